#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 passes: a few launches of one front-end path.

    python3 tools/prof_step.py whisper [batch] [steps] [n_mels] [samples per clip] [frames per output row]
    python3 tools/prof_step.py kaldi   [batch] [steps]
    python3 tools/prof_step.py resample [batch] [steps]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

what = sys.argv[1] if len(sys.argv) > 1 else "whisper"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
if what == "whisper":
    n_mels = int(sys.argv[4]) if len(sys.argv) > 4 else 80
    n_samples = int(sys.argv[5]) if len(sys.argv) > 5 else 480000        # uniform clips of this length
    n_frames = int(sys.argv[6]) if len(sys.argv) > 6 else 3000            # (compact output layouts: fewer frames per row)
    fe = HipWhisperFeatureExtractor(n_mels, device=dev)
    pcm = synth_batch_device(1, batch, n_samples, dev).reshape(-1)
    off = (torch.arange(batch, dtype=torch.int64) * n_samples).to(dev)
    ln = torch.full((batch,), n_samples, dtype=torch.int32, device=dev)
    out = torch.empty((batch, n_mels, n_frames), dtype=torch.float32, device=dev)
    for _ in range(steps):
        fe.extract(pcm, off, ln, out=out)
elif what == "kaldi":
    fe = HipSpeech2TextFeatureExtractor(device=dev)
    rng = np.random.default_rng(102000)          # the lengths of bench.py --workload s2t on rank 0
    lens = rng.integers(32000, 160001, size=batch)
    n_max = int(lens.max())
    pcm = synth_batch_device(2, batch, n_max, dev).reshape(-1)
    off = (torch.arange(batch, dtype=torch.int64) * n_max).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    t_max = max(fe.num_frames(int(n)) for n in lens)
    for _ in range(steps):
        fe.extract(pcm, off, ln, t_max)
elif what == "resample":
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    from speech_to_text_amd.synth import common_voice_lengths
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
    lens = (common_voice_lengths(np.random.default_rng(103), batch) * 3).tolist()
    n_max = (max(lens) + 63) // 64 * 64
    pcm = synth_batch_device(3, batch, n_max, dev, sample_rate=48000).reshape(-1)
    off = (torch.arange(batch, dtype=torch.int64) * n_max).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    for _ in range(steps):
        pre.run(pcm, off, ln, lens, 1, 48000)
torch.cuda.synchronize()
print("done", what, batch, steps)
