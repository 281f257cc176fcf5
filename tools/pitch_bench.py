"""Times sttfe_pitch_shift on a batch of CV-length clips (all active, steps cycling -2..2 without 0).
Usage: python tools/pitch_bench.py [batch] [seconds] [mode]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler
from speech_to_text_amd.preprocess import GpuAudioPreprocessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.5
mode = sys.argv[3] if len(sys.argv) > 3 else "as_constructed"
rng = np.random.default_rng(0)
n = int(secs * 16000)
clips = [(0.1 * rng.standard_normal(n)).astype(np.float32) for _ in range(B)]
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device="cuda:0")
p = pre.preprocess_batch(clips, 16000)
aug = GpuWaveformAugmenter(WaveformAugmentSampler(pitch_mode=mode))
steps = [(-2, -1, 1, 2)[b % 4] for b in range(B)]
for _ in range(3):
    aug.pitch_shift(p, steps)
torch.cuda.synchronize()
t = time.perf_counter()
K = 10
for _ in range(K):
    aug.pitch_shift(p, steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"pitch_shift {mode}: {B} clips x {secs} s: {dt * 1e3:.3f} ms per batch, {B * secs / dt / 1e6:.3f} M audio-s/s", flush=True)
