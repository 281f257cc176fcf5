#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline path ("one end-to-end line including pinned-H2D").

Host clips (pinned staging through PinnedIngestRing, float32 or int16) -> H2D on the copy stream -> same-rate
copy/convert + Whisper-80 log-mel on the compute stream, the copy of batch k+1 under the kernels of batch k.
    python3 tools/e2e_h2d.py [batch] [batches] [fill_threads]
This is NOT bench.py's `value` (inputs resident in HBM); it is the number DESIGN.md quotes beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd.preprocess import GpuAudioPreprocessor, PinnedIngestRing
from speech_to_text_amd.processors import HipWhisperFeatureExtractor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
THREADS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
base = (rng.standard_normal(480000) * 0.1).astype(np.float32)
fe = HipWhisperFeatureExtractor(80, device=dev)
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
for name, clips in (("float32", [base] * B), ("int16", [np.round(base * 30000).astype(np.int16)] * B)):
    ring = PinnedIngestRing(dev, slots=3, fill_threads=THREADS)

    def run(n):
        tickets = [ring.submit(clips), ring.submit(clips)]
        for k in range(n):
            pcm, off, ln, lens, ch = ring.take(tickets.pop(0))
            p = pre.run(pcm, off, ln, lens, ch, 16000)
            fe.extract(p.pcm, p.offsets, p.lengths, out=out)
            if k + 2 < n:
                tickets.append(ring.submit(clips))
        torch.cuda.synchronize()

    run(4)
    t0 = time.perf_counter()
    run(N)
    dt = time.perf_counter() - t0
    audio = B * N * 30.0
    gb = B * N * 480000 * (4 if name == "float32" else 2) / 1e9
    print(f"{name:8s} B={B} x {N} batches, {THREADS} fill thread(s): {audio / dt / 1e6:.3f} M audio-s/s end to end, {gb / dt:.1f} GB/s over PCIe "
          f"(includes the host memcpy into the pinned slot)", flush=True)
