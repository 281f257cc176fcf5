#!/usr/bin/env python3
"""Timing of the 48 kHz decimator on stereo (planar) input, 512 Common-Voice-length clips."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 512
lens = common_voice_lengths(np.random.default_rng(103000), B) * 3
n_max = (int(lens.max()) + 63) // 64 * 64
mono = synth_batch_device(103000, B, n_max, dev, sample_rate=48000)
# planar: clip b = C rows of lens[b] samples, packed back to back with a 64-sample aligned clip stride
stride = C * n_max
pcm = torch.zeros(B * stride, dtype=torch.float32, device=dev)
for b in range(B):
    n = int(lens[b])
    for c in range(C):
        pcm[b * stride + c * n: b * stride + (c + 1) * n] = mono[b, :n] * (1.0 if c == 0 else 0.5)
off = (torch.arange(B, dtype=torch.int64) * stride).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
hl = [int(v) for v in lens]
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
L = _lib.lib()
for _ in range(5):
    pre.run(pcm, off, ln, hl, C, 48000, sync_lengths=False)
torch.cuda.synchronize()
_lib.check(L.sttfe_timing_enable(1024)); _lib.check(L.sttfe_timing_reset()); _lib.check(L.sttfe_timing_set_stride(1))
for _ in range(10):
    pre.run(pcm, off, ln, hl, C, 48000, sync_lengths=False)
torch.cuda.synchronize()
ms, k = ctypes.c_double(), ctypes.c_int()
_lib.check(L.sttfe_timing_query(b"resample_peak", ctypes.byref(ms), ctypes.byref(k)))
t = ms.value / max(k.value, 1)
nin = 4 * C * float(lens.sum()); nout = 4 * float(lens.sum()) / 3
print(f"{C} channel(s) 48000 -> 16000: {t:.4f} ms per launch, {nin / 1e6:.0f} MB in + {nout / 1e6:.0f} MB out = {(nin + nout) / t / 1e9:.2f} TB/s")
