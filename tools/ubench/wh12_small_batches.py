#!/usr/bin/env python3
"""Latency of the Whisper path (both STFT launches + clamp / fill, HIP events around the call) on small uniform batches of
30 s clips, 12-wave kernel on / off.  Usage: python3 tools/ubench/wh12_small_batches.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

dev = torch.device("cuda:0")
L = _lib.lib()
N_MELS = int(sys.argv[1]) if len(sys.argv) > 1 else 80
fe = HipWhisperFeatureExtractor(N_MELS, device=dev)
print(f"{N_MELS} mel bins")
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    pcm = synth_batch_device(3, B, 480000, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
    ln = torch.full((B,), 480000, dtype=torch.int32, device=dev)
    out = torch.empty((B, N_MELS, 3000), dtype=torch.float32, device=dev)
    res = []
    for on in (0, 1):
        L.sttfe_set_whisper_uniform_kernel(on)
        for _ in range(20):
            fe.extract(pcm, off, ln, out=out)
        ts = []
        for _ in range(30):
            torch.cuda._sleep(200_000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fe.extract(pcm, off, ln, out=out)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(float(np.median(ts)))
    print(f"B = {B:4d}: general kernel {res[0] * 1e3:8.1f} us   12-wave kernel {res[1] * 1e3:8.1f} us   ({100 * (res[1] / res[0] - 1):+.0f} %)", flush=True)
L.sttfe_set_whisper_uniform_kernel(1)
