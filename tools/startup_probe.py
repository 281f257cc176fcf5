#!/usr/bin/env python3
"""Fixed cost of a whisper_stft_mel launch: uniform clips of k tiles each, k = 1, 3, 6, 12, 24 (512 clips, 80 bins)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
dev = torch.device("cuda:0")
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fe = HipWhisperFeatureExtractor(80, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
for tiles in (1, 3, 6, 12, 24, 48):
    n = tiles * 7680 - 200                       # exactly `tiles` computed tiles
    pcm = synth_batch_device(7, B, n, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * n).to(dev)
    ln = torch.full((B,), n, dtype=torch.int32, device=dev)
    L.sttfe_timing_enable(256)
    L.sttfe_timing_set_stride(1 << 30)
    for _ in range(40):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    L.sttfe_timing_reset()
    L.sttfe_timing_set_stride(1)
    for _ in range(10):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    t, k = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_stft_mel", ctypes.byref(t), ctypes.byref(k))
    L.sttfe_timing_enable(0)
    print(f"B={B} tiles/clip={tiles:3d}  stft_mel {t.value / k.value * 1e3:8.1f} us  = {t.value / k.value * 1e6 / (B * tiles):6.2f} ns per tile", flush=True)
