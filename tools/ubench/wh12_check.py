#!/usr/bin/env python3
"""The 12-wave STFT kernel for uniform batches (csrc/whisper_logmel12.inc) against the general kernel, switched per call with
sttfe_set_whisper_uniform_kernel: bit-identity of the features on a uniform and on a ragged batch (on the ragged one the
12-wave kernel returns at once), then both kernels' times on the headline workload.
Usage: python3 tools/ubench/wh12_check.py   (the commits of DESIGN.md section 4.1.1's experiments need
STTFE_LIB_PATH=tools/ubench/whisper12.so from `make wh12` there and select the kernel with STTFE_WH12=1)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

dev = torch.device("cuda:0")
L = _lib.lib()
fe = HipWhisperFeatureExtractor(80, device=dev)


def run(pcm, off, ln, B, proto):
    assert L.sttfe_set_whisper_uniform_kernel(1 if proto else 0) == 0
    out = torch.full((B, 80, 3000), -7.0, dtype=torch.float32, device=dev)
    fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    return out


ok = True
rng = np.random.default_rng(12)
for name, B, lens in (("uniform 30 s", 48, np.full(48, 480000)),
                      ("ragged", 96, np.concatenate([rng.integers(1, 480001, 80), [0, 1, 199, 200, 201, 23039, 23040, 23041, 46079, 464000, 464001, 479999, 479840, 479841, 240000, 7]]))):
    n_max = 480000
    pcm = synth_batch_device(5, B, n_max, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    a = run(pcm, off, ln, B, False)
    b = run(pcm, off, ln, B, True)
    same = torch.equal(a, b)
    ok &= same
    nd = int((a != b).sum())
    print(f"{name}: bit-identical = {same} ({nd} cells differ)", flush=True)
    if not same:
        idx = torch.nonzero(a != b)[:5].tolist()
        print("  first differences (clip, mel, frame):", idx, [(float(a[tuple(i)]), float(b[tuple(i)])) for i in idx])
if not ok:
    sys.exit(1)

B = 1024
pcm = synth_batch_device(1, B, 480000, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
ln = torch.full((B,), 480000, dtype=torch.int32, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
for proto, tiles in ((0, 0), (1, 21), (1, 7), (1, 9), (1, 63), (0, 0)):
    assert L.sttfe_set_whisper_uniform_kernel(proto) == 0
    assert L.sttfe_set_whisper_tiles_per_block(tiles) == 0
    L.sttfe_timing_enable(256)
    L.sttfe_timing_set_stride(1 << 30)
    for _ in range(40):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    L.sttfe_timing_reset()
    L.sttfe_timing_set_stride(1)
    for _ in range(8):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    t, n = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_stft_mel", ctypes.byref(t), ctypes.byref(n))
    t12, n12 = ctypes.c_double(), ctypes.c_int()
    if L.sttfe_timing_query(b"whisper_stft_mel12", ctypes.byref(t12), ctypes.byref(n12)) == 0:      # both STFT launches of a call
        t.value += t12.value
    t2, n2 = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_clamp_fill", ctypes.byref(t2), ctypes.byref(n2))
    L.sttfe_timing_enable(0)
    print(f"{'12-wave kernel   ' if proto else 'general kernel   '} tiles/run {tiles:2d}: kernel 1 {t.value / 8:.4f} ms   kernel 2 {t2.value / 8:.4f} ms", flush=True)
L.sttfe_set_whisper_tiles_per_block(0)
L.sttfe_set_whisper_uniform_kernel(1)
