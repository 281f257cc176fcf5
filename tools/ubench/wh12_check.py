#!/usr/bin/env python3
"""Experiment: the 12-wave-per-workgroup Whisper kernel 1 (csrc/whisper_logmel12.inc, `make -C speech_to_text_amd/csrc wh12`)
against the shipped kernel, both inside tools/ubench/whisper12.so (STTFE_WH12=1 selects the prototype per call):
bit-identity of the features on uniform and ragged batches, then the kernel times on the headline workload.
Usage: STTFE_LIB_PATH=$PWD/tools/ubench/whisper12.so python3 tools/ubench/wh12_check.py"""
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
    os.environ["STTFE_WH12"] = "1" if proto else "0"
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
    os.environ["STTFE_WH12"] = str(proto)
    os.environ["STTFE_WH12_TILES"] = str(tiles or 21)
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
    t2, n2 = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_clamp_fill", ctypes.byref(t2), ctypes.byref(n2))
    L.sttfe_timing_enable(0)
    print(f"{'12-wave prototype' if proto else 'shipped kernel   '} tiles/run {tiles:2d}: kernel 1 {t.value / 8:.4f} ms   kernel 2 {t2.value / 8:.4f} ms", flush=True)
