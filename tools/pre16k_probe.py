#!/usr/bin/env python3
"""Timing of the 16 kHz preprocessing chain (peak -> normalise + trim -> finalize) on 1024 Common-Voice-length clips."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device
dev = torch.device("cuda:0")
B = 1024
lens = common_voice_lengths(np.random.default_rng(104000), B)
n_max = (int(lens.max()) + 63) // 64 * 64
pcm = synth_batch_device(104000, B, n_max, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
hl = [int(v) for v in lens]
nbytes = 4 * float(lens.sum())
for norm, trim in [(True, False), (True, True), (False, True)]:
    pre = GpuAudioPreprocessor(normalize=norm, trim_silence=trim, min_duration=0.0, max_duration=1e9, device=dev)
    for _ in range(5):
        pre.run(pcm.clone(), off, ln, hl, 1, 16000, sync_lengths=False)
    torch.cuda.synchronize()
    src = [pcm.clone() for _ in range(20)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        pre.run(src[i], off, ln, hl, 1, 16000, sync_lengths=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    import ctypes
    from speech_to_text_amd import _lib
    L = _lib.lib()
    _lib.check(L.sttfe_timing_enable(1024)); _lib.check(L.sttfe_timing_reset()); _lib.check(L.sttfe_timing_set_stride(1))
    for i in range(10):
        pre.run(src[i], off, ln, hl, 1, 16000, sync_lengths=False)
    torch.cuda.synchronize()
    ks = {}
    for name in ("resample_peak", "normalize_trim"):
        ms, k = ctypes.c_double(), ctypes.c_int()
        _lib.check(L.sttfe_timing_query(name.encode(), ctypes.byref(ms), ctypes.byref(k)))
        ks[name] = round(ms.value / max(k.value, 1), 4)
    _lib.check(L.sttfe_timing_enable(0))
    print(f"normalize={norm} trim={trim}: {dt * 1e3:.3f} ms per batch wall ({nbytes / 1e6:.0f} MB of PCM); kernels {ks}", flush=True)
