#!/usr/bin/env python3
"""Diagnostic: how the launch time of the Whisper STFT kernel evolves over a long back-to-back run (clock ramp-up,
then the sustained level).  python3 tools/clock_profile.py [launches]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = 1024
dev = torch.device("cuda:0")
L = _lib.lib()
fe = HipWhisperFeatureExtractor(80, device=dev)
pcm = synth_batch_device(1, B, 480000, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
ln = torch.full((B,), 480000, dtype=torch.int32, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
done = 0
for chunk in [10, 10, 10, 20, 50, 100, 200, 300, 400, 400]:
    if done >= n:
        break
    L.sttfe_timing_enable(1024)
    for _ in range(chunk):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    t, k = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_stft_mel", ctypes.byref(t), ctypes.byref(k))
    L.sttfe_timing_enable(0)
    done += chunk
    print(f"launches {done - chunk:5d}..{done:5d}  t = {time.perf_counter() - t0:6.3f} s  avg {t.value / max(k.value, 1):.4f} ms", flush=True)
