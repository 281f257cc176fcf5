#!/usr/bin/env python3
"""Diagnostic: phase shares of the Whisper STFT kernel from a -DSTTFE_STAMPS build (STTFE_LIB_PATH)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
import numpy as np
from speech_to_text_amd.synth import common_voice_lengths
WL = os.environ.get("STAMP_WL", "full")          # full: 256 x 30 s clips; cv: 1024 Common-Voice-length clips
B = int(os.environ.get("STAMP_B", "256" if WL == "full" else "1024"))
N_MELS = int(os.environ.get("STAMP_MELS", "80"))
dev = torch.device("cuda:0")
L = _lib.lib()
tpb = int(os.environ.get("AB_TPB", "9" if WL == "full" else "0"))
L.sttfe_set_whisper_tiles_per_block(tpb)
nblk = 512                      # persistent grid: two workgroups per CU
buf = torch.zeros(nblk * 4 * 20, dtype=torch.int64, device=dev)
L.sttfe_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sttfe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
fe = HipWhisperFeatureExtractor(N_MELS, device=dev)
lens = np.full(B, 480000) if WL == "full" else common_voice_lengths(np.random.default_rng(104000), B)
n_max = (int(lens.max()) + 63) // 64 * 64
pcm = synth_batch_device(1, B, n_max, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
out = torch.empty((B, N_MELS, 3000), dtype=torch.float32, device=dev)
for _ in range(3):
    fe.extract(pcm, off, ln, out=out)
torch.cuda.synchronize()
raw = buf.view(nblk, 4, 20)
s = torch.cat([raw[:, :, :13], raw[:, :, 16:18]], dim=2).double()
n_tiles_total = int(np.minimum((np.minimum(lens, 480000) + 200 + 7679) // 7680, 63).sum())
tiles_per_wg = n_tiles_total / nblk     # tiles each workgroup walks on average
tot = s.sum(dim=2).mean()
busy = (raw[:, :, 14] - raw[:, :, 13]).double()
print(f"{WL} {N_MELS} mels: {n_tiles_total} tiles; per-wave lifetime (cycles; the counters of different XCDs are not aligned, so no kernel-wide span) mean {busy.mean():.0f} min {busy.min():.0f} max {busy.max():.0f}")
names = ["s1(p0)+exchange+s1 math(p1)", "epilogue", "exchange(p1)+s2(p0)+fetch", "s2(p1)+fetch+sep(p0)", "sep(p1)", "panel rows+PCM requests", "barrier B2", "mel phase (static) / tile min", "barrier B3", "mfma loops", "kernel start-up", "item top -> first tile", "after the last item", "item tail + hand-over barrier", "item decode"]
print(f"mean cycles per wave {tot:.0f} ({tot/tiles_per_wg:.0f} per tile)")
for i, n in enumerate(names[:15]):
    print(f"  {n:18s} {s[:, :, i].mean()/tiles_per_wg:8.0f} cycles/tile  {100*s[:, :, i].mean()/tot:5.1f}%")
