#!/usr/bin/env python3
"""Diagnostic: phase shares of the 3:1 decimator's tile loop from a -DSTTFE_STAMPS build
(make -C speech_to_text_amd/csrc stamps; STTFE_LIB_PATH=.../libsttfe_stamps.so python tools/stamps_decimate.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device
dev = torch.device("cuda:0")
L = _lib.lib()
B = 512
nblk = 5 * 256
buf = torch.zeros(nblk * 4 * 10, dtype=torch.int64, device=dev)
L.sttfe_debug_set_rs_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sttfe_debug_set_rs_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
lens = common_voice_lengths(np.random.default_rng(103000), B) * 3
n_max = (int(lens.max()) + 63) // 64 * 64
pcm = synth_batch_device(103000, B, n_max, dev, sample_rate=48000).reshape(-1)
if os.environ.get("STAMP_S16"):
    pcm = torch.clamp(torch.round(pcm * 32767.0), -32768, 32767).to(torch.int16)
off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
for _ in range(3):
    buf.zero_()
    pre.run(pcm, off, ln, [int(v) for v in lens], 1, 48000, sync_lengths=False)
torch.cuda.synchronize()
raw = buf.view(nblk, 4, 10).double()
live = raw[:, :, 9] > 0
tiles = int(((lens + 2) // 3 + 1023) // 1024).sum() if False else int((((lens + 2) // 3 + 1023) // 1024).sum())
per_wg = tiles / max(int(live[:, 0].sum()), 1)
names = ["deposit (cvt + LDS writes)", "barrier after deposit", "request (addresses + loads issued)", "row reads + FMAs", "store + peak",
         "barrier at the tile top", "loop overhead / item switch", "start-up (prefix, first request)", "wait for the tile's samples (vmcnt 0)", "lifetime"]
life = raw[:, :, 9][live].mean()
print(f"{int(live[:, 0].sum())} workgroups, {tiles} tiles, {per_wg:.1f} tiles per workgroup, mean wave lifetime {life:.0f} cycles (100 MHz counter)")
for i, n in enumerate(names):
    v = raw[:, :, i][live].mean()
    print(f"  {n:40s} {v:9.0f}  {v / per_wg:8.1f} per tile  {100 * v / life:5.1f} %")
