#!/usr/bin/env python3
"""Diagnostic: phase shares of the wave-independent Whisper STFT kernel from a -DSTTFE_STAMPS build
(STTFE_LIB_PATH=.../libsttfe_stamps.so STTFE_WH_KERNEL=8|12)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device, common_voice_lengths
W = int(os.environ.get("STTFE_WH_KERNEL", "8"))
WL = os.environ.get("STAMP_WL", "full")
B = int(os.environ.get("STAMP_B", "256" if WL == "full" else "1024"))
N_MELS = int(os.environ.get("STAMP_MELS", "80"))
dev = torch.device("cuda:0")
L = _lib.lib()
nblk = 256
buf = torch.zeros(nblk * W * 20, dtype=torch.int64, device=dev)
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
raw = buf.view(nblk, W, 20).double()
busy = raw[:, :, 14] - raw[:, :, 13]
frames = np.minimum((np.minimum(lens, 480000) + 200 + 159) // 160, 3000)
n_wt = int(((frames + 11) // 12).sum())
per_wave = n_wt / (nblk * W)
names = ["FFT (both passes)", "panel rows", "PCM requests", "mel phase", "tile min + atomic", "item switch / loop top", "after last item"]
tot = raw[:, :, :7].sum(dim=2).mean()
print(f"{WL} {N_MELS} mels W={W}: ~{n_wt} wave tiles, {per_wave:.1f} per wave; wave lifetime mean {busy.mean():.0f} min {busy.min():.0f} max {busy.max():.0f}")
print(f"mean stamped cycles per wave {tot:.0f} ({tot/per_wave:.0f} per tile)")
for i, n in enumerate(names):
    print(f"  {n:24s} {raw[:, :, i].mean()/per_wave:8.0f} cycles/tile  {100*raw[:, :, i].mean()/tot:5.1f}%")
