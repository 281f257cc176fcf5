#!/usr/bin/env python3
"""Diagnostic: phase shares of the Whisper STFT kernel from a -DSTTFE_STAMPS build (STTFE_LIB_PATH)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
B = 256
dev = torch.device("cuda:0")
L = _lib.lib()
tpb = int(os.environ.get("AB_TPB", "9"))
L.sttfe_set_whisper_tiles_per_block(tpb)
nblk = 512                      # persistent grid: two workgroups per CU
buf = torch.zeros(nblk * 4 * 10, dtype=torch.int64, device=dev)
L.sttfe_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sttfe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
fe = HipWhisperFeatureExtractor(80, device=dev)
pcm = synth_batch_device(1, B, 480000, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
ln = torch.full((B,), 480000, dtype=torch.int32, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
for _ in range(3):
    fe.extract(pcm, off, ln, out=out)
torch.cuda.synchronize()
s = buf.view(nblk, 4, 10).double()
tiles_per_wg = 63 * B / nblk     # full-length clips: tiles each workgroup walks on average
tot = s.sum(dim=2).mean()
names = ["s1(p0)+exchange+s1 math(p1)", "epilogue", "exchange(p1)+s2(p0)+fetch", "s2(p1)+fetch+sep(p0)", "sep(p1)", "panel rows+PCM requests", "barrier B2", "tile min", "barrier B3", "mfma loops"]
print(f"mean cycles per wave {tot:.0f} ({tot/tiles_per_wg:.0f} per tile)")
for i, n in enumerate(names[:10]):
    print(f"  {n:18s} {s[:, :, i].mean()/tiles_per_wg:8.0f} cycles/tile  {100*s[:, :, i].mean()/tot:5.1f}%")
