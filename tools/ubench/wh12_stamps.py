#!/usr/bin/env python3
"""Phase times of the 12-wave STFT kernel for uniform batches, per wave role (make -C speech_to_text_amd/csrc stamps):
STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_stamps.so python3 tools/ubench/wh12_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
B = 1024
N = int(os.environ.get("STAMP_CLIP_SAMPLES", "480000"))        # uniform clips of this length (480000 = 30 s)
dev = torch.device("cuda:0")
L = _lib.lib()
assert L.sttfe_set_whisper_uniform_kernel(1) == 0      # the diagnostic builds start with it off
nblk = 256
buf = torch.zeros(nblk * 12 * 20, dtype=torch.int64, device=dev)
L.sttfe_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sttfe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
fe = HipWhisperFeatureExtractor(80, device=dev)
pcm = synth_batch_device(1, B, N, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * N).to(dev)
ln = torch.full((B,), N, dtype=torch.int32, device=dev)
NF = int(os.environ.get("STAMP_N_FRAMES", "3000"))            # compact output layouts: frames per row
out = torch.empty((B, 80, NF), dtype=torch.float32, device=dev)
for _ in range(3):
    fe.extract(pcm, off, ln, out=out)
torch.cuda.synchronize()
raw = buf.view(nblk, 12, 20).double()
tiles = B * ((min(N // 160 + 1, NF) + 47) // 48) / nblk
print(f'{B} uniform clips of {N} samples, {NF} frames per output row, {tiles:.1f} tiles per workgroup')
names = ["s1: window, DFT, twiddles, slab", "wait for the panel (freed)", "panel rows + signal", "mel: wait for the panel", "mel units + keys", "loop top", "s2: DFT, partner, power", "tail"]
for role, sl in (("transform waves 0-7", slice(0, 8)), ("mel waves 8-11", slice(8, 12))):
    r = raw[:, sl, :]
    life = (r[:, :, 14] - r[:, :, 13]).mean()
    print(f"{role}: lifetime {life:.0f} cycles = {life / tiles:.0f} per tile")
    for i, n in enumerate(names):
        print(f"   {n:26s} {r[:, :, i].mean() / tiles:8.0f} cycles/tile")
