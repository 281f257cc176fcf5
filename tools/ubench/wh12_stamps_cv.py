#!/usr/bin/env python3
"""Phase times of the 12-wave STFT kernel on a RAGGED batch (1024 Common-Voice-length clips, config 4's shape), per wave
role, beside tools/ubench/wh12_stamps.py's uniform batch (make -C speech_to_text_amd/csrc stamps):
STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_stamps.so python3 tools/ubench/wh12_stamps_cv.py [n_mels]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device
n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = 1024
dev = torch.device("cuda:0")
L = _lib.lib()
assert L.sttfe_set_whisper_uniform_kernel(1) == 0
nblk = 256
buf = torch.zeros(nblk * 12 * 20, dtype=torch.int64, device=dev)
L.sttfe_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sttfe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
fe = HipWhisperFeatureExtractor(n_mels, device=dev)
lens = common_voice_lengths(np.random.default_rng(3), B)
stride = int((lens.max() + 63) // 64 * 64)
pcm = synth_batch_device(1, B, stride, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * stride).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
out = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=dev)
for _ in range(3):
    fe.extract(pcm, off, ln, out=out)
torch.cuda.synchronize()
raw = buf.view(nblk, 12, 20).double()
tiles = float(np.sum((lens // 160 + 1 + 47) // 48)) / nblk
print(f"{B} clips, {tiles * nblk:.0f} tiles = {tiles:.1f} per workgroup, n_mels {n_mels}")
names = ["s1: window, DFT, twiddles, slab", "wait for the panel (freed)", "panel rows + signal", "mel: wait for the panel", "mel units + keys", "loop top (+ item switch)", "s2: DFT, partner, power", "tail"]
for role, sl in (("transform waves 0-7", slice(0, 8)), ("mel waves 8-11", slice(8, 12))):
    r = raw[:, sl, :]
    life = (r[:, :, 14] - r[:, :, 13])
    print(f"{role}: lifetime mean {life.mean():.0f} max {life.max():.0f} min {life.min():.0f} cycles = {life.mean() / tiles:.0f} per tile")
    for i, n in enumerate(names):
        print(f"   {n:26s} {r[:, :, i].mean() / tiles:8.0f} cycles/tile")
