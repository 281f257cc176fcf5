#!/usr/bin/env python3
"""Race hunt for the 12-wave kernel on RAGGED batches (item plan in LDS, runs of one to a few tiles, masks, per-clip
scale): many launches on batches of Common-Voice-like and uniformly random lengths, 80 and 128 bins, different run lengths,
each compared bit for bit with the general kernel's output.
Usage: python3 tools/ubench/wh12_stress_ragged.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
L = _lib.lib()
fes = {80: HipWhisperFeatureExtractor(80, device=dev), 128: HipWhisperFeatureExtractor(128, device=dev)}
rng = np.random.default_rng(11)
bad = launches = 0
for it in range(iters):
    n_mels = 80 if it % 3 else 128
    B = int(rng.choice([37, 300, 512, 1024, 2500]))
    lens = common_voice_lengths(rng, B) if it % 2 else rng.integers(0, 480001, B)
    n_max = int(max(lens.max(), 1))
    pcm = synth_batch_device(300 + it, B, n_max, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    k = (0, 2, 16)[it % 3]
    tbl = None
    if k:
        t = np.zeros((B, k, 3), dtype=np.int32)
        t[:, :, 0] = rng.integers(0, 2, size=(B, k))
        start = np.where(t[:, :, 0] == 1, rng.integers(0, 3000, size=(B, k)), rng.integers(0, n_mels, size=(B, k)))
        t[:, :, 1] = start
        t[:, :, 2] = start + rng.integers(0, 80, size=(B, k))
        tbl = torch.tensor(t, device=dev)
    scale = torch.tensor(rng.uniform(0.5, 300.0, B).astype(np.float32), device=dev) if it % 4 == 3 else None
    L.sttfe_set_whisper_tiles_per_block((0, 1, 2, 3, 7)[it % 5])
    ref = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=dev)
    out = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=dev)
    L.sttfe_set_whisper_uniform_kernel(0)
    fes[n_mels].extract(pcm, off, ln, out=ref, mask_table=tbl, clip_scale=scale)
    L.sttfe_set_whisper_uniform_kernel(1)
    for rep in range(3):
        out.fill_(-9.0)
        fes[n_mels].extract(pcm, off, ln, out=out, mask_table=tbl, clip_scale=scale)
        launches += 1
        if not torch.equal(out, ref):
            bad += 1
            print(f"iteration {it} (B={B}, {n_mels} bins, {k} masks) repeat {rep}: {(out != ref).sum().item()} cells differ", flush=True)
    del ref, out, pcm
    if it % 10 == 9:
        print(f"{it + 1} batches, {launches} launches compared, {bad} mismatches", flush=True)
L.sttfe_set_whisper_tiles_per_block(0)
print("launches:", launches, "mismatching:", bad)
sys.exit(1 if bad else 0)
