#!/usr/bin/env python3
"""Race hunt for the 12-wave kernel's LDS-counter hand-over: many launches on large uniform batches (different seeds,
run lengths and near-uniform length mixes), each compared bit for bit with the general kernel's output.
Usage: python3 tools/ubench/wh12_stress.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
L = _lib.lib()
fe = HipWhisperFeatureExtractor(80, device=dev)
B = 768
rng = np.random.default_rng(7)
bad = 0
ref = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
out = torch.empty((B, 80, 3000), dtype=torch.float32, device=dev)
for it in range(iters):
    pcm = synth_batch_device(100 + it, B, 480000, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
    lens = np.full(B, 480000) if it % 2 == 0 else rng.integers(480000 - 7000, 480001, B)     # within a tile of 30 s: uniform
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    L.sttfe_set_whisper_tiles_per_block((0, 1, 5, 21, 63)[it % 5])
    L.sttfe_set_whisper_uniform_kernel(0)
    fe.extract(pcm, off, ln, out=ref)
    L.sttfe_set_whisper_uniform_kernel(1)
    for rep in range(3):
        out.fill_(-9.0)
        fe.extract(pcm, off, ln, out=out)
        if not torch.equal(out, ref):
            bad += 1
            print(f"iteration {it} repeat {rep}: {(out != ref).sum().item()} cells differ", flush=True)
    if it % 10 == 9:
        print(f"{it + 1} batches x 3 launches compared, {bad} mismatches", flush=True)
L.sttfe_set_whisper_tiles_per_block(0)
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)
