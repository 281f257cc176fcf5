#!/usr/bin/env python3
"""Float64 Speech2Text path on random batches: the default kernels (wave-owned items, register-column CMVN) against the
split statistics + normalisation pair (STTFE_KALDI_SPLIT_CMVN, read per call), bit for bit -- 1 ... 1500 clips of 0 ... 30 s,
with and without SpecAugment masks, means / variances individually off.
Usage: python3 tools/ab/kaldi_stress.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
rng = np.random.default_rng(21)
fes = [HipSpeech2TextFeatureExtractor(device=dev), HipSpeech2TextFeatureExtractor(device=dev, normalize_vars=False),
       HipSpeech2TextFeatureExtractor(device=dev, normalize_means=False)]
bad = 0
for it in range(iters):
    B = int(rng.choice([1, 7, 64, 256, 1100, 1500]))
    hi = int(rng.choice([4000, 160000, 330000, 480000])) if B <= 256 else int(rng.choice([4000, 40000]))
    lens = rng.integers(0, hi + 1, B)
    n_max = int(max(lens.max(), 1))
    pcm = synth_batch_device(500 + it, B, n_max, dev).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    fe = fes[it % 3]
    t_max = max(max(fe.num_frames(int(n)) for n in lens), 1)
    tbl = None
    if it % 2:
        k = 3
        t = np.zeros((B, k, 3), dtype=np.int32)
        t[:, :, 0] = rng.integers(0, 2, size=(B, k))
        start = np.where(t[:, :, 0] == 1, rng.integers(0, t_max, size=(B, k)), rng.integers(0, 80, size=(B, k)))
        t[:, :, 1] = start
        t[:, :, 2] = start + rng.integers(0, 40, size=(B, k))
        tbl = torch.tensor(t, device=dev)
    a, am = fe.extract(pcm, off, ln, t_max, mask_table=tbl)
    os.environ["STTFE_KALDI_SPLIT_CMVN"] = "1"
    b, bm = fe.extract(pcm, off, ln, t_max, mask_table=tbl)
    del os.environ["STTFE_KALDI_SPLIT_CMVN"]
    same = bool(((a == b) | (a.isnan() & b.isnan())).all()) and bool((am == bm).all())
    if not same:
        bad += 1
        print(f"iteration {it}: B={B}, t_max={t_max}: {int(((a != b) & ~(a.isnan() & b.isnan())).sum())} cells differ", flush=True)
    if it % 8 == 7:
        print(f"{it + 1} batches compared, {bad} mismatches", flush=True)
print("batches:", iters, "mismatching:", bad)
sys.exit(1 if bad else 0)
