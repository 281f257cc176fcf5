#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run once, in the build container).

What the fixtures are
---------------------
The reference (Alexey511/speech_to_text) holds NO numerical golden vectors for its
feature path, and its arithmetic lives in the pip dependency
``transformers`` (reference requirements.txt:11).  ``transformers`` IS importable in the
build container, so the goldens for rows W and K are the outputs of

    transformers.WhisperFeatureExtractor()            (80 mel)
    transformers.WhisperFeatureExtractor(feature_size=128)
    transformers.Speech2TextFeatureExtractor()

constructed from default arguments (no Hub fetch) and called exactly the way the reference
calls them (src/data.py:338-355, inference.py:148-152) on seeded synthetic PCM.
Version note: installed transformers is 5.15.0, the reference pins 4.57.0.

Rows R (resample) and S (SpecAugment masks) have no importable oracle (torchaudio is not
installed).  ``selfgen_*.npz`` hold outputs of the repo's own restatement (oracle/) and
only guard against regressions / RNG drift -- they pin nothing external.

Only inputs and expected outputs are stored (data, not code).  Whisper outputs for short
clips are stored as the first ``keep`` frames plus the constant tail value.
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from speech_to_text_amd.synth import synth_clip  # noqa: E402


def whisper_cases():
    rng = np.random.default_rng(1001)
    cases = {
        "c0_0p3s": synth_clip(rng, 4800, zero_block=False),
        "c1_1s": synth_clip(rng, 16000, zero_block=False),
        "c2_2p5s_odd": synth_clip(rng, 40001, zero_block=False),
        "c3_zero": np.zeros(8000, dtype=np.float32),
        "c4_trailing_zeros": np.concatenate([synth_clip(rng, 12000, zero_block=False),
                                             np.zeros(9000, dtype=np.float32)]),
        "c5_zero_block": synth_clip(rng, 32000, zero_block=True),
        "c6_tiny": synth_clip(rng, 37, zero_block=False),
    }
    return cases


def gen_whisper():
    from transformers import WhisperFeatureExtractor
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor() if n_mels == 80 else WhisperFeatureExtractor(feature_size=128)
        store = {}
        for name, clip in whisper_cases().items():
            out = fe(clip, sampling_rate=16000, return_tensors="pt").input_features[0].numpy()
            assert out.shape == (n_mels, 3000) and out.dtype == np.float32
            keep = min(3000, clip.shape[0] // 160 + 4)
            tail = out[:, keep:]
            assert tail.size == 0 or np.all(tail == tail.flat[0]), name
            store[f"{name}__pcm"] = clip
            store[f"{name}__head"] = out[:, :keep]
            store[f"{name}__tail"] = np.float32(tail.flat[0] if tail.size else 0.0)
        # a batch call (list of arrays), as TorchWhisperFeatureExtractor-style callers use it
        clips = [whisper_cases()[k] for k in ("c0_0p3s", "c2_2p5s_odd", "c3_zero")]
        outb = fe(clips, sampling_rate=16000, return_tensors="pt").input_features.numpy()
        for i, k in enumerate(("c0_0p3s", "c2_2p5s_odd", "c3_zero")):
            keep = store[f"{k}__head"].shape[1]
            assert np.array_equal(outb[i][:, :keep], store[f"{k}__head"]), "batched != single"
        np.savez_compressed(os.path.join(HERE, f"whisper_{n_mels}.npz"), **store)
    # one full-length case (31 s -> truncated to 30 s), 80 mel only, full output kept
    fe = WhisperFeatureExtractor()
    rng = np.random.default_rng(1002)
    clip = synth_clip(rng, 31 * 16000, zero_block=True)
    out = fe(clip, sampling_rate=16000, return_tensors="pt").input_features[0].numpy()
    np.savez_compressed(os.path.join(HERE, "whisper_80_full31s.npz"), seed=np.int64(1002),
                        n_samples=np.int64(31 * 16000), out=out)


def gen_s2t():
    from transformers import Speech2TextFeatureExtractor
    fe = Speech2TextFeatureExtractor()
    rng = np.random.default_rng(2001)
    cases = {
        "k0_0p3s": synth_clip(rng, 4800, zero_block=False),
        "k1_1s": synth_clip(rng, 16000, zero_block=False),
        "k2_2p5s_odd": synth_clip(rng, 40001, zero_block=False),
        "k3_zero_block": synth_clip(rng, 48000, zero_block=True),
        "k4_min": synth_clip(rng, 400 + 160 * 2, zero_block=False),     # exactly 3 frames
    }
    store = {}
    for name, clip in cases.items():
        o = fe(clip, sampling_rate=16000, return_tensors="pt")
        store[f"{name}__pcm"] = clip
        store[f"{name}__feat"] = o.input_features[0].numpy()
        store[f"{name}__mask"] = o.attention_mask[0].numpy()
    # truncation as the reference does it (src/data.py:349-355) with a small max_length
    o = fe(cases["k2_2p5s_odd"], sampling_rate=16000, return_tensors="pt", truncation=True, max_length=100)
    store["k2_trunc100__feat"] = o.input_features[0].numpy()
    store["k2_trunc100__mask"] = o.attention_mask[0].numpy()
    np.savez_compressed(os.path.join(HERE, "s2t_80.npz"), **store)


def gen_s2t_cmvn_flags():
    """normalize_means / normalize_vars individually off (HF:...speech_to_text.py:141-163), round 3."""
    from transformers import Speech2TextFeatureExtractor
    rng = np.random.default_rng(2002)
    cases = {"c0_0p7s": synth_clip(rng, 11200, zero_block=False), "c1_2s_zero_block": synth_clip(rng, 32000, zero_block=True)}
    store = {}
    for name, clip in cases.items():
        store[f"{name}__pcm"] = clip
        for tag, means, varis in (("mean_only", True, False), ("var_only", False, True)):
            fe = Speech2TextFeatureExtractor(normalize_means=means, normalize_vars=varis)
            store[f"{name}__{tag}"] = fe(clip, sampling_rate=16000, return_tensors="pt").input_features[0].numpy()
    np.savez_compressed(os.path.join(HERE, "s2t_80_cmvn_flags.npz"), **store)


def gen_selfgen():
    import torch
    from oracle import resample as R
    from oracle import specaugment as S
    rng = np.random.default_rng(3001)
    store = {}
    for sr in (48000, 32000, 44100):
        n = int(0.25 * sr) + 7
        clip = synth_clip(rng, n, sample_rate=sr, zero_block=False)
        store[f"rs_{sr}__pcm"] = clip
        store[f"rs_{sr}__out"] = R.resample(clip, sr, 16000)
    tuples = []
    torch.manual_seed(104)
    random.seed(104)
    for i in range(64):
        masks = S.draw_clip_masks(128, 3000)
        tuples.append(np.asarray([(i, a, s, e) for a, s, e in masks], dtype=np.int32).reshape(-1, 4))
    store["masks_seed104_128x3000"] = np.concatenate(tuples, axis=0)
    np.savez_compressed(os.path.join(HERE, "selfgen_resample_masks.npz"), **store)


if __name__ == "__main__":
    gen_whisper()
    gen_s2t()
    gen_s2t_cmvn_flags()
    gen_selfgen()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
