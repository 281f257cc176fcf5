"""Self-consistency checks for oracle rows R / P2 / P3 / S (torchaudio absent: parity UNPINNED,
see oracle/__init__.py and DESIGN.md)."""
import random

import numpy as np
import torch

from conftest import load_golden
from oracle import resample as R
from oracle import specaugment as S


def test_kernel_shapes_and_dc_gain():
    k, width, orig, new = R.sinc_resample_kernel(48000, 16000)
    assert (width, orig, new) == (19, 3, 1) and k.shape == (1, 41)
    assert abs(float(k.sum()) - 1.0) < 2e-3
    k, width, orig, new = R.sinc_resample_kernel(32000, 16000)
    assert (width, orig, new) == (13, 2, 1) and k.shape == (1, 28)
    k, width, orig, new = R.sinc_resample_kernel(44100, 16000)
    assert (orig, new) == (441, 160) and k.shape == (160, 2 * width + 441)


def test_tone_gain_and_length():
    for sr in (48000, 32000):
        n = sr // 2 + 5
        t = np.arange(n) / sr
        x = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
        y = R.resample(x, sr, 16000)
        assert y.shape[0] == int(np.ceil(16000 * n / sr))
        ref = 0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(y.shape[0]) / 16000.0)
        mid = slice(200, y.shape[0] - 200)
        assert np.max(np.abs(y[mid] - ref[mid])) < 5e-3


def test_selfgen_regression():
    z = load_golden("selfgen_resample_masks.npz")
    for sr in (48000, 32000, 44100):
        got = R.resample(z[f"rs_{sr}__pcm"], sr, 16000)
        assert np.max(np.abs(got - z[f"rs_{sr}__out"])) < 1e-6
    torch.manual_seed(104)
    random.seed(104)
    rows = []
    for i in range(64):
        rows += [(i, a, s, e) for a, s, e in S.draw_clip_masks(128, 3000)]
    assert np.array_equal(np.asarray(rows, dtype=np.int32).reshape(-1, 4), z["masks_seed104_128x3000"])


def test_mask_bounds():
    torch.manual_seed(7)
    random.seed(7)
    n_aug = 0
    for _ in range(400):
        masks = S.draw_clip_masks(80, 3000)
        assert len(masks) in (0, 4)
        n_aug += bool(masks)
        for axis, s, e in masks:
            size, param = (3000, 20) if axis == S.AXIS_TIME else (80, 30)
            assert 0 <= s <= e <= size and e - s < param
        if masks:
            assert [m[0] for m in masks] == [S.AXIS_TIME, S.AXIS_TIME, S.AXIS_FREQ, S.AXIS_FREQ]
    assert 100 < n_aug < 220          # gate p = 0.4


def test_normalize_trim_gate():
    x = np.array([0, 0.001, 0.2, -0.5, 0.1, 0.0009, 0], dtype=np.float32)
    y = R.peak_normalize(x)
    assert y.dtype == np.float32 and abs(float(np.abs(y).max()) - 1.0) < 1e-6
    assert R.trim_bounds(y, 0.01) == (2, 5)
    assert R.trim_bounds(np.zeros(4, np.float32)) is None
    out = R.preprocess(x[None, :], 16000, trim_silence=True, min_duration=0.0, max_duration=1.0)
    assert out["audio"].shape == (3,) and abs(out["duration"] - 3 / 16000) < 1e-12
    assert R.preprocess(x[None, :], 16000, min_duration=1.0, max_duration=30.0) is None
    st = np.stack([x, -x])
    assert np.all(R.to_mono(st) == 0)
