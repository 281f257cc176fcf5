"""Pitch shift, CPU side: the oracle's sanity anchors and the host-only plan function.

torchaudio is absent and the reference holds no vectors for this row: parity with a real install is UNPINNED.
What can be pinned here: the oracle's stretch runs on the real torch.stft / torch.istft; a stationary tone keeps
its frequency in "as_constructed" mode (the reference's transform never resamples: tempo change only) and moves by
2^(n/12) in "functional" mode; the per-sample sinc resampler equals the table resampler of row R."""
import ctypes

import numpy as np
import pytest

from oracle import pitch_shift as P
from oracle.resample import resample


def tone(n, f=440.0):
    return (0.3 * np.sin(2 * np.pi * f * np.arange(n) / 16000.0)).astype(np.float32)


def dominant_hz(v):
    seg = v[4000:20000].astype(np.float64) * np.hanning(16000)
    return np.argmax(np.abs(np.fft.rfft(seg))) * 16000.0 / 16000


def test_zero_steps_is_stft_istft_identity():
    x = tone(20000) + 0.1 * np.random.default_rng(0).standard_normal(20000).astype(np.float32)
    for mode in ("as_constructed", "functional"):
        assert np.abs(P.pitch_shift(x, 0, 16000, mode) - x).max() <= 2e-6


@pytest.mark.parametrize("n_steps", [-2, -1, 1, 2])
def test_modes_on_a_tone(n_steps):
    x = tone(32000)
    kept = P.pitch_shift(x, n_steps, 16000, "as_constructed")
    moved = P.pitch_shift(x, n_steps, 16000, "functional")
    assert kept.shape == moved.shape == x.shape and kept.dtype == np.float32
    assert abs(dominant_hz(kept) - 440.0) <= 1.0
    assert abs(dominant_hz(moved) - 440.0 * 2 ** (n_steps / 12)) <= 1.5
    if n_steps > 0:                                  # rate < 1: longer, cropped; rate > 1: shorter, zero-padded
        assert np.abs(kept[-2000:]).max() > 0.1
    else:
        pad = 32000 - int(round(32000 / P.stretch_rate(n_steps)))
        assert np.all(kept[-pad:] == 0.0) and pad > 1000


def test_per_sample_sinc_equals_table_resampler():
    x = np.random.default_rng(1).standard_normal(6000).astype(np.float32)
    for src, dst in [(3, 2), (11, 10), (9, 10), (160, 147)]:
        a, b = resample(x, src, dst), P.sinc_resample_any(x, src, dst)
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-6


def _plan(lengths, steps, mode=0, rate=16000):
    from speech_to_text_amd import _lib
    L = _lib.lib()
    lens = np.asarray(lengths, np.int32)
    st = np.asarray(steps, np.int64).astype(np.int32)
    layout = np.full((len(lengths), 8), -1, np.int64)
    nbytes = ctypes.c_size_t(0)
    rc = L.sttfe_pitch_shift_plan(lens.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p), len(lengths),
                                  rate, mode, layout.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nbytes))
    return rc, layout, nbytes.value


def test_plan_layout_and_errors():
    from speech_to_text_amd.augment import PITCH_NONE
    rc, lay, nbytes = _plan([16000, 5000, 23457], [2, PITCH_NONE, -1], mode=1)
    assert rc == 0 and np.all(lay[1] == 0)
    for b, (n, s) in {0: (16000, 2), 2: (23457, -1)}.items():
        rate = P.stretch_rate(s)
        T = 1 + n // 128
        assert lay[b, 4] & 0xffffffff == T and lay[b, 4] >> 32 == int(np.ceil(T / rate))
        assert lay[b, 5] & 0xffffffff == n and lay[b, 5] >> 32 == int(round(n / rate))
        assert np.frombuffer(lay[b, 6].tobytes(), np.float64)[0] == rate
        orig = int(16000 / rate)
        g = np.gcd(orig, 16000)
        assert lay[b, 7] & 0xffffffff == orig // g and lay[b, 7] >> 32 == 16000 // g
        assert np.all(lay[b, :4] % 64 == 0) and lay[b, 0] < lay[b, 1] < lay[b, 2] == lay[b, 3]
    assert lay[2, 0] > lay[0, 3] and nbytes > 4 * lay[2, 3]
    assert _plan([16000], [2], mode=0)[2] < _plan([16000], [2], mode=1)[2]        # no waveform buffer without resample
    assert _plan([256], [1])[0] == -1                                             # reflect padding impossible
    assert _plan([257], [1])[0] == 0
    assert _plan([16000], [1], mode=7)[0] == -1 and _plan([16000], [49])[0] == -5
    assert _plan([], [])[0] == 0


def test_sampler_pitch_mode_argument():
    from speech_to_text_amd.augment import WaveformAugmentSampler
    with pytest.raises(ValueError):
        WaveformAugmentSampler(pitch_mode="shift")
    assert WaveformAugmentSampler(pitch_mode="off").pitch_mode == "off"
