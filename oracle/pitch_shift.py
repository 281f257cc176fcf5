"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the pitch shift of
the reference's waveform augmentation (``src/data.py:133-136,163-167``: ``torchaudio.transforms.PitchShift(
sample_rate, n_steps=0)`` whose ``n_steps`` attribute is overwritten with ``random.randint(-2, 2)`` per clip).

torchaudio (pinned 2.5.1 in the reference's requirements.txt:3) is not installed here and the reference holds
no golden vectors for this row: **parity unpinned**.  The algorithm is restated from the torchaudio 2.5.1
text (``functional.pitch_shift`` = ``_stretch_waveform`` -> sinc resample -> ``_fix_waveform_shape``;
``functional.phase_vocoder``); the torch pieces it is made of (``torch.stft``, ``torch.istft``, ``torch.cumsum``,
``torch.polar``, ``torch.linspace``) are the real ones.

Two modes, because the reference's way of using the transform changes what it computes:

* ``"as_constructed"`` -- what ``src/data.py`` runs.  The transform fixes ``orig_freq = int(sample_rate / rate)``
  in its constructor, where n_steps is 0, so ``orig_freq == sample_rate`` and ``forward`` skips the resample:
  the clip is time-stretched by 2**(-n_steps/12) with the phase vocoder and then cropped / zero-padded back
  to its length (a tempo change, not a pitch change).  (With ``orig_freq == sample_rate`` the transform's
  lazily initialised resample kernel is never materialised either; whether torch's lazy-module hook then
  rejects the first call cannot be checked here without torchaudio.)
* ``"functional"`` -- ``torchaudio.functional.pitch_shift(x, sr, n_steps)``: the same stretch followed by the
  sinc resample from ``int(sr / rate)`` to ``sr`` (hann-windowed sinc, width 6, rolloff 0.99).
"""
from __future__ import annotations

import math
from math import gcd

import numpy as np
import torch

N_FFT = 512
HOP = N_FFT // 4


def stretch_rate(n_steps: int, bins_per_octave: int = 12) -> float:
    return 2.0 ** (-float(n_steps) / bins_per_octave)


def phase_vocoder(spec: torch.Tensor, rate: float, phase_advance: torch.Tensor) -> torch.Tensor:
    """(freq, T) complex64 -> (freq, ceil(T / rate)) complex64: magnitudes interpolated between neighbouring
    frames, phases accumulated from the wrapped frame-to-frame phase differences."""
    if rate == 1.0:
        return spec
    T = spec.shape[-1]
    steps = torch.arange(0, T, rate, dtype=torch.float32)
    alphas = steps % 1.0
    first_phase = spec[..., :1].angle()
    padded = torch.nn.functional.pad(spec, [0, 2])
    lo = padded.index_select(-1, steps.long())
    hi = padded.index_select(-1, (steps + 1).long())
    dphi = hi.angle() - lo.angle() - phase_advance
    dphi = dphi - 2 * math.pi * torch.round(dphi / (2 * math.pi))
    dphi = dphi + phase_advance
    acc = torch.cumsum(torch.cat([first_phase, dphi[..., :-1]], dim=-1), -1)
    mag = alphas * hi.abs() + (1 - alphas) * lo.abs()
    return torch.polar(mag, acc)


def stretch_waveform(x: torch.Tensor, n_steps: int) -> torch.Tensor:
    """(n,) float32 -> (round(n / rate),) float32."""
    n = x.shape[-1]
    rate = stretch_rate(n_steps)
    window = torch.hann_window(N_FFT)
    spec = torch.stft(x, n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    advance = torch.linspace(0, math.pi * HOP, spec.shape[-2])[..., None]
    out = phase_vocoder(spec, rate, advance)
    return torch.istft(out, n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window=window,
                       length=int(round(n / rate)))


def sinc_resample_any(x: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """torchaudio's sinc_interp_hann resample (width 6, rolloff 0.99) with the taps evaluated per output
    sample instead of from the (new x taps) table, which has 16000 x 17000 entries for these ratios."""
    g = gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return x.astype(np.float32)
    base = min(orig, new) * 0.99
    width = int(math.ceil(6 * orig / base))
    n = x.shape[0]
    n_out = int(math.ceil(new * n / orig))
    j = np.arange(n_out, dtype=np.int64)
    blk, ph = j // new, j % new
    k = np.arange(2 * width + orig, dtype=np.int64)                         # tap index within the table row
    # only taps with |t| < 6 are non-zero: k in (width - 6 orig/base + ph*orig/new ...); evaluate a band
    centre = width + (ph * orig) // new                                     # tap nearest t = 0
    half = width + 1
    out = np.zeros(n_out, dtype=np.float64)
    xp = np.concatenate([np.zeros(width, np.float32), x.astype(np.float32), np.zeros(width + orig, np.float32)])
    for d in range(-half, half + 1):
        kk = centre + d
        ok = (kk >= 0) & (kk < k.shape[0])
        # the phase term is a float32 quotient in torch (int64 arange / int), see oracle/resample.py
        t = ((-ph).astype(np.float32) / np.float32(new)).astype(np.float64) + (kk - width) / orig
        t = t * base
        t = np.clip(t, -6.0, 6.0)
        win = np.cos(t * math.pi / 6 / 2) ** 2
        tp = t * math.pi
        tap = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp)) * win * (base / orig)
        src = blk * orig + kk
        ok &= src < xp.shape[0]
        val = np.where(ok, xp[np.where(ok, src, 0)] * tap.astype(np.float32), 0.0)
        out += val
    return out.astype(np.float32)


def pitch_shift(x: np.ndarray, n_steps: int, sample_rate: int = 16000, mode: str = "as_constructed") -> np.ndarray:
    """One mono clip (n,) float32 -> (n,) float32."""
    if mode not in ("as_constructed", "functional"):
        raise ValueError(mode)
    n = x.shape[0]
    y = stretch_waveform(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), n_steps).numpy()
    if mode == "functional":
        y = sinc_resample_any(y, int(sample_rate / stretch_rate(n_steps)), sample_rate)
    out = np.zeros(n, dtype=np.float32)
    m = min(n, y.shape[0])
    out[:m] = y[:m]
    return out
