"""Oracle for rows P1, R, P2, P3, P4 of SURVEY.md section 8a: mono, resample, normalise, trim, gate.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call sites:
  src/data.py:66-67      mono mean over channels
  src/data.py:70-72      torchaudio.transforms.Resample(orig, 16000)(audio)   (also inference.py:111-113)
  src/data.py:78-81      x / (max|x| + 1e-8) if max > 0                         (also inference.py:125)
  src/data.py:84-91      silence trim: |x| > 0.01 * max|x|, slice first..last+1
  src/data.py:95-98      duration gate

Row R's arithmetic lives in torchaudio==2.5.1 (reference requirements.txt:3), which is
NOT installed here and whose source is not on disk.  ``sinc_resample_kernel`` /
``resample`` restate its published algorithm (functional._get_sinc_resample_kernel /
_apply_sinc_resample_kernel with the transform's defaults: sinc_interp_hann,
lowpass_filter_width=6, rolloff=0.99, kernel built in float64 and cast to float32,
zero padding (width, width + orig), conv1d stride orig, crop to ceil(new * n / orig))
as written down in SURVEY.md section 8a row R.  **Parity with a real torchaudio install
is unpinned**; tests check self-consistency only (DC gain, tone gain, lengths).
"""
from __future__ import annotations

import math

import numpy as np


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99):
    """Returns (kernels float32 (new, 2*width + orig), width, orig, new) with orig/new reduced by gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig = int(orig_freq) // g
    new = int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    # torch: int64 arange / int -> float32 true division, then promoted to float64 by the add
    phase = (np.arange(0, -new, -1).astype(np.float32) / np.float32(new)).astype(np.float64)[:, None]
    t = (phase + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    with np.errstate(divide="ignore", invalid="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t)
    kern = kern * window * scale
    return kern.astype(np.float32), width, orig, new


def resample(wave: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """1-D float32 -> 1-D float32 of length ceil(new * n / orig) (row R)."""
    wave = np.asarray(wave, dtype=np.float32).reshape(-1)
    if int(orig_freq) == int(new_freq):
        return wave.copy()
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    n = wave.shape[0]
    padded = np.zeros(n + 2 * width + orig, dtype=np.float64)
    padded[width:width + n] = wave
    taps = kern.shape[1]
    n_out = (padded.shape[0] - taps) // orig + 1
    idx = orig * np.arange(n_out)[:, None] + np.arange(taps)[None, :]
    frames = padded[idx]                                        # (n_out, taps)
    out = frames @ kern.astype(np.float64).T                    # (n_out, new): phases interleaved
    out = out.reshape(-1)
    target = int(math.ceil(new * n / orig))
    return out[:target].astype(np.float32)


def to_mono(audio: np.ndarray) -> np.ndarray:
    """(C, n) -> (n,)  src/data.py:66-67 (float32 mean)."""
    audio = np.asarray(audio, dtype=np.float32)
    if audio.ndim == 1:
        return audio
    if audio.shape[0] > 1:
        return audio.mean(axis=0, dtype=np.float32)
    return audio[0]


def peak_normalize(wave: np.ndarray) -> np.ndarray:
    """src/data.py:78-81, float32 arithmetic."""
    wave = np.asarray(wave, dtype=np.float32)
    if wave.size == 0:
        return wave
    m = np.max(np.abs(wave))
    if m > 0:
        return (wave / (m + np.float32(1e-8))).astype(np.float32)
    return wave


def trim_bounds(wave: np.ndarray, silence_threshold: float = 0.01):
    """src/data.py:84-91 -> (start, end) or None when nothing is above threshold."""
    energy = np.abs(np.asarray(wave, dtype=np.float32))
    if energy.size == 0:
        return None
    thr = np.float32(energy.max()) * np.float32(silence_threshold)
    nz = np.nonzero(energy > thr)[0]
    if nz.size == 0:
        return None
    return int(nz[0]), int(nz[-1]) + 1


def preprocess(audio: np.ndarray, original_sample_rate: int, sample_rate: int = 16000,
               normalize: bool = True, trim_silence: bool = False, silence_threshold: float = 0.01,
               min_duration: float = 0.0, max_duration: float = 1e9):
    """AudioPreprocessor.preprocess (src/data.py:54-108) -> dict(audio (n,), duration) or None."""
    wave = to_mono(audio)
    if original_sample_rate != sample_rate:
        wave = resample(wave, original_sample_rate, sample_rate)
    duration = wave.shape[0] / sample_rate
    if normalize:
        wave = peak_normalize(wave)
    if trim_silence:
        b = trim_bounds(wave, silence_threshold)
        if b is not None:
            wave = wave[b[0]:b[1]]
            duration = (b[1] - b[0]) / sample_rate
    if not (min_duration <= duration <= max_duration):
        return None
    return {"audio": wave, "sample_rate": sample_rate, "duration": duration}
