"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the reference's
waveform-domain augmentations.

Follows ``AudioAugmentationCPU.apply_time_domain_augmentations`` (reference ``src/data.py:152-183``) and the
torchaudio 2.5.1 transforms it calls (``Vol`` amplitude gain + clamp, ``Fade`` linear, ``SpeedPerturbation`` ->
``functional.speed`` -> sinc resample).  torchaudio is not installed in the build container and the reference
holds no golden vectors for this row: **parity unpinned** -- the transforms are restated from the torchaudio
2.5.1 algorithm text (as rows R and S of SURVEY.md 8c), the torch pieces they are made of (``torch.randn``,
``torch.linspace``, ``torch.clamp``, ``random``) are the real ones.

Draw order per clip (every gate consumes one ``random.random()`` whether it passes or not)::

    noise  gate -> torch.randn_like(audio)                    (global torch CPU generator)
    speed  gate -> torch.randint(len(factors), ())            factors = np.linspace(lo, hi, steps)
    pitch  gate -> random.randint(lo, hi)                     (semitones; applied by oracle/pitch_shift.py)
    volume gate -> random.uniform(lo_db, hi_db); gain = 10 ** (db / 20)
    fade   gate -> fade samples = int(fade_duration * sample_rate)
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from math import gcd
from typing import Any, Optional

import numpy as np
import torch

from .pitch_shift import pitch_shift as _pitch_shift
from .resample import resample as _resample


@dataclass
class ClipDraw:
    noise_factor: Optional[float] = None      # with `noise`: (1, n) standard-normal tensor
    noise: Optional[torch.Tensor] = None
    speed_factor: Optional[float] = None
    pitch_steps: Optional[int] = None
    gain: Optional[float] = None
    fade_samples: Optional[int] = None


def _get(cfg: Any, key: str, default):
    return getattr(cfg, key, default) if cfg is not None else default


def draw_time_domain(cfg: Any, n_samples: int, sample_rate: int = 16000) -> ClipDraw:
    """The random draws of one clip, in the reference's order (src/data.py:155-181)."""
    d = ClipDraw()
    if _get(cfg, "add_noise", True) and random.random() < _get(cfg, "noise_probability", 0.3):
        d.noise_factor = float(_get(cfg, "noise_factor", 0.02))
        d.noise = torch.randn(1, n_samples)                                   # randn_like(audio), audio (1, n) f32
    if _get(cfg, "speed_perturbation", True) and random.random() < _get(cfg, "speed_probability", 0.3):
        lo, hi = _get(cfg, "speed_factor_range", (0.9, 1.1))
        factors = np.linspace(lo, hi, int(_get(cfg, "speed_num_steps", 3))).tolist()      # src/data.py:145-149
        d.speed_factor = float(factors[int(torch.randint(len(factors), ()))])
    if _get(cfg, "pitch_shift", True) and random.random() < _get(cfg, "pitch_probability", 0.3):
        d.pitch_steps = random.randint(*_get(cfg, "pitch_shift_range", (-2, 2)))
    if _get(cfg, "volume_perturbation", True) and random.random() < _get(cfg, "volume_probability", 0.3):
        gain_db = random.uniform(*_get(cfg, "volume_range_db", (-10, 10)))
        d.gain = 10 ** (gain_db / 20.0)
    if _get(cfg, "fade_inout", True) and random.random() < _get(cfg, "fade_probability", 0.2):
        d.fade_samples = int(_get(cfg, "fade_duration", 0.1) * sample_rate)
    return d


def speed_rates(factor: float, sample_rate: int = 16000):
    """torchaudio.functional.speed: (source, target) rates of the resample, reduced by their gcd."""
    source, target = int(factor * sample_rate), int(sample_rate)
    g = gcd(source, target)
    return source // g, target // g


def apply_time_domain(x: np.ndarray, d: ClipDraw, sample_rate: int = 16000,
                      pitch_mode: Optional[str] = "as_constructed") -> np.ndarray:
    """One mono clip (n,) float32 -> augmented clip, torch CPU float32 arithmetic.  ``pitch_mode`` None skips the
    pitch shift (its draw stays consumed)."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).reshape(1, -1)
    if d.noise_factor is not None:
        t = t + d.noise * d.noise_factor                                      # src/data.py:156-157
    if d.speed_factor is not None and d.speed_factor != 1.0:
        src, dst = speed_rates(d.speed_factor, sample_rate)
        if src != dst:
            t = torch.from_numpy(_resample(t.numpy().reshape(-1), src, dst)).reshape(1, -1)
    if d.pitch_steps is not None and pitch_mode is not None:                  # src/data.py:164-167
        t = torch.from_numpy(_pitch_shift(t.numpy().reshape(-1), int(d.pitch_steps), sample_rate, pitch_mode)).reshape(1, -1)
    if d.gain is not None:
        t = torch.clamp(t * d.gain, -1, 1)                                    # torchaudio Vol, gain_type="amplitude"
    if d.fade_samples is not None:
        n, L = int(d.fade_samples), t.shape[-1]
        fade_in = torch.cat((torch.linspace(0, 1, n), torch.ones(L - n))).clamp_(0, 1)
        fade_out = torch.cat((torch.ones(L - n), -torch.linspace(0, 1, n) + 1)).clamp_(0, 1)
        t = fade_in * fade_out * t                                            # torchaudio Fade.forward
    return t.reshape(-1).numpy()
