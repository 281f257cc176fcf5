"""Oracle for row S of SURVEY.md section 8a: SpecAugment time / frequency masks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: ``AudioAugmentationCPU.apply_spectrogram_augmentations`` (src/data.py:185-194)
with ``torchaudio.transforms.TimeMasking(time_mask_param=20)`` and
``FrequencyMasking(freq_mask_param=30)`` (src/data.py:129-130); defaults from
src/config.py:39-44 (probability 0.4, 2 time masks then 2 frequency masks).

torchaudio==2.5.1 is not installed; ``mask_along_axis`` is restated from its published
algorithm: with p = 1.0 and mask_value = 0.0,

    value     = torch.rand(1) * mask_param            (drawn first)
    min_value = torch.rand(1) * (axis_len - value)    (drawn second)
    start = floor(min_value); end = floor(min_value) + floor(value)
    x[..., start:end (along axis)] = 0.0

``torch.rand`` and Python ``random`` are available, so the draw sequence is reproduced
exactly from the seeds.  **Parity with a real torchaudio install is unpinned.**
"""
from __future__ import annotations

import random

import numpy as np
import torch

AXIS_FREQ = 0
AXIS_TIME = 1


def draw_mask(axis_len: int, mask_param: int):
    """One mask_along_axis draw -> (start, end) ints; consumes two torch.rand(1)."""
    if mask_param < 1:
        return 0, 0
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (axis_len - value)
    start = int(min_value.long().item())
    end = start + int(value.long().item())
    return start, end


def draw_clip_masks(n_freq: int, n_time: int, *, spec_augment: bool = True, probability: float = 0.4,
                    time_mask_max: int = 20, freq_mask_max: int = 30, num_time_masks: int = 2,
                    num_freq_masks: int = 2):
    """Gate + draws for ONE clip in the reference order -> list of (axis, start, end).

    Consumes one ``random.random()`` and, when the gate passes, 2 torch.rand(1) per mask.
    """
    masks = []
    if spec_augment and random.random() < probability:
        for _ in range(num_time_masks):
            s, e = draw_mask(n_time, time_mask_max)
            masks.append((AXIS_TIME, s, e))
        for _ in range(num_freq_masks):
            s, e = draw_mask(n_freq, freq_mask_max)
            masks.append((AXIS_FREQ, s, e))
    return masks


def apply_masks(feat: np.ndarray, masks) -> np.ndarray:
    """feat (freq, time) -> masked copy (masked_fill with 0.0)."""
    out = np.array(feat, copy=True)
    for axis, s, e in masks:
        if axis == AXIS_TIME:
            out[:, s:e] = 0.0
        else:
            out[s:e, :] = 0.0
    return out
