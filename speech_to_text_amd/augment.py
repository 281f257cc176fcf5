"""SpecAugment for the GPU collate mode: the host draws the mask bounds, the kernels apply them.

Counterpart of ``AudioAugmentationCPU.apply_spectrogram_augmentations`` (reference
``src/data.py:185-194``) with ``torchaudio.transforms.TimeMasking(time_mask_param=...)`` /
``FrequencyMasking(freq_mask_param=...)`` (``src/data.py:129-130``).  A mask is four integers, so the
random draws stay on the host, in the reference's order and from the same generators (Python
``random`` for the gate, the global torch CPU generator for the bounds): seeding both reproduces the
reference's masks bit for bit.  The (B, n_masks, 3) int32 table {axis, start, end} is handed to
``sttfe_whisper_logmel`` / ``sttfe_kaldi_fbank_cmvn``, which write 0.0 over the masked spans in the
same pass that normalises the features.

Draw order per clip (torchaudio 2.5.1 ``functional.mask_along_axis`` with p = 1.0, mask_value = 0.0)::

    random.random() < spec_augment_probability              # gate, consumed even when it fails
    for each time mask, then each frequency mask:
        value     = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (axis_len - value)
        start, end = floor(min_value), floor(min_value) + floor(value)
"""
from __future__ import annotations

import random
from typing import Any, List, Sequence, Tuple

import torch

AXIS_FREQ = 0        # STTFE_AXIS_FREQ
AXIS_TIME = 1        # STTFE_AXIS_TIME


class SpecAugmentSampler:
    """Draws SpecAugment masks in the reference's RNG order.  ``config`` = AugmentationConfig-like."""

    def __init__(self, config: Any = None, **overrides):
        get = lambda k, d: overrides.get(k, getattr(config, k, d) if config is not None else d)
        self.spec_augment = bool(get("spec_augment", True))
        self.probability = float(get("spec_augment_probability", 0.4))
        self.time_mask_max_size = int(get("time_mask_max_size", 20))
        self.freq_mask_max_size = int(get("freq_mask_max_size", 30))
        self.num_time_masks = int(get("num_time_masks", 2))
        self.num_freq_masks = int(get("num_freq_masks", 2))

    @property
    def masks_per_clip(self) -> int:
        return self.num_time_masks + self.num_freq_masks

    @staticmethod
    def _draw(axis_len: int, mask_param: int) -> Tuple[int, int]:
        if mask_param < 1:                       # mask_along_axis returns the input untouched, no draws
            return 0, 0
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (axis_len - value)
        start = int(min_value.long().item())
        return start, start + int(value.long().item())

    def draw_clip(self, n_freq: int, n_time: int) -> List[Tuple[int, int, int]]:
        """Masks of one (n_freq, n_time) clip: [] when the gate fails, else time masks then freq masks."""
        if not (self.spec_augment and random.random() < self.probability):
            return []
        masks = [(AXIS_TIME, *self._draw(n_time, self.time_mask_max_size)) for _ in range(self.num_time_masks)]
        masks += [(AXIS_FREQ, *self._draw(n_freq, self.freq_mask_max_size)) for _ in range(self.num_freq_masks)]
        return masks

    def mask_table(self, shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
        """One draw per clip, in batch order -> (B, masks_per_clip, 3) int32 host tensor.

        Clips whose gate failed get no-op rows (start == end == 0).
        """
        B, M = len(shapes), max(self.masks_per_clip, 1)
        tbl = torch.zeros((B, M, 3), dtype=torch.int32)
        for b, (n_freq, n_time) in enumerate(shapes):
            for i, (axis, s, e) in enumerate(self.draw_clip(int(n_freq), int(n_time))):
                tbl[b, i, 0], tbl[b, i, 1], tbl[b, i, 2] = axis, s, e
        return tbl
