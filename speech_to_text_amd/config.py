"""The configuration fields the front-end reads, with the reference's names and defaults.

Mirror of the hot-path subset of ``src/config.py`` (``AugmentationConfig`` :12-44 -- the SpecAugment
fields and the waveform-domain ones the GPU front-end applies -- and ``DataConfig`` :67-89).  Any object with these attributes works (the
reference's own OmegaConf / dataclass config objects included); nothing else of the reference's
configuration system is rebuilt here.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class AugmentationConfig:
    enabled: bool = False                    # src/config.py:14
    add_noise: bool = True                   # :17
    noise_probability: float = 0.3           # :18
    noise_factor: float = 0.02               # :19
    speed_perturbation: bool = True          # :21
    speed_probability: float = 0.3           # :22
    speed_factor_range: tuple = (0.9, 1.1)   # :23
    speed_num_steps: int = 3                 # :24
    pitch_shift: bool = True                 # :26 (sttfe_pitch_shift; GpuFrontEnd(pitch_mode=...))
    pitch_probability: float = 0.3           # :27
    pitch_shift_range: tuple = (-2, 2)       # :28
    volume_perturbation: bool = True         # :30
    volume_probability: float = 0.3          # :31
    volume_range_db: tuple = (-10, 10)       # :32
    fade_inout: bool = True                  # :34
    fade_probability: float = 0.2            # :35
    fade_duration: float = 0.1               # :36 seconds
    spec_augment: bool = True                # src/config.py:39
    spec_augment_probability: float = 0.4    # :40
    time_mask_max_size: int = 20             # :41 frames
    freq_mask_max_size: int = 30             # :42 frequency bins
    num_time_masks: int = 2                  # :43
    num_freq_masks: int = 2                  # :44


@dataclass
class DataConfig:
    max_duration: float = 30.0               # src/config.py:67
    min_duration: float = 0.5                # :68
    sample_rate: int = 16000                 # :69
    normalize: bool = True                   # :72
    trim_silence: bool = True                # :73
    silence_threshold: float = 0.01          # :74
    processing_and_augmentation_device: str = "cpu"   # :76 'cpu' | 'cuda'
    pin_memory: bool = True                  # :82
    max_source_positions: int = 6000         # :85 Speech2Text frame cap
    max_target_positions: int = 1024         # :86
    augmentation: AugmentationConfig = field(default_factory=AugmentationConfig)   # :89
