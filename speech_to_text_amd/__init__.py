"""sttfe -- MI355X-native audio front-end (log-mel / Kaldi fbank / resample / SpecAugment).

Host side of the drop-in for the feature-extraction path of Alexey511/speech_to_text
(reference ``src/processors.py`` / ``src/data.py``).  All arithmetic runs in hand-written HIP
kernels behind the C-ABI library ``libsttfe.so`` (``include/sttfe.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .augment import GpuWaveformAugmenter, SpecAugmentSampler, WaveformAugmentSampler, WaveformDraw  # noqa: F401
from .config import AugmentationConfig, DataConfig  # noqa: F401
from .data import AudioCollator, GpuFrontEnd  # noqa: F401
from .feature_cache import FeatureCacheDataset, FeatureCacheWriter  # noqa: F401
from .preprocess import GpuAudioPreprocessor, PackedAudio, PinnedIngestRing, ResamplePlan  # noqa: F401
from .processors import (BatchFeature, HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor,  # noqa: F401
                         TorchWhisperFeatureExtractor, pack_clips)
