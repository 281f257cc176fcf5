"""Diagnostic: pitch shift error vs the oracle on the smoke clip family (prints max / rms / where)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle.pitch_shift import pitch_shift
from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import synth_batch
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device="cuda:0")
for seed, lens in [(41, [16000, 52000]), (42, [48000, 80000]), (43, [160000])]:
    clips = synth_batch(seed, lens)
    for st in (2, -1):
        packed = pre.preprocess_batch(clips, 16000)
        out = GpuWaveformAugmenter(WaveformAugmentSampler()).pitch_shift(packed, [st] * len(clips))
        for b, c in enumerate(clips):
            got = out.clip(b)[0].cpu().numpy()
            exp = pitch_shift(c, st)
            e = np.abs(got - exp)
            i = int(np.argmax(e))
            print(seed, len(c), st, "peak", float(np.abs(exp).max()), "max", float(e.max()), "at", i, "rms", float(np.sqrt((e ** 2).mean())),
                  "n>1e-4", int((e > 1e-4 * np.abs(exp).max()).sum()), flush=True)
