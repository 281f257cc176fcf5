"""GPU parity of the waveform-domain augmentations through the C ABI
(``sttfe_waveform_augment`` + the resampler for speed perturbation) vs oracle/waveform_aug.py.

Bars: noise (host-drawn, as the reference draws it) and the clamped amplitude gain are float32 products and sums
with the same roundings -> bit-exact.  The linear fade multiplies by torch.linspace ramps; torch's vectorised CPU
linspace may differ from its scalar formula (which the kernel uses) in the last bit of a ramp value
-> |d| <= 2.5e-7 * |x|, x = the sample before the fade (near the end 1 - ramp cancels, so the bound is absolute in x).
Speed perturbation is the generic polyphase resampler (float32 accumulation of <= 23 products vs the oracle's
float64) -> |d| <= 2e-6, as for row R.  torchaudio is absent: parity with a real install is UNPINNED."""
import random

import numpy as np
import pytest
import torch

from oracle import waveform_aug as W
from speech_to_text_amd.synth import synth_clip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def pack(clips):
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    return pre.preprocess_batch(clips, 16000)


def clips_of(lens, seed=0, scale=1.0):
    return [synth_clip(np.random.default_rng(seed + i), n, zero_block=False) * np.float32(scale) for i, n in enumerate(lens)]


def draws_to_oracle(d):
    return W.ClipDraw(noise_factor=d.noise_factor, noise=None if d.noise is None else d.noise.reshape(1, -1),
                      speed_factor=d.speed_factor, gain=d.gain, fade_samples=d.fade_samples)


def test_noise_gain_fade_in_place():
    from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler, WaveformDraw
    lens = [16000, 4801, 3200, 48000, 9, 20000]
    clips = clips_of(lens, 5, scale=2.5)                  # |x| > 1 so the gain clamp bites
    s = WaveformAugmentSampler(noise_source="host")
    torch.manual_seed(1)
    draws = [
        WaveformDraw(noise_factor=0.02, noise=torch.randn(lens[0])),
        WaveformDraw(gain=10 ** (7.3 / 20)),
        WaveformDraw(fade_samples=1600),
        WaveformDraw(noise_factor=0.05, noise=torch.randn(lens[3]), gain=10 ** (-4.0 / 20), fade_samples=1600),
        WaveformDraw(),
        WaveformDraw(noise_factor=0.02, noise=torch.randn(lens[5]), gain=1.7),
    ]
    p = pack(clips)
    buf = p.pcm.data_ptr()
    out = GpuWaveformAugmenter(s).apply(p, draws)
    assert out.pcm.data_ptr() == buf and out.host_lengths == lens            # in place, no speed change
    for i, (c, d) in enumerate(zip(clips, draws)):
        exp = W.apply_time_domain(c, draws_to_oracle(d))
        got = out.clip(i)[0].cpu().numpy()
        if d.fade_samples is None:
            assert np.array_equal(got, exp), i
        else:
            od = draws_to_oracle(d)
            od.fade_samples = None
            pre = W.apply_time_domain(c, od)                                 # the samples the fade multiplies
            assert np.all(np.abs(got - exp) <= 2.5e-7 * np.abs(pre) + 1e-12), (i, float(np.abs(got - exp).max()))
            assert got[0] == 0.0 and got[-1] == 0.0
    assert out.keep == [True] * len(lens)


def test_fade_longer_than_clip_is_dropped():
    from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler, WaveformDraw
    clips = clips_of([1000, 5000], 9)
    out = GpuWaveformAugmenter(WaveformAugmentSampler()).apply(pack(clips), [WaveformDraw(fade_samples=1600)] * 2)
    assert out.keep == [False, True]
    assert np.array_equal(out.clip(0)[0].cpu().numpy(), clips[0])            # left untouched


def test_speed_perturbation_with_the_other_stages():
    from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler, WaveformDraw
    lens = [16000, 30001, 8000, 24000, 12345]
    clips = clips_of(lens, 21)
    s = WaveformAugmentSampler(noise_source="host")
    torch.manual_seed(2)
    draws = [
        WaveformDraw(speed_factor=0.9),
        WaveformDraw(noise_factor=0.02, noise=torch.randn(lens[1]), speed_factor=1.1, gain=1.5, fade_samples=1600),
        WaveformDraw(gain=0.5),                                              # no speed change: copied
        WaveformDraw(speed_factor=1.0, fade_samples=1600),                   # factor 1.0 = identity
        WaveformDraw(speed_factor=1.1),
    ]
    out = GpuWaveformAugmenter(s).apply(pack(clips), draws)
    assert out.lengths.cpu().tolist() == out.host_lengths
    for i, (c, d) in enumerate(zip(clips, draws)):
        exp = W.apply_time_domain(c, draws_to_oracle(d))
        got = out.clip(i)[0].cpu().numpy()
        assert got.shape == exp.shape, (i, got.shape, exp.shape)
        if d.speed_factor in (None, 1.0) and d.fade_samples is None:
            assert np.array_equal(got, exp), i
        else:
            assert np.abs(got - exp).max() <= 3e-6, (i, float(np.abs(got - exp).max()))


def test_sampler_driven_batch_device_noise_statistics():
    """noise_source="device": same distribution, different stream -> check the statistics, not the bits."""
    from speech_to_text_amd.augment import GpuWaveformAugmenter, WaveformAugmentSampler
    s = WaveformAugmentSampler(noise_probability=1.0, speed_perturbation=False, pitch_shift=False,
                               volume_perturbation=False, fade_inout=False, noise_factor=0.02)
    random.seed(5)
    n = 200000
    clips = [np.zeros(n, np.float32) for _ in range(4)]
    out = GpuWaveformAugmenter(s).apply(pack(clips), [s.draw_clip(n) for _ in clips])
    for i in range(4):
        y = out.clip(i)[0].cpu().numpy().astype(np.float64)
        assert abs(y.mean()) < 3 * 0.02 / np.sqrt(n) * 1.5 and abs(y.std() - 0.02) < 2e-4
