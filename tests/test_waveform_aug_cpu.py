"""Host side of the waveform-domain augmentations: the sampler must consume the
Python `random` and torch CPU generators exactly like the oracle's restatement of the reference's
``apply_time_domain_augmentations`` (src/data.py:152-183), clip after clip.  torchaudio is absent:
parity of the transforms with a real install is UNPINNED (oracle/waveform_aug.py)."""
import random

import numpy as np
import torch

from oracle import waveform_aug as W
from speech_to_text_amd.augment import WaveformAugmentSampler


class Cfg:
    add_noise = True
    noise_probability = 0.5
    noise_factor = 0.02
    speed_perturbation = True
    speed_probability = 0.5
    speed_factor_range = (0.9, 1.1)
    speed_num_steps = 3
    pitch_shift = True
    pitch_probability = 0.5
    pitch_shift_range = (-2, 2)
    volume_perturbation = True
    volume_probability = 0.5
    volume_range_db = (-10, 10)
    fade_inout = True
    fade_probability = 0.5
    fade_duration = 0.1


def test_draw_order_matches_oracle():
    lens = [16000, 40001, 777, 52000, 31999, 16001, 8000, 123456]
    random.seed(11)
    torch.manual_seed(11)
    exp = [W.draw_time_domain(Cfg, n) for n in lens]
    random.seed(11)
    torch.manual_seed(11)
    s = WaveformAugmentSampler(Cfg, noise_source="host")
    got = [s.draw_clip(n) for n in lens]
    seen = set()
    for e, g, n in zip(exp, got, lens):
        assert (e.noise_factor, e.speed_factor, e.pitch_steps, e.gain, e.fade_samples) == \
               (g.noise_factor, g.speed_factor, g.pitch_steps, g.gain, g.fade_samples)
        if e.noise is not None:
            assert torch.equal(e.noise.reshape(-1), g.noise) and g.noise.shape[0] == n
        seen |= {k for k in ("noise_factor", "speed_factor", "pitch_steps", "gain", "fade_samples") if getattr(e, k) is not None}
    assert len(seen) == 5                    # every branch was exercised by this seed
    # the generators end in the same state as the oracle's
    a, b = random.random(), float(torch.rand(1))
    random.seed(11)
    torch.manual_seed(11)
    [W.draw_time_domain(Cfg, n) for n in lens]
    assert (a, b) == (random.random(), float(torch.rand(1)))


def test_disabled_stages_draw_nothing():
    s = WaveformAugmentSampler(add_noise=False, speed_perturbation=False, pitch_shift=False,
                               volume_perturbation=False, fade_inout=False)
    random.seed(3)
    first = random.random()
    random.seed(3)
    d = s.draw_clip(16000)
    assert (d.noise_factor, d.speed_factor, d.pitch_steps, d.gain, d.fade_samples) == (None,) * 5
    assert random.random() == first          # no gate consumed


def test_speed_rates_and_lengths():
    s = WaveformAugmentSampler(Cfg)
    assert s.speed_factors == [0.9, 1.0, 1.1]
    assert s.speed_rates(0.9) == (9, 10) and s.speed_rates(1.1) == (11, 10) == W.speed_rates(1.1)
    d = W.ClipDraw(speed_factor=0.9)
    x = np.zeros(16001, np.float32)
    assert W.apply_time_domain(x, d).shape[0] == s.out_length(16001, d) == (10 * 16001 + 8) // 9
    d = W.ClipDraw(speed_factor=1.0)
    assert s.out_length(500, d) == 500


def test_oracle_fade_and_gain_properties():
    x = np.linspace(-2, 2, 4000).astype(np.float32)
    y = W.apply_time_domain(x, W.ClipDraw(gain=0.5))
    assert np.array_equal(y, np.clip(x * np.float32(0.5), -1, 1))
    y = W.apply_time_domain(np.ones(4000, np.float32), W.ClipDraw(fade_samples=1600))
    assert y[0] == 0.0 and y[-1] == 0.0 and np.all(y[1600:2400] == 1.0)
    assert np.all(np.diff(y[:1600]) > 0) and np.all(np.diff(y[-1600:]) < 0)


def test_any_speed_ratio_is_accepted_and_the_sampler_needs_no_library(monkeypatch):
    """torchaudio accepts any speed factor (src/data.py:144-150); so does the sampler.  speed_num_steps=4 gives the factor
    0.9667 -> int(15466.67) = 15466 : 16000 = 7733 : 8000 after reduction, which runs through the table-free resampler
    (tests/test_preprocess_gpu.py::test_reduced_rates_beyond_1024...).  Building the sampler is host configuration +
    RNG state only: it must not load libsttfe.so (CPU-only dataloader workers, config validation)."""
    from speech_to_text_amd import _lib
    from speech_to_text_amd.augment import WaveformAugmentSampler

    def boom(*a, **k):
        raise AssertionError("the sampler constructor touched the library")
    monkeypatch.setattr(_lib, "lib", boom)
    WaveformAugmentSampler()                                   # defaults: 9:10, 1:1, 11:10
    WaveformAugmentSampler(speed_num_steps=5)                  # 9:10, 19:20, 21:20, 11:10
    s4 = WaveformAugmentSampler(speed_num_steps=4)
    assert s4.speed_rates(s4.speed_factors[1]) == (7733, 8000)
    WaveformAugmentSampler(speed_num_steps=7)
