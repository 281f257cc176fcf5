"""Row S: the product's host-side mask sampler draws exactly what the oracle (and the self-generated
golden) draws from the same seeds.  torchaudio is absent: parity with a real install is UNPINNED
(oracle/specaugment.py); what is pinned here is the RNG order and the bound arithmetic."""
import random

import numpy as np
import torch

from conftest import load_golden
from oracle import specaugment as S
from speech_to_text_amd.augment import AXIS_FREQ, AXIS_TIME, SpecAugmentSampler
from speech_to_text_amd.config import AugmentationConfig


def test_sampler_matches_oracle_draw_for_draw():
    cfg = AugmentationConfig(enabled=True)
    sampler = SpecAugmentSampler(cfg)
    shapes = [(80, 3000), (128, 3000), (80, 498), (80, 17), (80, 1)] * 40
    torch.manual_seed(11)
    random.seed(11)
    got = [sampler.draw_clip(f, t) for f, t in shapes]
    torch.manual_seed(11)
    random.seed(11)
    exp = [S.draw_clip_masks(f, t) for f, t in shapes]
    assert got == exp
    assert (AXIS_FREQ, AXIS_TIME) == (S.AXIS_FREQ, S.AXIS_TIME)


def test_mask_table_matches_golden_seed104():
    z = load_golden("selfgen_resample_masks.npz")
    sampler = SpecAugmentSampler(AugmentationConfig(enabled=True))
    torch.manual_seed(104)
    random.seed(104)
    tbl = sampler.mask_table([(128, 3000)] * 64)
    assert tbl.shape == (64, 4, 3) and tbl.dtype == torch.int32
    rows = [(b, *tbl[b, i].tolist()) for b in range(64) for i in range(4) if tbl[b, i, 2] > tbl[b, i, 1] or tbl[b, i].any()]
    gold = z["masks_seed104_128x3000"]
    # golden rows are (clip, axis, start, end) for gated-in clips, including zero-width masks
    gated = sorted(set(gold[:, 0].tolist()))
    exp = np.zeros((64, 4, 3), dtype=np.int32)
    for b in gated:
        exp[b] = gold[gold[:, 0] == b][:, 1:]
    assert np.array_equal(tbl.numpy(), exp)


def test_disabled_flags_consume_no_randomness():
    sampler = SpecAugmentSampler(AugmentationConfig(enabled=True, spec_augment=False))
    random.seed(3)
    a = random.random()
    random.seed(3)
    assert sampler.draw_clip(80, 3000) == []
    assert random.random() == a
    zero = SpecAugmentSampler(AugmentationConfig(enabled=True, spec_augment_probability=1.0,
                                                 time_mask_max_size=0, freq_mask_max_size=0))
    st = torch.get_rng_state()
    assert zero.draw_clip(80, 3000) == [(AXIS_TIME, 0, 0)] * 2 + [(AXIS_FREQ, 0, 0)] * 2
    assert torch.equal(torch.get_rng_state(), st)


def test_bulk_uniform_draws_equal_successive_single_draws():
    """SpecAugmentSampler.draw_clip takes a clip's uniform numbers in one torch.rand(2 k) call; the reference draws
    them one torch.rand(1) at a time: same numbers, same order, same generator state afterwards."""
    for n in (2, 8, 16, 17, 64):
        torch.manual_seed(11)
        single = torch.cat([torch.rand(1) for _ in range(n)])
        after_single = torch.rand(1)
        torch.manual_seed(11)
        bulk = torch.rand(n)
        after_bulk = torch.rand(1)
        assert torch.equal(single, bulk) and torch.equal(after_single, after_bulk)


def test_batched_mask_table_equals_per_clip_draws():
    """``mask_table`` draws the whole batch at once (the gates from Python's generator, every passing clip's uniform
    numbers from ONE ``torch.rand`` call): the table and both generators' states afterwards are those of per-clip
    ``draw_clip`` calls in batch order -- for the default configuration, with an axis switched off, without frequency
    masks, with every clip passing and with nothing to draw."""
    import random

    import torch

    from speech_to_text_amd.augment import SpecAugmentSampler
    shapes = [(80 if i % 3 else 128, 100 + 7 * i) for i in range(300)]
    for kw in ({}, {"time_mask_max_size": 0}, {"num_freq_masks": 0}, {"spec_augment_probability": 1.0},
               {"freq_mask_max_size": 0, "time_mask_max_size": 0}, {"spec_augment": False}):
        s = SpecAugmentSampler(None, **kw)
        random.seed(5)
        torch.manual_seed(5)
        got = s.mask_table(shapes)
        after_got = (random.random(), torch.rand(1).item())
        random.seed(5)
        torch.manual_seed(5)
        exp = torch.zeros((len(shapes), max(s.masks_per_clip, 1), 3), dtype=torch.int32)
        for b, (n_freq, n_time) in enumerate(shapes):
            for i, (axis, start, end) in enumerate(s.draw_clip(n_freq, n_time)):
                exp[b, i, 0], exp[b, i, 1], exp[b, i, 2] = axis, start, end
        after_exp = (random.random(), torch.rand(1).item())
        assert torch.equal(got, exp), kw
        assert after_got == after_exp, kw
