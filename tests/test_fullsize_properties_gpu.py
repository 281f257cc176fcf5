"""Size-independent properties at BASELINE.json's full batch sizes (the oracle would take minutes there).

config 3 (512 Common-Voice-length clips @48 kHz): the resampler is linear and power-of-two scaling is exact
in every stage, so resample(x / 4) == resample(x) / 4 bit for bit, the peak-normalised result is
unchanged up to the 1e-8 in the denominator, and lengths follow ceil(n / 3).
config 4 (1024 clips, 128 mels, SpecAugment): masked spans are exactly 0.0 and every other element is
bit-identical to the unmasked run; padding frames hold the clamped constant."""
import random

import numpy as np
import pytest
import torch

from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_config3_resample_linearity_and_lengths():
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    B = 512
    lens = (common_voice_lengths(np.random.default_rng(103), B) * 3 + np.arange(B) % 3).tolist()
    n_max = (max(lens) + 63) // 64 * 64
    pcm = synth_batch_device(103, B, n_max, DEV, sample_rate=48000)
    off = (torch.arange(B, dtype=torch.int64) * n_max).to(DEV)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    raw = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    a = raw.run(pcm.reshape(-1), off, ln, lens, 1, 48000)
    b = raw.run((pcm * 0.25).reshape(-1), off, ln, lens, 1, 48000)
    assert a.host_lengths == [-(-n // 3) for n in lens] and a.lengths.cpu().tolist() == a.host_lengths
    offs = a.offsets.cpu().tolist()
    for i in range(B):                                    # the alignment gaps between clips are uninitialised
        sl = slice(offs[i], offs[i] + a.host_lengths[i])
        assert torch.equal(b.pcm[sl], a.pcm[sl] * 0.25), i
    peaks = raw.last_peaks.clone()
    nrm = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    c = nrm.run(pcm.reshape(-1), off, ln, lens, 1, 48000)
    for i in (0, 17, 511):
        clip = c.clip(i)[0]
        assert abs(float(clip.abs().max()) - 1.0) < 1e-6
        assert torch.equal(clip, a.clip(i)[0] / (a.clip(i)[0].abs().max() + 1e-8))
    assert float(peaks.min()) > 0


def test_config4_specaugment_masks_exact_at_full_batch():
    from speech_to_text_amd.augment import SpecAugmentSampler
    from speech_to_text_amd.config import AugmentationConfig
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    B, M = 1024, 128
    lens = common_voice_lengths(np.random.default_rng(104), B)
    n_max = (int(lens.max()) + 63) // 64 * 64
    pcm = synth_batch_device(104, B, n_max, DEV).reshape(-1)
    off = (torch.arange(B, dtype=torch.int64) * n_max).to(DEV)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    fe = HipWhisperFeatureExtractor(M, device=DEV)
    base = fe.extract(pcm, off, ln)
    torch.manual_seed(104)
    random.seed(104)
    tbl = SpecAugmentSampler(AugmentationConfig(enabled=True)).mask_table([(M, 3000)] * B)
    got = fe.extract(pcm, off, ln, mask_table=tbl)
    keep = torch.ones((B, M, 3000), dtype=torch.bool, device=DEV)
    t = tbl.to(DEV)
    fr = torch.arange(3000, device=DEV)[None, :]
    ml = torch.arange(M, device=DEV)[None, :]
    for i in range(tbl.shape[1]):
        axis, s, e = t[:, i, 0:1], t[:, i, 1:2], t[:, i, 2:3]
        keep &= ~((axis == 1) & (fr >= s) & (fr < e))[:, None, :]
        keep &= ~((axis == 0) & (ml >= s) & (ml < e))[:, :, None]
    assert 0.3 < float((tbl[:, :, 2] > tbl[:, :, 1]).any(dim=1).float().mean()) < 0.5     # gate p = 0.4
    assert torch.equal(got[keep], base[keep])
    assert float(got[~keep].abs().max()) == 0.0
    # padding region: constant per clip, equal to max(-1.5, (max - 8 + 4) / 4) of that clip
    tail = base[:, :, 2900:]
    assert torch.equal(tail, tail[:, :1, :1].expand_as(tail))
    lmax = base.amax(dim=(1, 2)) * 4.0 - 4.0
    assert float((tail[:, 0, 0] - torch.maximum((lmax - 8.0 + 4.0) * 0.25, torch.full_like(lmax, -1.5))).abs().max()) <= 1e-6
