"""The persistent kernels walk (clip, run) work items.  Ragged batches up to a size limit are compacted through an LDS
prefix of the clips' run counts (Whisper: 3072 clips, Kaldi: 1024); larger or uniform batches keep the dense
run x clip space.  Both spaces must give the same bits: one call over a batch beyond the limit (dense) against the
same clips in chunks below it (compacted), with clips of every awkward length in the middle of the batch."""
import numpy as np
import pytest
import torch

from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def ragged_lengths(n, hi, seed, specials):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, hi, size=n)
    pos = rng.choice(n, size=len(specials), replace=False)
    lens[pos] = specials
    return lens


def packed(lens, seed):
    n_max = (int(lens.max()) + 63) // 64 * 64
    pcm = synth_batch_device(seed, len(lens), n_max, DEV).reshape(-1)
    off = (torch.arange(len(lens), dtype=torch.int64) * n_max).to(DEV)
    return pcm, off, torch.tensor(lens, dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_dense_and_compacted_items_agree(n_mels):
    lens = ragged_lengths(3100, 40000, 5, [1, 199, 200, 201, 7479, 7480, 7481, 15160, 39999, 160, 0])
    pcm, off, ln = packed(lens, 11)
    fe = HipWhisperFeatureExtractor(n_mels, device=DEV)
    full = fe.extract(pcm, off, ln)                                   # 3100 clips: dense items
    for c0, c1 in [(0, 1500), (1500, 3100), (37, 38), (100, 612)]:    # compacted (and a one-clip, uniform, dense) call
        part = fe.extract(pcm, off[c0:c1], ln[c0:c1])
        assert torch.equal(part, full[c0:c1]), (c0, c1)
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_kaldi_dense_and_compacted_items_agree(precision):
    lens = ragged_lengths(1100, 60000, 6, [0, 1, 399, 400, 559, 560, 561, 5519, 5520, 59999, 399, 0])
    pcm, off, ln = packed(lens, 12)
    fe = HipSpeech2TextFeatureExtractor(device=DEV, stft_precision=precision)
    t_max = max(fe.num_frames(int(n)) for n in lens)
    full, mask = fe.extract(pcm, off, ln, t_max, apply_cmvn=False)    # 1100 clips: dense items
    for c0, c1 in [(0, 600), (600, 1100), (5, 6), (300, 812)]:
        part, pmask = fe.extract(pcm, off[c0:c1], ln[c0:c1], t_max, apply_cmvn=False)
        assert torch.equal(part, full[c0:c1]), (precision, c0, c1)
        assert torch.equal(pmask, mask[c0:c1])
