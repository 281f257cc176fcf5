"""Offline feature cache: what goes in comes out bit for bit, with the keys and shapes the
reference's AudioCollator "cpu" branch consumes (src/data.py:462-518)."""
import pickle

import numpy as np
import torch

from speech_to_text_amd.config import DataConfig
from speech_to_text_amd.data import AudioCollator
from speech_to_text_amd.feature_cache import FeatureCacheDataset, FeatureCacheWriter, whisper_computed_frames


def fake_whisper(lens, n_mels=8, n_frames=3000, seed=0):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((len(lens), n_mels, n_frames)).astype(np.float32)
    for b, n in enumerate(lens):
        f[b, :, whisper_computed_frames(n):] = np.float32(-0.5 - 0.01 * b)      # the padding constant
    return torch.from_numpy(f)


def test_computed_frames():
    assert whisper_computed_frames(0) == 48 and whisper_computed_frames(480000) == 3000
    assert whisper_computed_frames(7480) == 48 and whisper_computed_frames(7481) == 96      # 7680 * k - 200
    assert whisper_computed_frames(10 ** 9) == 3000


def test_whisper_round_trip_two_ranks(tmp_path):
    lens0, lens1 = [16000, 480000, 80000], [1, 123456]
    f0, f1 = fake_whisper(lens0, seed=1), fake_whisper(lens1, seed=2)
    w0 = FeatureCacheWriter(str(tmp_path), "whisper", 8, shard_clips=2, rank=0)
    w0.add_batch([0, 2, 4], f0, lens0, texts=["a", "b", "c"], labels=[[1, 2], [3], []])
    w0.close()
    w1 = FeatureCacheWriter(str(tmp_path), "whisper", 8, shard_clips=2, rank=1)
    w1.add_batch([1, 3], f1, lens1, texts=["d", "e"], labels=[[9], [8, 7, 6]])
    w1.close()
    ds = pickle.loads(pickle.dumps(FeatureCacheDataset(str(tmp_path))))       # picklable, like the reference's datasets
    assert len(ds) == 5 and [ds[i]["id"] for i in range(5)] == [0, 1, 2, 3, 4]
    exp = {0: f0[0], 2: f0[1], 4: f0[2], 1: f1[0], 3: f1[1]}
    for i in range(5):
        s = ds[i]
        assert s["input_features"].shape == (8, 3000) and torch.equal(s["input_features"], exp[i])
    assert ds[3]["labels"].tolist() == [8, 7, 6] and ds[1]["text"] == "d"
    batch = AudioCollator(DataConfig(), "whisper")([ds[0], None, ds[3]])
    assert batch["input_features"].shape == (2, 8, 3000) and batch["labels"].tolist() == [[1, 2, -100], [8, 7, 6]]
    size = sum(p.stat().st_size for p in tmp_path.glob("*.bin"))
    assert size == 4 * 8 * sum(whisper_computed_frames(n) for n in lens0 + lens1)       # padding is not stored


def test_speech2text_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    frames = [98, 400, 1]
    f = torch.from_numpy(rng.standard_normal((3, 80, 400)).astype(np.float32))
    w = FeatureCacheWriter(str(tmp_path), "speech2text", 80)
    w.add_batch(["x", "y", "z"], f, [0, 0, 0], frames=frames)
    w.close()
    ds = FeatureCacheDataset(str(tmp_path))
    for i, t in enumerate(frames):
        assert torch.equal(ds[i]["input_features"], f[i, :, :t])
    batch = AudioCollator(DataConfig(), "speech2text", transpose_features=True)([ds[0], ds[1], ds[2]])
    assert batch["input_features"].shape == (3, 400, 80) and batch["attention_mask"].sum(1).tolist() == [98, 400, 1]


def test_augmented_features_are_refused(tmp_path):
    """The padding constant is stored once per clip: SpecAugment output (a zeroed mel row, a time mask inside the
    padding) is not constant behind the last computed tile, and the writer says so instead of reading back wrong."""
    import pytest
    import torch
    from speech_to_text_amd.feature_cache import FeatureCacheWriter
    feats = torch.full((2, 80, 3000), -0.5)
    feats[:, :, :96] = torch.randn(2, 80, 96)
    w = FeatureCacheWriter(str(tmp_path), "whisper", 80)
    w.add_batch([0, 1], feats, [8000, 8000])                        # fine: constant tail
    masked = feats.clone()
    masked[1, 0, :] = 0.0                                           # frequency mask over mel bin 0
    with pytest.raises(ValueError, match="not constant behind"):
        w.add_batch([2, 3], masked, [8000, 8000])
    masked = feats.clone()
    masked[0, :, 2000:2010] = 0.0                                   # time mask inside the padding
    with pytest.raises(ValueError, match="not constant behind"):
        w.add_batch([4, 5], masked, [8000, 8000])


def test_compact_batches_round_trip_and_by_length_chunks(tmp_path):
    """Compact batches (only whisper_compact_frames(lengths) columns, the last of them padding) read back as the same
    30 s tensors as full batches; precompute_corpus(by_length=True) visits every clip once, shortest first."""
    from speech_to_text_amd.feature_cache import whisper_compact_frames
    from speech_to_text_amd.sharding import precompute_corpus
    lens = [16000, 123456, 80000, 1, 7481, 160 * 52 - 200, 160 * 52 - 199]
    assert whisper_compact_frames([160 * 52 - 200]) == 56 and whisper_compact_frames([160 * 52 - 199]) == 60
    assert whisper_compact_frames([]) == 8 and whisper_compact_frames([480000, 3]) == 3000
    full = fake_whisper(lens, seed=5)
    for b, n in enumerate(lens):                       # constant from the first frame that sees no audio, as the kernel's
        full[b, :, (n + 200 + 159) // 160:] = full[b, 0, -1]
    seen = []

    def extract(idx):
        T = whisper_compact_frames([lens[i] for i in idx])
        assert all((lens[i] + 200 + 159) // 160 + 4 <= T for i in idx) and T % 4 == 0
        return full[idx][:, :, :T].clone()

    wc, wf = FeatureCacheWriter(str(tmp_path / "c"), "whisper", 8), FeatureCacheWriter(str(tmp_path / "f"), "whisper", 8)
    for idx, feats in precompute_corpus(lens, 0, 1, 3, extract, balanced=False, by_length=True):
        wc.add_batch(idx, feats, [lens[i] for i in idx])
        wf.add_batch(idx, full[idx], [lens[i] for i in idx])
        seen.extend(idx)
    wc.close(), wf.close()
    assert [lens[i] for i in seen] == sorted(lens) and sorted(seen) == list(range(len(lens)))
    a, b = FeatureCacheDataset(str(tmp_path / "c")), FeatureCacheDataset(str(tmp_path / "f"))
    for i in range(len(lens)):
        assert a[i]["id"] == b[i]["id"] == i
        assert torch.equal(a[i]["input_features"], full[i]) and torch.equal(b[i]["input_features"], full[i])


def test_a_compact_batch_that_breaks_the_contract_is_refused(tmp_path):
    """a compact (n_frames < 3000) batch whose longest clip needs more columns would be stored with a wrong
    padding constant; add_batch knows the lengths and refuses it."""
    import pytest
    import torch
    from speech_to_text_amd.feature_cache import FeatureCacheWriter, whisper_compact_frames
    w = FeatureCacheWriter(str(tmp_path / "c"), "whisper", 80)
    n = [16000, 52000]
    T = whisper_compact_frames(n)
    feats = torch.full((2, 80, T), -0.25)
    w.add_batch([0, 1], feats, n)                                  # fits
    with pytest.raises(ValueError, match="too short for its longest clip"):
        w.add_batch([2, 3], feats[:, :, :T - 4], n)
    w.close()
