"""Replays the reference's synthetic collator contracts (tests/test_dataloader.py:525-845) on the oracle."""
import numpy as np
import pytest

from oracle.collate import collate


def _sample(rng, n_freq, t, n_lab):
    return {"input_features": rng.standard_normal((n_freq, t)).astype(np.float32),
            "labels": rng.integers(0, 1000, size=n_lab), "text": f"t{t}"}


def test_whisper_fixed_shape_and_label_pad():
    rng = np.random.default_rng(0)
    batch = [_sample(rng, 80, 3000, n) for n in (5, 9, 7)]
    out = collate(batch, "cpu", "whisper", False)
    assert out["input_features"].shape == (3, 80, 3000) and "attention_mask" not in out
    assert out["labels"].shape == (3, 9) and out["labels"][0, 5:].tolist() == [-100] * 4
    assert set(out) == {"input_features", "labels", "text"}


def test_speech2text_variable_length_mask_and_transpose():
    rng = np.random.default_rng(1)
    batch = [_sample(rng, 80, t, 4) for t in (500, 1500, 2000)]
    out = collate(batch, "cpu", "speech2text", True)
    assert out["input_features"].shape == (3, 2000, 80)
    assert out["attention_mask"].dtype == np.int64
    assert out["attention_mask"].sum(axis=1).tolist() == [500, 1500, 2000]
    assert np.all(out["input_features"][0, 500:] == 0)
    out2 = collate(batch, "cpu", "speech2text", False)
    assert out2["input_features"].shape == (3, 80, 2000)
    assert np.array_equal(out2["input_features"].transpose(0, 2, 1), out["input_features"])


def test_none_filtering_and_errors():
    rng = np.random.default_rng(2)
    assert collate([None, None]) is None and collate([]) is None
    out = collate([None, _sample(rng, 80, 10, 3)], "cpu", "whisper", False)
    assert out["input_features"].shape == (1, 80, 10)
    with pytest.raises(ValueError, match="Unsupported processing device"):
        collate([_sample(rng, 80, 10, 3)], "tpu")


def test_cuda_mode_raw_audio_padding():
    batch = [{"audio": np.ones((1, n), np.float32), "text": str(n)} for n in (16000, 24000, 8000)]
    out = collate(batch, "cuda")
    assert out["audio"].shape == (3, 1, 24000) and out["texts"] == ["16000", "24000", "8000"]
    assert np.all(out["audio"][0, 0, 16000:] == 0) and set(out) == {"audio", "texts"}
