"""The product's AudioCollator keeps the reference's contract (src/data.py:417-529), replaying the
reference's own synthetic collator tests (tests/test_dataloader.py:525-845) and cross-checking every
output against the oracle restatement (oracle/collate.py)."""
import logging
import pickle

import numpy as np
import pytest
import torch

from oracle.collate import collate as oracle_collate
from speech_to_text_amd.config import DataConfig
from speech_to_text_amd.data import AudioCollator


def _samples(seed, times, n_freq=80):
    g = torch.Generator().manual_seed(seed)
    return [{"input_features": torch.randn(n_freq, t, generator=g),
             "labels": torch.randint(0, 1000, (5 + i,), generator=g), "text": f"sample {i}",
             "path": "x.mp3", "duration": 1.0} for i, t in enumerate(times)]


def _np(batch):
    return [None if s is None else {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in s.items()} for s in batch]


def test_whisper_batch_and_label_padding():                      # tests/test_dataloader.py:525-560
    col = AudioCollator(DataConfig(), "whisper", False)
    batch = _samples(0, [3000, 3000, 3000])
    out = col(batch)
    assert out["input_features"].shape == (3, 80, 3000) and out["input_features"].dtype == torch.float32
    assert out["labels"].shape == (3, 7) and out["labels"].dtype == torch.int64
    assert out["labels"][0, 5:].tolist() == [-100, -100]
    assert "attention_mask" not in out and set(out) == {"input_features", "labels", "text"}
    exp = oracle_collate(_np(batch), "cpu", "whisper", False)
    assert np.array_equal(out["input_features"].numpy(), exp["input_features"])
    assert np.array_equal(out["labels"].numpy(), exp["labels"]) and out["text"] == exp["text"]


def test_speech2text_variable_lengths():                         # tests/test_dataloader.py:562-664
    batch = _samples(1, [500, 1500, 2000])
    out = AudioCollator(DataConfig(), "speech2text", True)(batch)
    assert out["input_features"].shape == (3, 2000, 80)
    assert out["attention_mask"].dtype == torch.long and out["attention_mask"].sum(dim=1).tolist() == [500, 1500, 2000]
    assert float(out["input_features"][0, 500:].abs().max()) == 0.0
    exp = oracle_collate(_np(batch), "cpu", "speech2text", True)
    assert np.array_equal(out["input_features"].numpy(), exp["input_features"])
    assert np.array_equal(out["attention_mask"].numpy(), exp["attention_mask"])
    out2 = AudioCollator(DataConfig(), "speech2text", False)(batch)
    assert out2["input_features"].shape == (3, 80, 2000)
    w = AudioCollator(DataConfig(), "whisper", False)(_samples(2, [1000, 2000, 3000]))     # :666-698
    assert w["input_features"].shape == (3, 80, 3000) and "attention_mask" not in w


def test_cuda_mode_raw_audio():                                  # tests/test_dataloader.py:700-736
    cfg = DataConfig(processing_and_augmentation_device="cuda")
    batch = [{"audio": torch.ones(1, n), "text": f"t{n}"} for n in (16000, 24000, 8000)]
    out = AudioCollator(cfg, "whisper")(batch)
    assert out["audio"].shape == (3, 1, 24000) and out["texts"] == ["t16000", "t24000", "t8000"]
    assert float(out["audio"][0, 0, 16000:].abs().max()) == 0.0 and float(out["audio"][2, 0, :8000].min()) == 1.0
    assert out["audio_lengths"].tolist() == [16000, 24000, 8000]
    exp = oracle_collate(_np(batch), "cuda")
    assert np.array_equal(out["audio"].numpy(), exp["audio"])


def test_none_samples_and_errors(caplog):                        # tests/test_dataloader.py:738-760,822-828,1155
    col = AudioCollator(DataConfig(), "whisper", False)
    with caplog.at_level(logging.INFO):
        assert col([None, None]) is None and col([]) is None
    assert "Empty batch after filtering None samples, skipping..." in caplog.text
    out = col([None] + _samples(3, [100]))
    assert out["input_features"].shape == (1, 80, 100)
    with pytest.raises(ValueError, match="Unsupported processing device"):
        AudioCollator(DataConfig(processing_and_augmentation_device="tpu"), "whisper")(_samples(4, [10]))
    assert isinstance(pickle.loads(pickle.dumps(col)), AudioCollator)     # DataLoader workers (:376-382)
