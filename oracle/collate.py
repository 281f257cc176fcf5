"""Oracle for rows T, C, C' of SURVEY.md section 8a: transpose, collate, raw-audio collate.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``AudioCollator.__call__`` (src/data.py:442-529) with NumPy:
  src/data.py:455-460   drop None samples; empty batch -> None
  src/data.py:468-477   zero-pad (freq, time) features along time to the batch max -> (B, freq, T_max)
  src/data.py:480-487   speech2text: attention_mask int64, 1 on valid frames
  src/data.py:493-496   transpose_features -> (B, T_max, freq)
  src/data.py:500       labels padded with -100
  src/data.py:519-527   "cuda" branch: zero-pad raw audio -> (B, 1, n_max) + texts
  src/data.py:528-529   anything else -> ValueError("Unsupported processing device ...")

The reference's own synthetic collator tests (tests/test_dataloader.py:525-845) pin
these shapes / dtypes / pad values; tests/test_collate_cpu.py replays them.
"""
from __future__ import annotations

import numpy as np


def collate(batch, device: str = "cpu", model_type: str = "whisper", transpose_features: bool = False):
    batch = [s for s in batch if s is not None]
    if not batch:
        return None
    if device == "cpu":
        feats = [np.asarray(s["input_features"]) for s in batch]          # each (freq, time)
        lengths = [f.shape[1] for f in feats]
        t_max = max(lengths)
        out = np.zeros((len(feats), feats[0].shape[0], t_max), dtype=feats[0].dtype)
        for i, f in enumerate(feats):
            out[i, :, :f.shape[1]] = f
        result = {}
        if model_type.lower() == "speech2text":
            mask = np.ones((len(feats), t_max), dtype=np.int64)
            for i, n in enumerate(lengths):
                mask[i, n:] = 0
            result["attention_mask"] = mask
        if transpose_features:
            out = out.transpose(0, 2, 1)
        labels = [np.asarray(s["labels"]) for s in batch]
        l_max = max(l.shape[0] for l in labels)
        lab = np.full((len(labels), l_max), -100, dtype=np.int64)
        for i, l in enumerate(labels):
            lab[i, :l.shape[0]] = l
        result.update({"input_features": out, "labels": lab,
                       "text": [s.get("text", "") for s in batch]})
        return result
    if device == "cuda":
        audio = [np.asarray(s["audio"]).reshape(-1) for s in batch]
        n_max = max(a.shape[0] for a in audio)
        out = np.zeros((len(audio), 1, n_max), dtype=np.float32)
        for i, a in enumerate(audio):
            out[i, 0, :a.shape[0]] = a
        return {"audio": out, "texts": [s["text"] for s in batch]}
    raise ValueError(f"Unsupported processing device: {device}. Must be 'cpu' or 'cuda'.")
