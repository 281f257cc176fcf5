"""Offline feature cache (BASELINE config 5): precomputed features on disk + a Dataset.

The reference recomputes the features of the validation / test splits on every run (``evaluation.py:213-234``
builds the same ``CommonVoiceDataset`` -> MP3 decode -> extractor path as training, without augmentation).
With the GPU front-end a whole split is extracted once (``sharding.precompute_corpus``), written here, and
later epochs / runs read it back through ``FeatureCacheDataset``, whose samples have the keys and shapes of
``CommonVoiceDataset.__getitem__`` in "cpu" mode (``src/data.py:382-390``: ``input_features`` (freq, time),
``labels``, ``text``), so the reference's ``AudioCollator`` "cpu" branch consumes them unchanged.

Layout of a cache directory (plain files, nothing executable)::

    index_rNN.json          one per writer rank NN: {"version", "model_type", "n_mels", "n_frames",
                            "shards": [{"file", "clips": [...]}]}, clip = {"id", "frames", "pad", "offset", "text", "labels"}
    shard_rNN_00000.bin     float32, clip after clip: (n_mels, frames) row-major

Whisper features are constant behind the last computed 48-frame tile of a clip (zero padding to 30 s after
log / clamp / scale), so only ``frames`` columns are stored and ``pad`` is that constant: a 5 s clip costs 0.17 MB
instead of 0.96 MB.  That holds for UN-AUGMENTED features only (the cache is for the evaluation splits): a SpecAugment
frequency mask turns rows of the padding into 0.0 and a time mask can fall inside it, so ``add_batch`` checks that
everything behind ``frames`` is one constant per clip and refuses the batch otherwise.
Speech2Text features are stored whole ((freq, T), T = the clip's frame count).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

VERSION = 1
_TILE, _HOP, _NFFT = 48, 160, 400


def whisper_computed_frames(n_samples: int, n_frames: int = 3000) -> int:
    """Columns of a Whisper clip that are not the padding constant (the kernel's 48-frame tiles that see audio)."""
    n = max(0, min(int(n_samples), _HOP * n_frames))
    tiles = (n + _NFFT // 2 + _HOP * _TILE - 1) // (_HOP * _TILE)
    return min(n_frames, max(tiles, 1) * _TILE)


def whisper_compact_frames(n_samples: Sequence[int], n_frames: int = 3000) -> int:
    """Smallest ``n_frames`` for ``HipWhisperFeatureExtractor.extract(..., n_frames=)`` whose columns equal the 30 s
    result's for every clip of the batch AND whose last 4 columns are padding for every clip: frame t covers samples
    [160 t - 200, 160 t + 200), so clip b is constant from frame ceil((n_b + 200) / 160) on; rounded up to the
    kernel's 4-column groups, plus one group.  ``min(n_frames, .)``: a 30 s clip has no padding column."""
    n = max([0] + [min(int(v), _HOP * n_frames) for v in n_samples])
    first_const = (n + _NFFT // 2 + _HOP - 1) // _HOP
    return min(n_frames, (first_const + 3) // 4 * 4 + 4)


class FeatureCacheWriter:
    def __init__(self, directory: str, model_type: str, n_mels: int, n_frames: int = 3000, shard_clips: int = 1024,
                 rank: int = 0):
        self.dir = directory
        os.makedirs(directory, exist_ok=True)
        self.model_type = model_type.lower()
        if self.model_type not in ("whisper", "speech2text"):
            raise ValueError(f"Unsupported model type: {model_type}")
        self.n_mels, self.n_frames, self.shard_clips, self.rank = int(n_mels), int(n_frames), int(shard_clips), int(rank)
        self.shards: List[Dict[str, Any]] = []
        self._fh = None
        self._cur: Optional[Dict[str, Any]] = None
        self._pos = 0

    def _roll(self):
        if self._fh is not None:
            self._fh.close()
        name = f"shard_r{self.rank:02d}_{len(self.shards):05d}.bin"
        self._cur = {"file": name, "clips": []}
        self.shards.append(self._cur)
        self._fh = open(os.path.join(self.dir, name), "wb")
        self._pos = 0

    def add_batch(self, ids: Sequence[Any], features: torch.Tensor, n_samples: Sequence[int],
                  texts: Optional[Sequence[str]] = None, labels: Optional[Sequence[Sequence[int]]] = None,
                  frames: Optional[Sequence[int]] = None):
        """``features``: (B, n_mels, T) device or host tensor ((B, T, n_mels) time-major Speech2Text batches are
        transposed by the caller).  ``n_samples``: valid 16 kHz samples per clip (Whisper) ; ``frames``: valid
        frame counts (Speech2Text; default: all T)."""
        B = int(features.shape[0])
        if self.model_type == "whisper":
            # a compact batch (extract(..., n_frames=whisper_compact_frames(lengths))) has fewer columns than 3000:
            # its last column is still every clip's padding constant
            T = int(features.shape[2])
            if T < self.n_frames and whisper_compact_frames(n_samples, self.n_frames) > T:
                # the compact contract is the CALLER's promise (the kernel cannot check lengths against n_frames without
                # a synchronisation): here the lengths are on the host, so a batch that breaks it is refused instead of
                # being stored with a wrong padding constant
                raise ValueError(f"compact batch of {T} frames is too short for its longest clip: "
                                 f"whisper_compact_frames(n_samples) = {whisper_compact_frames(n_samples, self.n_frames)}")
            keep = [min(whisper_computed_frames(n, self.n_frames), T) for n in n_samples]
        else:
            keep = [int(f) for f in (frames if frames is not None else [features.shape[2]] * B)]
        t_max = max(keep) if keep else 0
        host = features[:, :, :t_max].to("cpu", non_blocking=False).contiguous().numpy()
        tail = features[:, 0, -1].to("cpu").numpy() if features.shape[2] > 0 else np.zeros(B, np.float32)
        if self.model_type == "whisper" and features.shape[2] > 0:
            # the truncated tail must be ONE constant per clip (true for un-augmented features; SpecAugment output is not)
            kb = torch.as_tensor(keep, device=features.device)
            col = torch.arange(features.shape[2], device=features.device)
            behind = col[None, None, :] >= kb[:, None, None]
            same = (features == features[:, :1, -1:]) | ~behind
            if not bool(same.all()):
                bad = (~same).flatten(1).any(dim=1).nonzero().flatten().tolist()
                raise ValueError(f"clips {bad[:8]} are not constant behind their last computed tile (augmented "
                                 "features?): the cache stores un-augmented features only")
        for b in range(B):
            if self._cur is None or len(self._cur["clips"]) >= self.shard_clips:
                self._roll()
            block = np.ascontiguousarray(host[b, :, :keep[b]], dtype=np.float32)
            self._fh.write(block.tobytes())
            self._cur["clips"].append({
                "id": ids[b] if not isinstance(ids[b], (np.integer,)) else int(ids[b]),
                "frames": keep[b],
                "pad": float(tail[b]) if (self.model_type == "whisper" and keep[b] < self.n_frames) else 0.0,
                "offset": self._pos,
                "text": texts[b] if texts is not None else "",
                "labels": [int(v) for v in labels[b]] if labels is not None else [],
            })
            self._pos += block.size

    def close(self) -> str:
        if self._fh is not None:
            self._fh.close()
            self._fh = None
        path = os.path.join(self.dir, f"index_r{self.rank:02d}.json")
        with open(path, "w") as f:
            json.dump({"version": VERSION, "model_type": self.model_type, "n_mels": self.n_mels,
                       "n_frames": self.n_frames, "shards": self.shards}, f)
        return path


class FeatureCacheDataset(Dataset):
    """Reads every ``index_r*.json`` of a cache directory (one per writer rank).  Samples are ordered by clip id."""

    def __init__(self, directory: str):
        self.dir = directory
        self.entries: List[Dict[str, Any]] = []
        meta = None
        for name in sorted(os.listdir(directory)):
            if not (name.startswith("index_r") and name.endswith(".json")):
                continue
            with open(os.path.join(directory, name)) as f:
                idx = json.load(f)
            if idx.get("version") != VERSION:
                raise ValueError(f"{name}: unsupported cache version {idx.get('version')}")
            head = (idx["model_type"], idx["n_mels"], idx["n_frames"])
            if meta is not None and head != meta:
                raise ValueError(f"{name}: cache written with different settings {head} vs {meta}")
            meta = head
            for sh in idx["shards"]:
                for c in sh["clips"]:
                    self.entries.append(dict(c, file=sh["file"]))
        if meta is None:
            raise FileNotFoundError(f"no index_r*.json in {directory}")
        self.model_type, self.n_mels, self.n_frames = meta
        self.entries.sort(key=lambda c: (str(type(c["id"])), c["id"]))
        self._maps: Dict[str, np.memmap] = {}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_maps"] = {}                      # memmaps are reopened in DataLoader workers
        return d

    def __len__(self) -> int:
        return len(self.entries)

    def __getitem__(self, i: int) -> Dict[str, Any]:
        c = self.entries[i]
        m = self._maps.get(c["file"])
        if m is None:
            m = np.memmap(os.path.join(self.dir, c["file"]), dtype=np.float32, mode="r")
            self._maps[c["file"]] = m
        frames = int(c["frames"])
        block = np.asarray(m[c["offset"]:c["offset"] + self.n_mels * frames]).reshape(self.n_mels, frames)
        if self.model_type == "whisper" and frames < self.n_frames:
            full = np.full((self.n_mels, self.n_frames), np.float32(c["pad"]), dtype=np.float32)
            full[:, :frames] = block
            block = full
        return {"input_features": torch.from_numpy(np.array(block, dtype=np.float32, copy=True)),
                "labels": torch.tensor(c["labels"], dtype=torch.long), "text": c["text"], "id": c["id"]}
