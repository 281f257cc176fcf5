"""Clip sharding across the GPUs of one node -- no collective on the hot path.

Every stage of the front-end is per clip (the only reductions -- peak, trim bounds, Whisper max, CMVN
mean / std -- stay inside a clip: reference ``src/data.py:79,85-90``, ``src/processors.py:206``,
``HF:...speech_to_text.py:150-155``), so a corpus or a batch is split by clip: one process per GPU, each
owning a disjoint set of clips.  The reference itself has no distributed code at all
(``train.py:1257``: one process, one device); this module is the multi-GPU shape of the same path.

``all_gather_features`` is the one optional exchange step: when a single consumer wants the whole
batch, feature shards are gathered with ``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI
for the "nccl" backend; "gloo" on CPU for tests).  xGMI is point to point, so the direct all-gather
puts each peer's shard on its own link: time ~ shard_bytes / per-link bandwidth.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch


def shard_indices(n_clips: int, rank: int, world_size: int) -> List[int]:
    """Round-robin: clip i -> rank i mod world_size."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_clips, world_size))


def balanced_shards(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Variable-length clips: longest-first greedy assignment balancing the total samples per rank.

    Deterministic (ties go to the lower rank), so every rank computes the same partition locally
    and no communication is needed to agree on it.
    """
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(lengths[i])
    for s in shards:
        s.sort()
    return shards


def all_gather_features(local: torch.Tensor, owned: Sequence[int], n_total: int, group=None) -> torch.Tensor:
    """Gather per-rank feature shards into the full (n_total, ...) batch on every rank, in clip order.

    ``local`` is (n_local, ...) for the clips ``owned`` (global indices, any partition).  Shards are
    padded to the largest shard so that one ``all_gather_into_tensor`` moves everything.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts) if counts else 0
    tail = tuple(local.shape[1:])
    padded = torch.zeros((n_max,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    idx = torch.full((n_max,), -1, dtype=torch.int64, device=local.device)
    idx[:local.shape[0]] = torch.as_tensor(list(owned), dtype=torch.int64, device=local.device)
    all_feat = torch.empty((world * n_max,) + tail, dtype=local.dtype, device=local.device)
    all_idx = torch.empty(world * n_max, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(all_feat, padded, group=group)
    dist.all_gather_into_tensor(all_idx, idx, group=group)
    out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
    valid = all_idx >= 0
    out[all_idx[valid]] = all_feat[valid]
    return out


def precompute_corpus(lengths: Sequence[int], rank: int, world_size: int, chunk: int,
                      extract: Callable[[List[int]], object], balanced: bool = True,
                      by_length: bool = False) -> Iterable[Tuple[List[int], object]]:
    """Offline feature precompute driver (BASELINE config 5): this rank's clips in chunks of ``chunk``.

    ``extract(indices)`` runs the front-end on those clips (whatever it returns is yielded with the
    indices).  With ``balanced`` the partition balances total samples, else it is round-robin.
    ``by_length``: the rank's clips are chunked in order of length (ties by index), so that the clips of a chunk are
    about equally long and a compact extraction (``extract(..., n_frames=whisper_compact_frames(chunk lengths))``)
    writes next to no padding; the yielded indices say which clip is which.
    """
    mine = balanced_shards(lengths, world_size)[rank] if balanced else shard_indices(len(lengths), rank, world_size)
    if by_length:
        mine = sorted(mine, key=lambda i: (int(lengths[i]), i))
    for c0 in range(0, len(mine), chunk):
        idx = mine[c0:c0 + chunk]
        yield idx, extract(idx)
