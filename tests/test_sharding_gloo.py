"""Multi-process (gloo, world_size 2) checks of the clip-sharding path used by bench.py --gpus N and the
corpus precompute driver: partitions are disjoint and complete, and the optional all-gather returns
the batch in clip order.  CPU only: the per-clip work is a stand-in (the kernels are tested on the GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speech_to_text_amd.sharding import all_gather_features, balanced_shards, precompute_corpus, shard_indices


def test_partitions_are_disjoint_and_complete():
    for n, w in [(0, 4), (1, 8), (30000, 8), (17, 3)]:
        parts = [shard_indices(n, r, w) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(n))
    lengths = np.random.default_rng(105).integers(17280, 169280, size=999)
    for w in (1, 2, 4, 8):
        parts = balanced_shards(lengths, w)
        assert sorted(i for p in parts for i in p) == list(range(999))
        loads = [int(lengths[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(lengths.max())
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = np.random.default_rng(7).integers(1000, 9000, size=n_clips)
        feats, owned = [], []
        for idx, out in precompute_corpus(lengths, rank, world, chunk=5,
                                          extract=lambda ids: torch.stack([torch.full((3, 4), float(i)) for i in ids])):
            owned += idx
            feats.append(out)
        local = torch.cat(feats) if feats else torch.zeros((0, 3, 4))
        full = all_gather_features(local, owned, n_clips)
        expect = torch.arange(n_clips, dtype=torch.float32)[:, None, None].expand(n_clips, 3, 4)
        ok = bool(torch.equal(full, expect))
        # max-over-ranks timing reduction as bench.py does it
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, len(owned), float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [23, 1])
def test_gloo_world2_precompute_and_gather(n_clips):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sum(r[2] for r in res) == n_clips
    assert [r[3] for r in res] == [2.0, 2.0]
