"""BASELINE config 5 (offline corpus precompute, clip-sharded) on the GPU at chunk size 1024: the chunked path
(sharding.precompute_corpus + sliced offsets over one resident PCM buffer, exactly what bench.py --workload corpus
runs) gives the same bits as one unchunked call, and the union over the shards of G = 2 and G = 8 ranks (round-robin
and sample-balanced) equals the G = 1 result.  The unchunked call itself is held to the oracle on a sample of clips."""
import numpy as np
import pytest
import torch

from speech_to_text_amd.sharding import precompute_corpus
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
N_CLIPS, CHUNK = 2148, 1024            # 1024 + 1024 + 100: two full chunks and a ragged last one


@pytest.fixture(scope="module")
def corpus():
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    lens = common_voice_lengths(np.random.default_rng(105), N_CLIPS)
    n_max = (int(lens.max()) + 63) // 64 * 64
    pcm2d = synth_batch_device(105, N_CLIPS, n_max, DEV)
    offsets = (torch.arange(N_CLIPS, dtype=torch.int64) * n_max).to(DEV)
    lengths = torch.tensor(lens, dtype=torch.int32, device=DEV)
    full = fe.extract(pcm2d.reshape(-1), offsets, lengths)           # one call over the whole corpus
    return fe, lens, pcm2d, offsets, lengths, full


def run_shard(corpus, rank, world, balanced):
    fe, lens, pcm2d, offsets, lengths, full = corpus
    pcm = pcm2d.reshape(-1)
    seen = []
    out = torch.empty((CHUNK, 80, 3000), dtype=torch.float32, device=DEV)     # one chunk buffer, reused like bench.py's

    def extract(idx):
        sel = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
        return fe.extract(pcm, offsets[sel], lengths[sel], out=out[:len(idx)])

    for idx, feats in precompute_corpus([int(v) for v in lens], rank, world, CHUNK, extract, balanced=balanced):
        assert len(idx) <= CHUNK
        sel = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
        assert torch.equal(feats, full[sel]), (rank, world, idx[0])
        seen.extend(idx)
    return seen


def test_chunked_equals_unchunked_bit_for_bit(corpus):
    seen = run_shard(corpus, 0, 1, balanced=False)
    assert seen == list(range(N_CLIPS))                                # three chunks, in order
    # contiguous slices of offsets / lengths (bench.py's own chunk loop) as well
    fe, lens, pcm2d, offsets, lengths, full = corpus
    for c0 in range(0, N_CLIPS, CHUNK):
        c1 = min(c0 + CHUNK, N_CLIPS)
        assert torch.equal(fe.extract(pcm2d.reshape(-1), offsets[c0:c1], lengths[c0:c1]), full[c0:c1])


@pytest.mark.parametrize("world,balanced", [(2, False), (8, False), (2, True), (8, True)])
def test_union_over_ranks_equals_one_rank(corpus, world, balanced):
    seen = []
    for rank in range(world):
        seen.extend(run_shard(corpus, rank, world, balanced))
    assert sorted(seen) == list(range(N_CLIPS))                        # every clip exactly once, each bit-identical


def test_unchunked_call_matches_the_oracle_on_a_sample(corpus):
    from oracle.whisper import whisper_features
    fe, lens, pcm2d, offsets, lengths, full = corpus
    pick = [0, 1, 1023, 1024, 2047, 2147]
    clips = [pcm2d[i, :int(lens[i])].cpu().numpy() for i in pick]
    exp = whisper_features(clips, 80)
    got = full[torch.as_tensor(pick, device=DEV)].cpu().numpy()
    assert float(np.abs(got - exp).max()) <= 1e-4
