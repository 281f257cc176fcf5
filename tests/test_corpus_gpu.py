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


def test_compact_chunks_by_length_equal_the_30s_tensor(corpus, tmp_path):
    """The compact output contract ("frames + padding constant" instead of the 30 s tensor): chunks in
    order of length, each extracted with n_frames = whisper_compact_frames(chunk lengths).  Every column is bit-identical
    to the 3000-frame result, everything behind it equals the chunk tensor's last column, and a feature cache written
    from the compact chunks reads back exactly what one written from the full tensors does."""
    from speech_to_text_amd.feature_cache import FeatureCacheDataset, FeatureCacheWriter, whisper_compact_frames
    fe, lens, pcm2d, offsets, lengths, full = corpus
    pcm = pcm2d.reshape(-1)
    host_lens = [int(v) for v in lens]
    w_compact = FeatureCacheWriter(str(tmp_path / "compact"), "whisper", 80)
    w_full = FeatureCacheWriter(str(tmp_path / "full"), "whisper", 80)
    frames_seen, order = [], []

    def extract(idx):
        T = whisper_compact_frames([host_lens[i] for i in idx])
        frames_seen.append(T)
        sel = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
        return fe.extract(pcm, offsets[sel], lengths[sel], n_frames=T)

    for idx, feats in precompute_corpus(host_lens, 0, 1, CHUNK, extract, balanced=False, by_length=True):
        sel = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
        T = feats.shape[2]
        assert T % 4 == 0 and T < 3000
        assert torch.equal(feats, full[sel][:, :, :T])
        assert bool((full[sel][:, :, T:] == feats[:, :, -1:]).all())
        pos = list(range(0, len(idx), 37))                            # a sample of the chunk goes to the two caches
        keep = [idx[p] for p in pos]
        w_compact.add_batch(keep, feats[pos], [host_lens[i] for i in keep])
        w_full.add_batch(keep, full[torch.as_tensor(keep, dtype=torch.int64, device=DEV)], [host_lens[i] for i in keep])
        order.extend(idx)
    assert sorted(order) == list(range(N_CLIPS))
    assert [host_lens[i] for i in order] == sorted(host_lens)          # chunks in order of length
    assert frames_seen == sorted(frames_seen) and frames_seen[0] < frames_seen[-1]
    w_compact.close()
    w_full.close()
    a, b = FeatureCacheDataset(str(tmp_path / "compact")), FeatureCacheDataset(str(tmp_path / "full"))
    assert len(a) == len(b) > 10
    for i in range(len(a)):
        assert a[i]["id"] == b[i]["id"] and torch.equal(a[i]["input_features"], b[i]["input_features"])
        assert torch.equal(a[i]["input_features"], full[a[i]["id"]].cpu())


def test_compact_frames_rule_and_argument_checks(corpus):
    from speech_to_text_amd.feature_cache import whisper_compact_frames
    fe, lens, pcm2d, offsets, lengths, full = corpus
    assert whisper_compact_frames([0]) == 8 and whisper_compact_frames([1]) == 8
    assert whisper_compact_frames([160 * 4 - 200]) == 8 and whisper_compact_frames([160 * 4 - 199]) == 12
    assert whisper_compact_frames([480000]) == 3000 and whisper_compact_frames([10, 500000]) == 3000
    with pytest.raises(ValueError):
        fe.extract(pcm2d.reshape(-1), offsets[:4], lengths[:4], n_frames=1002)
    with pytest.raises(ValueError):
        fe.extract(pcm2d.reshape(-1), offsets[:4], lengths[:4], n_frames=3004)
    # the compact contract is checked where the lengths are known on the host
    hl = [int(v) for v in lens[:4]]
    with pytest.raises(ValueError, match="too short for the longest clip"):
        fe.extract(pcm2d.reshape(-1), offsets[:4], lengths[:4], n_frames=whisper_compact_frames(hl) - 4, host_lengths=hl)
    fe.extract(pcm2d.reshape(-1), offsets[:4], lengths[:4], n_frames=whisper_compact_frames(hl), host_lengths=hl)
    # the rule is tight at its boundary: clips cut to 160 k - 200 (+1) samples
    for n in (160 * 52 - 200, 160 * 52 - 199, 160 * 50 - 199, 37, 0):
        ln = torch.tensor([n, n // 2], dtype=torch.int32, device=DEV)
        T = whisper_compact_frames([n, n // 2])
        ref = fe.extract(pcm2d.reshape(-1), offsets[:2], ln)
        got = fe.extract(pcm2d.reshape(-1), offsets[:2], ln, n_frames=T)
        assert torch.equal(got, ref[:, :, :T]) and bool((ref[:, :, T:] == got[:, :, -1:]).all()), n


@pytest.mark.parametrize("n_frames", [8, 28, 200, 500, 3000])
def test_short_loud_clips_compact_and_full_against_the_oracle(n_frames):
    """ADVICE r2 (fill pass of the clamp / fill kernel): clips of a few hundred samples have pad_frame * 4 < 128 bytes,
    and compact rows of fewer than 32 frames put several mel rows into one 128-byte line -- the lanes that complete a
    row's last line must not touch the fill segment of the next row or rows of another workgroup.  Loud clips make the
    clamp floor (max - 8) differ from the unclamped -1.5 kernel 1 leaves behind, so a stale cell shows.  Checked against
    the ORACLE (not against the same kernel), several times over: a race would not show on every run."""
    from oracle.whisper import whisper_features
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    rng = np.random.default_rng(77)
    sizes = [0, 37, 199, 200, 201, 333, 440, 441, 600, 999, 1000, 160 * (n_frames - 4) - 200 if n_frames < 3000 else 4000]
    sizes = [min(max(n, 0), max(160 * (n_frames - 4) - 200, 0)) for n in sizes]
    clips = [(0.9 * rng.standard_normal(n)).astype(np.float32).clip(-1, 1) for n in sizes]
    exp = whisper_features(clips, 80)[:, :, :n_frames]
    n_max = 4096 if n_frames == 3000 else max(max(sizes), 64)
    n_max = (n_max + 63) // 64 * 64
    pcm = torch.zeros((len(clips), n_max), dtype=torch.float32, device=DEV)
    for i, c in enumerate(clips):
        pcm[i, :len(c)] = torch.from_numpy(c)
    offsets = (torch.arange(len(clips), dtype=torch.int64) * n_max).to(DEV)
    lengths = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    for _ in range(5):
        got = fe.extract(pcm.reshape(-1), offsets, lengths, n_frames=n_frames).cpu().numpy()
        assert got.shape == exp.shape
        assert float(np.abs(got - exp).max()) <= 1e-4
