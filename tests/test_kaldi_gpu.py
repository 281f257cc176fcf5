"""GPU parity of the Speech2Text fbank + CMVN path (row K) through the C ABI: HIP vs oracle vs HF goldens.

Tolerance: BASELINE.json north_star -- max abs error <= 1e-4, flat, on every cell of the raw log-mel and of the
CMVN output, against the HF goldens and against the float64 oracle.  That is the bar of the DEFAULT mode
(``stft_precision="float64"``, STTFE_KALDI_F64): DC removal, pre-emphasis, window and the 512-point DFT in float64,
spectra rounded to complex64 afterwards, which is the arithmetic of HF's NumPy path (HF:audio_utils.py:957-993).

The optional float32 mode (``stft_precision="float32"``; what torchaudio's Kaldi fbank computes when torchaudio is
installed, HF speech_to_text.py:113-121 -- not runnable here, so unpinned) has a float32 STFT noise floor of ~1e-8 of
the frame's strongest FFT bin in amplitude: a mel bin whose power is r times the frame's peak bin power carries a
relative error of ~2e-8 / sqrt(r).  For that mode only, the tests use the element-wise bound

    |d ln mel|  <=  1e-4 + 8e-8 * sqrt(peak_bin_power_of_frame_pair / mel)
    |d cmvn|    <=  (b + mean_t b) / std + |z| * rms_t(b) / std + 2e-5 (1 + |z|)

i.e. 1e-4 wherever the bin is within ~55 dB of the frame peak, looser only below.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from speech_to_text_amd.synth import synth_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def fe():
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
    fe = HipSpeech2TextFeatureExtractor(device="cuda:0")
    assert fe.stft_precision == "float64"          # the default mode is the one held to the flat 1e-4
    return fe


@pytest.fixture(scope="module")
def fe32():
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
    return HipSpeech2TextFeatureExtractor(device="cuda:0", stft_precision="float32")


def test_mel_bank_matches_oracle(fe):
    from oracle.kaldi import mel_filter_bank_kaldi
    assert np.abs(fe.mel_filters - mel_filter_bank_kaldi(80).astype(np.float32)).max() <= 1e-7


def raw_bound(clip, raw_ref):
    """Element-wise bound on |d ln mel| from the float32 STFT noise floor (module docstring)."""
    from oracle.kaldi import kaldi_power
    pk = kaldi_power(clip).max(axis=1)
    # frames 2k and 2k+1 share one complex FFT (z = a + i b), so they share one float32 noise floor
    pk2 = pk.copy()
    n2 = pk.shape[0] // 2 * 2
    pk2[0:n2:2] = np.maximum(pk[0:n2:2], pk[1:n2:2])
    pk2[1:n2:2] = pk2[0:n2:2]
    peak = np.log(np.maximum(pk2[:raw_ref.shape[0], None], 1e-30))
    return 1e-4 + 8e-8 * np.exp(0.5 * np.maximum(peak - raw_ref, 0.0))


def cmvn_bound(bound, raw_ref):
    """Propagates the raw log-mel bound through (x - mean) / std (module docstring)."""
    std = raw_ref.std(axis=0, keepdims=True)
    z = np.abs(raw_ref - raw_ref.mean(axis=0, keepdims=True)) / std
    b_mean = bound.mean(axis=0, keepdims=True)
    b_rms = np.sqrt((bound ** 2).mean(axis=0, keepdims=True))
    return (bound + b_mean) / std + z * b_rms / std + 2e-5 * (1.0 + z)


def check_clip(fe, clip, exp_cmvn, max_length=None):
    """HIP vs oracle / golden for one clip: raw log-mel and CMVN output.  float64 mode: flat 1e-4; float32 mode: the
    element-wise bounds of the module docstring."""
    from oracle.kaldi import kaldi_fbank
    from speech_to_text_amd.processors import pack_clips
    raw_ref = kaldi_fbank(clip).astype(np.float64)
    if max_length:
        raw_ref = raw_ref[:max_length]
    T = raw_ref.shape[0]
    pcm, off, ln = pack_clips([clip], torch.device("cuda:0"))
    raw, _ = fe.extract(pcm, off, ln, T, max_frames=max_length or 0, apply_cmvn=False)
    d_raw = np.abs(raw[0].cpu().numpy() - raw_ref)
    out, _ = fe.extract(pcm, off, ln, T, max_frames=max_length or 0)
    d = np.abs(out[0].cpu().numpy() - exp_cmvn)
    if fe.stft_precision == "float64":
        assert float(d_raw.max()) <= TOL, float(d_raw.max())
        assert float(d.max()) <= TOL, float(d.max())
    else:
        bound = raw_bound(clip, raw_ref)
        assert (d_raw <= bound).all(), float((d_raw / bound).max())
        cb = cmvn_bound(bound, raw_ref)
        assert (d <= cb).all(), float((d / cb).max())
    return float(d_raw.max()), float(d.max()), float(np.median(d))


@pytest.mark.parametrize("mode", ["float64", "float32"])
def test_matches_hf_goldens(mode, fe, fe32):
    fe = fe if mode == "float64" else fe32
    z = load_golden("s2t_80.npz")
    names = sorted({k.split("__")[0] for k in z.files if k.endswith("__pcm")})
    for name in names:
        o = fe(z[f"{name}__pcm"], sampling_rate=16000, return_tensors="pt")
        exp = z[f"{name}__feat"]
        assert o.input_features[0].shape == exp.shape, name
        assert o.attention_mask.dtype == torch.int32
        assert np.array_equal(o.attention_mask[0].cpu().numpy(), z[f"{name}__mask"])
        if mode == "float64":
            err = float(np.abs(o.input_features[0].cpu().numpy() - exp).max())
            assert err <= TOL, (name, err)
        stats = check_clip(fe, z[f"{name}__pcm"], exp)
        print(mode, name, "max |d raw| %.2e  max |d cmvn| %.2e  median %.2e" % stats)
    o = fe(z["k2_2p5s_odd__pcm"], sampling_rate=16000, truncation=True, max_length=100)
    assert o.input_features.shape == (1, 100, 80)
    assert np.array_equal(o.attention_mask[0].cpu().numpy(), z["k2_trunc100__mask"])
    check_clip(fe, z["k2_2p5s_odd__pcm"], z["k2_trunc100__feat"], max_length=100)


def test_error_distribution(fe, fe32):
    """float64 mode: every raw log-mel value and every CMVN output within 1e-4 of the float64 oracle.  float32 mode:
    the looser part of its bound only applies to bins far below the frame peak -- >= 95 % of the raw values and
    >= 90 % of the CMVN outputs within 1e-4, median error below 1e-5."""
    from oracle.kaldi import kaldi_fbank, speech2text_features
    from speech_to_text_amd.processors import pack_clips
    clips = synth_batch(2002, [48000, 80000, 33333])
    pcm, off, ln = pack_clips(list(clips), torch.device("cuda:0"))
    exp, _, tl = speech2text_features(clips)
    for f in (fe, fe32):
        raw, _ = f.extract(pcm, off, ln, 498, apply_cmvn=False)
        raw = raw.cpu().numpy()
        got = f(clips, sampling_rate=16000).input_features.cpu().numpy()
        d_raw = np.concatenate([np.abs(raw[i, :n] - kaldi_fbank(clips[i])).ravel() for i, n in enumerate(tl)])
        d_out = np.concatenate([np.abs(got[i, :n] - exp[i, :n]).ravel() for i, n in enumerate(tl)])
        print(f.stft_precision, "raw: median %.1e p99 %.1e max %.1e within1e-4 %.4f | cmvn: median %.1e p99 %.1e max %.1e "
              "within1e-4 %.4f" % (np.median(d_raw), np.percentile(d_raw, 99), d_raw.max(), (d_raw <= TOL).mean(),
                                   np.median(d_out), np.percentile(d_out, 99), d_out.max(), (d_out <= TOL).mean()))
        if f.stft_precision == "float64":
            assert d_raw.max() <= TOL and d_out.max() <= TOL
        else:
            assert (d_raw <= TOL).mean() >= 0.95 and np.median(d_raw) <= 1e-5
            assert (d_out <= TOL).mean() >= 0.90 and np.median(d_out) <= 2e-5


@pytest.mark.parametrize("mode", ["float64", "float32"])
def test_batch_dynamic_padding_vs_oracle(mode, fe, fe32):
    """BASELINE config 2 shape at a size the oracle finishes in seconds: ragged 2-10 s clips."""
    from oracle.kaldi import kaldi_fbank, speech2text_features
    fe = fe if mode == "float64" else fe32
    rng = np.random.default_rng(102)
    lengths = rng.integers(32000, 160001, size=24).tolist() + [400, 559, 560, 16000, 399, 100]
    clips = synth_batch(102, lengths)
    o = fe(clips, sampling_rate=16000)
    exp, mask, tl = speech2text_features(clips)
    got = o.input_features.cpu().numpy()
    assert got.shape == exp.shape
    assert np.array_equal(o.attention_mask.cpu().numpy(), mask)
    for i, n in enumerate(tl):
        assert np.all(got[i, n:] == 0)
        if n >= 8:
            d = np.abs(got[i, :n] - exp[i, :n])
            if mode == "float64":
                assert float(d.max()) <= TOL, (i, n, float(d.max()))
            else:
                raw_ref = kaldi_fbank(clips[i]).astype(np.float64)
                cb = cmvn_bound(raw_bound(clips[i], raw_ref), raw_ref)
                assert (d <= cb).all(), (i, n, float((d / cb).max()))


def test_raw_fbank_without_cmvn_and_modes_agree(fe, fe32):
    """do_ceptral_normalize=False returns ln(mel) itself (HF:...speech_to_text.py:256-262 skip the normalisation);
    the two STFT precisions agree to the float32 mode's noise floor."""
    from oracle.kaldi import kaldi_fbank
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
    clip = synth_batch(31, [24000])[0]
    raw_fe = HipSpeech2TextFeatureExtractor(device="cuda:0", do_ceptral_normalize=False)
    got = raw_fe(clip, sampling_rate=16000).input_features[0].cpu().numpy()
    assert float(np.abs(got - kaldi_fbank(clip)).max()) <= TOL
    a = fe(clip, sampling_rate=16000).input_features
    b = fe32(clip, sampling_rate=16000).input_features
    assert float((a - b).abs().median()) <= 2e-5


def test_specaugment_masks_and_int64_mask(fe):
    from speech_to_text_amd.processors import pack_clips
    clips = synth_batch(11, [48000, 20000])
    base = fe(clips, sampling_rate=16000).input_features
    tbl = torch.tensor([[[1, 10, 25], [1, 290, 298], [0, 3, 30], [0, 79, 80]],
                        [[1, 0, 0], [1, 100, 119], [0, 0, 0], [0, 40, 41]]], dtype=torch.int32)
    got = fe(clips, sampling_rate=16000, mask_table=tbl).input_features
    exp = base.clone()
    for b in range(2):
        for axis, s, e in tbl[b].tolist():
            if axis == 1:
                exp[b, s:e, :] = 0.0
            else:
                exp[b, :, s:e] = 0.0
    assert torch.equal(got, exp)
    pcm, off, ln = pack_clips(list(clips), torch.device("cuda:0"))
    feats, attn = fe.extract(pcm, off, ln, 298, attention_dtype=torch.int64)
    assert attn.dtype == torch.int64 and attn.sum(dim=1).tolist() == [298, 123]


def test_cmvn_property_full_batch(fe):
    """Size-independent property at BASELINE config 2's batch (256 ragged clips): every valid block
    has zero mean / unit population variance per bin and padding is exactly zero."""
    rng = np.random.default_rng(5)
    lengths = rng.integers(32000, 160001, size=256)
    clips = [torch.randn(int(n), generator=torch.Generator().manual_seed(int(n))) * 0.1 for n in lengths]
    o = fe(clips, sampling_rate=16000)
    f = o.input_features.double()
    m = o.attention_mask.double().unsqueeze(-1)
    n = m.sum(dim=1)
    mean = (f * m).sum(dim=1) / n
    var = ((f - mean.unsqueeze(1)) ** 2 * m).sum(dim=1) / n
    assert float(mean.abs().max()) < 1e-4 and float((var - 1).abs().max()) < 1e-3
    assert float((f * (1 - m)).abs().max()) == 0.0


def test_fused_and_split_cmvn_are_bit_identical_and_long_clips(fe, monkeypatch):
    """float64 mode normalises through one fused kernel (a clip's columns in LDS) when the shapes allow it and through
    the statistics + normalisation kernel pair otherwise (here: forced by the A/B knob, and a 27 s clip whose columns do
    not fit): same additions in the same order, so the outputs are bit-identical; the long clip is checked against the
    oracle."""
    from oracle.kaldi import speech2text_features
    clips = synth_batch(77, [48000, 123457, 16000, 400, 399])
    a = fe(clips, sampling_rate=16000).input_features.cpu()
    monkeypatch.setenv("STTFE_KALDI_SPLIT_CMVN", "1")
    b = fe(clips, sampling_rate=16000).input_features.cpu()
    monkeypatch.delenv("STTFE_KALDI_SPLIT_CMVN")
    assert bool(((a == b) | (a.isnan() & b.isnan())).all())           # the one-frame clip has std 0: NaN, as in HF
    assert not a[:3].isnan().any()
    long_clip = synth_batch(78, [432000])                            # 2698 frames: beyond the LDS form's panel
    got = fe(long_clip, sampling_rate=16000).input_features.cpu().numpy()
    exp, _, tl = speech2text_features(long_clip)
    assert got.shape == exp.shape and tl[0] == 2698
    assert float(np.abs(got - exp).max()) <= TOL
    # the register form's three shapes (8 columns x 8-row blocks up to 1024 frames, 4 columns x 16-row blocks up to 2048
    # and up to 3072) against the split pair, and a 32 s clip that only the split pair takes
    for seed, lens in ((80, [163600, 99999, 16000]), (81, [250000, 327680, 40000, 399]), (82, [432000, 491520, 70001]),
                       (83, [512000, 16000])):
        clips = synth_batch(seed, lens)
        a = fe(clips, sampling_rate=16000).input_features.cpu()
        monkeypatch.setenv("STTFE_KALDI_SPLIT_CMVN", "1")
        b = fe(clips, sampling_rate=16000).input_features.cpu()
        monkeypatch.delenv("STTFE_KALDI_SPLIT_CMVN")
        assert a.shape == b.shape and bool(((a == b) | (a.isnan() & b.isnan())).all()), lens


def test_fused_cmvn_chain_every_block_remainder(fe, monkeypatch):
    """The fused CMVN kernels (columns in registers for clips of up to 1024 frames, the LDS-panel form up to 2048) walk a
    column in eight-row blocks (a DPP operand per row) with up to seven rows left over: clips of 1 ... 72 frames (every
    remainder, with and without full blocks, across the sixteen-block groups and the LDS form's look-ahead) must give the
    bits of the one-thread-per-column kernel pair, which adds row by row.  With SpecAugment masks on both axes too."""
    clips = synth_batch(79, [400 + 160 * (t - 1) for t in range(1, 73)])
    masks = torch.zeros((len(clips), 3, 3), dtype=torch.int32)       # axis 1 = time, 0 = mel (as in the test above)
    masks[:, 0] = torch.tensor([1, 3, 21])
    masks[:, 1] = torch.tensor([0, 7, 18])
    masks[:, 2] = torch.tensor([1, 60, 90])
    for m in (None, masks):
        outs = {}
        for knob in (None, "STTFE_KALDI_LDS_CMVN", "STTFE_KALDI_SPLIT_CMVN"):
            if knob:
                monkeypatch.setenv(knob, "1")
            o = fe(clips, sampling_rate=16000, mask_table=m) if m is not None else fe(clips, sampling_rate=16000)
            outs[knob] = (o.input_features.cpu(), o.attention_mask.cpu())
            if knob:
                monkeypatch.delenv(knob)
        a, am = outs[None]
        assert a.shape[1] == 72
        for knob in ("STTFE_KALDI_LDS_CMVN", "STTFE_KALDI_SPLIT_CMVN"):
            b, bm = outs[knob]
            assert a.shape == b.shape and bool(((a == b) | (a.isnan() & b.isnan())).all()), knob
            assert bool((am == bm).all())
        assert not a[1:].isnan().any()                                # (one frame: std 0, NaN as in HF)


def test_float64_item_sizes_and_chunked_launches_give_the_same_bits(fe):
    """The float64 kernel's work items belong to waves and are 1 ... 8 frame pairs long (sized by the host from the
    batch, STTFE_KALDI_ITEM_PAIRS overrides; read once per process), and batches of more than 1024 clips go as several
    launches: every item size must give the bits of the automatic one (child processes print a digest), and a
    1100-clip batch the bits of its two parts extracted separately."""
    import hashlib
    import os
    import subprocess
    import sys
    lens = [48000, 123457, 16000, 400, 399, 0, 559, 560, 561, 720, 33333, 159999, 160000, 80001, 879, 880, 881]
    clips = synth_batch(91, lens)
    ref = fe(clips, sampling_rate=16000).input_features.cpu().numpy()
    digest = hashlib.sha256(np.nan_to_num(ref, nan=12345.0).tobytes()).hexdigest()
    code = r"""
import hashlib, numpy as np
from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
from speech_to_text_amd.synth import synth_batch
fe = HipSpeech2TextFeatureExtractor(device="cuda:0")
lens = %r
out = fe(synth_batch(91, lens), sampling_rate=16000).input_features.cpu().numpy()
print("digest", hashlib.sha256(np.nan_to_num(out, nan=12345.0).tobytes()).hexdigest())
""" % (lens,)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pairs in ("1", "3", "8", "64"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, STTFE_KALDI_ITEM_PAIRS=pairs),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and ("digest " + digest) in r.stdout, (pairs, r.stdout[-500:], r.stderr[-2000:])
    # more clips than one launch takes
    from speech_to_text_amd.processors import pack_clips
    rng = np.random.default_rng(92)
    many = synth_batch(92, rng.integers(400, 4000, size=1100).tolist())
    pcm, off, ln = pack_clips(list(many), torch.device("cuda:0"))
    t_max = max(fe.num_frames(len(c)) for c in many)
    whole, _ = fe.extract(pcm, off, ln, t_max)
    a, _ = fe.extract(pcm, off[:1024].contiguous(), ln[:1024].contiguous(), t_max)
    b, _ = fe.extract(pcm, off[1024:].contiguous(), ln[1024:].contiguous(), t_max)
    parts = torch.cat([a, b]).cpu()
    whole = whole.cpu()
    assert bool(((whole == parts) | (whole.isnan() & parts.isnan())).all())


@pytest.mark.parametrize("mode", ["float64", "float32"])
@pytest.mark.parametrize("tag,means,varis", [("mean_only", True, False), ("var_only", False, True)])
def test_cmvn_means_or_vars_alone_match_hf_goldens(mode, tag, means, varis, monkeypatch):
    """normalize_means / normalize_vars individually off (HF:...speech_to_text.py:141-163) as two flags
    of the CMVN kernels, against goldens from the real HF extractor.  float64 mode: flat 1e-4 (fused and two-kernel
    forms); float32 mode: the raw log-mel differs by the float32 STFT's noise floor (module docstring), so the mean-only
    output is held to the raw bound and the variance-only output to the raw bound divided by the bin's std."""
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
    from oracle.kaldi import kaldi_fbank
    z = load_golden("s2t_80_cmvn_flags.npz")
    for split in ([False, True] if mode == "float64" else [False]):
        if split:
            monkeypatch.setenv("STTFE_KALDI_SPLIT_CMVN", "1")
        fx = HipSpeech2TextFeatureExtractor(normalize_means=means, normalize_vars=varis, stft_precision=mode, device="cuda:0")
        for name in sorted({k.split("__")[0] for k in z.files if k.endswith("__pcm")}):
            clip, exp = z[f"{name}__pcm"], z[f"{name}__{tag}"]
            got = fx(clip, sampling_rate=16000, return_tensors="pt").input_features[0].cpu().numpy()
            assert got.shape == exp.shape
            d = np.abs(got - exp)
            if mode == "float64":
                assert float(d.max()) <= TOL, (name, tag, split, float(d.max()))
            else:
                raw_ref = kaldi_fbank(clip).astype(np.float64)
                bound = raw_bound(clip, raw_ref)
                std = raw_ref.std(axis=0, keepdims=True)
                b_rms = np.sqrt((bound ** 2).mean(axis=0, keepdims=True))
                if varis:       # y = x / std: d y <= d x / std + |x| / std * (d std / std)
                    zz = np.abs(raw_ref) / std
                    bound = bound / std + zz * b_rms / std + 2e-5 * (1.0 + zz)
                else:           # y = x - mean
                    bound = bound + bound.mean(axis=0, keepdims=True) + 2e-5
                assert (d <= bound).all(), (name, tag, float((d / bound).max()))
    # both off is "no CMVN"
    fx = HipSpeech2TextFeatureExtractor(normalize_means=False, normalize_vars=False, device="cuda:0")
    clip = z["c0_0p7s__pcm"]
    got = fx(clip, sampling_rate=16000, return_tensors="pt").input_features[0].cpu().numpy()
    assert float(np.abs(got - kaldi_fbank(clip)).max()) <= TOL
