"""Pins oracle/kaldi.py against the HF Speech2TextFeatureExtractor goldens (CPU only)."""
import numpy as np

from conftest import load_golden
from oracle import kaldi as K

TOL = 2e-5      # same NumPy calls as HF's fallback path -> expected (near) bit-exact


def _cases():
    z = load_golden("s2t_80.npz")
    names = sorted({k.split("__")[0] for k in z.files if k.endswith("__pcm")})
    return z, names


def test_oracle_matches_hf_goldens():
    z, names = _cases()
    for name in names:
        feat, mask, lengths = K.speech2text_features([z[f"{name}__pcm"]])
        exp = z[f"{name}__feat"]
        assert feat.shape[1:] == exp.shape and feat.dtype == np.float32, name
        assert np.array_equal(mask[0], z[f"{name}__mask"]), name
        err = float(np.max(np.abs(feat[0] - exp)))
        assert err <= TOL, (name, err)


def test_truncation_matches_hf():
    z, _ = _cases()
    feat, mask, lengths = K.speech2text_features([z["k2_2p5s_odd__pcm"]], truncation=True, max_length=100)
    assert feat.shape == (1, 100, 80) and lengths[0] == 100
    assert float(np.max(np.abs(feat[0] - z["k2_trunc100__feat"]))) <= TOL
    assert np.array_equal(mask[0], z["k2_trunc100__mask"])


def test_frame_count_and_batch_padding():
    assert K.num_frames(399) == 0 and K.num_frames(400) == 1 and K.num_frames(559) == 1
    assert K.num_frames(560) == 2 and K.num_frames(16000) == 98
    z, _ = _cases()
    clips = [z["k0_0p3s__pcm"], z["k1_1s__pcm"], z["k4_min__pcm"]]
    feat, mask, lengths = K.speech2text_features(clips)
    assert feat.shape == (3, 98, 80)
    assert lengths.tolist() == [28, 98, 3]
    assert mask.sum(axis=1).tolist() == [28, 98, 3]
    assert np.all(feat[0, 28:] == 0) and np.all(feat[2, 3:] == 0)
    # per-clip CMVN: each valid block has zero mean / unit population std per bin
    for i, n in enumerate(lengths):
        blk = feat[i, :n].astype(np.float64)
        assert np.abs(blk.mean(axis=0)).max() < 1e-3
        assert np.abs(blk.std(axis=0) - 1.0).max() < 1e-3


def test_mel_bank_kaldi_properties():
    fb = K.mel_filter_bank_kaldi(80)
    assert fb.shape == (257, 80) and (fb >= 0).all() and fb.max() <= 1.0
    assert int((fb > 0).sum(axis=1).max()) <= 2
