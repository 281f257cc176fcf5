"""The reference's own TorchWhisperFeatureExtractor semantics (row W', src/processors.py:148-246) through
sttfe_whisper_logmel_torch: padding=True (max over 3001 frames, then crop) and padding=False (variable length),
against oracle.whisper.torch_whisper_features.  Tolerance 1e-4 (north_star).  Parity with the reference class itself
is unpinned: it cannot be constructed here (its mel bank needs torchaudio), so the oracle
restates src/processors.py:180-239 with HF's mel bank."""
import numpy as np
import pytest
import torch

from speech_to_text_amd.synth import synth_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def tfe():
    from speech_to_text_amd.processors import TorchWhisperFeatureExtractor
    return TorchWhisperFeatureExtractor(device="cuda:0")


def test_padding_false_matches_oracle(tfe):
    from oracle.whisper import torch_whisper_features
    lens = [16000, 52001, 201, 7680, 7679, 7681, 479999, 480000, 480160, 483680, 240123, 959]
    clips = synth_batch(501, lens)
    for group in (clips[:6], clips[6:], [clips[2]], [clips[9], clips[0]]):
        got = tfe([torch.from_numpy(c) for c in group], sampling_rate=16000, padding=False).input_features
        exp = torch_whisper_features(group, 80, padding=False)
        assert tuple(got.shape) == exp.shape, (got.shape, exp.shape)
        err = np.abs(got.cpu().numpy() - exp).reshape(len(group), -1).max(axis=1)
        assert float(err.max()) <= TOL, err
    # frames past a clip's own are exactly padding_value; the longest clip decides the width (cropped to 3000)
    got = tfe(clips[:3], sampling_rate=16000, padding=False).input_features
    assert got.shape == (3, 80, 1 + 52001 // 160)
    assert float(got[0, :, 1 + 16000 // 160:].abs().max()) == 0.0 and float(got[2, :, 2:].abs().max()) == 0.0
    assert tfe([clips[9]], sampling_rate=16000, padding=False).input_features.shape == (1, 80, 3000)


def test_padding_value_and_errors(tfe):
    from oracle.whisper import torch_whisper_features
    from speech_to_text_amd.processors import TorchWhisperFeatureExtractor
    clips = synth_batch(502, [4000, 12345])
    fe = TorchWhisperFeatureExtractor(device="cuda:0", padding_value=-1.5)
    got = fe(clips, sampling_rate=16000, padding=False, return_tensors="np").input_features
    exp = torch_whisper_features(clips, 80, padding=False, padding_value=-1.5)
    assert isinstance(got, np.ndarray) and float(np.abs(got - exp).max()) <= TOL
    assert np.all(got[0, :, 26:] == -1.5)
    with pytest.raises(ValueError, match="longer than 200 samples"):
        tfe([clips[0][:200]], sampling_rate=16000, padding=False)
    with pytest.raises(ValueError, match="Expected sampling_rate=16000"):
        tfe(clips, sampling_rate=8000, padding=False)


def test_padding_true_takes_the_max_over_3001_frames(tfe):
    """Energy concentrated in the last 200 samples of a full 30 s clip sits almost entirely in STFT frame 3000, which HF
    drops before the per-clip max and the reference class keeps (src/processors.py:206-219): the clamp floors differ."""
    from oracle.whisper import torch_whisper_features, whisper_features
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    rng = np.random.default_rng(7)
    clip = (rng.standard_normal(480000) * 1e-4).astype(np.float32)
    clip[-200:] += (0.9 * np.sin(2 * np.pi * 1000 * np.arange(200) / 16000)).astype(np.float32)
    others = synth_batch(503, [480000, 30000, 480001])
    clips = [clip] + list(others)
    got = tfe(clips, sampling_rate=16000, padding=True).input_features.cpu().numpy()
    exp = torch_whisper_features(clips, 80, padding=True)
    assert got.shape == (4, 80, 3000)
    assert float(np.abs(got - exp).max()) <= TOL
    hf = HipWhisperFeatureExtractor(80, device="cuda:0")(clips, sampling_rate=16000).input_features.cpu().numpy()
    assert float(np.abs(hf - whisper_features(clips, 80)).max()) <= TOL
    # the two semantics really differ on the first clip (its floor comes from frame 3000) and agree on the short one
    assert float(np.abs(got[0] - hf[0]).max()) > 0.01
    assert np.array_equal(got[2], hf[2])


def test_padding_false_clips_longer_than_30s_use_the_max_over_all_their_frames(tfe):
    """VERDICT r2 item 7 / ADVICE: the reference transforms the whole clip, takes the maximum over ALL its frames and
    crops to 3000 afterwards (src/processors.py:116-146, 226-239).  Clips of 30.3 s, 31.5 s and 40 s whose loudest part
    lies BEHIND the 3000th frame: the clamp floor (max - 8) of the kept frames then comes from frames that are not
    stored.  One quiet long clip and short clips in the same batch."""
    from oracle.whisper import torch_whisper_features
    lens = [484000, 504001, 640000, 500000, 16000, 480000]
    clips = synth_batch(601, lens)
    for i in (0, 1, 2):                                     # quiet head, loud tail
        n = lens[i]
        clips[i][:481000] *= np.float32(1e-3)
        clips[i][481000:] = np.clip(clips[i][481000:] * np.float32(3.0), -1, 1)
    got = tfe(clips, sampling_rate=16000, padding=False).input_features.cpu().numpy()
    exp = torch_whisper_features(clips, 80, padding=False)
    assert got.shape == exp.shape == (6, 80, 3000)
    assert float(np.abs(got - exp).max()) <= TOL
    # the tail decided the floor: the head of clip 0 sits on it in many cells
    floor = exp[0].min()
    assert (exp[0] == floor).mean() > 0.05 and float(np.abs(got[0].min() - floor)) <= TOL
