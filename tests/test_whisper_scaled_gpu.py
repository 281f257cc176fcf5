"""Rows P2 + W fused (``sttfe_whisper_logmel_scaled``): the peak normalisation x / (max|x| + 1e-8) of
``AudioPreprocessor.preprocess`` (src/data.py:78-81) folded into the Whisper feature kernel as a per-clip floor and
bias of its epilogue, against (a) the unfused chain (normalise, then ``sttfe_whisper_logmel``) and (b) the oracle on
the normalised waveform.  Bars: (a) the two differ by the float32 rounding of s * x per sample, which the float32
STFT turns into a relative change of ~1e-8 of the frame's peak bin: invisible (median < 5e-7, 99 % of cells < 5e-6)
except in cells 70-80 dB below the clip's maximum, where it reaches 4e-5 of normalised log-mel -- so max <= 1e-4 like
row W itself; (b) the 1e-4 of row W."""
import numpy as np
import pytest
import torch

from oracle.whisper import whisper_features
from speech_to_text_amd.processors import HipWhisperFeatureExtractor, pack_clips
from speech_to_text_amd.synth import synth_clip

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def batch():
    rng = np.random.default_rng(77)
    lens = [16000, 480000, 52345, 8000, 123457, 300, 64000]
    amps = [0.5, 0.031, 2.5e-4, 1.0, 7.0, 0.2, 0.0]                     # loud, quiet, very quiet, ..., silent
    clips = [(synth_clip(rng, n, zero_block=(i % 2 == 0)) * np.float32(a)).astype(np.float32)
             for i, (n, a) in enumerate(zip(lens, amps))]
    peaks = np.asarray([np.abs(c).max() for c in clips], np.float32)
    scale = np.where(peaks > 0, np.float32(1.0) / (peaks + np.float32(1e-8)), np.float32(1.0)).astype(np.float32)
    normed = [c / (p + np.float32(1e-8)) if p > 0 else c for c, p in zip(clips, peaks)]   # src/data.py:78-81
    return clips, normed, scale


@pytest.mark.parametrize("n_mels", [80, 128])
def test_scaled_features_equal_normalise_then_extract(n_mels):
    clips, normed, scale = batch()
    fe = HipWhisperFeatureExtractor(n_mels, device=DEV)
    masks = torch.zeros((len(clips), 4, 3), dtype=torch.int32)
    masks[1, 0] = torch.tensor([1, 100, 130])                            # a time mask and a frequency mask
    masks[1, 1] = torch.tensor([0, 10, 25])
    masks[3, 0] = torch.tensor([1, 40, 60])
    pcm, off, ln = pack_clips(clips, DEV, fe.n_samples)
    fused = fe.extract(pcm, off, ln, masks, clip_scale=torch.from_numpy(scale).to(DEV)).cpu().numpy()
    pcm_n, off_n, ln_n = pack_clips(normed, DEV, fe.n_samples)
    plain = fe.extract(pcm_n, off_n, ln_n, masks).cpu().numpy()
    assert np.isfinite(fused).all()
    d = np.abs(fused - plain)
    assert d.max() <= 1e-4 and np.median(d) <= 5e-7 and (d > 5e-6).mean() <= 0.01, (float(d.max()), float(np.median(d)))
    exp = whisper_features(normed, n_mels)
    for b in (0, 2, 4, 5, 6):                                            # the unmasked clips against the oracle
        assert np.abs(fused[b] - exp[b]).max() <= 1e-4, b
    assert np.all(fused[1][:, 100:130] == 0.0) and np.all(fused[1][10:25] == 0.0)


def test_scale_of_one_is_the_plain_entry_point():
    clips, _, _ = batch()
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    pcm, off, ln = pack_clips(clips, DEV, fe.n_samples)
    a = fe.extract(pcm, off, ln)
    b = fe.extract(pcm, off, ln, clip_scale=torch.ones(len(clips), device=DEV))
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        fe.extract(pcm, off, ln, clip_scale=torch.ones(3, device=DEV))


def test_preprocessor_defers_the_normalisation():
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    clips, normed, scale = batch()
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    from speech_to_text_amd.preprocess import pack_planar
    pcm, off, ln, lens, ch = pack_planar(clips, DEV)
    deferred = pre.run(pcm, off, ln, lens, ch, 16000, defer_normalize=True)
    assert np.array_equal(deferred.scale.cpu().numpy(), scale)
    assert np.array_equal(deferred.clip(0)[0].cpu().numpy(), clips[0])                 # samples left as they were
    pcm, off, ln, lens, ch = pack_planar(clips, DEV)
    done = pre.run(pcm, off, ln, lens, ch, 16000)
    assert done.scale is None and np.array_equal(done.clip(0)[0].cpu().numpy(), normed[0])
    a = fe.extract(deferred.pcm, deferred.offsets, deferred.lengths, clip_scale=deferred.scale)
    b = fe.extract(done.pcm, done.offsets, done.lengths)
    d = (a - b).abs()
    assert d.max().item() <= 1e-4 and d.median().item() <= 5e-7
