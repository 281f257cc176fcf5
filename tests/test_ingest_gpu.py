"""Decoded-PCM ingest on the GPU: int16 clips through ``sttfe_resample_peak_s16`` and the
pinned staging ring.  Bar: bit-exact against the float32 path fed with ``x / 32768`` (what ``torchaudio.load``
hands the reference at src/data.py:313) -- the conversion is exact in float32, everything after it is the same code."""
import numpy as np
import pytest
import torch

from speech_to_text_amd.synth import synth_clip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def s16_clips(sr, lens, seed, channels=1):
    out = []
    for i, n in enumerate(lens):
        rows = [np.clip(np.round(synth_clip(np.random.default_rng(seed + 7 * i + c), n, sample_rate=sr, zero_block=False)
                                 * 30000.0), -32768, 32767).astype(np.int16) for c in range(channels)]
        out.append(rows[0] if channels == 1 else np.stack(rows))
    return out


@pytest.mark.parametrize("sr,channels", [(48000, 1), (16000, 1), (44100, 2), (32000, 2)])
def test_int16_path_equals_float_path(sr, channels):
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=True, min_duration=0.0, max_duration=1e9, device=DEV)
    clips = s16_clips(sr, [sr // 3 + 1, 2 * sr + 5, 777, 4097], 40, channels)
    a = pre.preprocess_batch(clips, sr)
    b = pre.preprocess_batch([c.astype(np.float32) / np.float32(32768.0) for c in clips], sr)
    assert a.host_lengths == b.host_lengths and a.keep == b.keep
    for i in range(len(clips)):
        assert torch.equal(a.clip(i), b.clip(i)), i


def test_mixed_dtypes_fall_back_to_float_on_the_host():
    from speech_to_text_amd.preprocess import pack_planar
    c16 = s16_clips(16000, [1000], 3)[0]
    pcm, off, ln, lens, ch = pack_planar([c16, c16.astype(np.float32) / np.float32(32768.0)], torch.device(DEV))
    assert pcm.dtype == torch.float32 and lens == [1000, 1000]
    assert torch.equal(pcm[:1000], pcm[int(off[1]):int(off[1]) + 1000])


def test_pinned_ring_overlaps_and_matches_direct_path():
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor, PinnedIngestRing
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    ring = PinnedIngestRing(DEV, slots=2)
    batches = [s16_clips(48000, [48000, 30000 + 100 * k, 96001], 100 + k) for k in range(5)]     # wraps the ring twice
    tickets = [ring.submit(batches[0]), ring.submit(batches[1])]
    for k in range(5):
        pcm, off, ln, lens, ch = ring.take(tickets.pop(0))
        assert pcm.dtype == torch.int16
        got = pre.run(pcm, off, ln, lens, ch, 48000)
        if k + 2 < 5:
            tickets.append(ring.submit(batches[k + 2]))              # next copy runs under this batch's kernels
        exp = pre.preprocess_batch(batches[k], 48000)
        assert got.host_lengths == exp.host_lengths
        for i in range(3):
            assert torch.equal(got.clip(i), exp.clip(i)), (k, i)
