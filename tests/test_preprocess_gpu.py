"""GPU parity of rows P1 / R / P2 / P3 / P4 (mono mix, resample, peak normalise, trim, gate) through
the C ABI vs the oracle.

Tolerances: the resampler accumulates 41 (48 kHz) / 28 (32 kHz) / ~35 (44.1 kHz) float32 products;
the oracle accumulates in float64 -> |d| <= 2e-6 for |x| <= 1 (stated: 41 * 2^-24 * max|x| * sum|h|).
Peak normalisation and the trim bounds are bit-exact given the same resampled samples (IEEE float32
division; integer indices).  torchaudio itself is absent: parity of the filter with a real install is
UNPINNED (oracle/resample.py)."""
import numpy as np
import pytest
import torch

from oracle import resample as R
from speech_to_text_amd.synth import synth_clip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make(sr, n, seed, zero_edges=0):
    x = synth_clip(np.random.default_rng(seed), n, sample_rate=sr, zero_block=False)
    if zero_edges:
        x[:zero_edges] = 0
        x[-2 * zero_edges:] *= 1e-4
    return x


@pytest.mark.parametrize("sr", [48000, 32000, 44100, 96000, 8000, 24000, 22050, 11025])
def test_resample_matches_oracle(sr):
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    lens = [sr // 2 + 5, 3 * sr + 1, 1, 2, 3, 40, 41, 4097, 1024 * 3 * 2, 1024 * 3 * 2 + 1]
    clips = [make(sr, n, 100 + i) for i, n in enumerate(lens)]
    p = pre.preprocess_batch(clips, sr)
    assert p.sample_rate == 16000
    peaks = pre.last_peaks.cpu().numpy()
    for i, c in enumerate(clips):
        exp = R.resample(c, sr, 16000)
        assert p.host_lengths[i] == exp.shape[0]
        got = p.clip(i)[0].cpu().numpy()
        assert np.abs(got - exp).max() <= 2e-6, (sr, lens[i], float(np.abs(got - exp).max()))
        assert peaks[i] == np.abs(got).max()
    assert p.lengths.cpu().tolist() == p.host_lengths


@pytest.mark.parametrize("src,dst,channels", [(2062, 2048, 1), (15466, 16000, 1), (16000, 15466, 2), (2048, 2062, 1)])
def test_reduced_rates_beyond_1024_run_without_a_table(src, dst, channels):
    """ratios whose reduced rates exceed 1024 (1031:1024, 7733:8000 -- the speed factor 0.9667 --
    and their inverses) have no tap table (it would be up to 248 MB); resample_onthefly_kernel evaluates the taps where
    they are used.  Held to the oracle, which does build the table, at the tolerance of the tabled kernels."""
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor, ResamplePlan
    plan = ResamplePlan(src, dst)
    assert plan.taps_row == 0 and max(plan.orig, plan.new) > 1024            # header-only plan
    pre = GpuAudioPreprocessor(sample_rate=dst, normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9,
                               device=DEV)
    lens = [3 * plan.orig + 17, plan.orig - 1, 1, 40, 2 * plan.orig]
    rng = np.random.default_rng(9)
    clips = [make(16000, n, 300 + i) for i, n in enumerate(lens)]
    if channels == 2:
        clips = [np.stack([c, (0.5 * rng.standard_normal(len(c))).astype(np.float32)]) for c in clips]
    p = pre.preprocess_batch(clips, src)
    peaks = pre.last_peaks.cpu().numpy()
    for i, c in enumerate(clips):
        exp = R.resample(R.to_mono(c), src, dst)
        assert p.host_lengths[i] == exp.shape[0]
        got = p.clip(i)[0].cpu().numpy()
        assert np.abs(got - exp).max() <= 2e-6, (src, dst, lens[i], float(np.abs(got - exp).max()))
        assert peaks[i] == np.abs(got).max()


def test_same_rate_peak_normalise_bit_exact():
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    clips = [make(16000, n, 7 + i) * np.float32(s) for i, (n, s) in enumerate([(16000, 0.3), (777, 1.0), (5, 1e-3), (48001, 0.9)])]
    clips.append(np.zeros(1000, np.float32))                         # max == 0 -> left unchanged (src/data.py:80)
    p = pre.preprocess_batch(clips, 16000)
    for i, c in enumerate(clips):
        assert np.array_equal(p.clip(i)[0].cpu().numpy(), R.peak_normalize(c)), i


@pytest.mark.parametrize("normalize", [True, False])
def test_trim_bounds_and_gate_exact(normalize):
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=normalize, trim_silence=True, silence_threshold=0.01,
                               min_duration=0.5, max_duration=2.0, device=DEV)
    clips = [make(16000, 24000, 1, zero_edges=3000), make(16000, 16000, 2), np.zeros(16000, np.float32),
             make(16000, 40000, 3), make(16000, 4000, 4), make(16000, 20000, 5, zero_edges=9000)]
    p = pre.preprocess_batch(clips, 16000)
    for i, c in enumerate(clips):
        exp = R.preprocess(c[None, :], 16000, normalize=normalize, trim_silence=True, silence_threshold=0.01,
                           min_duration=0.5, max_duration=2.0)
        assert p.keep[i] == (exp is not None), i
        full = R.preprocess(c[None, :], 16000, normalize=normalize, trim_silence=True, min_duration=0.0, max_duration=1e9)
        assert p.host_lengths[i] == full["audio"].shape[0], i
        assert np.array_equal(p.clip(i)[0].cpu().numpy(), full["audio"]), i
    single = pre.preprocess(torch.from_numpy(clips[0])[None, :], 16000)
    assert single["audio"].shape == (1, p.host_lengths[0]) and single["sample_rate"] == 16000
    assert abs(single["duration"] - p.host_lengths[0] / 16000) < 1e-12
    assert pre.preprocess(clips[3], 16000) is None                   # 2.5 s > max_duration


def test_stereo_48k_full_chain_vs_oracle():
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=True, min_duration=0.0, max_duration=1e9, device=DEV)
    a, b = make(48000, 60000, 21, zero_edges=5000), make(48000, 60000, 22)
    st = np.stack([a, b * np.float32(0.5)])
    p = pre.preprocess_batch([st, st[::-1].copy()], 48000)
    exp = R.preprocess(st, 48000, normalize=True, trim_silence=True, min_duration=0.0, max_duration=1e9)
    for i in range(2):
        got = p.clip(i)[0].cpu().numpy()
        # the trim bounds depend on samples at the threshold: allow them to move by the resampler tolerance
        assert abs(got.shape[0] - exp["audio"].shape[0]) <= 2
        n = min(got.shape[0], exp["audio"].shape[0])
        # align on the end when the starts differ
        assert min(np.abs(got[:n] - exp["audio"][:n]).max(), np.abs(got[-n:] - exp["audio"][-n:]).max()) <= 5e-6


def test_resample_then_whisper_config3_shape():
    """BASELINE config 3 at a size the oracle finishes in seconds: 48 kHz clips -> resample + normalise
    -> Whisper log-mel, everything device-resident, vs oracle chain (tolerance 1e-4)."""
    from oracle.whisper import whisper_features
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    clips = [make(48000, n, 300 + i) for i, n in enumerate([247200, 51840, 507840, 100001])]
    p = pre.preprocess_batch(clips, 48000)
    got = fe.extract(p.pcm, p.offsets, p.lengths).cpu().numpy()
    exp = whisper_features([R.preprocess(c[None, :], 48000, trim_silence=False, min_duration=0.0, max_duration=1e9)["audio"]
                            for c in clips], 80)
    assert np.abs(got - exp).max() <= 1e-4


def test_decimator_item_boundaries_and_both_launch_forms():
    """The 48 -> 16 kHz decimator walks (clip, run of 8 tiles of 1024 outputs) items with persistent workgroups for up to
    2048 clips and one item per workgroup beyond that: clip lengths on and around tile / item boundaries against the
    oracle, and a 2100-clip batch (second form) bit-identical to the same clips in batches of 700 (first form)."""
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    sr, item = 48000, 3 * 1024 * 8
    lens = [item, item + 1, item - 1, 2 * item + 3 * 1024, 3 * 1024, 3 * 1024 + 2, 17 * 3 * 1024 + 7, 5, 3 * item - 2]
    clips = [make(sr, n, 300 + i) for i, n in enumerate(lens)]
    p = pre.preprocess_batch(clips, sr)
    peaks = pre.last_peaks.cpu().numpy()
    for i, c in enumerate(clips):
        exp = R.resample(c, sr, 16000)
        got = p.clip(i)[0].cpu().numpy()
        assert got.shape == exp.shape and np.abs(got - exp).max() <= 2e-6, (lens[i], float(np.abs(got - exp).max()))
        assert peaks[i] == np.abs(got).max()
    rng = np.random.default_rng(9)
    many = [make(sr, int(n), 1000 + i) for i, n in enumerate(rng.integers(1, 9000, size=2100))]
    big = pre.preprocess_batch(many, sr)
    big_peaks = pre.last_peaks.cpu().numpy()
    for c0 in range(0, 2100, 700):
        part = pre.preprocess_batch(many[c0:c0 + 700], sr)
        part_peaks = pre.last_peaks.cpu().numpy()
        for i in range(700):
            assert torch.equal(big.clip(c0 + i)[0], part.clip(i)[0]), c0 + i
        assert np.array_equal(big_peaks[c0:c0 + 700], part_peaks)
    for i in (0, 699, 2099):
        assert np.abs(big.clip(i)[0].cpu().numpy() - R.resample(many[i], sr, 16000)).max() <= 2e-6


def test_decimator_empty_clips_between_others():
    """Clips without samples have no item in the decimator's prefix (it repeats there): their neighbours' items must
    still resolve to the right clip, their output length is 0 and their peak key stays 0."""
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=DEV)
    sr = 48000
    lens = [0, 5000, 0, 0, 3 * 1024 * 8 + 5, 0, 1, 0, 70001, 0]
    n_max = 70016
    host = np.zeros((len(lens), n_max), np.float32)
    for i, n in enumerate(lens):
        if n:
            host[i, :n] = make(sr, n, 500 + i)
    pcm = torch.from_numpy(host).to(DEV).reshape(-1)
    off = (torch.arange(len(lens), dtype=torch.int64) * n_max).to(DEV)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    p = pre.run(pcm, off, ln, lens, 1, sr, sync_lengths=False)
    assert p.lengths.cpu().tolist() == [(n + 2) // 3 for n in lens]
    peaks = pre.last_peaks.cpu().numpy()
    for i, n in enumerate(lens):
        if n == 0:
            assert peaks[i] == 0.0
            continue
        exp = R.resample(host[i, :n], sr, 16000)
        o = int(p.offsets[i].item())
        got = p.pcm[o:o + exp.shape[0]].cpu().numpy()
        assert np.abs(got - exp).max() <= 2e-6, (i, n)
        assert peaks[i] == np.abs(got).max()


def test_decimator_persistent_form_in_a_child_process():
    """STTFE_DECIMATE_PERSIST=1 (read once per process) selects the persistent-workgroup form of the decimator: a child
    process runs it on clips around the item boundaries, float and int16, and holds it to the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import resample as R
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import synth_clip
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device="cuda:0")
item = 3 * 1024 * 8
lens = [item, item + 1, 2 * item + 3 * 1024, 5, 17 * 3 * 1024 + 7, 3 * item - 2, 40000]
clips = [synth_clip(np.random.default_rng(900 + i), n, sample_rate=48000, zero_block=False) for i, n in enumerate(lens)]
for as16 in (False, True):
    src = [np.clip(np.round(c * 32767.0), -32768, 32767).astype(np.int16) for c in clips] if as16 else clips
    p = pre.preprocess_batch(src, 48000)
    for i, c in enumerate(src):
        ref = R.resample(c.astype(np.float32) / 32768.0 if as16 else c, 48000, 16000)
        got = p.clip(i)[0].cpu().numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6, (as16, lens[i])
print("persistent form ok")
'''
    env = dict(os.environ, STTFE_DECIMATE_PERSIST="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "persistent form ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
