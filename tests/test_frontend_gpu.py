"""GPU collate mode end to end (rows S, T, C, C'): AudioCollator("cuda") -> GpuFrontEnd == the
reference's per-clip CPU path + "cpu" collate, restated by the oracle.  SpecAugment masks bit-exact
from the same host seeds; features within 1e-4 (Whisper) / the Kaldi bound of tests/test_kaldi_gpu.py."""
import random

import numpy as np
import pytest
import torch

from oracle import specaugment as S
from oracle.collate import collate as oracle_collate
from speech_to_text_amd.config import AugmentationConfig, DataConfig
from speech_to_text_amd.data import AudioCollator, GpuFrontEnd
from speech_to_text_amd.synth import synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class FakeTokenizer:
    """Stands in for the HF tokenizer (host-side, out of scope): text -> ids of its characters."""
    def __call__(self, text, return_tensors="pt", truncation=False, max_length=None):
        ids = [ord(c) % 1000 for c in text]
        if truncation and max_length:
            ids = ids[:max_length]
        return type("Enc", (), {"input_ids": torch.tensor([ids], dtype=torch.long)})()


def raw_batch(seed, lengths):
    clips = synth_batch(seed, lengths)
    samples = [{"audio": torch.from_numpy(c)[None, :], "text": "txt" * (i + 1)} for i, c in enumerate(clips)]
    samples.insert(1, None)                                           # a failed clip (src/data.py:307-321)
    # SpecAugment only, unless a test switches the waveform-domain stages back on
    aug = AugmentationConfig(enabled=True, add_noise=False, speed_perturbation=False, pitch_shift=False,
                             volume_perturbation=False, fade_inout=False)
    cfg = DataConfig(processing_and_augmentation_device="cuda", augmentation=aug)
    return clips, cfg, AudioCollator(cfg, "whisper")(samples)


def test_whisper_train_batch_with_specaugment_bit_exact_masks():
    from oracle.whisper import whisper_features
    clips, cfg, raw = raw_batch(31, [48000, 80000, 16000, 123457, 480000])
    front = GpuFrontEnd(cfg, "whisper", False, tokenizer=FakeTokenizer(), split="train", feature_size=128, device=DEV)
    torch.manual_seed(104)
    random.seed(104)
    out = front(raw)
    torch.manual_seed(104)
    random.seed(104)
    masks = [S.draw_clip_masks(128, 3000) for _ in clips]
    assert any(masks) and not all(masks)
    exp = whisper_features(clips, 128)
    got = out["input_features"].cpu().numpy()
    assert got.shape == (5, 128, 3000)
    for b in range(5):
        e = S.apply_masks(exp[b], masks[b])
        assert np.abs(got[b] - e).max() <= 1e-4
        for axis, s, t in masks[b]:                                   # masked spans are exactly 0.0
            sl = got[b][:, s:t] if axis == S.AXIS_TIME else got[b][s:t, :]
            assert not sl.any()
    assert out["labels"].shape == (5, 15) and out["labels"][0, 3:].tolist() == [-100] * 12
    assert out["text"] == raw["texts"] and "attention_mask" not in out
    # eval split: no augmentation, no RNG consumed
    st = random.getstate()
    ev = GpuFrontEnd(cfg, "whisper", False, split="validation", feature_size=128, device=DEV)(raw)
    assert random.getstate() == st and np.abs(ev["input_features"].cpu().numpy() - exp).max() <= 1e-4
    assert "labels" not in ev


@pytest.mark.parametrize("transpose", [True, False])
def test_speech2text_batch_matches_cpu_collate(transpose):
    from oracle.kaldi import speech2text_features
    clips, cfg, raw = raw_batch(32, [48000, 33333, 80000, 16000])
    cfg.augmentation.spec_augment_probability = 1.0
    front = GpuFrontEnd(cfg, "speech2text", transpose, tokenizer=FakeTokenizer(), split="train", device=DEV,
                        max_source_positions=400, max_target_positions=4)
    torch.manual_seed(5)
    random.seed(5)
    out = front(raw)
    torch.manual_seed(5)
    random.seed(5)
    feats, mask, tl = speech2text_features(clips, max_length=400, truncation=True)
    assert tl.tolist() == [298, 206, 400, 98]
    samples = []
    for b, n in enumerate(tl):
        f = feats[b, :n].T                                            # (freq, time), src/data.py:356-358
        f = S.apply_masks(f, S.draw_clip_masks(80, int(n), probability=1.0))
        samples.append({"input_features": f, "labels": np.arange(3), "text": "x"})
    exp = oracle_collate(samples, "cpu", "speech2text", transpose)
    got = out["input_features"].cpu().numpy()
    assert got.shape == exp["input_features"].shape == ((4, 400, 80) if transpose else (4, 80, 400))
    assert out["attention_mask"].dtype == torch.int64
    assert np.array_equal(out["attention_mask"].cpu().numpy(), exp["attention_mask"])
    zdiff = np.argwhere((got == 0.0) != (exp["input_features"] == 0.0))
    assert zdiff.shape[0] == 0, (zdiff[:8].tolist(), zdiff.shape[0])   # masks + padding: identical zero pattern
    assert np.abs(got - exp["input_features"]).max() <= 5e-3          # Kaldi fp32-vs-fp64 bound, see test_kaldi_gpu.py
    assert np.median(np.abs(got - exp["input_features"])) <= 2e-5
    assert out["labels"].shape == (4, 4)                              # truncated to max_target_positions


def test_reference_batch_without_lengths_and_empty():
    _, cfg, raw = raw_batch(33, [16000, 8000])
    del raw["audio_lengths"]
    front = GpuFrontEnd(cfg, "whisper", split="test", device=DEV)
    out = front(raw)
    assert out["input_features"].shape == (2, 80, 3000) and front(None) is None


def test_whisper_train_batch_with_waveform_and_spec_augmentation():
    """The whole train-split chain of CommonVoiceDataset.__getitem__ (src/data.py:326-361) for a batch: per clip
    the waveform draws + augmentations, the extractor, then the SpecAugment draws -- same generators, same order."""
    from oracle import waveform_aug as W
    from oracle.whisper import whisper_features
    clips, cfg, raw = raw_batch(34, [48000, 80000, 16000, 123457, 30000, 64000])
    a = cfg.augmentation
    a.add_noise = a.speed_perturbation = a.pitch_shift = a.volume_perturbation = a.fade_inout = True
    a.noise_probability = a.speed_probability = a.volume_probability = a.fade_probability = 0.6
    front = GpuFrontEnd(cfg, "whisper", False, split="train", device=DEV, noise_source="host")
    torch.manual_seed(7)
    random.seed(7)
    out = front(raw)
    torch.manual_seed(7)
    random.seed(7)
    got = out["input_features"].cpu().numpy()
    kinds = set()
    for b, c in enumerate(clips):
        d = W.draw_time_domain(a, c.shape[0])
        masks = S.draw_clip_masks(80, 3000)
        kinds |= {k for k in ("noise_factor", "speed_factor", "pitch_steps", "gain", "fade_samples")
                  if getattr(d, k) is not None}
        e = S.apply_masks(whisper_features([W.apply_time_domain(c, d)], 80)[0], masks)
        err = np.abs(got[b] - e)
        if d.pitch_steps in (None, 0):
            assert err.max() <= 1e-4, (b, d.speed_factor, d.gain, d.fade_samples)
        else:
            # the phase vocoder's float32 phase rounding (tests/test_pitch_shift_gpu.py) leaves ~1e-4 of the peak in
            # the waveform, which the weakest mel bins of a frame see as up to a few 1e-3 of normalised log-mel
            assert err.max() <= 5e-3 and err.mean() <= 2e-5, (b, d.pitch_steps, float(err.max()), float(err.mean()))
    assert len(kinds) == 5                                            # the seed exercises every stage


def test_feature_cache_round_trip_of_gpu_features(tmp_path):
    """Offline precompute (BASELINE config 5): GPU features -> cache on disk -> Dataset == the features themselves;
    the cache keeps only the computed tiles, so this also pins `whisper_computed_frames` to the kernel's tiling."""
    from speech_to_text_amd.feature_cache import FeatureCacheDataset, FeatureCacheWriter
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    from speech_to_text_amd.sharding import precompute_corpus
    lens = [16000, 7480, 7481, 480000, 100000, 1, 52000]
    clips = synth_batch(35, lens)
    fe = HipWhisperFeatureExtractor(80, device=DEV)
    direct = fe(clips, sampling_rate=16000, return_tensors="pt").input_features
    w = FeatureCacheWriter(str(tmp_path), "whisper", 80, shard_clips=3)
    for idx, feats in precompute_corpus(lens, 0, 1, 4, lambda ids: fe([clips[i] for i in ids], sampling_rate=16000,
                                                                      return_tensors="pt").input_features):
        w.add_batch(idx, feats, [lens[i] for i in idx], texts=[f"t{i}" for i in idx])
    w.close()
    ds = FeatureCacheDataset(str(tmp_path))
    assert len(ds) == len(lens)
    for i in range(len(lens)):
        assert ds[i]["id"] == i and ds[i]["text"] == f"t{i}"
        assert torch.equal(ds[i]["input_features"], direct[i].cpu()), i


def test_more_masks_than_the_old_limit_and_the_error_path():
    """num_time_masks / num_freq_masks are free config fields (src/config.py:43-44): 5 + 5 masks per clip work and stay
    bit-exact; beyond STTFE_MAX_MASKS (16) the call fails loudly instead of dropping masks."""
    from oracle.whisper import whisper_features
    from speech_to_text_amd import _lib
    clips = synth_batch(33, [48000, 80000, 16000])
    samples = [{"audio": torch.from_numpy(c)[None, :], "text": "t"} for c in clips]

    def front_for(nt, nf):
        aug = AugmentationConfig(enabled=True, add_noise=False, speed_perturbation=False, pitch_shift=False,
                                 volume_perturbation=False, fade_inout=False, spec_augment_probability=1.0,
                                 num_time_masks=nt, num_freq_masks=nf)
        cfg = DataConfig(processing_and_augmentation_device="cuda", augmentation=aug)
        return cfg, GpuFrontEnd(cfg, "whisper", False, split="train", feature_size=80, device=DEV)

    cfg, front = front_for(5, 5)
    raw = AudioCollator(cfg, "whisper")(samples)
    torch.manual_seed(7)
    random.seed(7)
    got = front(raw)["input_features"].cpu().numpy()
    torch.manual_seed(7)
    random.seed(7)
    masks = [S.draw_clip_masks(80, 3000, probability=1.0, num_time_masks=5, num_freq_masks=5) for _ in clips]
    assert all(len(m) == 10 for m in masks)
    exp = whisper_features(clips, 80)
    for b in range(3):
        assert np.abs(got[b] - S.apply_masks(exp[b], masks[b])).max() <= 1e-4
        for axis, s, t in masks[b]:
            assert not (got[b][:, s:t] if axis == S.AXIS_TIME else got[b][s:t, :]).any()
    cfg, front = front_for(9, 8)                                       # 17 masks per clip
    with pytest.raises(_lib.SttfeError, match="invalid argument"):
        front(AudioCollator(cfg, "whisper")(samples))


@pytest.mark.parametrize("split", ["train", "validation"])
def test_pinned_batches_are_read_in_place_and_give_the_same_result(split):
    """Round 3: a batch in pinned memory (DataLoader(pin_memory=True)) is not copied as a padded
    (B, n_max) tensor -- the same-rate copy kernel reads the valid samples over PCIe.  Same features, masks and labels as
    the pageable batch (same seeds), for float32; an int16 batch equals the float32 batch of x / 32768."""
    clips, cfg, raw = raw_batch(33, [48000, 80000, 16000, 123457])
    assert "audio_lengths" in raw
    front = GpuFrontEnd(cfg, "whisper", False, tokenizer=FakeTokenizer(), split=split, device=DEV)

    def run(batch):
        torch.manual_seed(7)
        random.seed(7)
        return front(batch)

    ref = run(raw)
    pinned = dict(raw, audio=raw["audio"].pin_memory())
    got = run(pinned)
    assert torch.equal(got["input_features"], ref["input_features"]) and torch.equal(got["labels"], ref["labels"])
    assert front._held and front._held[-1][1] is pinned["audio"]          # kept alive while the GPU reads it
    s16 = torch.clamp(torch.round(raw["audio"] * 32767.0), -32768, 32767).to(torch.int16)
    got16 = run(dict(raw, audio=s16.pin_memory()))
    ref16 = run(dict(raw, audio=s16.to(torch.float32) / 32768.0))
    assert torch.equal(got16["input_features"], ref16["input_features"])
    torch.cuda.synchronize()
