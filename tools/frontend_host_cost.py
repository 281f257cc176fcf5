#!/usr/bin/env python3
"""Wall time of one GpuFrontEnd call on a raw-audio batch (collator "cuda" mode), all augmentations on: how much of it
is host work (RNG draws in the reference's order, packing, H2D) and how much the kernels.
    python3 tools/frontend_host_cost.py [batch] [pitch_mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd import AudioCollator, GpuFrontEnd
from speech_to_text_amd.config import AugmentationConfig, DataConfig
from speech_to_text_amd.synth import common_voice_lengths, synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pitch_mode = sys.argv[2] if len(sys.argv) > 2 else "as_constructed"
lens = common_voice_lengths(np.random.default_rng(7), B).tolist()
clips = synth_batch(7, lens)
samples = [{"audio": torch.from_numpy(c)[None, :], "text": "t"} for c in clips]
cfg = DataConfig(processing_and_augmentation_device="cuda", augmentation=AugmentationConfig(enabled=True))
raw = AudioCollator(cfg, "whisper")(samples)
raw["audio_lengths"] = torch.tensor(lens)
front = GpuFrontEnd(cfg, "whisper", False, split="train", device="cuda:0", pitch_mode=pitch_mode)
audio = float(sum(lens)) / 16000
pinned = dict(raw, audio=raw["audio"].pin_memory())                      # what DataLoader(pin_memory=True) hands over
pinned16 = dict(raw, audio=torch.clamp(torch.round(raw["audio"] * 32767.0), -32768, 32767).to(torch.int16).pin_memory())
for name, batch in (("pageable float32 batch", raw), ("pinned float32 batch (zero-copy ingest)", pinned),
                    ("pinned int16 batch (zero-copy ingest)", pinned16)):
    for _ in range(3):
        front(batch)
    torch.cuda.synchronize()
    N = 10
    t0 = time.perf_counter()
    for _ in range(N):
        out = front(batch)
    t_host = (time.perf_counter() - t0) / N          # the call returns when the host work is done
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / N
    print(f"GpuFrontEnd, {B} Common-Voice-length clips ({audio:.0f} audio-s), all augmentations on, pitch_mode={pitch_mode}, "
          f"{name}: {t_all * 1e3:.2f} ms per batch ({t_host * 1e3:.2f} ms until the call returns) -> "
          f"{audio / t_all / 1e3:.1f} k audio-s/s", flush=True)
raw = pinned
try:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        front(raw)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(18)
except Exception as e:
    print("profile failed", e)
