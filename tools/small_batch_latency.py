#!/usr/bin/env python3
"""Per-call GPU time (two HIP events around the call, median of 30) of the front-end's entry points on small batches:
what a serving loop sends.  Usage: python3 tools/small_batch_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device

dev = torch.device("cuda:0")


def timed(fn):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(30):
        torch.cuda._sleep(200_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e3


s2t = HipSpeech2TextFeatureExtractor(device=dev)
s2t32 = HipSpeech2TextFeatureExtractor(device=dev, stft_precision="float32")
wh = HipWhisperFeatureExtractor(80, device=dev)
pre = GpuAudioPreprocessor(normalize=True, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
for B in (1, 4, 16, 64):
    for secs in (5, 10):
        n = 16000 * secs
        pcm = synth_batch_device(3, B, n, dev).reshape(-1)
        off = (torch.arange(B, dtype=torch.int64) * n).to(dev)
        ln = torch.full((B,), n, dtype=torch.int32, device=dev)
        t_max = s2t.num_frames(n)
        a = timed(lambda: s2t.extract(pcm, off, ln, t_max))
        b = timed(lambda: s2t32.extract(pcm, off, ln, t_max))
        c = timed(lambda: wh.extract(pcm, off, ln))
        n48 = 48000 * secs
        pcm48 = synth_batch_device(4, B, n48, dev, sample_rate=48000).reshape(-1)
        off48 = (torch.arange(B, dtype=torch.int64) * n48).to(dev)
        ln48 = torch.full((B,), n48, dtype=torch.int32, device=dev)
        host = [n48] * B
        d = timed(lambda: pre.run(pcm48, off48, ln48, host, 1, 48000, sync_lengths=False))
        print(f"B = {B:3d} x {secs:2d} s: Speech2Text float64 {a:7.1f} us, float32 {b:7.1f} us; Whisper-80 (30 s padded) {c:7.1f} us; "
              f"48 -> 16 kHz resample + normalise {d:7.1f} us", flush=True)
