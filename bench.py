#!/usr/bin/env python3
"""Headline benchmark: Whisper 80-mel log-mel front-end on full 30 s clips (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" = one pass of the hot path (sttfe_whisper_logmel: STFT + mel + log +
per-clip clamp) over one batch of `--batch` synthetic 30 s clips that already sit in HBM.  Clips are
sharded by rank (weak scaling: every rank processes its own batch; no collective on the data path).
Rank 0 prints ONE JSON line.  value = audio-seconds processed by all ranks / max-over-ranks time.

roofline: dominant kernel = whisper_stft_mel_kernel; achieved = algorithmic bytes per launch
(2.88 MB per 80-mel 30 s clip: 1.92 MB PCM read once + 0.96 MB features written once, SURVEY.md
section 8d, times the clips in one launch) / its average launch duration, measured with HIP events
recorded on the launch stream inside libsttfe.so (sttfe_timing_*).  peak = 8 TB/s HBM3E.
cpu_baseline: the reference's CPU extractor (HF WhisperFeatureExtractor, torch-CPU path) on a
bounded sample of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SAMPLES = 480000
CLIP_SECONDS = 30.0
HBM_PEAK_GBS = 8000.0


def cpu_baseline(n_mels: int, clips_np):
    """Times the reference CPU extractor on `clips_np` (list of float32 arrays).  ~10-20 s of CPU work."""
    cores = torch.get_num_threads()
    try:
        from transformers import WhisperFeatureExtractor
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        fn = lambda: fe(clips_np, sampling_rate=16000, return_tensors="pt").input_features
        kind = "reference"
        what = "transformers.WhisperFeatureExtractor (torch-CPU path)"
    except Exception:                                     # transformers missing on the box
        from oracle.whisper import whisper_features_f32
        fn = lambda: whisper_features_f32(clips_np, n_mels)
        kind = "port"
        what = "oracle.whisper.whisper_features_f32"
    fn()                                                  # warm-up
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 50:
        t0 = time.perf_counter()
        fn()
        t_total += time.perf_counter() - t0
        reps += 1
    audio_s = len(clips_np) * CLIP_SECONDS * reps
    return {"value": audio_s / t_total, "unit": "audio-s/s", "cores": cores, "kind": kind,
            "sample": f"{len(clips_np)} x 30 s clips x {reps} passes of {what}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="clips per GPU per step")
    ap.add_argument("--n-mels", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from speech_to_text_amd import _lib
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    from speech_to_text_amd.synth import synth_batch_device

    L = _lib.lib()
    fe = HipWhisperFeatureExtractor(args.n_mels, device=device)
    B = args.batch
    # clip-sharded corpus: rank r owns clips [r*B, (r+1)*B) (seeded per rank)
    pcm2d = synth_batch_device(1000 + rank, B, N_SAMPLES, device)
    pcm = pcm2d.reshape(-1)
    offsets = (torch.arange(B, dtype=torch.int64) * N_SAMPLES).to(device)
    lengths = torch.full((B,), N_SAMPLES, dtype=torch.int32, device=device)
    out = torch.empty((B, args.n_mels, 3000), dtype=torch.float32, device=device)

    def step():
        fe.extract(pcm, offsets, lengths, out=out)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.check(L.sttfe_timing_enable(8192))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    _lib.check(L.sttfe_timing_query(b"whisper_stft_mel", ctypes.byref(tot_ms), ctypes.byref(launches)))
    clamp_ms, clamp_n = ctypes.c_double(), ctypes.c_int()
    _lib.check(L.sttfe_timing_query(b"whisper_clamp_mask", ctypes.byref(clamp_ms), ctypes.byref(clamp_n)))
    _lib.check(L.sttfe_timing_enable(0))

    if rank == 0:
        bytes_per_clip = 4 * N_SAMPLES + 4 * args.n_mels * 3000
        clips_per_launch = B * args.steps / max(launches.value, 1)
        avg_ms = tot_ms.value / max(launches.value, 1)
        achieved = bytes_per_clip * clips_per_launch / (avg_ms * 1e-3) / 1e9
        audio_s = world * B * CLIP_SECONDS * args.steps
        result = {
            "metric": "log-mel audio-sec/sec (80-mel, 30 s clips)" if args.n_mels == 80
                      else f"log-mel audio-sec/sec ({args.n_mels}-mel, 30 s clips)",
            "value": audio_s / elapsed,
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Whisper {args.n_mels}-mel log-mel, {B} x 30 s clips @16 kHz per GPU "
                                   "(BASELINE configs[0] shape at GPU batch; inputs resident in HBM)",
                       "clips_per_gpu": B, "n_mels": args.n_mels, "n_frames": 3000,
                       "sharding": f"clip-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whisper_stft_mel_kernel", "avg_launch_ms": avg_ms,
                         "clips_per_launch": clips_per_launch, "launches": launches.value,
                         "clamp_kernel_avg_ms": clamp_ms.value / max(clamp_n.value, 1),
                         "whole_step_frac": bytes_per_clip * B * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = [pcm2d[i].cpu().numpy() for i in range(32)]
            result["cpu_baseline"] = cpu_baseline(args.n_mels, sample)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
