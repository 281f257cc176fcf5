#!/usr/bin/env python3
"""Headline benchmark: Whisper 80-mel log-mel front-end on full 30 s clips (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from a launcher)

One process per GPU.  A "step" = one pass of the hot path over one batch of synthetic clips that already
sit in HBM.  Clips are sharded by rank (weak scaling: every rank processes its own batch; no collective
on the data path).  Rank 0 prints ONE JSON line.  value = audio-seconds processed by all ranks /
max-over-ranks time.

--workload selects the BASELINE.json configuration (default: the headline one):
  whisper80         configs[0] shape at GPU batch: 1024 x 30 s clips @16 kHz -> (1024, 80, 3000)   [headline]
  s2t               configs[1]: Speech2Text fbank + CMVN, 256 ragged 2-10 s clips, dynamic padding (default mode:
                    float64 STFT + NumPy-order CMVN, bit-compatible with the HF NumPy path)
  s2t_f32           the same in the float32-STFT mode (torchaudio-Kaldi-style arithmetic, the faster kernel)
  resample_whisper  configs[2]: 48 kHz Common-Voice-length clips -> resample + normalise -> Whisper-80, batch 512
  resample_whisper_s16  the same with int16 decoded PCM (sttfe_resample_peak_s16; SURVEY 8d: reported as its own line)
  whisper128_aug    configs[3]: Whisper 128-mel + SpecAugment masks, Common-Voice-length clips, batch 1024
  corpus            configs[4]: offline precompute of a 30 000-clip (~43 h) Common-Voice-length corpus, clip i on rank
                    i mod G, chunks of 1024 clips, Whisper-80 features (strong scaling: the corpus is fixed)
  corpus_compact    the same corpus under the compact output contract of the feature cache: chunks in order of length,
                    each extracted with n_frames = whisper_compact_frames(chunk) (its columns are bit-identical to the
                    30 s tensor's, the rest is one constant per clip and is not written)

roofline: dominant kernel of the workload; achieved = algorithmic bytes per launch (SURVEY.md section 8d:
input PCM needed + output written, each once) / its average launch duration, measured with HIP events
recorded on the launch stream inside libsttfe.so (sttfe_timing_*).  peak = 8 TB/s HBM3E.
traffic = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x 2
+ WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), when that file covers the workload.
cpu_baseline: the reference's CPU extractor (HF, torch-CPU path) on a bounded sample of the same
workload, rank 0, N = 1 only.
--streams n (default 1): n batches in flight on n HIP streams; a serial timed region runs first (kernel timings,
serial_ms_per_step), then the same K steps pipelined (ms_per_step and value).  DESIGN.md section 5.0 has the lines.
"""
from __future__ import annotations

import argparse
import contextlib
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
HBM_PEAK_GBS = 8000.0


def cpu_baseline(workload: str, n_mels: int, clips_np):
    """Times the reference CPU extractor on `clips_np` (list of float32 arrays @16 kHz).  ~10-20 s of CPU work."""
    cores = torch.get_num_threads()
    s2t = workload == "s2t"
    try:
        if s2t:
            from transformers import Speech2TextFeatureExtractor
            fe = Speech2TextFeatureExtractor()
            fn = lambda: fe(clips_np, sampling_rate=16000, return_tensors="pt", padding=True).input_features
            what = "transformers.Speech2TextFeatureExtractor (NumPy path)"
        else:
            from transformers import WhisperFeatureExtractor
            fe = WhisperFeatureExtractor(feature_size=n_mels)
            fn = lambda: fe(clips_np, sampling_rate=16000, return_tensors="pt").input_features
            what = "transformers.WhisperFeatureExtractor (torch-CPU path)"
        kind = "reference"
        fn()                                              # warm-up
    except Exception:                                     # transformers missing on the box
        if s2t:
            from oracle.kaldi import speech2text_features
            fn = lambda: speech2text_features(clips_np)
            what = "oracle.kaldi.speech2text_features"
        else:
            from oracle.whisper import whisper_features_f32
            fn = lambda: whisper_features_f32(clips_np, n_mels)
            what = "oracle.whisper.whisper_features_f32"
        kind = "port"
        fn()
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 50:
        t0 = time.perf_counter()
        fn()
        t_total += time.perf_counter() - t0
        reps += 1
    audio_s = sum(len(c) for c in clips_np) / 16000.0 * reps
    return {"value": audio_s / t_total, "unit": "audio-s/s", "cores": cores, "kind": kind,
            "sample": f"{len(clips_np)} clips ({audio_s / reps:.0f} audio-s) x {reps} passes of {what}"}


def computed_frames(n16):
    """Frames of each clip that the Whisper STFT kernel computes and writes: whole 48-frame tiles up to the first tile
    that lies entirely in the zero padding (csrc/whisper_logmel.hip, first_pad_tile_of), at most 3000."""
    n16 = np.asarray(n16, dtype=np.int64)
    return np.minimum(3000, (n16 + 200 + 7679) // 7680 * 48)


def profile_traffic(workload: str, clips_per_launch: float):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (None if not covered)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path)).get(workload)
        if rec and abs(rec["clips_per_launch"] - clips_per_launch) < 0.5:
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def e2e_lines(wl: str, device, n_mels: int, sr: int, seed: int, batches: int = 10):
    """SURVEY.md 8(d) "one end-to-end line including pinned-H2D": host clips -> PinnedIngestRing (host memcpy into a
    pinned slot by a few threads, asynchronous H2D on the ring's stream, the copy of batch k + 1 under the kernels of
    batch k) -> [48 -> 16 kHz resample + peak normalise] -> Whisper features, for float32 and for int16 decoded PCM.
    Measured AFTER the timed region on a bounded number of batches; reported beside `value`, never mixed into it."""
    import torch
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor, PinnedIngestRing
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    from speech_to_text_amd.synth import common_voice_lengths, synth_batch
    B = 256
    rng = np.random.default_rng(seed)
    if wl == "whisper80":
        base = synth_batch(seed, [N_SAMPLES] * 4, sample_rate=sr)
        clips32 = [base[i % 4] for i in range(B)]
    else:
        lens = (common_voice_lengths(rng, 16) * 3).tolist()
        base = synth_batch(seed, lens, sample_rate=sr)
        clips32 = [base[i % 16] for i in range(B)]
    clips16 = [np.clip(np.round(c * 32767.0), -32768, 32767).astype(np.int16) for c in clips32[:16]]
    clips16 = [clips16[i % len(clips16)] for i in range(B)] if wl != "whisper80" else [clips16[i % 4] for i in range(B)]
    fe = HipWhisperFeatureExtractor(n_mels, device=device)
    pre = GpuAudioPreprocessor(normalize=(wl != "whisper80"), trim_silence=False, min_duration=0.0, max_duration=1e9,
                               device=device)
    out = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=device)
    audio_s = sum(len(c) for c in clips32) / sr
    lines = {}
    for name, clips in (("float32", clips32), ("int16", clips16)):
        ring = PinnedIngestRing(device, slots=3, fill_threads=4)

        def run(n):
            tickets = [ring.submit(clips), ring.submit(clips)]
            for k in range(n):
                pcm, off, ln, lens_h, ch = ring.take(tickets.pop(0))
                p = pre.run(pcm, off, ln, lens_h, ch, sr, sync_lengths=False)
                fe.extract(p.pcm, p.offsets, p.lengths, out=out)
                if k + 2 < n:
                    tickets.append(ring.submit(clips))
            torch.cuda.synchronize()

        run(3)
        t0 = time.perf_counter()
        run(batches)
        dt = time.perf_counter() - t0
        wire = sum(c.nbytes for c in clips) * batches
        lines[name] = {"value": audio_s * batches / dt, "unit": "audio-s/s", "pcie_gbs": wire / dt / 1e9,
                       "ms_per_batch": dt / batches * 1e3}
        del ring
    return {"path": "host clips -> PinnedIngestRing (4 fill threads, 3 pinned slots, async H2D) -> "
                    + ("" if wl == "whisper80" else "48 -> 16 kHz resample + peak normalise -> ") + f"Whisper {n_mels}-mel",
            "clips_per_batch": B, "batches": batches, **lines}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, the same command line), relay rank 0's JSON line, return the worst
    exit code.  The parent never touches the GPU and never replaces itself; children are stopped by PID.
    Every rank's stdout and stderr go to files of their own in a temporary directory; the tail of every rank that exits
    non-zero (and of the ranks stopped because of it) is printed to stderr, so that a failure on rank > 0 -- an RCCL
    rendezvous error, a fault on one GPU -- is not lost."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="sttfe_bench_ranks_")
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out_path, err_path = os.path.join(logdir, f"rank{r}.out"), os.path.join(logdir, f"rank{r}.err")
        logs.append((out_path, err_path))
        with open(out_path, "w") as fo, open(err_path, "w") as fe:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=fo, stderr=fe, text=True))
    worst = 0
    failed, stopped = [], []
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                if r in stopped:
                    continue                    # ended by the terminate() below, not a failure of its own
                worst = worst or rc
                failed.append((r, rc))
                for q in pending:               # a dead rank leaves the others waiting at a barrier: stop them
                    if q not in stopped:
                        stopped.append(q)
                        procs[q].terminate()
        time.sleep(0.05)

    def tail(path, n_lines=40):
        try:
            with open(path) as f:
                return "".join(f.readlines()[-n_lines:])
        except OSError:
            return ""

    for r, rc in failed:
        sys.stderr.write(f"[bench.py] rank {r} of {n} exited with code {rc}; last lines of its stderr / stdout:\n"
                         f"{tail(logs[r][1])}{tail(logs[r][0], 10)}")
    for q in stopped:
        sys.stderr.write(f"[bench.py] rank {q} of {n} was stopped after that failure; last lines of its stderr:\n"
                         f"{tail(logs[q][1], 15)}")
    if not failed:
        sys.stderr.write(tail(logs[0][1], 200))          # rank 0's own notices, as when it ran in this process
    sys.stdout.write(tail(logs[0][0], 10 ** 6))
    sys.stdout.flush()
    sys.stderr.flush()
    if failed:
        sys.stderr.write(f"[bench.py] per-rank logs kept in {logdir}\n")
    else:
        import shutil
        shutil.rmtree(logdir, ignore_errors=True)
    return worst


def run_stub(args, world: int, rank: int) -> None:
    """The benchmark's multi-rank control flow with a host-only step (STTFE_BENCH_STUB=1, gloo)."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    audio_seconds = 100.0 * (rank + 1)            # rank-dependent, so that a SUM that misses a rank shows
    step = lambda: time.sleep(0.002 * (rank + 1))

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    fail_rank = os.environ.get("STTFE_BENCH_STUB_FAIL_RANK")          # test hook: this rank dies after the first barrier
    if fail_rank is not None and int(fail_rank) == rank:
        sys.stderr.write(f"stub: rank {rank} fails on purpose (STTFE_BENCH_STUB_FAIL_RANK)\n")
        sys.stderr.flush()
        os._exit(7)
    for _ in range(args.warmup):
        step()
    t_settle = time.perf_counter()
    flag = torch.zeros(1, dtype=torch.int32)
    settle_rounds = 0
    while True:
        done = time.perf_counter() - t_settle >= min(args.settle_seconds, 0.05) * (rank + 1)
        if dist is not None:
            flag.fill_(1 if done else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            done = bool(flag.item())
        if done:
            break
        step()
        settle_rounds += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    audio_total = audio_seconds
    rounds = torch.tensor([settle_rounds], dtype=torch.int64)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        a = torch.tensor([audio_seconds], dtype=torch.float64)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        audio_total = float(a.item())
        lo, hi = rounds.clone(), rounds.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rounds = torch.stack([lo, hi]).reshape(-1)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": audio_total * args.steps / elapsed, "unit": "audio-s/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "audio_total": audio_total, "settle_rounds": [int(v) for v in rounds.tolist()]}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="whisper80",
                    choices=["whisper80", "s2t", "s2t_f32", "resample_whisper", "resample_whisper_s16", "whisper128_aug", "corpus", "corpus_compact"])
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (0 = the workload's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (host clips -> features) lines")
    ap.add_argument("--fused-normalize", action="store_true",
                    help="resample_whisper workloads: peak normalisation folded into the Whisper kernel "
                         "(sttfe_whisper_logmel_scaled) instead of a pass of its own")
    ap.add_argument("--trim-silence", action="store_true",
                    help="resample_whisper workloads: silence trimming on (SURVEY.md 8d: config 3's second line)")
    ap.add_argument("--event-stride", type=int, default=2,
                    help="HIP events bracket every n-th launch of each kernel in the timed region (2: ten timed launches "
                         "at --steps 20): an event pair drains the queue around its launch (a headline step takes 1.035 ms "
                         "without events, 1.109 ms with the four events of its two kernels), so timing every launch "
                         "would cost `value` 7 %%")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps run after the W warm-up steps until this much wall time has passed: the GPU "
                         "clocks of the pool's boxes take ~30 launches to settle (1.20 -> 1.02 ms per launch)")
    ap.add_argument("--streams", type=int, default=1,
                    help="n > 1: consecutive steps go round-robin to n HIP streams, each with an output tensor of its own "
                         "(a serving loop with n batches in flight): the memory-bound kernels of one batch (resample, "
                         "clamp / fill, CMVN) run under the issue-bound feature kernel of the next.  `value` is then the "
                         "pipelined rate; the kernel timings of `roofline` come from a serial timed region run first")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        # not under a launcher: this process (which has not touched the GPU) starts the N ranks itself
        raise SystemExit(spawn_ranks(args.gpus))
    # STTFE_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a one-GPU box -- the ranks share GPU 0 and the
    # barrier / reductions go over gloo (RCCL refuses two ranks on one device).  Never set for a real measurement.
    # STTFE_BENCH_STUB=1: the same control flow (barriers, lock-step settle, MAX / SUM reductions, one JSON line) with
    # no GPU at all -- a step is a short host sleep -- so that tests/test_bench_ranks_cpu.py can cover N > 1 on CPU.
    stub = os.environ.get("STTFE_BENCH_STUB") == "1"
    if stub:
        return run_stub(args, world, rank)
    rehearse = os.environ.get("STTFE_BENCH_REHEARSE") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # STTFE_BENCH_FORCE_DIST=1: a single rank started by torch.distributed.run still forms its (one-rank) RCCL group, so
    # that the nccl branch -- group set-up, barriers, the MAX / SUM reductions on device tensors, tear-down -- can be
    # exercised on a one-GPU box (profiles/r3_bench_rccl_one_rank.json).  Changes nothing that is measured.
    if world > 1 or os.environ.get("STTFE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout is for the ONE JSON
            # line, so fd 1 points at stderr while the group forms (the first barrier creates the communicator)
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=device)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
    red_device = torch.device("cpu") if rehearse else device

    from speech_to_text_amd import _lib
    from speech_to_text_amd.augment import SpecAugmentSampler
    from speech_to_text_amd.config import AugmentationConfig
    from speech_to_text_amd.preprocess import GpuAudioPreprocessor
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor
    from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device

    L = _lib.lib()
    wl = args.workload
    s16 = wl == "resample_whisper_s16"
    if s16:
        wl = "resample_whisper"
    s2t_f32 = wl == "s2t_f32"
    if s2t_f32:
        wl = "s2t"
    compact = wl == "corpus_compact"
    if compact:
        wl = "corpus"
    seed = {"whisper80": 101, "s2t": 102, "resample_whisper": 103, "whisper128_aug": 104, "corpus": 105}[wl] * 1000 + rank
    rng = np.random.default_rng(seed)
    B = args.batch or {"whisper80": 1024, "s2t": 256, "resample_whisper": 512, "whisper128_aug": 1024, "corpus": 1024}[wl]
    if wl == "corpus":
        from speech_to_text_amd.sharding import shard_indices
        all_lens = common_voice_lengths(np.random.default_rng(105), 30000)          # the same corpus on every rank
        mine = shard_indices(30000, rank, world)
        corpus_lens = all_lens[mine]
        if compact:
            # the compact output contract: this rank's clips in order of length (host metadata, known before the pass),
            # every chunk extracted with n_frames = whisper_compact_frames(chunk) instead of 3000
            corpus_lens = np.sort(corpus_lens, kind="stable")
    n_mels = 128 if wl == "whisper128_aug" else 80
    extra_kernels = []
    n_slots = max(args.streams, 1)
    side = [torch.cuda.Stream(device) for _ in range(n_slots)] if n_slots > 1 else []
    pipelined = [False]                    # switched on for the pipelined timed region only

    def stream_of(k):
        return torch.cuda.stream(side[k]) if pipelined[0] else contextlib.nullcontext()

    step_bytes = None                      # algorithmic bytes of a whole step when they differ from the dominant kernel's

    # ------- synthetic inputs resident in HBM (clip-sharded corpus: rank r owns its own B clips) -------
    if wl == "whisper80":
        lens = np.full(B, N_SAMPLES, dtype=np.int64)
        sr = 16000
    elif wl == "s2t":
        lens = rng.integers(32000, 160001, size=B)
        sr = 16000
    elif wl == "resample_whisper":
        lens = common_voice_lengths(rng, B) * 3                     # decoded PCM at 48 kHz
        sr = 48000
    elif wl == "corpus":
        lens = corpus_lens
        sr = 16000
    else:
        lens = common_voice_lengths(rng, B)
        sr = 16000
    n_max = (int(lens.max()) + 63) // 64 * 64
    pcm2d = synth_batch_device(seed, len(lens), n_max, device, sample_rate=sr)
    pcm = pcm2d.reshape(-1)
    if s16:                                        # decoded PCM as the decoder delivers it
        pcm = torch.clamp(torch.round(pcm * 32767.0), -32768, 32767).to(torch.int16)
    offsets = (torch.arange(len(lens), dtype=torch.int64) * n_max).to(device)
    lengths = torch.tensor(lens, dtype=torch.int32, device=device)
    host_lens = [int(v) for v in lens]
    audio_seconds = float(lens.sum()) / sr

    if wl == "corpus":
        # this rank's shard of the corpus, chunk by chunk; features of a chunk overwrite the previous chunk's
        # (an offline precompute would hand them to the writer here)
        fe = HipWhisperFeatureExtractor(n_mels, device=device)
        outs = [torch.empty((B, n_mels, 3000), dtype=torch.float32, device=device) for _ in range(n_slots)]
        kernel, kname = "whisper_stft_mel", "whisper_stft_mel_kernel"
        n_local = len(lens)
        if compact:
            from speech_to_text_amd.feature_cache import whisper_compact_frames
            chunk_T = {c0: whisper_compact_frames(host_lens[c0:c0 + B]) for c0 in range(0, n_local, B)}
        else:
            chunk_T = {c0: 3000 for c0 in range(0, n_local, B)}
        out_cols = sum(T * (min(c0 + B, n_local) - c0) for c0, T in chunk_T.items())
        step_bytes = float(4 * np.minimum(lens, N_SAMPLES).sum() + 4 * n_mels * out_cols)   # per pass over the shard
        # roofline.frac credits the STFT kernel with what IT moves: its input and the columns of the tiles it computes
        # (the padding behind them is written by the clamp / fill kernel and counts in whole_step_frac only)
        own_cols = sum(int(np.minimum(computed_frames(lens[c0:min(c0 + B, n_local)]), T).sum()) for c0, T in chunk_T.items())
        bytes_per_launch = float(4 * np.minimum(lens, N_SAMPLES).sum() + 4 * n_mels * own_cols)
        chunk = [0]

        def step(slot=0):
            for c0 in range(0, n_local, B):
                c1 = min(c0 + B, n_local)
                k = chunk[0] % n_slots                       # n_slots > 1: the chunks are the batches in flight
                chunk[0] += 1
                T = chunk_T[c0]
                with stream_of(k):
                    fe.extract(pcm, offsets[c0:c1], lengths[c0:c1],
                               out=outs[k].view(-1)[:(c1 - c0) * n_mels * T].view(c1 - c0, n_mels, T))
    elif wl in ("whisper80", "resample_whisper", "whisper128_aug"):
        fe = HipWhisperFeatureExtractor(n_mels, device=device)
        outs = [torch.empty((B, n_mels, 3000), dtype=torch.float32, device=device) for _ in range(n_slots)]
        kernel, kname = "whisper_stft_mel", "whisper_stft_mel_kernel"
        valid16 = np.minimum(-(-lens // 3) if sr == 48000 else lens, N_SAMPLES)
        # roofline.frac: the bytes the STFT kernel itself moves -- its 16 kHz input and the columns of the tiles it
        # computes; the 30 s padding behind them is the clamp / fill kernel's and counts in whole_step_frac only
        # (crediting the kernel with the whole tensor flattered it on ragged batches)
        bytes_per_launch = float(4 * valid16.sum() + 4 * n_mels * computed_frames(valid16).sum())
        step_bytes = float(4 * valid16.sum() + 4 * n_mels * 3000 * B)
        if wl == "resample_whisper":
            # the whole step's algorithmic bytes count the input at the SOURCE rate: 4 n_in (2 n_in for
            # int16 PCM) + the feature tensor; the 16 kHz intermediate is not algorithmic traffic
            step_bytes = float((2 if s16 else 4) * lens.sum() + 4 * n_mels * 3000 * B)
        if wl == "whisper80":
            step = lambda slot=0: fe.extract(pcm, offsets, lengths, out=outs[slot])
        elif wl == "whisper128_aug":
            sampler = SpecAugmentSampler(AugmentationConfig(enabled=True))
            torch.manual_seed(104)
            import random
            random.seed(104)
            table = sampler.mask_table([(n_mels, 3000)] * B).to(device)      # drawn once: RNG is host work
            step = lambda slot=0: fe.extract(pcm, offsets, lengths, mask_table=table, out=outs[slot])
        else:
            pre = GpuAudioPreprocessor(normalize=True, trim_silence=args.trim_silence, min_duration=0.0, max_duration=1e9,
                                       device=device)
            extra_kernels = ["resample_peak", "normalize_trim"]

            if args.fused_normalize:
                # rows P2 + W fused: the normalise pass disappears into the feature kernel's epilogue
                extra_kernels = ["resample_peak"]

                def step(slot=0):
                    p = pre.run(pcm, offsets, lengths, host_lens, 1, 48000, defer_normalize=True, sync_lengths=False)
                    fe.extract(p.pcm, p.offsets, p.lengths, out=outs[slot], clip_scale=p.scale)
            else:
                def step(slot=0):
                    p = pre.run(pcm, offsets, lengths, host_lens, 1, 48000, sync_lengths=False)   # no duration gate: no D2H
                    fe.extract(p.pcm, p.offsets, p.lengths, out=outs[slot])
    else:
        fe = HipSpeech2TextFeatureExtractor(device=device, stft_precision="float32" if s2t_f32 else "float64")
        t_max = max(fe.num_frames(n) for n in host_lens)
        kernel, kname = "kaldi_fbank", ("kaldi_fbank_kernel" if s2t_f32 else "kaldi_fbank64_kernel")
        extra_kernels = ["kaldi_cmvn"]
        bytes_per_launch = float(4 * lens.sum() + (4 * 80 + 4) * t_max * B)
        step = lambda slot=0: fe.extract(pcm, offsets, lengths, t_max)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        if wl == "corpus" or not pipelined[0]:
            for _ in range(n):
                step()                                   # corpus: the chunks pick their own streams
        else:
            for i in range(n):
                with stream_of(i % n_slots):
                    step(i % n_slots)

    # The event pool is created before the warm-up (and parked with a huge stride): creating it between the settle
    # phase and the timed region left the GPU idle for 2 ms, after which the next ~15 launches ran up to 10 % slower
    # (rocprofv3 trace: 0.98 ms per launch before the gap, 1.02 .. 1.10 .. 1.00 ms after it) -- the clocks drop that fast.
    _lib.check(L.sttfe_timing_enable(8192))
    _lib.check(L.sttfe_timing_set_stride(1 << 30))
    barrier()                    # N > 1: the communicator is set up here, not in front of the timed region
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # the same K steps straight after the W warm-up steps, before the settle phase: what a caller that runs one short
    # burst sees (clocks not settled yet); reported as `cold_ms_per_step`, never as `value`
    barrier()
    t_cold = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    cold_elapsed = time.perf_counter() - t_cold
    t_settle = time.perf_counter()
    flag = torch.zeros(1, dtype=torch.int32, device=red_device) if dist is not None else None
    while True:                                                       # untimed, like the warm-up steps
        done = time.perf_counter() - t_settle >= args.settle_seconds
        if dist is not None:
            # the ranks leave the settle phase in the same iteration, so that none of them idles at the barrier
            # below while the others finish (an idle GPU drops its clocks within a millisecond or two)
            flag.fill_(1 if done else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            done = bool(flag.item())
        if done:
            break
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    _lib.check(L.sttfe_timing_reset())
    _lib.check(L.sttfe_timing_set_stride(max(args.event_stride, 1)))
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = serial_elapsed = time.perf_counter() - t0
    if n_slots > 1:
        # the pipelined timed region: no events (a pair drains the queues around its launch), a few untimed steps to
        # fill the pipeline, then the same K steps round-robin over the streams
        _lib.check(L.sttfe_timing_set_stride(1 << 30))
        pipelined[0] = True
        run_steps(max(2 * n_slots, 4))
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        pipelined[0] = False
    if dist is not None:
        t = torch.tensor([elapsed, cold_elapsed, serial_elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold_elapsed, serial_elapsed = float(t[0].item()), float(t[1].item()), float(t[2].item())
        a = torch.tensor([audio_seconds], dtype=torch.float64, device=red_device)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        audio_total = float(a.item())
    else:
        audio_total = audio_seconds

    def query(name):
        ms, n = ctypes.c_double(), ctypes.c_int()
        _lib.check(L.sttfe_timing_query(name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    others = {k: query(k) for k in (["whisper_clamp_fill"] if kernel == "whisper_stft_mel" else []) + extra_kernels}
    if kernel == "whisper_stft_mel":
        # Calls that qualify for the 12-wave kernel of uniform batches (80 bins, no masks, no scale) launch both STFT kernels
        # and the lengths on the device decide which one works; the other returns at once.  The one that did the work is the
        # dominant kernel of the line, the empty launch is listed beside it.
        # (batches of up to 3072 such clips are computed by the 12-wave kernel alone, uniform or ragged: one STFT launch)
        t12, n12 = query("whisper_stft_mel12")
        t1, n1 = query("whisper_stft_mel")
        if n12 > 0 and t12 / n12 > t1 / max(n1, 1):
            kernel, kname = "whisper_stft_mel12", "whisper_stft_mel12_kernel"
            if n1 > 0:
                others["whisper_stft_mel (returns at once: the batch is uniform)"] = (t1, n1)
        elif n12 > 0:
            others["whisper_stft_mel12 (returns at once: the batch is not uniform)"] = (t12, n12)
    tot_ms, timed_launches = query(kernel)
    seen = ctypes.c_int()
    _lib.check(L.sttfe_timing_seen(kernel.encode(), ctypes.byref(seen)))
    launches = seen.value
    _lib.check(L.sttfe_timing_set_stride(1))
    _lib.check(L.sttfe_timing_enable(0))

    if rank == 0:
        n_clips_step = len(lens)
        clips_per_launch = n_clips_step * args.steps / max(launches, 1)
        avg_ms = tot_ms / max(timed_launches, 1)
        achieved = bytes_per_launch * (clips_per_launch / n_clips_step) / (avg_ms * 1e-3) / 1e9
        names = {"whisper80": "log-mel audio-sec/sec (80-mel, 30 s clips)",
                 "s2t": "Speech2Text fbank+CMVN audio-sec/sec (80-mel, 2-10 s clips)",
                 "resample_whisper": "48->16 kHz resample + log-mel audio-sec/sec (80-mel, Common-Voice-length clips)",
                 "whisper128_aug": "log-mel + SpecAugment audio-sec/sec (128-mel, Common-Voice-length clips)",
                 "corpus": "offline log-mel precompute audio-sec/sec (80-mel, 30 000 Common-Voice-length clips)"}
        shapes = {"whisper80": f"Whisper 80-mel log-mel, {B} x 30 s clips @16 kHz per GPU (BASELINE configs[0] shape "
                               "at GPU batch; inputs resident in HBM)",
                  "s2t": f"BASELINE configs[1]: Speech2Text fbank + utterance CMVN, {B} ragged 2-10 s clips, dynamic padding",
                  "resample_whisper": f"BASELINE configs[2]: 48 kHz PCM -> resample + peak normalise -> Whisper 80-mel, {B} clips",
                  "whisper128_aug": f"BASELINE configs[3]: Whisper 128-mel + SpecAugment masks, {B} clips",
                  "corpus": f"BASELINE configs[4]: 30 000-clip (~43 h) corpus, clip i on rank i mod {world}, chunks of {B}"
                            + (" in order of length, compact output (frames + padding constant per clip, no 30 s padding written)"
                               if compact else "")}
        result = {
            "metric": names[wl],
            "value": audio_total * args.steps / elapsed,
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_s": args.settle_seconds,
            "ms_per_step": elapsed / args.steps * 1e3,
            "cold_ms_per_step": cold_elapsed / args.steps * 1e3,
            "streams": n_slots,
            "serial_ms_per_step": serial_elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if wl == "corpus" else "weak",
            "vs_baseline": None,
            "dtype": "f64" if (wl == "s2t" and not s2t_f32) else "f32",
            "data": "synthetic",
            "config": {"workload": shapes[wl] + (" (int16 decoded PCM)" if s16 else "")
                                   + ((" (float32 STFT mode)" if s2t_f32 else " (float64 STFT + NumPy-order CMVN: HF NumPy-path parity mode)") if wl == "s2t" else "")
                                   + (" (normalisation fused into the feature kernel)" if args.fused_normalize and wl == "resample_whisper" else "")
                                   + (" (silence trimming on)" if args.trim_silence and wl == "resample_whisper" else ""), "clips_per_gpu": B, "n_mels": n_mels,
                       "sharding": f"clip-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": profile_traffic(args.workload + ("_fused" if args.fused_normalize else ""), clips_per_launch),
                         "kernel": kname, "avg_launch_ms": avg_ms, "clips_per_launch": clips_per_launch,
                         "launches": launches, "timed_launches": timed_launches, "event_stride": max(args.event_stride, 1),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "other_kernels_avg_ms": {k: (v[0] / max(v[1], 1)) for k, v in others.items()},
                         "step_algorithmic_bytes": step_bytes if step_bytes is not None else bytes_per_launch,
                         "whole_step_frac": (step_bytes if step_bytes is not None else bytes_per_launch)
                                            * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            n_s = 32 if wl != "s2t" else 16
            div = 3 if sr == 48000 else 1
            sample = []
            for i in range(min(n_s, len(lens))):
                c = pcm2d[i, :host_lens[i]].cpu().numpy()
                sample.append(np.ascontiguousarray(c[::div]) if div > 1 else c)     # CPU side gets 16 kHz audio
            result["cpu_baseline"] = cpu_baseline(wl, n_mels, sample)
        else:
            result["cpu_baseline"] = None
        if world == 1 and not args.no_e2e and wl in ("whisper80", "resample_whisper"):
            result["e2e"] = e2e_lines(wl, device, n_mels, sr, seed)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
