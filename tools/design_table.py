#!/usr/bin/env python3
"""Print the rows of DESIGN.md section 5.0's table from profiles/<round>_bench_<workload>.json (nothing typed by hand).
    python3 tools/design_table.py r3"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
ROWS = [("whisper80", "`whisper80` (headline, 1024 × 30 s)"),
        ("s2t", "`s2t` (256 ragged 2–10 s clips, float64 mode = default)"),
        ("s2t_f32", "`s2t_f32` (the same, float32 STFT mode)"),
        ("resample_whisper", "`resample_whisper` (512 × CV-length @48 kHz)"),
        ("resample_whisper_fused", "`resample_whisper --fused-normalize`"),
        ("resample_whisper_s16", "`resample_whisper_s16` (int16 PCM in)"),
        ("resample_whisper_trim", "`resample_whisper --trim-silence`"),
        ("whisper128_aug", "`whisper128_aug` (1024 × CV-length, 128 bins, masks)"),
        ("corpus", "`corpus` (30 000 CV-length clips, chunks of 1024)"),
        ("corpus_compact", "`corpus_compact` (chunks in order of length, compact output)")]
def short(k):
    return k.replace("_kernel", "").replace(" (returns at once: the batch is uniform)", " (empty launch)").replace(
        " (returns at once: the batch is not uniform)", " (empty launch)")
for wl, label in ROWS:
    d = json.loads(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{wl}.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    kernels = [f"{short(r['kernel'])} {r['avg_launch_ms']:.3f}"] + [f"{short(k)} {v:.3f}" for k, v in r["other_kernels_avg_ms"].items()]
    traffic = "" if r.get("traffic") is None else f"{r['traffic'] / 1e6:.0f} MB vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB"
    cpu = "" if not d.get("cpu_baseline") else f"{d['cpu_baseline']['value'] / 1e3:.1f} k ({d['cpu_baseline']['cores']} threads)"
    print(f"| {label} | **{d['value'] / 1e6:.2f} M** | {d['ms_per_step']:.3f} ({d['cold_ms_per_step']:.3f}) | {', '.join(kernels)} ms | "
          f"{r['frac']:.3f} / {r['whole_step_frac']:.3f} | {traffic} | {cpu} |")
for wl in ("whisper80", "resample_whisper", "resample_whisper_fused", "s2t", "whisper128_aug"):
    p = os.path.join(ROOT, "profiles", f"{rnd}_bench_streams2_{wl}.json")
    if os.path.exists(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f"streams2 {wl}: serial {d['serial_ms_per_step']:.3f} -> pipelined {d['ms_per_step']:.3f} ms, whole_step_frac {d['roofline']['whole_step_frac']:.3f}")
d = json.loads(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_whisper80.json")).read())
print("e2e headline:", json.dumps(d.get("e2e"))[:600])
d = json.loads(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_resample_whisper.json")).read())
print("e2e config 3:", json.dumps(d.get("e2e"))[:600])
