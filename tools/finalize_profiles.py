#!/usr/bin/env python3
"""Turn the pass directories of tools/profile_round.sh into the committed summaries.

    python3 tools/finalize_profiles.py <round tag, e.g. r2> <gpurun_out prefix, e.g. gpurun_out/r2c_>

For every workload directory <prefix><workload>/ : profiles/<round>_<workload>_kernel_stats.csv (the rocprofv3
--kernel-trace --stats table), profiles/<round>_<workload>_pmc.json (per-dispatch means of every counter, per kernel)
and one entry of profiles/traffic.json: HBM bytes per launch of the workload's dominant kernel,
FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024 (the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports
half of a streaming read).  bench.py reports that figure as `roofline.traffic` when the launch shape matches.
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, prefix = sys.argv[1], sys.argv[2]
DOMINANT = {"whisper80": "whisper_stft_mel12_kernel", "whisper128_aug": "whisper_stft_mel12_kernel",
            "resample_whisper": "whisper_stft_mel12_kernel", "resample_whisper_s16": "whisper_stft_mel12_kernel",
            "resample_whisper_fused": "whisper_stft_mel12_kernel",
            "s2t": "kaldi_fbank64_kernel", "s2t_f32": "kaldi_fbank_kernel"}
CLIPS = {"whisper80": 1024, "whisper128_aug": 1024, "resample_whisper": 512, "resample_whisper_s16": 512,
         "resample_whisper_fused": 512, "s2t": 256, "s2t_f32": 256}
traffic_path = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for d in sorted(glob.glob(prefix + "*")):
    wl = os.path.basename(d)[len(os.path.basename(prefix)):]
    if wl not in DOMINANT or not os.path.isdir(d):
        continue
    cmd = open(os.path.join(d, "trace.log")).read()
    stats = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(ROOT, "profiles", f"{rnd}_{wl}_kernel_stats.csv"))
    pmc = os.path.join(ROOT, "profiles", f"{rnd}_{wl}_pmc.json")
    flags = " --fused-normalize" if wl.endswith("_fused") else ""
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), d, pmc,
                    f"bash tools/profile_round.sh <tag> {wl.replace('_fused', '')}{flags}  (one rocprofv3 run per pass over `python3 "
                    f"bench.py --workload {wl.replace('_fused', '')} --steps 4 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-e2e{flags}`)"],
                   check=True, stdout=subprocess.DEVNULL)
    counters = json.load(open(pmc))["counters"]
    per_kernel = {}
    for k, cs in counters.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            per_kernel[k.replace("void ", "")] = cs["FETCH_SIZE"]["mean_per_dispatch"] * 2048.0 + cs["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
    dom = DOMINANT[wl]
    rec = counters[[k for k in counters if dom in k][0]]
    traffic[wl] = {
        "clips_per_launch": CLIPS[wl],
        "hbm_bytes_per_launch": rec["FETCH_SIZE"]["mean_per_dispatch"] * 2048.0 + rec["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0,
        "fetch_size_kib": rec["FETCH_SIZE"]["mean_per_dispatch"],
        "write_size_kib": rec["WRITE_SIZE"]["mean_per_dispatch"],
        "kernel": dom,
        "all_kernels_hbm_bytes_per_launch": per_kernel,
        "source": f"profiles/{rnd}_{wl}_pmc.json",
        "formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 correction)",
    }
    print(wl, "%.1f MB per launch of %s" % (traffic[wl]["hbm_bytes_per_launch"] / 1e6, dom))
json.dump(traffic, open(traffic_path, "w"), indent=1)
