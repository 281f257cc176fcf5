#!/usr/bin/env python3
"""Summarise the per-pass rocprofv3 outputs of tools/pmc_passes.sh into one JSON (per-dispatch means per kernel).

    python3 tools/summarize_pmc.py gpurun_out/<tag> profiles/<name>.json "<command that produced the passes>" [extra kernel name parts ...]
"""
import collections
import csv
import glob
import json
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
keep = ("whisper", "kaldi", "resample", "decimate", "copy_peak", "normalize", "trim", "waveform", "pv_") + tuple(sys.argv[4:])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{src}/*/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])          # one row per XCD / dimension: sum them
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("sttfe::", "")
    for (disp, counter), v in per_dispatch.items():
        acc[names[disp]][counter].append(v)
out = {
    "command": cmd,
    "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a streaming read "
            "(MI355X_MICROARCH.md, HBM) -> read bytes = FETCH_SIZE * 1024 * 2",
    "counters": {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(cs.items())}
                 for k, cs in sorted(acc.items()) if any(t in k for t in keep)},
}
json.dump(out, open(dst, "w"), indent=1)
for k, cs in out["counters"].items():
    print(k)
    for c, v in cs.items():
        print(f"   {c:30s} {v['mean_per_dispatch']:.4g}  ({v['dispatches']} dispatches)")
