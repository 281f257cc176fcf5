#!/usr/bin/env python3
"""Generate profiles/summary.md for a round from the committed files alone -- nothing in it is typed by hand.

    python3 tools/make_summary.py r3

Inputs (all under profiles/): <round>_bench_kernel_stats.csv + <round>_bench_under_rocprof.json (the settled
`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline` run), <round>_bench_<w>.json
(bench lines without the profiler), <round>_<w>_kernel_stats.csv and <round>_<w>_pmc.json (tools/profile_round.sh passes).
The formulas are printed with the numbers."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
N_CU, N_SIMD, N_SE = 256, 1024, 32          # MI355X: 256 CUs, 4 SIMDs each, 32 shader engines (SQ_BUSY_CYCLES is summed over them)
# resident waves per SIMD of each dominant kernel (launch bounds / LDS), units one wave of it handles, for the per-unit columns
KERNELS = {
    "whisper_stft_mel_kernel": {"unit": "wave-tile (12 frames)"},
    "whisper_stft_mel12_kernel": {"unit": "workgroup tile (48 frames)"},
    "kaldi_fbank64_kernel": {"unit": "frame pair"},
    "kaldi_fbank_kernel": {"unit": "wave-tile"},
    "decimate_peak_kernel": {"unit": "wave-tile (256 outputs)"},
}
lines = [f"# Generated summary of the {rnd} profiles", "",
         "Written by `python3 tools/make_summary.py " + rnd + "` from the files in this directory; do not edit by hand.", ""]


def load_json(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except json.JSONDecodeError:
        return json.loads(txt.splitlines()[-1])


def kernel_rows(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("sttfe::", "")
        if not name or name.startswith("at::") or name.startswith("__amd") or "elementwise" in name or "distribution" in name:
            continue
        rows.append((name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    return rows


# ---- 1. the settled bench trace against the bench line of the same run
stats = os.path.join(P, f"{rnd}_bench_kernel_stats.csv")
line = load_json(f"{rnd}_bench_under_rocprof.json")
if os.path.exists(stats) and line:
    rf = line["roofline"]
    lines += ["## Headline run under the profiler", "",
              "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline` "
              f"(`{rnd}_bench_kernel_stats.csv`; the JSON line of the same run: `{rnd}_bench_under_rocprof.json`).", "",
              "| kernel | launches | avg us | min us | max us |", "|---|---|---|---|---|"]
    avg_of = {}
    for name, calls, avg, mn, mx in kernel_rows(stats):
        lines.append(f"| `{name}` | {calls} | {avg:.1f} | {mn:.1f} | {mx:.1f} |")
        avg_of[name] = avg
    k = rf["kernel"]
    trace_avg = next((v for n, v in avg_of.items() if k in n), None)
    if trace_avg:
        frac_trace = rf["algorithmic_bytes_per_launch"] / (trace_avg * 1e-6) / 1e9 / rf["peak"]
        lines += ["", f"`{k}`: trace average {trace_avg:.1f} us over every launch of the run (warm-up and settle phase included); "
                  f"bench.py's HIP events on the timed launches of the same run {rf['avg_launch_ms'] * 1e3:.1f} us.  "
                  f"`roofline.frac` = algorithmic bytes per launch ({rf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB) / duration / {rf['peak']:.0f} GB/s: "
                  f"**{frac_trace:.3f} from the trace average, {rf['frac']:.3f} printed by bench.py** "
                  f"({100 * (frac_trace / rf['frac'] - 1):+.1f} %).", ""]

# ---- 2. bench lines without the profiler
rows = []
for path in sorted(glob.glob(os.path.join(P, f"{rnd}_bench_*.json"))):
    w = os.path.basename(path)[len(rnd) + 7:-5]
    if w == "under_rocprof":
        continue
    r = load_json(os.path.basename(path))
    if not r or "roofline" not in r:
        continue
    rf = r["roofline"]
    others = ", ".join(f"{k} {v * 1e3:.0f}" for k, v in rf.get("other_kernels_avg_ms", {}).items())
    e2e = r.get("e2e")
    e2e_txt = "" if not e2e else "; ".join(f"{k}: {v['value'] / 1e6:.2f} M audio-s/s, {v.get('pcie_gbs', 0):.1f} GB/s" for k, v in e2e.items() if isinstance(v, dict) and "value" in v)
    rows.append(f"| `{w}` | {r['value'] / 1e6:.2f} M | {r['ms_per_step']:.3f} | `{rf['kernel']}` {rf['avg_launch_ms'] * 1e3:.0f} | {others} | "
                f"{rf['frac']:.3f} | {rf.get('whole_step_frac', 0):.3f} | {'' if rf.get('traffic') is None else '%.0f' % (rf['traffic'] / 1e6)} / {rf['algorithmic_bytes_per_launch'] / 1e6:.0f} | {e2e_txt} |")
if rows:
    lines += ["## Bench lines (no profiler, one gpurun call, `tools/bench_all.sh`)", "",
              "| workload | audio-s/s | ms / step | dominant kernel, us | other kernels, us | roofline.frac | whole_step_frac | PMC traffic / algorithmic MB | e2e |",
              "|---|---|---|---|---|---|---|---|---|"] + rows + [""]

# ---- 3. per-workload PMC digests
for path in sorted(glob.glob(os.path.join(P, f"{rnd}_*_pmc.json"))):
    w = os.path.basename(path)[len(rnd) + 1:-9]
    d = load_json(os.path.basename(path))
    if not d or w.endswith("mem"):
        continue
    lines += [f"## PMC digest: `{w}`", "", f"`{d.get('command', '')}`", ""]
    ks = os.path.join(P, f"{rnd}_{w}_kernel_stats.csv")
    durs = {n: a for n, _, a, _, _ in kernel_rows(ks)} if os.path.exists(ks) else {}
    for kname, cs in d["counters"].items():
        g = lambda c: cs[c]["mean_per_dispatch"] if c in cs else None
        if g("SQ_BUSY_CYCLES") is None or g("SQ_INSTS_VALU") is None:
            continue
        cyc = g("SQ_BUSY_CYCLES") / N_SE                       # kernel duration in shader cycles
        dur = next((v for n, v in durs.items() if n.split("<")[0] in kname or kname.split("<")[0] in n), None)
        out = [f"**`{kname}`** -- {'' if dur is None else '%.1f us average in the trace pass; ' % dur}{cyc:,.0f} shader cycles per launch (`SQ_BUSY_CYCLES / 32` shader engines), {g('SQ_WAVES'):,.0f} waves."]
        out.append(f"- instructions per launch: VALU {g('SQ_INSTS_VALU'):.3g}, SALU {g('SQ_INSTS_SALU'):.3g}, LDS {g('SQ_INSTS_LDS'):.3g}, "
                   f"VMEM read {g('SQ_INSTS_VMEM_RD'):.3g} / write {g('SQ_INSTS_VMEM_WR'):.3g}"
                   + (f", MFMA {g('SQ_INSTS_VALU_MFMA_MOPS_F32') / 4:.3g} (`SQ_INSTS_VALU_MFMA_MOPS_F32 / 4`: a 16x16x4 f32 MFMA counts 4)" if g("SQ_INSTS_VALU_MFMA_MOPS_F32") else ""))
        if w == "whisper80" and "whisper_stft_mel12" in kname:
            wt = 1024 * 63                                       # clips x 48-frame tiles: one tile of the 12-wave workgroup
            out.append(f"- per 48-frame tile of a workgroup (`/ (1024 clips x 63 tiles)` = {wt:,}; 8 transform + 4 mel waves): VALU {g('SQ_INSTS_VALU') / wt:.0f}, "
                       f"SALU {g('SQ_INSTS_SALU') / wt:.0f}, LDS {g('SQ_INSTS_LDS') / wt:.0f}, VMEM {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / wt:.0f}, "
                       f"MFMA {g('SQ_INSTS_VALU_MFMA_MOPS_F32') / 4 / wt:.0f}; {cyc / 252:,.0f} cycles per tile of a workgroup "
                       f"(`cycles / 252 tiles per workgroup`, 256 workgroups)")
        elif w == "whisper80" and "whisper_stft_mel" in kname:
            out.append("- (the batch is uniform: this kernel reads the lengths and returns, `whisper_stft_mel12_kernel` computes the call)")
        if g("SQ_ACTIVE_INST_VALU") is not None:
            valu = 4 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * cyc)
            out.append(f"- VALU pipe busy **{100 * valu:.0f} %** = `4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)`; "
                       f"cycles per VALU instruction {4 * g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU'):.2f}")
        if g("SQ_LDS_IDX_ACTIVE") is not None:
            out.append(f"- LDS busy **{100 * g('SQ_LDS_IDX_ACTIVE') / (N_CU * cyc):.0f} %** = `SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)`, "
                       f"bank conflicts {100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.0f} % of the active cycles (`SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE`)")
        if g("SQ_VALU_MFMA_BUSY_CYCLES"):
            out.append(f"- MFMA pipe busy **{100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (N_SIMD * cyc):.0f} %** = `SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)`")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
            wc = g("SQ_WAVE_CYCLES")
            out.append(f"- a wave's time: issuing {100 * g('SQ_ACTIVE_INST_ANY') / wc:.0f} %, waiting to issue {100 * g('SQ_WAIT_INST_ANY') / wc:.0f} %, "
                       f"in `s_waitcnt` {100 * g('SQ_WAIT_ANY') / wc:.0f} % (`SQ_ACTIVE_INST_ANY`, `SQ_WAIT_INST_ANY`, `SQ_WAIT_ANY` over `SQ_WAVE_CYCLES`)")
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            out.append(f"- HBM side: read {g('FETCH_SIZE') * 2048 / 1e6:.1f} MB (`FETCH_SIZE x 1024 x 2`, the gfx950 correction), "
                       f"written {g('WRITE_SIZE') * 1024 / 1e6:.1f} MB (`WRITE_SIZE x 1024`) per launch")
        lines += out + [""]

open(os.path.join(P, "summary.md"), "w").write("\n".join(lines) + "\n")
print("wrote profiles/summary.md,", len(lines), "lines")
