#!/bin/bash
# Usage (GPU box): tools/ab_workloads.sh "<workloads>" libA.so libB.so ...  -- A B A B order, dominant-kernel ms per launch
wls=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for rep in 1 2; do
  for lib in "$@"; do
    for w in $wls; do
      STTFE_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib $w failed"; tail -3 gpurun_out/ab.err; exit 1; }
      python - "$lib" "$w" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); rf = r["roofline"]
print("%-40s %-18s step %.4f ms  kernel %.4f ms  others %s" % (sys.argv[1], sys.argv[2], r["ms_per_step"], rf["avg_launch_ms"], {k: round(v, 4) for k, v in rf["other_kernels_avg_ms"].items()}), flush=True)
PY
    done
  done
done
