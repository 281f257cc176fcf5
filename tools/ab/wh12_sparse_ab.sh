#!/bin/bash
# A/B of the 12-wave kernel's mel projection on one box: band-sparse packed VALU (default) against MFMAs (STTFE_WH12_MFMA=1)
# tools/ab/wh12_sparse_ab.sh <workload> ...
run() {
  echo "== $1 $2"
  env $2 timeout -k 10 200 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), round(r['frac'],4), {k[:24]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
for wl in "$@"; do
  run $wl STTFE_WH12_MFMA=1 || exit 1
  run $wl STTFE_WH12_MFMA=0 || exit 1
  run $wl STTFE_WH12_MFMA=1 || exit 1
  run $wl STTFE_WH12_MFMA=0 || exit 1
done
