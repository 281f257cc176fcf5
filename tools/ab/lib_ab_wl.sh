#!/bin/bash
# A/B of another build of the library against the shipped one on several workloads: tools/ab/lib_ab_wl.sh <other.so> <workload> ...
so=$1; shift
run() {
  env $2 timeout -k 10 200 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 $3', round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), round(r['frac'],4), {k[:24]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
for wl in "$@"; do
  run $wl A=1 shipped || exit 1
  run $wl STTFE_LIB_PATH=$PWD/$so other || exit 1
  run $wl A=1 shipped || exit 1
  run $wl STTFE_LIB_PATH=$PWD/$so other || exit 1
done
