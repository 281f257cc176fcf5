#!/bin/bash
# A/B of several builds of the library on one bench workload: tools/ab/lib_ab_many.sh <workload> <a.so> <b.so> ... (shipped first and last)
wl=$1; shift
run() {
  echo "== $1"
  env $2 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), round(r['frac'],4), {k[:24]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
run shipped A=1 || exit 1
for so in "$@"; do run "$so" STTFE_LIB_PATH=$PWD/$so || exit 1; done
run shipped A=1 || exit 1
for so in "$@"; do run "$so" STTFE_LIB_PATH=$PWD/$so || exit 1; done
