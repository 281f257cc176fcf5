#!/bin/bash
# Run length of the 12-wave kernel's items (STTFE_WH_TILES_PER_BLOCK; 0 = the kernel's own choice) on three workloads
run() {
  env STTFE_WH_TILES_PER_BLOCK=$2 timeout -k 10 200 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 tiles_per_block $2:', round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), round(r['frac'],4))"
}
for t in 0 4 8 11 16 21 32 63 0; do run whisper80 $t || exit 1; done
for w in resample_whisper whisper128_aug; do for t in 0 1 2 3 4 6 0; do run $w $t || exit 1; done; done
