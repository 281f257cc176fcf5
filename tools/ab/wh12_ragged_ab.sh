#!/bin/bash
# ragged batches on the 12-wave Whisper kernel (default) against the general kernel (STTFE_WH12_RAGGED=0), bench workloads;
# TPB="1 2 3": also with fixed tiles per item (STTFE_WH_TILES_PER_BLOCK); HEADLINE=0 leaves the headline out
run() {
  echo "== $1: $2 $3"
  env $4 timeout -k 10 240 python bench.py --workload $2 $3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), round(r['frac'],3), round(r['whole_step_frac'],3), {k[:22]:round(v,4) for k,v in r['other_kernels_avg_ms'].items()}, round(d['value']/1e6,2), 'M')"
}
one() {
  run general $1 "$2" STTFE_WH12_RAGGED=0 || exit 1
  run wh12 $1 "$2" A=1 || exit 1
  for t in ${TPB:-}; do run "wh12 tiles/item $t" $1 "$2" STTFE_WH_TILES_PER_BLOCK=$t || exit 1; done
}
one resample_whisper "--steps 20 --warmup 5"
one corpus "--steps 5 --warmup 2"
one corpus_compact "--steps 5 --warmup 2"
[ "${HEADLINE:-1}" = 1 ] && one whisper80 "--steps 20 --warmup 5"
