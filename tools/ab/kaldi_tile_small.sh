#!/bin/bash
run() {
  env $3 timeout -k 10 120 python bench.py --workload s2t --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'batch', $2, 'step', round(d['ms_per_step'],4), 'fbank64', round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
for b in 16 64 128; do
  run base $b STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1
  for tf in 32 16 8; do run "tile $tf" $b STTFE_KALDI_TILE_FRAMES=$tf || exit 1; done
done
