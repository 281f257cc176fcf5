#!/bin/bash
# float64 Kaldi kernel: tile size (STTFE_KALDI_TILE_FRAMES = 32 / 16 / 8, unset = automatic) by batch size on config 2's clip lengths
run() {
  env $3 timeout -k 10 120 python bench.py --workload s2t --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'batch', $2, 'step', round(d['ms_per_step'],4), 'fbank64', round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['other_kernels_avg_ms'].items()}, round(d['value']/1e6,2), 'M audio-s/s')"
}
[ -f speech_to_text_amd/libsttfe_base.so ] && { run base 256 STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1; }
for b in 256 512 1024; do
  for tf in 32 16 8; do run "tile $tf" $b STTFE_KALDI_TILE_FRAMES=$tf || exit 1; done
  run "auto" $b A=1 || exit 1
done
