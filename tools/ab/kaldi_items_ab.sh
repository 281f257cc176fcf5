#!/bin/bash
# float64 Kaldi kernel: frame pairs per (wave-owned) item, STTFE_KALDI_ITEM_PAIRS = 1 / 2 / 4 / 8 and automatic, by batch size
# on config 2's clip lengths; base = the library of the previous commit (workgroup-owned 32-frame items)
run() {
  env $3 timeout -k 10 120 python bench.py --workload s2t --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'batch', $2, 'step', round(d['ms_per_step'],4), 'fbank64', round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['other_kernels_avg_ms'].items()}, round(d['value']/1e6,2), 'M audio-s/s')"
}
for b in ${BATCHES:-16 64 128 256 512 1024}; do
  [ -f speech_to_text_amd/libsttfe_base.so ] && { run base $b STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1; }
  for ip in 1 2 4 8; do run "pairs $ip" $b STTFE_KALDI_ITEM_PAIRS=$ip || exit 1; done
  run "auto" $b A=1 || exit 1
done
