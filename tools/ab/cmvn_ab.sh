#!/bin/bash
# A/B of kaldi_cmvn_seq_kernel builds / timing-only variants on config 2 (one gpurun call): prints ms of the two kernels
run() {
  echo "== $1"
  env $2 timeout -k 10 120 python bench.py --workload s2t --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
run base STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1
run new STTFE_CMVN_VARIANT=0 || exit 1
run base STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1
run new STTFE_CMVN_VARIANT=0 || exit 1
for v in 1 2 3 4 5 6 7; do run "variant $v" STTFE_CMVN_VARIANT=$v || exit 1; done
