#!/bin/bash
# A/B of the float64 CMVN kernels on config 2 (one gpurun call): ms per step and per kernel
#   base = a library built from the previous commit (speech_to_text_amd/libsttfe_base.so), reg = columns in registers
#   (default), lds = the LDS-panel form (STTFE_KALDI_LDS_CMVN=1), split = statistics + normalisation kernels
run() {
  echo "== $1"
  env $2 timeout -k 10 120 python bench.py --workload s2t --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_ms'],4), {k:round(v,4) for k,v in r['other_kernels_avg_ms'].items()})"
}
[ -f speech_to_text_amd/libsttfe_base.so ] && { run base STTFE_LIB_PATH=$PWD/speech_to_text_amd/libsttfe_base.so || exit 1; }
run reg A=1 || exit 1
run lds STTFE_KALDI_LDS_CMVN=1 || exit 1
run reg A=1 || exit 1
run lds STTFE_KALDI_LDS_CMVN=1 || exit 1
run split STTFE_KALDI_SPLIT_CMVN=1 || exit 1
