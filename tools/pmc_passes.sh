#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/pmc_passes.sh <tag> <prof_step args...>
# Runs the kernel-trace pass and the separate PMC passes, one rocprofv3 run each, into gpurun_out/<tag>/.
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 tools/prof_step.py $ARGS > $out/$name.log 2>&1; }
ARGS="$*"
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_LT_64 SQ_THREAD_CYCLES_VALU
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run grbm --pmc GRBM_GUI_ACTIVE
echo passes done
