#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/profile_round.sh <tag> <bench.py workload> [more bench.py flags]
# One rocprofv3 run per pass over the SAME short bench.py command: the kernel trace, then each PMC group on its own
# (never combined with tracing), into gpurun_out/<tag>/<pass>/.  Stops at the first pass that fails.
tag=$1; wl=$2; shift 2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
CMD="bench.py --workload $wl --steps 4 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-e2e $*"
run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 $CMD > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }
    echo "pass $name ok"
}
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_LT_64 SQ_THREAD_CYCLES_VALU
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
echo "passes done: $CMD"
