set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r1e_ifetch; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/a -- python3 tools/prof_step.py whisper 1024 2 > $out/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES --output-format csv -d $out/b -- python3 tools/prof_step.py whisper 1024 2 > $out/b.log 2>&1
echo ok
