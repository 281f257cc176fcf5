set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/bench_all.sh r3b | tail -10
for w in whisper80 resample_whisper whisper128_aug s2t; do timeout -k 10 240 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --streams 2 > gpurun_out/r3b/streams2_$w.json 2>/dev/null; done
timeout -k 10 240 python bench.py --workload resample_whisper --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --streams 2 --fused-normalize > gpurun_out/r3b/streams2_resample_whisper_fused.json 2>/dev/null
timeout -k 10 240 python bench.py --workload resample_whisper --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --trim-silence > gpurun_out/r3b/resample_whisper_trim.json 2>/dev/null
echo bench lines done
mkdir -p gpurun_out/r3bench
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3bench/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r3bench/bench.json 2> gpurun_out/r3bench/bench.err
find gpurun_out/r3bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3bench/kernel_stats.csv
echo settled trace done
