#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/bench_all.sh <tag>  -- every bench.py workload line into gpurun_out/<tag>/
tag=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
out=gpurun_out/$tag
mkdir -p $out
run() {
    name=$1; shift
    timeout -k 10 240 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("%-28s %.3f ms/step (cold %.3f)  %.2f M audio-s/s  %s %.4f ms  others %s  frac %.3f whole %.3f traffic %s cpu %s" % (
    sys.argv[2], r["ms_per_step"], r.get("cold_ms_per_step", 0), r["value"] / 1e6, rf["kernel"], rf["avg_launch_ms"],
    {k: round(v, 4) for k, v in rf["other_kernels_avg_ms"].items()}, rf["frac"], rf["whole_step_frac"],
    None if rf["traffic"] is None else round(rf["traffic"] / 1e6, 1),
    None if not r.get("cpu_baseline") else round(r["cpu_baseline"]["value"])))
PY
}
run whisper80 --steps 20 --warmup 5
run s2t --workload s2t --steps 20 --warmup 5
run s2t_f32 --workload s2t_f32 --steps 20 --warmup 5
run resample_whisper --workload resample_whisper --steps 20 --warmup 5 --no-cpu-baseline
run resample_whisper_fused --workload resample_whisper --steps 20 --warmup 5 --no-cpu-baseline --fused-normalize
run resample_whisper_s16 --workload resample_whisper_s16 --steps 20 --warmup 5 --no-cpu-baseline
run whisper128_aug --workload whisper128_aug --steps 20 --warmup 5 --no-cpu-baseline
run corpus --workload corpus --steps 5 --warmup 2 --no-cpu-baseline
run corpus_compact --workload corpus_compact --steps 5 --warmup 2 --no-cpu-baseline
