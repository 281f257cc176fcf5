"""bench.py --gpus N starts its own ranks: the N > 1 control flow (rendezvous, barriers, lock-step settle phase,
MAX-over-ranks time, SUM-over-ranks audio, ONE JSON line from rank 0, worst exit code) on CPU with gloo.
STTFE_BENCH_STUB=1 replaces the GPU step with a host sleep; everything else is the code the GPU run uses."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run(args, extra_env=None, timeout=120):
    env = dict(os.environ, STTFE_BENCH_STUB="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_without_a_launcher():
    r = run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--settle-seconds", "0.05"])
    assert r.returncode == 0, r.stderr[-2000:]
    # (gloo itself prints a "[Gloo] Rank 0 is connected ..." notice on stdout; RCCL does not)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout                      # rank 0 only, one line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["audio_total"] == 300.0                      # 100 (rank 0) + 200 (rank 1): SUM over ranks
    # MAX over ranks: rank 1's steps sleep 4 ms each, rank 0's 2 ms
    assert j["ms_per_step"] >= 3.9
    assert abs(j["value"] - 300.0 * 5 / (j["ms_per_step"] * 5e-3)) <= 1e-6 * j["value"]
    assert j["settle_rounds"][0] == j["settle_rounds"][1]  # the ranks left the settle phase in the same iteration


def test_gpus_1_runs_in_process():
    r = run(["--gpus", "1", "--steps", "3", "--warmup", "0", "--settle-seconds", "0.01"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip())
    assert j["n_gpus"] == 1 and j["audio_total"] == 100.0


def test_a_failing_rank_fails_the_run():
    # WORLD_SIZE that contradicts --gpus inside the children: every rank exits non-zero, the parent reports it
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "nonsense"])
    assert r.returncode != 0


def test_a_rank_above_zero_that_dies_leaves_its_diagnostics():
    """rank 1 exits with code 7 right after the first barrier (rank 0 is then stuck in a collective
    and has to be stopped): the parent returns 7, prints the tail of rank 1's stderr and says which ranks it stopped."""
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--settle-seconds", "0.05"],
            extra_env={"STTFE_BENCH_STUB_FAIL_RANK": "1"}, timeout=180)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 of 2 exited with code 7" in r.stderr
    assert "fails on purpose" in r.stderr                              # the failing rank's own message came through
    assert "rank 0 of 2 was stopped" in r.stderr
    assert "per-rank logs kept in" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no result line from a failed run
