"""`python bench.py --gpus N` must start N ranks itself: the parent spawns one child
per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and fails unless all N ranks
finished.  Rehearsed here on CPU with the stub worker (gloo, no GPU, no library): the distributed plumbing — barrier,
max-over-ranks of the wall time, per-rank gather — is the real code path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1", *extra],
                          capture_output=True, text=True, timeout=300, env=e)


def test_launcher_runs_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 128 and j["scaling"] == "weak"
    assert len(j["per_rank_ms"]) == 2 and j["steps"] == 3 and j["warmup"] == 1
    # whole-job value = units of ALL ranks over the max-over-ranks time (rank 1's stub step is the slower one)
    assert j["per_rank_ms"][1] > j["per_rank_ms"][0]
    assert j["ms_per_step"] >= max(j["per_rank_ms"]) * 0.99
    assert abs(j["value"] - 2 * 64 * 488 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6


def test_launcher_fails_when_a_rank_dies():
    r = _run("--gpus", "2", "--stub-fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_worker_refuses_a_world_size_mismatch():
    r = _run("--gpus", "4", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_single_rank_needs_no_launcher():
    r = _run("--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and len(j["per_rank_ms"]) == 1
