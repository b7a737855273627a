"""bench.py's launcher: `--gpus N` without WORLD_SIZE must start N ranks itself (the parent never touches the GPU), and a
line is printed only when the number of ranks that ran equals --gpus.  --dry-launch makes the ranks report their environment
and exit, so this runs on a CPU-only machine (gloo rendezvous on 127.0.0.1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_starts_two_ranks_with_distinct_local_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=clean_env(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert sorted(x["local_rank"] for x in out["ranks"]) == [0, 1]
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1]


def test_three_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-launch"], env=clean_env(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["n_gpus"] == 3 and sorted(x["local_rank"] for x in out["ranks"]) == [0, 1, 2]


def test_rank_count_must_equal_gpus():
    """One rank running with --gpus 2 (what a plain `python bench.py --gpus 2` used to do silently) is refused."""
    env = clean_env()
    env["WORLD_SIZE"] = "1"
    env["RANK"] = "0"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""


def test_a_failing_rank_fails_the_launcher_and_prints_nothing():
    # an argument the ranks reject: every rank exits non-zero, the parent must too
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "nope"], env=clean_env(), capture_output=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""
