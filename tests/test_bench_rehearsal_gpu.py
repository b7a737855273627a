"""bench.py's N-rank flow on the one-GPU box: PM_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo
instead of RCCL.  SIX ranks -- the box allows six processes on its card (the eight-wide paths are covered by the eight-entry device
list in one process, tests/test_multi_device.py, and by eight gloo ranks on the CPU, tests/test_shard_gloo.py).  The numbers of such a
run mean nothing and the line says so; what is checked is the flow: the census, the strong split by cells, one compact line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_six_rank_rehearsal_of_the_strong_split():
    env = dict(os.environ, PM_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "6", "--config", "c2", "--scaling", "strong", "--dp-pairs", "6000",
                        "--path", "dp", "--no-ride-alongs", "--steps", "2", "--warmup", "1", "--long-form", "none"],
                       env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 6 and out["scaling"] == "strong" and out["rehearsal"] is True
    assert out["config"]["pairs_in_job"] == 6000 and 0 < out["config"]["pairs_per_rank"] < 6000
    assert out["value"] > 0 and out["oracle_check"] is True
