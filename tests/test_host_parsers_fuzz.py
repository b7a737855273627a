"""Malformed input must end in an error code, never in a crash: the host parsers of the product are fed mutated
files (truncations, byte flips, token deletions).  Runs the loader in a child process so a crash is caught; CPU only
(pm_workload_load needs no device).  When a mutated job still parses, the unit list must equal the oracle's or the
oracle must reject it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
from paramugsy_amd import capi
from paramugsy_amd.translate import Workload
try:
    w = Workload.load(sys.argv[1], sys.argv[2], sys.argv[3:], allow_parse_error=True)
    t = w.tables()
    print("OK", t.n_units, w.parse_error is not None)
except capi.PmError as e:
    print("ERR", e.code)
"""


def mutate(rng, data: bytes) -> bytes:
    b = bytearray(data)
    kind = int(rng.integers(0, 5))
    if kind == 0 and len(b) > 10:  # truncate
        return bytes(b[:int(rng.integers(1, len(b)))])
    if kind == 1:  # flip bytes
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(32, 127))
        return bytes(b)
    if kind == 2:  # delete a line
        lines = data.split(b"\n")
        del lines[int(rng.integers(0, len(lines)))]
        return b"\n".join(lines)
    if kind == 3:  # duplicate a line
        lines = data.split(b"\n")
        k = int(rng.integers(0, len(lines)))
        lines.insert(k, lines[k])
        return b"\n".join(lines)
    # huge / negative numbers
    return data.replace(b" 1", b" -99999999999999999999", 1) if rng.random() < 0.5 else data.replace(b"\n1", b"\n18446744073709551616", 1)


@pytest.mark.parametrize("seed", range(12))
def test_mutated_jobs_never_crash_the_loader(seed, tmp_path):
    rng = np.random.default_rng(seed)
    case = os.path.join(GOLDEN, "translate_reverse")
    files = {"profiles-l/profiles": None, "profiles-r/profiles": None, "nucmer_0.delta": None}
    for rel in files:
        files[rel] = open(os.path.join(case, rel), "rb").read()
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT))
    for trial in range(6):
        d = tmp_path / ("t%d" % trial)
        (d / "profiles-l").mkdir(parents=True)
        (d / "profiles-r").mkdir()
        victim = list(files)[int(rng.integers(0, 3))]
        for rel, data in files.items():
            (d / rel).write_bytes(mutate(rng, data) if rel == victim else data)
        r = subprocess.run([sys.executable, str(script), str(d / "profiles-l"), str(d / "profiles-r"), str(d / "nucmer_0.delta")],
                           capture_output=True, timeout=60)
        assert r.returncode == 0, (victim, r.stderr[-300:])
        assert r.stdout.startswith(b"OK") or r.stdout.startswith(b"ERR"), r.stdout
