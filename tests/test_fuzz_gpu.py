"""A few seconds of each differential fuzzer of tools/ (fixed first seeds, so the cases are the same every time): the library against
the upstream binaries byte for byte, the translate job against the oracle unit by unit, make / untranslate against their
transcriptions, the DP under random kernel choices against its oracle.  The long runs are in profiles/r02_*_fuzz.txt and profiles/r03_*_fuzz.txt."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_tool(name, seconds, seed, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name), str(seconds), str(seed)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def test_library_against_the_upstream_binaries(oracle_build):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "m_translate")):
        pytest.skip("oracle/_ref was not built where /root/reference exists")
    out = run_tool("ref_fuzz.py", 8, 900001)
    assert "all equal" in out and "DIFFERENT" not in out and "DISAGREE" not in out


def test_translate_job_against_the_oracle(oracle_build):
    out = run_tool("translate_fuzz.py", 8, 900001)
    assert "every unit equals the oracle" in out and "MISMATCH" not in out


def test_make_and_untranslate_against_their_transcriptions(oracle_build):
    out = run_tool("stage_fuzz.py", 6, 900001)
    assert "all equal the transcriptions" in out and "DIFFERENT" not in out


def test_dp_under_random_kernel_choices_against_its_oracle(oracle_build):
    out = run_tool("dp_fuzz.py", 8, 900001)
    assert "all equal the oracle" in out and "MISMATCH" not in out


def test_dp_device_list_engine_with_workers_sharing_the_gpu(oracle_build):
    """Only the device-list entry (two to four host threads, each with a host-fed engine on non-blocking streams, on the one GPU): the
    arrangement in which round 3's fuzzing found the fill kernel's error word being cleared too late (profiles/r03_dp_fuzz.txt)."""
    out = run_tool("dp_fuzz.py", 10, 204600, PM_FUZZ_ENGINE="multi")
    assert "all equal the oracle" in out and "MISMATCH" not in out
