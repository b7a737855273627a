import os
import subprocess
import sys

import pytest

# torch first: its wheel carries its own HIP runtime under the same soname as /opt/rocm's (libamdhip64.so.7), and the process
# uses whichever is loaded first.  With libparamugsy_amd.so loaded first, torch's kernels meet a runtime they were not built
# for ("no ROCm-capable device"); the other way round both work -- the order bench.py has.  (The executables under bin/ run on
# /opt/rocm's runtime, so the suite exercises both.)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover -- the CPU-side tests do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_BUILD = os.path.join(ROOT, "oracle", "_build")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_build():
    """Build the CPU oracle (test infrastructure) if its binaries are not there yet."""
    need = ["libpm_oracle.so", "oracle_m_translate", "oracle_m_sort_delta", "oracle_maf_analyzer", "oracle_units", "libdp_oracle.so"]
    if not all(os.path.exists(os.path.join(ORACLE_BUILD, n)) for n in need):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True)
    return ORACLE_BUILD


@pytest.fixture(scope="session")
def ref_dir():
    """oracle/_ref: the upstream reference's own binaries.  Present in the dev container (and shipped to the GPU
    box as prebuilt files); tests that diff against it live are skipped where it is absent."""
    if not os.path.exists(os.path.join(REF_DIR, "m_translate")):
        pytest.skip("oracle/_ref not built (no /root/reference here); golden fixtures cover this")
    return REF_DIR


@pytest.fixture(scope="session")
def hip_lib():
    from paramugsy_amd import capi
    return capi.lib()
