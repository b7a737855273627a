"""The callers either side of the translate path, chained on the GPU as the task script chains them
(lib/base/mugsy_profiles_task.ml:40-58): `mugsy_profiles make` on both MAFs, then `m_translate` on the result.
Checked against the CPU chain: make oracle (Python transcription of the OCaml source) -> translate oracle / upstream binary."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from paramugsy_amd import synth

pytestmark = pytest.mark.gpu


def test_make_then_translate_equals_cpu_chain(oracle_build, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_oracle
    rng = np.random.default_rng(2026)
    lg, rg = ["L0.chr", "L1.chr"], ["R0.chr", "R1.chr", "R2.chr"]
    mafs = {"l": synth.side_to_maf_text(synth.gen_side(rng, lg, 30000, 40, mean_cols=300, gap_rate=0.03)),
            "r": synth.side_to_maf_text(synth.gen_side(rng, rg, 30000, 40, mean_cols=300, gap_rate=0.03))}
    delta = tmp_path / "n.delta"
    delta.write_text(synth.gen_delta_text(rng, lg, rg, 30000, 30000, 120, mean_len=900))
    (tmp_path / "nucmer.list").write_text(str(delta) + "\n")
    exe = os.path.join(ROOT, "bin", "mugsy_profiles")
    for side in ("l", "r"):
        (tmp_path / ("%s.maf" % side)).write_text(mafs[side])
        r = subprocess.run([exe, "make", "-in_maf", str(tmp_path / ("%s.maf" % side)), "-out_dir", str(tmp_path / ("gpu-" + side)),
                            "-basename", side], capture_output=True)
        assert r.returncode == 0, r.stderr
        os.makedirs(tmp_path / ("cpu-" + side))
        prof, fasta = make_oracle.make(mafs[side], side)
        (tmp_path / ("cpu-" + side) / "profiles").write_text(prof)
        (tmp_path / ("cpu-" + side) / "sequences.fasta").write_text(fasta)
        assert filecmp.cmp(tmp_path / ("gpu-" + side) / "profiles", tmp_path / ("cpu-" + side) / "profiles", shallow=False)
        assert filecmp.cmp(tmp_path / ("gpu-" + side) / "sequences.fasta", tmp_path / ("cpu-" + side) / "sequences.fasta", shallow=False)
    # same directory names on both sides so that the first output line (which echoes them) is comparable
    os.rename(tmp_path / "gpu-l", tmp_path / "profiles-l")
    os.rename(tmp_path / "gpu-r", tmp_path / "profiles-r")
    r = subprocess.run([exe, "translate", "-profiles_left", "profiles-l", "-profiles_right", "profiles-r", "-nucmer_list", "nucmer.list",
                        "-out_delta", "gpu.delta"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    cpu = ref if os.path.exists(ref) else os.path.join(oracle_build, "oracle_m_translate")
    assert subprocess.run([cpu, "profiles-l", "profiles-r", "nucmer.list", "cpu.delta"], cwd=tmp_path).returncode == 0
    assert filecmp.cmp(tmp_path / "gpu.delta", tmp_path / "cpu.delta", shallow=False)
    assert os.path.getsize(tmp_path / "gpu.delta") > 2000
