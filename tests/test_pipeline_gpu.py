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


def _stage_inputs(tmp_path, seed=7, blocks=60, entries=150):
    rng = np.random.default_rng(seed)
    lg, rg = ["L0.chr", "L1.chr"], ["R0.chr", "R1.chr"]
    for side, names in (("l", lg), ("r", rg)):
        (tmp_path / ("%s.maf" % side)).write_text(
            synth.side_to_maf_text(synth.gen_side(rng, names, 40000, blocks, mean_cols=300, gap_rate=0.03)))
    paths = []
    for d in range(2):
        p = tmp_path / ("n%d.delta" % d)
        p.write_text(synth.gen_delta_text(rng, lg, rg, 40000, 40000, entries, mean_len=900))
        paths.append(str(p))
    (tmp_path / "nucmer.list").write_text("".join(p + "\n" for p in paths))


def _three_processes(exe, tmp_path, tag):
    for side in ("l", "r"):
        r = subprocess.run([exe, "make", "-in_maf", "%s.maf" % side, "-out_dir", "%s-%s" % (tag, side), "-basename", side], cwd=tmp_path,
                           capture_output=True)
        assert r.returncode == 0, r.stderr


def test_stage_in_one_process_writes_the_same_bytes_as_three(oracle_build, tmp_path):
    """`mugsy_profiles stage` (make + make + translate in one process, rows handed over in memory) against the three commands
    of lib/base/mugsy_profiles_task.ml:40-58 run one after the other, and against the CPU translate on the same profiles."""
    exe = os.path.join(ROOT, "bin", "mugsy_profiles")
    _stage_inputs(tmp_path)
    _three_processes(exe, tmp_path, "profiles")  # profiles-l, profiles-r
    r = subprocess.run([exe, "translate", "-profiles_left", "profiles-l", "-profiles_right", "profiles-r", "-nucmer_list", "nucmer.list",
                        "-out_delta", "three.delta"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    os.rename(tmp_path / "profiles-l", tmp_path / "three-l")
    os.rename(tmp_path / "profiles-r", tmp_path / "three-r")
    r = subprocess.run([exe, "stage", "-left_maf", "l.maf", "-left_dir", "profiles-l", "-left_basename", "l", "-right_maf", "r.maf",
                        "-right_dir", "profiles-r", "-right_basename", "r", "-nucmer_list", "nucmer.list", "-out_delta", "stage.delta"],
                       cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    for side in ("l", "r"):
        for name in ("profiles", "sequences.fasta", "profiles.soa"):
            assert filecmp.cmp(tmp_path / ("profiles-" + side) / name, tmp_path / ("three-" + side) / name, shallow=False), (side, name)
    assert filecmp.cmp(tmp_path / "stage.delta", tmp_path / "three.delta", shallow=False)
    assert os.path.getsize(tmp_path / "stage.delta") > 5000
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    cpu = ref if os.path.exists(ref) else os.path.join(oracle_build, "oracle_m_translate")
    assert subprocess.run([cpu, "profiles-l", "profiles-r", "nucmer.list", "cpu.delta"], cwd=tmp_path).returncode == 0
    assert filecmp.cmp(tmp_path / "stage.delta", tmp_path / "cpu.delta", shallow=False)


def test_binary_side_file_is_used_only_while_it_matches_the_text(oracle_build, tmp_path):
    """m_translate reads <dir>/profiles.soa instead of parsing <dir>/profiles -- same output with and without it -- and goes back
    to the text when the text file has changed since (here: a row removed from the text; the stale side file must be ignored)."""
    exe = os.path.join(ROOT, "bin", "mugsy_profiles")
    mt = os.path.join(ROOT, "bin", "m_translate")
    _stage_inputs(tmp_path, seed=8)
    _three_processes(exe, tmp_path, "profiles")
    run = lambda out, env=None: subprocess.run([mt, "profiles-l", "profiles-r", "nucmer.list", out], cwd=tmp_path, capture_output=True,
                                                env=dict(os.environ, **(env or {})))
    assert run("with.delta").returncode == 0
    assert run("without.delta", {"PM_NO_SOA": "1"}).returncode == 0
    assert filecmp.cmp(tmp_path / "with.delta", tmp_path / "without.delta", shallow=False)
    # drop the first record of the left text file: header line, gap lines, "0", text line
    lines = (tmp_path / "profiles-l" / "profiles").read_text().split("\n")
    cut = lines.index("0") + 2
    (tmp_path / "profiles-l" / "profiles").write_text("\n".join(lines[cut:]))
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    cpu = ref if os.path.exists(ref) else os.path.join(oracle_build, "oracle_m_translate")
    assert subprocess.run([cpu, "profiles-l", "profiles-r", "nucmer.list", "cpu.delta"], cwd=tmp_path).returncode == 0
    assert run("stale.delta").returncode == 0
    assert filecmp.cmp(tmp_path / "stale.delta", tmp_path / "cpu.delta", shallow=False)
    assert not filecmp.cmp(tmp_path / "stale.delta", tmp_path / "with.delta", shallow=False)


def test_resident_worker_serves_several_nodes(oracle_build, tmp_path):
    """`mugsy_profiles serve`: one process, one HIP context, several tree nodes; every answer `done 0`, outputs as the one-shot
    commands write them; a bad command answers non-zero and the worker lives on."""
    exe = os.path.join(ROOT, "bin", "mugsy_profiles")
    nodes = []
    for k in range(2):
        d = tmp_path / ("node%d" % k)
        d.mkdir()
        _stage_inputs(d, seed=20 + k, blocks=30, entries=60)
        r = subprocess.run([exe, "stage", "-left_maf", "l.maf", "-left_dir", "one-l", "-left_basename", "l", "-right_maf", "r.maf",
                            "-right_dir", "one-r", "-right_basename", "r", "-nucmer_list", "nucmer.list", "-out_delta", "one.delta"], cwd=d,
                           capture_output=True)
        assert r.returncode == 0, r.stderr
        nodes.append(d)
    cmds = []
    for d in nodes:
        cmds.append("\t".join(["stage", "-left_maf", str(d / "l.maf"), "-left_dir", str(d / "srv-l"), "-left_basename", "l", "-right_maf",
                               str(d / "r.maf"), "-right_dir", str(d / "srv-r"), "-right_basename", "r", "-nucmer_list",
                               str(d / "nucmer.list"), "-out_delta", str(d / "srv.delta")]))
    cmds.insert(1, "make\t-in_maf\t/nonexistent.maf\t-out_dir\t%s\t-basename\tx" % (tmp_path / "bad"))
    r = subprocess.run([exe, "serve"], input=("\n".join(cmds) + "\nquit\n").encode(), capture_output=True, timeout=300)
    answers = [ln for ln in r.stdout.decode().split("\n") if ln.startswith("done")]
    assert answers == ["done 0", "done 2", "done 0"], (answers, r.stderr)
    for d in nodes:
        # the first line of a delta file echoes the directory names: compare everything after it
        a = (d / "one.delta").read_bytes().split(b"\n", 1)[1]
        b = (d / "srv.delta").read_bytes().split(b"\n", 1)[1]
        assert a == b and len(a) > 1000
        assert filecmp.cmp(d / "one-l" / "profiles", d / "srv-l" / "profiles", shallow=False)
