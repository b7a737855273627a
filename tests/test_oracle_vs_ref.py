"""Live diff of the oracle against the upstream reference binaries (oracle/_ref) on fresh seeded inputs.
Skipped where oracle/_ref is absent.  CPU only."""
import filecmp
import os
import subprocess

import pytest

from paramugsy_amd import synth

MODES = {
    "typical": dict(),
    "gappy": dict(gap_rate=0.05, mean_gap=6.0, indel_rate=0.02, mean_indel=4.0, adjacent_prob=0.1, edge_gap_prob=0.5),
    "reverse": dict(genome_len=8000, n_blocks=30, mean_cols=120, gap_rate=0.08, indel_rate=0.05, mean_len=400,
                    entries_per_delta=80, rev_prob=0.5, delta_rev_prob=0.5, spacing=5),
    "tiny_blocks": dict(genome_len=3000, n_blocks=150, mean_cols=8, gap_rate=0.1, mean_gap=3.0, indel_rate=0.05, mean_indel=8.0,
                        mean_len=150, entries_per_delta=60, spacing=3, edge_gap_prob=0.4, adjacent_prob=0.15, delta_rev_prob=0.4),
}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_m_translate_bytes(mode, seed, oracle_build, ref_dir, tmp_path):
    w = synth.make_workload(str(tmp_path / "job"), seed * 1000 + len(mode), **MODES[mode])
    a, b = str(tmp_path / "ref.delta"), str(tmp_path / "ora.delta")
    ra = subprocess.run([os.path.join(ref_dir, "m_translate"), w.left_dir, w.right_dir, w.list_path, a])
    rb = subprocess.run([os.path.join(oracle_build, "oracle_m_translate"), w.left_dir, w.right_dir, w.list_path, b])
    assert ra.returncode == 0 and rb.returncode == 0
    assert filecmp.cmp(a, b, shallow=False)
    assert os.path.getsize(a) > 200  # the case is not vacuous
