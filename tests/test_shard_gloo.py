"""The N>1 path on CPU: two `gloo` ranks each translate a contiguous slice of the delta-file list and rank 0
gathers; the merged bytes equal a single run over the whole list.  The per-shard compute here is the CPU oracle
(injected as translate_fn) because no GPU exists in this container; on GPUs the same code calls the HIP path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from paramugsy_amd import shard, synth

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch.distributed as dist
import pyoracle
from paramugsy_amd import shard
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[3], rank=rank, world_size=world)
paths = [p for p in open(sys.argv[6]).read().split("\n") if p]
def fn(l, r, ps, out):
    rc = pyoracle.translate_files(l, r, list(ps), out)
    assert rc == 0
shard.translate_sharded(sys.argv[4], sys.argv[5], paths, sys.argv[7], rank, world, dist=dist, translate_fn=fn)
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_contiguous_and_balanced():
    for n in (0, 1, 7, 8, 9, 100):
        for w in (1, 2, 3, 8):
            parts = [shard.partition(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[k][1] == parts[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_merge_drops_a_repeated_header_at_a_seam():
    head = b"l/sequences.fasta r/sequences.fasta\nNUCMER\n"
    a = head + b">x y 10 10\n1 2 3 4 1 2 3\n0\n"
    b = head + b">x y 10 10\n5 6 7 8 1 2 3\n0\n>x z 10 9\n1 1 1 1 1 2 3\n0\n"
    c = head
    d = head + b">x z 10 9\n2 2 2 2 1 2 3\n0\n"
    merged = shard.merge_delta_outputs([a, b, c, d])
    assert merged == head + b">x y 10 10\n1 2 3 4 1 2 3\n0\n5 6 7 8 1 2 3\n0\n>x z 10 9\n1 1 1 1 1 2 3\n0\n2 2 2 2 1 2 3\n0\n"


@pytest.mark.parametrize("n_deltas", [2, 5])
def test_two_gloo_ranks_reproduce_the_single_process_bytes(n_deltas, oracle_build, tmp_path):
    import pyoracle
    # few genomes and many entries per file so that consecutive files often continue under the same header
    w = synth.make_workload(str(tmp_path / "job"), 900 + n_deltas, n_left=1, n_right=1, genome_len=20000, n_blocks=6,
                            mean_cols=3000, n_deltas=n_deltas, entries_per_delta=12, mean_len=900)
    single = str(tmp_path / "single.delta")
    assert pyoracle.translate_files(w.left_dir, w.right_dir, w.delta_paths, single) == 0
    merged = str(tmp_path / "merged.delta")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, w.left_dir, w.right_dir, w.list_path, merged])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    assert open(merged, "rb").read() == open(single, "rb").read()
    assert os.path.getsize(merged) > 300
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    if os.path.exists(ref):
        out = str(tmp_path / "ref.delta")
        subprocess.run([ref, w.left_dir, w.right_dir, w.list_path, out], check=True)
        assert open(merged, "rb").read() == open(out, "rb").read()


GPU_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
from paramugsy_amd import shard
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[3], rank=rank, world_size=world)
paths = [p for p in open(sys.argv[6]).read().split("\n") if p]
shard.translate_sharded(sys.argv[4], sys.argv[5], paths, sys.argv[7], rank, world, dist=dist, device=0)  # the HIP path
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_reproduce_the_single_process_bytes(oracle_build, tmp_path):
    """The real N>1 path: two processes, each translating its slice of the delta-file list on the GPU (both on device 0
    here: the box has one), host-side gather over gloo, merged bytes equal one upstream/oracle run over the whole list."""
    w = synth.make_workload(str(tmp_path / "job"), 77, n_left=2, n_right=2, genome_len=60000, n_blocks=40, n_deltas=5,
                            entries_per_delta=60, mean_len=1200)
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    cpu = ref if os.path.exists(ref) else os.path.join(oracle_build, "oracle_m_translate")
    single = str(tmp_path / "single.delta")
    assert subprocess.run([cpu, w.left_dir, w.right_dir, w.list_path, single]).returncode == 0
    merged = str(tmp_path / "merged.delta")
    script = tmp_path / "worker.py"
    script.write_text(GPU_WORKER.format(root=ROOT))
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, w.left_dir, w.right_dir, w.list_path, merged])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    assert open(merged, "rb").read() == open(single, "rb").read()
    assert os.path.getsize(merged) > 5000


DP_WORKER = r"""
import os, pickle, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import numpy as np
import torch.distributed as dist
from paramugsy_amd import dp, shard
rank, world, use_gpu = int(sys.argv[1]), int(sys.argv[2]), sys.argv[5] == "gpu"
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[3], rank=rank, world_size=world)
inputs = dp.synth_pairs(31, 23, 3, 180, indel_rate=0.03, vary_length=True)   # every rank builds the same batch
params = dp.make_params(3, 3)
fn = None
if not use_gpu:
    import pyoracle
    fn = lambda sub, p: pyoracle.dp_align(sub, p)
scores, paths = shard.align_sharded(inputs, params, rank, world, dist=dist, device=0, align_fn=fn)
if rank == 0:
    pickle.dump((np.asarray(scores), [np.asarray(p) for p in paths]), open(sys.argv[4], "wb"))
else:
    assert scores is None and paths is None
dist.barrier()
dist.destroy_process_group()
"""


def run_dp_ranks(tmp_path, mode, world=2):
    import pickle
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER.format(root=ROOT))
    out = str(tmp_path / "gathered.pkl")
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, out, mode]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    return pickle.load(open(out, "rb"))


def check_dp_gather(scores, paths):
    import pyoracle
    from paramugsy_amd import dp
    inputs = dp.synth_pairs(31, 23, 3, 180, indel_rate=0.03, vary_length=True)
    o_scores, o_paths = pyoracle.dp_align(inputs, dp.make_params(3, 3))
    assert np.array_equal(scores, o_scores) and len(paths) == len(o_paths) == 23
    assert all(np.array_equal(a, b) for a, b in zip(paths, o_paths))


@pytest.mark.parametrize("world", [2, 3])
def test_dp_pair_partition_gathers_in_pair_order(world, oracle_build, tmp_path):
    """The DP's N>1 path on CPU: `world` gloo ranks align contiguous slices of the pair list (the oracle stands in for
    the HIP path), rank 0 gathers; scores and paths come back in pair order and equal one run over all pairs."""
    check_dp_gather(*run_dp_ranks(tmp_path, "cpu", world))


def test_dp_eight_gloo_ranks_gather_in_pair_order(oracle_build, tmp_path):
    """The world size the scaling run has (8), on the CPU: eight gloo ranks, 23 pairs (some ranks get two, some three), gathered in pair
    order -- the census, the cuts and the point-to-point gather of shard.align_sharded at the width no GPU box of this build can run."""
    check_dp_gather(*run_dp_ranks(tmp_path, "cpu", 8))


@pytest.mark.gpu
def test_dp_two_ranks_on_the_gpu(oracle_build, tmp_path):
    check_dp_gather(*run_dp_ranks(tmp_path, "gpu"))


FAIL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch.distributed as dist
from paramugsy_amd import dp, shard
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[3], rank=rank, world_size=world)
inputs = dp.synth_pairs(31, 9, 2, 60)
def fn(sub, p):
    if rank == 1:
        raise ValueError("boom on rank 1")
    import pyoracle
    return pyoracle.dp_align(sub, p)
try:
    shard.align_sharded(inputs, dp.make_params(2, 2), rank, world, dist=dist, align_fn=fn)
    rc = 0
except RuntimeError as e:
    rc = 7
dist.destroy_process_group()
sys.exit(rc)
"""


def test_a_failing_rank_makes_every_rank_raise_instead_of_hanging(oracle_build, tmp_path):
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER.format(root=ROOT))
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port]) for r in range(2)]
    assert [p.wait(timeout=120) for p in procs] == [7, 7]


BLOCK_WORKER = r"""
import os, pickle, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import numpy as np
import torch.distributed as dist
from paramugsy_amd import dp, shard
rank, world, use_gpu = int(sys.argv[1]), int(sys.argv[2]), sys.argv[5] == "gpu"
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[3], rank=rank, world_size=world)
A, B = pickle.load(open(sys.argv[6], "rb"))
params = dp.make_params(2, 2)
fn = None
if not use_gpu:
    import pyoracle, dp_maf_oracle as ora
    def fn(sa, sb, p):
        scores, merged = [], []
        for a, b in zip(sa, sb):
            ca = np.array(ora.pack_block(a), dtype=np.uint8).reshape(-1, 8); cb = np.array(ora.pack_block(b), dtype=np.uint8).reshape(-1, 8)
            one = dp.DpInputs(ca, np.array([0, len(ca)], dtype=np.int64), cb, np.array([0, len(cb)], dtype=np.int64))
            s, pth = pyoracle.dp_align(one, p)
            scores.append(int(s[0])); merged.append(ora.emit_block(a, b, pth[0].tolist()))
        return np.array(scores, dtype=np.int32), merged
scores, blocks = shard.align_blocks_sharded(A, B, params, rank, world, dist=dist, device=0, block_fn=fn)
if rank == 0:
    pickle.dump((np.asarray(scores), blocks), open(sys.argv[4], "wb"))
else:
    assert scores is None and blocks is None
dist.barrier()
dist.destroy_process_group()
"""


def run_block_ranks(tmp_path, mode, world):
    import pickle
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b"ACGTacgt-N", dtype=np.uint8)
    mk = lambda: [[alphabet[rng.integers(0, 10, size=c)].tobytes() for _ in range(int(rng.integers(1, 4)))]
                  for c in rng.integers(1, 90, size=11)]
    A, B = mk(), mk()
    inp = str(tmp_path / "blocks.pkl")
    pickle.dump((A, B), open(inp, "wb"))
    script = tmp_path / "block_worker.py"
    script.write_text(BLOCK_WORKER.format(root=ROOT))
    out = str(tmp_path / "gathered_blocks.pkl")
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, out, mode, inp]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    return A, B, pickle.load(open(out, "rb"))


def check_block_gather(A, B, got):
    import pyoracle
    import dp_maf_oracle as ora
    from paramugsy_amd import dp
    scores, blocks = got
    assert len(blocks) == len(A) == 11
    for k in range(len(A)):
        ca = np.array(ora.pack_block(A[k]), dtype=np.uint8).reshape(-1, 8)
        cb = np.array(ora.pack_block(B[k]), dtype=np.uint8).reshape(-1, 8)
        one = dp.DpInputs(ca, np.array([0, len(ca)], dtype=np.int64), cb, np.array([0, len(cb)], dtype=np.int64))
        s, p = pyoracle.dp_align(one, dp.make_params(2, 2))
        assert scores[k] == s[0]
        assert [bytes(r) for r in blocks[k]] == ora.emit_block(A[k], B[k], p[0].tolist()), "pair %d" % k


@pytest.mark.parametrize("world", [2, 3])
def test_maf_blocks_gather_in_pair_order(world, oracle_build, tmp_path):
    """MAF blocks in, merged blocks out over `world` gloo ranks (the oracle stands in for the HIP path): rank 0 receives scores
    and merged blocks in pair order, equal to one run over all pairs."""
    check_block_gather(*run_block_ranks(tmp_path, "cpu", world))


@pytest.mark.gpu
def test_maf_blocks_two_ranks_on_the_gpu(oracle_build, tmp_path):
    check_block_gather(*run_block_ranks(tmp_path, "gpu", 2))
