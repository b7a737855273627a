"""The multi-device entries of the C ABI (include/paramugsy_amd.h, csrc/multi.hpp): a device list instead of one process per GPU.

CPU part: the partition rule and the seam join through the host-only entries (pm_partition, pm_delta_join_files), against
shard.py's rule and against a single run of the CPU oracle over the whole list.
GPU part (-m gpu): devices = {0, 0} -- two host threads, two slices, one GPU -- must print the bytes of the single-device call,
for the translate path and for the DP (host columns, MAF blocks in memory, MAF files); a failing slice fails the call."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from paramugsy_amd import capi, shard, synth


def join_files(parts, out):
    arr = (C.c_char_p * len(parts))(*[p.encode() for p in parts])
    capi.check(capi.lib().pm_delta_join_files(arr, len(parts), out.encode()))


def test_partition_is_shard_pys_rule():
    lo, hi = C.c_int64(), C.c_int64()
    for n in (0, 1, 7, 8, 9, 100, 12345):
        for w in (1, 2, 3, 8):
            for r in range(w):
                capi.check(capi.lib().pm_partition(n, w, r, C.byref(lo), C.byref(hi)))
                assert (lo.value, hi.value) == shard.partition(n, w, r)
    assert capi.lib().pm_partition(5, 0, 0, C.byref(lo), C.byref(hi)) == capi.PM_E_INVALID
    assert capi.lib().pm_partition(5, 2, 2, C.byref(lo), C.byref(hi)) == capi.PM_E_INVALID


def test_weighted_partition_cuts_at_equal_cells_and_is_shard_pys_rule():
    """pm_partition_weighted (csrc/multi.hpp) and shard.partition_weighted are one rule: contiguous slices, cut where the running sum
    of the weights comes closest to k / parts of the total; equal weights give pm_partition's slices.  On the ragged stand-in for
    BASELINE configs[2]/[3] (100 000 pairs, eight slices) every slice is within 2 % of an eighth of the cells -- by count the slices
    hold equal numbers of pairs, whatever their lengths."""
    from paramugsy_amd import dp
    lib = capi.lib()
    lib.pm_partition_weighted.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]

    def c_cuts(w, parts):
        w = np.ascontiguousarray(w, dtype=np.int64)
        cuts = np.zeros(parts + 1, dtype=np.int64)
        capi.check(lib.pm_partition_weighted(w.ctypes.data, len(w), parts, cuts.ctypes.data))
        return cuts.tolist()
    rng = np.random.default_rng(8)
    for trial in range(300):
        n = int(rng.integers(0, 60))
        parts = int(rng.integers(1, 9))
        kind = trial % 4
        w = (rng.integers(0, 5, n) if kind == 0 else rng.integers(1, 10**6, n) if kind == 1 else np.full(n, int(rng.integers(0, 9)))
             if kind == 2 else rng.integers(0, 2**62 // max(n, 1), n))
        cuts = c_cuts(w, parts)
        assert cuts == shard.partition_weighted(w.tolist(), parts)
        assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))
        if kind == 2:
            assert cuts == [shard.partition(n, parts, r)[0] for r in range(parts)] + [n]
    la, lb = dp.ragged_lengths(20261003, 100000)
    w = shard.pair_weights(la, lb)
    cuts = c_cuts(w, 8)
    assert cuts == shard.partition_weighted(w.tolist(), 8)
    cells = np.array([int((la[a:b] * lb[a:b]).sum()) for a, b in zip(cuts, cuts[1:])], dtype=np.float64)
    assert np.all(np.abs(cells / cells.mean() - 1) < 0.02), cells / cells.mean()
    by_count = np.array([int((la[a:b] * lb[a:b]).sum()) for a, b in (shard.partition(100000, 8, r) for r in range(8))], dtype=np.float64)
    assert np.abs(by_count / by_count.mean() - 1).max() < 0.05  # (a seeded i.i.d. batch is balanced by count, too: the rule matters for sorted or clustered lists)
    # a list sorted by length, as a producer that groups its segments would hand over: by count the first slice has 4 x the cells of the last
    order = np.argsort(-(la * lb), kind="stable")
    ws = w[order]
    cs = c_cuts(ws, 8)
    cells_sorted = np.array([int((la[order][a:b] * lb[order][a:b]).sum()) for a, b in zip(cs, cs[1:])], dtype=np.float64)
    assert np.all(np.abs(cells_sorted / cells_sorted.mean() - 1) < 0.02)
    assert lib.pm_partition_weighted(None, 3, 2, np.zeros(3, dtype=np.int64).ctypes.data) == capi.PM_E_INVALID
    assert lib.pm_partition_weighted(np.array([1, -1], dtype=np.int64).ctypes.data, 2, 2, np.zeros(3, dtype=np.int64).ctypes.data) == capi.PM_E_INVALID


def test_join_drops_a_repeated_header_at_a_seam(tmp_path):
    head = b"l/sequences.fasta r/sequences.fasta\nNUCMER\n"
    texts = [head + b">x y 10 10\n1 2 3 4 1 2 3\n0\n",
             head + b">x y 10 10\n5 6 7 8 1 2 3\n0\n>x z 10 9\n1 1 1 1 1 2 3\n0\n",
             head,
             head + b">x z 10 9\n2 2 2 2 1 2 3\n0\n",
             head + b">x zz 10 9\n3 3 3 3 1 2 3\n-1\n0\n"]  # a name that only starts like the one in force
    paths = []
    for k, t in enumerate(texts):
        p = str(tmp_path / ("part%d.delta" % k))
        open(p, "wb").write(t)
        paths.append(p)
    out = str(tmp_path / "joined.delta")
    join_files(paths, out)
    got = open(out, "rb").read()
    assert got == shard.merge_delta_outputs(texts)
    assert got == head + (b">x y 10 10\n1 2 3 4 1 2 3\n0\n5 6 7 8 1 2 3\n0\n>x z 10 9\n1 1 1 1 1 2 3\n0\n2 2 2 2 1 2 3\n0\n"
                          b">x zz 10 9\n3 3 3 3 1 2 3\n-1\n0\n")
    join_files([], out)
    assert open(out, "rb").read() == b""
    with pytest.raises(capi.PmError):
        join_files([str(tmp_path / "missing.delta")], out)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_joined_slices_equal_one_run_over_the_whole_list(world, oracle_build, tmp_path):
    """The CPU oracle's m_translate twin over contiguous slices of a delta-file list, joined by pm_delta_join_files, prints the
    bytes of one run over the whole list (several delta files share header pairs across the seams: the workload has 2 + 2
    genomes and 7 delta files)."""
    w = synth.make_workload(str(tmp_path / "job"), 77, n_left=2, n_right=2, genome_len=20000, n_blocks=10, n_deltas=7,
                            entries_per_delta=25, mean_len=700)
    exe = os.path.join(oracle_build, "oracle_m_translate")
    whole = str(tmp_path / "whole.delta")
    subprocess.run([exe, w.left_dir, w.right_dir, w.list_path, whole], check=True)
    parts = []
    for r in range(world):
        lo, hi = shard.partition(len(w.delta_paths), world, r)
        lp = str(tmp_path / ("list%d.txt" % r))
        open(lp, "w").write("".join(p + "\n" for p in w.delta_paths[lo:hi]))
        out = str(tmp_path / ("part%d.delta" % r))
        subprocess.run([exe, w.left_dir, w.right_dir, lp, out], check=True)
        parts.append(out)
    joined = str(tmp_path / "joined.delta")
    join_files(parts, joined)
    assert open(joined, "rb").read() == open(whole, "rb").read()
    assert open(whole, "rb").read().count(b">") > 3


# ------------------------------------------------------------------ on the GPU

@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_translate_over_a_device_list_prints_the_single_device_bytes(devices, tmp_path):
    from paramugsy_amd.translate import translate, translate_multi
    w = synth.make_workload(str(tmp_path / "job"), 91, n_left=3, n_right=3, genome_len=60000, n_blocks=40, n_deltas=7,
                            entries_per_delta=120, mean_len=900)
    one = str(tmp_path / "one.delta")
    translate(w.left_dir, w.right_dir, w.delta_paths, one)
    many = str(tmp_path / "many.delta")
    translate_multi(w.left_dir, w.right_dir, w.delta_paths, many, devices)
    assert open(many, "rb").read() == open(one, "rb").read()
    assert os.path.getsize(one) > 20000
    # the golden job printed by the upstream binary, through the device list
    case = os.path.join(GOLDEN, "translate_typical")
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    out = str(tmp_path / "golden.delta")
    translate_multi(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas, out, devices)
    body = open(out, "rb").read().split(b"\n", 2)[2]
    assert body == open(os.path.join(case, "expected.delta"), "rb").read().split(b"\n", 2)[2]


@pytest.mark.gpu
def test_translate_cli_takes_a_device_list(tmp_path):
    w = synth.make_workload(str(tmp_path / "job"), 92, n_left=2, n_right=2, genome_len=30000, n_blocks=20, n_deltas=4,
                            entries_per_delta=60, mean_len=800)
    one, env_many, flag_many = (str(tmp_path / n) for n in ("one.delta", "env.delta", "flag.delta"))
    subprocess.run([os.path.join(ROOT, "bin", "m_translate"), w.left_dir, w.right_dir, w.list_path, one], check=True)
    subprocess.run([os.path.join(ROOT, "bin", "m_translate"), w.left_dir, w.right_dir, w.list_path, env_many], check=True,
                   env=dict(os.environ, PARAMUGSY_DEVICES="0,0"))
    subprocess.run([os.path.join(ROOT, "bin", "mugsy_profiles"), "translate", "-profiles_left", w.left_dir, "-profiles_right", w.right_dir,
                    "-nucmer_list", w.list_path, "-out_delta", flag_many, "-devices", "0,0,0"], check=True)
    assert open(env_many, "rb").read() == open(one, "rb").read() == open(flag_many, "rb").read()
    r = subprocess.run([os.path.join(ROOT, "bin", "mugsy_profiles"), "translate", "-profiles_left", w.left_dir, "-profiles_right", w.right_dir,
                        "-nucmer_list", w.list_path, "-out_delta", flag_many, "-devices", "0,x"], capture_output=True)
    assert r.returncode == 2


@pytest.mark.gpu
def test_a_failing_slice_fails_the_call_and_keeps_what_precedes_it(tmp_path):
    """A delta file the parser cannot read in the SECOND slice: the reference, run over the whole list, would have printed every
    entry before it and died (m_translate.cc:722-728).  The device list does the same: slice 0 whole, slice 1 up to the failure,
    nothing of slice 2, and the call fails naming the device."""
    from paramugsy_amd.translate import translate, translate_multi
    w = synth.make_workload(str(tmp_path / "job"), 93, n_left=2, n_right=2, genome_len=30000, n_blocks=20, n_deltas=6,
                            entries_per_delta=50, mean_len=800)
    text = open(w.delta_paths[3]).read().split("\n")
    text[len(text) // 2] = "this is not an offset"
    open(w.delta_paths[3], "w").write("\n".join(text))
    one, many = str(tmp_path / "one.delta"), str(tmp_path / "many.delta")
    with pytest.raises(capi.PmError) as e1:
        translate(w.left_dir, w.right_dir, w.delta_paths, one)
    with pytest.raises(capi.PmError) as e3:
        translate_multi(w.left_dir, w.right_dir, w.delta_paths, many, [0, 0, 0])
    assert e1.value.code == e3.value.code == capi.PM_E_PARSE and "worker 1" in str(e3.value)
    assert open(many, "rb").read() == open(one, "rb").read()
    with pytest.raises(capi.PmError) as e:
        translate_multi(w.left_dir, w.right_dir, w.delta_paths, many, [0, 7])
    assert e.value.code == capi.PM_E_INVALID
    with pytest.raises(capi.PmError):
        translate_multi(w.left_dir, w.right_dir, w.delta_paths, many, [])


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_dp_over_a_device_list_equals_the_resident_batch(devices, oracle_build):
    import pyoracle
    from paramugsy_amd import dp
    la, lb = dp.ragged_lengths(5, 203, median=600, lo=1, hi=2500)
    la[17] = 0
    lb[100] = 0
    inputs = dp.synth_batch(6, la, lb, 3, 3)
    params = dp.make_params(3, 3)
    batch = dp.DpBatch(inputs, params)
    batch.run(True)
    r_scores, r_ops, r_nops = batch.fetch()
    batch.close()
    scores, ops, n_ops = dp.align_multi(inputs, params, devices)
    assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
    for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):
        assert np.array_equal(p, q)
    k = 40
    o_scores, o_paths = pyoracle.dp_align(shard.slice_pairs(inputs, 0, k), params)
    assert np.array_equal(scores[:k], o_scores)
    assert all(np.array_equal(p, q) for p, q in zip(dp.paths_of(inputs, ops, n_ops)[:k], o_paths))
    s2, _, _ = dp.align_multi(inputs, params, devices, with_paths=False)
    assert np.array_equal(s2, r_scores)
    # fewer pairs than workers: some slices are empty
    few = shard.slice_pairs(inputs, 0, 3)
    s3, o3, n3 = dp.align_multi(few, params, devices)
    assert np.array_equal(s3, r_scores[:3]) and np.array_equal(n3, r_nops[:3])
    # limits are checked before any worker starts
    with pytest.raises(capi.PmError):
        dp.align_multi(inputs, dp.make_params(3, 3, open_per_pair=-1), devices)


@pytest.mark.gpu
def test_eight_way_rehearsal_of_the_device_list_on_one_gpu(oracle_build):
    """the first real 8-GPU run must not be the first 8-way run of the code.  pm_dp_align_multi over an
    EIGHT-entry device list {0, ..., 0} (eight host threads and HIP contexts in one process -- the GPU box allows six PROCESSES on
    its card, so the eight-rank bench is rehearsed with gloo ranks on the CPU, tests/test_shard_gloo.py, and with six ranks on the
    card, tests/test_bench_rehearsal_gpu.py): on a reduced stand-in for configs[3] (2 000 ragged pairs) the eight weighted cuts are
    within 2 % of equal cells, the gathered scores and paths hash-equal to the one-device batch, and a slice that fails fails the call."""
    import hashlib
    from paramugsy_amd import dp
    la, lb = dp.ragged_lengths(20261003, 2000, median=400, lo=50, hi=2000)
    inputs = dp.synth_batch(7, la, lb, 4, 4)
    params = dp.make_params(4, 4)
    cuts = shard.partition_weighted(shard.pair_weights(la, lb).tolist(), 8)
    cells = [int((la[a:b] * lb[a:b]).sum()) for a, b in zip(cuts, cuts[1:])]
    assert len(cells) == 8 and max(cells) <= 1.02 * sum(cells) / 8 and min(cells) >= 0.98 * sum(cells) / 8
    batch = dp.DpBatch(inputs, params)
    batch.run(True)
    r_scores, r_ops, r_nops = batch.fetch()
    batch.close()
    digest = lambda *arrays: hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in arrays)).hexdigest()
    want = digest(r_scores, r_nops, *dp.paths_of(inputs, r_ops, r_nops))
    scores, ops, n_ops = dp.align_multi(inputs, params, [0] * 8)
    assert digest(scores, n_ops, *dp.paths_of(inputs, ops, n_ops)) == want
    # a pair the library refuses (a profile of 300 rows with weights that leave int16) in the LAST slice: every slice's work is thrown
    # away, the call fails, and the error names what failed
    bad = dp.DpInputs(inputs.cols_a.copy(), inputs.off_a, inputs.cols_b.copy(), inputs.off_b)
    bad.cols_b[int(inputs.off_b[1990]), 0] = 255
    bad.cols_b[int(inputs.off_b[1990]), 1] = 255
    with pytest.raises(capi.PmError):
        dp.align_multi(bad, dp.make_params(4, 4, match=127, mismatch=-127), [0] * 8)
    # and the list may name more workers than there are pairs
    few = shard.slice_pairs(inputs, 0, 5)
    s3, o3, n3 = dp.align_multi(few, params, [0] * 8)
    assert np.array_equal(s3, r_scores[:5]) and np.array_equal(n3, r_nops[:5])


@pytest.mark.gpu
def test_maf_blocks_over_a_device_list(oracle_build, tmp_path):
    from paramugsy_amd import dp
    from test_dp_maf import random_blocks
    rng = np.random.default_rng(8)
    A = random_blocks(rng, 11, max_rows=4, max_cols=200)
    B = random_blocks(rng, 11, max_rows=4, max_cols=200)
    params = dp.make_params(2, 2)
    s1, m1 = dp.align_blocks(A, B, params)
    for devices in ([0], [0, 0], [0, 0, 0]):
        s2, m2 = dp.align_blocks_multi(A, B, params, devices)
        assert np.array_equal(s1, s2) and m1 == m2

    def write(path, blocks, tag):
        with open(path, "wb") as f:
            f.write(b"##maf version=1\n")
            for k, b in enumerate(blocks):
                f.write(b"a score=0\n")
                for r, row in enumerate(b):
                    f.write(b"s %s.g%d %d %d + 100000 %s\n" % (tag, r, 10 * k, sum(ch not in b"-" for ch in row), row))
                f.write(b"\n")
    pa, pb = str(tmp_path / "a.maf"), str(tmp_path / "b.maf")
    write(pa, A, b"L")
    write(pb, B, b"R")
    one, many, cli = (str(tmp_path / n) for n in ("one.maf", "many.maf", "cli.maf"))
    dp.align_maf_files(pa, pb, params, one)
    dp.align_maf_files(pa, pb, params, many, devices=[0, 0, 0])
    assert open(one, "rb").read() == open(many, "rb").read() and os.path.getsize(one) > 1000
    # the CLI: `mugsy_profiles align ... -devices` (default penalties = make_params(1, 1) scaled by -rows^2)
    dp.align_maf_files(pa, pb, dp.make_params(2, 2), one)
    subprocess.run([os.path.join(ROOT, "bin", "mugsy_profiles"), "align", "-left_maf", pa, "-right_maf", pb, "-out_maf", cli, "-rows", "2",
                    "-devices", "0,0"], check=True)
    assert open(cli, "rb").read() == open(one, "rb").read()


@pytest.mark.gpu
def test_output_that_cannot_be_seeked_and_released_caches(tmp_path):
    """The job's text normally goes to the output file piece by piece through the descriptor, each piece at its own place; a sink that
    cannot seek (a named pipe) gets the pieces in order instead -- same bytes.  pm_release_caches() drops the kept staging and
    device buffers, and the next call builds them again."""
    import threading
    from paramugsy_amd.translate import translate
    w = synth.make_workload(str(tmp_path / "job"), 94, n_left=3, n_right=3, genome_len=60000, n_blocks=40, n_deltas=5,
                            entries_per_delta=150, mean_len=900)
    plain = str(tmp_path / "plain.delta")
    translate(w.left_dir, w.right_dir, w.delta_paths, plain)
    want = open(plain, "rb").read()
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    got = []
    reader = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    reader.start()
    translate(w.left_dir, w.right_dir, w.delta_paths, fifo)
    reader.join(timeout=60)
    assert got and got[0] == want
    assert len(want) > 2000
    assert capi.lib().pm_release_caches() == capi.PM_OK
    again = str(tmp_path / "again.delta")
    translate(w.left_dir, w.right_dir, w.delta_paths, again)
    assert open(again, "rb").read() == want
    assert capi.lib().pm_release_caches() == capi.PM_OK
