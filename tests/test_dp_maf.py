"""MAF blocks into the profile DP and out of it: pm_dp_pack_maf, pm_dp_emit_maf, pm_dp_align_maf against oracle/dp_maf_oracle.py
(this repo's own restatement: the reference has no DP stage) and against cases worked out by hand below."""
import os

import numpy as np
import pytest

from paramugsy_amd import dp

import dp_maf_oracle as ora


def test_oracle_hand_worked_pack():
    # column by column: "AC-n", "aGTN", "A--T" -> col0 {A:3}, col1 {C:1, G:1, gap:1}, col2 {T:1, gap:2}, col3 {T:1, other:2}
    cols = ora.pack_block([b"AC-n", b"aGTN", b"A--T"])
    assert cols == [[3, 0, 0, 0, 0, 0, 0, 0], [0, 1, 1, 0, 1, 0, 0, 0], [0, 0, 0, 1, 2, 0, 0, 0], [0, 0, 0, 1, 0, 2, 0, 0]]
    assert np.array_equal(np.array(cols, dtype=np.uint8), dp.pack_profile([b"AC-n", b"aGTN", b"A--T"]))


def test_oracle_hand_worked_emit():
    # A = {ACG, A-G} (3 columns), B = {CGT} (3 columns); path M D M I I: A0~B0, A1 alone, A2~B1, B2 alone... spans A (M D M = 3)
    # and B (M M I I would be 4) -> use M D M I: B consumes M, M, I = 3
    rows = ora.emit_block([b"ACG", b"A-G"], [b"CGT"], [0, 2, 0, 1])
    assert rows == [b"ACG-", b"A-G-", b"C-GT"]
    with pytest.raises(AssertionError):
        ora.emit_block([b"ACG"], [b"CGT"], [0, 0])


def random_blocks(rng, n, max_rows=6, max_cols=300):
    alphabet = np.frombuffer(b"ACGTacgt-N-ACGT", dtype=np.uint8)
    blocks = []
    for _ in range(n):
        rows, cols = int(rng.integers(1, max_rows + 1)), int(rng.integers(1, max_cols + 1))
        blocks.append([alphabet[rng.integers(0, len(alphabet), size=cols)].tobytes() for _ in range(rows)])
    return blocks


@pytest.mark.gpu
def test_pack_equals_oracle_on_random_blocks():
    rng = np.random.default_rng(3)
    blocks = random_blocks(rng, 40) + [[b"A"], [b"-", b"n"], [b"acgt" * 700] * 9]
    cols, off = dp.pack_maf(blocks)
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(b[0]) for b in blocks])]).tolist()
    for k, b in enumerate(blocks):
        assert cols[off[k]:off[k + 1]].tolist() == ora.pack_block(b), "block %d" % k
    # every column's six counts sum to the number of rows
    for k, b in enumerate(blocks):
        assert (cols[off[k]:off[k + 1], :6].sum(axis=1) == len(b)).all()


@pytest.mark.gpu
def test_blocks_through_the_dp_and_back(oracle_build):
    """pack -> DP -> emit on random block pairs: the packed columns feed the DP (scores and paths equal the scalar oracle's on the
    same packed columns), and the merged blocks equal the oracle's expansion, row by row; removing the '-' the path inserted
    gives every input row back."""
    import pyoracle
    rng = np.random.default_rng(4)
    A = random_blocks(rng, 25, max_rows=4, max_cols=200)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    B = []
    for a in A:  # B's rows: a's first row with substitutions, cut to one length per block (so the optimal paths carry gaps)
        src = np.frombuffer(a[0].upper().replace(b"N", b"A").replace(b"-", b"C"), dtype=np.uint8)
        keep = max(1, len(src) - int(rng.integers(0, 9)))
        B.append([np.where(rng.random(len(src)) < 0.1, rng.choice(bases, len(src)), src).astype(np.uint8)[:keep].tobytes()
                  for _ in range(int(rng.integers(1, 4)))])
    ca, oa = dp.pack_maf(A)
    cb, ob = dp.pack_maf(B)
    inputs = dp.DpInputs(ca, oa, cb, ob)
    params = dp.make_params(2, 2)
    batch = dp.DpBatch(inputs, params)
    batch.run(True)
    scores, ops, n_ops = batch.fetch()
    paths = batch.paths(ops, n_ops)
    batch.close()
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    assert np.array_equal(scores, o_scores) and all(np.array_equal(p, q) for p, q in zip(paths, o_paths))
    merged = dp.emit_maf(A, B, paths)
    for k in range(len(A)):
        assert merged[k] == ora.emit_block(A[k], B[k], paths[k].tolist()), "pair %d" % k
        for r, row in enumerate(A[k]):
            kept = bytes(ch for ch, op in zip(merged[k][r], paths[k]) if op != 1)
            assert kept == row
        for r, row in enumerate(B[k]):
            kept = bytes(ch for ch, op in zip(merged[k][len(A[k]) + r], paths[k]) if op != 2)
            assert kept == row


@pytest.mark.gpu
def test_align_blocks_in_one_call_equals_the_three_stages(oracle_build):
    """pm_dp_align_blocks (texts to the device once, columns and paths staying there) against pack -> DpBatch -> emit through host
    memory, on random block pairs including empty blocks and one-row blocks; and its refusal of a buffer that is too small."""
    from paramugsy_amd import capi
    rng = np.random.default_rng(14)
    A = random_blocks(rng, 30, max_rows=5, max_cols=260) + [[b"ACGT"], [b"-"], [b"AC-T", b"ACGT"]]
    B = random_blocks(rng, 30, max_rows=3, max_cols=260) + [[b"A"], [b"ACG"], [b"T"]]
    params = dp.make_params(3, 3)
    scores, merged = dp.align_blocks(A, B, params)
    ca, oa = dp.pack_maf(A)
    cb, ob = dp.pack_maf(B)
    batch = dp.DpBatch(dp.DpInputs(ca, oa, cb, ob), params)
    batch.run(traceback=True)
    s2, ops, n_ops = batch.fetch()
    paths = batch.paths(ops, n_ops)
    batch.close()
    assert np.array_equal(scores, s2)
    assert merged == dp.emit_maf(A, B, paths)
    for k in range(len(A)):
        assert merged[k] == ora.emit_block(A[k], B[k], paths[k].tolist())
    s0, m0 = dp.align_blocks([], [], params)
    assert len(s0) == 0 and m0 == []
    # a buffer that is too small: refused, with the needed size in out_off
    import ctypes as C
    ta, roa, bra = dp.flatten_blocks(A)
    tb, rob, brb = dp.flatten_blocks(B)
    n = len(A)
    sc, cols, out, off = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(16, np.uint8), np.zeros(n + 1, np.int64)
    rc = dp._lib().pm_dp_align_blocks(ta.ctypes.data, roa.ctypes.data, len(roa) - 1, bra.ctypes.data, tb.ctypes.data, rob.ctypes.data,
                                       len(rob) - 1, brb.ctypes.data, n, C.byref(params), sc.ctypes.data, cols.ctypes.data, out.ctypes.data, 16,
                                       off.ctypes.data, 0)
    assert rc != 0 and off[-1] == sum(len(m) * len(m[0]) for m in merged) and np.array_equal(sc, scores)


@pytest.mark.gpu
def test_emit_refuses_a_path_that_does_not_span_the_pair():
    from paramugsy_amd import capi
    with pytest.raises(capi.PmError) as e:
        dp.emit_maf([[b"ACG"]], [[b"CGT"]], [np.array([0, 0], dtype=np.uint8)])
    assert e.value.code == capi.PM_E_INVALID
    with pytest.raises(capi.PmError):
        dp.pack_maf([[b"ACG", b"AC"]])  # rows of one block differ in length


@pytest.mark.gpu
def test_align_maf_files_end_to_end(oracle_build, tmp_path):
    """Two MAF files -> a MAF file of merged blocks: `a score=` is the oracle's score for the packed blocks, every `s` line keeps
    its six leading fields, and the texts are the oracle's expansion of the oracle's path."""
    import pyoracle
    rng = np.random.default_rng(5)
    A = random_blocks(rng, 6, max_rows=3, max_cols=120)
    B = random_blocks(rng, 6, max_rows=3, max_cols=120)

    def write(path, blocks, tag):
        with open(path, "wb") as f:
            f.write(b"##maf version=1 scoring=test\n# a comment\n")
            for k, b in enumerate(blocks):
                f.write(b"a score=0 label=%d\n" % k)
                for r, row in enumerate(b):
                    size = sum(ch not in b"-" for ch in row)
                    f.write(b"s %s.g%d %d %d + 100000 %s\n" % (tag, r, 10 * k, size, row))
                f.write(b"\n")
    pa, pb, po = str(tmp_path / "a.maf"), str(tmp_path / "b.maf"), str(tmp_path / "out.maf")
    write(pa, A, b"L")
    write(pb, B, b"R")
    params = dp.make_params(2, 2)
    dp.align_maf_files(pa, pb, params, po)
    got = ora.parse_maf(po)
    assert len(got) == 6
    lines = open(po, "rb").read().split(b"\n")
    score_lines = [ln for ln in lines if ln.startswith(b"a score=")]
    for k in range(6):
        ca = np.array(ora.pack_block(A[k]), dtype=np.uint8)
        cb = np.array(ora.pack_block(B[k]), dtype=np.uint8)
        one = dp.DpInputs(ca, np.array([0, len(ca)], dtype=np.int64), cb, np.array([0, len(cb)], dtype=np.int64))
        s, p = pyoracle.dp_align(one, params)
        assert score_lines[k] == b"a score=%d" % s[0]
        heads, texts = got[k]
        assert texts == ora.emit_block(A[k], B[k], p[0].tolist())
        assert [h.split()[1] for h in heads] == [b"L.g%d" % r for r in range(len(A[k]))] + [b"R.g%d" % r for r in range(len(B[k]))]
    # mismatching block counts are refused
    from paramugsy_amd import capi
    write(pb, B[:5], b"R")
    with pytest.raises(capi.PmError):
        dp.align_maf_files(pa, pb, params, po)
    # two files without blocks: the header alone
    write(pa, [], b"L")
    write(pb, [], b"R")
    dp.align_maf_files(pa, pb, params, po)
    assert open(po, "rb").read() == b"##maf version=1 scoring=paramugsy_amd\n"
    # a block without rows on one side (an `a` line followed by nothing) aligns as an empty profile
    with open(pa, "wb") as f:
        f.write(b"a score=0\n\na score=0\ns L.g0 0 4 + 100 ACGT\n\n")
    with open(pb, "wb") as f:
        f.write(b"a score=0\ns R.g0 0 3 + 100 ACG\n\na score=0\ns R.g0 5 4 + 100 ACGT\n\n")
    dp.align_maf_files(pa, pb, params, po)
    got = ora.parse_maf(po)
    assert len(got) == 2 and got[0][1] == [b"ACG"] and got[1][1] == [b"ACGT", b"ACGT"]


@pytest.mark.gpu
def test_a_failed_call_leaves_an_existing_output_alone(tmp_path):
    """pm_dp_align_maf empties its output file only when it is about to write: a missing input, unequal block counts or a block of
    more than 255 rows leave a file that is already there as it was, and an output that is one of the (mapped) inputs is refused."""
    from paramugsy_amd import capi
    pa, pb, po = str(tmp_path / "a.maf"), str(tmp_path / "b.maf"), str(tmp_path / "out.maf")
    keep = b"precious bytes\n" * 100
    params = dp.make_params(1, 1)

    def two(path, n):
        with open(path, "wb") as f:
            for k in range(n):
                f.write(b"a score=0\ns g.%d 0 4 + 100 ACGT\n\n" % k)
    for make_inputs in (lambda: (two(pa, 2), os.remove(pb) if os.path.exists(pb) else None),  # an input is missing
                        lambda: (two(pa, 2), two(pb, 3)),                                      # unequal block counts
                        lambda: (two(pb, 1), open(pa, "wb").write(b"a score=0\n" + b"".join(b"s r%d 0 1 + 9 A\n" % r for r in range(300))))):
        make_inputs()
        with open(po, "wb") as f:
            f.write(keep)
        with pytest.raises(capi.PmError):
            dp.align_maf_files(pa, pb, params, po)
        assert open(po, "rb").read() == keep
    two(pa, 2)
    two(pb, 2)
    before = open(pa, "rb").read()
    with pytest.raises(capi.PmError, match="one of the input files"):
        dp.align_maf_files(pa, pb, params, pa)
    assert open(pa, "rb").read() == before
    with pytest.raises(capi.PmError, match="one of the input files"):
        dp.align_maf_files(pa, pb, params, pb, devices=[0, 0])
    # and a good call over a longer old file leaves no tail of it behind
    with open(po, "wb") as f:
        f.write(keep * 50)
    dp.align_maf_files(pa, pb, params, po)
    out = open(po, "rb").read()
    assert out.startswith(b"##maf") and b"precious" not in out and len(ora.parse_maf(po)) == 2


@pytest.mark.gpu
def test_score_less_a_lines_open_blocks(oracle_build, tmp_path):
    """The score after `a` is optional in MAF: a line that is just `a` (also with a trailing CR) starts a block, so the rows
    that follow are not appended to the block before it and block k of A still meets block k of B."""
    pa, pb, po = str(tmp_path / "a.maf"), str(tmp_path / "b.maf"), str(tmp_path / "out.maf")
    with open(pa, "wb") as f:
        f.write(b"##maf version=1\na\ns L.g0 0 4 + 100 ACGT\n\na\r\ns L.g0 9 3 + 100 TTG\r\n\n")
    with open(pb, "wb") as f:
        f.write(b"a score=1\ns R.g0 0 4 + 100 ACGT\n\na\ns R.g0 7 3 + 100 TTG\n\n")
    dp.align_maf_files(pa, pb, dp.make_params(1, 1), po)
    got = ora.parse_maf(po)
    assert len(got) == 2
    assert got[0][1] == [b"ACGT", b"ACGT"] and got[1][1] == [b"TTG", b"TTG"]


@pytest.mark.gpu
def test_large_maf_files_parsed_in_ranges_and_emitted_on_the_device(oracle_build, tmp_path):
    """Files big enough (> 4 MB) for the parser to cut them into ranges parsed side by side, with comment lines, CR line ends and
    blocks of different depths; the output file is the in-memory entry's merged blocks (pm_dp_align_blocks, checked against the
    oracle elsewhere) under `a score=` lines, every `s` line keeping its six leading fields, byte for byte."""
    rng = np.random.default_rng(77)
    n = 800
    A = random_blocks(rng, n, max_rows=6, max_cols=3500)
    B = random_blocks(rng, n, max_rows=6, max_cols=3500)
    A[7], B[11], A[n - 1], B[n - 1] = [], [], [b"ACGT" * 3], []

    def write(path, blocks, tag, cr):
        eol = b"\r\n" if cr else b"\n"
        with open(path, "wb") as f:
            f.write(b"##maf version=1" + eol + b"# comment" + eol)
            for k, b in enumerate(blocks):
                f.write((b"a" if k % 5 == 0 else b"a score=%d" % k) + eol)
                for r, row in enumerate(b):
                    f.write(b"s %s.g%d\t%d %d + 100000 %s" % (tag, r, 10 * k, sum(ch not in b"-" for ch in row), row) + eol)
                f.write(eol + (b"# between blocks" + eol if k % 7 == 0 else b""))
    pa, pb, po = str(tmp_path / "a.maf"), str(tmp_path / "b.maf"), str(tmp_path / "out.maf")
    write(pa, A, b"L", False)
    write(pb, B, b"R", True)
    assert os.path.getsize(pa) > (4 << 20) and os.path.getsize(pb) > (4 << 20)
    params = dp.make_params(3, 3)
    dp.align_maf_files(pa, pb, params, po)
    scores, merged = dp.align_blocks(A, B, params)
    want = [b"##maf version=1 scoring=paramugsy_amd\n"]
    for k in range(n):
        want.append(b"a score=%d\n" % scores[k])
        heads = [b"s L.g%d\t%d %d + 100000" % (r, 10 * k, sum(ch not in b"-" for ch in row)) for r, row in enumerate(A[k])]
        heads += [b"s R.g%d\t%d %d + 100000" % (r, 10 * k, sum(ch not in b"-" for ch in row)) for r, row in enumerate(B[k])]
        for h, t in zip(heads, merged[k]):
            want.append(h + b" " + t + b"\n")
        want.append(b"\n")
    assert open(po, "rb").read() == b"".join(want)
    many = str(tmp_path / "many.maf")
    dp.align_maf_files(pa, pb, params, many, devices=[0, 0, 0])
    assert open(many, "rb").read() == open(po, "rb").read()
