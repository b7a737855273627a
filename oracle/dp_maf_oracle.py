"""CPU ORACLE of the two byte transforms around the profile DP (csrc/dp_maf.hip).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
"Parity unpinned": the reference has no DP stage, so nothing upstream prints these bytes; the transforms restate
what the reference does around its own path -- a per-column fold over a block's rows (lib/profiles/m_make.ml:15-45) and the
expansion of a row's text along a walk, '-' where the walk skips it (lib/profiles/m_untranslate.ml:38-52) -- and are pinned
by the hand-worked cases in tests/test_dp_maf.py.  Plain Python loops, one symbol at a time."""
from typing import List, Sequence


def pack_block(rows: Sequence[bytes]) -> List[List[int]]:
    """One packed column {nA, nC, nG, nT, nGap, nOther, 0, 0} per block column; case-insensitive."""
    n = len(rows[0]) if rows else 0
    out = []
    for c in range(n):
        col = [0] * 8
        for r in rows:
            ch = chr(r[c]).upper()
            k = "ACGT-".find(ch)
            col[k if k >= 0 else 5] += 1
        out.append(col)
    return out


def emit_block(rows_a: Sequence[bytes], rows_b: Sequence[bytes], ops: Sequence[int]) -> List[bytes]:
    """The merged block: A's rows then B's rows, expanded along the path (0 = M, 1 = I: B only, 2 = D: A only)."""
    out = [bytearray() for _ in range(len(rows_a) + len(rows_b))]
    i = j = 0
    for op in ops:
        for r, row in enumerate(rows_a):
            out[r].append(row[i] if op != 1 else ord("-"))
        for r, row in enumerate(rows_b):
            out[len(rows_a) + r].append(row[j] if op != 2 else ord("-"))
        i += op != 1
        j += op != 2
    assert (not rows_a or i == len(rows_a[0])) and (not rows_b or j == len(rows_b[0])), "the path does not span the pair"
    return [bytes(x) for x in out]


def parse_maf(path: str):
    """[(list of `s` line heads, list of texts)] per block."""
    blocks = []
    for line in open(path, "rb").read().split(b"\n"):
        if line[:2] in (b"a ", b"a\t"):
            blocks.append(([], []))
        elif line[:2] in (b"s ", b"s\t"):
            f = line.split()
            blocks[-1][0].append(b" ".join(f[:6]))
            blocks[-1][1].append(f[6])
    return blocks
