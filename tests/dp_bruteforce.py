"""The DP's specification held against a DIFFERENT ALGORITHM: no recurrence, no matrices -- every alignment of a tiny pair is
written out, scored from the definition and compared (until round 5 the specification of oracle/dp_oracle.h was
checked only against restatements of the same three-matrix recurrence, by one author reading one text twice).

"Parity unpinned" stays: the reference has no DP (lib/maf/alignment.ml:8-10: a score is a string).  What this pins is
that the recurrence, its boundary rows and its tie order say what the prose definition below says.

THE DEFINITION (what a reader without the recurrence would write down).
  An alignment of profiles A (La columns) and B (Lb columns) is a string of ops that consumes A and B from (0, 0) to (La, Lb):
  M takes a column of each, I a column of B against a gap, D a column of A against a gap.
  Score: the sum-of-pairs score s(i, j) = sum_ab A[i][a] * B[j][b] * sub[a][b] of every M, minus the gap costs:
      every MAXIMAL run of I's, and every maximal run of D's, of length n costs  go + (n - 1) * ge      (go >= ge: the affine model)
  An I run directly followed by a D run (or the reverse) is two runs and pays two openings.
  The DP reports the maximum over all alignments, and ONE alignment that attains it.

GAP COLUMNS WITH LABELS (what makes the score additive, and the choice among optima sayable).
  Write every gap column as opening (Io, Do: cost go) or extending (Ie, De: cost ge; allowed only directly after a gap column of the
  same kind).  A run labelled o e e ... e is the definition's price; the labelled strings also contain runs that re-open (o e o e):
  for go >= ge those never beat the plain labelling, so the maximum over labelled strings IS the definition's maximum (asserted below
  for every go >= ge case, against a scorer that knows nothing of labels).
  Which optimum is reported: read the labelled strings BACKWARDS (from (La, Lb)); the reported one is the smallest optimal string in
  lexicographic order under   M < Io < Ie < Do < De   -- i.e. at the end prefer a match to an insertion to a deletion, and a gap
  column that opens its run to one that extends it -- which is oracle/dp_oracle.h's "H prefers diagonal, then E, then F; E and F
  prefer opening over extending", said without H, E and F.

TWO CORNERS OF THE SPECIFICATION THIS FOUND AND NOW STATES (oracle/dp_oracle.h carries them since round 5):
  * the boundary.  H[0][j] = -(go + (j-1) ge) and H[i][0] likewise are ONE run from the origin: along row 0 / column 0 the labels
    are forced (o e e ...).  For go >= ge that is what the free maximum would choose anyway.
  * go < ge (an extension dearer than an opening; no biological scoring has it, the limits allow it).  Inside the matrix a run may
    then re-open instead of extending, so a run of n costs go + (n - 1) * min(go, ge) there -- but not along the boundary, which
    keeps go + (n - 1) * ge.  The labelled enumeration below follows exactly that (re-opening allowed off the boundary only), so
    these cases are pinned too; the plain-definition cross-check is made for go >= ge only, where the definition is the textbook's.
"""
import itertools

M, IO, IE, DO, DE = 0, 1, 2, 3, 4  # the order of preference, read from the END of the alignment
OP_OF = {M: 0, IO: 1, IE: 1, DO: 2, DE: 2}


def column_score(ca, cb, sub):
    return sum(int(ca[a]) * int(cb[b]) * int(sub[a * 5 + b]) for a in range(5) for b in range(5))


def labelled_alignments(la, lb):
    """Every labelled op string from (0, 0) to (la, lb).  Xe only directly after Xo / Xe; on the boundary (i == 0 or j == 0 BEFORE the
    column is taken... i.e. while the path has not left row 0 / column 0) the run from the origin is one run: o, then e's."""
    out = []

    def go_on(i, j, prefix):
        if i == la and j == lb:
            out.append(tuple(prefix))
            return
        last = prefix[-1] if prefix else None
        if i < la and j < lb:
            go_on(i + 1, j + 1, prefix + [M])
        if j < lb:
            on_boundary = i == 0  # the cell reached, (0, j + 1), lies on row 0: H[0][j+1] is the one run from the origin
            if on_boundary:
                go_on(i, j + 1, prefix + [IO if j == 0 else IE])
            else:
                go_on(i, j + 1, prefix + [IO])
                if last in (IO, IE):
                    go_on(i, j + 1, prefix + [IE])
        if i < la:
            on_boundary = j == 0
            if on_boundary:
                go_on(i + 1, j, prefix + [DO if i == 0 else DE])
            else:
                go_on(i + 1, j, prefix + [DO])
                if last in (DO, DE):
                    go_on(i + 1, j, prefix + [DE])
    go_on(0, 0, [])
    return out


def score_labelled(seq, a, b, sub, go, ge):
    i = j = total = 0
    for x in seq:
        if x == M:
            total += column_score(a[i], b[j], sub)
            i, j = i + 1, j + 1
        elif x in (IO, IE):
            total -= go if x == IO else ge
            j += 1
        else:
            total -= go if x == DO else ge
            i += 1
    return total


def plain_alignments(la, lb):
    """Every UNLABELLED op string (0 = M, 1 = I, 2 = D) from (0, 0) to (la, lb): Delannoy(la, lb) of them."""
    out = []

    def go_on(i, j, prefix):
        if i == la and j == lb:
            out.append(tuple(prefix))
            return
        if i < la and j < lb:
            go_on(i + 1, j + 1, prefix + [0])
        if j < lb:
            go_on(i, j + 1, prefix + [1])
        if i < la:
            go_on(i + 1, j, prefix + [2])
    go_on(0, 0, [])
    return out


def score_plain(ops, a, b, sub, go, ge):
    """The definition as the prose has it: every M's sum of pairs, every MAXIMAL run of I's or of D's costs go + (len - 1) * ge."""
    i = j = total = 0
    for op, run in itertools.groupby(ops):
        n = len(list(run))
        if op == 0:
            for _ in range(n):
                total += column_score(a[i], b[j], sub)
                i, j = i + 1, j + 1
        elif op == 1:
            total -= go + (n - 1) * ge
            j += n
        else:
            total -= go + (n - 1) * ge
            i += n
    return total


def best_by_enumeration(a, b, sub, go, ge):
    """(score, ops of the preferred optimum, number of optimal UNLABELLED alignments) by writing every alignment out."""
    la, lb = len(a), len(b)
    seqs = labelled_alignments(la, lb)
    scored = [(score_labelled(s, a, b, sub, go, ge), s) for s in seqs]
    best = max(v for v, _ in scored)
    optimal = [s for v, s in scored if v == best]
    preferred = min(optimal, key=lambda s: tuple(reversed(s)))
    ops = [OP_OF[x] for x in preferred]
    if go >= ge:
        # the labels are bookkeeping: the textbook definition, scored without them, has the same maximum, and the reported alignment
        # attains it
        plain = [(score_plain(p, a, b, sub, go, ge), p) for p in plain_alignments(la, lb)]
        assert max(v for v, _ in plain) == best
        assert score_plain(tuple(ops), a, b, sub, go, ge) == best
        n_opt = sum(1 for v, _ in plain if v == best)
    else:
        n_opt = len({tuple(OP_OF[x] for x in s) for s in optimal})
    return best, ops, n_opt


def random_case(rng, max_len=5, max_rows=3, tie_heavy=True):
    """One tiny pair: 1..max_rows rows per side, lengths 0..max_len, a random 5 x 5 matrix, go != ge mostly (zero penalties and
    go < ge included).  tie_heavy: small matrix entries and small penalties, so that most cases have several optima."""
    import numpy as np
    ra, rb = int(rng.integers(1, max_rows + 1)), int(rng.integers(1, max_rows + 1))
    la, lb = int(rng.integers(0, max_len + 1)), int(rng.integers(0, max_len + 1))

    def cols(n, rows):
        c = np.zeros((n, 8), dtype=np.uint8)
        for k in range(n):
            for _ in range(rows):
                c[k, int(rng.integers(0, 5))] += 1
        return c
    a, b = cols(la, ra), cols(lb, rb)
    lim = 2 if tie_heavy else 9
    sub = [int(rng.integers(-lim, lim + 1)) for _ in range(25)]
    kind = int(rng.integers(0, 6))
    if kind == 0:
        go, ge = 0, 0
    elif kind == 1:
        go, ge = int(rng.integers(1, 6)), 0
    elif kind == 2:
        ge = int(rng.integers(1, 4))
        go = ge  # linear gaps
    elif kind == 3:
        go = int(rng.integers(0, 3))
        ge = go + int(rng.integers(1, 4))  # an extension dearer than an opening: the corner the docstring describes
    else:
        ge = int(rng.integers(0, 4))
        go = ge + int(rng.integers(1, 8))
    scale = ra * rb if rng.random() < 0.5 else 1
    return a, b, sub, go * scale, ge * scale
