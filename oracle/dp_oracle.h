/*
 * oracle/dp_oracle.h -- CPU ORACLE of the profile x profile DP.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * PARITY STATUS: "parity unpinned" -- NO REFERENCE COUNTERPART.  BASELINE.json's north_star asks for a
 * profile-vs-profile affine-gap DP "bit-exact vs lib/profiles_cpp", but the reference contains no DP, no
 * scores and no traceback anywhere.  The recurrence, the column encoding, the scoring
 * and the tie-breaking below are therefore SPECIFIED BY THIS REPO (DESIGN.md "Profile DP specification");
 * the GPU kernel is checked against this scalar restatement of that specification and nothing else.
 *
 * Specification (all arithmetic int32):
 *   column  = 8 bytes {nA, nC, nG, nT, nGap, 0, 0, 0}: how many of the profile's rows hold that symbol
 *   s(i,j)  = sum_{a,b} colA[i][a] * colB[j][b] * sub[a*5+b]          (sum of pairs, 5x5 matrix)
 *   E[i][j] = max(E[i][j-1] - gap_extend, H[i][j-1] - gap_open)       horizontal: consumes a column of B
 *   F[i][j] = max(F[i-1][j] - gap_extend, H[i-1][j] - gap_open)       vertical:   consumes a column of A
 *   H[i][j] = max(H[i-1][j-1] + s(i,j), E[i][j], F[i][j])
 *   H[0][0] = 0, H[i][0] = F[i][0] = -(gap_open + (i-1)*gap_extend), H[0][j] = E[0][j] likewise,
 *   E[i][0] = F[0][j] = DP_NEG_INF.  Global alignment: score = H[La][Lb].
 *   ties:   H prefers diagonal, then E, then F;  E and F prefer opening (from H) over extending.
 *   path    ops from (0,0) to (La,Lb): 0 = M (one column of each), 1 = I (column of B against a gap),
 *           2 = D (column of A against a gap).
 *
 * The same WITHOUT matrices (round 5; tests/dp_bruteforce.py writes every alignment of tiny pairs out and holds this file to it):
 *   score   = max over all op strings of  sum of s(i,j) over the M's  -  sum over MAXIMAL runs of I's and of D's of
 *             (gap_open + (len - 1) * gap_extend);  an I run directly followed by a D run is two runs, two openings.
 *   path    = label every gap column as opening its run or extending it; read the optimal labelled strings from the END; the one
 *             reported is the smallest in lexicographic order under  M < I-open < I-extend < D-open < D-extend.
 *   Two corners, stated so that nobody has to find them again:
 *   - the boundary rows H[0][j], H[i][0] are ONE run from the origin (labels forced: open, extend, extend, ...);
 *   - gap_open < gap_extend (allowed by the limits, used by no scoring scheme): inside the matrix a run may re-open instead of
 *     extending (E takes max(E - ge, H - go) and H may itself have come from E), so a run of n costs go + (n-1) * min(go, ge)
 *     there, while the boundary keeps go + (n-1) * ge.  For gap_open >= gap_extend the maximum is the textbook affine optimum.
 */
#ifndef DP_ORACLE_H
#define DP_ORACLE_H

#include <stdint.h>

#define DP_NEG_INF (-(1 << 29))
#define DP_OP_M 0
#define DP_OP_I 1
#define DP_OP_D 2

typedef struct dp_params {
  int32_t sub[25]; /* sub[a*5+b], symbols A,C,G,T,gap */
  int32_t gap_open;
  int32_t gap_extend;
} dp_params_t;

/* Full-matrix alignment.  ops must hold La+Lb bytes.  Returns 0, or -1 when out of memory. */
int dp_oracle_align(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p,
                    int32_t *score, uint8_t *ops, int32_t *n_ops);
/* Score only, two rows of memory (for large cases and for timing the CPU baseline). */
int32_t dp_oracle_score(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p);
/* Re-scores a path under the specification (size-independent check: a reported path must reproduce the
 * reported score).  Returns 0 and the score, or -1 when the path does not span (0,0)-(La,Lb). */
int dp_oracle_score_of_path(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p,
                            const uint8_t *ops, int32_t n_ops, int32_t *score);
/* the same for pairs [first, end) of a batch in the product's fetch layout (path k = the last n_ops[k] bytes of its slot) */
void dp_oracle_score_of_paths(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t first, int64_t end,
                              const dp_params_t *p, const uint8_t *ops, const int32_t *n_ops, int32_t *rc, int32_t *scores);
/* A batch of pairs in the product's layout (cols_*: concatenated columns, off_*: n_pairs+1 column offsets).
 * ops_off[k] .. : each pair owns la+lb bytes at ops + (off_a[k] + off_b[k]). */
int dp_oracle_align_batch(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                          const dp_params_t *p, int32_t *scores, uint8_t *ops, int32_t *n_ops);
void dp_oracle_score_batch(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                           const dp_params_t *p, int32_t *scores);

#endif
