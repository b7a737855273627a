// dp_internal.hpp -- layouts shared by the DP fill kernel (dp_kernels.hip) and the checkpoint walk (dp_walk.hip).
//
// NO REFERENCE COUNTERPART: the profile x profile DP is specified by this repo (oracle/dp_oracle.h).
//
// Two ways to get the path of a pair:
//   bits        the fill kernel shifts 4 decision bits per cell into words (0.5 byte per cell) and a walk reads them.
//               14 VALU instructions per cell, 8 of them the decision bits.
//   checkpoints the fill kernel computes scores only (6 VALU instructions per cell) and keeps
//                 * per row and per lane (= per C columns) the two values the row hands to the next lane:
//                   {H~ - gop, E~} of the lane's last column, stored step by step, one coalesced 512-byte store per step;
//                 * every DP_CK_R steps every lane's column state {H~ - gop, F~} of its C columns;
//               the walk then re-runs the recurrence WITH decision bits only inside the C-column x DP_CK_R-row blocks the
//               path passes through (at most La / DP_CK_R + Lb / C + 1 of them), a few lanes per pair.
#pragma once

#include <hip/hip_runtime.h>

namespace pm {

typedef long long i64;
typedef unsigned long long u64;

#define DP_NEG_INF (-(1 << 29))

struct DpParamsD {
  int sub[25];
  int go;
  int ge;
};

// Block geometry of the checkpoints (compile-time; -DDP_CK_R=.. -DDP_CK_W=.. for experiments):
//   DP_CK_W  lanes of the fill kernel per column group: only the last lane of a group stores its column checkpoints, so a
//            block of the walk is DP_CK_W * C columns wide;
//   DP_CK_R  rows per block (a power of two).  The lanes of a group take their row checkpoints one step apart, after the
//            SAME row of A, so the blocks are rectangles: lane l stores after the step t with (t + 1 - l % DP_CK_W) a multiple
//            of DP_CK_R.
#ifndef DP_CK_R
#define DP_CK_R 32
#endif
#ifndef DP_CK_W
#define DP_CK_W 2
#endif
static_assert((DP_CK_R & (DP_CK_R - 1)) == 0 && DP_CK_R >= 16, "DP_CK_R is a power of two");
static_assert(DP_CK_W == 1 || DP_CK_W == 2 || DP_CK_W == 4, "DP_CK_W in {1, 2, 4}");

// Words (4 bytes) of checkpoint storage one pair needs.  Per stripe (64 lanes x C columns of B):
//   col[t][group]     int2 {H~ - gop, E~} of the group's last column after the row its last lane was on at step t
//   row[m][lane][c]   int2 {H~ - gop, F~} of column c after the lane's (m + 1)-th checkpoint step
__host__ __device__ inline i64 dp_ck_steps(i64 la) { return la + 63; }
__host__ __device__ inline i64 dp_ck_nck(i64 la) { return (la + 63) / DP_CK_R; }
__host__ __device__ inline i64 dp_ck_stripes(i64 lb, int C) { return (lb + 64 * C - 1) / (64 * C); }
__host__ __device__ inline i64 dp_ck_col_words_per_step() { return (64 / DP_CK_W) * 2; }
__host__ __device__ inline i64 dp_ck_words(i64 la, i64 lb, int C) {
  return dp_ck_stripes(lb, C) * (dp_ck_steps(la) * dp_ck_col_words_per_step() + dp_ck_nck(la) * 64 * 2 * C);
}
// column checkpoint of fill lane `lane` (the last of its group) at step t of stripe s
__host__ __device__ inline i64 dp_ck_col_word(i64 la, i64 s, i64 t, int lane) {
  return (s * dp_ck_steps(la) + t) * dp_ck_col_words_per_step() + (lane / DP_CK_W) * 2;
}
__host__ __device__ inline i64 dp_ck_row_word(i64 la, i64 lb, int C, i64 s, i64 m, int lane) {
  return dp_ck_stripes(lb, C) * dp_ck_steps(la) * dp_ck_col_words_per_step() + ((s * dp_ck_nck(la) + m) * 64 + lane) * 2 * C;
}
// bytes the fill kernel writes for one pair in checkpoint mode (rows of A it is on, not steps)
__host__ __device__ inline i64 dp_ck_bytes_written(i64 la, i64 lb, int C) {
  return dp_ck_stripes(lb, C) * (la * dp_ck_col_words_per_step() + dp_ck_nck(la) * 64 * 2 * C) * 4;
}

// dp_walk.hip: one launch walks the paths of pairs [first_pair, first_pair + n) from their checkpoints.
// cols_per_lane (of the fill kernel) in {8, 16}; lanes_per_pair a power of two for which dp_walk_lanes_ok() holds.
int dp_launch_walk(int cols_per_lane, int lanes_per_pair, const u64 *cols_a, const i64 *off_a, const u64 *cols_b, const i64 *off_b,
                   i64 first_pair, i64 n, const i64 *tb_off, const unsigned *ck, unsigned char *ops, int *n_ops, const DpParamsD &P,
                   hipStream_t stream);

bool dp_walk_lanes_ok(int cols_per_lane, int lanes_per_pair);

} // namespace pm
