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
  int rows_a; // the number of symbols every column of A holds when that is one number for the whole batch (UNI kernels), else 0
};

// Block geometry of the checkpoints (compile-time; -DDP_CK_R=.. -DDP_CK_W=.. for experiments):
//   DP_CK_W  lanes of the fill kernel per column group: only the last lane of a group stores its column checkpoints, so a
//            block of the walk is DP_CK_W * C columns wide;
//   DP_CK_R  rows per block (a power of two).  The lanes of a group take their row checkpoints one step apart, after the
//            SAME row of A, so the blocks are rectangles: lane l stores after the step t with (t + 1 - l % DP_CK_W) a multiple
//            of DP_CK_R.
#ifndef DP_CK_R
#define DP_CK_R 64
#endif
#ifndef DP_CK_W
#define DP_CK_W 4
#endif
static_assert((DP_CK_R & (DP_CK_R - 1)) == 0 && DP_CK_R >= 16, "DP_CK_R is a power of two");
static_assert(DP_CK_W == 1 || DP_CK_W == 2 || DP_CK_W == 4, "DP_CK_W in {1, 2, 4}");

// Stripes.  The columns of B are cut into stripes of 64 lanes x cs columns; a stripe's lane keeps its cs columns in registers
// while the wavefront sweeps the rows of A.  Full stripes have cs = C columns per lane (C = 16, or 8 for small batches).  A stripe
// costs (La + 63) steps whatever is left of B, so a ragged batch computed 1.23 x its cells in padding (round 3).  With `tail`
// (C = 16 only) what is left after the full stripes is therefore cut into NARROWER stripes, in quarters of a full one: up to 256
// columns left -> one stripe of 4 columns per lane; up to 512 -> one of 8; up to 768 -> one of 8 and one of 4; more -> a full one.
// A step of a narrow stripe costs (6 cs + 5) / 101 of a full stripe's.  Everything below -- the checkpoints' layout, the walk's
// blocks -- is addressed by COLUMN (or by column group: DP_CK_W * C columns, the width of a block of the walk), so that it is the
// same whatever the widths of the stripes the columns were computed in.
struct DpStripe {
  int jb; // first column of B (0-based)
  int cs; // columns per lane
};
// (cs is 16, 8 or 4: shifts, not the division a variable divisor compiles to)
__host__ __device__ inline int dp_cs_shift(int cs) { return cs == 16 ? 4 : (cs == 8 ? 3 : 2); }
__host__ __device__ inline int dp_tail_quarters(i64 lb, int C, int tail) { // quarters of a full stripe the remainder needs; 0: none
  const i64 rem = lb % (64 * C);
  return !tail || C != 16 || rem == 0 ? 0 : (int)((rem + 16 * C - 1) / (16 * C));
}
__host__ __device__ inline i64 dp_ck_stripes(i64 lb, int C, int tail) {
  const int q = dp_tail_quarters(lb, C, tail);
  return tail && C == 16 ? lb / (64 * C) + (q == 0 ? 0 : (q == 3 ? 2 : 1)) : (lb + 64 * C - 1) / (64 * C);
}
__host__ __device__ inline DpStripe dp_stripe(i64 lb, int C, int tail, i64 s) {
  const i64 full = lb / (64 * C);
  DpStripe st = {(int)(s * 64 * C), C};
  if(tail && C == 16 && s >= full) {
    const int q = dp_tail_quarters(lb, C, tail);
    if(q == 1) {
      st.cs = 4;
    }
    else if(q == 2) {
      st.cs = 8;
    }
    else if(q == 3) {
      st.jb = (int)(full * 64 * C) + (s > full ? 32 * C : 0);
      st.cs = s > full ? 4 : 8;
    }
  }
  return st;
}
__host__ __device__ inline DpStripe dp_stripe_of_col(i64 lb, int C, int tail, i64 j) {
  const i64 full = lb / (64 * C);
  i64 s = j / (64 * C);
  if(tail && C == 16 && s >= full && dp_tail_quarters(lb, C, tail) == 3 && j - full * 64 * C >= 32 * C) {
    s = full + 1;
  }
  return dp_stripe(lb, C, tail, s);
}
// the columns the stripes cover (the cells computed are this x the rows of A)
__host__ __device__ inline i64 dp_padded_cols(i64 lb, int C, int tail) {
  const int q = dp_tail_quarters(lb, C, tail);
  return tail && C == 16 ? (lb / (64 * C)) * 64 * C + q * 16 * C : ((lb + 64 * C - 1) / (64 * C)) * 64 * C;
}
// what a pair costs the wavefront that works through its stripes, in VALU instructions (6 per cell + about 5 per step): the
// measure of dp_batch_plan's processing order and of its tiers
__host__ __device__ inline i64 dp_fill_cost(i64 la, i64 lb, int C, int tail) {
  i64 per_step = 0;
  const i64 n = dp_ck_stripes(lb, C, tail);
  for(i64 s = 0; s < n; ++s) {
    per_step += 6 * dp_stripe(lb, C, tail, s).cs + 5;
  }
  return per_step * (la + 63);
}

// Words (4 bytes) of checkpoint storage one pair needs:
//   col[group][t]     int2 {H~ - gop, E~} of the group's last column after the row its last lane was on at step t
//   row[stripe][m][j] int2 {H~ - gop, F~} of column j of the stripe after its lane's (m + 1)-th checkpoint step
// Round 5: every store of a checkpoint is a whole, ALIGNED 128-byte line.  A pair's workspace is a multiple of 128 bytes (so every
// pair's begins on a line: the chunks' bases are multiples of 256), a group's column checkpoints are dp_ck_stride(la) entries apart
// -- the steps rounded up to 16, the entries of one 128-byte store -- and the row checkpoints begin on a line and give every lane of
// the fill kernel a line-aligned run of its own (below).  Until then a 128-byte store of column checkpoints straddled two lines
// ((la + 63) * 8 is a multiple of 128 for no length anybody uses), and the 64-byte write requests HBM saw were 1.22 x the bytes.
__host__ __device__ inline i64 dp_ck_steps(i64 la) { return la + 63; }
__host__ __device__ inline i64 dp_ck_stride(i64 la) { return (la + 63 + 15) & ~(i64)15; }
__host__ __device__ inline i64 dp_ck_nck(i64 la) { return (la + 63) / DP_CK_R; }
__host__ __device__ inline i64 dp_ck_groups(i64 lb, int C, int tail) { return dp_padded_cols(lb, C, tail) / (DP_CK_W * C); }
__host__ __device__ inline i64 dp_ck_words(i64 la, i64 lb, int C, int tail) {
  return (dp_ck_groups(lb, C, tail) * dp_ck_stride(la) * 2 + dp_ck_nck(la) * dp_padded_cols(lb, C, tail) * 2 + 31) & ~(i64)31;
}
// column checkpoint of column group g (written by the last lane of the group) at step t.  Layout [group][step]: the steps of one
// group are contiguous, which is how all three readers go through them (the next stripe's seam: lane 63's group, 64 rows at a
// time; the walk's left edge: one group, the rows of a block); the writer's 16 lanes hit 16 lines per step, each line completed
// by 16 consecutive steps while it sits in L2.  ([step][group] made the seam read 8 bytes of every 128-byte line: 20 GB of
// FETCH_SIZE on the headline batch.)
__host__ __device__ inline i64 dp_ck_col_word(i64 la, i64 g, i64 t) { return (g * dp_ck_stride(la) + t) * 2; }
// row checkpoint m of column j, which lies in stripe st: [stripe][m][lane][c] with c = the column's place in its lane -- a lane's cs
// columns are cs * 8 consecutive bytes: one line with 16 columns a lane, written by that lane alone in ONE step (16-byte stores
// back to back), so the line is whole when it leaves L2, and the walk's top edge is the consecutive lines of the block's lanes.
// History: round 3 had this order with 8-byte stores; round 4 turned it lane-minor ([c][lane]: a store of one c then covers 64
// consecutive slots -- but only every fourth lane stores at a step, the lanes of a group being one step apart, so every line was
// filled in four steps a few microseconds apart and left L2 in pieces: WRITE_SIZE 1.22 x the checkpoints, profiles/r04_store_ablation.txt).
__host__ __device__ inline i64 dp_ck_row_word(i64 la, i64 lb, int C, int tail, DpStripe st, i64 m, i64 j) {
  const int r = (int)(j - st.jb);
  return dp_ck_groups(lb, C, tail) * dp_ck_stride(la) * 2 + ((i64)st.jb * dp_ck_nck(la) + m * 64 * st.cs + r) * 2;
}
// the first fill lane of column group g (in the stripe it lies in), and the lanes of a group there
__host__ __device__ inline int dp_group_lane0(int C, DpStripe st, i64 g) { return ((int)g * DP_CK_W * C - st.jb) >> dp_cs_shift(st.cs); }
__host__ __device__ inline int dp_group_lanes(int C, DpStripe st) { return (DP_CK_W * C) >> dp_cs_shift(st.cs); }
// bytes the fill kernel writes for one pair in checkpoint mode (rows of A it is on, not steps)
__host__ __device__ inline i64 dp_ck_bytes_written(i64 la, i64 lb, int C, int tail) {
  return (dp_ck_groups(lb, C, tail) * la * 2 + dp_ck_nck(la) * dp_padded_cols(lb, C, tail) * 2) * 4;
}

// ------------------------------------------------------------------------------------------------------------------
// The band (dp_walk.hip): for a batch of few pairs the walk is a chain of blocks on one wavefront per pair with the rest of the
// chip idle, so the decision bits of the DP_BAND_BLOCKS row blocks of every column group that lie around the straight line
// from (0, 0) to (La, Lb) are computed up front, all at once; the walk reads them and recomputes only where the path leaves
// the band.  dp_band_row_block: the row block of column group gg (first fill lane l0) that the line crosses at the group's
// middle column.
#ifndef DP_BAND_BLOCKS
#define DP_BAND_BLOCKS 3
#endif
__host__ __device__ inline int dp_band_row_block(int la, int lb, int bw, int gg, int l0) {
  // (only the two kernels of dp_walk.hip call this, with the same arguments, so all that matters is that it is a function of
  // them: double arithmetic instead of a 64-bit integer division, which costs the walk kernel some 70 registers)
  const double jm = fmin((double)gg * bw + bw / 2, (double)lb);
  int im = (int)(jm * (double)la / (double)(lb > 0 ? lb : 1) + 0.5);
  im = im < 1 ? 1 : (im > la ? la : im);
  return (im - 1 + l0) / DP_CK_R;
}
// bytes of one block's decisions as the walk keeps them in LDS: DP_CK_R rows of `lanes_per_pair` words holding 4 bits for each
// of the lane's columns (1, 2 or 4 bytes)
__host__ __device__ inline int dp_band_block_bytes(int cols_per_lane, int lanes_per_pair) {
  const int c2 = cols_per_lane * DP_CK_W / lanes_per_pair;
  return DP_CK_R * lanes_per_pair * (c2 <= 2 ? 1 : (c2 <= 4 ? 2 : 4));
}
struct DpBand {          // device pointers; work == nullptr: no band
  const int *work;       // (pair, column group) of every work item of the band kernel, two ints each
  i64 n_work;
  unsigned *bits;        // block b of group gg of pair p at block index off[p] + gg * DP_BAND_BLOCKS + b
  const i64 *off;        // indexed by pair
};

// ------------------------------------------------------------------------------------------------------------------
// Device code shared by the fill kernel and the walk: the hand-scheduled cell of the recurrence.
#if defined(__HIPCC__)
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(int a, int b, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), acc, false);
}

__device__ __forceinline__ int pack16(int lo, int hi) { return (int)(((unsigned)lo & 0xffffu) | ((unsigned)hi << 16)); }

// One cell of the recurrence, hand-scheduled.  The compiler's version of the same cell spends one more VALU op on
// a register copy (the old H of the row above must survive as the next cell's diagonal while the new H is written);
// here the next cell's diagonal term is folded into its score (dn = hop + gap-row dot) BEFORE hop is overwritten, so
// every per-column register is updated in place: 14 ops with int8 weights (DOT4), 15 with int16.
//   dp   in : diag + dot2(gap row)  for this cell          dn  out: the same for the next cell (unless LAST)
//   hl       : H~ - gop of the cell to the left            hop in/out: H~ - gop of the row above / of this cell
//   ax, az   : A's base counts (four int8, or two int16 pairs), ay : A's (nGap, 1)
// gfx950 needs 3 independent instructions between a dot op and a different op that reads its result: the orders
// below keep >= 3 everywhere (score -> max3, dot2 -> the next cell's score).
#define PM_CELL_E_T "v_sub_u32 %[t], %[hl], %[e]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\tv_max_i32 %[e], %[e], %[hl]\n\t"
#define PM_CELL_F_T "v_sub_u32 %[t], %[hop], %[f]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\tv_max_i32 %[f], %[f], %[hop]\n\t"
#define PM_CELL_E "v_max_i32 %[e], %[e], %[hl]\n\t"
#define PM_CELL_F "v_max_i32 %[f], %[f], %[hop]\n\t"
#define PM_CELL_SCORE4 "v_dot4c_i32_i8 %[dp], %[ax], %[w0]\n\t"
#define PM_CELL_SCORE2 "v_dot2_i32_i16 %[dp], %[ax], %[w0], %[dp]\n\tv_dot2_i32_i16 %[dp], %[az], %[w1], %[dp]\n\t"
#define PM_CELL_NEXT "v_dot2_i32_i16 %[dn], %[ay], %[w2n], %[hop]\n\t"
// uniform depth (every column of A holds the same number of symbols): the gap row of the score is folded into the base
// weights and a per-column constant, so the next cell's diagonal is a plain add (full-rate) instead of a dot2 (half-rate)
#define PM_CELL_NEXT_U "v_add_u32 %[dn], %[hop], %[w2n]\n\t"
#define PM_CELL_H "v_max3_i32 %[h], %[dp], %[e], %[f]\n\t"
#define PM_CELL_H_T                                                                                                    \
  "v_max3_i32 %[h], %[dp], %[e], %[f]\n\tv_sub_u32 %[t], %[dp], %[h]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\t"   \
  "v_sub_u32 %[t], %[e], %[f]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\t"
#define PM_CELL_OUT "v_subrev_u32 %[hop], %[gop], %[h]"
#define PM_CELL_OPERANDS                                                                                               \
  [dp] "+v"(dp), [dn] "=&v"(dn), [e] "+v"(e), [f] "+v"(f), [hop] "+v"(hop), [acc] "+v"(acc), [t] "=&v"(t), [h] "=&v"(h)  \
      : [hl] "v"(hl), [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [w0] "v"(w0), [w1] "v"(w1), [w2n] "v"(w2n), [gop] "s"(gop)
// LAST: 0 = a column in the middle of the lane's run; 1 = the lane's last column, nothing prepared for a next cell (the walk, which
// forms its neighbour's diagonal itself); 2 = the lane's last column, and dn = the diagonal term of the RIGHT NEIGHBOUR's first cell
// (w2n = that column's constant, dp_fill_kernel: the neighbour takes it with one DPP move instead of a move and an add).  1 and 2 end
// in an s_nop that keeps hop two wait states away from the DPP read that follows the step.
template <bool TRACE, int LAST, bool DOT4, bool UNI = false>
__device__ __forceinline__ void dp_cell(int &dp, int &dn, int &e, int &f, int &hop, unsigned &acc, int hl, int ax, int ay,
                                        int az, int w0, int w1, int w2n, int gop) {
  int t, h;
#define PM_CELL_ASM(BODY) asm volatile(BODY : PM_CELL_OPERANDS)
#define PM_CELL_TAIL(BODY) \
  if(LAST == 2) {          \
    PM_CELL_ASM(BODY "\n\ts_nop 1"); \
  }                        \
  else {                   \
    PM_CELL_ASM(BODY);     \
  }
  if(TRACE) {
    if(LAST == 1) {
      if(DOT4) {
        PM_CELL_ASM(PM_CELL_E_T PM_CELL_SCORE4 PM_CELL_F_T PM_CELL_H_T PM_CELL_OUT "\n\ts_nop 1");
      }
      else {
        PM_CELL_ASM(PM_CELL_E_T PM_CELL_SCORE2 PM_CELL_F_T PM_CELL_H_T PM_CELL_OUT "\n\ts_nop 1");
      }
    }
    else if(UNI) {
      if(DOT4) {
        PM_CELL_TAIL(PM_CELL_E_T PM_CELL_SCORE4 PM_CELL_F_T PM_CELL_NEXT_U PM_CELL_H_T PM_CELL_OUT)
      }
      else {
        PM_CELL_TAIL(PM_CELL_E_T PM_CELL_SCORE2 PM_CELL_F_T PM_CELL_NEXT_U PM_CELL_H_T PM_CELL_OUT)
      }
    }
    else {
      if(DOT4) {
        PM_CELL_TAIL(PM_CELL_E_T PM_CELL_SCORE4 PM_CELL_F_T PM_CELL_NEXT PM_CELL_H_T PM_CELL_OUT)
      }
      else {
        PM_CELL_TAIL(PM_CELL_E_T PM_CELL_SCORE2 PM_CELL_F_T PM_CELL_NEXT PM_CELL_H_T PM_CELL_OUT)
      }
    }
  }
  else { // score only: 6 (7) ops; s_nops stand in for the decision ops that separate the dot ops from their readers
    if(LAST == 1) {
      if(DOT4) {
        PM_CELL_ASM(PM_CELL_E PM_CELL_SCORE4 PM_CELL_F "s_nop 1\n\t" PM_CELL_H PM_CELL_OUT "\n\ts_nop 1");
      }
      else {
        PM_CELL_ASM(PM_CELL_E PM_CELL_SCORE2 PM_CELL_F "s_nop 1\n\t" PM_CELL_H PM_CELL_OUT "\n\ts_nop 1");
      }
    }
    else if(UNI) {
      if(DOT4) {
        PM_CELL_TAIL(PM_CELL_E PM_CELL_SCORE4 PM_CELL_F PM_CELL_NEXT_U "s_nop 0\n\t" PM_CELL_H PM_CELL_OUT)
      }
      else {
        PM_CELL_TAIL(PM_CELL_E PM_CELL_SCORE2 PM_CELL_F PM_CELL_NEXT_U "s_nop 0\n\t" PM_CELL_H PM_CELL_OUT)
      }
    }
    else {
      if(DOT4) {
        PM_CELL_TAIL(PM_CELL_E PM_CELL_SCORE4 PM_CELL_F PM_CELL_NEXT "s_nop 0\n\t" PM_CELL_H PM_CELL_OUT)
      }
      else {
        PM_CELL_TAIL(PM_CELL_E PM_CELL_SCORE2 PM_CELL_F PM_CELL_NEXT "s_nop 0\n\t" PM_CELL_H PM_CELL_OUT)
      }
    }
  }
#undef PM_CELL_TAIL
#undef PM_CELL_ASM
}

// B's packed column -> the cell's weight registers: w[a] = sum_b count[b] * sub[a][b]; w0/w1 hold the four base weights
// (four int8 with DOT4, else two int16 pairs), w2 = (w[gap], go + ge): A's third pair is (nGap, 1), the 1 picks up gop + 2 * ge.
// UNI (every column of A holds rows_a symbols, so nGap = rows_a - nA - nC - nG - nT): the score sum_a A[a] * w[a] is rewritten
// as sum_{a < 4} A[a] * (w[a] - w[gap]) + rows_a * w[gap]; w0/w1 hold the differences and w2 the whole per-column constant
// rows_a * w[gap] + go + ge (a plain int, added instead of dotted).
template <bool DOT4, bool UNI = false>
__device__ __forceinline__ void dp_column_weights(u64 col, bool in, const DpParamsD &P, int &w0, int &w1, int &w2, int rows_a = 0) {
  int cb[5], w[5];
#pragma unroll
  for(int b = 0; b < 5; ++b) {
    cb[b] = (int)((col >> (8 * b)) & 0xff);
  }
#pragma unroll
  for(int a = 0; a < 5; ++a) {
    int acc = 0;
#pragma unroll
    for(int b = 0; b < 5; ++b) {
      acc += cb[b] * P.sub[a * 5 + b];
    }
    w[a] = acc;
  }
  if(UNI) {
    w2 = in ? rows_a * w[4] + P.go + P.ge : 0;
#pragma unroll
    for(int a = 0; a < 4; ++a) {
      w[a] -= w[4];
    }
  }
  if(DOT4) {
    w0 = (int)(((unsigned)w[0] & 0xffu) | (((unsigned)w[1] & 0xffu) << 8) | (((unsigned)w[2] & 0xffu) << 16) | ((unsigned)w[3] << 24));
    w1 = 0;
  }
  else {
    w0 = pack16(w[0], w[1]);
    w1 = pack16(w[2], w[3]);
  }
  if(!UNI) {
    w2 = pack16(w[4], in ? P.go + P.ge : 0);
  }
}

// A's packed column -> what the cell reads per row: x = nA, nC, nG, nT as four int8 (DOT4) or (nA, nC) as an int16 pair,
// y = (nGap, 1), z = (nG, nT) as an int16 pair (unused with DOT4), w = 0.
template <bool DOT4>
__device__ __forceinline__ int4 dp_expand_row(u64 col) {
  int4 v = make_int4(0, 0, 0, 0);
  if(DOT4) {
    v.x = (int)(col & 0xffffffffull);
  }
  else {
    v.x = (int)(col & 0xff) | ((int)((col >> 8) & 0xff) << 16);
    v.z = (int)((col >> 16) & 0xff) | ((int)((col >> 24) & 0xff) << 16);
  }
  v.y = (int)((col >> 32) & 0xff) | (1 << 16);
  return v;
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// The walk beside the fill of the SAME launch (round 5).  A pair's checkpoints are plain stores: they sit in the L2 of the XCD whose
// CU wrote them until that L2 writes them back, and the eight L2s are not coherent with each other -- which is why the walk used to
// be a kernel of its own behind the fill kernel (a kernel's end makes everything visible everywhere).  But a wavefront on the SAME
// XCD reads that L2.  So the fill wavefront that has finished a pair (one wavefront per pair, all stripes; its stores acknowledged:
// s_waitcnt vmcnt(0)) appends the pair's position to the list of the XCD it really runs on -- HW_REG_XCC_ID, read, not inferred from
// blockIdx -- and a persistent walk kernel launched beside the fill kernel (mode 1: a wavefront or two per SIMD, the fill keeps the
// rest) has every group take the next entry of the list of the XCD IT really runs on, wait until the entry is written, and walk
// that pair while the fill goes on.  No placement is assumed: a pair lies in exactly one list, whatever the dispatcher did, and an
// entry is taken exactly once (a ticket per XCD).  When the fill kernel has ended, everything is visible everywhere: a second launch
// of the walk kernel behind it (mode 2, the whole chip) takes what is left of all eight lists.  The fill never waits for a walker,
// the early walkers leave the fill most of every SIMD, and every wait ends when the fill has (n_filled) -- PROVIDED the fill kernel runs
// beside them.  Where kernels are run one at a time (rocprofv3 --pmc serialises the dispatches of all streams, and took the walkers
// first: the profile refresh of round 5 hung there) it never starts while they wait; so a walker first waits, for 30 ms at most, for the
// fill kernel's sign that it is running (n_filled[1]) and leaves if it does not come -- the launch behind the fill walks everything then.
struct DpEarly {   // device pointers; mode 0: the walk as a kernel of its own (order[0 .. n), one pair per group)
  int mode;        // 1: beside the fill kernel, the lists of this wavefront's own XCD only; 2: behind it, all lists
  int n_expected;  // pairs the fill kernel will publish
  int *taken;      // [8] entries handed out per XCD
  const int *n_filled; // pairs published so far (incremented AFTER the entry is written); n_filled[1]: the fill kernel has started
  const int *list; // [8][n]: position + 1, 0 = not written yet
};
// what the fill kernel needs to publish a finished pair (null pointers: it does not)
struct DpFilled {
  int *resv;     // [8] entries reserved per XCD
  int *n_filled;
  int *list;     // [8][n]
  int n;
};
// HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this wavefront runs on
#define DP_GETREG_XCC_ID ((3 << 11) | (0 << 6) | 20)

// dp_walk.hip: one launch walks the paths of the pairs order[0 .. n) from their checkpoints (tb_off is indexed by pair).
// cols_per_lane (of the fill kernel) in {8, 16}; lanes_per_pair a power of two for which dp_walk_lanes_ok() holds.
int dp_launch_walk(int cols_per_lane, int lanes_per_pair, bool dot4, const u64 *cols_a, const i64 *off_a, const u64 *cols_b, const i64 *off_b,
                   const int *order, i64 n, const i64 *tb_off, const unsigned *ck, unsigned char *ops, int *n_ops, const DpParamsD &P,
                   const DpBand &band, int tail, int urgent, hipStream_t stream, const DpEarly *early = nullptr, unsigned early_groups = 0);

bool dp_walk_lanes_ok(int cols_per_lane, int lanes_per_pair);

} // namespace pm
