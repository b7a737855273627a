// dp_kernels.hip -- profile x profile affine-gap DP (global alignment), gfx950.
//
// NO REFERENCE COUNTERPART: orbitz/paramugsy contains no DP, no scores and no traceback.  The
// computation is specified by this repo (oracle/dp_oracle.h, DESIGN.md "Profile DP specification") because
// BASELINE.json's metric is profile-DP GCUPS; the kernel is checked against that scalar oracle only.
//
// Mapping (DESIGN.md "dp_fill_kernel"):
//   * one 64-lane wavefront per profile pair, one pair per workgroup, thousands of pairs per launch;
//   * the columns of B are cut into stripes of 64 lanes x C columns; a lane keeps its C columns' state
//     (H-gap_open of the previous row, F, and the columns' substitution weights) in registers for the whole
//     stripe: 4-5 VGPRs per column;
//   * the wave sweeps the rows of A as an anti-diagonal wavefront: at step t lane l is on row t-l.  The two
//     values a row hands to the next lane (H-gap_open and E of the lane's last column) move with one
//     v_mov_b32_dpp wave_shr:1 each;
//   * A's packed columns are loaded 64 at a time (one coalesced 512-byte load per 64 steps), expanded to
//     int8/int16 lanes and staged in a 128-row LDS ring (every row at two places 128 apart, so that a block's reads are
//     consecutive); each lane reads its row with one ds_read_b64 (b96 with int16 weights);
//   * the column score is sum-of-pairs = v_dot4_i32_i8 + v_dot2_i32_i16 (or 3 x v_dot2_i32_i16) accumulating onto
//     the diagonal (the last half lane of the gap-row dot carries gap_open so that the stored H-gap_open needs no
//     correction);
//   * max-plus recurrence in int32 VALU (no MFMA: nothing to contract), carried in SKEWED coordinates
//     V~[i][j] = V[i][j] + (i+j)*gap_extend for V in {H,E,F}: extending a gap then costs nothing
//     (E~[i][j] = max(E~[i][j-1], H~[i][j-1] - (gap_open-gap_extend))), which removes the two "- gap_extend"
//     subtractions per cell; every comparison is between values of one cell, so all decisions are unchanged, and
//     the score is un-skewed once at the end;
//   * traceback: 4 decision bits per cell, shifted into a word with v_alignbit_b32 (one op per bit); four steps'
//     words are staged in LDS and leave as one contiguous tile: tb[stripe][tile][lane][4 steps][C/8 words];
//   * stripe boundary (last column of a stripe, per row): written by lane 63, read back 64 rows at a time.
// Algorithmic HBM traffic per cell: 0.5 byte of traceback written + (8 bytes per column of A per stripe +
// 8 bytes per column of B) read, i.e. ~0.5 B/cell for kilobase profiles; the kernel is VALU-issue bound
// (14 VALU ops per cell with int8 weights, 15 with int16: dp_cell below), not HBM bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "dp_internal.hpp"
#include "pm_internal.hpp"

namespace pm {

#define DPP_WAVE_SHR1 0x138

// What the fill kernel leaves behind for the path (dp_internal.hpp): nothing (scores only), 4 decision bits per cell,
// or the row/column checkpoints the walk of dp_walk.hip recomputes blocks from.
enum { DP_MODE_SCORE = 0, DP_MODE_BITS = 1, DP_MODE_CKPT = 2 };

// lane l takes lane l-1's value; lane 0 takes `lane0`.  Needs all 64 lanes enabled.
__device__ __forceinline__ int from_left(int lane0, int v) {
  return __builtin_amdgcn_update_dpp(lane0, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}

// Words of traceback one pair needs.  Layout tb[stripe][tile][lane][4 steps][C/8 words]: a tile is four consecutive
// steps; a lane's words of one tile are contiguous (16 or 32 bytes), so the backward walk, which follows one lane
// through consecutive steps, finds four cells' decisions in one place instead of in four 256/512-byte rows.
__host__ __device__ inline i64 dp_tb_words(i64 la, i64 lb, int C) {
  i64 W = 64 * C;
  i64 stripes = (lb + W - 1) / W;
  i64 tiles = (la + 63 + 3) / 4;
  return stripes * tiles * 64 * 4 * (C / 8);
}

// What a pair's stripes share (uniform over the wavefront).
struct DpFillPair {
  const u64 *A, *B;
  int la, lb, steps, cstride, tiles, n_stripes, gop, ge; // cstride: entries between two groups' column checkpoints (dp_ck_stride)
  unsigned *tbp;   // the pair's workspace: decision bits, or checkpoints
  int2 *bp;        // bits / score mode: the seam, one {H~ - gop, E~} per row of A
  i64 row_base;    // checkpoint mode: word offset of the row checkpoints in the workspace
  int nck;         // checkpoint mode: row checkpoints per lane
  int ng, team, tw;
  int *gp;         // ng > 1: the pair's progress words in global memory
  int *pipe_error;
  // tiles (dp_fill_tiles_kernel): this call runs the 64-step blocks [t_begin, t_end) of the stripe and hands the lanes' state on
  int t_begin, t_end;
  unsigned *state; // the stripe's lane state between two tiles: [dword][lane], 2 C + 4 dwords
  int *tdone;      // steps of the stripe done and saved so far
};

// One work item of dp_fill_tiles_kernel: the blocks [t_begin, t_end) of stripe s of the pair at position pos of the launch.
struct DpTile {
  int pos, s, t_begin, t_end;
};

// One stripe of a pair: 64 lanes x CS columns of B from column st.jb on, against all rows of A.  C is the batch's columns per lane
// (the full stripes' CS, and what the checkpoints' column groups are made of); the narrow last stripes of dp_internal.hpp have CS < C.
//
// A step (lane l on row t - l of A) is 6 CS VALU instructions of cells and, since round 4, three more:
//   * what the lane's row takes over from the lane to its left -- {H~ - gop, E~} of that lane's last column, and the diagonal term
//     of the lane's first cell -- arrives with three v_mov_b32_dpp wave_shr:1.  The diagonal term (H~ - gop of the row above, plus
//     the column's constant) is formed by the LEFT lane, in the slot of its last cell that used to be an s_nop (dp_cell, LAST = 2),
//     so the receiving lane no longer keeps last step's hand-over and adds to it (a move and an add per step);
//   * lane 0 has no lane to its left: its three values are the stripe's left boundary (the seam of the stripe before, or the DP's
//     column 0).  They are written into LDS next to the rows of A when those are staged (every 64 steps), so they arrive in the
//     DPP instructions' `old` operand with the lane's read of its row -- no v_readlane / v_writelane per step;
//   * a block is always 64 steps (the steps behind the stripe's last have an empty EXEC mask), so the step loop has a constant
//     trip count and unrolls by four with its LDS and checkpoint addresses as immediate offsets.
// TILED (dp_fill_tiles_kernel): the call runs only the blocks [pp.t_begin, pp.t_end) of the stripe.  What a lane carries from block to
// block -- hop[], f[], e, dn_last and the boundary's hl_carry: 2 CS + 3 dwords -- is loaded from pp.state at the start (after the
// tile before has published it: pp.tdone) and stored there at the end; A's 64 rows before the first block are staged again (the
// lanes right of lane 0 are still on them).  Everything the wavefront shares with other wavefronts then goes the way it goes between
// the workgroups of a pair (ng > 1): progress words in global memory, the seam read and written past the XCD's L2.
template <int C, int CS, int MODE, bool DOT4, int NW, bool UNI, bool TILED = false>
__device__ __forceinline__ void dp_fill_stripe(const DpFillPair &pp, const DpParamsD &P, const int s, const DpStripe st, int4 *ring, int4 *sbnd,
                                               int2 *cstage, unsigned *tbstage, int *progress, const int wv, const int lane, int &result) {
  constexpr bool TRACE = MODE == DP_MODE_BITS;
  constexpr bool CKPT = MODE == DP_MODE_CKPT;
  constexpr bool SYNC = NW > 1 || TILED; // the stripe to the left is another wavefront's
  static_assert(!TILED || (NW == 1 && !TRACE), "tiles: one wavefront per workgroup, no decision bits");
  constexpr int TBW = TRACE ? CS / 8 : 1;
  constexpr int TBS = TRACE ? 64 * TBW : 1;
  constexpr bool PACKED = DOT4 && UNI; // a row of A is one dword: the boundary's three ride in the same 16-byte ring entry
  constexpr int GW = DP_CK_W * C / CS; // lanes per column group
  static_assert(!TRACE || (CS == C && CS % 8 == 0), "decision bits: whole words per lane per step, full stripes only");
  const int la = pp.la, lb = pp.lb, steps = pp.steps, gop = pp.gop, ng = pp.ng, team = pp.team;
  const u64 *A = pp.A, *B = pp.B;
  unsigned *tbp = pp.tbp;
  const int jb = st.jb;
  const int j0 = jb + lane * CS; // this lane's first column of B (0-based)
  const i64 g0 = jb / (DP_CK_W * C); // the stripe's first column group
  int w0[CS], w1[CS], w2[CS], hop[CS], f[CS];
#pragma unroll
  for(int c = 0; c < CS; ++c) {
    const int j = j0 + c;
    const bool in = j < lb;
    dp_column_weights<DOT4, UNI>(in ? B[j] : 0ull, in, P, w0[c], w1[c], w2[c], P.rows_a);
    hop[c] = -2 * gop; // H~[0][j+1] - gop, H~[0][j] = -gop for j >= 1
    f[c] = DP_NEG_INF;
  }
  const int w2nb = __shfl_down(w2[0], 1);                        // the constant of the right neighbour's first column
  const int w2_l0 = __builtin_amdgcn_readfirstlane(w2[0]);       // and of lane 0's
  int e = DP_NEG_INF; // the running E of this lane's row; between steps: what the lane hands to its right neighbour
  int dn_last = 0;    // between steps: the diagonal term of the right neighbour's first cell
  // H~ - gop above-left of lane 0's first cell of the block's first row: H~[0][jb] - gop at the top, then the boundary's last row
  int hl_carry = (jb == 0 ? 0 : -gop) - gop;
  if constexpr(TILED) {
    if(pp.t_begin > 0) {
      // the tile before this one (drawn earlier from the queue, so running or done) has saved the lanes' state
      int spins = 0;
      while(__hip_atomic_load(pp.tdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pp.t_begin) {
        if((spins & 1023) == 0 && __hip_atomic_load(pp.pipe_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          break; // the launch has already failed somewhere: do not wait, let the grid drain
        }
        __builtin_amdgcn_s_sleep(8);
        if(++spins > (1 << 22)) {
          if(lane == 0) {
            atomicOr(pp.pipe_error, 1);
          }
          break;
        }
      }
      asm volatile("" ::: "memory");
      const int *sv = reinterpret_cast<const int *>(pp.state) + lane;
#pragma unroll
      for(int c = 0; c < CS; ++c) {
        hop[c] = __hip_atomic_load(sv + c * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f[c] = __hip_atomic_load(sv + (CS + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      e = __hip_atomic_load(sv + 2 * CS * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dn_last = __hip_atomic_load(sv + (2 * CS + 1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hl_carry = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv + (2 * CS + 2) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      // rows [t_begin - 64, t_begin) of A: lane l of the first block's step t is on row t - l
      const int rp = pp.t_begin - 64 + lane;
      const int4 vp = rp < la ? dp_expand_row<DOT4>(A[rp]) : make_int4(0, 0, 0, 0);
      if constexpr(PACKED) {
        const int4 ent = make_int4(vp.x, 0, 0, 0);
        ring[rp & 127] = ent;
        ring[(rp & 127) + 128] = ent;
      }
      else {
        ring[rp & 127] = vp;
        ring[(rp & 127) + 128] = vp;
      }
    }
  }
  if(jb > 0 && ng == 1) { // lane 63's stores of the stripe to the left must be visible to every lane's loads
    if constexpr(CKPT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (they are inline asm, the fence below does not know of them)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // (several workgroups: the seam is read with agent-scope atomic loads)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // Column checkpoints ({H~ - gop, E~} of a group's last column, row by row): the lanes that close a group put theirs into LDS step
  // by step -- cstage[group][step & 15], 17 slots a group so that the lanes' writes fall into different banks -- and every 16 steps
  // the wavefront writes the 16 steps of every group out, 16 lanes per group: whole 128-byte lines, four groups per store (a store
  // per lane and step was 16 partial-line write requests per step; L2 write requests are what this kernel runs short of)
  constexpr int NGRP = 64 / GW;
  const int2 *const ck_col = reinterpret_cast<const int2 *>(tbp) + g0 * pp.cstride; // group 0 of the stripe, step 0
  const bool closes_group = (lane & (GW - 1)) == GW - 1;
  unsigned cs_put = (unsigned)(size_t)cstage + (unsigned)(lane / GW) * 136u;              // + (step & 15) * 8
  const unsigned cs_get = (unsigned)(size_t)cstage + (unsigned)(lane >> 4) * 136u + (unsigned)(lane & 15) * 8u; // + 4 i groups
  const unsigned ck_put = ((unsigned)(lane >> 4) * (unsigned)pp.cstride + (unsigned)(lane & 15)) * 8u;           // the same in HBM

  for(int t0 = TILED ? pp.t_begin : 0; t0 < (TILED ? min(pp.t_end, steps) : steps); t0 += 64) {
    {
      // stage rows [t0, t0+63] of A (one coalesced 8-byte load per lane, expanded) and the left boundary of the same rows
      const int r = t0 + lane;
      const int4 v = r < la ? dp_expand_row<DOT4>(A[r]) : make_int4(0, 0, 0, 0);
      if(SYNC && s > 0) {
        // rows [t0, t0+63] of the left stripe's boundary must have been published by its wave
        const int need = ((s - 1) / team) * la + min(t0 + 64, la);
        int spins = 0;
        if(ng > 1) {
          int *word = pp.gp + (s - 1) % team;
          while(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if((spins & 1023) == 0 && __hip_atomic_load(pp.pipe_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              break; // the launch has already failed somewhere: do not wait, let the grid drain
            }
            __builtin_amdgcn_s_sleep(4);
            if(++spins > (1 << 22)) {
              if(lane == 0) {
                atomicOr(pp.pipe_error, 1);
              }
              break;
            }
          }
          asm volatile("" ::: "memory");
        }
        else {
          volatile int *word = &progress[(s - 1) % NW];
          while(*word < need) {
            if((spins & 1023) == 0 && __hip_atomic_load(pp.pipe_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              break;
            }
            __builtin_amdgcn_s_sleep(2);
            if(++spins > (1 << 22)) {
              if(lane == 0) {
                atomicOr(pp.pipe_error, 1);
              }
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
      }
      // what the column left of the stripe hands to lane 0 for row r: the seam of the stripe to the left, or for the first stripe
      // the DP's own column 0, H~[r+1][0] - gop = -2 gop with no E
      int2 b = make_int2(-2 * gop, DP_NEG_INF);
      if(jb > 0 && r < la) {
        const int2 *src;
        if constexpr(CKPT) { // lane 63 closes a column group: its column checkpoints in the stripe to the left are the seam
          src = reinterpret_cast<const int2 *>(tbp + dp_ck_col_word(la, g0 - 1, (i64)r + 63));
        }
        else {
          src = pp.bp + r;
        }
        if(ng > 1) {
          const unsigned long long q =
              __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          b = make_int2((int)(unsigned)q, (int)(unsigned)(q >> 32));
        }
        else {
          b = *src;
        }
      }
      // the diagonal term of lane 0's first cell on row r: H~ - gop of the boundary one row up, plus the column's constant
      const int hl_up = from_left(hl_carry, b.x);
      hl_carry = __builtin_amdgcn_readlane(b.x, 63);
      const int dg = UNI ? hl_up + w2_l0 : dot2(v.y, w2_l0, hl_up);
      if constexpr(PACKED) {
        const int4 ent = make_int4(v.x, b.x, dg, b.y);
        ring[r & 127] = ent;
        ring[(r & 127) + 128] = ent;
      }
      else {
        ring[r & 127] = v;
        ring[(r & 127) + 128] = v;
        sbnd[lane] = make_int4(b.x, dg, b.y, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // where this lane's row of step t0 lies in the doubled ring: place (rbase & 127) + (row - rbase) with rbase = t0 - 64, the oldest
    // row the block can read: between 1 and 191 for the rows of the block.  (LDS byte addresses: the low half of the generic ones.)
    unsigned rrow = (unsigned)(size_t)ring + (unsigned)((((t0 - 64) & 127) + 64 - lane) * 16);
    unsigned brow = (unsigned)(size_t)sbnd;
    // one step; K = t & 3 is a constant, so that four steps share their address registers (immediate offsets)
    auto step = [&](auto kc, const int t) __attribute__((always_inline)) {
      constexpr int K = decltype(kc)::value;
      // the lane's row of A; and (lane 0) the boundary's values for it.  Separate dwords: a 16-byte read comes back as a register
      // tuple, and taking the DPP instructions' tied operands out of a tuple costs copies.
      int ax, ay = 0, az = 0, b_hl, b_dg, b_e;
      if constexpr(PACKED) {
        asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\t"
                     "ds_read_b32 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b_hl), "=&v"(b_dg), "=&v"(b_e), "=&v"(ax)
                     : "v"(rrow), "n"(K * 16 + 4), "n"(K * 16 + 8), "n"(K * 16 + 12), "n"(K * 16)
                     : "memory");
      }
      else { // the boundary: the same address in every lane, row t of the block's 64
        asm volatile("ds_read_b32 %0, %6 offset:%8\n\tds_read_b32 %1, %6 offset:%9\n\tds_read_b32 %2, %6 offset:%10\n\t"
                     "ds_read_b32 %3, %7 offset:%11\n\tds_read_b32 %4, %7 offset:%12\n\tds_read_b32 %5, %7 offset:%13\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b_hl), "=&v"(b_dg), "=&v"(b_e), "=&v"(ax), "=&v"(ay), "=&v"(az)
                     : "v"(brow), "v"(rrow), "n"(K * 16), "n"(K * 16 + 4), "n"(K * 16 + 8), "n"(K * 16), "n"(K * 16 + 4), "n"(K * 16 + 8)
                     : "memory");
      }
      // from the lane to the left (all 64 lanes enabled); hop[CS-1], e and dn_last change only in the steps a lane is on a row, so
      // they are what its right neighbour needs one step later
      const int ho_in = from_left(b_hl, hop[CS - 1]);
      const int d0 = from_left(b_dg, dn_last);
      e = from_left(b_e, e);
      // the lanes that are on a row of A at this step, 0 <= t - lane < la, are lanes max(0, t - la + 1) .. min(63, t): the mask is built
      // on the scalar unit and becomes EXEC as it is (no per-lane compare: v_cmp is a half-rate VALU instruction, once a step); it
      // is empty in the steps behind the stripe's last
      const int lane_lo = min(63, max(0, t - la + 1)), lane_hi = min(63, t);
      const unsigned long long on_a_row = t < steps ? (~0ull >> (63 - max(0, lane_hi - lane_lo))) << lane_lo : 0ull;
      if(__builtin_amdgcn_inverse_ballot_w64(on_a_row)) {
        unsigned accw[TBW];
#pragma unroll
        for(int k = 0; k < TBW; ++k) {
          asm volatile("" : "=v"(accw[k])); // no initial value needed: 8 cells x 4 bits shift every old bit out
        }
        {
          int dd[2];
          dd[0] = d0;
#pragma unroll
          for(int c = 0; c < CS; ++c) {
            const int hl = c == 0 ? ho_in : hop[c == 0 ? 0 : c - 1];
            if(c == CS - 1) {
              dp_cell<TRACE, 2, DOT4, UNI>(dd[c & 1], dd[(c + 1) & 1], e, f[c], hop[c], accw[TRACE ? c / 8 : 0], hl, ax, ay, DOT4 ? ax : az,
                                           w0[c], DOT4 ? w0[c] : w1[c], w2nb, gop);
            }
            else {
              dp_cell<TRACE, 0, DOT4, UNI>(dd[c & 1], dd[(c + 1) & 1], e, f[c], hop[c], accw[TRACE ? c / 8 : 0], hl, ax, ay, DOT4 ? ax : az,
                                           w0[c], DOT4 ? w0[c] : w1[c], w2[c + 1 < CS ? c + 1 : c], gop);
            }
          }
          dn_last = dd[CS & 1];
        }
        if constexpr(TRACE) { // into this lane's slot of the tile being assembled in LDS
#pragma unroll
          for(int k = 0; k < TBW; ++k) {
            tbstage[K * TBS + lane * TBW + k] = accw[k];
          }
        }
        if constexpr(!CKPT) {
          if(lane == 63 && s + 1 < pp.n_stripes) {
            if(ng > 1) {
              __hip_atomic_store(reinterpret_cast<unsigned long long *>(pp.bp + (t - 63)),
                                 (unsigned long long)(unsigned)hop[CS - 1] | ((unsigned long long)(unsigned)e << 32), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            }
            else {
              pp.bp[t - 63] = make_int2(hop[CS - 1], e);
            }
          }
        }
        if constexpr(CKPT) { // what this row hands to the next column group, into the staging of the current 16 steps
          if(closes_group) {
            const unsigned long long he = (unsigned long long)(unsigned)hop[CS - 1] | ((unsigned long long)(unsigned)e << 32);
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(cs_put), "v"(he), "n"(K * 8) : "memory");
          }
        }
      }
      if constexpr(CKPT) {
        // the lane's column state every DP_CK_R steps, the lanes of a group one step apart (after the same row of A): lane l
        // stores after the steps t with t + 1 - l % GW a positive multiple of DP_CK_R.  A scalar test lets GW of every DP_CK_R
        // steps through to the per-lane test (kept behind it by an opaque asm, or the compiler would hoist the vector compare
        // into every step).
        const int tm = (t + 1) & (DP_CK_R - 1);
        if(tm < GW && t < steps) {
          const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); // the lane, not kept for this
          int tc;
          asm volatile("v_sub_u32 %0, %1, %2" : "=v"(tc) : "s"(t + 1), "v"(ln & (GW - 1)));
          if(tc > 0 && (tc & (DP_CK_R - 1)) == 0) {
            // [lane][c]: the lane's CS columns are CS * 8 consecutive, line-aligned bytes, written here and now, store after store --
            // the line is whole when it leaves L2 (dp_internal.hpp).  8-byte stores: a 16-byte one wants its four registers side by
            // side, and the copies that takes cost the step loop of the 96-register kernels 16 spilled registers (measured at the
            // compiler).  One store after the other through inline asm: left to itself the compiler assembles all of them side by
            // side first, 32 registers that the step loop then does without
            const char *rowp = reinterpret_cast<const char *>(tbp + pp.row_base + ((i64)jb * pp.nck + (i64)(tc / DP_CK_R - 1) * 64 * CS) * 2) +
                               (unsigned)ln * (unsigned)(CS * 8);
#pragma unroll
            for(int c = 0; c < CS; ++c) {
              const unsigned long long hf = (unsigned long long)(unsigned)hop[c] | ((unsigned long long)(unsigned)f[c] << 32);
              asm volatile("global_store_dwordx2 %0, %1, off offset:%2" ::"v"(rowp), "v"(hf), "n"(c * 8) : "memory");
            }
          }
        }
      }
      if constexpr(TRACE) if((K == 3 && t < steps) || t == steps - 1) {
        // tile complete (or the stripe's last, partial tile): every lane writes its own 4 x TBW words, contiguously;
        // the wave's store covers one contiguous 1-2 KiB tile
        unsigned *dst = tbp + (((i64)s * pp.tiles + (t >> 2)) * 64 + lane) * (4 * TBW);
#pragma unroll
        for(int q = 0; q < 4; ++q) {
#pragma unroll
          for(int k = 0; k < TBW; ++k) {
            dst[q * TBW + k] = tbstage[q * TBS + lane * TBW + k];
          }
        }
      }
    };
    for(int tb = 0; tb < 64; tb += 4) {
      step(std::integral_constant<int, 0>(), t0 + tb);
      step(std::integral_constant<int, 1>(), t0 + tb + 1);
      step(std::integral_constant<int, 2>(), t0 + tb + 2);
      step(std::integral_constant<int, 3>(), t0 + tb + 3);
      rrow += 64;
      brow += 64;
      if constexpr(CKPT) {
        cs_put += 32;
        if((tb & 12) == 12) { // 16 steps staged: out with them (the slots of steps behind the stripe's last are not written)
          cs_put -= 128;
          const int t16 = t0 + tb - 12;
          const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
          if(t16 + (ln & 15) < steps) {
            const int2 *base = ck_col + t16;
            // (the compiler does not count these stores: whoever reads them in this kernel waits for vmcnt(0) itself)
            unsigned long long he[NGRP / 4];
#pragma unroll
            for(int i = 0; i < NGRP / 4; ++i) {
              asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(he[i]) : "v"(cs_get), "n"(i * 4 * 136) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for(int i = 0; i < NGRP / 4; ++i) {
              if(SYNC && i == NGRP / 4 - 1 && ng > 1) { // lane 63's group is the seam for another workgroup: past this XCD's L2
                asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(ck_put), "v"(he[i]), "s"(base + (i64)i * 4 * pp.cstride) : "memory");
              }
              else {
                asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(ck_put), "v"(he[i]), "s"(base + (i64)i * 4 * pp.cstride) : "memory");
              }
            }
          }
        }
      }
    }
    if(SYNC) {
      // lane 63 has stored the boundary of rows < t0 + 64 - 63: publish the count (cumulated over this wave's stripes)
      const int done = (s / team) * la + max(0, min(t0 + 1, la));
      if(ng > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // lane 63's seam stores have been acknowledged
        if(lane == 0) {
          __hip_atomic_store(pp.gp + pp.tw, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      else {
        if constexpr(CKPT) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if(lane == 0) {
          *(volatile int *)&progress[wv] = done;
        }
      }
    }
  }
  if constexpr(TILED) {
    if(pp.t_end < steps) { // not the stripe's last tile: the lanes' state for the next one, then the word that says it is there
      int *sv = reinterpret_cast<int *>(pp.state) + lane;
#pragma unroll
      for(int c = 0; c < CS; ++c) {
        __hip_atomic_store(sv + c * 64, hop[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sv + (CS + c) * 64, f[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(sv + 2 * CS * 64, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sv + (2 * CS + 1) * 64, dn_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sv + (2 * CS + 2) * 64, hl_carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // those, and the checkpoints' inline-asm stores
      if(lane == 0) {
        __hip_atomic_store(pp.tdone, pp.t_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if(s == pp.n_stripes - 1 && (!TILED || pp.t_end >= steps)) {
    const int jj = lb - 1 - jb;
    const int cstar = jj % CS;
    int hv = hop[0];
#pragma unroll
    for(int c = 1; c < CS; ++c) {
      hv = c == cstar ? hop[c] : hv;
    }
    result = __builtin_amdgcn_readlane(hv, jj / CS) + gop - (la + lb) * pp.ge; // un-skew
  }
}

// DOT4: every count of A and every ACGT weight of B fits int8 (checked at batch creation), so the four base terms of
// the column score are one v_dot4_i32_i8 instead of two v_dot2_i32_i16.
// NW: wavefronts per pair.  NW = 1 is the mapping described above.  NW > 1 (used when a launch has too few pairs to
// fill the chip with one wave each) gives the pair a workgroup of NW waves; wave w runs stripes w, w+NW, ... and a
// stripe may start a 64-row block as soon as the stripe to its left has published those rows of its boundary:
// a pipeline of stripes, synchronised through one progress word per wave in LDS.  Waits are bounded (a timeout sets
// *pipe_error and lets the wave run on, so the grid always drains).
// NG (run time, `ng`): workgroups per pair.  A workgroup sits on one CU, so with fewer pairs than CUs most of the chip would idle:
// the pair's stripes are then dealt round robin to the NG * NW waves of NG consecutive workgroups, the progress words live
// in global memory (`gprog`, this launch's own, zeroed by the host before the launch), and the seam values and the progress words
// are written and read as agent-scope atomics (sc1: past the XCD's L2), ordered by the wave's own s_waitcnt vmcnt(0) -- agent-scope
// release / acquire FENCES write back and invalidate the whole L2 of the XCD every time and made 128 pairs of 32 x 10 kbp 3.5x slower.
// The host asks for this only while every workgroup can have a CU of its own (n * NG <= CUs).  No workgroup can wait for ever on
// one that has not started, by construction: a workgroup does not take its (pair, team slot) from blockIdx -- HIP promises no
// dispatch order -- but from a ticket it draws at entry (the word after the progress words, zeroed with them).  The tickets
// drawn so far are 0 .. T-1 and their workgroups have all started; a pair whose NG tickets are all below T has every one of its
// waves running and depends on nothing else, so it finishes whatever the rest of the device is doing; the one pair that straddles
// T holds fewer than NG <= 16 workgroups, which cannot fill the chip, so the dispatcher goes on starting workgroups and T
// grows.  The bounded waits stay as the last line of defence, and a wave that finds *pipe_error set (by any wave of the grid)
// stops waiting at once, so a failed launch drains in milliseconds instead of paying the time-out at every block.
// tail: the narrow last stripes of dp_internal.hpp (kernels with 16 columns per lane that store no decision bits).
// A finished pair for the walk that runs beside this launch (DpFilled / DpEarly, dp_internal.hpp): the wavefront's stores have been
// acknowledged, so they are in this XCD's L2, where a walker of the same XCD reads them; the entry goes into that XCD's list, and
// the count of published pairs goes up only when the entry is written.
__device__ __forceinline__ void dp_publish_filled(const DpFilled &fl, int pos, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if(lane == 0) {
    const int x = (int)(__builtin_amdgcn_s_getreg(DP_GETREG_XCC_ID) & 7);
    const int slot = __hip_atomic_fetch_add(fl.resv + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fl.list + (i64)x * fl.n + slot, pos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(fl.n_filled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int C, int MODE, bool DOT4, int NW, bool UNI>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(DOT4 && NW <= 4 ? 5 : 4)))
dp_fill_kernel(const u64 *__restrict__ cols_a, const i64 *__restrict__ off_a, const u64 *__restrict__ cols_b,
               const i64 *__restrict__ off_b, const int *__restrict__ order, const i64 *__restrict__ tb_off, unsigned *__restrict__ tb,
               int2 *__restrict__ bnd, int *__restrict__ scores, int *__restrict__ pipe_error, DpParamsD P, int ng, int *__restrict__ gprog,
               int *__restrict__ started, int tail, DpFilled fl) {
  static_assert(C % 8 == 0, "whole traceback words per lane per step");
  if(NW > 1) { // several wavefronts per pair: a launch of few, long pairs (a tier beside the launch of the rest): ahead of those at issue
    __builtin_amdgcn_s_setprio(2);
  }
  else if(fl.list) {
    // walkers run beside this launch (DpEarly): the fill goes first at issue, the walkers take the cycles it leaves (a fifth of them:
    // the fill kernel issues a VALU instruction in four cycles of five) -- side by side at EQUAL priority every instruction of a walker
    // is one the fill does not issue, and the fill took 1.4 ms longer for a walk that costs 1.0 ms behind it (profiles/r05_early_walk.txt).
    // (The fill ahead of the walk kernels of EARLIER chunks too, always: measured, no difference -- 146.4 / 146.1 ms on 50 000 pairs.)
    __builtin_amdgcn_s_setprio(1);
    if(blockIdx.x == 0 && threadIdx.x == 0) { // "this launch is running": what the walkers beside it wait for before they wait for anything else
      __hip_atomic_store(fl.n_filled + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if(started && threadIdx.x == 0) { // dp_gate_kernel: once every workgroup of this launch has started, the next chunk's may
    __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  constexpr bool TRACE = MODE == DP_MODE_BITS;
  constexpr bool CKPT = MODE == DP_MODE_CKPT;
  constexpr bool TAILS = C == 16 && !TRACE; // the narrow stripes are compiled in
  constexpr int TBW = C / 8;
  constexpr bool PACKED = DOT4 && UNI;
  // A's expanded rows, 128 rows deep, every row written at two places 128 apart: the 127 rows a 64-step block reads (64 lanes one row
  // apart, 64 steps) then lie at consecutive places whatever the block, and a lane's read address only advances, 16 bytes a step
  __shared__ int4 ring_all[NW][256];
  __shared__ int4 sbnd_all[NW][PACKED ? 1 : 64]; // the left boundary of the block's 64 rows, where it does not fit the ring's entries
  __shared__ int2 cstage_all[NW][CKPT ? 16 * 17 : 1]; // checkpoint mode: the column checkpoints of the current 16 steps (dp_fill_stripe)
  __shared__ int progress[NW]; // per wave: rows of boundary published so far, cumulated over the wave's stripes
  constexpr int TBS = TRACE ? 64 * TBW : 1;
  __shared__ unsigned tbstage_all[NW][(TRACE ? 4 : 1) * TBS]; // the decisions of the current tile (4 steps), per wave
  const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  unsigned bid = blockIdx.x;
  if(NW > 1 && ng > 1) { // the ticket (see above): the order in which workgroups START, whatever their blockIdx
    __shared__ unsigned ticket;
    if(threadIdx.x == 0) {
      ticket = (unsigned)__hip_atomic_fetch_add(gprog + (i64)gridDim.x * NW, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    bid = ticket;
  }
  const int pos = ng > 1 ? (int)(bid / (unsigned)ng) : (int)bid;
  DpFillPair pp;
  pp.ng = ng;
  pp.team = NW * ng;                                                // waves of this pair
  pp.tw = ng > 1 ? wv * ng + (int)(bid % (unsigned)ng) : wv;        // this wave's place among them: consecutive stripes go to
                                                                    // different workgroups
  pp.gp = ng > 1 ? gprog + (i64)pos * pp.team : nullptr;
  pp.pipe_error = pipe_error;
  const i64 pair = order[pos]; // the launch's pairs in processing order (dp_batch_plan)
  const i64 a0 = off_a[pair], b0 = off_b[pair];
  const int la = (int)(off_a[pair + 1] - a0), lb = (int)(off_b[pair + 1] - b0);
  const int tl = TAILS ? tail : 0;
  pp.A = cols_a + a0;
  pp.B = cols_b + b0;
  pp.la = la;
  pp.lb = lb;
  pp.tbp = (TRACE || CKPT) ? tb + tb_off[pair] : nullptr;
  pp.bp = bnd + a0;
  pp.gop = P.go - P.ge; // cost of opening over extending, the only gap constant left in skewed coordinates
  pp.ge = P.ge;
  pp.n_stripes = (int)dp_ck_stripes(lb, C, tl);
  pp.steps = la + 63;
  pp.cstride = (int)dp_ck_stride(la);
  pp.tiles = (pp.steps + 3) / 4;
  pp.row_base = dp_ck_groups(lb, C, tl) * dp_ck_stride(la) * 2;
  pp.nck = (int)dp_ck_nck(la);
  int result = 0;
  if(la == 0 || lb == 0) { // one profile empty: a single gap run
    int n = la + lb;
    if(lane == 0 && pp.tw == 0) {
      scores[pair] = n == 0 ? 0 : -(P.go + (n - 1) * P.ge);
    }
    if constexpr(NW == 1) {
      if(fl.list) {
        dp_publish_filled(fl, pos, lane);
      }
    }
    return;
  }
  if(NW > 1) {
    if(lane == 0) {
      progress[wv] = 0;
    }
    __syncthreads();
  }
  for(int s = pp.tw; s < pp.n_stripes; s += pp.team) {
    const DpStripe st = dp_stripe(lb, C, tl, s);
    if(!TAILS || st.cs == C) {
      dp_fill_stripe<C, C, MODE, DOT4, NW, UNI>(pp, P, s, st, ring_all[wv], sbnd_all[wv], cstage_all[wv], tbstage_all[wv], progress, wv, lane, result);
    }
    else if constexpr(TAILS) {
      if(st.cs == 8) {
        dp_fill_stripe<C, 8, MODE, DOT4, NW, UNI>(pp, P, s, st, ring_all[wv], sbnd_all[wv], cstage_all[wv], tbstage_all[wv], progress, wv, lane, result);
      }
      else {
        dp_fill_stripe<C, 4, MODE, DOT4, NW, UNI>(pp, P, s, st, ring_all[wv], sbnd_all[wv], cstage_all[wv], tbstage_all[wv], progress, wv, lane, result);
      }
    }
  }
  if(lane == 0 && pp.tw == (pp.n_stripes - 1) % pp.team) {
    scores[pair] = result;
  }
  if constexpr(NW == 1) { // one wavefront has run all the pair's stripes: the pair is filled (the host passes the lists to such launches only)
    if(fl.list) {
      dp_publish_filled(fl, pos, lane);
    }
  }
}

// Tiles from a queue (round 5).  A wavefront of dp_fill_kernel works through a whole stripe -- La + 63 steps -- so a launch of few long
// pairs meets the chip's wavefront slots as uneven rounds: 512 pairs of 32 x 10 kbp are 5 120 stripe-long jobs for 4 096 slots, and
// the second half of the launch ran at one wavefront per SIMD (3 204 GCUPS against 4 153 for 4 096 pairs; a ragged launch's longest
// pairs needed launches of their own, the tiers).  Here the unit of work is a TILE: the 64-step blocks [t_begin, t_end) of one
// stripe, the lanes' state handed from tile to tile through memory (dp_fill_stripe, TILED), the stripes of a pair pipelined through
// the progress words the workgroups of a pair already use.  The grid is persistent -- as many one-wavefront workgroups as the chip
// holds -- and every wavefront takes the next tile from a queue until the queue is empty.
// Nobody waits for ever, by construction: the host lists the tiles in an order in which every tile comes after the two it depends
// on -- the tile before it in its stripe, and the tile of the stripe to its left that covers the same steps of the pair's clock
// (stripe s lags stripe s - 1 by 64 steps on that clock, so that a tile needs rows of the seam only from the left tile of its own
// number) -- and a wavefront draws tickets in that order: whatever a tile waits for was drawn earlier, so it is running on a resident
// wavefront or done, and the tile with the lowest unfinished ticket never waits.  The bounded waits and the error word stay as the
// last line of defence.  The order the host chooses (dp_tiles_build) is longest remaining chain first.
template <int C, int MODE, bool DOT4, bool UNI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DOT4 ? 5 : 4)))
dp_fill_tiles_kernel(const u64 *__restrict__ cols_a, const i64 *__restrict__ off_a, const u64 *__restrict__ cols_b,
                     const i64 *__restrict__ off_b, const int *__restrict__ order, const i64 *__restrict__ tb_off, unsigned *__restrict__ tb,
                     int2 *__restrict__ bnd, int *__restrict__ scores, int *__restrict__ pipe_error, DpParamsD P,
                     const DpTile *__restrict__ tiles, int n_tiles, const i64 *__restrict__ sync_off, i64 total_stripes, int *__restrict__ sync_words,
                     unsigned *__restrict__ state, int *__restrict__ started, int tail, int n_pos) {
  static_assert(MODE != DP_MODE_BITS, "tiles: scores and checkpoints only");
  if(started && threadIdx.x == 0) {
    __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // pairs with an empty profile have no tiles: their score is a single gap run, written here, a pair per thread of the grid.
  // (Not as a tile of its own with a `continue` in the loop below: a lane-0 store followed by `continue` inside the persistent loop
  // came out of the compiler's control-flow structurizer with the other 63 lanes running on into the tile body -- found by the
  // probe that filled the workspace with a pattern and ran the queue one ticket further at a time, tools/debug/.)
  for(int p = (int)(blockIdx.x * 64 + threadIdx.x); p < n_pos; p += (int)(gridDim.x * 64)) {
    const i64 pr = order[p];
    const int la_p = (int)(off_a[pr + 1] - off_a[pr]), lb_p = (int)(off_b[pr + 1] - off_b[pr]);
    if(la_p == 0 || lb_p == 0) {
      const int n = la_p + lb_p;
      scores[pr] = n == 0 ? 0 : -(P.go + (n - 1) * P.ge);
    }
  }
  constexpr bool CKPT = MODE == DP_MODE_CKPT;
  constexpr bool TAILS = C == 16;
  constexpr bool PACKED = DOT4 && UNI;
  __shared__ int4 ring[256];
  __shared__ int4 sbnd[PACKED ? 1 : 64];
  __shared__ int2 cstage[CKPT ? 16 * 17 : 1];
  __shared__ unsigned tbstage[1];
  __shared__ int progress[1];
  const int lane = threadIdx.x;
  int *const queue = sync_words + 2 * total_stripes; // the ticket counter, behind the progress words and the tiles-done words
  for(;;) {
    int ticket = 0;
    if(lane == 0) {
      ticket = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if(ticket >= n_tiles) {
      break;
    }
    const DpTile tile = tiles[ticket];
    const int pos = tile.pos, s = tile.s;
    DpFillPair pp;
    pp.ng = 2;           // the seam and the progress words the way several workgroups of a pair share them
    pp.team = 1 << 24;   // more than any pair has stripes: the progress words are one per stripe, not cumulated
    pp.tw = s;
    const i64 so = sync_off[pos];
    pp.gp = sync_words + so;
    pp.tdone = sync_words + total_stripes + so + s;
    pp.state = state + (so + s) * (i64)(64 * (2 * C + 4));
    pp.t_begin = tile.t_begin;
    pp.t_end = tile.t_end;
    pp.pipe_error = pipe_error;
    const i64 pair = order[pos];
    const i64 a0 = off_a[pair], b0 = off_b[pair];
    const int la = (int)(off_a[pair + 1] - a0), lb = (int)(off_b[pair + 1] - b0);
    const int tl = TAILS ? tail : 0;
    pp.A = cols_a + a0;
    pp.B = cols_b + b0;
    pp.la = la;
    pp.lb = lb;
    pp.tbp = CKPT ? tb + tb_off[pair] : nullptr;
    pp.bp = bnd + a0;
    pp.gop = P.go - P.ge;
    pp.ge = P.ge;
    pp.n_stripes = (int)dp_ck_stripes(lb, C, tl);
    pp.steps = la + 63;
    pp.cstride = (int)dp_ck_stride(la);
    pp.tiles = (pp.steps + 3) / 4;
    pp.row_base = dp_ck_groups(lb, C, tl) * dp_ck_stride(la) * 2;
    pp.nck = (int)dp_ck_nck(la);
    int result = 0;
    const DpStripe st = dp_stripe(lb, C, tl, s);
    if(!TAILS || st.cs == C) {
      dp_fill_stripe<C, C, MODE, DOT4, 1, UNI, true>(pp, P, s, st, ring, sbnd, cstage, tbstage, progress, 0, lane, result);
    }
    else if constexpr(TAILS) {
      if(st.cs == 8) {
        dp_fill_stripe<C, 8, MODE, DOT4, 1, UNI, true>(pp, P, s, st, ring, sbnd, cstage, tbstage, progress, 0, lane, result);
      }
      else {
        dp_fill_stripe<C, 4, MODE, DOT4, 1, UNI, true>(pp, P, s, st, ring, sbnd, cstage, tbstage, progress, 0, lane, result);
      }
    }
    if(s == pp.n_stripes - 1 && pp.t_end >= pp.steps) { // (uniform condition outside, the lane inside: see above)
      if(lane == 0) {
        scores[pair] = result;
      }
    }
    // the next tile's staging must not overtake this tile's last LDS reads (one wavefront, LDS operations in order: a compiler fence)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// One wavefront per pair walks the stored decisions from (la, lb) back to (0, 0).  A lane-per-pair walk pays
// one dependent HBM load per step; here the 64 lanes fetch the decisions of the next 64 cells along the current
// direction (diagonal in state H, along the row in state E, along the column in state F) in one gather, a
// ballot finds how far the run continues, and the whole run is emitted at once.  The path is written
// right-aligned into the pair's (la+lb)-byte slot: its last n_ops bytes, first op first.
template <int C>
__global__ void __launch_bounds__(64)
dp_traceback_kernel(const i64 *__restrict__ off_a, const i64 *__restrict__ off_b, const int *__restrict__ order, const i64 *__restrict__ tb_off,
                    const unsigned *__restrict__ tb, unsigned char *__restrict__ ops, int *__restrict__ n_ops) {
  const int lane = threadIdx.x;
  const i64 pair = order[blockIdx.x];
  const int la = (int)(off_a[pair + 1] - off_a[pair]), lb = (int)(off_b[pair + 1] - off_b[pair]);
  const unsigned *tbp = tb + tb_off[pair];
  unsigned char *out = ops + off_a[pair] + off_b[pair];
  constexpr int W = 64 * C;
  const i64 tiles = (la + 63 + 3) / 4;
  int i = la, j = lb, state = 0; // wave-uniform
  int at = la + lb;
  int guard = 2 * (la + lb) + 8; // every iteration but a state switch consumes a cell; a switch is followed by one
  while(i > 0 && j > 0 && guard-- > 0) {
    const int di = state != 1, dj = state != 2; // H: diagonal, E: along the row, F: along the column
    const int ci = i - lane * di, cj = j - lane * dj;
    const bool valid = ci >= 1 && cj >= 1;
    unsigned nib = 0;
    if(valid) {
      const int jj = cj - 1;
      const int s = jj / W, l = (jj % W) / C, c = jj % C;
      const int t = ci - 1 + l; // the step at which lane l was on row ci
      const unsigned word = tbp[(((i64)s * tiles + (t >> 2)) * 64 + l) * (4 * (C / 8)) + (t & 3) * (C / 8) + c / 8];
      nib = (word >> (4 * (7 - (c & 7)))) & 15u;
    }
    // a lane continues the run when its cell keeps the walk going in the same direction and state
    const unsigned keep = state == 0 ? (~nib & 2u) : (state == 1 ? (nib & 8u) : (nib & 4u));
    const unsigned long long cont = __ballot(valid && keep != 0);
    const unsigned long long vmask = __ballot(valid);
    const int run = cont == ~0ull ? 64 : __builtin_ctzll(~cont);
    if(state == 0) {
      if(run == 0) { // the cell itself is not diagonal: switch to the gap state it names, no move
        const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nib, 0);
        state = (n0 & 1u) ? 2 : 1;
        continue;
      }
      if(lane < run) {
        out[at - 1 - lane] = 0;
      }
      at -= run;
      i -= run;
      j -= run;
    }
    else {
      // the cells that extend are consumed in this state; the first one that does not is consumed too and
      // returns the walk to H
      int take = run;
      int next = state;
      if(run < 64 && ((vmask >> run) & 1ull)) {
        take = run + 1;
        next = 0;
      }
      if(lane < take) {
        out[at - 1 - lane] = (unsigned char)state; // 1 = I (state E), 2 = D (state F)
      }
      at -= take;
      if(state == 1) {
        j -= take;
      }
      else {
        i -= take;
      }
      state = next;
    }
  }
  // one profile exhausted: the rest is a single gap run
  const int rest = i + j;
  const unsigned char op = i == 0 ? 1 : 2;
  for(int k = lane; k < rest; k += 64) {
    out[at - 1 - k] = op;
  }
  at -= rest;
  if(lane == 0) {
    n_ops[pair] = la + lb - at;
  }
}

// Largest ACGT count of any column (stats[0]), largest and smallest symbol total (bytes 0-4) of any column (stats[1], and
// 2047 - stats[4] so that a zeroed word is the neutral element): what decides between the int8 and the int16 weights and whether
// every column holds the same number of symbols.  One pass over the packed columns, grid-stride, one atomic per wavefront.
__global__ void dp_column_stats_kernel(const u64 *cols, i64 n, int *stats, int *stats_min) {
  int max_base = 0, max_rows = 0, max_inv = 0;
  for(i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
    const u64 c = cols[k];
    int sum = 0;
#pragma unroll
    for(int b = 0; b < 5; ++b) {
      const int v = (int)((c >> (8 * b)) & 0xff);
      sum += v;
      if(b < 4) {
        max_base = max(max_base, v);
      }
    }
    max_rows = max(max_rows, sum);
    max_inv = max(max_inv, 2047 - sum);
  }
#pragma unroll
  for(int d = 32; d >= 1; d >>= 1) {
    max_base = max(max_base, __shfl_xor(max_base, d));
    max_rows = max(max_rows, __shfl_xor(max_rows, d));
    max_inv = max(max_inv, __shfl_xor(max_inv, d));
  }
  if((threadIdx.x & 63) == 0) {
    atomicMax(stats + 0, max_base);
    atomicMax(stats + 1, max_rows);
    atomicMax(stats_min, max_inv);
  }
}

} // namespace pm

using namespace pm;

#include "dp_batch.hpp"

namespace pm {

int dp_batch_check_params(const pm_dp_params_t *params) {
  // int16 weights: |sum_b count * sub| must stay below 2^15 (255 rows x |sub| <= 127)
  for(int k = 0; k < 25; ++k) {
    if(params->sub[k] < -127 || params->sub[k] > 127) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: substitution entries must be within [-127, 127]");
    }
  }
  if(params->gap_open < 0 || params->gap_extend < 0 || params->gap_open + params->gap_extend > 32767) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: gap penalties must be >= 0 and gap_open + gap_extend <= 32767");
  }
  return PM_OK;
}

// The error word of the fill kernel's bounded waits, cleared so that the clear has LANDED when this returns: the kernels run on
// the caller's streams, which need not wait for the null stream (hipStreamNonBlocking: pm_dp_stream_*), and a wave that finds the
// word set -- by a hipMemset that has not run yet over whatever the allocation held before -- skips its waits.  A blocking copy from
// host memory is complete on return; hipMemset is not.
int dp_clear_pipe_error(pm_dp_batch *h) {
  const int zero = 0;
  PM_HIP(hipMemcpy(h->pipe_error.p, &zero, 4, hipMemcpyHostToDevice));
  return PM_OK;
}

// The process's default options (pm_dp_set_default_options): what a batch is created with when its creator passes none.
static std::mutex g_default_options_lock;
static pm_dp_options_t g_default_options = {};

int dp_batch_init(pm_dp_batch *h, const pm_dp_params_t *params, int64_t tb_budget_bytes, int device, const pm_dp_options_t *options) {
  h->device = device;
  if(options) {
    h->opt = *options;
  }
  else {
    std::lock_guard<std::mutex> hold(g_default_options_lock);
    h->opt = g_default_options;
  }
  const pm_dp_options_t &o = h->opt;
  if(o.path_mode < 0 || o.path_mode > 2 || (o.cols_per_lane != 0 && o.cols_per_lane != 8 && o.cols_per_lane != 16) ||
     (o.waves_per_pair != 0 && o.waves_per_pair != 1 && o.waves_per_pair != 2 && o.waves_per_pair != 4 && o.waves_per_pair != 8 &&
      o.waves_per_pair != 16) ||
     o.groups_per_pair < 0 || o.band < 0 || o.band > 2 || o.walk_lanes < 0 || (o.slots != 0 && (o.slots < 2 || o.slots > 8)) || o.split < 0 ||
     o.split > 64 || o.segment_cells < 0 || o.tier_min_pairs < 0 || o.tile_steps < 0 || (o.tile_steps > 1 && o.tile_steps % 64 != 0) || o.early_walk < 0 || o.early_walk > 2) {
    return fail(PM_E_INVALID, "pm_dp_options_t: a field is out of range");
  }
  // (a batch may be initialised again for its next use -- the kept batches of dp_maf.hip: every field from the options, every time)
  h->n_slots = o.slots ? o.slots : 3; // parts of the path workspace / fill streams of a batch that needs several chunks
  h->waves_override = o.waves_per_pair;
  h->cols_per_lane = o.cols_per_lane == 8 ? 8 : 16;
  h->cols_forced = o.cols_per_lane != 0;
  h->mode_auto = o.path_mode == 0; // else fixed; chosen per batch in dp_batch_plan
  h->ckpt = o.path_mode != 1;
  h->walk_lanes = o.walk_lanes && dp_walk_lanes_ok(h->cols_per_lane, o.walk_lanes) ? o.walk_lanes : 0;
  memcpy(h->params.sub, params->sub, sizeof h->params.sub);
  h->params.go = params->gap_open;
  h->params.ge = params->gap_extend;
  // int8 path: counts of A <= 127 and |sum_b B[j][b] * sub[a][b]| <= (rows of B's column) * max|sub[a][.]| <= 127 for a in ACGT
  h->max_sub_acgt = 0;
  h->max_sub_all = 0;
  for(int a = 0; a < 5; ++a) {
    for(int b = 0; b < 5; ++b) {
      const int v = std::abs(params->sub[a * 5 + b]);
      h->max_sub_all = std::max(h->max_sub_all, v);
      if(a < 4) {
        h->max_sub_acgt = std::max(h->max_sub_acgt, v);
      }
    }
  }
  if(tb_budget_bytes <= 0) {
    tb_budget_bytes = dp_default_budget_bytes();
  }
  h->tb_budget_bytes = tb_budget_bytes;
  PM_TRY(h->pipe_error.alloc(4));
  PM_TRY(dp_clear_pipe_error(h));
  PM_TRY(h->stats.alloc(32));
  return PM_OK;
}

static int grow(DevBuf &b, size_t bytes) {
  if(b.p && b.bytes >= std::max<size_t>(bytes, 16)) {
    return PM_OK;
  }
  return b.alloc(bytes);
}

int dp_batch_reserve(pm_dp_batch *h, i64 cap_pairs, i64 cap_a, i64 cap_b) {
  PM_TRY(grow(h->cols_a, (size_t)cap_a * 8));
  PM_TRY(grow(h->cols_b, (size_t)cap_b * 8));
  PM_TRY(grow(h->d_off_a, (size_t)(cap_pairs + 1) * 8));
  PM_TRY(grow(h->d_off_b, (size_t)(cap_pairs + 1) * 8));
  PM_TRY(grow(h->d_tb_off, (size_t)(cap_pairs + 1) * 8));
  PM_TRY(grow(h->bnd, (size_t)cap_a * 8));
  PM_TRY(grow(h->scores, (size_t)cap_pairs * 4));
  PM_TRY(grow(h->n_ops, (size_t)cap_pairs * 4));
  PM_TRY(grow(h->ops, (size_t)(cap_a + cap_b)));
  return PM_OK;
}

int dp_batch_load(pm_dp_batch *h, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                  int64_t n_pairs, hipStream_t stream) {
  const int64_t a0 = off_a[0], b0 = off_b[0];
  for(int64_t k = 0; k < n_pairs; ++k) {
    if(off_a[k + 1] < off_a[k] || off_b[k + 1] < off_b[k] || off_a[k + 1] - off_a[k] > (1 << 24) || off_b[k + 1] - off_b[k] > (1 << 24)) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: bad profile length");
    }
    if((off_a[k + 1] - off_a[k] + off_b[k + 1] - off_b[k]) * (int64_t)h->params.ge >= (1 << 28)) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: (La + Lb) * gap_extend must stay below 2^28");
    }
  }
  h->n_pairs = n_pairs;
  h->seg_first.clear();
  h->seg_events_armed = false;
  h->off_a.resize((size_t)n_pairs + 1);
  h->off_b.resize((size_t)n_pairs + 1);
  for(int64_t k = 0; k <= n_pairs; ++k) {
    h->off_a[(size_t)k] = off_a[k] - a0;
    h->off_b[(size_t)k] = off_b[k] - b0;
  }
  h->total_a = h->off_a[(size_t)n_pairs];
  h->total_b = h->off_b[(size_t)n_pairs];
  if((h->total_a > 0 && !cols_a) || (h->total_b > 0 && !cols_b)) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: null columns");
  }
  PM_TRY(dp_batch_reserve(h, n_pairs, h->total_a, h->total_b));
  // the offsets go through pinned staging when the batch has it (so that the copy is asynchronous), else straight from the vectors
  const i64 *src_a = h->off_a.data(), *src_b = h->off_b.data();
  if(h->pinned && h->pinned_bytes >= (size_t)(4 * (n_pairs + 1)) * 8 + 64 + 32) {
    i64 *pa = (i64 *)h->pinned, *pb = pa + (n_pairs + 1);
    memcpy(pa, src_a, (size_t)(n_pairs + 1) * 8);
    memcpy(pb, src_b, (size_t)(n_pairs + 1) * 8);
    src_a = pa;
    src_b = pb;
  }
  PM_HIP(hipMemcpyAsync(h->d_off_a.p, src_a, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
  PM_HIP(hipMemcpyAsync(h->d_off_b.p, src_b, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
  if(h->total_a > 0) {
    // (hipMemcpyDefault: pm_dp_align_maf hands over columns that are already in device memory)
    PM_HIP(hipMemcpyAsync(h->cols_a.p, cols_a + a0 * 8, (size_t)h->total_a * 8, hipMemcpyDefault, stream));
  }
  if(h->total_b > 0) {
    PM_HIP(hipMemcpyAsync(h->cols_b.p, cols_b + b0 * 8, (size_t)h->total_b * 8, hipMemcpyDefault, stream));
  }
  // the ranges of the uploaded columns, found on the device (four 4-byte words back)
  PM_HIP(hipMemsetAsync(h->stats.p, 0, 32, stream));
  if(h->total_a > 0) {
    dp_column_stats_kernel<<<1024, 256, 0, stream>>>((const u64 *)h->cols_a.p, h->total_a, (int *)h->stats.p, (int *)h->stats.p + 4);
  }
  if(h->total_b > 0) {
    dp_column_stats_kernel<<<1024, 256, 0, stream>>>((const u64 *)h->cols_b.p, h->total_b, (int *)h->stats.p + 2, (int *)h->stats.p + 5);
  }
  PM_HIP(hipGetLastError());
  int *dst = h->host_stats;
  if(h->pinned) {
    dst = (int *)((char *)h->pinned + h->pinned_bytes - 32);
  }
  PM_HIP(hipMemcpyAsync(dst, h->stats.p, 32, hipMemcpyDeviceToHost, stream));
  return PM_OK;
}

int dp_batch_load_segments(pm_dp_batch *h, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                           int64_t n_pairs, int segments, int *stats_first, hipStream_t stream) {
  const int64_t a0 = off_a[0], b0 = off_b[0];
  return dp_batch_load_segments_from(h, off_a, off_b, n_pairs, segments, stats_first, stream,
                                     [&](int side, i64, i64, i64 c0, i64 c1, hipStream_t st) {
                                       if(side == 0) {
                                         PM_HIP(hipMemcpyAsync((char *)h->cols_a.p + c0 * 8, cols_a + (a0 + c0) * 8, (size_t)(c1 - c0) * 8,
                                                               hipMemcpyHostToDevice, st));
                                       }
                                       else {
                                         PM_HIP(hipMemcpyAsync((char *)h->cols_b.p + c0 * 8, cols_b + (b0 + c0) * 8, (size_t)(c1 - c0) * 8,
                                                               hipMemcpyHostToDevice, st));
                                       }
                                       return (int)PM_OK;
                                     },
                                     cols_a != nullptr, cols_b != nullptr);
}

// The same with the columns of a segment produced by `fill(side, first pair, end pair, first column, end column, stream)` -- a copy
// from host columns above, or row texts copied and packed on the device (dp_maf.hip).
int dp_batch_load_segments_from(pm_dp_batch *h, const int64_t *off_a, const int64_t *off_b, int64_t n_pairs, int segments, int *stats_first,
                                hipStream_t stream, const std::function<int(int, i64, i64, i64, i64, hipStream_t)> &fill, bool have_a,
                                bool have_b) {
  const int64_t a0 = off_a[0], b0 = off_b[0];
  for(int64_t k = 0; k < n_pairs; ++k) {
    if(off_a[k + 1] < off_a[k] || off_b[k + 1] < off_b[k] || off_a[k + 1] - off_a[k] > (1 << 24) || off_b[k + 1] - off_b[k] > (1 << 24)) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: bad profile length");
    }
    if((off_a[k + 1] - off_a[k] + off_b[k + 1] - off_b[k]) * (int64_t)h->params.ge >= (1 << 28)) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: (La + Lb) * gap_extend must stay below 2^28");
    }
  }
  h->n_pairs = n_pairs;
  h->off_a.resize((size_t)n_pairs + 1);
  h->off_b.resize((size_t)n_pairs + 1);
  for(int64_t k = 0; k <= n_pairs; ++k) {
    h->off_a[(size_t)k] = off_a[k] - a0;
    h->off_b[(size_t)k] = off_b[k] - b0;
  }
  h->total_a = h->off_a[(size_t)n_pairs];
  h->total_b = h->off_b[(size_t)n_pairs];
  if((h->total_a > 0 && !have_a) || (h->total_b > 0 && !have_b)) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: null columns");
  }
  PM_TRY(dp_batch_reserve(h, n_pairs, h->total_a, h->total_b));
  if(!h->pinned || h->pinned_bytes < (size_t)(4 * (n_pairs + 1)) * 8 + 64 + 32) { // offsets, workspace offsets, order, statistics
    if(h->pinned) {
      (void)hipHostFree(h->pinned);
      h->pinned = nullptr;
    }
    h->pinned_bytes = (size_t)(4 * (n_pairs + 1)) * 8 + 64 + 32;
    PM_HIP(hipHostMalloc(&h->pinned, h->pinned_bytes, hipHostMallocDefault));
  }
  i64 *pa = (i64 *)h->pinned, *pb = pa + (n_pairs + 1);
  memcpy(pa, h->off_a.data(), (size_t)(n_pairs + 1) * 8);
  memcpy(pb, h->off_b.data(), (size_t)(n_pairs + 1) * 8);
  PM_HIP(hipMemcpyAsync(h->d_off_a.p, pa, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
  PM_HIP(hipMemcpyAsync(h->d_off_b.p, pb, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
  PM_HIP(hipMemsetAsync(h->stats.p, 0, 32, stream));
  // segments of about equal numbers of columns (A + B), cut at pair boundaries -- but no more of them than leave every fill
  // launch about 5e9 cells: a launch of a few thousand short pairs cannot fill the chip (2.4 wavefronts per SIMD at 2 500 pairs),
  // and four such launches cost more than the upload they hide (10 000 pairs of 2 x 1 kbp: 5.1 ms in four segments, 4.6 in two)
  double cells = 0;
  {
    for(int64_t k = 0; k < n_pairs; ++k) {
      cells += (double)(h->off_a[(size_t)k + 1] - h->off_a[(size_t)k]) * (double)(h->off_b[(size_t)k + 1] - h->off_b[(size_t)k]);
    }
    const double per_segment = h->opt.segment_cells > 0 ? (double)h->opt.segment_cells : 5e9; // (tests cut tiny batches into many segments)
    segments = (int)std::min<int64_t>(segments, std::max<int64_t>(1, (int64_t)(cells / per_segment)));
    // A segment is a launch of its own (dp_batch_plan).  Pairs of one length balance themselves; a RAGGED launch -- its longest pair
    // more than twice the mean -- needs pairs in the tens of thousands to keep the chip evenly loaded beside its few long ones (the
    // ragged 100 k-pair batch from host memory: 131 ms in 4 segments of 25 000 pairs, 179 in 8; profiles/r04_stream.txt)
    if(h->opt.segment_cells <= 0 && n_pairs > 0) {
      double longest = 0;
      for(int64_t k = 0; k < n_pairs; ++k) {
        longest = std::max(longest, (double)(h->off_a[(size_t)k + 1] - h->off_a[(size_t)k]) * (double)(h->off_b[(size_t)k + 1] - h->off_b[(size_t)k]));
      }
      if(longest > 2.0 * cells / (double)n_pairs) {
        segments = (int)std::min<int64_t>(segments, std::max<int64_t>(1, n_pairs / 25000));
      }
      // ... and a launch of few LONG pairs should bring a few rounds' worth of stripes for the chip's 4 096 resident wavefronts
      // (4 096 pairs of 32 x 10 kbp, ten stripes each: 120 ms in four segments of 1 024 pairs, profiles/r04_stream.txt)
      // -- and a launch of a small batch at least one: 512 pairs of 32 x 10 kbp in four launches of 128 pairs, 1 280 stripes each,
      // took 28.9 ms from host memory against the resident step's 15.6.
      {
        int64_t stripes = 0;
        for(int64_t k = 0; k < n_pairs; ++k) {
          stripes += (h->off_b[(size_t)k + 1] - h->off_b[(size_t)k] + 1023) / 1024;
        }
        segments = (int)std::min<int64_t>(segments, std::max<int64_t>(1, stripes / (cells >= 1e11 ? 16384 : 4096)));
      }
    }
  }
  segments = (int)std::max<int64_t>(1, std::min<int64_t>(segments, std::max<int64_t>(n_pairs, 1)));
  h->seg_first.assign(1, 0);
  {
    const i64 total = h->total_a + h->total_b;
    // A large batch (one that runs a chunk per segment, dp_batch_plan_layout) gets a short segment in front, an eighth of the others:
    // nothing can run before the first segment is up, and an eighth of a segment is still a launch that fills the chip
    // (a round of stripes for the 4 096 resident wavefronts and a thousand pairs at least -- a launch of a few hundred long pairs is a
    // poor one: 4 096 pairs of 32 x 10 kbp with 410 in front 143 ms, without 110 -- and at most a quarter of a regular segment)
    if(segments >= 2 && cells >= 1e11 && h->opt.segment_cells <= 0) {
      const i64 want = total / ((i64)segments * 8);
      i64 k = 0, stripes = 0;
      while(k < n_pairs && (h->off_a[(size_t)k] + h->off_b[(size_t)k] < want || stripes < 4096 || k < 1024)) {
        stripes += (h->off_b[(size_t)k + 1] - h->off_b[(size_t)k] + 1023) / 1024;
        ++k;
      }
      if(k < n_pairs && h->off_a[(size_t)k] + h->off_b[(size_t)k] <= total / ((i64)segments * 4)) {
        h->seg_first.push_back(k);
      }
    }
    for(int sgi = 1; sgi < segments; ++sgi) {
      const i64 want = total * sgi / segments;
      i64 k = h->seg_first.back();
      while(k < n_pairs && h->off_a[(size_t)k] + h->off_b[(size_t)k] < want) {
        ++k;
      }
      if(k > h->seg_first.back() && k < n_pairs) {
        h->seg_first.push_back(k);
      }
    }
    h->seg_first.push_back(n_pairs);
  }
  const size_t nseg = h->seg_first.size() - 1;
  while(h->ev_seg.size() < nseg) {
    hipEvent_t e = nullptr;
    PM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->ev_seg.push_back(e);
  }
  for(size_t sgi = 0; sgi < nseg; ++sgi) {
    const i64 lo = h->seg_first[sgi], hi = h->seg_first[sgi + 1];
    const i64 sa0 = h->off_a[(size_t)lo], sa1 = h->off_a[(size_t)hi], sb0 = h->off_b[(size_t)lo], sb1 = h->off_b[(size_t)hi];
    if(sa1 > sa0) {
      PM_TRY(fill(0, lo, hi, sa0, sa1, stream));
      dp_column_stats_kernel<<<256, 256, 0, stream>>>((const u64 *)h->cols_a.p + sa0, sa1 - sa0, (int *)h->stats.p, (int *)h->stats.p + 4);
    }
    if(sb1 > sb0) {
      PM_TRY(fill(1, lo, hi, sb0, sb1, stream));
      dp_column_stats_kernel<<<256, 256, 0, stream>>>((const u64 *)h->cols_b.p + sb0, sb1 - sb0, (int *)h->stats.p + 2, (int *)h->stats.p + 5);
    }
    PM_HIP(hipGetLastError());
    if(sgi == 0 && stats_first) {
      PM_HIP(hipMemcpyAsync(stats_first, h->stats.p, 32, hipMemcpyDeviceToHost, stream));
    }
    if(sgi + 1 == nseg) {
      PM_HIP(hipMemcpyAsync((char *)h->pinned + h->pinned_bytes - 32, h->stats.p, 32, hipMemcpyDeviceToHost, stream));
    }
    PM_HIP(hipEventRecord(h->ev_seg[sgi], stream));
  }
  h->seg_events_armed = true;
  return PM_OK;
}

const int *dp_batch_final_stats(const pm_dp_batch *h) { // where the load's last statistics copy lands
  return h->pinned ? (const int *)((const char *)h->pinned + h->pinned_bytes - 32) : h->host_stats;
}

int dp_batch_plan(pm_dp_batch *h, hipStream_t stream) {
  return dp_batch_plan_with(h, dp_batch_final_stats(h), stream);
}

// lanes per pair of the checkpoint walk for the launch of the n pairs at positions [first, first + n): as few as still give the launch
// about two wavefronts per SIMD (1 024 SIMDs) -- but at least as many as leave a lane four columns of a block: with eight columns
// per lane a block's decisions are 4-byte words, the workgroup's LDS 24 KB, and a CU holds six wavefronts of a kernel that lives on
// hiding latency (the ragged 100 k-pair batch's walk, alone on the chip: 18.8 ms with 8 lanes per pair, 15.0 with 16).  And 32 lanes
// (two columns a lane: 6 KB of LDS and 96 VGPRs, five wavefronts per SIMD where 16 lanes leave three) for LONG pairs: their walk is
// a chain of a hundred blocks and more, bound by latency, and more resident wavefronts hide it, although a block then costs 1.46 x
// the instructions (95 steps of 2 cells against 79 of 4).  Measured (profiles/r04_walk_lanes.txt; 8 / 16 / 32 lanes): 12 500 pairs of
// 8 x 4 096 44.5 / 42.3 / 41.4 ms a step; the ragged eighth 19.8 / 15.0 / 14.5; 10 000 pairs of 2 x 1 000, whose walk is a third of the
// step and bound by issue, 3.40 / 3.13 / 3.23: 32 lanes from a mean La + Lb of 3 000 columns on.
static int dp_walk_lanes_rule(const pm_dp_batch *h, i64 n, double mean_columns) { // mean_columns: La + Lb averaged over the launch's pairs
  int lpp = h->walk_lanes;
  if(lpp == 0) {
    lpp = 2;
    while(!dp_walk_lanes_ok(h->cols_per_lane, lpp)) {
      lpp *= 2;
    }
    while(lpp < 32 && h->cols_per_lane * DP_CK_W / lpp > 4 && dp_walk_lanes_ok(h->cols_per_lane, lpp * 2)) {
      lpp *= 2;
    }
    while(lpp < 32 && dp_walk_lanes_ok(h->cols_per_lane, lpp * 2) && (n * lpp <= 2 * 1024 * 64 || mean_columns >= 3000.0)) {
      lpp *= 2;
    }
  }
  return lpp;
}
static int dp_walk_lanes_for(const pm_dp_batch *h, i64 first, i64 n) { // positions [first, first + n) of the processing order
  double columns = 0;
  for(i64 q = first; q < first + n; ++q) {
    const i64 k = h->order[(size_t)q];
    columns += (double)(h->off_a[k + 1] - h->off_a[k] + h->off_b[k + 1] - h->off_b[k]);
  }
  return dp_walk_lanes_rule(h, n, n > 0 ? columns / (double)n : 0.0);
}

// Does the fill launch of the pairs at positions [first, first + n) take its work from a queue of tiles (dp_fill_tiles_kernel), and in
// tiles of how many steps?  0: no.  By itself: a launch whose stripe-long jobs are neither few enough to be resident all at once
// (the waves-per-pair choices of dp_launch_fill) nor so many that the rounds even out -- between half and eight times the chip's
// 4 096 wavefront slots -- and long enough to be cut (the longest pair two tiles or more).  From the lengths and the options alone,
// so that the layout (no tiers for such a launch) and the launch agree.
static int dp_tiles_rule(const pm_dp_batch *h, i64 first, i64 n, bool bits) {
  if(h->opt.tile_steps == 1 || bits || h->cols_per_lane != 16 || h->waves_override != 0 || n <= 0) {
    return 0;
  }
  if(h->opt.tile_steps >= 64) {
    return h->opt.tile_steps;
  }
  if(!h->seg_first.empty()) {
    // a batch that is arriving from the host (dp_stream.hip): the list would be made, and its buffers allocated, in the middle of the
    // pipeline of uploads and launches (measured: the headline batch from pinned memory 328 -> 437 ms with a tiled first chunk)
    return 0;
  }
  const i64 slots = 4096;
  i64 jobs = 0, max_steps = 0, min_steps = (i64)1 << 40;
  for(i64 q = first; q < first + n; ++q) {
    const i64 k = h->order[(size_t)q];
    const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
    jobs += la > 0 ? dp_ck_stripes(lb, h->cols_per_lane, h->tail) : 0;
    max_steps = std::max(max_steps, la > 0 && lb > 0 ? la + 63 : 0);
    min_steps = std::min(min_steps, la + 63);
  }
  // ... and a launch of LIKE pairs: in a ragged one the tiers (launches of the longest pairs with their own early walks) are worth
  // more than the balance (measured: one GPU's eighth of the ragged batch 3 440 GCUPS from the queue, 3 970 with tiers)
  // Tiles of a quarter of the longest stripe, 1 024 steps at least: a tile costs its hand-over and, more, the waits of wavefronts that
  // now depend on each other (measured, profiles/r05_dp_tiles.txt: 512 pairs of 32 x 10 kbp 3 295 GCUPS without tiles, 3 217 / 3 418 /
  // 3 433 / 3 256 in tiles of 512 / 1 024 / 2 048 / 4 096 steps; 1 024 pairs 3 463 without, 3 383 / 3 660 / 3 790 / 3 848)
  const i64 T = std::max<i64>(1024, ((max_steps / 4 + 63) / 64) * 64);
  return max_steps >= 2 * T && 2 * min_steps >= max_steps && jobs >= slots / 2 && jobs <= 8 * slots ? (int)T : 0;
}

// The tiles of that launch, in the order the wavefronts draw them: longest remaining chain first.  On the pair's clock stripe s lags
// stripe s - 1 by one 64-step block; tile k of a stripe is what the stripe does during blocks [k Tb, (k + 1) Tb) of that clock, so a
// tile depends on the tile before it in its stripe and on tile k of the stripe to its left, nothing else.  A tile's rank is the
// number of clock tiles its pair still has from it on; the list is sorted by falling rank, then by position, then by stripe -- an
// order in which every tile comes after those it depends on (the tile before it has a higher rank; the left tile the same rank, the
// same position and a lower stripe number).
static int dp_tiles_build(pm_dp_batch *h, i64 first, i64 n, int T, DpTilePlan &plan) {
  struct Ranked {
    DpTile t;
    int rank;
  };
  std::vector<Ranked> all;
  std::vector<i64> sync_off((size_t)n);
  const i64 Tb = T / 64;
  i64 total_stripes = 0;
  for(i64 pos = 0; pos < n; ++pos) {
    const i64 k = h->order[(size_t)(first + pos)];
    const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
    sync_off[(size_t)pos] = total_stripes;
    if(la == 0 || lb == 0) {
      continue; // no tiles: the kernel's first lines write the score of the single gap run
    }
    const i64 S = dp_ck_stripes(lb, h->cols_per_lane, h->tail), nblk = (la + 63 + 63) / 64;
    const i64 K = (nblk + S - 1 + Tb - 1) / Tb;
    for(i64 s = 0; s < S; ++s) {
      for(i64 kk = 0; kk < K; ++kk) {
        const i64 b0 = std::max<i64>(kk * Tb - s, 0), b1 = std::min<i64>((kk + 1) * Tb - s, nblk);
        if(b1 > b0) {
          all.push_back(Ranked{DpTile{(int)pos, (int)s, (int)(b0 * 64), (int)(b1 * 64)}, (int)(K - kk)});
        }
      }
    }
    total_stripes += S;
  }
  if(all.size() >= ((size_t)1 << 30)) {
    return fail(PM_E_INVALID, "pm_dp_batch: too many tiles in one launch");
  }
  std::stable_sort(all.begin(), all.end(), [](const Ranked &x, const Ranked &y) { return x.rank > y.rank; });
  std::vector<DpTile> tiles(all.size());
  for(size_t i = 0; i < all.size(); ++i) {
    tiles[i] = all[i].t;
  }
  plan.first = first;
  plan.n = n;
  plan.tile_steps = T;
  plan.n_tiles = (i64)tiles.size();
  plan.total_stripes = total_stripes;
  PM_TRY(plan.tiles.alloc(tiles.size() * sizeof(DpTile)));
  PM_TRY(plan.sync_off.alloc((size_t)n * 8));
  PM_TRY(plan.sync_words.alloc((size_t)(2 * total_stripes + 1) * 4));
  PM_TRY(plan.state.alloc((size_t)total_stripes * 64 * (size_t)(2 * h->cols_per_lane + 4) * 4));
  PM_HIP(hipMemcpy(plan.tiles.p, tiles.data(), tiles.size() * sizeof(DpTile), hipMemcpyHostToDevice));
  PM_HIP(hipMemcpy(plan.sync_off.p, sync_off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  return PM_OK;
}

int dp_batch_plan_with(pm_dp_batch *h, const int *st, hipStream_t stream) {
  PM_TRY(dp_batch_plan_variant(h, st));
  return dp_batch_plan_layout(h, stream);
}

// What the column statistics decide: whether the batch can be run at all, and the kernel's arithmetic (int8 or int16 weights, the
// uniform-depth form).  Nothing of the layout below depends on it -- the host-fed engine lays a batch out while its first segment is
// still on the way and comes here when the statistics are (dp_stream.hip).
int dp_batch_plan_variant(pm_dp_batch *h, const int *st) {
  const i64 n_pairs = h->n_pairs;
  const int max_a = st[0], max_colsum_b = st[3];
  if((int64_t)max_colsum_b * h->max_sub_all > 32767) { // the column weights are int16 lanes of v_dot2_i32_i16
    return fail(PM_E_INVALID, "pm_dp_batch_create: (rows of a column of B) x max|sub| exceeds 32767");
  }
  // every score the kernel carries must stay within +-2^28 (the skewed H~ = H + (i + j) * gap_extend as well, and the
  // decision bits are signs of 32-bit differences against the -2^29 sentinel): bound the largest magnitude any cell can
  // reach from the column statistics and refuse the batch otherwise
  {
    const int64_t max_colsum_a = st[1];
    int64_t worst = 0;
    for(int64_t k = 0; k < n_pairs; ++k) {
      const int64_t la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      const int64_t m = max_colsum_a * max_colsum_b * h->max_sub_all * std::min(la, lb) + (la + lb) * (int64_t)h->params.ge +
                        2 * (int64_t)h->params.go;
      worst = std::max(worst, m);
    }
    if(worst >= ((int64_t)1 << 28)) {
      return fail(PM_E_INVALID, "pm_dp_batch_create: scores could leave +-2^28 (rows(A) x rows(B) x max|sub| x min(La, Lb) + (La + Lb) x gap_extend "
                                "+ 2 x gap_open too large)");
    }
  }
  h->dot4 = max_a <= 127 && max_colsum_b * h->max_sub_acgt <= 127;
  // uniform depth: every column of A holds the same number of symbols (bytes 0-4), as the rows of a MAF block without N's do;
  // the gap row of the score then folds into the base weights (w[a] - w[gap]) and a per-column constant (dp_column_weights)
  {
    const int min_colsum_a = 2047 - st[4];
    int max_sub_gap = 0; // max |sub[gap][b]|
    for(int b = 0; b < 5; ++b) {
      max_sub_gap = std::max(max_sub_gap, std::abs(h->params.sub[4 * 5 + b]));
    }
    const bool uniform = h->total_a > 0 && st[1] == min_colsum_a && st[1] > 0;
    const int64_t diff_bound = (int64_t)max_colsum_b * (h->max_sub_acgt + max_sub_gap); // |w[a] - w[gap]|
    h->uni = uniform && diff_bound <= 32767 && (int64_t)st[1] * max_colsum_b * max_sub_gap < (1 << 28);
    if(h->uni) {
      // the differences are what the int8 lanes would hold: when they do not fit but the plain weights do, the general int8
      // kernel (dot4 + dot2) beats the uniform int16 one (2 x dot2 + add)
      const bool uni_dot4 = max_a <= 127 && diff_bound <= 127;
      if(!uni_dot4 && h->dot4) {
        h->uni = false;
      }
      else {
        h->dot4 = uni_dot4;
      }
    }
    if(h->opt.no_uniform_depth && h->uni) {
      h->uni = false;
      h->dot4 = max_a <= 127 && max_colsum_b * h->max_sub_acgt <= 127;
    }
    h->params.rows_a = h->uni ? st[1] : 0;
  }
  if(h->opt.int16_weights) {
    h->dot4 = false;
  }
  return PM_OK;
}

// Everything that follows from the pairs' lengths and the options: columns per lane, where the paths come from, the processing
// order, chunks and their workspace, tiers, the band.
int dp_batch_plan_layout(pm_dp_batch *h, hipStream_t stream) {
  const i64 n_pairs = h->n_pairs;
  h->tile_plans.clear(); // (hipFree waits for the device: no launch of the layout before still reads them)
  // Columns of B per lane: 16, or 8 for a batch of a few hundred pairs at most whose profiles fit one 1 024-column stripe --
  // two stripes of 512 then, so twice the wavefronts a pair can keep busy, each with half the work per step (256 pairs of
  // 2 x 1 kbp: 0.60 -> 0.46 ms).  Longer pairs already have stripes to run side by side, and narrower stripes only make the
  // checkpoint walk's blocks smaller and its chain longer -- which stops mattering once the walk reads its blocks from the
  // band: batches with fewer pairs than the chip has CUs take 8 columns whatever their length (128 pairs of 32 x 10 kbp: fill
  // 7.7 -> 6.9 ms, walk 0.85 -> 1.16 ms; from ~250 pairs up 16 columns win again).
  // the offset tables go through the batch's pinned staging when it has one (dp_stream.hip)
  const bool staged = h->pinned && h->pinned_bytes >= (size_t)(4 * (n_pairs + 1)) * 8 + 64 + 32;
  const int band_env = h->opt.band == 1 ? 0 : (h->opt.band == 2 ? 1 : -1); // 0: never, 1: whenever it fits, -1: chosen below
  if(!h->cols_forced) {
    i64 max_lb = 0;
    for(i64 k = 0; k < n_pairs; ++k) {
      max_lb = std::max(max_lb, h->off_b[k + 1] - h->off_b[k]);
    }
    h->cols_per_lane = (n_pairs <= 512 && max_lb <= 1024) || (n_pairs <= 200 && !staged && band_env != 0) ? 8 : 16;
    if(h->walk_lanes && !dp_walk_lanes_ok(h->cols_per_lane, h->walk_lanes)) {
      h->walk_lanes = 0;
    }
  }
  // The band of the walk (dp_internal.hpp) pays while the walk alone would leave the chip mostly idle: up to one wavefront per
  // SIMD (measured: 2 048 pairs of 8 x 4 kbp 2.69 -> 1.93 ms, 4 096 pairs 2.74 -> 3.56 ms; 1 000 pairs of 2 x 1 kbp 0.47 -> 0.28 ms,
  // 4 000 pairs 0.59 -> 0.90 ms)
  auto band_pays = [](i64 n, int lanes) { return n * lanes <= 1024 * 64; };
  // Paths from checkpoints or from stored decision bits?  The checkpoint fill is 2.3x faster per cell but its walk is a chain
  // of blocks with ~15 us of latency each, whatever the batch size; a small batch is better off storing the bits.  Measured on
  // MI355X (profiles/r02_dp_mode_sweep.txt): bits 2.3 T cells/s, checkpoint fill 5.2 T cells/s, a lone pair's fill 0.36 us per
  // step (1.55x that with bits).  Checkpoints when what the fill saves exceeds the walk's chain.
  if(h->mode_auto) {
    double cells = 0, chain_blocks = 0, lone_fill_s = 0;
    const i64 W = 64 * h->cols_per_lane;
    for(i64 k = 0; k < n_pairs; ++k) {
      const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      cells += (double)la * (double)lb;
      chain_blocks = std::max(chain_blocks, (double)(la / DP_CK_R + lb / (DP_CK_W * h->cols_per_lane) + 1));
      const i64 stripes = (lb + W - 1) / W;
      i64 nw = 1;
      while(nw < 16 && nw * 2 <= stripes + (nw >= 8 ? 6 : 0) && n_pairs * nw < 4096) {
        nw *= 2;
      }
      // a pair's waves share one CU: beyond one wave per SIMD they take turns
      const double share = std::max(1.0, (double)std::min<i64>(nw, stripes) / 4.0);
      lone_fill_s = std::max(lone_fill_s, (double)((stripes + nw - 1) / nw) * (double)(la + 63) * 0.36e-6 * share);
    }
    const double fill_saved_s = std::max(cells * (1.0 / 2.3e12 - 1.0 / 5.2e12), 0.55 * lone_fill_s);
    // a block of the chain: ~15 us recomputed; ~4 us read back from the band, whose kernel costs a launch more (~30 us)
    const bool band_likely = !staged && band_env != 0 && (band_env > 0 || band_pays(n_pairs, dp_walk_lanes_rule(h, n_pairs, n_pairs > 0 ? (double)(h->total_a + h->total_b) / (double)n_pairs : 0.0)));
    h->ckpt = fill_saved_s > (band_likely ? 30e-6 + chain_blocks * 4e-6 : chain_blocks * 15e-6);
  }
  // the narrow last stripes (dp_internal.hpp): wherever the path comes from checkpoints and a lane has 16 columns
  h->tail = h->ckpt && h->cols_per_lane == 16 && !h->opt.full_stripes;
  auto need_words = [&](i64 k) {
    i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
    return h->ckpt ? dp_ck_words(la, lb, h->cols_per_lane, h->tail) : dp_tb_words(la, lb, h->cols_per_lane);
  };
  // processing order: a wavefront works through its pair's stripes one after the other, so a launch lasts at least as long as
  // its longest pair; pairs are therefore taken longest first (stripes x steps, ties in input order), which also puts pairs of
  // like length into the same chunk, where the waves-per-pair choice of dp_launch_fill can split the long ones.  A batch
  // whose columns arrive in segments (dp_stream.hip) is cut into chunks in the input order -- a chunk never reaches across a
  // segment boundary, so its fill kernel waits for one segment, and its results are one range of the caller's arrays -- and
  // ordered longest first INSIDE every chunk (below).  (It kept the input order altogether until round 4: the ragged 100 k-pair
  // batch from host memory ran at 1 960 GCUPS against the resident batch's 4 950, profiles/r04_stream.txt.)
  const bool segmented = !h->seg_first.empty();
  h->order.resize((size_t)n_pairs);
  for(i64 k = 0; k < n_pairs; ++k) {
    h->order[(size_t)k] = (int)k;
  }
  std::vector<i64> cost;
  if(!h->opt.keep_order) {
    cost.resize((size_t)n_pairs);
    for(i64 k = 0; k < n_pairs; ++k) {
      const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      cost[(size_t)k] = dp_fill_cost(la, lb, h->cols_per_lane, h->tail);
    }
    if(!segmented) {
      std::stable_sort(h->order.begin(), h->order.end(), [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
    }
  }
  i64 total_words = 0;
  h->cells = 0;
  for(i64 k = 0; k < n_pairs; ++k) {
    total_words += need_words(k);
    h->cells += (h->off_a[k + 1] - h->off_a[k]) * (h->off_b[k + 1] - h->off_b[k]);
  }
  // chunks: consecutive pairs whose workspace fits the budget.  A batch that fits is one chunk; otherwise the budget is cut
  // in n_slots parts that consecutive chunks use in turn, so that the path kernel of one chunk can run beside the fill kernels
  // of the next, and those beside each other where one drains and the next fills in (dp_run)
  // the workspace is allocated to what the chunks need; if the device cannot give that much (other allocations beside this
  // batch) the budget is halved and the batch cut into more chunks, down to 256 MiB
  // A batch that fits is one chunk -- one fill launch, one path launch -- unless its path kernel is worth hiding: a batch of
  // at least 8 192 pairs whose fill takes 20 ms or more and whose walk would add 8 % or more to it is cut in two chunks, each with a
  // part of the workspace of its own, the path kernel of the first beside the fill kernel of the second (nothing is lost at the
  // cut: dp_gate_kernel).  Measured on one MI355X: 12 500 pairs of 8 x 4 096 (one GPU's eighth of the headline batch) 43.9 -> 42.2 ms
  // (profiles/r04_dp_shares.txt); 25 000 pairs +0.4 %; the ragged 100 k-pair batch 109.6 -> 108.1 ms.  Not for smaller work: the path
  // kernel is real work for the same SIMDs, and launches of a few milliseconds lose more at their ends than the overlap gains (10 k
  // pairs of 2 x 1 kbp: 2.88 ms whole, 3.28 in two; more than two chunks: 12 500 pairs 44.7 ms in four).  opt.split fixes the number.
  i64 split = 1;
  bool uneven = false;
  double first_share = 0.5;
  if(h->opt.split >= 1) {
    split = !staged ? std::min<i64>(h->opt.split, std::max<i64>(n_pairs, 1)) : 1;
  }
  else if(h->ckpt && !staged && h->seg_first.empty() && n_pairs >= 2 * 4096) {
    double padded_cells = 0, walk_cells = 0;
    for(i64 k = 0; k < n_pairs; ++k) {
      const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      padded_cells += (double)(la + 63) * (double)dp_padded_cols(lb, h->cols_per_lane, h->tail);
      walk_cells += (double)DP_CK_R * (double)(la + lb); // the blocks a path crosses: about 64 (La + Lb) cells
    }
    const double fill_s = padded_cells / 5.6e12, walk_s = walk_cells / 1.2e12; // measured rates of the two kernels, each alone
    if(fill_s >= 0.020 && walk_s >= 0.08 * (fill_s + walk_s)) {
      split = 2;
      // The second chunk is the smaller one -- a quarter of the workspace, but no fewer than 4 096 pairs' worth -- so that the path
      // kernel left alone at the end is the short one; the first has the second chunk's fill kernel to run beside (a quarter of the
      // fill is still twice the first chunk's walk).  4 096 pairs: below that a launch is one of two wavefronts per pair (25 000 pairs
      // of 8 x 4 096: 79.0 ms in halves, 76.7 at 3 : 1; 12 500: 42.2 in halves, 43.9 at 3 : 1 -- 3 125 pairs in the second).
      uneven = true;
      first_share = 1.0 - std::max(0.25, 4096.0 / (double)n_pairs);
    }
  }
  // (tests and the fuzzer cut tiny batches into many segments, opt.segment_cells: theirs of four segments' worth and more go this way too)
  const double per_segment_from = h->opt.segment_cells > 0 ? 4.0 * (double)h->opt.segment_cells : 1e11;
  const bool per_segment = segmented && (h->seg_first.size() == 2 || (double)h->cells >= per_segment_from); // chunks end where segments end (below)
  const bool sorted_chunks = !h->opt.keep_order && (!segmented || per_segment);                // ... and are ordered longest first
  bool pipelined = false;
  for(i64 budget = h->tb_budget_bytes;; budget /= 2) {
    const bool one_chunk = total_words <= budget / 4;
    i64 budget_words = one_chunk ? (uneven ? (i64)((double)total_words * first_share) : (total_words + split - 1) / split) + 64 : budget / (4 * h->n_slots);
    h->slot_reuse = !one_chunk;
    // chunk_first: positions in `order`; chunk_tb: the word offset of every position's pair inside its chunk's workspace
    h->chunk_first.assign(1, 0);
    h->chunk_tb.clear();
    h->tb_words_cap = 0;
    // Greedy: a chunk takes pairs while they fit; in a batch that is arriving in segments it also ends where a segment ends (with
    // segments no larger than chunks the first chunk starts when the FIRST segment is up: the headline batch in 8 segments 15 ms into
    // the call, profiles/r04_stream.txt), and its pairs are then put longest first.
    // (Not a small batch in several segments: 10 000 pairs of 2 x 1 kbp in two segments are 5.9 ms as ONE chunk in the input order
    // with a fill launch per segment -- the second segment's upload beside the first one's kernel -- and 7.0 as two chunks.)
    if(segmented) {
      for(i64 k = 0; k < n_pairs; ++k) {
        h->order[(size_t)k] = (int)k; // (a second attempt with a smaller budget cuts other chunks)
      }
    }
    size_t next_seg = 1; // the first segment boundary behind the chunk's start
    for(i64 c_lo = 0; c_lo < n_pairs || h->chunk_tb.empty();) {
      while(per_segment && next_seg < h->seg_first.size() && h->seg_first[next_seg] <= c_lo) {
        ++next_seg;
      }
      const i64 seg_end = per_segment && next_seg < h->seg_first.size() ? h->seg_first[next_seg] : n_pairs;
      i64 used = 0, q = c_lo;
      while(q < n_pairs && q < seg_end) {
        const i64 need = need_words(h->order[(size_t)q]);
        if(q > c_lo && used + need > budget_words) {
          break;
        }
        used += need;
        ++q;
      }
      if(per_segment && !cost.empty()) {
        std::stable_sort(h->order.begin() + c_lo, h->order.begin() + q, [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
      }
      std::vector<i64> cur;
      cur.reserve((size_t)(q - c_lo));
      i64 at_words = 0;
      for(i64 p = c_lo; p < q; ++p) {
        cur.push_back(at_words);
        at_words += need_words(h->order[(size_t)p]);
      }
      h->chunk_tb.push_back(cur);
      h->chunk_first.push_back(q);
      h->tb_words_cap = std::max(h->tb_words_cap, used);
      if(one_chunk && uneven) {
        budget_words = total_words; // the second chunk takes the rest
      }
      c_lo = q;
      if(n_pairs == 0) {
        break;
      }
    }
    // (cutting the last chunk once more, so that the path kernel that runs alone at the very end is a short one, was measured and
    // costs more than it saves: the headline batch 310.2 -> 312.4 ms, profiles/r03_dp_chunks.txt)
    pipelined = h->chunk_tb.size() > 1;
    h->tb_half_words = pipelined ? ((h->tb_words_cap + 63) / 64) * 64 : 0;
    const i64 parts = pipelined ? (h->slot_reuse ? std::min<i64>(h->n_slots, (i64)h->chunk_tb.size()) : (i64)h->chunk_tb.size()) : 1;
    h->chunk_base.assign(h->chunk_tb.size(), 0);
    i64 tb_total = pipelined ? parts * h->tb_half_words : h->tb_words_cap;
    if(pipelined && !h->slot_reuse) { // every chunk has a part of its own, as large as it needs
      i64 at = 0;
      for(size_t c = 0; c < h->chunk_tb.size(); ++c) {
        h->chunk_base[c] = at;
        const i64 c_words = h->chunk_tb[c].empty() ? 0 : h->chunk_tb[c].back() + need_words(h->order[(size_t)(h->chunk_first[c + 1] - 1)]);
        at += ((c_words + 63) / 64) * 64;
      }
      tb_total = at;
    }
    else {
      for(size_t c = 0; c < h->chunk_tb.size(); ++c) {
        h->chunk_base[c] = pipelined ? (i64)(c % (size_t)h->n_slots) * h->tb_half_words : 0;
      }
    }
    const int rc = grow(h->tb, (size_t)tb_total * 4);
    if(rc == PM_OK) {
      break;
    }
    (void)hipGetLastError(); // the failed allocation's sticky error
    if(budget <= ((i64)256 << 20) || (i64)h->chunk_tb.size() >= n_pairs) {
      return rc; // one pair per chunk already, or nothing sensible left to try
    }
  }
  if(pipelined) {
    if(!h->path_stream) { // (at the highest or the lowest priority: measured, within the run-to-run spread, profiles/r03_c2_chunks_priority.txt)
      PM_HIP(hipStreamCreateWithFlags(&h->path_stream, hipStreamNonBlocking));
    }
    while((int)h->fill_streams.size() < h->n_slots - 1) {
      hipStream_t st = nullptr;
      PM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      h->fill_streams.push_back(st);
    }
    if(!h->ev_begin) {
      PM_HIP(hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming));
    }
    while(h->ev_fill.size() < h->chunk_tb.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      PM_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      h->ev_fill.push_back(a);
      PM_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      h->ev_path.push_back(b);
    }
  }
  // tiers of long pairs (dp_batch.hpp): per chunk, the pairs that alone would take more than half of what the launch takes when the
  // chip is evenly loaded (steps of all its pairs / 4 096 wavefronts) go first, in launches of at most 256 and 512 pairs -- small
  // enough for dp_launch_fill to give them 8 and 4 wavefronts each (a third launch of the next 1 024 pairs, at 4 wavefronts each,
  // was the last to finish: 14.2 ms beside 11.6 for the rest).  Uniform batches have no such pairs.  PM_DP_NO_TIERS=1: not.
  h->chunk_tiers.assign(h->chunk_tb.size(), std::vector<i64>());
  if(!h->opt.no_tiers && sorted_chunks && h->waves_override == 0) {
    bool any = false;
    for(size_t c = 0; c < h->chunk_tb.size(); ++c) {
      const i64 c_lo = h->chunk_first[c], c_hi = h->chunk_first[c + 1], n = c_hi - c_lo;
      // (the wavefronts an evenly loaded chip runs side by side: 4 096; the fuzzer lowers the mark so that tiny batches get tiers too)
      const i64 W = h->opt.tier_min_pairs > 0 ? h->opt.tier_min_pairs : 4096;
      if(n < W) {
        continue; // dp_launch_fill already gives such a launch several wavefronts per pair
      }
      if(dp_tiles_rule(h, c_lo, n, !h->ckpt) > 0) {
        continue; // the launch takes tiles from a queue: its longest pairs' tiles go first, on as many wavefronts as they have stripes
      }
      auto cost_at = [&](i64 q) {
        const i64 k = h->order[(size_t)q];
        return dp_fill_cost(h->off_a[k + 1] - h->off_a[k], h->off_b[k + 1] - h->off_b[k], h->cols_per_lane, h->tail);
      };
      double total = 0;
      for(i64 q = c_lo; q < c_hi; ++q) {
        total += (double)cost_at(q);
      }
      // only a launch that its longest pair bounds: that pair's steps against the steps every wavefront gets when the chip is
      // evenly loaded.  A wavefront that has a SIMD to itself steps about three times as fast as one of five sharing it, so the
      // longest pair finishes with the others up to a ratio of about three (measured on the ragged stand-in: ratio 3.9 at
      // 12 500 pairs, 1 877 -> 2 625 GCUPS with the tiers; 1.9 at 25 000 pairs, 3 550 -> 3 386: the launches of several
      // wavefronts per pair are the less efficient ones)
      if((double)cost_at(c_lo) < 3.0 * total / (double)W) {
        continue;
      }
      const double limit = total / (double)W / 2.0;
      i64 heavy = 0; // the order is longest first: the heavy pairs are a prefix
      while(heavy < n && heavy < std::max<i64>(3 * W / 16, 1) && (double)cost_at(c_lo + heavy) > limit) {
        ++heavy;
      }
      if(heavy == 0 || heavy * 2 > n) {
        continue;
      }
      std::vector<i64> &cuts = h->chunk_tiers[c];
      i64 at = 0;
      for(i64 size : {std::max<i64>(W / 16, 1), std::max<i64>(W / 8, 1)}) {
        if(at >= heavy) {
          break;
        }
        at = std::min(heavy, at + size);
        cuts.push_back(c_lo + at);
      }
      any = true;
    }
    if(any) {
      while(h->tier_streams.size() < 3) {
        // at the highest priority: the tiers are what the step waits for, and the runtime gives every priority level hardware queues
        // of its own -- the seven streams of a ragged batch in several chunks no longer share four (profiles/r04_stream.txt)
        hipStream_t st = nullptr;
        int prio_least = 0, prio_greatest = 0;
        PM_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        PM_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_greatest));
        h->tier_streams.push_back(st);
      }
      while(h->tier_events.size() < 4) {
        hipEvent_t e = nullptr;
        PM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->tier_events.push_back(e);
      }
    }
  }
  // the workspace offset of every pair (indexed by pair) and the processing order go to the device
  {
    const size_t npad = (size_t)std::max<i64>(n_pairs, 1);
    i64 *flat = nullptr;
    int *ord = nullptr;
    std::vector<i64> tmp;
    std::vector<int> tmp_o;
    if(staged) {
      flat = (i64 *)h->pinned + 2 * (n_pairs + 1);
      ord = (int *)((i64 *)h->pinned + 3 * (n_pairs + 1));
    }
    else {
      tmp.resize(npad);
      tmp_o.resize(npad);
      flat = tmp.data();
      ord = tmp_o.data();
    }
    size_t q = 0;
    for(size_t c = 0; c < h->chunk_tb.size(); ++c) {
      for(i64 v : h->chunk_tb[c]) {
        flat[(size_t)h->order[q]] = v;
        ord[q] = h->order[q];
        ++q;
      }
    }
    PM_TRY(grow(h->d_order, npad * 4));
    // The band (dp_internal.hpp): a one-chunk batch whose walk would leave most of the chip idle (at most two wavefronts per
    // SIMD) gets the decisions around every pair's diagonal computed up front.  Not for a batch that is reloaded slice after
    // slice from pinned staging (dp_stream.hip): its tables would have to follow every slice.
    h->band_work_items = 0;
    if(h->ckpt && !staged && h->chunk_tb.size() == 1 && n_pairs > 0 && band_env != 0) {
      const int lpp = dp_walk_lanes_for(h, 0, n_pairs);
      const i64 bw = (i64)h->cols_per_lane * DP_CK_W;
      const i64 block_bytes = dp_band_block_bytes(h->cols_per_lane, lpp);
      i64 items = 0;
      for(i64 k = 0; k < n_pairs; ++k) {
        const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
        items += la > 0 ? (lb + bw - 1) / bw : 0;
      }
      const i64 band_bytes = items * DP_BAND_BLOCKS * block_bytes;
      if(items > 0 && (band_env > 0 || band_pays(n_pairs, lpp)) && band_bytes <= ((i64)(band_env > 0 ? 32 : 4) << 30)) {
        std::vector<int> work((size_t)items * 2);
        std::vector<i64> boff((size_t)n_pairs);
        i64 at = 0;
        for(i64 q2 = 0; q2 < n_pairs; ++q2) { // in processing order: the longest pairs' blocks first
          const i64 k = h->order[(size_t)q2];
          const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
          const i64 groups = la > 0 ? (lb + bw - 1) / bw : 0;
          boff[(size_t)k] = at * DP_BAND_BLOCKS;
          for(i64 g = 0; g < groups; ++g) {
            work[(size_t)(at + g) * 2] = (int)k;
            work[(size_t)(at + g) * 2 + 1] = (int)g;
          }
          at += groups;
        }
        // the band is an accelerator, not a need: a device that has no room for it walks without
        if(grow(h->d_band_work, (size_t)items * 8) == PM_OK && grow(h->d_band_off, (size_t)n_pairs * 8) == PM_OK &&
           grow(h->band_bits, (size_t)band_bytes) == PM_OK) {
          PM_HIP(hipStreamSynchronize(stream));
          PM_HIP(hipMemcpy(h->d_band_work.p, work.data(), (size_t)items * 8, hipMemcpyHostToDevice));
          PM_HIP(hipMemcpy(h->d_band_off.p, boff.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice));
          h->band_work_items = items;
          h->band_lanes = lpp;
        }
        else {
          (void)hipGetLastError();
        }
      }
    }
    if(n_pairs > 0) {
      if(staged) {
        PM_HIP(hipMemcpyAsync(h->d_tb_off.p, flat, (size_t)n_pairs * 8, hipMemcpyHostToDevice, stream));
        PM_HIP(hipMemcpyAsync(h->d_order.p, ord, (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream));
      }
      else { // pageable sources: blocking copies, the vectors die with this scope
        PM_HIP(hipStreamSynchronize(stream));
        PM_HIP(hipMemcpy(h->d_tb_off.p, flat, (size_t)n_pairs * 8, hipMemcpyHostToDevice));
        PM_HIP(hipMemcpy(h->d_order.p, ord, (size_t)n_pairs * 4, hipMemcpyHostToDevice));
      }
    }
  }
  return PM_OK;
}

} // namespace pm

extern "C" {

int pm_dp_set_default_options(const pm_dp_options_t *options) {
  std::lock_guard<std::mutex> hold(pm::g_default_options_lock);
  if(options) {
    pm::g_default_options = *options;
  }
  else {
    pm::g_default_options = pm_dp_options_t{};
  }
  return PM_OK;
}

int pm_dp_batch_create(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                       const pm_dp_params_t *params, int64_t tb_budget_bytes, int device, pm_dp_batch_t **out) {
  return pm_dp_batch_create_opt(cols_a, off_a, cols_b, off_b, n_pairs, params, nullptr, tb_budget_bytes, device, out);
}

int pm_dp_batch_create_opt(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                           const pm_dp_params_t *params, const pm_dp_options_t *options, int64_t tb_budget_bytes, int device,
                           pm_dp_batch_t **out) {
  if(!out) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: null out");
  }
  *out = nullptr;
  if(n_pairs < 0 || !off_a || !off_b || !params) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: null argument");
  }
  PM_TRY(use_device(device));
  if(off_a[0] != 0 || off_b[0] != 0) {
    return fail(PM_E_INVALID, "pm_dp_batch_create: offsets must start at 0");
  }
  PM_TRY(dp_batch_check_params(params));
  pm_dp_batch *h = new(std::nothrow) pm_dp_batch();
  if(!h) {
    return fail(PM_E_INVALID, "out of host memory");
  }
  int rc = dp_batch_init(h, params, tb_budget_bytes, device, options);
  if(!rc) {
    rc = dp_batch_load(h, cols_a, off_a, cols_b, off_b, n_pairs, nullptr);
  }
  if(!rc && hipStreamSynchronize(nullptr) != hipSuccess) {
    rc = fail(PM_E_HIP, "upload failed");
  }
  if(!rc) {
    rc = dp_batch_plan(h, nullptr);
  }
  if(rc) {
    pm_dp_batch_destroy(h);
    return rc;
  }
  *out = h;
  return PM_OK;
}

} // extern "C"

// In front of the fill kernel of chunk c + 1, on its stream: holds it back until every workgroup of chunk c's fill kernel has
// STARTED, i.e. until that launch has nothing left to dispatch and the chip begins to drain -- chunk c + 1's workgroups then take
// the SIMDs as they fall free, instead of competing with chunk c's from the start (two fill kernels side by side would both
// finish late, and the path kernel of chunk c with them) or waiting for the last of them to end (the drain left idle).  One lane
// polls.  Nothing depends on it for correctness, so the wait is simply bounded (about a second).
__global__ void dp_gate_kernel(const int *__restrict__ started, int total) {
  if(threadIdx.x == 0) {
    for(unsigned spins = 0; spins < (1u << 18); ++spins) {
      if(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) {
        break;
      }
      __builtin_amdgcn_s_sleep(127);
    }
  }
}

// The gate itself: the polling kernel above, with its bounded wait.  Round 5 tried the runtime's own stream operation instead
// (hipStreamWaitValue32 on the counter; probed by itself first, tools/ubench/wait_value.hip: it does hold a stream back until a kernel
// on another stream has counted its workgroups in) hoping to take the gates out of the kernel trace -- and found that the runtime
// implements it as a polling kernel of its own (`__amd_rocclr_streamOpsWait`: 66 calls x 40.4 ms in the headline's trace, exactly
// where dp_gate_kernel's 66 x 40.4 ms had been), WITHOUT a time-out: under `rocprofv3 --pmc`, which runs kernels one at a time in
// whatever order it picks them from the queues, a wait picked before the kernel it waits for never ends, where the bounded kernel
// gives up after a second and lets the pass go on slowly (the profile refresh hung on its first counter pass).  So: this kernel.
static int dp_gate(hipStream_t stream, int *started, int total) {
  dp_gate_kernel<<<1, 64, 0, stream>>>(started, total);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// The fill kernel of chunk c into workspace `tbw`.  started: the chunk's counter of started workgroups, or null; *groups: the launch's
// workgroups.
// filled: where a launch of one wavefront per pair publishes its finished pairs for the walk beside it (null: nobody listens);
// *filled_used: whether this launch does (it does not when it takes tiles or gives a pair several wavefronts).
static int dp_launch_fill(pm_dp_batch *h, i64 first, i64 n, unsigned *tbw, int traceback, hipStream_t stream, int *started = nullptr,
                          i64 *groups = nullptr, const DpFilled *filled = nullptr, bool *filled_used = nullptr) {
  if(filled_used) {
    *filled_used = false;
  }
  const i64 *tb_off = (const i64 *)h->d_tb_off.p;
  const int *order = (const int *)h->d_order.p + first;
  static const int cus_of_device = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    return hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
  }();
  // tiles from a queue (dp_fill_tiles_kernel)?  The list is made at the launch's first pass and kept with the layout
  if(const int T = dp_tiles_rule(h, first, n, traceback && !h->ckpt)) {
    DpTilePlan *plan = nullptr;
    for(std::unique_ptr<DpTilePlan> &p : h->tile_plans) {
      if(p->first == first && p->n == n && p->tile_steps == T) {
        plan = p.get();
      }
    }
    if(!plan) {
      std::unique_ptr<DpTilePlan> made(new DpTilePlan());
      PM_TRY(dp_tiles_build(h, first, n, T, *made));
      plan = made.get();
      h->tile_plans.push_back(std::move(made));
    }
    // progress words, tiles-done words and the ticket counter start from zero every pass (the lanes' states are written before read)
    PM_HIP(hipMemsetAsync(plan->sync_words.p, 0, (size_t)(2 * plan->total_stripes + 1) * 4, stream));
    const i64 slots = (i64)std::max(cus_of_device, 1) * 4 * (h->dot4 ? 5 : 4); // what the chip holds of this kernel at once
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(std::max<i64>(plan->n_tiles, (n + 63) / 64), slots));
#define DP_LAUNCH_TILES(TR, D4, UN)                                                                                                          \
  dp_fill_tiles_kernel<16, TR, D4, UN><<<grid, 64, 0, stream>>>((const u64 *)h->cols_a.p, (const i64 *)h->d_off_a.p, (const u64 *)h->cols_b.p, \
                                                                (const i64 *)h->d_off_b.p, order, tb_off, tbw, (int2 *)h->bnd.p,             \
                                                                (int *)h->scores.p, (int *)h->pipe_error.p, h->params,                      \
                                                                (const DpTile *)plan->tiles.p, (int)plan->n_tiles,                          \
                                                                (const i64 *)plan->sync_off.p, plan->total_stripes, (int *)plan->sync_words.p, \
                                                                (unsigned *)plan->state.p, started, h->tail ? 1 : 0, (int)n)
#define DP_LAUNCH_TILES_D4(TR)            \
  if(h->dot4 && h->uni) {                 \
    DP_LAUNCH_TILES(TR, true, true);      \
  }                                       \
  else if(h->dot4) {                      \
    DP_LAUNCH_TILES(TR, true, false);     \
  }                                       \
  else if(h->uni) {                       \
    DP_LAUNCH_TILES(TR, false, true);     \
  }                                       \
  else {                                  \
    DP_LAUNCH_TILES(TR, false, false);    \
  }
    if(traceback) {
      DP_LAUNCH_TILES_D4(DP_MODE_CKPT)
    }
    else {
      DP_LAUNCH_TILES_D4(DP_MODE_SCORE)
    }
#undef DP_LAUNCH_TILES_D4
#undef DP_LAUNCH_TILES
    PM_HIP(hipGetLastError());
    if(groups) {
      *groups = (i64)grid;
    }
    return PM_OK;
  }
  // waves per pair: one, unless the launch has too few pairs to fill the chip (1 024 SIMDs x 4 waves) and the pairs
  // have several stripes to pipeline
  int nw = 1, ng = 1;
  {
    i64 max_stripes = 0, max_la = 0;
    for(i64 q = first; q < first + n; ++q) {
      const i64 k = h->order[(size_t)q];
      i64 lbk = h->off_b[k + 1] - h->off_b[k];
      max_stripes = std::max(max_stripes, dp_ck_stripes(lbk, h->cols_per_lane, h->tail));
      max_la = std::max(max_la, h->off_a[k + 1] - h->off_a[k]);
    }
    bool fits = (max_stripes + 2) * max_la < ((i64)1 << 30); // the progress word is an int
    // enough waves to give every SIMD about four (1 024 SIMDs), as far as the pairs have stripes to run side by side
    if(fits && max_stripes >= 2 && n < 4096) {
      // 16 waves (one workgroup filling a CU) once a pair has 10 stripes or more: all of them run side by side
      while(nw < 16 && nw * 2 <= max_stripes + (nw >= 8 ? 6 : 0) && n * nw < 4096) {
        nw *= 2;
      }
    }
    if(h->waves_override == 1 ||
       ((h->waves_override == 2 || h->waves_override == 4 || h->waves_override == 8 || h->waves_override == 16) && fits)) {
      nw = h->waves_override;
    }
    // workgroups per pair: with at most half as many pairs as CUs, as many as give every workgroup a CU of its own and still two
    // stripes; the waves of a workgroup then follow from the stripes it gets (consecutive stripes go to different workgroups)
    const int cus = cus_of_device;
    // (also for a launch that runs beside another fill launch of this batch -- dp_run's fill streams, the tiers: every launch has
    // progress words of its own, and the tickets make the order in which workgroups start irrelevant)
    const int groups_env = h->opt.groups_per_pair ? h->opt.groups_per_pair : -1; // 1: never; 2, 4, ..: that many; -1: chosen here
    if(fits && groups_env > 1 && h->waves_override >= 2 && n * groups_env <= cus) { // both forced (tests)
      ng = groups_env;
    }
    else if(fits && max_stripes >= 4 && groups_env != 1 && h->waves_override == 0) {
      int want = 1;
      while(want < 16 && n * want * 2 <= cus && want * 4 <= max_stripes) {
        want *= 2;
      }
      if(groups_env > 1) {
        want = groups_env;
      }
      if(want > 1 && n * want <= cus && (i64)want * 2 <= max_stripes) { // every workgroup of the launch must be resident
        ng = want;
        const i64 per_group = (max_stripes + ng - 1) / ng;
        nw = 2;
        while(nw < 16 && nw < per_group) {
          nw *= 2;
        }
      }
    }
  }
  int *gprog = nullptr;
  if(ng > 1) {
    // one progress word per wave of the launch, then the ticket counter the workgroups draw their places from: this launch's own
    // piece of the arena dp_run has sized for all the launches of a pass (two launches of a pass may overlap on their streams)
    const size_t words = (size_t)(n * ng * nw + 1);
    if(h->gprog_used + words > h->gprog.bytes / sizeof(int)) {
      ng = 1; // (cannot happen: DP_GPROG_WORDS_PER_LAUNCH bounds n * ng * nw + 1)
      nw = std::min(nw, 16);
    }
    else {
      gprog = (int *)h->gprog.p + h->gprog_used;
      h->gprog_used += words;
      PM_HIP(hipMemsetAsync(gprog, 0, words * sizeof(int), stream));
    }
  }
  DpFilled fl = {nullptr, nullptr, nullptr, 0};
  if(filled && nw == 1 && ng == 1) {
    fl = *filled;
    if(filled_used) {
      *filled_used = true;
    }
  }
#define DP_LAUNCH_FILL(CC, TR, D4, NWV, UN)                                                                                                  \
  dp_fill_kernel<CC, TR, D4, NWV, UN><<<(unsigned)(n * ng), 64 * NWV, 0, stream>>>((const u64 *)h->cols_a.p, (const i64 *)h->d_off_a.p,    \
                                                                                   (const u64 *)h->cols_b.p, (const i64 *)h->d_off_b.p,    \
                                                                                   order, tb_off, tbw, (int2 *)h->bnd.p, (int *)h->scores.p, \
                                                                                   (int *)h->pipe_error.p, h->params, ng, gprog, started, \
                                                                                   h->tail ? 1 : 0, fl)
#define DP_LAUNCH_FILL_D4(CC, TR, NWV)         \
  if(h->dot4 && h->uni) {                      \
    DP_LAUNCH_FILL(CC, TR, true, NWV, true);   \
  }                                            \
  else if(h->dot4) {                           \
    DP_LAUNCH_FILL(CC, TR, true, NWV, false);  \
  }                                            \
  else if(h->uni) {                            \
    DP_LAUNCH_FILL(CC, TR, false, NWV, true);  \
  }                                            \
  else {                                       \
    DP_LAUNCH_FILL(CC, TR, false, NWV, false); \
  }
#define DP_LAUNCH_FILL_TR(CC, NWV)                \
  if(traceback && h->ckpt) {                      \
    DP_LAUNCH_FILL_D4(CC, DP_MODE_CKPT, NWV)      \
  }                                               \
  else if(traceback) {                            \
    DP_LAUNCH_FILL_D4(CC, DP_MODE_BITS, NWV)      \
  }                                               \
  else {                                          \
    DP_LAUNCH_FILL_D4(CC, DP_MODE_SCORE, NWV)     \
  }
#define DP_LAUNCH_FILL_NW(CC)    \
  switch(nw) {                   \
  case 16:                       \
    DP_LAUNCH_FILL_TR(CC, 16)    \
    break;                       \
  case 8:                        \
    DP_LAUNCH_FILL_TR(CC, 8)     \
    break;                       \
  case 4:                        \
    DP_LAUNCH_FILL_TR(CC, 4)     \
    break;                       \
  case 2:                        \
    DP_LAUNCH_FILL_TR(CC, 2)     \
    break;                       \
  default:                       \
    DP_LAUNCH_FILL_TR(CC, 1)     \
    break;                       \
  }
  if(h->cols_per_lane == 16) {
    DP_LAUNCH_FILL_NW(16)
  }
  else {
    DP_LAUNCH_FILL_NW(8)
  }
#undef DP_LAUNCH_FILL_NW
#undef DP_LAUNCH_FILL_TR
#undef DP_LAUNCH_FILL_D4
#undef DP_LAUNCH_FILL
  PM_HIP(hipGetLastError());
  if(groups) {
    *groups = n * ng;
  }
  return PM_OK;
}

// The path kernel of the pairs at positions [first, first + n) (of one chunk) from the chunk's workspace `tbw`: the checkpoint walk,
// or the walk over stored decision bits.
static int dp_launch_path(pm_dp_batch *h, i64 first, i64 n, const unsigned *tbw, hipStream_t stream, bool urgent = false,
                          const DpEarly *early = nullptr, unsigned early_groups = 0, int lanes = 0) {
  if(n <= 0) {
    return PM_OK;
  }
  const i64 *tb_off = (const i64 *)h->d_tb_off.p;
  const int *order = (const int *)h->d_order.p + first;
  if(h->ckpt) {
    const int lpp = lanes ? lanes : dp_walk_lanes_for(h, first, n);
    if(early && early->mode != 0) { // beside / behind the fill kernel of the same launch: no band (such launches are large)
      const DpBand none = {nullptr, 0, nullptr, nullptr};
      return dp_launch_walk(h->cols_per_lane, lpp, h->dot4, (const u64 *)h->cols_a.p, (const i64 *)h->d_off_a.p, (const u64 *)h->cols_b.p,
                            (const i64 *)h->d_off_b.p, order, n, tb_off, tbw, (unsigned char *)h->ops.p, (int *)h->n_ops.p, h->params, none,
                            h->tail ? 1 : 0, urgent ? 1 : 0, stream, early, early_groups);
    }
    DpBand band = {nullptr, 0, nullptr, nullptr};
    if(h->band_work_items > 0 && h->band_lanes == lpp) {
      band.work = (const int *)h->d_band_work.p;
      band.n_work = h->band_work_items;
      band.bits = (unsigned *)h->band_bits.p;
      band.off = (const i64 *)h->d_band_off.p;
    }
    return dp_launch_walk(h->cols_per_lane, lpp, h->dot4, (const u64 *)h->cols_a.p, (const i64 *)h->d_off_a.p, (const u64 *)h->cols_b.p,
                          (const i64 *)h->d_off_b.p, order, n, tb_off, tbw, (unsigned char *)h->ops.p, (int *)h->n_ops.p, h->params, band,
                          h->tail ? 1 : 0, urgent ? 1 : 0, stream);
  }
  if(h->cols_per_lane == 16) {
    dp_traceback_kernel<16><<<(unsigned)n, 64, 0, stream>>>((const i64 *)h->d_off_a.p, (const i64 *)h->d_off_b.p, order, tb_off, tbw,
                                                            (unsigned char *)h->ops.p, (int *)h->n_ops.p);
  }
  else {
    dp_traceback_kernel<8><<<(unsigned)n, 64, 0, stream>>>((const i64 *)h->d_off_a.p, (const i64 *)h->d_off_b.p, order, tb_off, tbw,
                                                           (unsigned char *)h->ops.p, (int *)h->n_ops.p);
  }
  PM_HIP(hipGetLastError());
  return PM_OK;
}

// One pass over every chunk.  One chunk: fill, then path, on `stream`.  Several chunks: the fill kernels of the workspace's parts
// on `stream` and on the batch's own fill streams, each behind a gate kernel that lets it go when the chunk before it has nothing
// left to dispatch; path kernels on the batch's path stream, chunk c's beside the fill kernels of the chunks after it; `stream`
// ends up waiting for the last path kernels, so the caller sees one asynchronous operation on its stream.
// Timed: device time of the fill and of the path kernels, summed over the chunks (events around every launch on its stream), and
// the time during which some fill kernel ran (h->last_fill_busy_ms).
namespace pm {
// Default path workspace: 60 % of the device's memory (172 GB of an MI355X's 288 GB).  A launch ends with the chip draining -- the
// last round of pairs fills only part of it -- so the fewer launches the better: the 100 000 ragged pairs of bench.py's c2 (139 GB
// of checkpoints) and configs[4]'s 4 096 deep pairs (108 GB) run as ONE fill launch and one path launch (4 100 and 4 060 GCUPS
// against 3 900 and 3 490 in 48 GiB chunks); a batch that needs more (the headline batch: 424 GB) is cut into chunks of a third of
// the budget whose fill kernels overlap on streams of their own (dp_run).  The workspace is only as large as the chunks need.
int64_t dp_default_budget_bytes() {
  size_t free_b = 0, total_b = 0;
  if(hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
    return (int64_t)(total_b / 10 * 6);
  }
  return (int64_t)96 << 30;
}

int dp_run(pm_dp_batch *h, hipStream_t stream, int traceback, float *ms_fill, float *ms_path) {
  const size_t nc = h->chunk_tb.size();
  const bool timed = ms_fill || ms_path;
  const bool pipelined = nc > 1 && h->path_stream && traceback;
  if(timed && h->tv_fill0.size() < nc) {
    for(std::vector<hipEvent_t> *v : {&h->tv_fill0, &h->tv_fill1, &h->tv_path0, &h->tv_path1}) {
      while(v->size() < nc) {
        hipEvent_t e = nullptr;
        PM_HIP(hipEventCreate(&e));
        v->push_back(e);
      }
    }
  }
  h->tv_tier_chunk.clear();
  unsigned *tb = (unsigned *)h->tb.p;
  // two fill streams (see dp_batch.hpp): the fill kernel of an odd chunk is held back only by the path kernel that frees its half
  // of the workspace, not by the fill kernel of the chunk before it
  const int K = h->n_slots;
  // (also for a batch that is still arriving in segments -- the host-fed engine: a chunk lies inside one segment, waits for it and
  // goes as one launch.  The uploads run far ahead of the kernels, 57 GB/s against the 22 GB/s of columns the fill kernels use up,
  // so only the first chunk ever waits; a launch per (chunk, segment) on one stream, as round 3 had them, cost the headline batch
  // 342 instead of 297 ms of kernels: profiles/r04_stream.txt)
  const bool two_fills = pipelined && (int)h->fill_streams.size() >= K - 1 && h->ev_begin;
  // progress words for every launch of the pass that may ask for several workgroups per pair (dp_launch_fill takes a piece each):
  // n * ng <= CUs and at most 16 waves per workgroup bound a launch's
  {
    size_t launches = 0;
    for(size_t c = 0; c < nc; ++c) {
      launches += 1 + (c < h->chunk_tiers.size() ? h->chunk_tiers[c].size() : 0) + (h->seg_first.empty() ? 0 : h->seg_first.size());
    }
    const size_t per_launch = 1024 * 16 + 1; // (an MI355X has 256 CUs; room for four times as many)
    if(h->gprog.bytes < launches * per_launch * sizeof(int)) {
      PM_TRY(h->gprog.alloc(launches * per_launch * sizeof(int))); // (hipFree of the old arena waits for the device: no launch of an earlier pass still uses it)
    }
    h->gprog_used = 0;
  }
  int *started = nullptr;
  if(two_fills) {
    PM_TRY(grow(h->fill_started, nc * sizeof(int)));
    started = (int *)h->fill_started.p;
    PM_HIP(hipMemsetAsync(started, 0, nc * sizeof(int), stream)); // ordered before every fill stream's work by ev_begin
    h->chunk_groups.assign(nc, 0);
    PM_HIP(hipEventRecord(h->ev_begin, stream));
    for(int k = 0; k < K - 1; ++k) {
      PM_HIP(hipStreamWaitEvent(h->fill_streams[(size_t)k], h->ev_begin, 0));
    }
  }
  const bool gate_on = !h->opt.no_gate;
  // The walk beside the fill kernel of its own launch (DpEarly, dp_internal.hpp): for the launches whose walk nothing else hides -- a
  // batch of one chunk, and the last chunk of a batch of several (the chunks before it walk beside the fill kernels that follow them).
  // prepare: the lists the fill kernel publishes into, zeroed on the fill stream in front of it; begin: the early walkers on their own
  // stream (a wavefront per SIMD, two where the walk is a large share of the work); finish (in place of the path kernel): the same
  // kernel behind the fill kernel over the whole chip for what is left, and the chunk is done when the early walkers are, too.
  struct EarlyWalk {
    bool used = false;
    size_t slot = 0;
    i64 first = 0, n = 0;
    int *base = nullptr;
    DpFilled fl = {nullptr, nullptr, nullptr, 0};
  } ew;
  auto early_prepare = [&](size_t c, i64 first, i64 n, hipStream_t fill_stream) -> int {
    ew = EarlyWalk();
    // By itself: for the launches whose walk nothing else hides (a batch of one chunk, the last chunk of a batch of several) when the
    // pairs are LONG (a mean of 3 000 columns and more in La + Lb).  Measured on one MI355X (profiles/r05_early_walk.txt), with the fill
    // kernel ahead of the walkers at issue (s_setprio: the walkers take the cycles the fill leaves): one GPU's eighth of the ragged
    // batch 16.3 -> 14.9 ms a step, of the headline batch 41.3 -> 40.3.  NOT for short pairs: 10 000 pairs of 2 x 1 000 finish in two
    // bursts, a round of wavefronts each -- there is nothing to walk for the first half of the launch and too much at its end, and the
    // walkers that are in the middle of a pair when the fill ends hold the launch behind it up (3.03 -> 3.37 ms; at equal priority
    // 3.72).  opt.early_walk: 1 never, 2 wherever the launch allows it.
    if(!traceback || !h->ckpt || h->opt.early_walk == 1 || h->band_work_items > 0 || !h->seg_first.empty() || n <= 0) {
      return PM_OK;
    }
    if(h->opt.early_walk != 2) {
      double columns = 0;
      for(i64 q = first; q < first + n; ++q) {
        const i64 k = h->order[(size_t)q];
        columns += (double)(h->off_a[k + 1] - h->off_a[k] + h->off_b[k + 1] - h->off_b[k]);
      }
      if(c + 1 != nc || n < 2048 || columns < 3000.0 * (double)n) {
        return PM_OK;
      }
    }
    if(!h->early_stream) {
      PM_HIP(hipStreamCreateWithFlags(&h->early_stream, hipStreamNonBlocking));
    }
    // lists per chunk -- or per part of the workspace where the chunks take those in turn: there the fill stream of chunk c has waited
    // for chunk c - K to be done with everything (ev_path) before it zeroes the lists chunk c - K used.  (The fuzzer's find of round 5:
    // a batch that fits, cut into five chunks with parts of their own, shared three sets of lists -- chunk 3 zeroed the lists chunk 0's
    // walkers were still waiting on, and they waited for a count that never came.)
    const size_t slots = h->slot_reuse ? (size_t)std::max(K, 1) : std::max<size_t>(nc, 1);
    while(h->ev_early_ready.size() < slots) {
      hipEvent_t a = nullptr, b = nullptr;
      PM_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      h->ev_early_ready.push_back(a);
      PM_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      h->ev_early_done.push_back(b);
    }
    i64 n_max = 0;
    for(size_t cc = 0; cc < nc; ++cc) {
      n_max = std::max(n_max, h->chunk_first[cc + 1] - h->chunk_first[cc]);
    }
    const size_t slot_ints = 96 + 8 * (size_t)n_max; // taken[8], resv[8], the count: a 128-byte line each, then the lists
    if(h->early_slot_ints < slot_ints || h->early_buf.bytes < slots * slot_ints * sizeof(int)) {
      PM_TRY(h->early_buf.alloc(slots * slot_ints * sizeof(int))); // (hipFree of the old lists waits for the device)
      h->early_slot_ints = slot_ints;
    }
    ew.slot = c % slots;
    ew.first = first;
    ew.n = n;
    ew.base = (int *)h->early_buf.p + ew.slot * h->early_slot_ints;
    ew.fl = DpFilled{ew.base + 32, ew.base + 64, ew.base + 96, (int)n};
    PM_HIP(hipMemsetAsync(ew.base, 0, (96 + 8 * (size_t)n) * sizeof(int), fill_stream));
    PM_HIP(hipEventRecord(h->ev_early_ready[ew.slot], fill_stream));
    return PM_OK;
  };
  auto early_begin = [&](const unsigned *tbw_) -> int { // after the fill launch that publishes (ew.used)
    PM_HIP(hipStreamWaitEvent(h->early_stream, h->ev_early_ready[ew.slot], 0));
    const DpEarly e1 = {1, (int)ew.n, ew.base, ew.base + 64, ew.base + 96};
    // how many pairs are walked at a time: a wavefront per SIMD, two where the walk is more than a seventh of the launch's work
    double fill_cells = 0, walk_cells = 0;
    for(i64 q = ew.first; q < ew.first + ew.n; ++q) {
      const i64 k = h->order[(size_t)q];
      const double la = (double)(h->off_a[k + 1] - h->off_a[k]), lb = (double)(h->off_b[k + 1] - h->off_b[k]);
      fill_cells += la * lb;
      walk_cells += (double)DP_CK_R * (la + lb);
    }
    (void)fill_cells;
    (void)walk_cells;
    // one wavefront per SIMD, of the walk's smallest footprint (32 lanes per pair: 96 VGPRs and 6 KB of LDS, so that the fill kernel
    // keeps four of its five wavefronts per SIMD) wherever the fill kernel's blocks allow it
    const int lpp = h->walk_lanes ? h->walk_lanes : (dp_walk_lanes_ok(h->cols_per_lane, 32) ? 32 : dp_walk_lanes_for(h, ew.first, ew.n));
    const unsigned groups = (unsigned)std::min<i64>(ew.n, (i64)1024 * (64 / lpp));
    PM_TRY(dp_launch_path(h, ew.first, ew.n, tbw_, h->early_stream, false, &e1, groups, lpp));
    PM_HIP(hipEventRecord(h->ev_early_done[ew.slot], h->early_stream));
    return PM_OK;
  };
  auto early_finish = [&](const unsigned *tbw_, hipStream_t ps_) -> int { // on the path stream, behind the fill kernel
    const DpEarly e2 = {2, (int)ew.n, ew.base, ew.base + 64, ew.base + 96};
    PM_TRY(dp_launch_path(h, ew.first, ew.n, tbw_, ps_, false, &e2, (unsigned)ew.n));
    PM_HIP(hipStreamWaitEvent(ps_, h->ev_early_done[ew.slot], 0));
    return PM_OK;
  };
  long prev_chunk = -1; // the last chunk that had pairs
  hipStream_t caller_stream = stream;
  i64 tiers_end = 0;       // the current chunk: the position behind its tiers (their path kernels ride on their own streams)
  size_t tiers_joined = 0; // and how many tier streams its end has to wait for
  for(size_t c = 0; c < nc; ++c) {
    if(h->chunk_first[c + 1] - h->chunk_first[c] <= 0) {
      continue;
    }
    const size_t slot = c % (size_t)K;
    unsigned *tbw = tb + (c < h->chunk_base.size() ? h->chunk_base[c] : 0);
    hipStream_t ps = pipelined ? h->path_stream : caller_stream;
    stream = two_fills && slot ? h->fill_streams[slot - 1] : caller_stream; // this chunk's fill stream
    if(pipelined && h->slot_reuse && c >= (size_t)K) {
      PM_HIP(hipStreamWaitEvent(stream, h->ev_path[c - (size_t)K], 0)); // the part is free again
    }
    if(two_fills && gate_on && prev_chunk >= 0 && h->chunk_groups[(size_t)prev_chunk] > 0) {
      PM_TRY(dp_gate(stream, started + prev_chunk, (int)std::min<i64>(h->chunk_groups[(size_t)prev_chunk], 0x7fffffff)));
    }
    if(timed) {
      PM_HIP(hipEventRecord(h->tv_fill0[c], stream));
    }
    // the chunk's pairs.  A batch that is arriving in segments: the chunk lies inside one of them (dp_batch_plan) and waits for it
    {
      const i64 c_lo = h->chunk_first[c], c_hi = h->chunk_first[c + 1];
      i64 at = c_lo;
      tiers_end = c_lo;
      tiers_joined = 0;
      ew = EarlyWalk();
      size_t spans = 0; // segments the chunk's pairs lie in
      for(size_t sg = 0; sg + 1 < h->seg_first.size(); ++sg) {
        spans += h->seg_first[sg + 1] > c_lo && h->seg_first[sg] < c_hi ? 1 : 0;
      }
      for(size_t sg = 0; sg + 1 < h->seg_first.size() && at < c_hi; ++sg) {
        if(!(h->seg_first[sg + 1] > c_lo && h->seg_first[sg] < c_hi)) {
          continue;
        }
        if(sg < h->ev_seg.size() && h->seg_events_armed) {
          PM_HIP(hipStreamWaitEvent(stream, h->ev_seg[sg], 0)); // the segment's columns are in HBM
        }
        if(spans > 1 && !two_fills) { // a small batch, in the input order (dp_batch_plan): a launch per segment, as the segments arrive
          const i64 s_hi = std::min(h->seg_first[sg + 1], c_hi);
          if(s_hi > at) {
            PM_TRY(dp_launch_fill(h, at, s_hi - at, tbw, traceback, stream));
            at = s_hi;
          }
        }
      }
      // the chunk's longest pairs in small launches of their own, on side streams, beside the launch of the rest
      if(at == c_lo && c < h->chunk_tiers.size() && !h->chunk_tiers[c].empty() && h->tier_streams.size() >= h->chunk_tiers[c].size() &&
         !h->tier_events.empty()) {
        // A tier's path kernel follows its fill kernel on the tier's own stream: the walk of the longest pair is a chain of a few
        // hundred blocks of about 15 us each -- milliseconds, whatever else the chip does -- and has to start as early as it can (the
        // ragged eighth's one path kernel after everything: 5.3 ms, of which the rest of its pairs needed 2).
        // The launch of the rest is held back (dp_gate_kernel) until every workgroup of the tiers has STARTED: when all of them become
        // ready at once -- the passes of a caller that does not wait in between -- whichever the dispatcher takes first fills the chip,
        // and the tiers, the launches the step waits for, got what the 11 700 workgroups of the rest left over (measured: a pass
        // 15.1 ms with a wait in between, 20.0 without; the trace showed the rest starting 12 us ahead of the tiers).
        const std::vector<i64> &cuts = h->chunk_tiers[c];
        PM_TRY(grow(h->tier_started, nc * sizeof(int)));
        int *tier_started = (int *)h->tier_started.p + c;
        PM_HIP(hipMemsetAsync(tier_started, 0, sizeof(int), stream));
        PM_HIP(hipEventRecord(h->tier_events[0], stream)); // whatever this chunk's fill waits for, the side streams wait for too
        i64 tier_groups = 0;
        for(size_t tier = 0; tier < cuts.size(); ++tier) {
          hipStream_t ts = h->tier_streams[tier];
          PM_HIP(hipStreamWaitEvent(ts, h->tier_events[0], 0));
          i64 groups = 0;
          const size_t tslot = h->tv_tier_chunk.size();
          if(timed) {
            while(h->tv_tier0.size() <= tslot) {
              hipEvent_t e0 = nullptr, e1 = nullptr;
              PM_HIP(hipEventCreate(&e0));
              h->tv_tier0.push_back(e0);
              PM_HIP(hipEventCreate(&e1));
              h->tv_tier1.push_back(e1);
            }
            PM_HIP(hipEventRecord(h->tv_tier0[tslot], ts));
          }
          PM_TRY(dp_launch_fill(h, at, cuts[tier] - at, tbw, traceback, ts, tier_started, &groups));
          if(timed) {
            PM_HIP(hipEventRecord(h->tv_tier1[tslot], ts));
            h->tv_tier_chunk.push_back((int)c);
          }
          tier_groups += groups;
          if(traceback) {
            PM_TRY(dp_launch_path(h, at, cuts[tier] - at, tbw, ts, true));
          }
          PM_HIP(hipEventRecord(h->tier_events[1 + tier], ts));
          at = cuts[tier];
        }
        tiers_end = at; // the chunk's own path kernel starts here
        if(gate_on && tier_groups > 0) {
          PM_TRY(dp_gate(stream, tier_started, (int)std::min<i64>(tier_groups, 0x7fffffff)));
        }
        PM_TRY(early_prepare(c, at, c_hi - at, stream));
        PM_TRY(dp_launch_fill(h, at, c_hi - at, tbw, traceback, stream, nullptr, nullptr, ew.base ? &ew.fl : nullptr, &ew.used));
        if(ew.used) {
          PM_TRY(early_begin(tbw));
        }
        tiers_joined = cuts.size();
        if(!traceback) { // no path kernel to wait for them: the chunk's fill stream does
          for(size_t tier = 0; tier < cuts.size(); ++tier) {
            PM_HIP(hipStreamWaitEvent(stream, h->tier_events[1 + tier], 0));
          }
        }
        at = c_hi;
      }
      if(at < c_hi) {
        const bool whole = at == c_lo; // one launch for the chunk: the next chunk's gate can count its workgroups
        i64 groups = 0;
        if(whole) {
          PM_TRY(early_prepare(c, at, c_hi - at, stream));
        }
        PM_TRY(dp_launch_fill(h, at, c_hi - at, tbw, traceback, stream, two_fills && whole ? started + c : nullptr, &groups,
                              whole && ew.base ? &ew.fl : nullptr, &ew.used));
        if(ew.used) {
          PM_TRY(early_begin(tbw));
        }
        if(two_fills && whole) {
          h->chunk_groups[c] = groups;
        }
      }
      prev_chunk = (long)c;
    }
    if(timed) {
      PM_HIP(hipEventRecord(h->tv_fill1[c], stream));
    }
    if(traceback) {
      if(pipelined) {
        PM_HIP(hipEventRecord(h->ev_fill[c], stream));
        PM_HIP(hipStreamWaitEvent(ps, h->ev_fill[c], 0));
      }
      if(timed) {
        PM_HIP(hipEventRecord(h->tv_path0[c], ps));
      }
      if(ew.used && ew.first == tiers_end && ew.n == h->chunk_first[c + 1] - tiers_end) {
        PM_TRY(early_finish(tbw, ps));
      }
      else {
        PM_TRY(dp_launch_path(h, tiers_end, h->chunk_first[c + 1] - tiers_end, tbw, ps));
      }
      if(timed) {
        PM_HIP(hipEventRecord(h->tv_path1[c], ps));
      }
      for(size_t tier = 0; tier < tiers_joined; ++tier) { // the chunk is done when its tiers' kernels are, too
        PM_HIP(hipStreamWaitEvent(ps, h->tier_events[1 + tier], 0));
      }
      if(pipelined) {
        PM_HIP(hipEventRecord(h->ev_path[c], ps));
      }
    }
  }
  stream = caller_stream;
  if(pipelined) { // the caller's stream is done when the last two path kernels are (each behind its chunk's fill kernel, and the
                  // path stream runs them in chunk order: every earlier kernel of either fill stream is behind them too)
    for(size_t c = nc >= (size_t)K ? nc - (size_t)K : 0; c < nc; ++c) {
      PM_HIP(hipStreamWaitEvent(stream, h->ev_path[c], 0));
    }
  }
  h->last_stream = stream;
  if(timed) {
    PM_HIP(hipStreamSynchronize(stream));
    float acc_fill = 0, acc_path = 0;
    std::vector<std::pair<float, float> > spans;
    size_t first_chunk = 0;
    while(first_chunk + 1 < nc && h->chunk_first[first_chunk + 1] - h->chunk_first[first_chunk] <= 0) {
      ++first_chunk;
    }
    for(size_t c = 0; c < nc; ++c) {
      if(h->chunk_first[c + 1] - h->chunk_first[c] <= 0) {
        continue;
      }
      float a = 0, b = 0;
      PM_HIP(hipEventElapsedTime(&a, h->tv_fill0[c], h->tv_fill1[c]));
      if(traceback) {
        PM_HIP(hipEventElapsedTime(&b, h->tv_path0[c], h->tv_path1[c]));
      }
      acc_fill += a;
      acc_path += b;
      // where the launch lies on the step's clock (the first chunk's start): for the time during which SOME fill kernel ran
      float at = 0;
      if(c != first_chunk) {
        PM_HIP(hipEventElapsedTime(&at, h->tv_fill0[first_chunk], h->tv_fill0[c]));
      }
      spans.push_back(std::make_pair(at, at + a));
    }
    // the tiers' fill kernels: launches of their own on their own streams, and the ones the step waits for
    for(size_t k = 0; k < h->tv_tier_chunk.size(); ++k) {
      float a = 0, at = 0;
      PM_HIP(hipEventElapsedTime(&a, h->tv_tier0[k], h->tv_tier1[k]));
      PM_HIP(hipEventElapsedTime(&at, h->tv_fill0[first_chunk], h->tv_tier0[k])); // (may be negative: a tier starts before its chunk's launch)
      acc_fill += a;
      spans.push_back(std::make_pair(at, at + a));
    }
    std::sort(spans.begin(), spans.end());
    float busy = 0, upto = -1e30f;
    for(const std::pair<float, float> &sp : spans) {
      const float lo = std::max(sp.first, upto);
      if(sp.second > lo) {
        busy += sp.second - lo;
        upto = sp.second;
      }
    }
    h->last_fill_busy_ms = busy;
    if(ms_fill) {
      *ms_fill = acc_fill;
    }
    if(ms_path) {
      *ms_path = acc_path;
    }
  }
  return PM_OK;
}
} // namespace pm

extern "C" {

int pm_dp_batch_run(pm_dp_batch_t *h, int traceback, void *hip_stream) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_run: null batch");
  }
  int rc = use_device(h->device);
  if(rc) {
    return rc;
  }
  return dp_run(h, (hipStream_t)hip_stream, traceback, nullptr, nullptr);
}

int pm_dp_batch_run_profiled(pm_dp_batch_t *h, int traceback, void *hip_stream, float *ms_fill, float *ms_traceback) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_run_profiled: null batch");
  }
  int rc = use_device(h->device);
  if(rc) {
    return rc;
  }
  float a = 0, b = 0;
  rc = dp_run(h, (hipStream_t)hip_stream, traceback, &a, &b);
  if(ms_fill) {
    *ms_fill = a;
  }
  if(ms_traceback) {
    *ms_traceback = b;
  }
  return rc;
}

int pm_dp_batch_fill_busy_ms(pm_dp_batch_t *h, float *ms) {
  if(!h || !ms) {
    return fail(PM_E_INVALID, "pm_dp_batch_fill_busy_ms: null argument");
  }
  *ms = h->last_fill_busy_ms;
  return PM_OK;
}

int pm_dp_batch_fetch(pm_dp_batch_t *h, int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_fetch: null batch");
  }
  int rc = use_device(h->device);
  if(rc) {
    return rc;
  }
  PM_HIP(hipStreamSynchronize(h->last_stream));
  {
    int perr = 0;
    PM_HIP(hipMemcpy(&perr, h->pipe_error.p, 4, hipMemcpyDeviceToHost));
    if(perr) {
      (void)dp_clear_pipe_error(h); // reported: the batch can be run again
      return fail(PM_E_HIP, "dp_fill_kernel: a stripe timed out waiting for its left neighbour (results invalid)");
    }
  }
  if(scores && h->n_pairs > 0) {
    PM_HIP(hipMemcpy(scores, h->scores.p, (size_t)h->n_pairs * 4, hipMemcpyDeviceToHost));
  }
  if(n_ops && h->n_pairs > 0) {
    PM_HIP(hipMemcpy(n_ops, h->n_ops.p, (size_t)h->n_pairs * 4, hipMemcpyDeviceToHost));
  }
  if(ops && h->total_a + h->total_b > 0) {
    PM_HIP(hipMemcpy(ops, h->ops.p, (size_t)(h->total_a + h->total_b), hipMemcpyDeviceToHost));
  }
  return PM_OK;
}

int pm_dp_batch_info(pm_dp_batch_t *h, int64_t *cells, int64_t *traceback_bytes_per_run, int64_t *input_bytes, int32_t *n_chunks) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_info: null batch");
  }
  if(cells) {
    *cells = h->cells;
  }
  if(traceback_bytes_per_run) {
    i64 words = 0;
    for(i64 k = 0; k < h->n_pairs; ++k) {
      i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      if(h->ckpt) { // what the fill kernel writes: the column checkpoints of the rows it is on + the row checkpoints
        words += dp_ck_bytes_written(la, lb, h->cols_per_lane, h->tail) / 4;
      }
      else {
        words += dp_tb_words(la, lb, h->cols_per_lane);
      }
    }
    *traceback_bytes_per_run = words * 4;
  }
  if(input_bytes) {
    // every stripe re-reads A's columns; B's columns are read once
    i64 bytes = h->total_b * 8;
    for(i64 k = 0; k < h->n_pairs; ++k) {
      i64 lb = h->off_b[k + 1] - h->off_b[k];
      bytes += dp_ck_stripes(lb, h->cols_per_lane, h->tail) * (h->off_a[k + 1] - h->off_a[k]) * 8;
    }
    *input_bytes = bytes;
  }
  if(n_chunks) {
    *n_chunks = (int32_t)h->chunk_tb.size();
  }
  return PM_OK;
}

int pm_dp_batch_geometry(pm_dp_batch_t *h, int64_t *padded_cells, int32_t *narrow_last_stripes, int64_t *fill_launches) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_geometry: null batch");
  }
  if(padded_cells) { // what the fill kernel computes: the columns its stripes cover x the rows of A
    i64 cells = 0;
    for(i64 k = 0; k < h->n_pairs; ++k) {
      const i64 la = h->off_a[k + 1] - h->off_a[k], lb = h->off_b[k + 1] - h->off_b[k];
      cells += la > 0 && lb > 0 ? la * dp_padded_cols(lb, h->cols_per_lane, h->tail) : 0;
    }
    *padded_cells = cells;
  }
  if(narrow_last_stripes) {
    *narrow_last_stripes = h->tail ? 1 : 0;
  }
  if(fill_launches) { // per pass: one per chunk and one per tier of a chunk
    i64 n = 0;
    for(size_t c = 0; c < h->chunk_tb.size(); ++c) {
      n += 1 + (c < h->chunk_tiers.size() ? (i64)h->chunk_tiers[c].size() : 0);
    }
    *fill_launches = n;
  }
  return PM_OK;
}

int pm_dp_batch_variant(pm_dp_batch_t *h, int32_t *cols_per_lane, int32_t *dot4, int32_t *valu_ops_per_cell) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_variant: null batch");
  }
  if(cols_per_lane) {
    *cols_per_lane = h->cols_per_lane;
  }
  if(dot4) {
    *dot4 = (h->dot4 ? 1 : 0) | (h->uni ? 2 : 0);
  }
  if(valu_ops_per_cell) {
    // per cell: column score (dot4 + dot2 = 2, or 3 x dot2), E 3, F 3, H + two decision bits 5, H - open 1;
    // without the decision bits (checkpoint mode): score 2 or 3, E 1, F 1, H 1, H - open 1
    *valu_ops_per_cell = h->ckpt ? (h->dot4 ? 2 : 3) + 4 : (h->dot4 ? 2 : 3) + 3 + 3 + 5 + 1; // UNI: one of them a full-rate add
  }
  return PM_OK;
}

int pm_dp_batch_chunks(pm_dp_batch_t *h, int64_t *first_position, int32_t capacity, int32_t *order) {
  if(!h || (!first_position && capacity > 0)) {
    return fail(PM_E_INVALID, "pm_dp_batch_chunks: null argument");
  }
  const int32_t nc = (int32_t)h->chunk_tb.size();
  for(int32_t c = 0; c <= nc && c < capacity; ++c) {
    first_position[c] = h->chunk_first[c];
  }
  if(order) {
    for(i64 q = 0; q < h->n_pairs; ++q) {
      order[q] = h->order[(size_t)q];
    }
  }
  return PM_OK;
}

int pm_dp_batch_path_mode(pm_dp_batch_t *h, int32_t *checkpoints, int32_t *block_rows, int32_t *block_columns) {
  if(!h) {
    return fail(PM_E_INVALID, "pm_dp_batch_path_mode: null batch");
  }
  if(checkpoints) {
    *checkpoints = h->ckpt ? 1 : 0;
  }
  if(block_rows) {
    *block_rows = h->ckpt ? DP_CK_R : 0;
  }
  if(block_columns) {
    *block_columns = h->ckpt ? DP_CK_W * h->cols_per_lane : 0;
  }
  return PM_OK;
}

void pm_dp_batch_destroy(pm_dp_batch_t *h) {
  if(!h) {
    return;
  }
  (void)hipSetDevice(h->device);
  delete h;
}

} // extern "C"
