// dp_walk.hip -- the path of every pair from the checkpoints the score-only fill kernel left (dp_internal.hpp), gfx950.
//
// NO REFERENCE COUNTERPART; checked against oracle/dp_oracle.c only.
//
// A group of LPP lanes (64 / LPP pairs per wavefront) walks one pair from (La, Lb) back to (0, 0).
// The walk needs the four decisions of the cells it visits; the fill kernel stored none.  So, block by block along the path:
//   1. the block is the C * DP_CK_W columns of one column group of the fill kernel x the DP_CK_R rows between two of that
//      group's row checkpoints;
//   2. its top edge ({H~ - gop, F~} of the columns) comes from the row checkpoint, its left edge ({H~ - gop, E~} per row)
//      from the column checkpoints of the group to the left; A's rows (expanded as the fill kernel's LDS ring holds them) and
//      the left edge are staged in LDS;
//   3. the group re-runs the recurrence inside the block as a small anti-diagonal wavefront (lane q owns BW / LPP columns,
//      neighbours exchange with v_mov_b32_dpp: row_shr:1 for groups of 16 lanes, which are DPP rows -- the group's first lane
//      keeps the instruction's `old` operand, the left edge -- and wave_shr:1 plus a select otherwise) with the fill kernel's own
//      hand-scheduled cell (dp_cell, 14 VALU instructions with the four decision bits), 4 bits per cell into LDS;
//   4. the walk follows the decisions until it leaves the block through its top or its left edge, a whole run at a time.
// For launches of few pairs the same kernel runs first in a second mode (band_mode 1) over (pair, column group) work items and
// computes the blocks around the straight line between the corners, all at once, into global memory; the walk (band_mode 2)
// then copies such a block's bits into LDS instead of recomputing it (dp_internal.hpp, "The band").
// Same arithmetic, in the same skewed coordinates, as dp_fill_kernel (V~[i][j] = V[i][j] + (i + j) * gap_extend), so every
// decision is the one the one-pass kernel would have stored.  A path crosses at most La / R + Lb / BW + 1 blocks, i.e. the
// walk recomputes about La * BW + Lb * R cells of the La * Lb: 2-6 % at kilobase lengths.  It is bound by VALU issue like the
// fill kernel (the groups of a launch keep every SIMD busy), so what counts is instructions per recomputed cell.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dp_internal.hpp"
#include "pm_internal.hpp"

namespace pm {

#define DPP_WAVE_SHR1 0x138

// lane l takes lane l-1's value (lane 0 keeps its own); the first lane of every group replaces what it gets
__device__ __forceinline__ int from_left_lane(int v) { return __builtin_amdgcn_update_dpp(v, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }
// the same inside rows of 16 lanes: the first lane of every row keeps `first` (row_shr:1 without bound_ctrl)
#define DPP_ROW_SHR1 0x111
__device__ __forceinline__ int from_left_in_row(int first, int v) { return __builtin_amdgcn_update_dpp(first, v, DPP_ROW_SHR1, 0xf, 0xf, false); }

template <int N> struct BitsWord { typedef unsigned type; };
template <> struct BitsWord<1> { typedef unsigned char type; };
template <> struct BitsWord<2> { typedef unsigned char type; };
template <> struct BitsWord<4> { typedef unsigned short type; };
template <bool DOT4> struct RowWord { typedef int4 type; };
template <> struct RowWord<true> { typedef int2 type; };

template <int C, int LPP, bool DOT4>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(C == 16 && LPP == 32 ? 5 : (C == 16 && LPP == 16 && DOT4 ? 4 : 1))))
dp_walk_kernel(const u64 *__restrict__ cols_a, const i64 *__restrict__ off_a, const u64 *__restrict__ cols_b, const i64 *__restrict__ off_b,
               const int *__restrict__ order, i64 n, const i64 *__restrict__ tb_off, const unsigned *__restrict__ ck, unsigned char *__restrict__ ops,
               int *__restrict__ n_ops, DpParamsD P, DpBand band, int band_mode, int tail, int urgent, DpEarly early) {
  // band_mode 0: the walk, every block recomputed; 1: no walk, the band's blocks computed and stored (work items = the band's
  // (pair, column group) list); 2: the walk, blocks inside the band read back, the others recomputed
  // A walk is a chain of blocks, each waiting for the one before.  The walk of a tier -- a few hundred long pairs beside the fill
  // kernel of the rest, and the step waits for its longest chain -- goes first at issue (`urgent`).  Not every walk: the path kernel
  // of a chunk beside the next chunk's fill kernel is throughput work like that one (12 500 pairs of 8 x 4 096 in two chunks:
  // 42.2 ms a step, 42.6 with the walk ahead at issue).
  if(urgent) {
    __builtin_amdgcn_s_setprio(3);
  }
  constexpr int R = DP_CK_R;
  constexpr int BW = C * DP_CK_W; // columns of a block
  constexpr int C2 = BW / LPP;    // columns per lane inside a block
  constexpr int G = 64 / LPP;     // pairs per wavefront
  static_assert(C2 >= 1 && C2 <= 8 && C % C2 == 0 && C2 * LPP == BW, "a lane's columns lie in one lane of the fill kernel");
  typedef typename BitsWord<C2>::type bits_t;
  // per group: A's rows of the block, expanded as the cell reads them (the int8 path reads 8 bytes of the 16), then the block's left
  // edge row by row -- one array, so that a step's two reads share an address register (the left edge at a constant distance); the
  // left edge of the row ABOVE the block apart
  constexpr int AW = DOT4 ? 1 : 2; // int2 per row of A
  __shared__ int2 sh_al[G][(AW + 1) * R];
  __shared__ int2 sh_top[G];
  __shared__ bits_t sh_bits[G][R][LPP];
  const int grp = threadIdx.x / LPP, q = threadIdx.x % LPP;
  constexpr int BLOCK_WORDS = R * LPP * (int)sizeof(bits_t) / 4;
  const int go = P.go, ge = P.ge, gop = go - ge;
  // the group's current pair
  bool have = false;
  i64 pair = 0;
  int ggw = 0, la = 0, lb = 0;
  const u64 *A = cols_a, *B = cols_b;
  const unsigned *ckp = ck;
  unsigned char *out = ops;
  int i = 0, j = 0, state = 0; // DP coordinates of the walk (cell (i, j) = row i-1 of A against column j-1 of B)
  int at = 0;
  auto take = [&](i64 p) {
    pair = p;
    const i64 a0 = off_a[p], b0 = off_b[p];
    la = (int)(off_a[p + 1] - a0);
    lb = (int)(off_b[p + 1] - b0);
    A = cols_a + a0;
    B = cols_b + b0;
    ckp = ck + tb_off[p];
    out = ops + a0 + b0;
    i = la;
    j = lb;
    state = 0;
    at = la + lb;
    have = true;
  };
  // beside / behind the fill kernel of the same launch (DpEarly, dp_internal.hpp): a group takes its pairs from the XCDs' lists
  const int my_xcc = early.mode ? (int)(__builtin_amdgcn_s_getreg(DP_GETREG_XCC_ID) & 7) : 0;
  int list_x = early.mode == 2 ? (int)((blockIdx.x * G + grp) & 7) : my_xcc; // the list the group draws from (mode 2: all, in turn)
  int ticket = -1;         // the entry of that list the group has drawn and not yet found written
  int lists_tried = 0;     // mode 2: lists found exhausted in a row
  int polls = 0;           // mode 1: looks at an entry that was not there yet
  bool group_done = early.mode == 0;
  if(early.mode == 1) { // is the fill kernel running beside this launch (DpEarly)?  Bounded: 2 048 looks, some 15 us apart
    int running = 0;
    for(int tries = 0; tries < 2048 && !running; ++tries) {
      if(threadIdx.x == 0) {
        running = __hip_atomic_load(early.n_filled + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                  __hip_atomic_load(early.n_filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= early.n_expected;
      }
      running = __shfl(running, 0);
      if(!running) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
      }
    }
    if(!running) {
      group_done = true; // no fill kernel beside us: nothing to wait for here
    }
  }
  {
    const i64 idx = (i64)blockIdx.x * G + grp;
    if(band_mode == 1) {
      if(idx < n) {
        ggw = band.work[2 * idx + 1];
        take(band.work[2 * idx]);
      }
    }
    else if(early.mode == 0 && idx < n) {
      take(order[idx]); // the launch's pairs in processing order
    }
  }

  for(int pass = 0;; ++pass) {
    bool live;
    if(band_mode == 1) { // the band's blocks of this column group, top to bottom; (i, j) = the block's bottom right cell
      if(pass == DP_BAND_BLOCKS) {
        break;
      }
      const int l0w = dp_group_lane0(C, dp_stripe_of_col(lb, C, tail, (i64)ggw * BW), ggw);
      const int kb = dp_band_row_block(la, lb, BW, ggw, l0w) - DP_BAND_BLOCKS / 2 + pass;
      const int top = kb * R - l0w;
      live = have && kb >= 0 && top < la && top + R >= 1;
      i = min(la, top + R);
      j = min(lb, (ggw + 1) * BW);
    }
    else {
      if(!(have && i > 0 && j > 0)) { // the same for all lanes of the group
        if(have) { // one profile exhausted: the rest is a single gap run
          const int rest = i + j;
          const unsigned char op = i == 0 ? 1 : 2;
          for(int k2 = q; k2 < rest; k2 += LPP) {
            out[at - 1 - k2] = op;
          }
          at -= rest;
          if(q == 0) {
            n_ops[pair] = la + lb - at;
          }
          have = false;
        }
      }
      if(early.mode != 0 && !have && !group_done) {
        // the group's next pair.  Draw an entry of the list (one lane draws, the group's first lane hands it round), then see whether
        // the fill kernel has written it; if not, the group sits this round out -- unless every pair has been published, in which
        // case an entry still empty will stay empty: the list has ended.
        // (the count of published pairs is ONE word that every waiting group of the chip would poll: it is looked at when a group is
        // about to draw, and every sixteenth time an entry is found empty -- polled every round by a thousand wavefronts it became a hot
        // spot in one memory channel that the fill kernel's checkpoint stores queued behind: the fill took 3.1 ms instead of 1.9)
        if(ticket < 0 && early.mode == 1 &&
           __hip_atomic_load(early.n_filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= early.n_expected) {
          // the fill kernel has ended (or is about to): what is left belongs to the launch behind it, which has the whole chip -- a
          // wavefront or two per SIMD drawing on would only drag the launch's end out (measured: 10 000 pairs of 2 x 1 000, 4.1 ms
          // with early walkers that drew until the lists were empty, profiles/r05_early_walk.txt)
          group_done = true;
        }
        else if(ticket < 0) {
          int t = 0;
          if(q == 0) {
            t = __hip_atomic_fetch_add(early.taken + list_x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          ticket = __shfl(t, grp * LPP);
        }
        int entry = 0;
        if(!group_done && ticket < early.n_expected) {
          entry = __hip_atomic_load(early.list + (i64)list_x * early.n_expected + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if(entry == 0 && (early.mode == 2 || (++polls & 15) == 0) &&
             __hip_atomic_load(early.n_filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= early.n_expected) {
            // all published: entries are written before the count goes up, so a second look is final
            entry = __hip_atomic_load(early.list + (i64)list_x * early.n_expected + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(entry == 0) {
              ticket = early.n_expected; // this list has ended
            }
          }
        }
        if(entry != 0) {
          take(order[entry - 1]);
          ticket = -1;
          lists_tried = 0;
        }
        else if(!group_done && ticket >= early.n_expected) {
          ticket = -1;
          if(early.mode == 2 && ++lists_tried < 8) {
            list_x = (list_x + 1) & 7; // behind the fill kernel: the next XCD's list
          }
          else {
            group_done = true;
          }
        }
      }
      if(!__any(have)) {
        if(__all(group_done)) {
          break;
        }
        if(early.mode == 1) { // waiting for the fill kernel: leave the SIMD, and the memory system, to it (some 15 us between two looks)
          __builtin_amdgcn_s_sleep(127);
          __builtin_amdgcn_s_sleep(127);
          __builtin_amdgcn_s_sleep(127);
          __builtin_amdgcn_s_sleep(127);
        }
        continue;
      }
      if(early.mode == 1 && __any(have && at == la + lb && i == la && j == lb)) {
        // a pair just taken: this CU's L1 may hold lines of an earlier pass over the same workspace
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      live = have && i > 0 && j > 0;
    }
    // ---- the block the walk is in: column group gg (lanes l0 .. of the stripe of the fill kernel it lies in), row block k
    const int gg = live ? (j - 1) / BW : 0;
    const DpStripe st = dp_stripe_of_col(lb, C, tail, (i64)gg * BW);
    const int l0 = dp_group_lane0(C, st, gg);
    const int k = live ? (i - 1 + l0) / R : 0;
    const int i0 = max(0, k * R - l0);
    // the walk only moves up and left: rows below row i - 1 and columns right of column j - 1 are never visited and no
    // cell that is depends on them, so the block is cut to rows [i0, i) and to the lanes up to the one that owns column j - 1
    const int i1 = live ? i : i0;
    const int nrows = i1 - i0;
    const int j0 = gg * BW;
    const int qmax = live ? (j - 1 - j0) / C2 : -1;
    // a block of the band is read back instead
    bool in_band = false;
    i64 band_block = 0;
    if(band_mode != 0 && live) {
      const int b = k - dp_band_row_block(la, lb, BW, gg, l0) + DP_BAND_BLOCKS / 2;
      in_band = band_mode == 2 && b >= 0 && b < DP_BAND_BLOCKS;
      band_block = band.off[pair] + (i64)gg * DP_BAND_BLOCKS + b;
    }
    const bool comp = live && !in_band;
    const bool mine = comp && q <= qmax; // this lane has cells to recompute
    // ---- this lane's columns: the cell's weight registers, as in dp_fill_kernel
    int w0[C2], w1[C2], w2[C2], hop[C2], f[C2];
#pragma unroll
    for(int c = 0; c < C2; ++c) {
      const int jc = j0 + q * C2 + c;
      const bool in = comp && jc < lb;
      dp_column_weights<DOT4>(in ? B[jc] : 0ull, in, P, w0[c], w1[c], w2[c]);
    }
    // ---- top edge: the state of the lane's columns after row i0 - 1 (the group's row checkpoint k - 1)
    const bool has_top = comp && k * R - l0 >= 1;
    if(has_top) {
#pragma unroll
      for(int c = 0; c < C2; ++c) {
        const int2 v = *reinterpret_cast<const int2 *>(ckp + dp_ck_row_word(la, lb, C, tail, st, k - 1, j0 + q * C2 + c));
        hop[c] = v.x;
        f[c] = v.y;
      }
    }
    else {
#pragma unroll
      for(int c = 0; c < C2; ++c) {
        hop[c] = -2 * gop; // H~[0][j] - gop, H~[0][j] = -gop for j >= 1
        f[c] = DP_NEG_INF;
      }
    }
    // ---- A's rows and the left edge (rows i0 - 1 .. i1 - 1) into LDS
    if(in_band) {
      const unsigned *src = band.bits + band_block * BLOCK_WORDS;
      unsigned *dst = reinterpret_cast<unsigned *>(&sh_bits[grp][0][0]);
#pragma unroll 4
      for(int w = q; w < BLOCK_WORDS; w += LPP) {
        dst[w] = src[w];
      }
    }
    if(comp) {
      for(int r = q; r < nrows; r += LPP) {
        const int4 v = dp_expand_row<DOT4>(A[i0 + r]);
        if constexpr(DOT4) {
          sh_al[grp][r] = make_int2(v.x, v.y);
        }
        else {
          *reinterpret_cast<int4 *>(&sh_al[grp][2 * r]) = v;
        }
      }
      // the lane left of the group: the last lane of the group to the left, in the stripe that group lies in
      const DpStripe stl = dp_stripe_of_col(lb, C, tail, gg > 0 ? (i64)gg * BW - 1 : 0);
      const int ll = gg > 0 ? (gg * BW - 1 - stl.jb) >> dp_cs_shift(stl.cs) : 0;
      for(int rr = q; rr <= nrows; rr += LPP) {
        const int row = i0 - 1 + rr; // row of A; -1 is the DP's row 0
        int2 v;
        if(gg == 0) {
          v = make_int2(row < 0 ? -gop : -2 * gop, DP_NEG_INF); // H~[row + 1][0] - gop
        }
        else if(row < 0) {
          v = make_int2(-2 * gop, DP_NEG_INF); // H~[0][j0] - gop, j0 >= 1
        }
        else {
          v = *reinterpret_cast<const int2 *>(ckp + dp_ck_col_word(la, gg - 1, (i64)row + ll));
        }
        if(rr == 0) {
          sh_top[grp] = v;
        }
        else {
          sh_al[grp][AW * R + rr - 1] = v;
        }
      }
    }
    __syncthreads();
    // ---- the block's cells, anti-diagonal over the group's lanes: at step u lane q is on row r = u - q of the block
    int e = DP_NEG_INF;
    bits_t *brow = &sh_bits[grp][0][q];
    // H~ - gop above-left of the lane's first column
    const int2 top = sh_top[grp];
    int diag_in = LPP == 16 ? from_left_in_row(top.x, hop[C2 - 1]) : from_left_lane(hop[C2 - 1]);
    if(LPP != 16 && q == 0) {
      diag_in = top.x;
    }
    // steps of this block: the longest of the wavefront's groups (rows it recomputes + its lanes' skew), a scalar
    int trip = 0;
    {
      const int need = comp && nrows > 0 ? nrows + qmax : 0; // the same in all lanes of a group
#pragma unroll
      for(int g = 0; g < G; ++g) {
        trip = max(trip, __builtin_amdgcn_readlane(need, g * LPP));
      }
      trip = __builtin_amdgcn_readfirstlane(trip); // (the loop counter on the scalar unit)
    }
    const unsigned rows_mine = mine ? (unsigned)nrows : 0u; // (unsigned)r < rows_mine: this lane is on one of its rows
    // A's row and the left edge are read one step ahead, so that the reads' latency is not on the step's path; the row index
    // wraps instead of being clamped or predicated (a row that is read and not used costs nothing), and every lane reads the
    // left edge although only the group's first uses it (no exec juggling).  Two steps per iteration (an odd trip count runs one step
    // more: no lane is on a row in it), so that "this step's" and "next step's" registers swap roles instead of being copied.
    int r = -q;
    const char *const rows = reinterpret_cast<const char *>(&sh_al[grp][0]);
    unsigned off = (unsigned)(r & (R - 1)) * 8u; // byte offset of the lane's row among the block's R
    auto fetch = [&](int4 &a_to, int2 &l_to) __attribute__((always_inline)) {
      if constexpr(DOT4) {
        const int2 t2 = *reinterpret_cast<const int2 *>(rows + off);
        a_to.x = t2.x;
        a_to.y = t2.y;
      }
      else {
        a_to = *reinterpret_cast<const int4 *>(rows + 2 * off);
      }
      l_to = *reinterpret_cast<const int2 *>(rows + AW * R * 8 + off);
    };
    auto step = [&](const int4 &a, const int2 &lb_, int4 &a_to, int2 &l_to) __attribute__((always_inline)) {
      int ho_in, e_in;
      if(LPP == 16) { // a DPP row is a group: its first lane keeps the `old` operand, the left edge
        ho_in = from_left_in_row(lb_.x, hop[C2 - 1]);
        e_in = from_left_in_row(lb_.y, e);
      }
      else {
        ho_in = from_left_lane(hop[C2 - 1]);
        e_in = from_left_lane(e);
        if(q == 0) {
          ho_in = lb_.x;
          e_in = lb_.y;
        }
      }
      off = (off + 8u) & (unsigned)(R * 8 - 1);
      fetch(a_to, l_to);
      if((unsigned)r < rows_mine) {
        const int ax = a.x, ay = a.y, az = DOT4 ? 0 : a.z;
        unsigned acc;
        asm volatile("" : "=v"(acc)); // every bit that is read gets shifted in below
        e = e_in;
        int dd[2];
        asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(dd[0]) : "v"(ay), "v"(w2[0]), "v"(diag_in));
#pragma unroll
        for(int c = 0; c < C2; ++c) {
          const int hl = c == 0 ? ho_in : hop[c == 0 ? 0 : c - 1];
          if(c == C2 - 1) {
            dp_cell<true, true, DOT4>(dd[c & 1], dd[(c + 1) & 1], e, f[c], hop[c], acc, hl, ax, ay, DOT4 ? ax : az, w0[c],
                                      DOT4 ? w0[c] : w1[c], 0, gop);
          }
          else {
            dp_cell<true, false, DOT4>(dd[c & 1], dd[(c + 1) & 1], e, f[c], hop[c], acc, hl, ax, ay, DOT4 ? ax : az, w0[c],
                                       DOT4 ? w0[c] : w1[c], w2[c + 1 < C2 ? c + 1 : c], gop);
          }
        }
        diag_in = ho_in;
        brow[r * LPP] = (bits_t)acc;
      }
      ++r;
    };
    int4 row_a = make_int4(0, 0, 0, 0), row_b = make_int4(0, 0, 0, 0);
    int2 left_a, left_b;
    fetch(row_a, left_a);
    for(int u = 0; u < trip; u += 2) {
      step(row_a, left_a, row_b, left_b);
      step(row_b, left_b, row_a, left_a);
    }
    __syncthreads();
    if(band_mode == 1) {
      if(live) {
        unsigned *dst = band.bits + band_block * BLOCK_WORDS;
        const unsigned *src = reinterpret_cast<const unsigned *>(&sh_bits[grp][0][0]);
#pragma unroll 4
        for(int w = q; w < BLOCK_WORDS; w += LPP) {
          dst[w] = src[w];
        }
      }
      __syncthreads();
      continue;
    }
    // ---- follow the decisions until the walk leaves the block.  The group's lanes look at the next LPP cells along the
    // current direction (diagonal in state H, along the row in E, along the column in F); a ballot finds how far the run
    // goes and the whole run is emitted at once, lane q writing op q of it.
    bool inb = live;
    while(__any(inb)) {
      const int di = state != 1, dj = state != 2;
      const int ci = i - q * di, cj = j - q * dj;
      const bool cv = inb && ci - 1 >= i0 && cj - 1 >= j0; // i0, j0 >= 0: also inside the DP
      unsigned nib = 0;
      if(cv) {
        const int cc = cj - 1 - j0;
        const unsigned word = sh_bits[grp][ci - 1 - i0][cc / C2];
        nib = (word >> (4 * (C2 - 1 - cc % C2))) & 15u;
      }
      const unsigned keep = state == 0 ? (~nib & 2u) : (state == 1 ? (nib & 8u) : (nib & 4u));
      const int sh = grp * LPP;
      const u64 gmask = LPP == 64 ? ~0ull : ((1ull << (LPP & 63)) - 1);
      const u64 gc = (__ballot(cv && keep != 0) >> sh) & gmask;
      const u64 gv = (__ballot(cv) >> sh) & gmask;
      const unsigned gf = (unsigned)(__ballot((nib & 1u) != 0) >> sh) & 1u;
      const int run = gc == gmask ? LPP : __builtin_ctzll(~gc);
      if(inb) {
        if(state == 0) {
          if(run == 0) { // the cell itself is not diagonal: switch to the gap state it names, no move
            state = gf ? 2 : 1;
          }
          else {
            if(q < run) {
              out[at - 1 - q] = 0;
            }
            at -= run;
            i -= run;
            j -= run;
          }
        }
        else {
          // the cells that extend are consumed in this state; the first one that does not is consumed too and returns
          // the walk to H (if it lies in this block; otherwise the next block goes on in this state)
          int take = run, next = state;
          if(run < LPP && ((gv >> run) & 1ull)) {
            take = run + 1;
            next = 0;
          }
          if(q < take) {
            out[at - 1 - q] = (unsigned char)state; // 1 = I (state E), 2 = D (state F)
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
        inb = i > 0 && j > 0 && i - 1 >= i0 && j - 1 >= j0;
      }
    }
    __syncthreads();
  }
}

template <int C, int LPP, bool DOT4>
static int launch_walk(const u64 *cols_a, const i64 *off_a, const u64 *cols_b, const i64 *off_b, const int *order, i64 n, const i64 *tb_off,
                       const unsigned *ck, unsigned char *ops, int *n_ops, const DpParamsD &P, const DpBand &band, int tail, int urgent, hipStream_t stream,
                       const DpEarly *early, unsigned early_groups) {
  constexpr int G = 64 / LPP;
  const DpEarly none = {0, 0, nullptr, nullptr, nullptr};
  if(early && early->mode != 0) { // beside (1) or behind (2) the fill kernel of the same launch: groups that take pairs from the XCDs' lists
    const unsigned eb = std::max(1u, (early_groups + G - 1) / G);
    dp_walk_kernel<C, LPP, DOT4><<<eb, 64, 0, stream>>>(cols_a, off_a, cols_b, off_b, order, n, tb_off, ck, ops, n_ops, P, band, 0, tail, urgent, *early);
    PM_HIP(hipGetLastError());
    return PM_OK;
  }
  const unsigned blocks = (unsigned)((n + G - 1) / G);
  const bool with_band = band.work != nullptr && band.n_work > 0;
  if(with_band) {
    const unsigned bblocks = (unsigned)((band.n_work + G - 1) / G);
    dp_walk_kernel<C, LPP, DOT4><<<bblocks, 64, 0, stream>>>(cols_a, off_a, cols_b, off_b, order, band.n_work, tb_off, ck, ops, n_ops, P, band, 1,
                                                             tail, urgent, none);
    PM_HIP(hipGetLastError());
  }
  dp_walk_kernel<C, LPP, DOT4><<<blocks, 64, 0, stream>>>(cols_a, off_a, cols_b, off_b, order, n, tb_off, ck, ops, n_ops, P, band, with_band ? 2 : 0,
                                                          tail, urgent, none);
  PM_HIP(hipGetLastError());
  return PM_OK;
}

int dp_launch_walk(int cols_per_lane, int lanes_per_pair, bool dot4, const u64 *cols_a, const i64 *off_a, const u64 *cols_b, const i64 *off_b,
                   const int *order, i64 n, const i64 *tb_off, const unsigned *ck, unsigned char *ops, int *n_ops, const DpParamsD &P,
                   const DpBand &band, int tail, int urgent, hipStream_t stream, const DpEarly *early, unsigned early_groups) {
  if(n <= 0) {
    return PM_OK;
  }
  // lanes per pair: a lane owns BW / LPP of the block's BW = C * DP_CK_W columns, 1 to 8 of them, inside one lane of the fill kernel
#define WALK(CC, LL)                                                                                                        \
  if constexpr((CC * DP_CK_W) % LL == 0 && (CC * DP_CK_W) / LL >= 1 && (CC * DP_CK_W) / LL <= 8 && CC % ((CC * DP_CK_W) / LL) == 0) { \
    if(lanes_per_pair == LL) {                                                                                              \
      return dot4 ? launch_walk<CC, LL, true>(cols_a, off_a, cols_b, off_b, order, n, tb_off, ck, ops, n_ops, P, band, tail, urgent, stream, early, early_groups) \
                  : launch_walk<CC, LL, false>(cols_a, off_a, cols_b, off_b, order, n, tb_off, ck, ops, n_ops, P, band, tail, urgent, stream, early, early_groups); \
    }                                                                                                                       \
  }
  if(cols_per_lane == 16) {
    WALK(16, 2) WALK(16, 4) WALK(16, 8) WALK(16, 16) WALK(16, 32) WALK(16, 64)
  }
  else {
    WALK(8, 2) WALK(8, 4) WALK(8, 8) WALK(8, 16) WALK(8, 32)
  }
#undef WALK
  return fail(PM_E_INVALID, "dp_launch_walk: lanes per pair not available for this block width");
}

// the group sizes dp_launch_walk accepts for a fill kernel with `cols_per_lane` columns per lane
bool dp_walk_lanes_ok(int cols_per_lane, int lanes_per_pair) {
  const int bw = cols_per_lane * DP_CK_W;
  if(lanes_per_pair < 2 || lanes_per_pair > 64 || (lanes_per_pair & (lanes_per_pair - 1)) || bw % lanes_per_pair) {
    return false;
  }
  const int c2 = bw / lanes_per_pair;
  return c2 >= 1 && c2 <= 8 && cols_per_lane % c2 == 0 && !(cols_per_lane == 8 && lanes_per_pair == 64);
}

} // namespace pm
