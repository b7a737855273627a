// translate_job.hip -- kernels, device-resident batch ("job") and C ABI of the translate path.
//
// Data layout in HBM (all int64 unless noted; R2 = {start,end} 16-byte pairs so a gap is one dwordx4 load):
//   rows   : range[R2 n], length[n], gap_off[n+1], gaps[R2 G], pre[G+n], bad[int n]          (per side)
//   deltas : ref[R2 n], qry[R2 n], ref_off[n+1], qry_off[n+1],
//            {ref,qry}_gaps[2][R2 G], {ref,qry}_pre[2][G+n]   (orientation 0 as read, 1 reversed), bad[int n]
//   units  : delta[int U], left[int U], right[int U]
//   out    : status[int U]; per LIVE unit (the filter pass's survivors, in unit order: live index k) cnt_ent[k], cnt_off[k] -> exclusive
//            scans ent_off_l[n_live+1], off_off[n_live+1]; per unit again ent_off[U+1] (what pm_job_fetch hands out), expanded from those;
//            entries[E] + offsets[O]: 32-byte Entry32 records + int offsets for a job on the int tables (pm_job_fetch widens them),
//            pm_entry_t + int64 else (translate_device.hpp, EntRecT)
// Kernels: prepare_rows / prepare_deltas (once per job), translate_filter + the live list, translate_count, the prefix sums of
// the counts (flag_* / count_* kernels below), translate_emit.  One lane per unit (the merge is a sequential state machine; the batch supplies the
// parallelism), 64-lane workgroups so a 100 k-unit batch spreads over all 256 CUs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>

#include <rocprim/rocprim.hpp>

#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "pm_internal.hpp"
#include "translate_device.hpp"
#include "translate_store.hpp"

namespace pm {

// ------------------------------------------------------------------ kernels

__device__ __forceinline__ unsigned long long mag(i64 v) { return (unsigned long long)(v < 0 ? -v : v); }

// One thread per row: interleave the gap list, build the prefix table, validate ordering.  Instantiated for int64 (the
// tables every job has; also validates and records the largest magnitude in the tables, which decides whether the
// job may use the int tables) and for int (the same tables narrowed: range32/length32/gaps32/pre32).
template <typename I>
__global__ void prepare_rows_kernel(i64 n, const i64 *start, const i64 *end, const i64 *length, const i64 *gap_off, const i64 *gs,
                                    const i64 *ge, R2T<I> *range, I *length_out, R2T<I> *gaps, I *pre, int *bad,
                                    unsigned long long *maxabs) {
  i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(r >= n) {
    return;
  }
  range[r] = R2T<I>{(I)start[r], (I)end[r]};
  if(length_out) {
    length_out[r] = (I)length[r];
  }
  i64 o = gap_off[r], m = gap_off[r + 1] - o;
  I *p = pre + o + r;
  i64 acc = 0, prev_end = 0;
  int flag = 0;
  // two ORs of magnitudes (an OR is below 2^k exactly when every one of them is): sequence POSITIONS, and everything counted in
  // COLUMNS -- with the row's span, which bounds every difference of two positions inside it (translate_device.hpp, type P)
  const unsigned long long big_pos = mag(start[r]) | mag(end[r]);
  unsigned long long big = mag(length[r]) | mag(rlen(R2{start[r], end[r]}));
  for(i64 k = 0; k < m; ++k) {
    R2 g{gs[o + k], ge[o + k]};
    if(g.s > g.e || (k > 0 && g.s <= prev_end)) {
      flag = 1;
    }
    prev_end = g.e;
    big |= mag(g.s) | mag(g.e);
    gaps[o + k] = R2T<I>{(I)g.s, (I)g.e};
    p[k] = (I)acc;
    acc += rlen(g);
    big |= mag(acc);
  }
  p[m] = (I)acc;
  if(bad) {
    bad[r] = flag;
  }
  if(maxabs) {
    atomicOr(maxabs, big_pos);
    atomicOr(maxabs + 1, big);
  }
}

// One thread per (entry, strand): both orientations of one gap list (m_delta.cc:94-146 for the reversed one).
// Instantiated like prepare_rows_kernel (bad/maxabs only in the int64 run).
template <typename I>
__global__ void prepare_deltas_kernel(i64 n, const i64 *rs, const i64 *re, const i64 *gap_off, const i64 *gs, const i64 *ge,
                                      R2T<I> *range, R2T<I> *g_fwd, I *pre_fwd, R2T<I> *g_rev, I *pre_rev, int *bad,
                                      unsigned long long *maxabs) {
  i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(d >= n) {
    return;
  }
  R2 rg{rs[d], re[d]};
  range[d] = R2T<I>{(I)rg.s, (I)rg.e};
  i64 o = gap_off[d], m = gap_off[d + 1] - o;
  I *pf = pre_fwd + o + d;
  I *pr = pre_rev + o + d;
  i64 acc = 0, prev_end = 0;
  int flag = 0;
  const unsigned long long big_pos = mag(rg.s) | mag(rg.e);
  unsigned long long big = mag(rlen(rg));
  for(i64 k = 0; k < m; ++k) {
    R2 g{gs[o + k], ge[o + k]};
    if(g.s > g.e || (k > 0 && g.s <= prev_end)) {
      flag = 1;
    }
    prev_end = g.e;
    big |= mag(g.s) | mag(g.e);
    g_fwd[o + k] = R2T<I>{(I)g.s, (I)g.e};
    pf[k] = (I)acc;
    acc += rlen(g);
    big |= mag(acc);
  }
  pf[m] = (I)acc;
  i64 columns = rlen(rg) + acc;
  i64 racc = 0;
  for(i64 k = 0; k < m; ++k) {
    R2 g{gs[o + (m - 1 - k)], ge[o + (m - 1 - k)]};
    R2 mg{columns - g.e + 1, columns - g.s + 1};
    g_rev[o + k] = R2T<I>{(I)mg.s, (I)mg.e};
    pr[k] = (I)racc;
    racc += rlen(mg);
  }
  pr[m] = (I)racc;
  if(bad && flag) {
    atomicOr(bad + d, 1);
  }
  if(maxabs) {
    atomicOr(maxabs, big_pos);
    atomicOr(maxabs + 1, big);
  }
}

// Which 64 units a workgroup takes.  Workgroups go to the eight XCDs in turn (workgroup b to XCD b mod 8) and every XCD has an L2 of its
// own: with workgroup b on units [64 b, 64 b + 64) every XCD walks the whole unit list -- and fetches every delta entry's gap lists and
// prefix tables into its own L2 (the unit list is ordered by entry; round 5's request counters: the emit pass read 145 MB from the
// memory side, all in 128-byte lines, for 69 MB of algorithmic reads).  Here XCD x takes the x-th eighth of the `nb` chunks of 64, so an
// entry's tables are fetched by one L2 (two at a seam).  A bijection of [0, nb); workgroups b >= nb have no chunk (returns nb or more).
__device__ __forceinline__ unsigned xcd_chunk(unsigned b, unsigned nb) {
  if(b >= nb) {
    return b;
  }
  const unsigned q = nb / 8, r = nb % 8, x = b % 8;
  return x * q + (x < r ? x : r) + b / 8;
}

// Filter pass: most (entry, left row, right row) triples the reference's loops visit end at the first overlap test
// (m_translate.cc:513).  One lane per unit runs just that prefix; the survivors ("live" units) are compacted so that
// the count and emit passes run on dense wavefronts instead of waiting for the few long lanes of every wavefront.
template <typename I, typename P>
__global__ void __launch_bounds__(64)
translate_filter_kernel(RowsT<I, P> left, RowsT<I, P> right, DeltasT<I, P> ds, i64 n_units, const int *u_delta, const int *u_left, const int *u_right,
                        int *status, i64 *cnt_ent, i64 *cnt_off, int *live_flag) {
  i64 u = (i64)xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if(u >= n_units) {
    return;
  }
  PVT<I, P> lp, rp, dr, dq;
  R2T<I> cols;
  bool live;
  int orientation;
  int st = unit_prefix(left, right, ds, u_delta[u], u_left[u], u_right[u], lp, rp, dr, dq, cols, live, orientation);
  status[u] = st;
  // (cnt_ent[u] and cnt_off[u] of a unit that is not live stay what pm_job_create set them to, 0: nothing ever writes them -- the
  // count pass writes the live units' -- and which units are live follows from the job's tables alone)
  live_flag[u] = (!st && live) ? 1 : 0;
}

__global__ void scatter_live_kernel(i64 n_units, const int *live_flag, const int *live_pos, int *live_units) {
  i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(u < n_units && live_flag[u]) {
    live_units[live_pos[u]] = (int)u;
  }
}

// ---- The step's prefix sums.  Between its passes a step needs three exclusive sums over the units (the live flags; the entries and
// the offsets each unit writes).  As three library scans they were seven launches and 60 us of a 320 us step, each a chained
// look-back over some 1.4 M elements (profiles/r04_translate_ablation.txt).  Here: tiles of 2 048 elements; one kernel adds up every
// tile, the next gives every tile the sum of the tiles before it (a block adds up to SCAN_MAX_TILES partial sums: a few KB out of
// the L2) and scans the tile; the two count arrays go through together, and the flags' kernel also writes the list of live units
// (scatter_live_kernel's work).  Jobs of more than SCAN_MAX_TILES tiles (8.4 M units) keep the library scans.
constexpr int SCAN_THREADS = 256, SCAN_PER = 8, SCAN_TILE = SCAN_THREADS * SCAN_PER, SCAN_MAX_TILES = 4096;
struct Sum2 {
  i64 a, b;
};
__device__ __forceinline__ int scan_add(int x, int y) { return x + y; }
__device__ __forceinline__ Sum2 scan_add(Sum2 x, Sum2 y) { return Sum2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ int scan_up(int v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ Sum2 scan_up(Sum2 v, int d) { return Sum2{__shfl_up(v.a, d), __shfl_up(v.b, d)}; }
__device__ __forceinline__ int scan_zero(int) { return 0; }
__device__ __forceinline__ Sum2 scan_zero(Sum2) { return Sum2{0, 0}; }
// exclusive sum of v over the block's SCAN_THREADS threads; total = the block's sum.  sh: SCAN_THREADS / 64 slots
template <typename T>
__device__ __forceinline__ T block_exclusive(T v, T *sh, T &total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for(int d = 1; d < 64; d <<= 1) {
    const T o = scan_up(inc, d);
    if(lane >= d) {
      inc = scan_add(inc, o);
    }
  }
  __syncthreads(); // sh may still be read from an earlier call
  if(lane == 63) {
    sh[wv] = inc;
  }
  __syncthreads();
  T before = scan_zero(v);
  total = scan_zero(v);
#pragma unroll
  for(int w = 0; w < SCAN_THREADS / 64; ++w) {
    if(w < wv) {
      before = scan_add(before, sh[w]);
    }
    total = scan_add(total, sh[w]);
  }
  T exc = scan_up(inc, 1);
  if(lane == 0) {
    exc = scan_zero(v);
  }
  return scan_add(before, exc);
}
// the sum of the tiles before this block's
template <typename T>
__device__ __forceinline__ T tiles_before(const T *partial, T *sh) {
  T acc = scan_zero(partial[0]);
  for(int k = threadIdx.x; k < (int)blockIdx.x; k += SCAN_THREADS) {
    acc = scan_add(acc, partial[k]);
  }
  T total;
  (void)block_exclusive(acc, sh, total);
  return total;
}

__global__ void __launch_bounds__(SCAN_THREADS) flag_tile_sums_kernel(i64 n, const int *__restrict__ flag, int *__restrict__ partial) {
  __shared__ int sh[SCAN_THREADS / 64];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  int acc = 0;
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    acc += k < n ? flag[k] : 0;
  }
  int total;
  (void)block_exclusive(acc, sh, total);
  if(threadIdx.x == 0) {
    partial[blockIdx.x] = total;
  }
}
// live_pos[u] = live flags before u (u = 0 .. n_units: the last one is the number of live units); live_units = the live ones in order
// (the tile through LDS both ways, as in count_scan_kernel below)
__device__ __forceinline__ int scan_pad(int x) { return x + (x >> 3); }
__global__ void __launch_bounds__(SCAN_THREADS)
flag_scan_scatter_kernel(i64 n, i64 n_units, const int *__restrict__ flag, const int *__restrict__ partial, int *__restrict__ live_pos,
                         int *__restrict__ live_units) {
  __shared__ int sh[SCAN_THREADS / 64];
  __shared__ int sh_f[SCAN_TILE + SCAN_TILE / 8];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  int g[SCAN_PER];
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) { // asked for before anything waits
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    g[i] = k < n ? flag[k] : 0;
  }
  const int before = tiles_before(partial, sh);
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    sh_f[scan_pad(i * SCAN_THREADS + (int)threadIdx.x)] = g[i];
  }
  __syncthreads();
  const i64 first = base + (i64)threadIdx.x * SCAN_PER;
  int f[SCAN_PER], sum = 0;
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    f[i] = sh_f[scan_pad((int)threadIdx.x * SCAN_PER + i)];
    sum += f[i];
  }
  int total;
  int at = before + block_exclusive(sum, sh, total);
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const i64 u = first + i;
    sh_f[scan_pad((int)threadIdx.x * SCAN_PER + i)] = at; // (the place the thread has just read: nobody else's)
    if(f[i] && u < n_units && u < n) {
      live_units[at] = (int)u;
    }
    at += f[i];
  }
  __syncthreads();
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    if(k < n) {
      live_pos[k] = sh_f[scan_pad(i * SCAN_THREADS + (int)threadIdx.x)];
    }
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
count_tile_sums_kernel(i64 n, const i64 *__restrict__ cnt_ent, const i64 *__restrict__ cnt_off, Sum2 *__restrict__ partial) {
  __shared__ Sum2 sh[SCAN_THREADS / 64];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  Sum2 acc{0, 0};
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    if(k < n) {
      acc.a += cnt_ent[k];
      acc.b += cnt_off[k];
    }
  }
  Sum2 total;
  (void)block_exclusive(acc, sh, total);
  if(threadIdx.x == 0) {
    partial[blockIdx.x] = total;
  }
}
// (the tile goes through LDS both ways: the global loads and stores are contiguous across the wavefront, a thread's own eight
// consecutive elements come out of LDS, one element of padding per eight keeping its reads off each other's banks)
__global__ void __launch_bounds__(SCAN_THREADS)
count_scan_kernel(i64 n, const i64 *__restrict__ cnt_ent, const i64 *__restrict__ cnt_off, const Sum2 *__restrict__ partial,
                  i64 *__restrict__ ent_off, i64 *__restrict__ off_off) {
  __shared__ Sum2 sh[SCAN_THREADS / 64];
  __shared__ i64 sh_e[SCAN_TILE + SCAN_TILE / 8], sh_o[SCAN_TILE + SCAN_TILE / 8];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  i64 ge[SCAN_PER], go[SCAN_PER];
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) { // asked for before anything waits
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    ge[i] = k < n ? cnt_ent[k] : 0;
    go[i] = k < n ? cnt_off[k] : 0;
  }
  const Sum2 before = tiles_before(partial, sh);
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const int x = scan_pad(i * SCAN_THREADS + (int)threadIdx.x);
    sh_e[x] = ge[i];
    sh_o[x] = go[i];
  }
  __syncthreads();
  i64 ce[SCAN_PER], co[SCAN_PER];
  Sum2 sum{0, 0};
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const int x = scan_pad((int)threadIdx.x * SCAN_PER + i);
    ce[i] = sh_e[x];
    co[i] = sh_o[x];
    sum.a += ce[i];
    sum.b += co[i];
  }
  Sum2 total;
  Sum2 at = scan_add(before, block_exclusive(sum, sh, total));
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) { // (a thread rewrites the places it has just read: nobody else's)
    const int x = scan_pad((int)threadIdx.x * SCAN_PER + i);
    sh_e[x] = at.a;
    sh_o[x] = at.b;
    at.a += ce[i];
    at.b += co[i];
  }
  __syncthreads();
#pragma unroll
  for(int i = 0; i < SCAN_PER; ++i) {
    const i64 k = base + i * SCAN_THREADS + threadIdx.x;
    if(k < n) {
      const int x = scan_pad(i * SCAN_THREADS + (int)threadIdx.x);
      ent_off[k] = sh_e[x];
      off_off[k] = sh_o[x];
    }
  }
}

// The counts and their sums are held per LIVE unit (round 5): the count pass stores a wavefront's 64 counts as one run instead of 64
// 8-byte stores scattered over the unit list (a third of the units are live: nearly every sector of the per-unit arrays was written for
// a quarter of its bytes), the sums are over a third of the elements, and the emit pass reads its lanes' offsets as runs.  What leaves
// the library per UNIT -- pm_job_fetch's unit_entry_off, the text pass's unit boundaries -- is this: the live units before unit u are
// live_pos[u] (the filter pass's exclusive sum; live_pos[U] = all of them), and units that are not live hold nothing, so
// ent_off[u] = ent_off_l[live_pos[u]].  totals = {entries, offsets} of the job.
__global__ void expand_offsets_kernel(i64 n_units, const int *__restrict__ live_pos, const i64 *__restrict__ ent_off_l, const i64 *__restrict__ off_off_l,
                                      i64 *__restrict__ ent_off, i64 *__restrict__ totals) {
  const i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(u <= n_units) {
    const int k = live_pos[u];
    const i64 e = ent_off_l[k];
    ent_off[u] = e;
    if(u == n_units) {
      totals[0] = e;
      totals[1] = off_off_l[k];
    }
  }
}

// The saved merge states live in HBM field by field (13 coordinate arrays, then 10 int arrays, each n_live long), so
// that the 64 lanes of a wavefront, which hold consecutive live units, store and load every field as one contiguous run
// instead of 64 separate records.
static_assert(sizeof(UnitStateT<i64>) == 13 * 8 + 14 * 4 && sizeof(UnitStateT<int>) == 26 * 4,
              "state_store/state_load lay UnitState out as 13 coordinate + 13 int fields (the int64 struct ends in 4 bytes of padding)");
template <typename I>
__device__ __forceinline__ void state_store(void *base, i64 n, i64 k, const UnitStateT<I> &s) {
  I *w = reinterpret_cast<I *>(base);
  int *iw = reinterpret_cast<int *>(w + 13 * n);
#pragma unroll
  for(int q = 0; q < 4; ++q) {
    w[(0 + q) * n + k] = s.ws[q];
    w[(4 + q) * n + k] = s.we[q];
    iw[(0 + q) * n + k] = s.lo[q];
    iw[(4 + q) * n + k] = s.n[q];
  }
  w[8 * n + k] = s.ref_start;
  w[9 * n + k] = s.query_start;
  w[10 * n + k] = s.column;
  w[11 * n + k] = s.last_column;
  w[12 * n + k] = s.query_columns;
  iw[8 * n + k] = s.orientation;
  iw[9 * n + k] = s.mirrored;
  iw[10 * n + k] = s.delta;
  iw[11 * n + k] = s.left;
  iw[12 * n + k] = s.right;
}

template <typename I>
__device__ __forceinline__ void state_load(const void *base, i64 n, i64 k, UnitStateT<I> &s) {
  const I *w = reinterpret_cast<const I *>(base);
  const int *iw = reinterpret_cast<const int *>(w + 13 * n);
#pragma unroll
  for(int q = 0; q < 4; ++q) {
    s.ws[q] = w[(0 + q) * n + k];
    s.we[q] = w[(4 + q) * n + k];
    s.lo[q] = iw[(0 + q) * n + k];
    s.n[q] = iw[(4 + q) * n + k];
  }
  s.ref_start = w[8 * n + k];
  s.query_start = w[9 * n + k];
  s.column = w[10 * n + k];
  s.last_column = w[11 * n + k];
  s.query_columns = w[12 * n + k];
  s.orientation = iw[8 * n + k];
  s.mirrored = iw[9 * n + k];
  s.delta = iw[10 * n + k];
  s.left = iw[11 * n + k];
  s.right = iw[12 * n + k];
}

// amdgpu_waves_per_eu(4): keep the register allocation at <= 128 VGPRs (4 waves per SIMD); the kernel is bound by the
// latency of dependent loads, so resident waves matter more than a few spare registers.
// I = int (the narrow tables): about 70-80 VGPRs, 6-7 waves per SIMD.  `narrow_trip` is set when a unit's int merge
// left its checked range (PM_ST_NARROW): pm_job_create then switches the job to the int64 tables.
// The int64 emit pass needs 162 VGPRs: held to 128 it spilled 38 of them to scratch, in the merge loop -- three waves per SIMD without
// spills are the faster (the bench job on the int64 tables 0.470 -> 0.396 ms a step; the count pass, 5 spills, is the same either way).
template <bool EMIT, typename I, typename P>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(I) == 8 && EMIT ? 3 : 4, 8)))
translate_kernel(RowsT<I, P> left, RowsT<I, P> right, DeltasT<I, P> ds, i64 n_units, const int *u_delta, const int *u_left, const int *u_right,
                 const int *live_units, const int *live_pos, int *status, i64 *cnt_ent, i64 *cnt_off, const i64 *ent_off,
                 const i64 *off_off, typename EntRecT<I>::type *entries, I *offsets, i64 ent_cap, i64 off_cap, int *overflow, void *states,
                 int *narrow_trip, int *slow_flag, const int *slow_units, i64 n_slow, i64 *scratch, const i64 *slow_scratch_off) {
  // slow_flag (COUNT): set for a unit whose gaps arrive out of the writer's merge order (Sink, translate_device.hpp).
  // slow_units (EMIT only; the FIX pass): the launch covers just those units, one lane each, with a scratch list for the open
  // segment's gaps at scratch + slow_scratch_off[lane]; it rewrites the offsets the EMIT pass wrote for them.
  const i64 n_live = live_pos[n_units];
  // (the launch covers all units; the live ones fill its first ceil(n_live / 64) workgroups, an eighth of them per XCD: xcd_chunk)
  i64 k = (EMIT && slow_units ? (i64)blockIdx.x : (i64)xcd_chunk(blockIdx.x, (unsigned)((n_live + 63) / 64))) * blockDim.x + threadIdx.x;
  if(EMIT) {
    // (EMIT: ent_off and off_off are the sums over the LIVE units, indexed by k -- expand_offsets_kernel above)
    if(ent_off[n_live] > ent_cap || off_off[n_live] > off_cap) { // uniform: buffers sized by an older run
      if(blockIdx.x == 0 && threadIdx.x == 0) {
        *overflow = 1;
      }
      return;
    }
  }
  i64 *fix = nullptr;
  i64 u = 0;
  bool active = true; // EMIT: false for a lane that only helps to store the wavefront's window
  // EMIT: the window of the wavefront's offsets (Sink::win).  8 KB: with it a CU holds 20 workgroups, five per SIMD -- more than the
  // kernel's registers allow anyway; the mean wavefront of the bench job holds 370 offsets
  constexpr int WIN = 8192 / (int)sizeof(I);
  I *win = nullptr;
  if constexpr(EMIT) {
    __shared__ I win_slots[WIN];
    win = win_slots;
  }
  i64 win_lo = 0, win_hi = 0;
  const bool fixing = EMIT && slow_units != nullptr;
  if(fixing) {
    if(k >= n_slow) {
      return;
    }
    fix = scratch + slow_scratch_off[k];
    u = slow_units[k];
    k = live_pos[u]; // the unit's place among the live ones: where the count pass left its merge start
  }
  else if(EMIT) {
    const i64 k0 = k - threadIdx.x;
    if(k0 >= n_live) { // the grid covers all units; only the live ones (compacted by the filter pass) have a lane
      return;
    }
    active = k < n_live;
    // consecutive live units: what lies between two of them holds nothing, so the wavefront's offsets are one run
    const i64 ob = active ? off_off[k] : 0, oe = active ? off_off[k + 1] : 0;
    win_lo = __shfl(ob, 0);
    win_hi = __shfl(oe, (int)(n_live - k0 < 64 ? n_live - k0 - 1 : 63));
    if(active && oe == ob && ent_off[k + 1] == ent_off[k]) {
      active = false; // the count pass found nothing to write for this unit (most left x right pairs of an entry)
    }
  }
  else {
    if(k >= n_live) {
      return;
    }
    u = live_units[k];
  }
  Sink<EMIT, I> sink;
  sink.n_ent = sink.n_off = sink.pend = sink.wpos = sink.last_start = 0;
  sink.last_row = 0;
  sink.ent = nullptr;
  sink.off = nullptr;
  sink.off_base = sink.off_cap = sink.ent_cap = 0;
  sink.disorder = 0;
  sink.fix = fix;
  sink.fix_n = 0;
  sink.fix_cap = 0;
  sink.win = win;
  sink.win_lo = win_lo;
  sink.win_cap = fixing ? 0 : WIN;
  if(EMIT && active) {
    sink.ent = entries + ent_off[k];
    sink.ent_cap = (I)(ent_off[k + 1] - ent_off[k]);
    sink.off = offsets;
    sink.off_base = off_off[k];
    sink.off_cap = (I)(off_off[k + 1] - off_off[k]);
    sink.fix_cap = sink.off_cap + 1; // the scratch list is 2 * (offsets of the unit) + 2 words (pm_job_create)
  }
  if constexpr(EMIT) {
    if(active) {
      // the count pass left this unit's merge start in states[k], with which entry and rows it is: no set-up to redo
      Merge<EMIT, I> m;
      m.sink = sink;
      UnitStateT<I> s;
      state_load<I>(states, n_live, k, s);
      unit_restore<EMIT>(left, right, ds, s.delta, s.left, s.right, s, m);
      (void)unit_merge<EMIT>(m);
    }
    if(!fixing) { // the wavefront's run of offsets, from its window: whole lines, 64 consecutive offsets a store
      __syncthreads();
      const i64 n = win_hi - win_lo < (i64)WIN ? win_hi - win_lo : (i64)WIN;
      for(i64 q = threadIdx.x; q < n; q += 64) {
        offsets[win_lo + q] = win[q];
      }
    }
  }
  else {
    const int d = u_delta[u], l = u_left[u], r = u_right[u];
    PVT<I, P> lp, rp, dr, dq;
    R2T<I> cols;
    bool live, proceed = false;
    int orientation;
    Merge<EMIT, I> m;
    m.sink = sink;
    int st = unit_prefix(left, right, ds, d, l, r, lp, rp, dr, dq, cols, live, orientation);
    if(!st && live) {
      st = unit_setup<EMIT>(lp, rp, dr, dq, cols, m, proceed);
      if(!st && proceed) {
        if(states) { // null only in the sizing pass of pm_job_create
          UnitStateT<I> s;
          unit_save<EMIT>(m, orientation, s);
          s.delta = d;
          s.left = l;
          s.right = r;
          state_store<I>(states, n_live, k, s);
        }
        st = unit_merge<EMIT>(m);
        sink = m.sink;
      }
    }
    if(sizeof(I) < 8 && st == PM_ST_NARROW) {
      atomicOr(narrow_trip, 1);
    }
    status[u] = st;
    cnt_ent[k] = sink.n_ent; // per live unit: a run per wavefront
    cnt_off[k] = sink.n_off;
    if(sink.disorder && slow_flag) {
      slow_flag[u] = 1;
      slow_flag[n_units] = 1; // "the job has such units"
    }
  }
}

// pm_job_fetch of a narrow job: its records and offsets in the C ABI's types
__global__ void widen_entries_kernel(i64 n, const Entry32 *in, pm_entry_t *out) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(k < n) {
    const Entry32 e = in[k];
    pm_entry_t w;
    w.ref_start = e.ref_start;
    w.ref_end = e.ref_end;
    w.qry_start = e.qry_start;
    w.qry_end = e.qry_end;
    w.offset_begin = e.offset_begin;
    w.n_offsets = e.n_offsets;
    out[k] = w;
  }
}
__global__ void widen_offsets_kernel(i64 n, const int *in, i64 *out) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(k < n) {
    out[k] = in[k];
  }
}

__global__ void p2s_batch_kernel(RowsD rows, i64 n, const int *row, const i64 *si, i64 *out, int *status) {
  i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(q >= n) {
    return;
  }
  int r = row[q];
  if(rows.bad[r]) {
    status[q] = PM_ST_MALFORMED_INPUT;
    out[q] = 0;
    return;
  }
  i64 v = 0;
  status[q] = profile_idx_of_seq_idx(row_view(rows, r), si[q], v);
  out[q] = v;
}

__global__ void s2p_batch_kernel(RowsD rows, i64 n, const int *row, const i64 *pi, i64 *out, int *status) {
  i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(q >= n) {
    return;
  }
  int r = row[q];
  if(rows.bad[r]) {
    status[q] = PM_ST_MALFORMED_INPUT;
    out[q] = 0;
    return;
  }
  i64 v = 0;
  bool none = false;
  int st = seq_idx_of_profile_idx(row_view(rows, r), pi[q], v, none);
  status[q] = st ? st : (none ? PM_ST_IS_NONE : PM_ST_OK);
  out[q] = v;
}

// The first launch of a kernel of this file makes the HIP runtime load the file's code object (tens of milliseconds in a fresh
// process): a caller that still has host work to do (parsing) asks for that load early, on another thread, with this.
__global__ void translate_warm_kernel() {}
int warm_translate_kernels() {
  translate_warm_kernel<<<1, 1>>>();
  PM_HIP(hipGetLastError());
  PM_HIP(hipDeviceSynchronize());
  return PM_OK;
}

// ------------------------------------------------------------------ the unit list, on the device
// The loops of _translate_delta (m_translate.cc:666-707): for every delta entry, the rows of its reference sequence on the left side
// and of its query sequence on the right side that overlap it, every (left, right) pair of them a work unit, in entry order, left
// rows outer.  Per side the host hands over the rows of every sequence sorted by forward start (_profile_map_of_dir, :188-207) as
// one array with offsets per sequence; the first candidate is std::lower_bound with "the row ends before the entry starts"
// (:175-178,682-695) -- restated here as the SAME halving search libstdc++ runs, so that the result is the reference's whatever the
// rows' order makes of the predicate -- and the candidates run on while they overlap the entry (m_range.hh:80-94).
struct EnumSideD {
  const i64 *seq_off;  // [sequences + 1] into seq_rows
  const int *seq_rows; // the side's rows, sequence by sequence, sorted by forward start
  const i64 *start, *end; // the rows' p_range (as uploaded)
};

__device__ __forceinline__ void enum_span(const EnumSideD &sd, int seq, i64 es, i64 ee, int &first, int &count) {
  first = 0;
  count = 0;
  if(seq < 0) {
    return;
  }
  const int *rows = sd.seq_rows + sd.seq_off[seq];
  const i64 n = sd.seq_off[seq + 1] - sd.seq_off[seq];
  const i64 key = es < ee ? es : ee;
  i64 lo = 0, len = n; // std::lower_bound(rows, rows + n, key, [](row, v) { return max(start, end) < v; })
  while(len > 0) {
    const i64 half = len >> 1;
    const int r = rows[lo + half];
    const i64 hi_end = sd.start[r] > sd.end[r] ? sd.start[r] : sd.end[r];
    if(hi_end < key) {
      lo = lo + half + 1;
      len = len - half - 1;
    }
    else {
      len = half;
    }
  }
  const i64 e_lo = key, e_hi = es < ee ? ee : es;
  i64 k = lo;
  for(; k < n; ++k) { // while the row overlaps the entry: max of the starts <= min of the ends
    const int r = rows[k];
    const i64 r_lo = sd.start[r] < sd.end[r] ? sd.start[r] : sd.end[r], r_hi = sd.start[r] < sd.end[r] ? sd.end[r] : sd.start[r];
    const i64 s = r_lo > e_lo ? r_lo : e_lo, e = r_hi < e_hi ? r_hi : e_hi;
    if(e - s < 0) {
      break;
    }
  }
  first = (int)lo;
  count = (int)(k - lo);
}

__global__ void enum_count_kernel(i64 n_entries, EnumSideD left, EnumSideD right, const int *ref_seq, const int *qry_seq, const i64 *ref_s,
                                  const i64 *ref_e, const i64 *qry_s, const i64 *qry_e, int4 *span, i64 *count) {
  const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(d > n_entries) {
    return;
  }
  if(d == n_entries) {
    count[d] = 0;
    return;
  }
  int l0, nl, r0, nr;
  enum_span(left, ref_seq[d], ref_s[d], ref_e[d], l0, nl);
  enum_span(right, qry_seq[d], qry_s[d], qry_e[d], r0, nr);
  if(ref_seq[d] < 0 || qry_seq[d] < 0) {
    nl = nr = 0;
  }
  span[d] = make_int4(l0, nl, r0, nr);
  count[d] = (i64)nl * nr;
}

__global__ void enum_fill_kernel(i64 n_units, i64 n_entries, EnumSideD left, EnumSideD right, const int *ref_seq, const int *qry_seq,
                                 const int4 *span, const i64 *unit_off, int *u_delta, int *u_left, int *u_right) {
  const i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(u >= n_units) {
    return;
  }
  i64 lo = 0, hi = n_entries; // the entry that owns unit u: the last d with unit_off[d] <= u
  while(hi - lo > 1) {
    const i64 mid = (lo + hi) >> 1;
    if(unit_off[mid] <= u) {
      lo = mid;
    }
    else {
      hi = mid;
    }
  }
  const int4 sp = span[lo];
  const i64 local = u - unit_off[lo];
  const i64 l = local / sp.w, r = local % sp.w; // left rows outer, right rows inner
  u_delta[u] = (int)lo;
  u_left[u] = left.seq_rows[left.seq_off[ref_seq[lo]] + sp.x + l];
  u_right[u] = right.seq_rows[right.seq_off[qry_seq[lo]] + sp.z + r];
}

// ------------------------------------------------------------------ the writer's text, on the device
// M_delta_stream_writer::write (m_delta_stream_writer.hh:55-82) for every entry of a job's result, in unit order: a `>` header
// line when the (left major name, right major name) pair differs from the pair of the last entry printed, the entry line
// `rs re qs qe 1 2 3`, one signed offset per line, the terminating 0 among them (deltas_of_gaps' output is what the emit
// pass left in `offsets`).  Three kernels around two scans:
//   text_units_kernel    per unit: "has entries" (for the header rule: which unit printed last before me) and the first
//                        failing unit (the reference dies inside it: what it had printed stays, nothing after it)
//   text_measure_kernel  per entry: its unit (upper bound in ent_off), whether it carries a header, its bytes
//   text_write_kernel    per entry: the bytes, at the exclusive scan of the measures
// One lane per entry: an entry is ~25-40 bytes of decimal text, its offset list 1-2 numbers on average.

__device__ __forceinline__ int udec_len(unsigned long long u) {
  int n = 1;
  if(u < 4294967296ull) { // 32-bit divisions for the usual case
    unsigned w = (unsigned)u;
    while(w >= 10u) {
      w /= 10u;
      ++n;
    }
    return n;
  }
  while(u >= 10ull) {
    u /= 10ull;
    ++n;
  }
  return n;
}

__device__ __forceinline__ int dec_len(i64 v) {
  return v < 0 ? 1 + udec_len(0ull - (unsigned long long)v) : udec_len((unsigned long long)v);
}

__device__ __forceinline__ char *put_dec(char *p, i64 v) {
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  if(v < 0) {
    *p++ = '-';
  }
  const int n = udec_len(u);
  if(u < 4294967296ull) {
    unsigned w = (unsigned)u;
    for(int k = n - 1; k >= 0; --k) {
      p[k] = (char)('0' + w % 10u);
      w /= 10u;
    }
  }
  else {
    for(int k = n - 1; k >= 0; --k) {
      p[k] = (char)('0' + (unsigned)(u % 10ull));
      u /= 10ull;
    }
  }
  return p + n;
}

__global__ void text_units_kernel(i64 U, const int *status, const i64 *ent_off, int *last_ne, int *first_fail) {
  const i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(u >= U) {
    return;
  }
  last_ne[u] = ent_off[u + 1] > ent_off[u] ? (int)u : -1;
  if(status[u] != PM_ST_OK) {
    atomicMin(first_fail, (int)u);
  }
}

struct TextNames {
  const char *bytes[2];   // the rows' major names back to back, left side / right side
  const i64 *off[2];      // [rows + 1]
  const int *id[2];       // equal names <-> equal ids
  const i64 *length[2];   // p_length of the rows (the header's two numbers)
};

template <bool WRITE, typename I>
__global__ void text_entries_kernel(i64 E, i64 U, const i64 *ent_off, const int *u_left, const int *u_right, const int *last_ne,
                                    const int *first_fail, const typename EntRecT<I>::type *entries, const I *offsets, TextNames names,
                                    int *e_unit, i64 *len, const i64 *pos, char *text) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= E) {
    return;
  }
  int u;
  if(!WRITE) {
    // the unit that owns entry k: the last u with ent_off[u] <= k (units without entries repeat their neighbour's offset)
    i64 lo = 0, hi = U; // invariant: ent_off[lo] <= k < ent_off[hi]
    while(hi - lo > 1) {
      const i64 mid = (lo + hi) >> 1;
      if(ent_off[mid] <= k) {
        lo = mid;
      }
      else {
        hi = mid;
      }
    }
    u = (int)lo;
  }
  else {
    u = e_unit[k];
    if(u < 0) {
      return; // behind the first failing unit: the reference never got there
    }
  }
  const int l = u_left[u], r = u_right[u];
  bool header = false;
  if(k == ent_off[u]) { // the unit's first entry: does the writer hold another name pair?
    const int p = u > 0 ? last_ne[u - 1] : -1; // the last unit before u that printed (inclusive max-scan of text_units_kernel's marks)
    // (the writer starts with the pair ("", ""), m_delta_stream_writer.hh:55-60)
    header = p < 0 ? (names.off[0][l + 1] > names.off[0][l] || names.off[1][r + 1] > names.off[1][r])
                   : (names.id[0][u_left[p]] != names.id[0][l] || names.id[1][u_right[p]] != names.id[1][r]);
  }
  const typename EntRecT<I>::type en = entries[k];
  if(!WRITE) {
    if(u > *first_fail) {
      e_unit[k] = -1;
      len[k] = 0;
      return;
    }
    i64 n = dec_len(en.ref_start) + dec_len(en.ref_end) + dec_len(en.qry_start) + dec_len(en.qry_end) + 3 + 7;
    if(header) {
      n += 1 + (names.off[0][l + 1] - names.off[0][l]) + 1 + (names.off[1][r + 1] - names.off[1][r]) + 1 + dec_len(names.length[0][l]) + 1 +
           dec_len(names.length[1][r]) + 1;
    }
    for(i64 o = 0; o < en.n_offsets; ++o) {
      n += dec_len(offsets[en.offset_begin + o]) + 1;
    }
    e_unit[k] = u;
    len[k] = n;
    return;
  }
  char *p = text + pos[k];
  if(header) {
    *p++ = '>';
    for(i64 c = names.off[0][l]; c < names.off[0][l + 1]; ++c) {
      *p++ = names.bytes[0][c];
    }
    *p++ = ' ';
    for(i64 c = names.off[1][r]; c < names.off[1][r + 1]; ++c) {
      *p++ = names.bytes[1][c];
    }
    *p++ = ' ';
    p = put_dec(p, names.length[0][l]);
    *p++ = ' ';
    p = put_dec(p, names.length[1][r]);
    *p++ = '\n';
  }
  p = put_dec(p, en.ref_start);
  *p++ = ' ';
  p = put_dec(p, en.ref_end);
  *p++ = ' ';
  p = put_dec(p, en.qry_start);
  *p++ = ' ';
  p = put_dec(p, en.qry_end);
  *p++ = ' ';
  *p++ = '1';
  *p++ = ' ';
  *p++ = '2';
  *p++ = ' ';
  *p++ = '3'; // the three error fields, m_delta_stream_writer.hh:71
  *p++ = '\n';
  for(i64 o = 0; o < en.n_offsets; ++o) {
    p = put_dec(p, offsets[en.offset_begin + o]);
    *p++ = '\n';
  }
}

// ------------------------------------------------------------------ device-resident tables

static int check_csr(const int64_t *off, int64_t n, const char *what) {
  if(n < 0) {
    return fail(PM_E_INVALID, std::string(what) + ": negative count");
  }
  if(!off) {
    return fail(PM_E_INVALID, std::string(what) + ": null offsets");
  }
  if(off[0] != 0) {
    return fail(PM_E_INVALID, std::string(what) + ": offsets must start at 0");
  }
  for(int64_t k = 0; k < n; ++k) {
    if(off[k + 1] < off[k]) {
      return fail(PM_E_INVALID, std::string(what) + ": offsets not ascending");
    }
    if(off[k + 1] - off[k] > 0x7fffffff) {
      return fail(PM_E_INVALID, std::string(what) + ": more than 2^31 gaps in one list");
    }
  }
  return PM_OK;
}

int upload_rows(const pm_rows_t *h, RowsStore &s, hipStream_t stream, unsigned long long *maxabs) {
  if(!h || h->n < 0 || (h->n > 0 && (!h->start || !h->end || !h->length))) {
    return fail(PM_E_INVALID, "rows: null array");
  }
  int rc = check_csr(h->gap_off, h->n, "rows.gap_off");
  if(rc) {
    return rc;
  }
  s.n = h->n;
  s.G = h->gap_off[h->n];
  if(s.G > 0 && (!h->gap_start || !h->gap_end)) {
    return fail(PM_E_INVALID, "rows: null gap arrays");
  }
  size_t n8 = (size_t)s.n * 8, g8 = (size_t)s.G * 8;
  PM_TRY(s.range.alloc(n8 * 2));
  PM_TRY(s.length.upload(h->length, n8, stream));
  PM_TRY(s.gap_off.upload(h->gap_off, n8 + 8, stream));
  PM_TRY(s.gaps.alloc(g8 * 2));
  PM_TRY(s.pre.alloc(g8 + n8));
  PM_TRY(s.bad.alloc((size_t)s.n * 4));
  PM_TRY(s.raw_s.upload(h->start, n8, stream));
  PM_TRY(s.raw_e.upload(h->end, n8, stream));
  PM_TRY(s.raw_gs.upload(h->gap_start, g8, stream));
  PM_TRY(s.raw_ge.upload(h->gap_end, g8, stream));
  if(maxabs) {
    PM_TRY(s.range32.alloc(n8));
    PM_TRY(s.length32.alloc(n8 / 2));
    PM_TRY(s.gaps32.alloc(g8));
    PM_TRY(s.pre32.alloc((g8 + n8) / 2));
  }
  if(s.n > 0) {
    unsigned blocks = (unsigned)((s.n + 255) / 256);
    prepare_rows_kernel<i64><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw_s.p, (const i64 *)s.raw_e.p, (const i64 *)s.length.p,
                                                         (const i64 *)s.gap_off.p, (const i64 *)s.raw_gs.p, (const i64 *)s.raw_ge.p,
                                                         (R2 *)s.range.p, nullptr, (R2 *)s.gaps.p, (i64 *)s.pre.p, (int *)s.bad.p, maxabs);
    if(maxabs) {
      prepare_rows_kernel<int><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw_s.p, (const i64 *)s.raw_e.p, (const i64 *)s.length.p,
                                                           (const i64 *)s.gap_off.p, (const i64 *)s.raw_gs.p, (const i64 *)s.raw_ge.p,
                                                           (R2T<int> *)s.range32.p, (int *)s.length32.p, (R2T<int> *)s.gaps32.p,
                                                           (int *)s.pre32.p, nullptr, nullptr);
    }
    PM_HIP(hipGetLastError());
  }
  return PM_OK;
}

struct DeltasStore {
  DevBuf ref, qry, ref_off, qry_off, bad;
  DevBuf ref_gaps[2], ref_pre[2], qry_gaps[2], qry_pre[2];
  DevBuf ref32, qry32, ref_gaps32[2], ref_pre32[2], qry_gaps32[2], qry_pre32[2]; // the same tables as int
  DevBuf raw[8];
  i64 n = 0, Gr = 0, Gq = 0;
  DeltasD view() const {
    DeltasD d;
    d.n = n;
    d.ref = (const R2 *)ref.p;
    d.qry = (const R2 *)qry.p;
    d.ref_off = (const i64 *)ref_off.p;
    d.qry_off = (const i64 *)qry_off.p;
    for(int o = 0; o < 2; ++o) {
      d.ref_gaps[o] = (const R2 *)ref_gaps[o].p;
      d.ref_pre[o] = (const i64 *)ref_pre[o].p;
      d.qry_gaps[o] = (const R2 *)qry_gaps[o].p;
      d.qry_pre[o] = (const i64 *)qry_pre[o].p;
    }
    d.bad = (const int *)bad.p;
    return d;
  }
  DeltasT<int, i64> view_mixed() const { // positions in 64 bits, the gap tables as int (translate_device.hpp)
    DeltasT<int, i64> d;
    d.n = n;
    d.ref = (const R2 *)ref.p;
    d.qry = (const R2 *)qry.p;
    d.ref_off = (const i64 *)ref_off.p;
    d.qry_off = (const i64 *)qry_off.p;
    for(int o = 0; o < 2; ++o) {
      d.ref_gaps[o] = (const R2T<int> *)ref_gaps32[o].p;
      d.ref_pre[o] = (const int *)ref_pre32[o].p;
      d.qry_gaps[o] = (const R2T<int> *)qry_gaps32[o].p;
      d.qry_pre[o] = (const int *)qry_pre32[o].p;
    }
    d.bad = (const int *)bad.p;
    return d;
  }
  DeltasT<int> view32() const {
    DeltasT<int> d;
    d.n = n;
    d.ref = (const R2T<int> *)ref32.p;
    d.qry = (const R2T<int> *)qry32.p;
    d.ref_off = (const i64 *)ref_off.p;
    d.qry_off = (const i64 *)qry_off.p;
    for(int o = 0; o < 2; ++o) {
      d.ref_gaps[o] = (const R2T<int> *)ref_gaps32[o].p;
      d.ref_pre[o] = (const int *)ref_pre32[o].p;
      d.qry_gaps[o] = (const R2T<int> *)qry_gaps32[o].p;
      d.qry_pre[o] = (const int *)qry_pre32[o].p;
    }
    d.bad = (const int *)bad.p;
    return d;
  }
};

static int upload_deltas(const pm_deltas_t *h, DeltasStore &s, hipStream_t stream, unsigned long long *maxabs) {
  if(!h || h->n < 0 || (h->n > 0 && (!h->ref_start || !h->ref_end || !h->qry_start || !h->qry_end))) {
    return fail(PM_E_INVALID, "deltas: null array");
  }
  int rc = check_csr(h->ref_gap_off, h->n, "deltas.ref_gap_off");
  if(rc) {
    return rc;
  }
  rc = check_csr(h->qry_gap_off, h->n, "deltas.qry_gap_off");
  if(rc) {
    return rc;
  }
  s.n = h->n;
  s.Gr = h->ref_gap_off[h->n];
  s.Gq = h->qry_gap_off[h->n];
  if((s.Gr > 0 && (!h->ref_gap_start || !h->ref_gap_end)) || (s.Gq > 0 && (!h->qry_gap_start || !h->qry_gap_end))) {
    return fail(PM_E_INVALID, "deltas: null gap arrays");
  }
  size_t n8 = (size_t)s.n * 8;
  PM_TRY(s.ref.alloc(n8 * 2));
  PM_TRY(s.qry.alloc(n8 * 2));
  PM_TRY(s.ref_off.upload(h->ref_gap_off, n8 + 8, stream));
  PM_TRY(s.qry_off.upload(h->qry_gap_off, n8 + 8, stream));
  PM_TRY(s.bad.alloc((size_t)s.n * 4));
  PM_HIP(hipMemsetAsync(s.bad.p, 0, (size_t)s.n * 4, stream));
  for(int o = 0; o < 2; ++o) {
    PM_TRY(s.ref_gaps[o].alloc((size_t)s.Gr * 16));
    PM_TRY(s.ref_pre[o].alloc((size_t)s.Gr * 8 + n8));
    PM_TRY(s.qry_gaps[o].alloc((size_t)s.Gq * 16));
    PM_TRY(s.qry_pre[o].alloc((size_t)s.Gq * 8 + n8));
    PM_TRY(s.ref_gaps32[o].alloc((size_t)s.Gr * 8));
    PM_TRY(s.ref_pre32[o].alloc((size_t)s.Gr * 4 + n8 / 2));
    PM_TRY(s.qry_gaps32[o].alloc((size_t)s.Gq * 8));
    PM_TRY(s.qry_pre32[o].alloc((size_t)s.Gq * 4 + n8 / 2));
  }
  PM_TRY(s.ref32.alloc(n8));
  PM_TRY(s.qry32.alloc(n8));
  PM_TRY(s.raw[0].upload(h->ref_start, n8, stream));
  PM_TRY(s.raw[1].upload(h->ref_end, n8, stream));
  PM_TRY(s.raw[2].upload(h->qry_start, n8, stream));
  PM_TRY(s.raw[3].upload(h->qry_end, n8, stream));
  PM_TRY(s.raw[4].upload(h->ref_gap_start, (size_t)s.Gr * 8, stream));
  PM_TRY(s.raw[5].upload(h->ref_gap_end, (size_t)s.Gr * 8, stream));
  PM_TRY(s.raw[6].upload(h->qry_gap_start, (size_t)s.Gq * 8, stream));
  PM_TRY(s.raw[7].upload(h->qry_gap_end, (size_t)s.Gq * 8, stream));
  if(s.n > 0) {
    unsigned blocks = (unsigned)((s.n + 255) / 256);
    prepare_deltas_kernel<i64><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw[0].p, (const i64 *)s.raw[1].p, (const i64 *)s.ref_off.p,
                                                           (const i64 *)s.raw[4].p, (const i64 *)s.raw[5].p, (R2 *)s.ref.p,
                                                           (R2 *)s.ref_gaps[0].p, (i64 *)s.ref_pre[0].p, (R2 *)s.ref_gaps[1].p,
                                                           (i64 *)s.ref_pre[1].p, (int *)s.bad.p, maxabs);
    prepare_deltas_kernel<i64><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw[2].p, (const i64 *)s.raw[3].p, (const i64 *)s.qry_off.p,
                                                           (const i64 *)s.raw[6].p, (const i64 *)s.raw[7].p, (R2 *)s.qry.p,
                                                           (R2 *)s.qry_gaps[0].p, (i64 *)s.qry_pre[0].p, (R2 *)s.qry_gaps[1].p,
                                                           (i64 *)s.qry_pre[1].p, (int *)s.bad.p, maxabs);
    prepare_deltas_kernel<int><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw[0].p, (const i64 *)s.raw[1].p, (const i64 *)s.ref_off.p,
                                                           (const i64 *)s.raw[4].p, (const i64 *)s.raw[5].p, (R2T<int> *)s.ref32.p,
                                                           (R2T<int> *)s.ref_gaps32[0].p, (int *)s.ref_pre32[0].p,
                                                           (R2T<int> *)s.ref_gaps32[1].p, (int *)s.ref_pre32[1].p, nullptr, nullptr);
    prepare_deltas_kernel<int><<<blocks, 256, 0, stream>>>(s.n, (const i64 *)s.raw[2].p, (const i64 *)s.raw[3].p, (const i64 *)s.qry_off.p,
                                                           (const i64 *)s.raw[6].p, (const i64 *)s.raw[7].p, (R2T<int> *)s.qry32.p,
                                                           (R2T<int> *)s.qry_gaps32[0].p, (int *)s.qry_pre32[0].p,
                                                           (R2T<int> *)s.qry_gaps32[1].p, (int *)s.qry_pre32[1].p, nullptr, nullptr);
    PM_HIP(hipGetLastError());
  }
  return PM_OK;
}

} // namespace pm

using namespace pm;

struct pm_job {
  int device = 0;
  RowsStore left, right;
  DeltasStore deltas;
  DevBuf u_delta, u_left, u_right;
  i64 n_units = 0;
  // cnt_ent, cnt_off, ent_off_l, off_off: per LIVE unit (index k); ent_off: per unit (expand_offsets_kernel); totals: {entries, offsets}
  DevBuf status, cnt_ent, cnt_off, ent_off, ent_off_l, off_off, totals, entries, offsets, overflow, scan_tmp;
  i64 n_live = -1; // live units of the job (its tables never change, so neither does this); -1 until the sizing pass has run
  DevBuf live_flag, live_pos, live_units, scan_tmp32, scan_partial;
  DevBuf states; // UnitState per live unit (null during the sizing pass of pm_job_create)
  DevBuf maxabs, narrow_trip;
  // units whose gaps arrive out of the writer's order (Sink): found by the sizing pass, emitted again by the FIX pass
  DevBuf slow_flag, slow_units, slow_scratch_off, slow_scratch;
  i64 n_slow = 0;
  // what the EMIT pass writes per entry / per offset (translate_device.hpp: Entry32 + int for a narrow job, pm_entry_t + int64 else)
  int64_t rec_bytes() const { return narrow ? (int64_t)sizeof(Entry32) : (int64_t)sizeof(pm_entry_t); }
  int64_t off_bytes() const { return narrow ? 4 : 8; }
  bool library_scans = false; // PM_TRANSLATE_LIBRARY_SCANS=1 at pm_job_create: the prefix sums of jobs above SCAN_MAX_TILES tiles, for any job (tests)
  bool narrow = false; // the job runs on the int tables (every table value below PM_NARROW_INPUT_LIMIT, no PM_ST_NARROW seen)
  // narrow, with the sequence positions -- and only them -- in 64 bits: a job whose positions pass PM_NARROW_INPUT_LIMIT while every
  // length, span and gap column is below it (translate_device.hpp, type P).  Only the filter and count passes ever see a position;
  // everything behind them (emit, fix, text, fetch) is the int job's
  bool wide_positions = false;
  size_t scan_tmp32_bytes = 0;
  size_t scan_tmp_bytes = 0;
  i64 ent_cap = 0, off_cap = 0;
  i64 n_entries = 0, n_offsets = 0;
  i64 input_bytes = 0;
  hipStream_t last_stream = nullptr;
  bool ran = false;
  // the delta text of the last run, formatted on the device (pm_job_text)
  DevBuf t_names[2], t_name_off[2], t_name_id[2], t_marks, t_last_ne, t_eunit, t_len, t_pos, t_text, t_scan_tmp, t_word;
  i64 text_bytes = -1;
};

// The four phases of one pass; `ev` (5 events) brackets them when the pass is being timed.
static int job_launch_pass(pm_job *j, hipStream_t stream, bool emit, hipEvent_t *ev) {
  i64 U = j->n_units;
  unsigned blocks = (unsigned)((U + 63) / 64);
  unsigned blocks256 = (unsigned)((U + 255) / 256);
  if(ev) {
    PM_HIP(hipEventRecord(ev[0], stream));
  }
  // 1. filter + compaction of the live units
  if(U > 0) {
    if(j->narrow && j->wide_positions) {
      translate_filter_kernel<int, i64><<<blocks, 64, 0, stream>>>(j->left.view_mixed(), j->right.view_mixed(), j->deltas.view_mixed(), U,
                                                                   (const int *)j->u_delta.p, (const int *)j->u_left.p, (const int *)j->u_right.p,
                                                                   (int *)j->status.p, (i64 *)j->cnt_ent.p, (i64 *)j->cnt_off.p,
                                                                   (int *)j->live_flag.p);
    }
    else if(j->narrow) {
      translate_filter_kernel<int, int><<<blocks, 64, 0, stream>>>(j->left.view32(), j->right.view32(), j->deltas.view32(), U,
                                                              (const int *)j->u_delta.p, (const int *)j->u_left.p, (const int *)j->u_right.p,
                                                              (int *)j->status.p, (i64 *)j->cnt_ent.p, (i64 *)j->cnt_off.p,
                                                              (int *)j->live_flag.p);
    }
    else {
      translate_filter_kernel<i64, i64><<<blocks, 64, 0, stream>>>(j->left.view(), j->right.view(), j->deltas.view(), U,
                                                              (const int *)j->u_delta.p, (const int *)j->u_left.p, (const int *)j->u_right.p,
                                                              (int *)j->status.p, (i64 *)j->cnt_ent.p, (i64 *)j->cnt_off.p,
                                                              (int *)j->live_flag.p);
    }
    PM_HIP(hipGetLastError());
  }
  const i64 n_scan = U + 1;
  const unsigned tiles = (unsigned)((n_scan + SCAN_TILE - 1) / SCAN_TILE);
  const bool own_scans = tiles <= (unsigned)SCAN_MAX_TILES && !j->library_scans;
  if(own_scans) {
    flag_tile_sums_kernel<<<tiles, SCAN_THREADS, 0, stream>>>(n_scan, (const int *)j->live_flag.p, (int *)j->scan_partial.p);
    flag_scan_scatter_kernel<<<tiles, SCAN_THREADS, 0, stream>>>(n_scan, U, (const int *)j->live_flag.p, (const int *)j->scan_partial.p,
                                                               (int *)j->live_pos.p, (int *)j->live_units.p);
    PM_HIP(hipGetLastError());
  }
  else {
    size_t tmp32 = j->scan_tmp32_bytes;
    PM_HIP(rocprim::exclusive_scan(j->scan_tmp32.p, tmp32, (int *)j->live_flag.p, (int *)j->live_pos.p, 0, (size_t)(U + 1),
                                   rocprim::plus<int>(), stream));
    if(U > 0) {
      scatter_live_kernel<<<blocks256, 256, 0, stream>>>(U, (const int *)j->live_flag.p, (const int *)j->live_pos.p, (int *)j->live_units.p);
      PM_HIP(hipGetLastError());
    }
  }
  if(ev) {
    PM_HIP(hipEventRecord(ev[1], stream));
  }
  // 2. count pass over the live units
  if(U > 0) {
#define PM_COUNT_ARGS                                                                                                              \
  U, (const int *)j->u_delta.p, (const int *)j->u_left.p, (const int *)j->u_right.p, (const int *)j->live_units.p,                 \
      (const int *)j->live_pos.p, (int *)j->status.p, (i64 *)j->cnt_ent.p, (i64 *)j->cnt_off.p, nullptr, nullptr, nullptr, nullptr, \
      0, 0, nullptr, j->states.p, (int *)j->narrow_trip.p, (int *)j->slow_flag.p, nullptr, 0, nullptr, nullptr
    if(j->narrow && j->wide_positions) {
      translate_kernel<false, int, i64><<<blocks, 64, 0, stream>>>(j->left.view_mixed(), j->right.view_mixed(), j->deltas.view_mixed(), PM_COUNT_ARGS);
    }
    else if(j->narrow) {
      translate_kernel<false, int, int><<<blocks, 64, 0, stream>>>(j->left.view32(), j->right.view32(), j->deltas.view32(), PM_COUNT_ARGS);
    }
    else {
      translate_kernel<false, i64, i64><<<blocks, 64, 0, stream>>>(j->left.view(), j->right.view(), j->deltas.view(), PM_COUNT_ARGS);
    }
#undef PM_COUNT_ARGS
    PM_HIP(hipGetLastError());
  }
  if(ev) {
    PM_HIP(hipEventRecord(ev[2], stream));
  }
  // 3. output offsets: sums over the live units' counts (n_live + 1 elements; in the sizing pass of pm_job_create, which does not know
  // n_live yet, over U + 1 -- the counts behind the live units' are zero), then the per-unit index and the totals
  {
    const i64 n_cscan = (j->n_live >= 0 ? j->n_live : U) + 1;
    const unsigned ctiles = (unsigned)((n_cscan + SCAN_TILE - 1) / SCAN_TILE);
    if(ctiles <= (unsigned)SCAN_MAX_TILES && !j->library_scans) {
      count_tile_sums_kernel<<<ctiles, SCAN_THREADS, 0, stream>>>(n_cscan, (const i64 *)j->cnt_ent.p, (const i64 *)j->cnt_off.p, (Sum2 *)j->scan_partial.p);
      count_scan_kernel<<<ctiles, SCAN_THREADS, 0, stream>>>(n_cscan, (const i64 *)j->cnt_ent.p, (const i64 *)j->cnt_off.p,
                                                           (const Sum2 *)j->scan_partial.p, (i64 *)j->ent_off_l.p, (i64 *)j->off_off.p);
      PM_HIP(hipGetLastError());
    }
    else {
      size_t tmp = j->scan_tmp_bytes;
      PM_HIP(rocprim::exclusive_scan(j->scan_tmp.p, tmp, (i64 *)j->cnt_ent.p, (i64 *)j->ent_off_l.p, (i64)0, (size_t)n_cscan,
                                     rocprim::plus<i64>(), stream));
      tmp = j->scan_tmp_bytes;
      PM_HIP(rocprim::exclusive_scan(j->scan_tmp.p, tmp, (i64 *)j->cnt_off.p, (i64 *)j->off_off.p, (i64)0, (size_t)n_cscan,
                                     rocprim::plus<i64>(), stream));
    }
    expand_offsets_kernel<<<(unsigned)((U + 256) / 256), 256, 0, stream>>>(U, (const int *)j->live_pos.p, (const i64 *)j->ent_off_l.p,
                                                                          (const i64 *)j->off_off.p, (i64 *)j->ent_off.p, (i64 *)j->totals.p);
    PM_HIP(hipGetLastError());
  }
  if(ev) {
    PM_HIP(hipEventRecord(ev[3], stream));
  }
  // 4. emit pass over the live units
  // (a job with wide positions: the int job's kernel on the int tables -- the emit pass restores the merge from the saved state and reads
  // gap lists; no position enters it, and the int tables' ranges, which would be truncated, are not read)
  if(emit && U > 0) {
#define PM_EMIT_ARGS_OF(I)                                                                                                        \
  U, (const int *)j->u_delta.p, (const int *)j->u_left.p, (const int *)j->u_right.p, (const int *)j->live_units.p,                \
      (const int *)j->live_pos.p, nullptr, nullptr, nullptr, (const i64 *)j->ent_off_l.p, (const i64 *)j->off_off.p,               \
      (EntRecT<I>::type *)j->entries.p, (I *)j->offsets.p, j->ent_cap, j->off_cap, (int *)j->overflow.p, j->states.p,              \
      (int *)j->narrow_trip.p, nullptr
    if(j->narrow) {
      translate_kernel<true, int, int><<<blocks, 64, 0, stream>>>(j->left.view32(), j->right.view32(), j->deltas.view32(), PM_EMIT_ARGS_OF(int),
                                                             nullptr, 0, nullptr, nullptr);
    }
    else {
      translate_kernel<true, i64, i64><<<blocks, 64, 0, stream>>>(j->left.view(), j->right.view(), j->deltas.view(), PM_EMIT_ARGS_OF(i64), nullptr, 0,
                                                             nullptr, nullptr);
    }
    PM_HIP(hipGetLastError());
    if(j->n_slow > 0) { // the FIX pass: the few units whose offsets are not what on-the-fly emission gives
      const unsigned fblocks = (unsigned)((j->n_slow + 63) / 64);
      if(j->narrow) {
        translate_kernel<true, int, int><<<fblocks, 64, 0, stream>>>(j->left.view32(), j->right.view32(), j->deltas.view32(), PM_EMIT_ARGS_OF(int),
                                                                (const int *)j->slow_units.p, j->n_slow, (i64 *)j->slow_scratch.p,
                                                                (const i64 *)j->slow_scratch_off.p);
      }
      else {
        translate_kernel<true, i64, i64><<<fblocks, 64, 0, stream>>>(j->left.view(), j->right.view(), j->deltas.view(), PM_EMIT_ARGS_OF(i64),
                                                                (const int *)j->slow_units.p, j->n_slow, (i64 *)j->slow_scratch.p,
                                                                (const i64 *)j->slow_scratch_off.p);
      }
      PM_HIP(hipGetLastError());
    }
#undef PM_EMIT_ARGS_OF
  }
  if(ev) {
    PM_HIP(hipEventRecord(ev[4], stream));
  }
  return PM_OK;
}

static int job_read_totals(pm_job *j, hipStream_t stream) {
  i64 tot[2] = {0, 0};
  PM_HIP(hipMemcpyAsync(tot, j->totals.p, 16, hipMemcpyDeviceToHost, stream));
  PM_HIP(hipStreamSynchronize(stream));
  j->n_entries = tot[0];
  j->n_offsets = tot[1];
  return PM_OK;
}

namespace pm {
const char *job_text_device(pm_job_t *j) { return (const char *)j->t_text.p; }
} // namespace pm

extern "C" {

} // extern "C"

// pm_job_create, with the unit list either given (units) or made on the device from the sides' per-sequence row lists (en).
static int job_create_impl(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                           const pm::EnumInput *en, const pm_translate_options_t &opt, int device, pm_job_t **out) {
  if(!out) {
    return fail(PM_E_INVALID, "pm_job_create: null out");
  }
  if(opt.coordinate_bits != 0 && opt.coordinate_bits != 32 && opt.coordinate_bits != 64) {
    return fail(PM_E_INVALID, "pm_job_create: options.coordinate_bits is 0, 32 or 64");
  }
  *out = nullptr;
  const pm_units_t no_units = {0, nullptr, nullptr, nullptr};
  if(en) {
    units = &no_units;
  }
  if(!units || units->n < 0 || (units->n > 0 && (!units->delta || !units->left || !units->right))) {
    return fail(PM_E_INVALID, "pm_job_create: bad units");
  }
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  if(!left || !right || !deltas) {
    return fail(PM_E_INVALID, "pm_job_create: null table");
  }
  for(int64_t u = 0; u < units->n; ++u) {
    if(units->delta[u] < 0 || units->delta[u] >= deltas->n || units->left[u] < 0 || units->left[u] >= left->n || units->right[u] < 0 ||
       units->right[u] >= right->n) {
      return fail(PM_E_INVALID, "pm_job_create: unit index out of range");
    }
  }
  pm_job *j = new(std::nothrow) pm_job();
  if(!j) {
    return fail(PM_E_INVALID, "out of host memory");
  }
  j->device = device;
  hipStream_t stream = nullptr;
#define JTRY(x)        \
  do {                 \
    int rc_ = (x);     \
    if(rc_) {          \
      pm_job_destroy(j); \
      return rc_;      \
    }                  \
  } while(0)
  const bool timing = opt.timing != 0;
  auto wall = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  };
  double lap_t = wall();
  auto lap = [&](const char *what) {
    if(timing) {
      (void)hipStreamSynchronize(stream);
      const double t = wall();
      fprintf(stderr, "[pm]   job create: %-34s %.4f s\n", what, t - lap_t);
      lap_t = t;
    }
  };
  JTRY(j->maxabs.alloc(16));
  JTRY(j->narrow_trip.alloc(4));
  if(hipMemsetAsync(j->maxabs.p, 0, 16, stream) != hipSuccess || hipMemsetAsync(j->narrow_trip.p, 0, 4, stream) != hipSuccess) {
    pm_job_destroy(j);
    return fail(PM_E_HIP, "hipMemsetAsync failed");
  }
  JTRY(upload_rows(left, j->left, stream, (unsigned long long *)j->maxabs.p));
  JTRY(upload_rows(right, j->right, stream, (unsigned long long *)j->maxabs.p));
  JTRY(upload_deltas(deltas, j->deltas, stream, (unsigned long long *)j->maxabs.p));
  lap("rows + deltas up, prepared");
  i64 U = j->n_units = units->n;
  if(en) {
    // the unit list on the device: spans and counts per entry, a scan, then one thread per unit
    const i64 D = deltas->n;
    DevBuf d_seq_off[2], d_seq_rows[2], d_eseq[2], d_span, d_count, d_uoff, d_tmp;
    for(int sd = 0; sd < 2; ++sd) {
      JTRY(d_seq_off[sd].upload(en->seq_off[sd], (size_t)(en->n_seq[sd] + 1) * 8, stream));
      JTRY(d_seq_rows[sd].upload(en->seq_rows[sd], (size_t)en->seq_off[sd][en->n_seq[sd]] * 4, stream));
      JTRY(d_eseq[sd].upload(en->entry_seq[sd], (size_t)D * 4, stream));
    }
    JTRY(d_span.alloc((size_t)(D + 1) * 16));
    JTRY(d_count.alloc((size_t)(D + 1) * 8));
    JTRY(d_uoff.alloc((size_t)(D + 1) * 8));
    EnumSideD L = {(const i64 *)d_seq_off[0].p, (const int *)d_seq_rows[0].p, (const i64 *)j->left.raw_s.p, (const i64 *)j->left.raw_e.p};
    EnumSideD R = {(const i64 *)d_seq_off[1].p, (const int *)d_seq_rows[1].p, (const i64 *)j->right.raw_s.p, (const i64 *)j->right.raw_e.p};
    enum_count_kernel<<<(unsigned)((D + 256) / 256), 256, 0, stream>>>(D, L, R, (const int *)d_eseq[0].p, (const int *)d_eseq[1].p,
                                                                       (const i64 *)j->deltas.raw[0].p, (const i64 *)j->deltas.raw[1].p,
                                                                       (const i64 *)j->deltas.raw[2].p, (const i64 *)j->deltas.raw[3].p,
                                                                       (int4 *)d_span.p, (i64 *)d_count.p);
    size_t tmp_b = 0;
    if(hipGetLastError() != hipSuccess ||
       rocprim::exclusive_scan(nullptr, tmp_b, (i64 *)d_count.p, (i64 *)d_uoff.p, (i64)0, (size_t)(D + 1), rocprim::plus<i64>(), stream) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "unit enumeration failed");
    }
    JTRY(d_tmp.alloc(tmp_b ? tmp_b : 8));
    i64 total = 0;
    if(rocprim::exclusive_scan(d_tmp.p, tmp_b, (i64 *)d_count.p, (i64 *)d_uoff.p, (i64)0, (size_t)(D + 1), rocprim::plus<i64>(), stream) != hipSuccess ||
       hipMemcpyAsync(&total, (i64 *)d_uoff.p + D, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "unit enumeration failed");
    }
    if(total < 0 || total >= ((i64)1 << 31)) {
      pm_job_destroy(j);
      return fail(PM_E_INVALID, "pm_job_create: more than 2^31 work units in one job");
    }
    U = j->n_units = total;
    JTRY(j->u_delta.alloc((size_t)U * 4));
    JTRY(j->u_left.alloc((size_t)U * 4));
    JTRY(j->u_right.alloc((size_t)U * 4));
    if(U > 0) {
      enum_fill_kernel<<<(unsigned)((U + 255) / 256), 256, 0, stream>>>(U, D, L, R, (const int *)d_eseq[0].p, (const int *)d_eseq[1].p,
                                                                        (const int4 *)d_span.p, (const i64 *)d_uoff.p, (int *)j->u_delta.p,
                                                                        (int *)j->u_left.p, (int *)j->u_right.p);
      if(hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { // the scratch tables die with this scope
        pm_job_destroy(j);
        return fail(PM_E_HIP, "unit enumeration failed");
      }
    }
    lap("units listed on the device");
  }
  else {
    JTRY(j->u_delta.upload(units->delta, (size_t)U * 4, stream));
    JTRY(j->u_left.upload(units->left, (size_t)U * 4, stream));
    JTRY(j->u_right.upload(units->right, (size_t)U * 4, stream));
  }
  JTRY(j->status.alloc((size_t)U * 4));
  JTRY(j->cnt_ent.alloc((size_t)(U + 1) * 8));
  JTRY(j->cnt_off.alloc((size_t)(U + 1) * 8));
  JTRY(j->ent_off.alloc((size_t)(U + 1) * 8));
  JTRY(j->ent_off_l.alloc((size_t)(U + 1) * 8));
  JTRY(j->off_off.alloc((size_t)(U + 1) * 8));
  JTRY(j->totals.alloc(16));
  JTRY(j->overflow.alloc(4));
  JTRY(j->live_flag.alloc((size_t)(U + 1) * 4));
  JTRY(j->live_pos.alloc((size_t)(U + 1) * 4));
  JTRY(j->live_units.alloc((size_t)(U + 1) * 4));
  JTRY(j->slow_flag.alloc((size_t)(U + 1) * 4));
  if(hipMemsetAsync(j->slow_flag.p, 0, (size_t)(U + 1) * 4, stream) != hipSuccess) {
    pm_job_destroy(j);
    return fail(PM_E_HIP, "hipMemsetAsync failed");
  }
  if(hipMemsetAsync(j->live_flag.p, 0, (size_t)(U + 1) * 4, stream) != hipSuccess) {
    pm_job_destroy(j);
    return fail(PM_E_HIP, "hipMemsetAsync failed");
  }
  if(hipMemsetAsync(j->cnt_ent.p, 0, (size_t)(U + 1) * 8, stream) != hipSuccess ||
     hipMemsetAsync(j->cnt_off.p, 0, (size_t)(U + 1) * 8, stream) != hipSuccess ||
     hipMemsetAsync(j->overflow.p, 0, 4, stream) != hipSuccess) {
    pm_job_destroy(j);
    return fail(PM_E_HIP, "hipMemsetAsync failed");
  }
  size_t tmp = 0;
  if(rocprim::exclusive_scan(nullptr, tmp, (i64 *)j->cnt_ent.p, (i64 *)j->ent_off.p, (i64)0, (size_t)(U + 1), rocprim::plus<i64>(),
                             stream) != hipSuccess) {
    pm_job_destroy(j);
    return fail(PM_E_HIP, "rocprim scan sizing failed");
  }
  j->scan_tmp_bytes = tmp;
  {
    size_t tmp32 = 0;
    if(rocprim::exclusive_scan(nullptr, tmp32, (int *)j->live_flag.p, (int *)j->live_pos.p, 0, (size_t)(U + 1), rocprim::plus<int>(), stream) !=
       hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "rocprim scan sizing failed");
    }
    j->scan_tmp32_bytes = tmp32;
    JTRY(j->scan_tmp32.alloc(tmp32 ? tmp32 : 8));
  }
  JTRY(j->scan_tmp.alloc(tmp ? tmp : 8));
  JTRY(j->scan_partial.alloc((size_t)SCAN_MAX_TILES * sizeof(Sum2)));
  lap("units up, per-unit arrays");
  // int or int64 tables?  int when every magnitude in the tables is below the limit (PM_TRANSLATE_WIDE=1 forces int64)
  {
    unsigned long long big[2] = {0, 0}; // [0] positions, [1] everything in columns
    if(hipMemcpyAsync(big, j->maxabs.p, 16, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "hipMemcpy failed");
    }
    j->library_scans = opt.library_scans != 0;
    j->narrow = big[1] < (unsigned long long)PM_NARROW_INPUT_LIMIT && opt.coordinate_bits != 64;
    // positions beyond the limit (a chromosome, a concatenated assembly): they stay in 64 bits, and only they.  coordinate_bits = 32
    // asks for the int tables outright where they are valid at all, so it keeps the positions narrow when they fit
    j->wide_positions = j->narrow && big[0] >= (unsigned long long)PM_NARROW_INPUT_LIMIT;
    if(big[0] >= (1ull << 62)) { // (positions whose differences could wrap in 64 bits: the spans in big[1] mean nothing then)
      j->narrow = j->wide_positions = false;
    }
  }
  // Size the outputs once: the inputs of a job never change, so neither do its output sizes.
  JTRY(job_launch_pass(j, stream, false, nullptr));
  JTRY(job_read_totals(j, stream));
  if(j->narrow) {
    int trip = 0;
    if(hipMemcpy(&trip, j->narrow_trip.p, 4, hipMemcpyDeviceToHost) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "hipMemcpy failed");
    }
    if(trip) { // some unit's int merge left its checked range: this job runs on the int64 tables
      j->narrow = false;
      j->wide_positions = false;
      // the sizing pass no longer stores zero counts for units that are not live: were the wide filter ever to keep fewer units than
      // the narrow one did, a stale count would corrupt the prefix sums.  The redo starts from cleared counts (once per job).
      if(hipMemsetAsync(j->cnt_ent.p, 0, (size_t)(U + 1) * 8, stream) != hipSuccess ||
         hipMemsetAsync(j->cnt_off.p, 0, (size_t)(U + 1) * 8, stream) != hipSuccess) {
        pm_job_destroy(j);
        return fail(PM_E_HIP, "hipMemsetAsync failed");
      }
      JTRY(job_launch_pass(j, stream, false, nullptr));
      JTRY(job_read_totals(j, stream));
    }
  }
  if(j->n_entries < 0 || j->n_offsets < 0 || j->n_entries > ((i64)1 << 36) || j->n_offsets > ((i64)1 << 38)) {
    pm_job_destroy(j);
    return fail(PM_E_INVALID, "pm_job_create: implausible output size (inconsistent input tables)");
  }
  j->ent_cap = j->n_entries;
  j->off_cap = j->n_offsets;
  lap("sizing pass");
  // units for the FIX pass (none unless the tables contradict themselves): their list, and room for the gaps of a segment --
  // a gap owns at least one of the unit's offsets, so twice the unit's offset count in words is always enough
  {
    int any = 0;
    if(hipMemcpy(&any, (int *)j->slow_flag.p + U, 4, hipMemcpyDeviceToHost) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "hipMemcpy failed");
    }
    if(any) {
      std::vector<int> flag((size_t)U), pos((size_t)U);
      std::vector<i64> cnt((size_t)U); // (the counts are per live unit: a flagged unit's is cnt[pos[u]])
      if(hipMemcpy(flag.data(), j->slow_flag.p, (size_t)U * 4, hipMemcpyDeviceToHost) != hipSuccess ||
         hipMemcpy(pos.data(), j->live_pos.p, (size_t)U * 4, hipMemcpyDeviceToHost) != hipSuccess ||
         hipMemcpy(cnt.data(), j->cnt_off.p, (size_t)U * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        pm_job_destroy(j);
        return fail(PM_E_HIP, "hipMemcpy failed");
      }
      std::vector<int> list;
      std::vector<i64> at;
      i64 words = 0;
      for(i64 u = 0; u < U; ++u) {
        if(flag[(size_t)u]) {
          list.push_back((int)u);
          at.push_back(words);
          words += 2 * cnt[(size_t)pos[(size_t)u]] + 2;
        }
      }
      if(words > ((i64)1 << 28)) { // 2 GiB of scratch: not a job anyone means
        pm_job_destroy(j);
        return fail(PM_E_INVALID, "pm_job_create: implausible output size (inconsistent input tables)");
      }
      j->n_slow = (i64)list.size();
      JTRY(j->slow_units.upload(list.data(), list.size() * 4, stream));
      JTRY(j->slow_scratch_off.upload(at.data(), at.size() * 8, stream));
      JTRY(j->slow_scratch.alloc((size_t)(words > 0 ? words : 1) * 8));
      if(hipStreamSynchronize(stream) != hipSuccess) { // the uploads read these vectors
        pm_job_destroy(j);
        return fail(PM_E_HIP, "hipStreamSynchronize failed");
      }
    }
  }
  {
    int n_live = 0;
    if(hipMemcpy(&n_live, (int *)j->live_pos.p + U, 4, hipMemcpyDeviceToHost) != hipSuccess) {
      pm_job_destroy(j);
      return fail(PM_E_HIP, "hipMemcpy failed");
    }
    j->n_live = n_live;
    JTRY(j->states.alloc((size_t)(n_live > 0 ? n_live : 1) * sizeof(UnitState))); // the int layout needs less
  }
  // (+ 32: a whole-sector store of staged offsets never starts inside the array and ends outside it, but keep a sector of slack)
  JTRY(j->entries.alloc((size_t)j->ent_cap * (size_t)j->rec_bytes()));
  JTRY(j->offsets.alloc((size_t)j->off_cap * (size_t)j->off_bytes() + 32));
  // free the SoA staging copies
  j->left.raw_s.release();
  j->left.raw_e.release();
  j->left.raw_gs.release();
  j->left.raw_ge.release();
  j->right.raw_s.release();
  j->right.raw_e.release();
  j->right.raw_gs.release();
  j->right.raw_ge.release();
  for(int k = 0; k < 8; ++k) {
    j->deltas.raw[k].release();
  }
  // algorithmic input bytes: the tables the unit kernels can touch + unit triples
  // (coordinates are 8 bytes each, or 4 when the job runs on the int tables)
  {
    const i64 c = j->narrow ? 4 : 8;
    j->input_bytes = (j->left.n + j->right.n) * (2 * c + c + 8 + 4) + (j->left.G + j->right.G) * (2 * c + c) +
                     j->deltas.n * (4 * c + 16 + 4) + (j->deltas.Gr + j->deltas.Gq) * (2 * c + c) + U * 12;
  }
  lap("states + output buffers");
#undef JTRY
  *out = j;
  return PM_OK;
}

namespace pm {
int job_create_enumerating(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const EnumInput *en,
                           const pm_translate_options_t &opt, int device, pm_job_t **out) {
  return job_create_impl(left, right, deltas, nullptr, en, opt, device, out);
}
int job_unit_at(pm_job_t *j, int64_t unit, int32_t out[3]) {
  if(!j || unit < 0 || unit >= j->n_units) {
    return fail(PM_E_INVALID, "job_unit_at: no such unit");
  }
  PM_TRY(use_device(j->device));
  PM_HIP(hipMemcpy(&out[0], (const int *)j->u_delta.p + unit, 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(&out[1], (const int *)j->u_left.p + unit, 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(&out[2], (const int *)j->u_right.p + unit, 4, hipMemcpyDeviceToHost));
  return PM_OK;
}
} // namespace pm

extern "C" {

int pm_job_create(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units, int device,
                  pm_job_t **out) {
  return job_create_impl(left, right, deltas, units, nullptr, pm::translate_options(nullptr), device, out);
}

int pm_job_create_opt(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                      const pm_translate_options_t *options, int device, pm_job_t **out) {
  return job_create_impl(left, right, deltas, units, nullptr, pm::translate_options(options), device, out);
}

/* the job's unit list (made on the device when the job came from pm_job_create_from_workload) */
int pm_job_units(pm_job_t *j, int64_t *n_units, int32_t *delta, int32_t *left, int32_t *right) {
  if(!j) {
    return fail(PM_E_INVALID, "pm_job_units: null job");
  }
  PM_TRY(use_device(j->device));
  if(n_units) {
    *n_units = j->n_units;
  }
  const size_t bytes = (size_t)j->n_units * 4;
  if(bytes > 0) {
    if(delta) {
      PM_HIP(hipMemcpy(delta, j->u_delta.p, bytes, hipMemcpyDeviceToHost));
    }
    if(left) {
      PM_HIP(hipMemcpy(left, j->u_left.p, bytes, hipMemcpyDeviceToHost));
    }
    if(right) {
      PM_HIP(hipMemcpy(right, j->u_right.p, bytes, hipMemcpyDeviceToHost));
    }
  }
  return PM_OK;
}

int pm_job_run(pm_job_t *j, void *hip_stream) {
  if(!j) {
    return fail(PM_E_INVALID, "pm_job_run: null job");
  }
  int rc = use_device(j->device);
  if(rc) {
    return rc;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  PM_TRY(job_launch_pass(j, stream, true, nullptr));
  j->last_stream = stream;
  j->ran = true;
  return PM_OK;
}

int pm_job_run_profiled(pm_job_t *j, void *hip_stream, float *ms_filter, float *ms_count, float *ms_scan, float *ms_emit) {
  if(!j) {
    return fail(PM_E_INVALID, "pm_job_run_profiled: null job");
  }
  int rc = use_device(j->device);
  if(rc) {
    return rc;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  struct Events { // destroyed on every way out, failed launches included
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    ~Events() {
      for(int k = 0; k < 5; ++k) {
        if(ev[k]) {
          (void)hipEventDestroy(ev[k]);
        }
      }
    }
  } events;
  hipEvent_t *ev = events.ev;
  for(int k = 0; k < 5; ++k) {
    PM_HIP(hipEventCreate(&ev[k]));
  }
  PM_TRY(job_launch_pass(j, stream, true, ev));
  PM_HIP(hipEventSynchronize(ev[4]));
  float ms[4] = {0, 0, 0, 0};
  for(int k = 0; k < 4; ++k) {
    PM_HIP(hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]));
  }
  float *outp[4] = {ms_filter, ms_count, ms_scan, ms_emit};
  for(int k = 0; k < 4; ++k) {
    if(outp[k]) {
      *outp[k] = ms[k];
    }
  }
  j->last_stream = stream;
  j->ran = true;
  return PM_OK;
}

int pm_job_sizes(pm_job_t *j, int64_t *n_entries, int64_t *n_offsets) {
  if(!j) {
    return fail(PM_E_INVALID, "pm_job_sizes: null job");
  }
  int rc = use_device(j->device);
  if(rc) {
    return rc;
  }
  if(!j->ran) {
    return fail(PM_E_INVALID, "pm_job_sizes: pm_job_run has not been called");
  }
  PM_TRY(job_read_totals(j, j->last_stream));
  int ovf = 0;
  PM_HIP(hipMemcpy(&ovf, j->overflow.p, 4, hipMemcpyDeviceToHost));
  if(ovf || j->n_entries > j->ent_cap || j->n_offsets > j->off_cap) {
    return fail(PM_E_HIP, "output buffers smaller than this run's output (inputs changed under the job?)");
  }
  if(n_entries) {
    *n_entries = j->n_entries;
  }
  if(n_offsets) {
    *n_offsets = j->n_offsets;
  }
  return PM_OK;
}

int pm_job_fetch(pm_job_t *j, int32_t *unit_status, int64_t *unit_entry_off, pm_entry_t *entries, int64_t *offsets) {
  int64_t ne = 0, no = 0;
  int rc = pm_job_sizes(j, &ne, &no);
  if(rc) {
    return rc;
  }
  i64 U = j->n_units;
  std::vector<int32_t> st_local;
  int32_t *st = unit_status;
  if(!st) {
    st_local.resize((size_t)U);
    st = st_local.data();
  }
  if(U > 0) {
    PM_HIP(hipMemcpy(st, j->status.p, (size_t)U * 4, hipMemcpyDeviceToHost));
  }
  if(unit_entry_off) {
    PM_HIP(hipMemcpy(unit_entry_off, j->ent_off.p, (size_t)(U + 1) * 8, hipMemcpyDeviceToHost));
  }
  if(j->narrow) { // the job holds 32-byte records and int offsets: widened on the device into the C ABI's types, then copied
    if(entries && ne > 0) {
      DevBuf wide;
      PM_TRY(wide.alloc((size_t)ne * sizeof(pm_entry_t)));
      widen_entries_kernel<<<(unsigned)((ne + 255) / 256), 256>>>(ne, (const Entry32 *)j->entries.p, (pm_entry_t *)wide.p);
      PM_HIP(hipGetLastError());
      PM_HIP(hipMemcpy(entries, wide.p, (size_t)ne * sizeof(pm_entry_t), hipMemcpyDeviceToHost));
    }
    if(offsets && no > 0) {
      DevBuf wide;
      PM_TRY(wide.alloc((size_t)no * 8));
      widen_offsets_kernel<<<(unsigned)((no + 255) / 256), 256>>>(no, (const int *)j->offsets.p, (i64 *)wide.p);
      PM_HIP(hipGetLastError());
      PM_HIP(hipMemcpy(offsets, wide.p, (size_t)no * 8, hipMemcpyDeviceToHost));
    }
  }
  else {
    if(entries && ne > 0) {
      PM_HIP(hipMemcpy(entries, j->entries.p, (size_t)ne * sizeof(pm_entry_t), hipMemcpyDeviceToHost));
    }
    if(offsets && no > 0) {
      PM_HIP(hipMemcpy(offsets, j->offsets.p, (size_t)no * 8, hipMemcpyDeviceToHost));
    }
  }
  for(i64 u = 0; u < U; ++u) {
    if(st[u] != PM_ST_OK) {
      char msg[128];
      snprintf(msg, sizeof msg, "unit %lld ended with status %d", (long long)u, (int)st[u]);
      return fail(PM_E_UNIT, msg);
    }
  }
  return PM_OK;
}

// One side's major names as the text kernels take them: bytes back to back, offsets, and an id per row (equal names, equal ids).
static int upload_names(pm_job *j, int side, const char *const *names, i64 n) {
  std::vector<i64> off((size_t)n + 1, 0);
  std::vector<int> id((size_t)n, 0);
  std::string blob;
  std::unordered_map<std::string, int> seen;
  for(i64 r = 0; r < n; ++r) {
    if(!names[r]) {
      return fail(PM_E_INVALID, "pm_job_text: null name");
    }
    const std::string name(names[r]);
    blob += name;
    off[(size_t)r + 1] = (i64)blob.size();
    id[(size_t)r] = seen.emplace(name, (int)seen.size()).first->second;
  }
  PM_TRY(j->t_names[side].upload(blob.data(), blob.size(), nullptr));
  PM_TRY(j->t_name_off[side].upload(off.data(), off.size() * 8, nullptr));
  PM_TRY(j->t_name_id[side].upload(id.data(), id.size() * 4, nullptr));
  return PM_OK;
}

int pm_job_text(pm_job_t *j, const char *const *left_major, const char *const *right_major, int64_t *n_bytes, int64_t *failed_unit,
                int32_t *failed_status) {
  return guarded("pm_job_text", [&]() -> int {
  if(!j || !n_bytes || (j->left.n > 0 && !left_major) || (j->right.n > 0 && !right_major)) {
    return fail(PM_E_INVALID, "pm_job_text: null argument");
  }
  PM_TRY(use_device(j->device));
  int64_t E = 0, O = 0;
  PM_TRY(pm_job_sizes(j, &E, &O)); // waits for the run; refuses a job that has not run
  const i64 U = j->n_units;
  hipStream_t stream = j->last_stream;
  PM_TRY(upload_names(j, 0, left_major, j->left.n));
  PM_TRY(upload_names(j, 1, right_major, j->right.n));
  PM_TRY(j->t_word.alloc(8));
  const int no_fail = 0x7fffffff;
  PM_HIP(hipMemcpyAsync(j->t_word.p, &no_fail, 4, hipMemcpyHostToDevice, stream));
  PM_TRY(j->t_marks.alloc((size_t)(U + 1) * 4));
  PM_TRY(j->t_last_ne.alloc((size_t)(U + 1) * 4));
  PM_TRY(j->t_eunit.alloc((size_t)(E + 1) * 4));
  PM_TRY(j->t_len.alloc((size_t)(E + 1) * 8));
  PM_TRY(j->t_pos.alloc((size_t)(E + 1) * 8));
  j->text_bytes = 0;
  if(U > 0) {
    text_units_kernel<<<(unsigned)((U + 255) / 256), 256, 0, stream>>>(U, (const int *)j->status.p, (const i64 *)j->ent_off.p, (int *)j->t_marks.p,
                                                                        (int *)j->t_word.p);
    PM_HIP(hipGetLastError());
    size_t tmp = 0;
    PM_HIP(rocprim::inclusive_scan(nullptr, tmp, (int *)j->t_marks.p, (int *)j->t_last_ne.p, (size_t)U, rocprim::maximum<int>(), stream));
    size_t tmp2 = 0;
    PM_HIP(rocprim::exclusive_scan(nullptr, tmp2, (i64 *)j->t_len.p, (i64 *)j->t_pos.p, (i64)0, (size_t)(E + 1), rocprim::plus<i64>(), stream));
    PM_TRY(j->t_scan_tmp.alloc(std::max(tmp, tmp2) + 16));
    tmp = j->t_scan_tmp.bytes;
    PM_HIP(rocprim::inclusive_scan(j->t_scan_tmp.p, tmp, (int *)j->t_marks.p, (int *)j->t_last_ne.p, (size_t)U, rocprim::maximum<int>(), stream));
  }
  TextNames names;
  for(int sd = 0; sd < 2; ++sd) {
    names.bytes[sd] = (const char *)j->t_names[sd].p;
    names.off[sd] = (const i64 *)j->t_name_off[sd].p;
    names.id[sd] = (const int *)j->t_name_id[sd].p;
  }
  names.length[0] = (const i64 *)j->left.length.p;
  names.length[1] = (const i64 *)j->right.length.p;
  PM_HIP(hipMemsetAsync((i64 *)j->t_len.p + E, 0, 8, stream));
  if(E > 0) {
#define PM_TEXT_LAUNCH(WRITE, I, LEN, POS, TEXT)                                                                                                   \
  text_entries_kernel<WRITE, I><<<(unsigned)((E + 255) / 256), 256, 0, stream>>>(                                                                   \
      E, U, (const i64 *)j->ent_off.p, (const int *)j->u_left.p, (const int *)j->u_right.p, (const int *)j->t_last_ne.p, (const int *)j->t_word.p, \
      (const EntRecT<I>::type *)j->entries.p, (const I *)j->offsets.p, names, (int *)j->t_eunit.p, LEN, POS, TEXT)
    if(j->narrow) {
      PM_TEXT_LAUNCH(false, int, (i64 *)j->t_len.p, nullptr, nullptr);
    }
    else {
      PM_TEXT_LAUNCH(false, i64, (i64 *)j->t_len.p, nullptr, nullptr);
    }
    PM_HIP(hipGetLastError());
    size_t tmp = j->t_scan_tmp.bytes;
    PM_HIP(rocprim::exclusive_scan(j->t_scan_tmp.p, tmp, (i64 *)j->t_len.p, (i64 *)j->t_pos.p, (i64)0, (size_t)(E + 1), rocprim::plus<i64>(), stream));
    i64 total = 0;
    PM_HIP(hipMemcpyAsync(&total, (i64 *)j->t_pos.p + E, 8, hipMemcpyDeviceToHost, stream));
    PM_HIP(hipStreamSynchronize(stream));
    if(total < 0 || total > ((i64)1 << 40)) {
      return fail(PM_E_INVALID, "pm_job_text: implausible text size");
    }
    if((i64)j->t_text.bytes < total) {
      PM_TRY(j->t_text.alloc((size_t)total));
    }
    if(j->narrow) {
      PM_TEXT_LAUNCH(true, int, nullptr, (const i64 *)j->t_pos.p, (char *)j->t_text.p);
    }
    else {
      PM_TEXT_LAUNCH(true, i64, nullptr, (const i64 *)j->t_pos.p, (char *)j->t_text.p);
    }
#undef PM_TEXT_LAUNCH
    PM_HIP(hipGetLastError());
    j->text_bytes = total;
  }
  int ff = no_fail;
  PM_HIP(hipMemcpyAsync(&ff, j->t_word.p, 4, hipMemcpyDeviceToHost, stream));
  PM_HIP(hipStreamSynchronize(stream));
  *n_bytes = j->text_bytes;
  if(failed_unit) {
    *failed_unit = ff == no_fail ? -1 : ff;
  }
  if(failed_status) {
    *failed_status = 0;
    if(ff != no_fail) {
      PM_HIP(hipMemcpy(failed_status, (int *)j->status.p + ff, 4, hipMemcpyDeviceToHost));
    }
  }
  return PM_OK;
  });
}

int pm_job_text_fetch_range(pm_job_t *j, char *out, int64_t first, int64_t n) {
  if(!j || j->text_bytes < 0 || first < 0 || n < 0 || first + n > j->text_bytes || (n > 0 && !out)) {
    return fail(PM_E_INVALID, "pm_job_text_fetch_range: bad range (or pm_job_text has not been called)");
  }
  PM_TRY(use_device(j->device));
  if(n > 0) {
    PM_HIP(hipMemcpy(out, (const char *)j->t_text.p + first, (size_t)n, hipMemcpyDeviceToHost));
  }
  return PM_OK;
}

int pm_job_text_fetch(pm_job_t *j, char *out) {
  if(!j || j->text_bytes < 0 || (j->text_bytes > 0 && !out)) {
    return fail(PM_E_INVALID, "pm_job_text_fetch: pm_job_text has not been called");
  }
  PM_TRY(use_device(j->device));
  if(j->text_bytes > 0) {
    PM_HIP(hipMemcpy(out, j->t_text.p, (size_t)j->text_bytes, hipMemcpyDeviceToHost));
  }
  return PM_OK;
}

int pm_job_algorithmic_bytes(pm_job_t *j, int64_t *bytes) {
  if(!j || !bytes) {
    return fail(PM_E_INVALID, "pm_job_algorithmic_bytes: null argument");
  }
  // inputs once + per unit status/counts/offsets + entries + offsets
  *bytes = j->input_bytes + j->n_units * (4 + 16 + 16) + j->ent_cap * j->rec_bytes() + j->off_cap * j->off_bytes();
  return PM_OK;
}

int pm_job_kernel_bytes(pm_job_t *j, int64_t *count_bytes, int64_t *emit_bytes, int64_t *n_live) {
  if(!j) {
    return fail(PM_E_INVALID, "pm_job_kernel_bytes: null job");
  }
  int live = 0;
  PM_HIP(hipMemcpy(&live, (int *)j->live_pos.p + j->n_units, 4, hipMemcpyDeviceToHost));
  const int64_t state_bytes = j->narrow ? (int64_t)sizeof(UnitStateT<int>) : (int64_t)sizeof(UnitState);
  // count pass: the tables once, the live list, status + two counts per live unit, one UnitState per live unit
  if(count_bytes) {
    *count_bytes = j->input_bytes + (int64_t)live * (4 + 4 + 16 + state_bytes);
  }
  // emit pass: the live list and the saved states, the gap lists again, output offsets, entries and offsets written
  if(emit_bytes) {
    *emit_bytes = (int64_t)live * (4 + 32 + state_bytes) + (j->left.G + j->right.G + j->deltas.Gr + j->deltas.Gq) * (j->narrow ? 8 : 16) +
                  j->ent_cap * j->rec_bytes() + j->off_cap * j->off_bytes();
  }
  if(n_live) {
    *n_live = live;
  }
  return PM_OK;
}

int pm_job_coordinate_bits(pm_job_t *j, int *bits) {
  if(!j || !bits) {
    return fail(PM_E_INVALID, "pm_job_coordinate_bits: null argument");
  }
  *bits = j->narrow ? 32 : 64;
  return PM_OK;
}

int pm_job_position_bits(pm_job_t *j, int *bits) {
  if(!j || !bits) {
    return fail(PM_E_INVALID, "pm_job_position_bits: null argument");
  }
  *bits = j->narrow && !j->wide_positions ? 32 : 64;
  return PM_OK;
}

void pm_job_destroy(pm_job_t *j) {
  if(!j) {
    return;
  }
  (void)hipSetDevice(j->device);
  delete j;
}

static int rows_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *in, int64_t *out, int32_t *status, int device,
                      bool p2s) {
  if(n < 0 || (n > 0 && (!row || !in || !out || !status))) {
    return fail(PM_E_INVALID, "rows batch: null array");
  }
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  if(!rows) {
    return fail(PM_E_INVALID, "rows batch: null rows");
  }
  for(int64_t q = 0; q < n; ++q) {
    if(row[q] < 0 || row[q] >= rows->n) {
      return fail(PM_E_INVALID, "rows batch: row index out of range");
    }
  }
  RowsStore s;
  hipStream_t stream = nullptr;
  PM_TRY(upload_rows(rows, s, stream));
  DevBuf d_row, d_in, d_out, d_st;
  PM_TRY(d_row.upload(row, (size_t)n * 4, stream));
  PM_TRY(d_in.upload(in, (size_t)n * 8, stream));
  PM_TRY(d_out.alloc((size_t)n * 8));
  PM_TRY(d_st.alloc((size_t)n * 4));
  if(n > 0) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if(p2s) {
      p2s_batch_kernel<<<blocks, 256, 0, stream>>>(s.view(), n, (const int *)d_row.p, (const i64 *)d_in.p, (i64 *)d_out.p, (int *)d_st.p);
    }
    else {
      s2p_batch_kernel<<<blocks, 256, 0, stream>>>(s.view(), n, (const int *)d_row.p, (const i64 *)d_in.p, (i64 *)d_out.p, (int *)d_st.p);
    }
    PM_HIP(hipGetLastError());
    PM_HIP(hipMemcpy(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    PM_HIP(hipMemcpy(status, d_st.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  return PM_OK;
}

int pm_rows_profile_idx_of_seq_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *seq_idx,
                                         int64_t *profile_idx, int32_t *status, int device) {
  return rows_batch(rows, n, row, seq_idx, profile_idx, status, device, true);
}

int pm_rows_seq_idx_of_profile_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *profile_idx,
                                         int64_t *seq_idx, int32_t *status, int device) {
  return rows_batch(rows, n, row, profile_idx, seq_idx, status, device, false);
}

} // extern "C"
