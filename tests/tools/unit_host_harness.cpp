// tests/tools/unit_host_harness.cpp -- TEST TOOL.  Runs the device-side unit arithmetic of
// paramugsy_amd/csrc/translate_device.hpp on the CPU (the functions are __host__ __device__), so that the
// exact code the kernels execute can be checked against the oracle without a GPU and under
// -fsanitize=address,undefined (GPU sanitizers are not available on the pool).  Built by
// tests/test_device_code_on_host.py with hipcc; never part of libparamugsy_amd.so.
//
// C entry point: same flat tables as pm_job_create, same outputs as pm_job_fetch.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../paramugsy_amd/csrc/translate_device.hpp"

using namespace pm;

namespace {

template <typename I, typename P>
struct HostRows {
  std::vector<R2T<P>> range;
  std::vector<R2T<I>> gaps;
  std::vector<I> length, pre;
  std::vector<i64> gap_off;
  std::vector<int> bad;
  RowsT<I, P> view() const {
    RowsT<I, P> d;
    d.n = (i64)range.size();
    d.range = range.data();
    d.length = length.data();
    d.gap_off = gap_off.data();
    d.gaps = gaps.data();
    d.pre = pre.data();
    d.bad = bad.data();
    return d;
  }
};

// the same construction as prepare_rows_kernel
template <typename I, typename P>
void prepare_rows(const pm_rows_t *h, HostRows<I, P> &s) {
  i64 n = h->n, G = h->gap_off[n];
  s.range.resize(n);
  s.length.resize(n);
  for(i64 r = 0; r < n; ++r) {
    s.length[r] = (I)h->length[r];
  }
  s.gap_off.assign(h->gap_off, h->gap_off + n + 1);
  s.gaps.resize(G + 1);
  s.pre.resize(G + n + 1);
  s.bad.resize(n + 1);
  for(i64 r = 0; r < n; ++r) {
    s.range[r] = R2T<P>{(P)h->start[r], (P)h->end[r]};
    i64 o = h->gap_off[r], m = h->gap_off[r + 1] - o;
    I *p = s.pre.data() + o + r;
    I acc = 0, prev_end = 0;
    int flag = 0;
    for(i64 k = 0; k < m; ++k) {
      R2T<I> g{(I)h->gap_start[o + k], (I)h->gap_end[o + k]};
      if(g.s > g.e || (k > 0 && g.s <= prev_end)) {
        flag = 1;
      }
      prev_end = g.e;
      s.gaps[o + k] = g;
      p[k] = acc;
      acc += rlen(g);
    }
    p[m] = acc;
    s.bad[r] = flag;
  }
}

template <typename I, typename P>
struct HostDeltas {
  std::vector<R2T<P>> ref, qry;
  std::vector<R2T<I>> rg[2], qg[2];
  std::vector<i64> ref_off, qry_off;
  std::vector<I> rp[2], qp[2];
  std::vector<int> bad;
};

// the same construction as prepare_deltas_kernel
template <typename I, typename P>
void prepare_strand(i64 n, const int64_t *rs, const int64_t *re, const int64_t *off, const int64_t *gs, const int64_t *ge,
                    std::vector<R2T<P>> &range, std::vector<R2T<I>> &gf, std::vector<I> &pf, std::vector<R2T<I>> &gr, std::vector<I> &pr,
                    std::vector<int> &bad) {
  i64 G = off[n];
  range.resize(n);
  gf.resize(G + 1);
  gr.resize(G + 1);
  pf.resize(G + n + 1);
  pr.resize(G + n + 1);
  for(i64 d = 0; d < n; ++d) {
    R2T<P> rg{(P)rs[d], (P)re[d]};
    range[d] = rg;
    i64 o = off[d], m = off[d + 1] - o;
    I acc = 0, prev_end = 0;
    for(i64 k = 0; k < m; ++k) {
      R2T<I> g{(I)gs[o + k], (I)ge[o + k]};
      if(g.s > g.e || (k > 0 && g.s <= prev_end)) {
        bad[d] = 1;
      }
      prev_end = g.e;
      gf[o + k] = g;
      pf[o + d + k] = acc;
      acc += rlen(g);
    }
    pf[o + d + m] = acc;
    I columns = (I)rlen(rg) + acc, racc = 0;
    for(i64 k = 0; k < m; ++k) {
      R2T<I> g{(I)gs[o + (m - 1 - k)], (I)ge[o + (m - 1 - k)]};
      R2T<I> mg{columns - g.e + 1, columns - g.s + 1};
      gr[o + k] = mg;
      pr[o + d + k] = racc;
      racc += rlen(mg);
    }
    pr[o + d + m] = racc;
  }
}

template <typename I, typename P>
int run_all(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                             int32_t *status, int64_t *unit_entry_off, int64_t *n_entries, int64_t *n_offsets, pm_entry_t *entries,
                             int64_t entries_cap, int64_t *offsets, int64_t offsets_cap) {
  HostRows<I, P> L, R;
  prepare_rows(left, L);
  prepare_rows(right, R);
  HostDeltas<I, P> D;
  i64 n = deltas->n;
  D.bad.assign(n + 1, 0);
  D.ref_off.assign(deltas->ref_gap_off, deltas->ref_gap_off + n + 1);
  D.qry_off.assign(deltas->qry_gap_off, deltas->qry_gap_off + n + 1);
  prepare_strand(n, deltas->ref_start, deltas->ref_end, deltas->ref_gap_off, deltas->ref_gap_start, deltas->ref_gap_end, D.ref, D.rg[0],
                 D.rp[0], D.rg[1], D.rp[1], D.bad);
  prepare_strand(n, deltas->qry_start, deltas->qry_end, deltas->qry_gap_off, deltas->qry_gap_start, deltas->qry_gap_end, D.qry, D.qg[0],
                 D.qp[0], D.qg[1], D.qp[1], D.bad);
  DeltasT<I, P> dv;
  dv.n = n;
  dv.ref = D.ref.data();
  dv.qry = D.qry.data();
  dv.ref_off = D.ref_off.data();
  dv.qry_off = D.qry_off.data();
  for(int o = 0; o < 2; ++o) {
    dv.ref_gaps[o] = D.rg[o].data();
    dv.ref_pre[o] = D.rp[o].data();
    dv.qry_gaps[o] = D.qg[o].data();
    dv.qry_pre[o] = D.qp[o].data();
  }
  dv.bad = D.bad.data();
  RowsT<I, P> lv = L.view(), rv = R.view();
  i64 U = units->n;
  std::vector<i64> cnt_e(U + 1, 0), cnt_o(U + 1, 0);
  std::vector<char> disorder(U + 1, 0);
  for(i64 u = 0; u < U; ++u) { // COUNT pass
    Sink<false, I> sink;
    memset(&sink, 0, sizeof sink);
    status[u] = run_unit<false>(lv, rv, dv, units->delta[u], units->left[u], units->right[u], sink);
    cnt_e[u] = sink.n_ent;
    cnt_o[u] = sink.n_off;
    disorder[u] = (char)sink.disorder;
  }
  std::vector<i64> eo(U + 1, 0), oo(U + 1, 0);
  for(i64 u = 0; u < U; ++u) {
    eo[u + 1] = eo[u] + cnt_e[u];
    oo[u + 1] = oo[u] + cnt_o[u];
  }
  memcpy(unit_entry_off, eo.data(), (size_t)(U + 1) * 8);
  *n_entries = eo[U];
  *n_offsets = oo[U];
  if(eo[U] > entries_cap || oo[U] > offsets_cap) {
    return 1; // caller re-calls with bigger buffers
  }
  // the EMIT pass writes its own record and offset types (Entry32 + int in the int instantiation); widened below for the caller
  typedef typename EntRecT<I>::type Rec;
  std::vector<Rec> rec((size_t)eo[U] + 1);
  std::vector<I> off((size_t)oo[U] + 8);
  // the wavefront's window of offsets (LDS on the device): here a "wavefront" of one unit and a window of five slots, so that units
  // are written through the window, past it, and both
  constexpr int WIN = 5;
  I win[WIN];
  for(i64 u = 0; u < U; ++u) { // EMIT pass
    Sink<true, I> sink;
    memset(&sink, 0, sizeof sink);
    sink.ent = rec.data() + eo[u];
    sink.ent_cap = (I)(eo[u + 1] - eo[u]);
    sink.off = off.data();
    sink.off_base = oo[u];
    sink.off_cap = (I)(oo[u + 1] - oo[u]);
    sink.win = win;
    sink.win_lo = oo[u];
    sink.win_cap = WIN;
    for(int k = 0; k < WIN; ++k) {
      win[k] = (I)0x5a5a5a5a; // whatever the previous unit left must not matter
    }
    int st = run_unit<true>(lv, rv, dv, units->delta[u], units->left[u], units->right[u], sink);
    if(st != status[u]) {
      return 2; // the two passes must agree
    }
    for(i64 q = 0; q < WIN && q < oo[u + 1] - oo[u]; ++q) { // what the wavefront does when its lanes are done
      off[(size_t)(oo[u] + q)] = win[q];
    }
    if(disorder[u]) { // the FIX pass of the library: the same unit again, its segments' gaps recorded and merged at commit
      std::vector<i64> scratch((size_t)(2 * (oo[u + 1] - oo[u]) + 2), 0);
      Sink<true, I> fx;
      memset(&fx, 0, sizeof fx);
      fx.ent = rec.data() + eo[u];
      fx.ent_cap = (I)(eo[u + 1] - eo[u]);
      fx.off = off.data();
      fx.off_base = oo[u];
      fx.off_cap = (I)(oo[u + 1] - oo[u]);
      fx.fix = scratch.data();
      fx.fix_cap = fx.off_cap + 1;
      st = run_unit<true>(lv, rv, dv, units->delta[u], units->left[u], units->right[u], fx);
      if(st != status[u]) {
        return 2;
      }
    }
  }
  for(i64 e = 0; e < eo[U]; ++e) {
    entries[e].ref_start = rec[(size_t)e].ref_start;
    entries[e].ref_end = rec[(size_t)e].ref_end;
    entries[e].qry_start = rec[(size_t)e].qry_start;
    entries[e].qry_end = rec[(size_t)e].qry_end;
    entries[e].offset_begin = rec[(size_t)e].offset_begin;
    entries[e].n_offsets = rec[(size_t)e].n_offsets;
  }
  for(i64 o = 0; o < oo[U]; ++o) {
    offsets[o] = off[(size_t)o];
  }
  return 0;
}

} // namespace

// How much merge every unit has in front of it (the kept gaps of its four lists: what unit_merge's step budget is made of), -1 for a
// unit the filter pass drops or whose set-up ends it: for sizing what a wavefront of 64 consecutive live units waits for.
extern "C" int unit_host_work(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units, int32_t *work) {
  typedef i64 I;
  HostRows<I, I> L, R;
  prepare_rows(left, L);
  prepare_rows(right, R);
  HostDeltas<I, I> D;
  i64 n = deltas->n;
  D.bad.assign(n + 1, 0);
  D.ref_off.assign(deltas->ref_gap_off, deltas->ref_gap_off + n + 1);
  D.qry_off.assign(deltas->qry_gap_off, deltas->qry_gap_off + n + 1);
  prepare_strand(n, deltas->ref_start, deltas->ref_end, deltas->ref_gap_off, deltas->ref_gap_start, deltas->ref_gap_end, D.ref, D.rg[0],
                 D.rp[0], D.rg[1], D.rp[1], D.bad);
  prepare_strand(n, deltas->qry_start, deltas->qry_end, deltas->qry_gap_off, deltas->qry_gap_start, deltas->qry_gap_end, D.qry, D.qg[0],
                 D.qp[0], D.qg[1], D.qp[1], D.bad);
  DeltasT<I, I> dv;
  dv.n = n;
  dv.ref = D.ref.data();
  dv.qry = D.qry.data();
  dv.ref_off = D.ref_off.data();
  dv.qry_off = D.qry_off.data();
  for(int o = 0; o < 2; ++o) {
    dv.ref_gaps[o] = D.rg[o].data();
    dv.ref_pre[o] = D.rp[o].data();
    dv.qry_gaps[o] = D.qg[o].data();
    dv.qry_pre[o] = D.qp[o].data();
  }
  dv.bad = D.bad.data();
  RowsT<I, I> lv = L.view(), rv = R.view();
  for(i64 u = 0; u < units->n; ++u) {
    PVT<I, I> lp, rp, dr, dq;
    R2T<I> cols;
    bool live, proceed = false;
    int orientation;
    work[u] = -1;
    int st = unit_prefix(lv, rv, dv, units->delta[u], units->left[u], units->right[u], lp, rp, dr, dq, cols, live, orientation);
    if(st || !live) {
      continue;
    }
    work[u] = 0; // live: a lane of the count and emit passes
    Merge<false, I> m;
    memset(&m, 0, sizeof m);
    st = unit_setup<false>(lp, rp, dr, dq, cols, m, proceed);
    if(!st && proceed) {
      work[u] = 1 + m.rows.v0.n + m.rows.v1.n + m.delta.v0.n + m.delta.v1.n;
    }
  }
  return 0;
}

extern "C" int unit_host_run(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                             int32_t *status, int64_t *unit_entry_off, int64_t *n_entries, int64_t *n_offsets, pm_entry_t *entries,
                             int64_t entries_cap, int64_t *offsets, int64_t offsets_cap) {
  return run_all<i64, i64>(left, right, deltas, units, status, unit_entry_off, n_entries, n_offsets, entries, entries_cap, offsets, offsets_cap);
}

// The int instantiation (the fast path of jobs whose tables are all below 2^25): same tables narrowed, same outputs.
// A status of PM_ST_NARROW (100) means the merge's range check tripped and the library would redo the job in int64.
extern "C" int unit_host_run_narrow(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                                    int32_t *status, int64_t *unit_entry_off, int64_t *n_entries, int64_t *n_offsets,
                                    pm_entry_t *entries, int64_t entries_cap, int64_t *offsets, int64_t offsets_cap) {
  return run_all<int, int>(left, right, deltas, units, status, unit_entry_off, n_entries, n_offsets, entries, entries_cap, offsets, offsets_cap);
}

// int columns, 64-bit sequence positions (round 5: a job whose positions need the `long` while its rows are short).
extern "C" int unit_host_run_wide_positions(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                                            int32_t *status, int64_t *unit_entry_off, int64_t *n_entries, int64_t *n_offsets,
                                            pm_entry_t *entries, int64_t entries_cap, int64_t *offsets, int64_t offsets_cap) {
  return run_all<int, i64>(left, right, deltas, units, status, unit_entry_off, n_entries, n_offsets, entries, entries_cap, offsets, offsets_cap);
}
