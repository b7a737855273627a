// translate_device.hpp -- device-side arithmetic of the translate path (gfx950).
//
// One lane owns one (delta entry x left row x right row) work unit and walks the reference's algorithm
// without allocating: every "sub profile" of the reference (a copied gap vector) is a view
// {parent gap array, index range, clip window, optional mirror}; every linear scan over a gap list
// (lib/profiles_lib/m_profile.cc:91-149,160-206) is a binary search over a per-gap prefix table built
// once per batch (translate_prepare_*).  The searches equal the scans on gap lists that are ascending and
// disjoint, which is what both producers of gap lists emit (lib/profiles/m_profile.ml:29-47,
// lib/profiles_lib/m_delta.cc:50-68); lists that are not are flagged at prepare time and every unit
// touching one ends in PM_ST_MALFORMED_INPUT instead of a different answer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/paramugsy_amd.h"

// The unit arithmetic is plain integer code; marking it host+device lets tests/tools/unit_host_harness.cpp run
// the very same functions on the CPU under sanitizers (GPU sanitizers are not available on the pool).  The
// library itself only ever calls them from kernels.
#define PM_HD __host__ __device__

namespace pm {

typedef long long i64;

// Everything below is a template over the coordinate type I.  I = long long is the reference's `long` and always
// valid.  I = int is the fast path the job takes when every number in its tables is below 2^25 in magnitude
// (bacterial genomes, the reference's domain, are a few Mbp): half the registers per lane, so nearly twice the
// resident wavefronts for a kernel that lives on latency hiding, and half the instructions.  It computes the same
// values as long as no intermediate leaves int: the searches and conversions only add or subtract a few table
// values, and the merge, which accumulates, checks what it carries after every step (Merge::narrow_overflow) and
// reports PM_ST_NARROW if anything leaves +-2^26; a job that sees PM_ST_NARROW anywhere is redone with
// I = long long (pm_job_create).
// A second type, P, is what SEQUENCE POSITIONS are held in (a row's range, an entry's two ranges, and what is derived from them inside
// unit_prefix / unit_setup); everything else -- lengths, gap columns, prefix tables, the whole merge and what it writes -- is in
// profile or entry COLUMNS, type I.  P = I for the two cases above.  I = int with P = long long (round 5) is the job whose
// positions need the reference's `long` (a chromosome beyond 2^25) while every row, entry and gap list is short: positions enter
// only as differences from a row's or entry's start, which are bounded by the row's length -- so the set-up subtracts in 64 bits, the
// difference is an int, and from there on it is the int job (its emit pass IS the int job's emit pass: no position in it).
#define PM_ST_NARROW 100 /* internal: never leaves the library */
#define PM_NARROW_INPUT_LIMIT (1 << 25)

template <typename I>
struct alignas(2 * sizeof(I)) R2T {
  I s;
  I e;
};
typedef R2T<i64> R2;

// Rows of one side, device resident.
template <typename I, typename P = I>
struct RowsT {
  i64 n;
  const R2T<P> *range; // [n]
  const I *length;     // [n]
  const i64 *gap_off;  // [n+1]
  const R2T<I> *gaps;  // [gap_off[n]]
  const I *pre;        // [gap_off[n] + n]: row r owns pre[gap_off[r] + r .. + n_r], pre[k] = gap columns before gap k, last = total
  const int *bad;      // [n] 1 when the row's gap list is not ascending+disjoint
};
typedef RowsT<i64> RowsD;

// Delta entries, device resident, in both orientations (o = 0 as read, 1 = M_delta_entry::reverse, m_delta.cc:94-146).
template <typename I, typename P = I>
struct DeltasT {
  i64 n;
  const R2T<P> *ref;  // [n] as read
  const R2T<P> *qry;  // [n]
  const i64 *ref_off; // [n+1]
  const i64 *qry_off; // [n+1]
  const R2T<I> *ref_gaps[2];
  const I *ref_pre[2]; // same indexing rule as RowsT::pre
  const R2T<I> *qry_gaps[2];
  const I *qry_pre[2];
  const int *bad;
};
typedef DeltasT<i64> DeltasD;

template <typename I>
PM_HD __forceinline__ I rlen(R2T<I> r) { return (r.s <= r.e ? r.e - r.s : r.s - r.e) + 1; } // m_range.hh:34
template <typename I>
PM_HD __forceinline__ bool fwd(R2T<I> r) { return r.s <= r.e; }                               // m_range.hh:36
template <typename I>
PM_HD __forceinline__ R2T<I> fwd_of(R2T<I> r) { return fwd(r) ? r : R2T<I>{r.e, r.s}; }
template <typename I>
PM_HD __forceinline__ I imax(I a, I b) { return a > b ? a : b; }
template <typename I>
PM_HD __forceinline__ I imin(I a, I b) { return a < b ? a : b; }

// m_range.hh:80-94
template <typename I>
PM_HD __forceinline__ bool overlap(R2T<I> a, R2T<I> b, R2T<I> &o) {
  R2T<I> fa = fwd_of(a), fb = fwd_of(b);
  o.s = imax(fa.s, fb.s);
  o.e = imin(fa.e, fb.e);
  return o.e - o.s >= 0;
}

// A profile as the conversions see it: range, p_length, gap list + prefix table.
template <typename I, typename P = I>
struct PVT {
  R2T<P> range;
  I len;
  const R2T<I> *g;
  const I *pre; // n+1 entries
  int n;
};
typedef PVT<i64> PV;

// a3, m_profile.cc:91-112.  The scan adds gap k while gap[k].s <= offset + pre[k] and stops at the first
// failure; gap[k].s - pre[k] is non-decreasing on ascending disjoint lists, so that first failure is a
// lower bound.
// `at` = the lower bound itself: the gaps before index `at` lie wholly before the returned column, gap `at` (if any)
// starts after it.
template <typename I, typename P>
PM_HD inline int profile_idx_of_seq_idx(const PVT<I, P> &p, P si, I &out, int &at) {
  R2T<P> f = fwd_of(p.range);
  if(!(f.s <= si && si <= f.e)) {
    return PM_ST_SEQ_IDX_OUT_OF_RANGE;
  }
  P d = p.range.s - si;
  I offset = (I)((d < 0 ? -d : d) + 1); // within the row: at most its length
  int lo = 0, hi = p.n;
  while(lo < hi) {
    int mid = (lo + hi) >> 1;
    if(p.g[mid].s - p.pre[mid] <= offset) {
      lo = mid + 1;
    }
    else {
      hi = mid;
    }
  }
  out = p.pre[lo] + offset;
  at = lo;
  return PM_ST_OK;
}

template <typename I, typename P>
PM_HD inline int profile_idx_of_seq_idx(const PVT<I, P> &p, P si, I &out) {
  int at;
  return profile_idx_of_seq_idx(p, si, out, at);
}

// a4, m_profile.cc:114-149.  First gap whose end is >= pi decides: inside it -> none, else the gaps
// before it are skipped.
template <typename I, typename P>
PM_HD inline int seq_idx_of_profile_idx(const PVT<I, P> &p, I pi, P &out, bool &none) {
  none = false;
  if(!(pi < p.len + 1)) {
    return PM_ST_PROFILE_IDX_OUT_OF_RANGE;
  }
  int lo = 0, hi = p.n;
  while(lo < hi) {
    int mid = (lo + hi) >> 1;
    if(p.g[mid].e < pi) {
      lo = mid + 1;
    }
    else {
      hi = mid;
    }
  }
  if(lo < p.n && p.g[lo].s <= pi) {
    none = true;
    return PM_ST_OK;
  }
  I offset = pi - p.pre[lo] - 1;
  out = fwd(p.range) ? p.range.s + (P)offset : p.range.s - (P)offset;
  return PM_ST_OK;
}

// The kept gaps of subset_profile as a view on the parent list.
template <typename I>
struct GapViewT {
  const R2T<I> *g; // parent list
  int lo;      // first kept gap
  int n;       // kept gaps
  I ws, we;  // clip window (the subset's s..e after ordering)
  bool mirror; // walk backwards and map column c -> L - c + 1 (m_translate.cc:559-570)
  I L;
};
typedef GapViewT<i64> GapView;

// (One load whichever way the view runs, the rest by selects: with a branch per direction the merge's pops -- executed for the few
// lanes that are at that point of that case -- were two loads each where the lanes of a wavefront differ, and what the merge costs is
// what its divergent lanes issue between them: profiles/r05_translate_merge.txt.)
template <typename I>
PM_HD __forceinline__ R2T<I> view_get(const GapViewT<I> &v, int i) {
  R2T<I> r = v.g[v.lo + (v.mirror ? v.n - 1 - i : i)];
  r = R2T<I>{imax(r.s, v.ws), imin(r.e, v.we)};
  const R2T<I> m{v.L - r.e + 1, v.L - r.s + 1};
  return v.mirror ? m : r;
}

// a5, m_profile.cc:160-206.  On an ascending disjoint list the gaps overlapping [s,e] are one index range.
// Returns status; `none` mirrors the reference's empty option; seq = the sub profile's p_range.
template <typename I, typename P>
PM_HD inline int subset_profile(const PVT<I, P> &p, I s, I e, GapViewT<I> &v, R2T<P> &seq, bool &none) {
  none = false;
  if(s <= 0 || p.len < s || e <= 0 || p.len < e) {
    return PM_ST_PROFILE_IDX_OUT_OF_RANGE;
  }
  if(s > e) {
    I t = s;
    s = e;
    e = t;
  }
  int lo = 0, hi = p.n;
  while(lo < hi) { // first gap with end >= s
    int mid = (lo + hi) >> 1;
    if(p.g[mid].e < s) {
      lo = mid + 1;
    }
    else {
      hi = mid;
    }
  }
  int first = lo;
  hi = p.n;
  while(lo < hi) { // first gap with start > e
    int mid = (lo + hi) >> 1;
    if(p.g[mid].s <= e) {
      lo = mid + 1;
    }
    else {
      hi = mid;
    }
  }
  v.g = p.g;
  v.lo = first;
  v.n = lo - first;
  v.ws = s;
  v.we = e;
  v.mirror = false;
  v.L = 0;
  if(v.n > 0) {
    R2T<I> a = view_get(v, 0);
    R2T<I> b = view_get(v, v.n - 1);
    if(v.n == 1 && a.s == s && a.e == e) {
      none = true;
      return PM_ST_OK;
    }
    if(a.s == s) {
      s = a.e + 1;
    }
    if(b.e == e) {
      e = b.s - 1;
    }
  }
  bool n1, n2;
  P ss = 0, se = 0;
  int st = seq_idx_of_profile_idx(p, s, ss, n1);
  if(st) {
    return st;
  }
  st = seq_idx_of_profile_idx(p, e, se, n2);
  if(st) {
    return st;
  }
  if(n1 || n2) {
    return PM_ST_IS_NONE;
  }
  seq = R2T<P>{ss, se};
  return PM_ST_OK;
}

// m_profile.cc:208-212: subset_profile(profile_idx_of_seq_idx(s), profile_idx_of_seq_idx(e)).  The two conversions
// already hold everything subset_profile would search for, because their results are columns that carry a
// sequence position (never a gap column).  With ps = pre[a] + offset from the lower bound a (ascending disjoint
// gaps, pre[k+1] = pre[k] + length of gap k):
//   * gaps k < a have g[k].s <= offset + pre[k], so g[k].e = g[k].s + len_k - 1 <= offset + pre[a] - 1 < ps;
//     gap a has g[a].s - pre[a] > offset, so g[a].s > ps: "first gap with end >= ps" is a, "first gap with start
//     > ps" is a as well;
//   * hence the kept gaps of [min(ps,pe), max(ps,pe)] are the index range between the two lower bounds, none of
//     them touches an end of the window (no trimming at :182-193, never the empty option), and
//     seq_idx_of_profile_idx of either end finds gap a starting after it and returns range.s +- (offset - 1),
//     which is the sequence index the column came from.
// What remains of subset_profile are its range checks.
template <typename I, typename P>
PM_HD inline int subset_seq(const PVT<I, P> &p, P s, P e, GapViewT<I> &v, R2T<P> &seq) {
  I ps, pe;
  int as, ae;
  int st = profile_idx_of_seq_idx(p, s, ps, as);
  if(st) {
    return st;
  }
  st = profile_idx_of_seq_idx(p, e, pe, ae);
  if(st) {
    return st;
  }
  if(ps <= 0 || p.len < ps || pe <= 0 || p.len < pe) { // m_profile.cc:163-166
    return PM_ST_PROFILE_IDX_OUT_OF_RANGE;
  }
  const bool swap = ps > pe;
  v.g = p.g;
  v.lo = swap ? ae : as;
  v.n = (swap ? as : ae) - v.lo;
  v.ws = swap ? pe : ps;
  v.we = swap ? ps : pe;
  v.mirror = false;
  v.L = 0;
  seq = swap ? R2T<P>{e, s} : R2T<P>{s, e};
  return PM_ST_OK;
}

// Four binary searches in lock step.  A unit's set-up is some twenty searches of four to twelve probes each, every probe a
// memory round trip the next one waits for; the searches of one stage do not depend on each other, so they probe together and a
// stage costs the round trips of its deepest search (the count pass 128 -> 116 us on the bench job, profiles/r04_translate_ablation.txt).
// KIND 0: g[k].s - pre[k] <= key (profile_idx_of_seq_idx's lower bound); 1: g[k].e < key (the first gap that ends at or after
// key); 2: g[k].s <= key (the first gap that starts after key).  A probe with lo == hi is idle.
template <typename I>
struct LbProbe {
  const R2T<I> *g;
  const I *pre;
  int lo, hi;
  I key;
};
template <int KIND, typename I>
PM_HD __forceinline__ bool lb_pred(const R2T<I> &g, I pre, I key) {
  return KIND == 0 ? g.s - pre <= key : (KIND == 1 ? g.e < key : g.s <= key);
}
template <typename I, typename P>
PM_HD __forceinline__ LbProbe<I> lb_probe(const PVT<I, P> &p, bool on, I key) {
  return LbProbe<I>{p.g, p.pre, 0, on ? p.n : 0, key};
}
template <int K0, int K1, int K2, int K3, typename I>
PM_HD __forceinline__ void lower_bounds4(LbProbe<I> &a, LbProbe<I> &b, LbProbe<I> &c, LbProbe<I> &d) {
  while((a.lo < a.hi) | (b.lo < b.hi) | (c.lo < c.hi) | (d.lo < d.hi)) {
    const bool oa = a.lo < a.hi, ob = b.lo < b.hi, oc = c.lo < c.hi, od = d.lo < d.hi;
    const int ma = (a.lo + a.hi) >> 1, mb = (b.lo + b.hi) >> 1, mc = (c.lo + c.hi) >> 1, md = (d.lo + d.hi) >> 1;
    R2T<I> ga{0, 0}, gb{0, 0}, gc{0, 0}, gd{0, 0};
    I pa = 0, pb = 0, pc = 0, pd = 0;
    if(oa) {
      ga = a.g[ma];
      if(K0 == 0) pa = a.pre[ma];
    }
    if(ob) {
      gb = b.g[mb];
      if(K1 == 0) pb = b.pre[mb];
    }
    if(oc) {
      gc = c.g[mc];
      if(K2 == 0) pc = c.pre[mc];
    }
    if(od) {
      gd = d.g[md];
      if(K3 == 0) pd = d.pre[md];
    }
    if(oa) {
      if(lb_pred<K0>(ga, pa, a.key)) a.lo = ma + 1; else a.hi = ma;
    }
    if(ob) {
      if(lb_pred<K1>(gb, pb, b.key)) b.lo = mb + 1; else b.hi = mb;
    }
    if(oc) {
      if(lb_pred<K2>(gc, pc, c.key)) c.lo = mc + 1; else c.hi = mc;
    }
    if(od) {
      if(lb_pred<K3>(gd, pd, d.key)) d.lo = md + 1; else d.hi = md;
    }
  }
}
// profile_idx_of_seq_idx's range test and offset (m_profile.cc:93-99): false = PM_ST_SEQ_IDX_OUT_OF_RANGE
template <typename I, typename P>
PM_HD __forceinline__ bool seq_idx_offset(const PVT<I, P> &p, P si, I &offset) {
  const R2T<P> f = fwd_of(p.range);
  const P d = p.range.s - si;
  offset = (I)((d < 0 ? -d : d) + 1); // inside the row (the only case the value is used in): at most the row's length
  return f.s <= si && si <= f.e;
}

// a11, m_translate.cc:24-139: two gap lists, one push-back slot each.  Row 0 = reference, 1 = query.
// Kept as scalars (no runtime-indexed arrays: those would go to scratch).
// The element a list stands at is held in registers (cur) and read again only when the list moves on: the merge decides every
// step from the four fronts, and read from memory at the start of every step they were four or five scattered loads a step -- a
// lane's addresses share no line with its neighbours', so what a unit pass costs is the number of such loads, not their latency
// (profiles/r04_translate_ablation.txt: also holding the element after it, to take the read off the step's path, bought nothing).
template <typename I>
struct PairCursorT {
  GapViewT<I> v0, v1;
  int at0, at1;
  bool held0, held1;
  R2T<I> hold0, hold1;
  R2T<I> cur0, cur1; // view_get(v, at); anything past the list's end

  PM_HD __forceinline__ void prime() { // after the views and at0/at1 are set
    const R2T<I> z{0, 0};
    cur0 = at0 < v0.n ? view_get(v0, at0) : z;
    cur1 = at1 < v1.n ? view_get(v1, at1) : z;
  }
  PM_HD __forceinline__ bool has(int r) const { return r ? (held1 || at1 < v1.n) : (held0 || at0 < v0.n); }
  PM_HD __forceinline__ R2T<I> front(int r) const {
    if(r) {
      return held1 ? hold1 : cur1;
    }
    return held0 ? hold0 : cur0;
  }
  PM_HD __forceinline__ bool done() const { return !has(0) && !has(1); }
  PM_HD __forceinline__ void pop(int r) {
    if(r) {
      if(held1) {
        held1 = false;
      }
      else {
        ++at1;
        if(at1 < v1.n) {
          cur1 = view_get(v1, at1);
        }
      }
    }
    else {
      if(held0) {
        held0 = false;
      }
      else {
        ++at0;
        if(at0 < v0.n) {
          cur0 = view_get(v0, at0);
        }
      }
    }
  }
  PM_HD __forceinline__ int push_back(int r, R2T<I> g) {
    if(r ? held1 : held0) {
      return PM_ST_ALREADY_UNNEXT;
    }
    if(r) {
      hold1 = g;
      held1 = true;
    }
    else {
      hold0 = g;
      held0 = true;
    }
    return PM_ST_OK;
  }
  // m_translate.cc:34-62
  PM_HD __forceinline__ int pick(I pos0, I pos1, bool &have, int &row, R2T<I> &gap) const {
    bool h0 = has(0), h1 = has(1);
    have = h0 || h1;
    if(h0 && h1) {
      R2T<I> g0 = front(0), g1 = front(1);
      I d0 = g0.s - pos0, d1 = g1.s - pos1;
      if(d0 < 0 || d1 < 0) {
        return PM_ST_ASSERT_GAP_BEHIND;
      }
      row = d0 <= d1 ? 0 : 1;
      gap = row ? g1 : g0;
    }
    else if(h0) {
      row = 0;
      gap = front(0);
    }
    else if(h1) {
      row = 1;
      gap = front(1);
    }
    return PM_ST_OK;
  }
};

// Output side of one unit.  COUNT pass: sizes only.  EMIT pass: writes into the unit's exact slot.
// The builder's two gap lists are normally not stored: each gap is turned into its signed offsets the moment it is
// added, which equals deltas_of_gaps (m_delta_stream_writer.hh:14-53) as long as gaps arrive in that merge's
// order.  The COUNT pass checks the order per gap (the counts do not depend on it) and marks a unit whose gaps ever
// arrive otherwise (`disorder`: only tables that contradict themselves, or a delta entry with a column gapped in both
// rows, get there).  Such units are emitted a second time by the FIX pass (`fix` != nullptr): the open segment's gaps
// are recorded in arrival order in a scratch list and commit() writes the writer's own two-list merge of them -- heads
// compared by start, ties to the query list -- over what the EMIT pass wrote.
// What the EMIT pass writes per entry.  Jobs in int64 coordinates write the C ABI's pm_entry_t (48 bytes) and int64 offsets; jobs
// whose tables fit the int instantiation -- all but pathological ones -- write a 32-byte record and int offsets: a record is then
// one 32-byte sector of HBM, written whole by its lane, and the job's output is 0.63 of the wide one's bytes.  (Round 2's PMC
// pass found 239 MB written for 126 MB of output: a 48-byte record at a 48-byte stride touches two sectors, a lone 8-byte offset
// costs a sector of its own; the lines of ~400 k resident lanes do not wait in the L2 to be completed.)  pm_job_fetch widens.
struct alignas(32) Entry32 {
  int ref_start, ref_end, qry_start, qry_end;
  i64 offset_begin;
  int n_offsets;
  int pad;
};
static_assert(sizeof(Entry32) == 32, "one HBM sector per record");
template <typename I>
struct EntRecT {
  typedef pm_entry_t type;
};
template <>
struct EntRecT<int> {
  typedef Entry32 type;
};

template <bool EMIT, typename I = i64>
struct Sink {
  typedef typename EntRecT<I>::type Rec;
  static constexpr int S = 32 / (int)sizeof(I); // offsets per 32-byte sector
  I n_ent;     // committed entries
  I n_off;     // committed offsets
  I pend;      // offsets of the open segment
  I wpos;      // deltas_of_gaps' running column
  I last_start;
  int last_row;
  Rec *ent;      // EMIT: this unit's first entry slot
  I *off;        // EMIT: whole offsets array
  i64 off_base;  // EMIT: this unit's first offset index
  I off_cap;     // EMIT: this unit's offset count (exact, from the COUNT pass)
  I ent_cap;
  int disorder;  // COUNT: a gap arrived out of the writer's merge order
  i64 *fix;      // FIX pass: the open segment's gaps in arrival order, two words each {start << 1 | row, end}; else nullptr
  I fix_n;
  I fix_cap;     // FIX pass: gaps the scratch list holds (the unit's offset count + 1).  A gap owns at least one offset of
                 // its segment, so a segment that commits never holds more; one that is later dropped may, and the
                 // gaps past the cap are then simply not recorded (nothing ever reads them)
  // EMIT: where a unit's offsets go.  The 64 lanes of a wavefront hold consecutive live units, so their offsets are ONE contiguous run
  // of the job's offsets array, [win_lo, win_lo + what the wavefront's units hold): the lanes write into a window of the wavefront's
  // own in LDS (`win`, win_cap slots from global index win_lo on) and the wavefront stores the run, whole lines at a time, when all its
  // lanes are done (the kernel does: translate_kernel).  An offset beyond the window (a wavefront with more offsets than win_cap) goes
  // to memory directly.  Until round 5 every lane staged two sectors of its own and wrote whole sectors -- but a unit holds six offsets
  // on average, a quarter of a sector, and the first and last sector of every unit left the L2 as partial writes twice (the
  // request counters: 109 MB written for 80 MB of output).  Every slot of a unit's exact range is written with its final value before
  // the unit ends (the range is the COUNT pass's count of the same run), a dropped segment's slots by the segment that replaces it.
  I *win;
  i64 win_lo;
  int win_cap;

  PM_HD __forceinline__ void stash(I at, I v) {
    if(at < off_cap) { // offsets of a segment that is later dropped may run past the exact slot
      const i64 g = off_base + at;
      const unsigned long long d = (unsigned long long)(g - win_lo);
      if(d < (unsigned long long)win_cap) {
        win[d] = v;
      }
      else {
        off[g] = v;
      }
    }
  }
  PM_HD __forceinline__ void finish() {} // (the window is stored by the wavefront, after its last lane's last commit)

  PM_HD __forceinline__ void put(I v) {
    if(EMIT && !fix) {
      stash(n_off + pend, v);
    }
    ++pend;
  }
  PM_HD __forceinline__ int gap(int row, R2T<I> g) {
    if(!EMIT && pend > 0 && (g.s < last_start || (g.s == last_start && last_row == 0 && row == 1))) {
      disorder = 1;
    }
    if(EMIT && fix && fix_n < fix_cap) {
      fix[2 * (i64)fix_n] = (i64)g.s * 2 + row;
      fix[2 * (i64)fix_n + 1] = (i64)g.e;
      ++fix_n;
    }
    I sign = row ? 1 : -1;
    put(sign * (g.s - wpos));
    // the gap's remaining columns are +-1 each (_push_ones, m_delta_stream_writer.hh:6-11).  Counted
    // arithmetically and written only inside the unit's slot, so a nonsensical gap length cannot stall a lane.
    I ones = rlen(g) - 1;
    if(EMIT && !fix) {
      I at = n_off + pend;
      I room = off_cap - at;
      I n = ones < room ? ones : room;
      for(I k = 0; k < n; ++k) {
        stash(at + k, sign);
      }
    }
    pend += ones;
    wpos = g.e;
    last_start = g.s;
    last_row = row;
    return PM_ST_OK;
  }
  PM_HD __forceinline__ void drop() {
    pend = 0;
    wpos = 0;
    fix_n = 0;
  }
  // FIX pass: deltas_of_gaps over the recorded gaps (m_delta_stream_writer.hh:14-53): two cursors, one per row, each taking
  // its row's gaps in arrival order; the smaller start goes first, ties to the query row
  PM_HD __forceinline__ void fix_write(I &at, I v) {
    if(at < off_cap) {
      off[off_base + at] = v;
    }
    ++at;
  }
  PM_HD __forceinline__ I fix_next(I from, int row) const {
    while(from < fix_n && (int)(fix[2 * (i64)from] & 1) != row) {
      ++from;
    }
    return from;
  }
  PM_HD __forceinline__ void fix_merge() {
    I r = fix_next(0, 0), q = fix_next(0, 1), at = n_off;
    i64 column = 0;
    while(r < fix_n || q < fix_n) {
      const bool take_ref = r < fix_n && (q >= fix_n || (fix[2 * (i64)r] >> 1) < (fix[2 * (i64)q] >> 1));
      const I k = take_ref ? r : q;
      const i64 gs = fix[2 * (i64)k] >> 1, ge = fix[2 * (i64)k + 1];
      const I sign = take_ref ? -1 : 1;
      fix_write(at, (I)(sign * (gs - column)));
      // the gap's remaining columns: length - 1 = |e - s|, as the COUNT and EMIT passes count them (M_range::length)
      for(i64 n = gs <= ge ? ge - gs : gs - ge; n > 0; --n) {
        fix_write(at, sign);
      }
      column = ge;
      if(take_ref) {
        r = fix_next(r + 1, 0);
      }
      else {
        q = fix_next(q + 1, 1);
      }
    }
    fix_write(at, 0);
  }
  PM_HD __forceinline__ void commit(R2T<I> ref, R2T<I> qry) {
    if(EMIT && fix) {
      fix_merge();
    }
    put(0);
    if(EMIT) {
      if(n_ent < ent_cap) {
        Rec e = Rec();
        e.ref_start = ref.s;
        e.ref_end = ref.e;
        e.qry_start = qry.s;
        e.qry_end = qry.e;
        e.offset_begin = off_base + n_off;
        e.n_offsets = pend;
        ent[n_ent] = e;
      }
    }
    ++n_ent;
    n_off += pend;
    drop();
  }
};

// a9 + a12: builder and merge state of one unit, all in registers.
template <bool EMIT, typename I = i64>
struct Merge {
  PairCursorT<I> rows;  // gaps of the two row profiles
  PairCursorT<I> delta; // gaps of the entry's own rows
  I ref_pos, query_pos, column, last_column;
  // builder (m_delta_builder.hh:9-87)
  I b_ref_start, b_ref_pos, b_query_start, b_query_pos, b_sum0, b_sum1;
  bool mirrored;
  I query_columns;
  Sink<EMIT, I> sink;

  PM_HD __forceinline__ void b_restart(I r, I q) {
    b_ref_start = b_ref_pos = r;
    b_query_start = b_query_pos = q;
    b_sum0 = b_sum1 = 0;
    sink.drop();
  }
  PM_HD __forceinline__ int b_add_gap(int row, R2T<I> d) { // m_delta_builder.hh:32-63
    I walked = row ? (b_query_pos - b_query_start) + b_sum1 : (b_ref_pos - b_ref_start) + b_sum0;
    R2T<I> g{d.s + walked + 1, d.e + walked + 1};
    if(row) {
      b_sum1 += rlen(g);
      b_ref_pos += d.e + 1;
      b_query_pos += d.s;
    }
    else {
      b_sum0 += rlen(g);
      b_ref_pos += d.s;
      b_query_pos += d.e + 1;
    }
    return sink.gap(row, g);
  }
  PM_HD __forceinline__ I qcol(I pi) const { return mirrored ? query_columns - pi + 1 : pi; }
  PM_HD __forceinline__ void b_finish() { // m_delta_builder.cc:7-22
    if(b_ref_start != b_ref_pos && b_query_start != b_query_pos) {
      sink.commit(R2T<I>{b_ref_start, b_ref_pos - 1}, R2T<I>{qcol(b_query_start), qcol(b_query_pos - 1)});
    }
  }
  PM_HD __forceinline__ void consume_delta_piece(int row, R2T<I> d) { // m_translate.cc:220-231
    if(row) {
      ref_pos += d.e + 1;
      query_pos += d.s;
    }
    else {
      ref_pos += d.s;
      query_pos += d.e + 1;
    }
    column += d.e + 1;
  }
  PM_HD __forceinline__ void close_segment(int row, R2T<I> g) { // m_translate.cc:309-316,436-443
    b_ref_pos += g.s;
    b_query_pos += g.s;
    if(row) { // m_translate.cc:233-244
      ref_pos += g.s;
      query_pos += g.e + 1;
    }
    else {
      ref_pos += g.e + 1;
      query_pos += g.s;
    }
    column += g.s;
    rows.pop(row);
    b_finish();
    b_restart(ref_pos, query_pos);
  }
  PM_HD __forceinline__ R2T<I> rel(int row, R2T<I> g) const {
    I base = row ? query_pos : ref_pos;
    return R2T<I>{g.s - base, g.e - base};
  }
  // I = int only: has any value the merge carries from step to step left +-2^26?
  // Why that proves the int merge computed what the int64 merge computes: every table value is below 2^25 in
  // magnitude (checked when the job is prepared; derived table values -- an entry's column count, its mirrored gaps
  // -- stay below 2^27).  With A = 2^26 and all carried values within +-A at the start of a step, the step's
  // intermediates are bounded by: a gap relative to a position 3A; the piece of a split 9A; a builder gap 12A+1; its
  // length, the largest of all, 24A+3; an accumulator after its update 25A+3 < 2^31.  Nothing wraps inside a step
  // that starts in range, and this check re-establishes the range after every step.
  PM_HD __forceinline__ bool narrow_overflow() const {
    const unsigned bias = 1u << 26;
    unsigned bad = 0;
#define PM_NARROW_CHECK(x) bad |= (((unsigned)(x) + bias) >> 27)
    PM_NARROW_CHECK(ref_pos);
    PM_NARROW_CHECK(query_pos);
    PM_NARROW_CHECK(column);
    PM_NARROW_CHECK(b_ref_pos);
    PM_NARROW_CHECK(b_query_pos);
    PM_NARROW_CHECK(b_ref_start);
    PM_NARROW_CHECK(b_query_start);
    PM_NARROW_CHECK(b_sum0);
    PM_NARROW_CHECK(b_sum1);
    PM_NARROW_CHECK(sink.pend);
    PM_NARROW_CHECK(sink.n_off);
    PM_NARROW_CHECK(sink.wpos);
    PM_NARROW_CHECK(sink.last_start);
    PM_NARROW_CHECK(rows.hold0.s);
    PM_NARROW_CHECK(rows.hold0.e);
    PM_NARROW_CHECK(rows.hold1.s);
    PM_NARROW_CHECK(rows.hold1.e);
    PM_NARROW_CHECK(delta.hold0.s);
    PM_NARROW_CHECK(delta.hold0.e);
    PM_NARROW_CHECK(delta.hold1.s);
    PM_NARROW_CHECK(delta.hold1.e);
#undef PM_NARROW_CHECK
    return bad != 0;
  }
  PM_HD inline int step() { // m_translate.cc:279-472
    bool have_p, have_d;
    int prow = 0, drow = 0;
    R2T<I> pgap{0, 0}, dgap{0, 0};
    int st = rows.pick(ref_pos, query_pos, have_p, prow, pgap);
    if(st) {
      return st;
    }
    st = delta.pick(column, column, have_d, drow, dgap);
    if(st) {
      return st;
    }
    // What the step does is decided first and done once: the reference's five cases end in one of two bodies (close the segment at a
    // row gap; add a piece of the entry's gap), and with a copy of the body per case the lanes of a wavefront, each in its own case,
    // ran every copy one after the other -- the merge is bound by the instructions its divergent lanes issue between them, not by its
    // loads' latency (profiles/r05_translate_merge.txt: the emit pass 118 -> 101 us, the int64 one 163 -> 137).
    enum { NONE, CLOSE, ADD, LAST } act = NONE;
    int arow = 0;
    R2T<I> ag{0, 0}, rest{0, 0};
    bool push_rest = false;
    if(have_p && have_d) {
      R2T<I> g = rel(prow, pgap);
      R2T<I> d{dgap.s - column, dgap.e - column};
      int other = prow ^ 1;
      bool other_within = rows.has(other) && rel(other, rows.front(other)).s <= d.e; // :246-268
      if(g.s <= d.s) {
        act = CLOSE;
        arow = prow;
        ag = g;
      }
      else if(d.e < g.s || (prow == drow && !other_within)) {
        act = ADD;
        arow = drow;
        ag = d;
      }
      else {
        // split the entry's gap in front of the row gap it runs into (:368-386 same row, :395-401 other row)
        I keep = prow == drow ? rel(other, rows.front(other)).s - d.s : g.s - d.s;
        act = ADD;
        arow = drow;
        ag = R2T<I>{d.s, d.s + keep - 1};
        rest = R2T<I>{dgap.s + keep, dgap.e};
        push_rest = true;
      }
    }
    else if(have_p) {
      act = CLOSE;
      arow = prow;
      ag = rel(prow, pgap);
    }
    else if(have_d) {
      act = ADD;
      arow = drow;
      ag = R2T<I>{dgap.s - column, dgap.e - column};
    }
    else if(column <= last_column) { // :464-470
      act = LAST;
    }
    if(act == CLOSE) {
      close_segment(arow, ag);
    }
    else if(act == ADD) {
      st = b_add_gap(arow, ag);
      consume_delta_piece(arow, ag);
      delta.pop(arow);
      if(push_rest) {
        int st2 = delta.push_back(arow, rest);
        if(!st) {
          st = st2;
        }
      }
    }
    else if(act == LAST) {
      I n = last_column - column + 1;
      b_ref_pos += n;
      b_query_pos += n;
      b_finish();
    }
    return st;
  }
};

template <typename I, typename P>
PM_HD __forceinline__ PVT<I, P> row_view(const RowsT<I, P> &rows, int r) {
  PVT<I, P> p;
  p.range = rows.range[r];
  p.len = rows.length[r];
  i64 o = rows.gap_off[r];
  p.n = (int)(rows.gap_off[r + 1] - o);
  p.g = rows.gaps + o;
  p.pre = rows.pre + o + r;
  return p;
}

// First part of a unit, up to the test that ends most of them (m_translate.cc:636-639 and :496-513): the
// overlap of the entry with both rows, the entry's two rows as profiles over its own columns, and the window of
// columns both rows cover.  `live` = the unit goes on to the subset/merge part.
template <typename I, typename P>
PM_HD inline int unit_prefix(const RowsT<I, P> &left, const RowsT<I, P> &right, const DeltasT<I, P> &ds, int d, int l, int r, PVT<I, P> &lp,
                             PVT<I, P> &rp, PVT<I, P> &dr, PVT<I, P> &dq, R2T<I> &cols, bool &live, int &orientation) {
  live = false;
  orientation = 0;
  if(left.bad[l] | right.bad[r] | ds.bad[d]) {
    return PM_ST_MALFORMED_INPUT;
  }
  lp = row_view(left, l);
  rp = row_view(right, r);
  R2T<P> de_ref = ds.ref[d], de_qry = ds.qry[d];
  R2T<P> ref_seq, query_seq;
  if(!overlap(de_ref, lp.range, ref_seq) || !overlap(de_qry, rp.range, query_seq)) {
    return PM_ST_OK; // :636-639
  }
  int o = fwd(de_ref) != fwd(lp.range) ? 1 : 0; // :210-217
  orientation = o;
  if(o) {
    de_ref = R2T<P>{de_ref.e, de_ref.s};
    de_qry = R2T<P>{de_qry.e, de_qry.s};
  }
  // the entry's two rows as profiles over its own columns (:496-506)
  i64 ro = ds.ref_off[d], qo = ds.qry_off[d];
  dr.range = de_ref;
  dr.n = (int)(ds.ref_off[d + 1] - ro);
  dr.g = ds.ref_gaps[o] + ro;
  dr.pre = ds.ref_pre[o] + ro + d;
  dr.len = (I)rlen(de_ref) + dr.pre[dr.n];
  dq.range = de_qry;
  dq.n = (int)(ds.qry_off[d + 1] - qo);
  dq.g = ds.qry_gaps[o] + qo;
  dq.pre = ds.qry_pre[o] + qo + d;
  dq.len = (I)rlen(de_qry) + dq.pre[dq.n];

  // :508-511: four profile_idx_of_seq_idx; their range tests in the reference's order, their searches together
  I o0, o1, o2, o3;
  if(!seq_idx_offset(dr, ref_seq.s, o0) || !seq_idx_offset(dr, ref_seq.e, o1) || !seq_idx_offset(dq, query_seq.s, o2) ||
     !seq_idx_offset(dq, query_seq.e, o3)) {
    return PM_ST_SEQ_IDX_OUT_OF_RANGE;
  }
  LbProbe<I> a = lb_probe(dr, true, o0), b = lb_probe(dr, true, o1), c = lb_probe(dq, true, o2), e = lb_probe(dq, true, o3);
  lower_bounds4<0, 0, 0, 0>(a, b, c, e);
  const I p0 = dr.pre[a.lo], p1 = dr.pre[b.lo], p2 = dq.pre[c.lo], p3 = dq.pre[e.lo];
  const R2T<I> d_ref_cols{p0 + o0, p1 + o1}, d_query_cols{p2 + o2, p3 + o3};
  live = overlap(d_ref_cols, d_query_cols, cols); // :513
  return PM_ST_OK;
}

// What the merge of a unit starts from, as plain numbers: enough to rebuild the Merge without redoing the ~20 binary
// searches of the set-up.  The count pass saves it per live unit, the emit pass restores it.
template <typename I>
struct UnitStateT {
  int lo[4], n[4];   // kept gaps of: left row, right row, entry's reference row, entry's query row
  I ws[4], we[4];  // their clip windows
  I ref_start, query_start, column, last_column, query_columns;
  int orientation;   // 1: the entry is used reversed (m_translate.cc:210-217)
  int mirrored;      // the right row is walked backwards (:557)
  int delta, left, right; // which entry and rows the unit is (set by the count pass: the emit pass reads them with the state, as runs)
};
typedef UnitStateT<i64> UnitState;

// Set-up of a unit that passed unit_prefix (m_translate.cc:527-610): `proceed` = the merge has to run.
template <bool EMIT, typename I, typename P>
PM_HD inline int unit_setup(const PVT<I, P> &lp, const PVT<I, P> &rp, const PVT<I, P> &dr, const PVT<I, P> &dq, R2T<I> cols, Merge<EMIT, I> &m,
                            bool &proceed) {
  proceed = false;
  // ---- :527-533: subset_profile (m_profile.cc:160-206) of the entry's two rows, side 0 = reference row, 1 = query row, stage by
  // stage for both at once.  Every stage is pure, so a side's status is found whatever the other side's is; the reference's (the
  // first side's first failure) is picked at the end.
  const PVT<I, P> *pv[2] = {&dr, &dq};
  GapViewT<I> *vv[2] = {&m.delta.v0, &m.delta.v1};
  int st_side[2] = {PM_ST_OK, PM_ST_OK};
  bool none_side[2] = {false, false}, on[2];
  I ws[2], we[2];
#pragma unroll
  for(int k = 0; k < 2; ++k) {
    I s = cols.s, e = cols.e;
    on[k] = !(s <= 0 || pv[k]->len < s || e <= 0 || pv[k]->len < e);
    if(!on[k]) {
      st_side[k] = PM_ST_PROFILE_IDX_OUT_OF_RANGE;
    }
    ws[k] = s <= e ? s : e;
    we[k] = s <= e ? e : s;
  }
  {
    // the kept gaps: from the first that ends at or after s to the first that starts after e (searched from the list's start:
    // every gap before the first kept one starts before s <= e)
    LbProbe<I> a = lb_probe(dr, on[0], ws[0]), b = lb_probe(dr, on[0], we[0]), c = lb_probe(dq, on[1], ws[1]), d = lb_probe(dq, on[1], we[1]);
    lower_bounds4<1, 2, 1, 2>(a, b, c, d);
    const int first[2] = {a.lo, c.lo}, last[2] = {b.lo, d.lo};
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      vv[k]->g = pv[k]->g;
      vv[k]->lo = first[k];
      vv[k]->n = last[k] - first[k];
      vv[k]->ws = ws[k];
      vv[k]->we = we[k];
      vv[k]->mirror = false;
      vv[k]->L = 0;
    }
  }
  // a kept gap that reaches an end of the window is cut off it (:182-193); the one gap that is the window: the empty option
  I cs[2] = {ws[0], ws[1]}, ce[2] = {we[0], we[1]};
  {
    R2T<I> ga[2] = {{0, 0}, {0, 0}}, gb[2] = {{0, 0}, {0, 0}};
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      if(on[k] && vv[k]->n > 0) {
        ga[k] = view_get(*vv[k], 0);
        gb[k] = view_get(*vv[k], vv[k]->n - 1);
      }
    }
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      if(on[k] && vv[k]->n > 0) {
        if(vv[k]->n == 1 && ga[k].s == ws[k] && ga[k].e == we[k]) {
          none_side[k] = true;
          on[k] = false;
        }
        else {
          if(ga[k].s == ws[k]) {
            cs[k] = ga[k].e + 1;
          }
          if(gb[k].e == we[k]) {
            ce[k] = gb[k].s - 1;
          }
        }
      }
    }
  }
  // the sequence positions of the window's ends: seq_idx_of_profile_idx (m_profile.cc:114-149) of both, both sides
  R2T<P> seq_side[2] = {{0, 0}, {0, 0}};
  {
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      if(on[k] && (!(cs[k] < pv[k]->len + 1) || !(ce[k] < pv[k]->len + 1))) {
        st_side[k] = PM_ST_PROFILE_IDX_OUT_OF_RANGE;
        on[k] = false;
      }
    }
    LbProbe<I> a = lb_probe(dr, on[0], cs[0]), b = lb_probe(dr, on[0], ce[0]), c = lb_probe(dq, on[1], cs[1]), d = lb_probe(dq, on[1], ce[1]);
    lower_bounds4<1, 1, 1, 1>(a, b, c, d);
    const int at[2][2] = {{a.lo, b.lo}, {c.lo, d.lo}};
    I gs[2][2], pr[2][2];
#pragma unroll
    for(int k = 0; k < 2; ++k) {
#pragma unroll
      for(int w = 0; w < 2; ++w) {
        gs[k][w] = 0;
        pr[k][w] = 0;
        if(on[k]) {
          pr[k][w] = pv[k]->pre[at[k][w]];
          if(at[k][w] < pv[k]->n) {
            gs[k][w] = pv[k]->g[at[k][w]].s;
          }
        }
      }
    }
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      if(on[k]) {
        const I pi[2] = {cs[k], ce[k]};
        P out[2];
        bool none = false;
#pragma unroll
        for(int w = 0; w < 2; ++w) {
          none = none || (at[k][w] < pv[k]->n && gs[k][w] <= pi[w]);
          const I offset = pi[w] - pr[k][w] - 1;
          out[w] = fwd(pv[k]->range) ? pv[k]->range.s + (P)offset : pv[k]->range.s - (P)offset;
        }
        if(none) {
          st_side[k] = PM_ST_IS_NONE;
        }
        seq_side[k] = R2T<P>{out[0], out[1]};
      }
    }
  }
  if(st_side[0]) return st_side[0]; // :527-529
  if(st_side[1]) return st_side[1]; // :531-533
  if(none_side[0] || none_side[1]) {
    return PM_ST_OK; // :535
  }
  const R2T<P> d_ref_seq = seq_side[0], d_query_seq = seq_side[1];
  // ---- :539-545: subset_seq of the two row profiles (see subset_seq above for why two conversions each are all of it)
  R2T<P> l_seq, r_seq;
  {
    const PVT<I, P> *rv[2] = {&lp, &rp};
    GapViewT<I> *rw[2] = {&m.rows.v0, &m.rows.v1};
    const P si[2][2] = {{d_ref_seq.s, d_ref_seq.e}, {d_query_seq.s, d_query_seq.e}};
    I off[2][2];
    bool ok[2];
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      const bool in0 = seq_idx_offset(*rv[k], si[k][0], off[k][0]), in1 = seq_idx_offset(*rv[k], si[k][1], off[k][1]);
      ok[k] = in0 && in1;
    }
    LbProbe<I> a = lb_probe(lp, ok[0], off[0][0]), b = lb_probe(lp, ok[0], off[0][1]), c = lb_probe(rp, ok[1], off[1][0]), d = lb_probe(rp, ok[1], off[1][1]);
    lower_bounds4<0, 0, 0, 0>(a, b, c, d);
    const int at[2][2] = {{a.lo, b.lo}, {c.lo, d.lo}};
    I pr[2][2];
#pragma unroll
    for(int k = 0; k < 2; ++k) {
#pragma unroll
      for(int w = 0; w < 2; ++w) {
        pr[k][w] = ok[k] ? rv[k]->pre[at[k][w]] : 0;
      }
    }
    int st_row[2];
    bool swp[2] = {false, false};
#pragma unroll
    for(int k = 0; k < 2; ++k) {
      const I ps = pr[k][0] + off[k][0], pe = pr[k][1] + off[k][1];
      if(!ok[k]) {
        st_row[k] = PM_ST_SEQ_IDX_OUT_OF_RANGE;
      }
      else if(ps <= 0 || rv[k]->len < ps || pe <= 0 || rv[k]->len < pe) { // m_profile.cc:163-166
        st_row[k] = PM_ST_PROFILE_IDX_OUT_OF_RANGE;
      }
      else {
        st_row[k] = PM_ST_OK;
        const bool swap = ps > pe;
        swp[k] = swap;
        rw[k]->g = rv[k]->g;
        rw[k]->lo = swap ? at[k][1] : at[k][0];
        rw[k]->n = (swap ? at[k][0] : at[k][1]) - rw[k]->lo;
        rw[k]->ws = swap ? pe : ps;
        rw[k]->we = swap ? ps : pe;
        rw[k]->mirror = false;
        rw[k]->L = 0;
      }
    }
    if(st_row[0]) return st_row[0]; // :539-541
    if(st_row[1]) return st_row[1]; // :543-545
    l_seq = swp[0] ? R2T<P>{d_ref_seq.e, d_ref_seq.s} : d_ref_seq;
    r_seq = swp[1] ? R2T<P>{d_query_seq.e, d_query_seq.s} : d_query_seq;
  }
  if(rlen(d_ref_seq) != rlen(l_seq) || rlen(d_query_seq) != rlen(r_seq)) {
    return PM_ST_ASSERT_SUB_LENGTHS; // :550-551
  }
  bool mirrored = fwd(rp.range) != fwd(dq.range); // :557
  m.rows.v1.mirror = mirrored;
  m.rows.v1.L = rp.len;
  // :572-581 convert l_seq.s and r_seq.s (r_seq.e when mirrored) back to columns: those are the ends of the windows
  // subset_seq has just derived them from (see there), so no search is repeated
  const I ref_start = m.rows.v0.ws;
  const I query_start = mirrored ? rp.len - m.rows.v1.we + 1 : m.rows.v1.ws;
  m.rows.at0 = m.rows.at1 = m.delta.at0 = m.delta.at1 = 0;
  m.rows.held0 = m.rows.held1 = m.delta.held0 = m.delta.held1 = false;
  m.rows.hold0 = m.rows.hold1 = m.delta.hold0 = m.delta.hold1 = R2T<I>{0, 0};
  m.ref_pos = ref_start;
  m.query_pos = query_start;
  m.column = cols.s;
  m.last_column = cols.e;
  m.mirrored = mirrored;
  m.query_columns = rp.len;
  m.b_restart(ref_start, query_start);
  m.rows.prime();
  m.delta.prime();
  proceed = true;
  return PM_ST_OK;
}

// The merge loop itself (m_translate.cc:612-618).  Same step budget as oracle/pm_oracle.cc: far above any
// terminating run; every lane reaches it.
template <bool EMIT, typename I>
PM_HD inline int unit_merge(Merge<EMIT, I> &m) {
  i64 budget = 4 * (i64)(m.rows.v0.n + m.rows.v1.n + m.delta.v0.n + m.delta.v1.n) + 64;
  int st = PM_ST_OK;
  while(!m.rows.done() || !m.delta.done()) {
    if(budget-- <= 0) {
      st = PM_ST_STEP_LIMIT;
      break;
    }
    st = m.step();
    if(st) {
      break;
    }
    if(sizeof(I) < 8 && m.narrow_overflow()) {
      return PM_ST_NARROW;
    }
  }
  if(!st) {
    st = m.step();
    if(sizeof(I) < 8 && m.narrow_overflow()) {
      return PM_ST_NARROW;
    }
  }
  return st;
}

template <bool EMIT, typename I>
PM_HD inline void unit_save(const Merge<EMIT, I> &m, int orientation, UnitStateT<I> &s) {
  const GapViewT<I> *v[4] = {&m.rows.v0, &m.rows.v1, &m.delta.v0, &m.delta.v1};
  for(int k = 0; k < 4; ++k) {
    s.lo[k] = v[k]->lo;
    s.n[k] = v[k]->n;
    s.ws[k] = v[k]->ws;
    s.we[k] = v[k]->we;
  }
  s.ref_start = m.ref_pos;
  s.query_start = m.query_pos;
  s.column = m.column;
  s.last_column = m.last_column;
  s.query_columns = m.query_columns;
  s.orientation = orientation;
  s.mirrored = m.mirrored ? 1 : 0;
}

template <bool EMIT, typename I, typename P>
PM_HD inline void unit_restore(const RowsT<I, P> &left, const RowsT<I, P> &right, const DeltasT<I, P> &ds, int d, int l, int r,
                               const UnitStateT<I> &s, Merge<EMIT, I> &m) {
  const R2T<I> *g[4] = {left.gaps + left.gap_off[l], right.gaps + right.gap_off[r], ds.ref_gaps[s.orientation] + ds.ref_off[d],
                    ds.qry_gaps[s.orientation] + ds.qry_off[d]};
  GapViewT<I> *v[4] = {&m.rows.v0, &m.rows.v1, &m.delta.v0, &m.delta.v1};
  for(int k = 0; k < 4; ++k) {
    v[k]->g = g[k];
    v[k]->lo = s.lo[k];
    v[k]->n = s.n[k];
    v[k]->ws = s.ws[k];
    v[k]->we = s.we[k];
    v[k]->mirror = false;
    v[k]->L = 0;
  }
  m.rows.v1.mirror = s.mirrored != 0;
  m.rows.v1.L = s.query_columns;
  m.rows.at0 = m.rows.at1 = m.delta.at0 = m.delta.at1 = 0;
  m.rows.held0 = m.rows.held1 = m.delta.held0 = m.delta.held1 = false;
  m.rows.hold0 = m.rows.hold1 = m.delta.hold0 = m.delta.hold1 = R2T<I>{0, 0};
  m.ref_pos = s.ref_start;
  m.query_pos = s.query_start;
  m.column = s.column;
  m.last_column = s.last_column;
  m.mirrored = s.mirrored != 0;
  m.query_columns = s.query_columns;
  m.b_restart(s.ref_start, s.query_start);
  m.rows.prime();
  m.delta.prime();
}

// One whole unit: _translate_delta_with_profiles (m_translate.cc:625-647) + _generate_delta (:474-621).
template <bool EMIT, typename I, typename P>
PM_HD inline int run_unit(const RowsT<I, P> &left, const RowsT<I, P> &right, const DeltasT<I, P> &ds, int d, int l, int r,
                               Sink<EMIT, I> &sink) {
  PVT<I, P> lp, rp, dr, dq;
  R2T<I> cols;
  bool live;
  int orientation;
  int st = unit_prefix(left, right, ds, d, l, r, lp, rp, dr, dq, cols, live, orientation);
  if(st || !live) {
    return st;
  }
  Merge<EMIT, I> m;
  m.sink = sink;
  bool proceed;
  st = unit_setup<EMIT>(lp, rp, dr, dq, cols, m, proceed);
  if(st || !proceed) {
    return st;
  }
  st = unit_merge<EMIT>(m);
  m.sink.finish();
  sink = m.sink;
  return st;
}

} // namespace pm
