// side_tools.hip -- the two tools of lib/profiles_cpp that compile upstream, on the GPU:
//   m_sort_delta  (lib/profiles_cpp/m_sort_delta.cc:58-91)            -> pm_sort_delta
//   maf_analyzer  (lib/profiles_cpp/maf_analyzer.cc:12-38,
//                  maf_analyzer_missing.cc:38-160, maf_read_stream.cc) -> pm_maf_analyzer
// Nothing in the reference calls them; they are built because BASELINE.json names their directory
// and its configs[0] is `maf_analyzer tests/highly_stitchable.maf`.  Text in/out stays on the host; the sort, the
// interval union and the complement run in kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <map>
#include <string>
#include <vector>

#include "pm_internal.hpp"
#include "translate_host.hpp"

namespace pm {

typedef long long i64;

// ------------------------------------------------------------------ a15: sort keys

// (header pair rank, ref start, query start, ref end, query end): m_sort_delta.cc:37-55, applied as one
// lexicographic key; the reference's two-level std::sort gives the same order up to ties.
struct DeltaKeyLess {
  const int *pair;
  const i64 *rs, *qs, *re, *qe;
  __device__ bool operator()(int a, int b) const {
    if(pair[a] != pair[b]) return pair[a] < pair[b];
    if(rs[a] != rs[b]) return rs[a] < rs[b];
    if(qs[a] != qs[b]) return qs[a] < qs[b];
    if(re[a] != re[b]) return re[a] < re[b];
    return qe[a] < qe[b];
  }
};

__global__ void iota_kernel(int n, int *idx) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k < n) {
    idx[k] = k;
  }
}

template <class Less>
static int sort_indices(int n, const Less &less, std::vector<int> &order) {
  order.resize((size_t)n);
  if(n == 0) {
    return PM_OK;
  }
  DevBuf in, out, tmp;
  PM_TRY(in.alloc((size_t)n * 4));
  PM_TRY(out.alloc((size_t)n * 4));
  iota_kernel<<<(n + 255) / 256, 256>>>(n, (int *)in.p);
  PM_HIP(hipGetLastError());
  size_t bytes = 0;
  PM_HIP(rocprim::merge_sort(nullptr, bytes, (int *)in.p, (int *)out.p, (size_t)n, less));
  PM_TRY(tmp.alloc(bytes ? bytes : 8));
  PM_HIP(rocprim::merge_sort(tmp.p, bytes, (int *)in.p, (int *)out.p, (size_t)n, less));
  PM_HIP(hipMemcpy(order.data(), out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return PM_OK;
}

// m_delta_stream_writer.hh:14-53 on the flat table
static void append_offsets(const DeltaTable &t, size_t d, std::vector<long long> &out) {
  size_t r = (size_t)t.ref_gap_off[d], r1 = (size_t)t.ref_gap_off[d + 1];
  size_t q = (size_t)t.qry_gap_off[d], q1 = (size_t)t.qry_gap_off[d + 1];
  long long column = 0;
  out.clear();
  while(r < r1 || q < q1) {
    bool take_ref = r < r1 && (q >= q1 || t.ref_gap_start[r] < t.qry_gap_start[q]);
    long long s = take_ref ? t.ref_gap_start[r] : t.qry_gap_start[q];
    long long e = take_ref ? t.ref_gap_end[r] : t.qry_gap_end[q];
    long long sign = take_ref ? -1 : 1;
    out.push_back(sign * (s - column));
    long long len = (s <= e ? e - s : s - e) + 1;
    for(long long k = len - 1; k > 0; --k) {
      out.push_back(sign);
    }
    column = e;
    if(take_ref) {
      ++r;
    }
    else {
      ++q;
    }
  }
  out.push_back(0);
}

static void put_i64(std::string &buf, long long v) {
  char tmp[24];
  int n = 0;
  unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
  do {
    tmp[n++] = (char)('0' + u % 10);
    u /= 10;
  } while(u);
  if(v < 0) {
    tmp[n++] = '-';
  }
  while(n) {
    buf.push_back(tmp[--n]);
  }
}

static bool same_gaps(const DeltaTable &t, int a, int b) {
  auto eq = [&](const std::vector<long long> &off, const std::vector<long long> &s, const std::vector<long long> &e) {
    long long na = off[a + 1] - off[a], nb = off[b + 1] - off[b];
    if(na != nb) {
      return false;
    }
    for(long long k = 0; k < na; ++k) {
      if(s[off[a] + k] != s[off[b] + k] || e[off[a] + k] != e[off[b] + k]) {
        return false;
      }
    }
    return true;
  };
  return eq(t.ref_gap_off, t.ref_gap_start, t.ref_gap_end) && eq(t.qry_gap_off, t.qry_gap_start, t.qry_gap_end);
}

// ------------------------------------------------------------------ a16: MAF rows

struct MafRows {
  std::vector<std::string> genomes; // sorted, as std::map iterates (maf_analyzer.cc:27-29)
  std::vector<int> genome;          // per row
  std::vector<long long> start, end; // forward range (abs), 1-based inclusive
  std::vector<long long> size_of;   // per genome: the last row's src_size (maf_analyzer_missing.cc:147)
};

// maf_read_stream.cc:7-45 and maf_read_stream.hh:23-47, including what the reference does at end of file:
// a final line without a newline sets eof, which ends the stream before a block that starts on it.
static int parse_maf(const std::string &text, MafRows &out) {
  struct Raw {
    std::string genome;
    long long s, e, src;
  };
  std::vector<Raw> rows;
  std::map<std::string, long long> sizes;
  const char *p = text.data(), *end = p + text.size();
  bool eof = false;
  auto getline = [&](const char *&b, const char *&e) {
    if(p >= end) {
      eof = true;
      b = e = end;
      return false;
    }
    b = p;
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    if(nl) {
      e = nl;
      p = nl + 1;
    }
    else {
      e = end;
      p = end;
      eof = true;
    }
    return true;
  };
  for(;;) {
    const char *b = nullptr, *e = nullptr;
    bool got;
    while((got = getline(b, e)) && (b == e || *b == '#')) {
    }
    if(eof || !got) {
      break;
    }
    if(*b != 'a') {
      break;
    }
    // `a <score> <label>`: two tokens after the marker character
    {
      const char *q = b + 1;
      int toks = 0;
      while(q < e) {
        while(q < e && (*q == ' ' || *q == '\t')) ++q;
        if(q < e) {
          ++toks;
          while(q < e && *q != ' ' && *q != '\t') ++q;
        }
      }
      if(toks < 2) {
        return fail(PM_E_PARSE, "MAF: `a` line needs a score and a label token (Maf_parse_error)");
      }
    }
    while(getline(b, e) && b < e && *b == 's') {
      std::vector<std::pair<const char *, const char *> > tok;
      const char *q = b;
      while(q < e) {
        while(q < e && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if(q < e) {
          const char *t0 = q;
          while(q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
          tok.push_back(std::make_pair(t0, q));
        }
      }
      if(tok.size() < 7) {
        return fail(PM_E_PARSE, "MAF: short `s` line (Maf_parse_error)");
      }
      auto num = [&](size_t k, long long &v) {
        char *endp = nullptr;
        std::string sx(tok[k].first, tok[k].second);
        v = strtoll(sx.c_str(), &endp, 10);
        return endp && *endp == 0 && !sx.empty();
      };
      Raw r;
      long long st, sz;
      r.genome.assign(tok[1].first, tok[1].second);
      if(!num(2, st) || !num(3, sz) || !num(5, r.src)) {
        return fail(PM_E_PARSE, "MAF: bad number on `s` line");
      }
      bool fwd = tok[4].second - tok[4].first == 1 && *tok[4].first == '+';
      long long rs, re; // m_range.hh:106-115
      if(fwd) {
        rs = st + 1;
        re = st + sz;
      }
      else {
        rs = r.src - st;
        re = r.src - st - (sz - 1);
      }
      r.s = std::min(rs, re);
      r.e = std::max(rs, re);
      sizes[r.genome] = r.src;
      rows.push_back(r);
    }
  }
  std::map<std::string, int> id;
  for(std::map<std::string, long long>::iterator it = sizes.begin(); it != sizes.end(); ++it) {
    id[it->first] = (int)out.genomes.size();
    out.genomes.push_back(it->first);
    out.size_of.push_back(it->second);
  }
  for(size_t k = 0; k < rows.size(); ++k) {
    out.genome.push_back(id[rows[k].genome]);
    out.start.push_back(rows[k].s);
    out.end.push_back(rows[k].e);
  }
  return PM_OK;
}

// ------------------------------------------------------------------ a17: coverage kernels

struct RowKeyLess {
  const int *genome;
  const i64 *start;
  __device__ bool operator()(int a, int b) const {
    if(genome[a] != genome[b]) return genome[a] < genome[b];
    if(start[a] != start[b]) return start[a] < start[b];
    return a < b;
  }
};

// Over rows sorted by (genome, start): head[k] = 1 when row k starts a new covered run (first of its genome or
// not touching its predecessor); *overlap is set when two rows of a genome overlap (then the reference's
// insertion is order dependent and the exact replay kernel below is used instead).
__global__ void coverage_heads_kernel(int n, const int *order, const int *genome, const i64 *start, const i64 *end, int *head,
                                      int *overlap) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) {
    return;
  }
  int r = order[k];
  int h = 1;
  if(k > 0) {
    int q = order[k - 1];
    if(genome[q] == genome[r]) {
      if(start[r] <= end[q]) {
        atomicOr(overlap, 1);
      }
      if(end[q] + 1 == start[r]) {
        h = 0;
      }
    }
  }
  head[k] = h;
}

// run_id = inclusive scan of head - 1.  One thread per row writes its run's start (heads) and end (row before the
// next head, or the last row).
__global__ void coverage_runs_kernel(int n, const int *order, const int *genome, const i64 *start, const i64 *end, const int *head,
                                     const int *run_of, int *run_genome, i64 *run_start, i64 *run_end) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) {
    return;
  }
  int r = order[k];
  int run = run_of[k] - 1;
  if(head[k]) {
    run_genome[run] = genome[r];
    run_start[run] = start[r];
  }
  if(k == n - 1 || head[k + 1]) {
    run_end[run] = end[r];
  }
}

// The reference's insertion replayed exactly (maf_analyzer_missing.cc:38-104), one lane per genome, rows in file
// order.  Only used when rows of a genome overlap, where the result depends on the order of insertion.
__global__ void coverage_replay_kernel(int n_genomes, const int *row_off, const int *rows, const i64 *start, const i64 *end,
                                       i64 *vs, i64 *ve, int *count) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n_genomes) {
    return;
  }
  int base = row_off[g], m = row_off[g + 1] - base;
  i64 *s = vs + base, *e = ve + base;
  int n = 0;
  for(int k = 0; k < m; ++k) {
    int r = rows[base + k];
    i64 rs = start[r], re = end[r];
    int at = 0;
    while(at < n && !(re < s[at])) { // :25-36
      ++at;
    }
    if(n == 0) {
      s[0] = rs;
      e[0] = re;
      n = 1;
      continue;
    }
    bool touches_prev = at != 0 && e[at - 1] + 1 == rs;
    bool touches_next = at != n && re + 1 == s[at];
    if(touches_prev && touches_next) {
      s[at] = s[at - 1];
      for(int j = at - 1; j + 1 < n; ++j) {
        s[j] = s[j + 1];
        e[j] = e[j + 1];
      }
      --n;
    }
    else if(touches_next) {
      s[at] = rs;
    }
    else if(touches_prev) {
      e[at - 1] = re;
    }
    else {
      for(int j = n; j > at; --j) {
        s[j] = s[j - 1];
        e[j] = e[j - 1];
      }
      s[at] = rs;
      e[at] = re;
      ++n;
    }
  }
  count[g] = n;
}

// _add_missing (maf_analyzer_missing.cc:106-135) per covered run: up to two missing ranges, written to fixed slots
// 2k and 2k+1 (valid flag per slot); the upstream quirks are kept (first range ends at end-1 of the FIRST covered
// run, :115; the loop stops one short of the last run, :119-126).
__global__ void coverage_missing_kernel(int n_runs, const int *run_genome, const i64 *run_start, const i64 *run_end, const i64 *size_of,
                                        i64 *ms, i64 *me, int *valid) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n_runs) {
    return;
  }
  int g = run_genome[k];
  bool first = k == 0 || run_genome[k - 1] != g;
  bool last = k == n_runs - 1 || run_genome[k + 1] != g;
  valid[2 * k] = valid[2 * k + 1] = 0;
  if(first && 1 < run_start[k]) {
    ms[2 * k] = 1;
    me[2 * k] = run_end[k] - 1;
    valid[2 * k] = 1;
  }
  if(!first && !last) {
    ms[2 * k] = run_end[k - 1] + 1;
    me[2 * k] = run_start[k] - 1;
    valid[2 * k] = 1;
  }
  if(last && run_end[k] < size_of[g]) {
    ms[2 * k + 1] = run_end[k] + 1;
    me[2 * k + 1] = size_of[g];
    valid[2 * k + 1] = 1;
  }
}

} // namespace pm

using namespace pm;

extern "C" {

int pm_sort_delta(const char *in_path, const char *out_path, int device) {
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  std::string text;
  FILE *fin = in_path ? fopen(in_path, "rb") : stdin;
  if(!fin) {
    return fail(PM_E_IO, std::string("cannot open ") + in_path);
  }
  bool ok = read_stream(fin, text);
  if(in_path) {
    fclose(fin);
  }
  if(!ok) {
    return fail(PM_E_IO, "read failed");
  }
  DeltaTable t;
  PM_TRY(parse_delta_text(text, in_path ? in_path : "<stdin>", t));
  int n = (int)t.ref_start.size();
  // header pairs ranked in std::string order (m_sort_delta.cc:19-45)
  std::map<std::pair<std::string, std::string>, int> rank;
  for(int d = 0; d < n; ++d) {
    rank[std::make_pair(t.ref_name[d], t.qry_name[d])] = 0;
  }
  int next = 0;
  for(std::map<std::pair<std::string, std::string>, int>::iterator it = rank.begin(); it != rank.end(); ++it) {
    it->second = next++;
  }
  std::vector<int> pair((size_t)n);
  for(int d = 0; d < n; ++d) {
    pair[d] = rank[std::make_pair(t.ref_name[d], t.qry_name[d])];
  }
  std::vector<int> order;
  {
    DevBuf d_pair, d_rs, d_qs, d_re, d_qe;
    PM_TRY(d_pair.upload(pair.data(), (size_t)n * 4, nullptr));
    PM_TRY(d_rs.upload(t.ref_start.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_qs.upload(t.qry_start.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_re.upload(t.ref_end.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_qe.upload(t.qry_end.data(), (size_t)n * 8, nullptr));
    DeltaKeyLess less{(const int *)d_pair.p, (const i64 *)d_rs.p, (const i64 *)d_qs.p, (const i64 *)d_re.p, (const i64 *)d_qe.p};
    PM_TRY(sort_indices(n, less, order));
  }
  // entries with identical keys: the reference's order among them is whatever libstdc++'s introsort leaves; it only
  // shows when their gap lists differ, and then this library refuses rather than guess
  for(int k = 1; k < n; ++k) {
    int a = order[k - 1], b = order[k];
    if(pair[a] == pair[b] && t.ref_start[a] == t.ref_start[b] && t.qry_start[a] == t.qry_start[b] && t.ref_end[a] == t.ref_end[b] &&
       t.qry_end[a] == t.qry_end[b] && !same_gaps(t, a, b)) {
      return fail(PM_E_INVALID, "two entries share header and coordinates but differ in gaps: upstream order is unspecified (std::sort)");
    }
  }
  FILE *fout = out_path ? fopen(out_path, "wb") : stdout;
  if(!fout) {
    return fail(PM_E_IO, std::string("cannot open ") + out_path);
  }
  std::string buf, last_ref, last_qry; // writer state starts at ("", ""), m_delta_stream_writer.hh:80
  std::vector<long long> offs;
  for(int k = 0; k < n; ++k) {
    int d = order[k];
    if(t.ref_name[d] != last_ref || t.qry_name[d] != last_qry) {
      buf.push_back('>');
      buf += t.ref_name[d];
      buf.push_back(' ');
      buf += t.qry_name[d];
      buf.push_back(' ');
      put_i64(buf, t.ref_len[d]);
      buf.push_back(' ');
      put_i64(buf, t.qry_len[d]);
      buf.push_back('\n');
      last_ref = t.ref_name[d];
      last_qry = t.qry_name[d];
    }
    put_i64(buf, t.ref_start[d]);
    buf.push_back(' ');
    put_i64(buf, t.ref_end[d]);
    buf.push_back(' ');
    put_i64(buf, t.qry_start[d]);
    buf.push_back(' ');
    put_i64(buf, t.qry_end[d]);
    buf += " 1 2 3\n";
    append_offsets(t, (size_t)d, offs);
    for(size_t j = 0; j < offs.size(); ++j) {
      put_i64(buf, offs[j]);
      buf.push_back('\n');
    }
    if(buf.size() > (1 << 20)) {
      fwrite(buf.data(), 1, buf.size(), fout);
      buf.clear();
    }
  }
  fwrite(buf.data(), 1, buf.size(), fout);
  if(out_path) {
    fclose(fout);
  }
  else {
    fflush(fout);
  }
  return PM_OK;
}

int pm_maf_analyzer(const char *maf_path, const char *out_path, int device) {
  if(!maf_path) {
    return fail(PM_E_INVALID, "pm_maf_analyzer: null path");
  }
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  std::string text;
  {
    FILE *f = fopen(maf_path, "rb");
    if(f) { // a missing file is an empty stream upstream (maf_analyzer.cc:13)
      read_stream(f, text);
      fclose(f);
    }
  }
  MafRows rows;
  PM_TRY(parse_maf(text, rows));
  int n = (int)rows.start.size(), G = (int)rows.genomes.size();
  std::vector<int> out_genome;
  std::vector<long long> out_s, out_e;
  if(n > 0) {
    DevBuf d_genome, d_start, d_end, d_size, d_head, d_run, d_overlap, d_tmp, d_order;
    PM_TRY(d_genome.upload(rows.genome.data(), (size_t)n * 4, nullptr));
    PM_TRY(d_start.upload(rows.start.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_end.upload(rows.end.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_size.upload(rows.size_of.data(), (size_t)G * 8, nullptr));
    std::vector<int> order;
    RowKeyLess less{(const int *)d_genome.p, (const i64 *)d_start.p};
    PM_TRY(sort_indices(n, less, order));
    PM_TRY(d_order.upload(order.data(), (size_t)n * 4, nullptr));
    PM_TRY(d_head.alloc((size_t)n * 4));
    PM_TRY(d_run.alloc((size_t)n * 4));
    PM_TRY(d_overlap.alloc(4));
    PM_HIP(hipMemset(d_overlap.p, 0, 4));
    unsigned blocks = (unsigned)((n + 255) / 256);
    coverage_heads_kernel<<<blocks, 256>>>(n, (const int *)d_order.p, (const int *)d_genome.p, (const i64 *)d_start.p, (const i64 *)d_end.p,
                                           (int *)d_head.p, (int *)d_overlap.p);
    PM_HIP(hipGetLastError());
    int overlap = 0;
    PM_HIP(hipMemcpy(&overlap, d_overlap.p, 4, hipMemcpyDeviceToHost));
    DevBuf d_rg, d_rs, d_re;
    int n_runs = 0;
    if(!overlap) {
      size_t bytes = 0;
      PM_HIP(rocprim::inclusive_scan(nullptr, bytes, (int *)d_head.p, (int *)d_run.p, (size_t)n, rocprim::plus<int>()));
      PM_TRY(d_tmp.alloc(bytes ? bytes : 8));
      PM_HIP(rocprim::inclusive_scan(d_tmp.p, bytes, (int *)d_head.p, (int *)d_run.p, (size_t)n, rocprim::plus<int>()));
      PM_HIP(hipMemcpy(&n_runs, (int *)d_run.p + (n - 1), 4, hipMemcpyDeviceToHost));
      PM_TRY(d_rg.alloc((size_t)n_runs * 4));
      PM_TRY(d_rs.alloc((size_t)n_runs * 8));
      PM_TRY(d_re.alloc((size_t)n_runs * 8));
      coverage_runs_kernel<<<blocks, 256>>>(n, (const int *)d_order.p, (const int *)d_genome.p, (const i64 *)d_start.p, (const i64 *)d_end.p,
                                            (const int *)d_head.p, (const int *)d_run.p, (int *)d_rg.p, (i64 *)d_rs.p, (i64 *)d_re.p);
      PM_HIP(hipGetLastError());
    }
    else {
      // rows of each genome in file order
      std::vector<int> row_off((size_t)G + 1, 0), grows((size_t)n);
      for(int k = 0; k < n; ++k) {
        ++row_off[(size_t)rows.genome[k] + 1];
      }
      for(int g = 0; g < G; ++g) {
        row_off[(size_t)g + 1] += row_off[g];
      }
      std::vector<int> fill(row_off.begin(), row_off.end() - 1);
      for(int k = 0; k < n; ++k) {
        grows[(size_t)fill[rows.genome[k]]++] = k;
      }
      DevBuf d_off, d_rows, d_vs, d_ve, d_cnt;
      PM_TRY(d_off.upload(row_off.data(), ((size_t)G + 1) * 4, nullptr));
      PM_TRY(d_rows.upload(grows.data(), (size_t)n * 4, nullptr));
      PM_TRY(d_vs.alloc((size_t)n * 8));
      PM_TRY(d_ve.alloc((size_t)n * 8));
      PM_TRY(d_cnt.alloc((size_t)G * 4));
      coverage_replay_kernel<<<(unsigned)((G + 63) / 64), 64>>>(G, (const int *)d_off.p, (const int *)d_rows.p, (const i64 *)d_start.p,
                                                                (const i64 *)d_end.p, (i64 *)d_vs.p, (i64 *)d_ve.p, (int *)d_cnt.p);
      PM_HIP(hipGetLastError());
      std::vector<int> cnt((size_t)G);
      std::vector<long long> vs((size_t)n), ve((size_t)n);
      PM_HIP(hipMemcpy(cnt.data(), d_cnt.p, (size_t)G * 4, hipMemcpyDeviceToHost));
      PM_HIP(hipMemcpy(vs.data(), d_vs.p, (size_t)n * 8, hipMemcpyDeviceToHost));
      PM_HIP(hipMemcpy(ve.data(), d_ve.p, (size_t)n * 8, hipMemcpyDeviceToHost));
      std::vector<int> rg;
      std::vector<long long> rs, re;
      for(int g = 0; g < G; ++g) {
        for(int k = 0; k < cnt[g]; ++k) {
          rg.push_back(g);
          rs.push_back(vs[(size_t)row_off[g] + k]);
          re.push_back(ve[(size_t)row_off[g] + k]);
        }
      }
      n_runs = (int)rg.size();
      PM_TRY(d_rg.upload(rg.data(), (size_t)n_runs * 4, nullptr));
      PM_TRY(d_rs.upload(rs.data(), (size_t)n_runs * 8, nullptr));
      PM_TRY(d_re.upload(re.data(), (size_t)n_runs * 8, nullptr));
    }
    DevBuf d_ms, d_me, d_valid;
    PM_TRY(d_ms.alloc((size_t)n_runs * 16));
    PM_TRY(d_me.alloc((size_t)n_runs * 16));
    PM_TRY(d_valid.alloc((size_t)n_runs * 8));
    coverage_missing_kernel<<<(unsigned)((n_runs + 255) / 256), 256>>>(n_runs, (const int *)d_rg.p, (const i64 *)d_rs.p, (const i64 *)d_re.p,
                                                                       (const i64 *)d_size.p, (i64 *)d_ms.p, (i64 *)d_me.p, (int *)d_valid.p);
    PM_HIP(hipGetLastError());
    std::vector<int> rg((size_t)n_runs), valid((size_t)n_runs * 2);
    std::vector<long long> ms((size_t)n_runs * 2), me((size_t)n_runs * 2);
    PM_HIP(hipMemcpy(rg.data(), d_rg.p, (size_t)n_runs * 4, hipMemcpyDeviceToHost));
    PM_HIP(hipMemcpy(valid.data(), d_valid.p, (size_t)n_runs * 8, hipMemcpyDeviceToHost));
    PM_HIP(hipMemcpy(ms.data(), d_ms.p, (size_t)n_runs * 16, hipMemcpyDeviceToHost));
    PM_HIP(hipMemcpy(me.data(), d_me.p, (size_t)n_runs * 16, hipMemcpyDeviceToHost));
    for(int k = 0; k < n_runs; ++k) {
      for(int j = 0; j < 2; ++j) {
        if(valid[(size_t)2 * k + j]) {
          out_genome.push_back(rg[k]);
          out_s.push_back(ms[(size_t)2 * k + j]);
          out_e.push_back(me[(size_t)2 * k + j]);
        }
      }
    }
  }
  FILE *fout = out_path ? fopen(out_path, "wb") : stdout;
  if(!fout) {
    return fail(PM_E_IO, std::string("cannot open ") + out_path);
  }
  std::string buf;
  size_t at = 0;
  for(int g = 0; g < G; ++g) { // maf_analyzer.cc:27-36
    buf += "--------\n";
    while(at < out_genome.size() && out_genome[at] == g) {
      buf += rows.genomes[g];
      buf.push_back('\t');
      put_i64(buf, out_s[at]);
      buf.push_back('\t');
      put_i64(buf, out_e[at]);
      buf.push_back('\n');
      ++at;
    }
  }
  fwrite(buf.data(), 1, buf.size(), fout);
  if(out_path) {
    fclose(fout);
  }
  else {
    fflush(fout);
  }
  return PM_OK;
}

} // extern "C"
