// untranslate.hip -- `mugsy_profiles untranslate`: the MAF mugsyWGA wrote over PROFILE names -> a MAF over the real
// genomes, the step after the translate path in the task script (lib/base/mugsy_profiles_task.ml:70-74).
// Reference (OCaml; cannot be built or run in this image, so RESTATED FROM SOURCE, NOT EXECUTED -- pinned by the
// hand-derived fixture tests/golden/untranslate_handmade and the Python transcription oracle/untranslate_oracle.py):
//   lib/profiles/m_untranslate.ml:15-221, lib/profiles/m_profile.ml:69-120 (reader), :163-239 (OCaml subset_profile)
// For every `s` line of the input and every row of the block it names: clip the row to the line's column range
// (binary searches on the same prefix tables as the translate path), derive the genome coordinates, and rebuild
// the row's text by walking the line's text: each non-gap column takes the next character of the row's own text
// (reversed and complemented when the strands differ).  Coordinates: one lane per (line, row); text: one thread
// per output byte with a scan of the line's non-gap columns.
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
#include "translate_store.hpp"

namespace pm {

struct UnitOut {
  int valid;   // 0: the row is all gaps over the range (None, m_untranslate.ml:106-109)
  int status;  // PM_ST_*
  int forward; // strand of the output line
  int reversed; // text walks the row backwards and is complemented
  i64 start, size; // MAF start / size
  i64 sub_pos, sub_len; // the row's text columns [sub_pos, sub_pos + sub_len) (0-based)
};

// m_profile.ml:189-239, the OCaml subset_profile: no lower-bound check, an all-gap end gives None (not an exception)
__device__ inline int subset_profile_ml(const PV &p, i64 s, i64 e, R2 &seq, bool &none) {
  none = false;
  if(s > e) {
    i64 t = s;
    s = e;
    e = t;
  }
  if(!(s < p.len + 1 && e < p.len + 1)) {
    return PM_ST_PROFILE_IDX_OUT_OF_RANGE;
  }
  int lo = 0, hi = p.n;
  while(lo < hi) {
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
  while(lo < hi) {
    int mid = (lo + hi) >> 1;
    if(p.g[mid].s <= e) {
      lo = mid + 1;
    }
    else {
      hi = mid;
    }
  }
  int n = lo - first;
  i64 qs = s, qe = e;
  if(n > 0) {
    R2 a{imax(p.g[first].s, s), imin(p.g[first].e, e)};
    R2 b{imax(p.g[lo - 1].s, s), imin(p.g[lo - 1].e, e)};
    if(n == 1 && a.s == s && a.e == e) {
      none = true;
      return PM_ST_OK;
    }
    if(a.s == s) {
      qs = a.e + 1;
    }
    if(b.e == e) {
      qe = b.s - 1;
    }
  }
  bool n1, n2;
  i64 ss = 0, se = 0;
  int st = seq_idx_of_profile_idx(p, qs, ss, n1);
  if(st) {
    return st;
  }
  st = seq_idx_of_profile_idx(p, qe, se, n2);
  if(st) {
    return st;
  }
  none = n1 || n2;
  seq = R2{ss, se};
  return PM_ST_OK;
}

// m_untranslate.ml:55-110 minus the text
__global__ void untranslate_units_kernel(RowsD rows, const i64 *row_src_size, const i64 *row_text_len, i64 n_units, const int *u_line,
                                         const int *u_row, const i64 *line_s, const i64 *line_e, const i64 *line_nongap, UnitOut *out) {
  i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(u >= n_units) {
    return;
  }
  UnitOut o;
  memset(&o, 0, sizeof o);
  int r = u_row[u], l = u_line[u];
  if(rows.bad[r]) {
    o.status = PM_ST_MALFORMED_INPUT;
    out[u] = o;
    return;
  }
  PV p = row_view(rows, r);
  i64 s = line_s[l], e = line_e[l];
  bool overlap_fwd = s <= e;
  R2 seq{0, 0};
  bool none = false;
  // String.sub of the row text happens before the None test (m_profile.ml:203) and throws when out of bounds
  i64 lo = s < e ? s : e, hi = s < e ? e : s;
  int st = PM_ST_OK;
  if(!(lo < p.len + 1 && hi < p.len + 1)) {
    st = PM_ST_PROFILE_IDX_OUT_OF_RANGE;
  }
  else if(row_text_len[r] > 0 && (lo - 1 < 0 || hi > row_text_len[r])) {
    st = PM_ST_TEXT_RANGE;
  }
  else {
    st = subset_profile_ml(p, s, e, seq, none);
  }
  if(st || none) {
    o.status = st;
    out[u] = o;
    return;
  }
  bool p_fwd = fwd(p.range);
  R2 real = overlap_fwd ? seq : R2{seq.e, seq.s}; // get_real_range, :55-60
  bool dir_fwd = overlap_fwd ? p_fwd : !p_fwd;
  o.size = rlen(real);
  o.start = fwd(real) ? real.s - 1 : row_src_size[r] - real.s; // get_start_size, :62-69
  o.forward = dir_fwd;
  o.reversed = p_fwd != dir_fwd;
  o.sub_pos = lo - 1;
  o.sub_len = hi - lo + 1;
  const i64 avail = row_text_len[r] == 0 ? 0 : o.sub_len; // characters expand_text may take from the row (m_untranslate.ml:38-52)
  if(line_nongap[l] > avail) {
    o.status = PM_ST_TEXT_RANGE;
    out[u] = o;
    return;
  }
  o.valid = 1;
  out[u] = o;
}

__device__ __forceinline__ unsigned char complement(unsigned char c) { // m_untranslate.ml:15-24
  switch(c) {
  case 'A': return 'T';
  case 'a': return 't';
  case 'T': return 'A';
  case 't': return 'a';
  case 'C': return 'G';
  case 'c': return 'g';
  case 'G': return 'C';
  case 'g': return 'c';
  default: return c;
  }
}

// expand_text (+ reverse + complement), m_untranslate.ml:38-52,88-98: one thread per output byte
__global__ void untranslate_text_kernel(i64 n_bytes, i64 n_units, const i64 *unit_text_off, const int *u_line, const int *u_row,
                                        const UnitOut *units, const i64 *line_off, const unsigned char *line_text, const int *nongap_before,
                                        const i64 *row_text_off, const unsigned char *row_text, unsigned char *out) {
  i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n_bytes) {
    return;
  }
  // unit of this byte: last u with unit_text_off[u] <= g.  The search runs once per workgroup, for the group's first
  // byte (wave-uniform addresses: scalar loads); a unit is hundreds of bytes, so a thread's own unit is that one or one
  // of the next few, found by walking; a thread that would walk far searches the rest instead.
  const i64 g0 = (i64)blockIdx.x * blockDim.x;
  i64 lo = 0, hi = n_units;
  while(hi - lo > 1) {
    i64 mid = (lo + hi) >> 1;
    if(unit_text_off[mid] <= g0) {
      lo = mid;
    }
    else {
      hi = mid;
    }
  }
  int walked = 0;
  while(lo + 1 < n_units && unit_text_off[lo + 1] <= g) {
    ++lo;
    if(++walked == 8) {
      hi = n_units;
      while(hi - lo > 1) {
        i64 mid = (lo + hi) >> 1;
        if(unit_text_off[mid] <= g) {
          lo = mid;
        }
        else {
          hi = mid;
        }
      }
      break;
    }
  }
  const UnitOut o = units[lo];
  if(!o.valid) {
    return;
  }
  int l = u_line[lo], r = u_row[lo];
  i64 at = line_off[l] + (g - unit_text_off[lo]);
  unsigned char ch = line_text[at];
  if(ch != '-') {
    i64 k = nongap_before[at] - nongap_before[line_off[l]];
    i64 src = o.reversed ? o.sub_pos + o.sub_len - 1 - k : o.sub_pos + k;
    ch = row_text[row_text_off[r] + src];
    if(o.reversed) {
      ch = complement(ch);
    }
  }
  out[g] = ch;
}

__global__ void nongap_flag_kernel(i64 n, const unsigned char *text, int *flag) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) {
    flag[i] = text[i] != '-';
  }
}

struct FullRows {
  std::vector<std::string> major, seq_name;
  std::vector<long long> start, end, length, src_size, gap_off, gap_start, gap_end, text_off;
  std::string text;
};

static std::string strip(const char *b, const char *e) {
  while(b < e && (*b == ' ' || *b == '\t' || *b == '\r' || *b == '\n')) ++b;
  while(e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r' || e[-1] == '\n')) --e;
  return std::string(b, e);
}

// m_profile.ml:69-120 with ~lite:false
static int parse_profiles_full(const std::string &path, FullRows &rows) {
  FILE *f = fopen(path.c_str(), "rb");
  if(!f) {
    return fail(PM_E_IO, "cannot open " + path);
  }
  std::string data;
  read_stream(f, data);
  fclose(f);
  const char *p = data.data(), *end = p + data.size();
  auto getline = [&](const char *&b, const char *&e) {
    if(p >= end) {
      return false;
    }
    b = p;
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    e = nl ? nl : end;
    p = nl ? nl + 1 : end;
    return true;
  };
  const char *b, *e;
  while(getline(b, e)) {
    std::vector<std::string> f7;
    const char *q = b;
    for(;;) { // String.split_on_chars ~on:[' ']: exactly seven fields
      const char *sp = (const char *)memchr(q, ' ', (size_t)(e - q));
      f7.push_back(std::string(q, sp ? sp : e));
      if(!sp) {
        break;
      }
      q = sp + 1;
    }
    if(f7.size() != 7) {
      return fail(PM_E_PARSE, path + ": Error reading profile index file line (m_profile.ml:116)");
    }
    long long v[4];
    for(int k = 0; k < 4; ++k) {
      char *endp = nullptr;
      v[k] = strtoll(f7[(size_t)3 + k].c_str(), &endp, 10);
      if(f7[(size_t)3 + k].empty() || *endp) {
        return fail(PM_E_PARSE, path + ": int_of_string failure in a profile header");
      }
    }
    rows.major.push_back(f7[0]);
    rows.seq_name.push_back(f7[2]);
    rows.start.push_back(v[0]);
    rows.end.push_back(v[1]);
    rows.length.push_back(v[2]);
    rows.src_size.push_back(v[3]);
    while(getline(b, e) && !(e - b == 1 && *b == '0')) {
      const char *sp = (const char *)memchr(b, ' ', (size_t)(e - b));
      if(!sp) {
        return fail(PM_E_PARSE, path + ": Invalid string reading profile index (m_profile.ml:75)");
      }
      rows.gap_start.push_back(strtoll(std::string(b, sp).c_str(), nullptr, 10));
      rows.gap_end.push_back(strtoll(std::string(sp + 1, e).c_str(), nullptr, 10));
    }
    rows.gap_off.push_back((long long)rows.gap_start.size());
    if(!getline(b, e)) {
      return fail(PM_E_PARSE, path + ": Early end of file (m_profile.ml:104)");
    }
    rows.text += strip(b, e);
    rows.text_off.push_back((long long)rows.text.size());
  }
  return PM_OK;
}

} // namespace pm

using namespace pm;

extern "C" int pm_untranslate(const char *const *profile_dirs, int n_dirs, const char *in_maf, const char *out_maf, int device) {
  if(n_dirs < 0 || (n_dirs > 0 && !profile_dirs) || !in_maf || !out_maf) {
    return fail(PM_E_INVALID, "pm_untranslate: null argument");
  }
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  FullRows rows;
  rows.gap_off.push_back(0);
  rows.text_off.push_back(0);
  for(int d = 0; d < n_dirs; ++d) {
    PM_TRY(parse_profiles_full(std::string(profile_dirs[d]) + "/profiles", rows));
  }
  int n_rows = (int)rows.start.size();
  std::map<std::string, std::vector<int> > by_block; // rows of a block in file order (m_untranslate.ml:26-36,153-166)
  for(int r = 0; r < n_rows; ++r) {
    by_block[rows.major[r]].push_back(r);
  }
  // the input MAF: pass-through lines and `s` lines
  std::string maf;
  {
    FILE *f = fopen(in_maf, "rb");
    if(!f) {
      return fail(PM_E_IO, std::string("cannot open ") + in_maf);
    }
    read_stream(f, maf);
    fclose(f);
  }
  struct Line {
    int kind;          // 0 pass through, 1 `s` line
    std::string text;  // pass-through text
    int s_index;       // index among the `s` lines
  };
  std::vector<Line> lines;
  std::vector<long long> line_s, line_e, line_nongap, line_off(1, 0);
  std::string line_text;
  std::vector<int> u_line, u_row, line_first_unit(1, 0);
  {
    const char *p = maf.data(), *end = p + maf.size();
    while(p < end) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      const char *b = p, *e = nl ? nl : end;
      p = nl ? nl + 1 : end;
      size_t len = (size_t)(e - b);
      if(len >= 6 && memcmp(b, "##maf ", 6) == 0) {
        continue; // m_untranslate.ml:129-133
      }
      if(len == 0 || *b == '#' || (len >= 8 && memcmp(b, "a score=", 8) == 0)) {
        lines.push_back(Line{0, std::string(b, e), -1});
        continue;
      }
      if(!(len >= 2 && b[0] == 's' && b[1] == ' ')) {
        return fail(PM_E_PARSE, "untranslate: Unknown line (m_untranslate.ml:148)");
      }
      std::vector<std::pair<const char *, const char *> > tok;
      const char *q = b;
      while(q < e) {
        while(q < e && (*q == ' ' || *q == '\t')) ++q;
        if(q < e) {
          const char *t0 = q;
          while(q < e && *q != ' ' && *q != '\t') ++q;
          tok.push_back(std::make_pair(t0, q));
        }
      }
      if(tok.size() != 7) {
        return fail(PM_E_PARSE, "untranslate: Unknown maf line (m_profile_stream.ml:21)");
      }
      long long st = strtoll(std::string(tok[2].first, tok[2].second).c_str(), nullptr, 10);
      long long sz = strtoll(std::string(tok[3].first, tok[3].second).c_str(), nullptr, 10);
      long long src = strtoll(std::string(tok[5].first, tok[5].second).c_str(), nullptr, 10);
      char d = *tok[4].first;
      if(tok[4].second - tok[4].first != 1 || (d != '+' && d != '-')) {
        return fail(PM_E_PARSE, "untranslate: Invalid direction");
      }
      std::string name(tok[1].first, tok[1].second);
      std::map<std::string, std::vector<int> >::const_iterator it = by_block.find(name);
      if(it == by_block.end()) {
        return fail(PM_E_PARSE, "untranslate: no profile block named " + name + " (Not_found)");
      }
      int li = (int)line_s.size();
      line_s.push_back(d == '+' ? st + 1 : src - st); // m_range.ml:60-65
      line_e.push_back(d == '+' ? st + sz : src - st - (sz - 1));
      long long ng = 0;
      for(const char *c = tok[6].first; c < tok[6].second; ++c) {
        ng += *c != '-';
      }
      line_nongap.push_back(ng);
      line_text.append(tok[6].first, tok[6].second);
      line_off.push_back((long long)line_text.size());
      for(size_t k = 0; k < it->second.size(); ++k) {
        u_line.push_back(li);
        u_row.push_back(it->second[k]);
      }
      line_first_unit.push_back((int)u_line.size());
      lines.push_back(Line{1, std::string(), li});
    }
  }
  long long n_units = (long long)u_line.size();
  std::vector<UnitOut> units((size_t)n_units);
  std::vector<long long> unit_text_off((size_t)n_units + 1, 0);
  for(long long u = 0; u < n_units; ++u) {
    int l = u_line[(size_t)u];
    unit_text_off[(size_t)u + 1] = unit_text_off[(size_t)u] + (line_off[(size_t)l + 1] - line_off[l]);
  }
  std::string out_text((size_t)unit_text_off[(size_t)n_units], '\0');
  if(n_units > 0) {
    pm_rows_t rv;
    rv.n = n_rows;
    rv.start = (const int64_t *)rows.start.data();
    rv.end = (const int64_t *)rows.end.data();
    rv.length = (const int64_t *)rows.length.data();
    rv.gap_off = (const int64_t *)rows.gap_off.data();
    rv.gap_start = (const int64_t *)rows.gap_start.data();
    rv.gap_end = (const int64_t *)rows.gap_end.data();
    RowsStore store;
    PM_TRY(upload_rows(&rv, store, nullptr));
    std::vector<long long> row_text_len((size_t)n_rows);
    for(int r = 0; r < n_rows; ++r) {
      row_text_len[r] = rows.text_off[(size_t)r + 1] - rows.text_off[r];
    }
    DevBuf d_src, d_tlen, d_ul, d_ur, d_ls, d_le, d_lng, d_units;
    PM_TRY(d_src.upload(rows.src_size.data(), (size_t)n_rows * 8, nullptr));
    PM_TRY(d_tlen.upload(row_text_len.data(), (size_t)n_rows * 8, nullptr));
    PM_TRY(d_ul.upload(u_line.data(), (size_t)n_units * 4, nullptr));
    PM_TRY(d_ur.upload(u_row.data(), (size_t)n_units * 4, nullptr));
    PM_TRY(d_ls.upload(line_s.data(), line_s.size() * 8, nullptr));
    PM_TRY(d_le.upload(line_e.data(), line_e.size() * 8, nullptr));
    PM_TRY(d_lng.upload(line_nongap.data(), line_nongap.size() * 8, nullptr));
    PM_TRY(d_units.alloc((size_t)n_units * sizeof(UnitOut)));
    untranslate_units_kernel<<<(unsigned)((n_units + 63) / 64), 64>>>(store.view(), (const i64 *)d_src.p, (const i64 *)d_tlen.p, n_units,
                                                                      (const int *)d_ul.p, (const int *)d_ur.p, (const i64 *)d_ls.p,
                                                                      (const i64 *)d_le.p, (const i64 *)d_lng.p, (UnitOut *)d_units.p);
    PM_HIP(hipGetLastError());
    i64 n_line_bytes = (i64)line_text.size(), n_bytes = unit_text_off[(size_t)n_units];
    if(n_bytes > 0) {
      DevBuf d_ltext, d_flag, d_scan, d_tmp, d_loff, d_rtoff, d_rtext, d_uoff, d_out;
      PM_TRY(d_ltext.upload(line_text.data(), (size_t)n_line_bytes, nullptr));
      PM_TRY(d_flag.alloc((size_t)n_line_bytes * 4));
      PM_TRY(d_scan.alloc((size_t)(n_line_bytes + 1) * 4));
      nongap_flag_kernel<<<(unsigned)((n_line_bytes + 255) / 256), 256>>>(n_line_bytes, (const unsigned char *)d_ltext.p, (int *)d_flag.p);
      PM_HIP(hipGetLastError());
      size_t bytes = 0;
      PM_HIP(rocprim::exclusive_scan(nullptr, bytes, (int *)d_flag.p, (int *)d_scan.p, 0, (size_t)n_line_bytes, rocprim::plus<int>()));
      PM_TRY(d_tmp.alloc(bytes ? bytes : 8));
      PM_HIP(rocprim::exclusive_scan(d_tmp.p, bytes, (int *)d_flag.p, (int *)d_scan.p, 0, (size_t)n_line_bytes, rocprim::plus<int>()));
      PM_TRY(d_loff.upload(line_off.data(), line_off.size() * 8, nullptr));
      PM_TRY(d_rtoff.upload(rows.text_off.data(), rows.text_off.size() * 8, nullptr));
      PM_TRY(d_rtext.upload(rows.text.data(), rows.text.size(), nullptr));
      PM_TRY(d_uoff.upload(unit_text_off.data(), unit_text_off.size() * 8, nullptr));
      PM_TRY(d_out.alloc((size_t)n_bytes));
      untranslate_text_kernel<<<(unsigned)((n_bytes + 255) / 256), 256>>>(n_bytes, n_units, (const i64 *)d_uoff.p, (const int *)d_ul.p,
                                                                          (const int *)d_ur.p, (const UnitOut *)d_units.p, (const i64 *)d_loff.p,
                                                                          (const unsigned char *)d_ltext.p, (const int *)d_scan.p,
                                                                          (const i64 *)d_rtoff.p, (const unsigned char *)d_rtext.p,
                                                                          (unsigned char *)d_out.p);
      PM_HIP(hipGetLastError());
      PM_HIP(hipMemcpy(&out_text[0], d_out.p, (size_t)n_bytes, hipMemcpyDeviceToHost));
    }
    PM_HIP(hipMemcpy(units.data(), d_units.p, (size_t)n_units * sizeof(UnitOut), hipMemcpyDeviceToHost));
  }
  FILE *fo = fopen(out_maf, "wb");
  if(!fo) {
    return fail(PM_E_IO, std::string("cannot open ") + out_maf);
  }
  std::string buf = "##maf version=1 scoring=paramugsy\n"; // m_untranslate.ml:218
  int result = PM_OK;
  for(size_t k = 0; k < lines.size() && result == PM_OK; ++k) {
    if(lines[k].kind == 0) {
      buf += lines[k].text;
      buf += '\n';
      continue;
    }
    int li = lines[k].s_index;
    for(int u = line_first_unit[li]; u < line_first_unit[(size_t)li + 1]; ++u) {
      const UnitOut &o = units[(size_t)u];
      if(o.status) {
        char msg[128];
        snprintf(msg, sizeof msg, "untranslate: line %d, row %d failed with status %d", li, u_row[(size_t)u], o.status);
        result = fail(o.status == PM_ST_MALFORMED_INPUT ? PM_E_MALFORMED : PM_E_UNIT, msg);
        break;
      }
      if(!o.valid) {
        continue;
      }
      int r = u_row[(size_t)u];
      buf += "s ";
      buf += rows.seq_name[r];
      buf += ' ';
      buf += std::to_string(o.start);
      buf += ' ';
      buf += std::to_string(o.size);
      buf += o.forward ? " + " : " - ";
      buf += std::to_string(rows.src_size[r]);
      buf += ' ';
      buf.append(out_text, (size_t)unit_text_off[(size_t)u], (size_t)(unit_text_off[(size_t)u + 1] - unit_text_off[(size_t)u]));
      buf += '\n';
    }
    if(buf.size() > (1 << 20)) {
      fwrite(buf.data(), 1, buf.size(), fo);
      buf.clear();
    }
  }
  fwrite(buf.data(), 1, buf.size(), fo);
  fclose(fo);
  return result;
}
