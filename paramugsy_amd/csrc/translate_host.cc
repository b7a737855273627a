// translate_host.cc -- host side of the translate path above the C ABI: file formats in, units out,
// delta text out.  Mirrors the reference's own host logic for this path:
//   read_profile_file            lib/profiles_lib/m_profile.cc:15-85      -> parse_profiles
//   M_delta_stream ctor / next   lib/profiles_lib/m_delta.cc:72-92,148-220 -> parse_delta_file
//   _split_gaps                  lib/profiles_lib/m_delta.cc:14-68        -> split_offsets
//   _profile_map_of_dir          lib/m_translate/m_translate.cc:188-207   -> build_side_index
//   _translate_delta (loops)     lib/m_translate/m_translate.cc:650-709   -> enumerate_units
//   M_delta_stream_writer::write lib/profiles_lib/m_delta_stream_writer.hh:55-82 -> on the device (pm_job_text, translate_job.hip)
//   translate + main's 2 lines   lib/m_translate/m_translate.cc:713-730, m_translate_main.cc:35-39 -> pm_translate_files
// The arithmetic of every work unit runs on the GPU (translate_job.hip); nothing here computes a translation.
//
// Parsing is strict where the reference's iostream extraction is lax: a token that is not wholly a decimal
// integer is a PM_E_PARSE error here (the reference would read its numeric prefix).  The producers of these
// files (lib/profiles/m_profile.ml:122-135 printf "%d"; MUMmer) never emit such tokens.
#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <iterator>
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "multi.hpp"
#include "pm_internal.hpp"
#include "translate_host.hpp"

namespace pm {
int warm_translate_kernels();                   // translate_job.hip
const char *job_text_device(pm_job_t *job);     // translate_job.hip: where pm_job_text left the text
void warm_text_staging();                       // below
}

#include <sys/stat.h>
#include <unistd.h>

namespace pm {

// ------------------------------------------------------------------ small text tools

static bool read_whole_file(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if(!f) {
    return false;
  }
  std::string buf;
  char chunk[1 << 16];
  size_t n;
  while((n = fread(chunk, 1, sizeof chunk, f)) > 0) {
    buf.append(chunk, n);
  }
  fclose(f);
  out.swap(buf);
  return true;
}

// A cursor over a text buffer that hands out '\n'-terminated lines the way std::getline does
// (a final unterminated line counts; an empty trailing remainder does not).
struct Lines {
  const char *p;
  const char *end;
  explicit Lines(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
  bool next(const char *&b, const char *&e) {
    if(p >= end) {
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
    }
    return true;
  }
};

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

static bool next_token(const char *&p, const char *e, const char *&tb, const char *&te) {
  while(p < e && is_space(*p)) {
    ++p;
  }
  if(p >= e) {
    return false;
  }
  tb = p;
  while(p < e && !is_space(*p)) {
    ++p;
  }
  te = p;
  return true;
}

static bool token_to_i64(const char *b, const char *e, long long &v) {
  if(b >= e) {
    return false;
  }
  bool neg = false;
  if(*b == '-' || *b == '+') {
    neg = *b == '-';
    ++b;
  }
  if(b >= e || e - b > 18) {
    return false;
  }
  long long acc = 0;
  for(; b < e; ++b) {
    if(*b < '0' || *b > '9') {
      return false;
    }
    acc = acc * 10 + (*b - '0');
  }
  v = neg ? -acc : acc;
  return true;
}

static bool next_i64(const char *&p, const char *e, long long &v) {
  const char *tb, *te;
  return next_token(p, e, tb, te) && token_to_i64(tb, te, v);
}

// ------------------------------------------------------------------ profiles

int parse_profiles(const std::string &path, Side &side) {
  std::string text;
  side = Side();
  side.gap_off.push_back(0);
  if(!read_whole_file(path, text)) {
    return PM_OK; // the reference's ifstream on a missing file is an empty stream: zero profiles (m_translate.cc:189-195)
  }
  Lines lines(text);
  const char *b, *e;
  while(lines.next(b, e)) {
    const char *p = b, *tb, *te;
    std::string major, minor, seq;
    long long start, end, length, src_size;
    if(!next_token(p, e, tb, te)) {
      return fail(PM_E_PARSE, path + ": empty profile header line");
    }
    major.assign(tb, te);
    if(!next_token(p, e, tb, te)) {
      return fail(PM_E_PARSE, path + ": short profile header");
    }
    minor.assign(tb, te);
    if(!next_token(p, e, tb, te)) {
      return fail(PM_E_PARSE, path + ": short profile header");
    }
    seq.assign(tb, te);
    if(!next_i64(p, e, start) || !next_i64(p, e, end) || !next_i64(p, e, length) || !next_i64(p, e, src_size)) {
      return fail(PM_E_PARSE, path + ": bad profile header numbers");
    }
    if(length < 0 || length > 0xffffffffLL || src_size < 0 || src_size > 0xffffffffLL) {
      return fail(PM_E_PARSE, path + ": p_length/p_src_size outside unsigned int (m_profile.cc:27-28)");
    }
    side.major.push_back(major);
    side.seq_name.push_back(seq);
    side.start.push_back(start);
    side.end.push_back(end);
    side.length.push_back(length);
    for(;;) {
      if(!lines.next(b, e)) {
        break; // m_profile.cc:45: the loop also ends at end of file
      }
      if(e - b == 1 && *b == '0') {
        break;
      }
      p = b;
      long long gs, ge;
      if(!next_i64(p, e, gs) || !next_i64(p, e, ge)) {
        return fail(PM_E_PARSE, path + ": bad gap line");
      }
      side.gap_start.push_back(gs);
      side.gap_end.push_back(ge);
    }
    side.gap_off.push_back((long long)side.gap_start.size());
    lines.next(b, e); // the text line, never needed here (lite = true, m_translate.cc:192)
  }
  return PM_OK;
}

// ------------------------------------------------------------------ binary side file
// `<dir>/profiles.soa`, written by the make stage next to `<dir>/profiles`: the rows as the flat arrays parse_profiles builds
// (everything but the row texts), so that the translate stage does not parse 10-100 MB of text it has no use for.  The text
// file stays the interchange format (untranslate and the reference's own tools read it); the side file is used only when it
// matches the text file it was written with (size recorded inside, not older than the text file), else the text is parsed.
// Layout: "PMSOA1\0\0", then int64 {rows, gaps, bytes of names, size of the profiles text file}, then int64 arrays
// start[rows], end[rows], length[rows], gap_off[rows + 1], gap_start[gaps], gap_end[gaps], then per row "major\0seq\0".
static const char SOA_MAGIC[8] = {'P', 'M', 'S', 'O', 'A', '1', 0, 0};

int write_side_soa(const std::string &dir, const Side &side, long long profiles_text_bytes) {
  const std::string path = dir + "/profiles.soa";
  FILE *f = fopen(path.c_str(), "wb");
  if(!f) {
    return fail(PM_E_IO, "cannot create " + path);
  }
  std::string names;
  for(size_t r = 0; r < side.start.size(); ++r) {
    names += side.major[r];
    names.push_back('\0');
    names += side.seq_name[r];
    names.push_back('\0');
  }
  const long long head[4] = {(long long)side.start.size(), (long long)side.gap_start.size(), (long long)names.size(), profiles_text_bytes};
  bool ok = fwrite(SOA_MAGIC, 1, 8, f) == 8 && fwrite(head, 8, 4, f) == 4;
  auto put = [&](const std::vector<long long> &v) {
    ok = ok && (v.empty() || fwrite(v.data(), 8, v.size(), f) == v.size());
  };
  put(side.start);
  put(side.end);
  put(side.length);
  put(side.gap_off);
  put(side.gap_start);
  put(side.gap_end);
  ok = ok && (names.empty() || fwrite(names.data(), 1, names.size(), f) == names.size());
  if(fclose(f) != 0 || !ok) {
    remove(path.c_str());
    return fail(PM_E_IO, "cannot write " + path);
  }
  return PM_OK;
}

// true when <dir>/profiles.soa was read into `side`; false (side untouched) when it is absent, stale or malformed
static bool read_side_soa(const std::string &dir, Side &side) {
  struct stat st_txt, st_soa;
  const std::string txt = dir + "/profiles", soa = dir + "/profiles.soa";
  if(stat(txt.c_str(), &st_txt) != 0 || stat(soa.c_str(), &st_soa) != 0) {
    return false;
  }
  if(st_soa.st_mtim.tv_sec < st_txt.st_mtim.tv_sec ||
     (st_soa.st_mtim.tv_sec == st_txt.st_mtim.tv_sec && st_soa.st_mtim.tv_nsec < st_txt.st_mtim.tv_nsec)) {
    return false; // the text file was rewritten after the side file
  }
  std::string blob;
  if(!read_whole_file(soa, blob) || blob.size() < 40 || memcmp(blob.data(), SOA_MAGIC, 8) != 0) {
    return false;
  }
  long long head[4];
  memcpy(head, blob.data() + 8, 32);
  const long long n = head[0], g = head[1], nb = head[2];
  if(n < 0 || g < 0 || nb < 0 || head[3] != (long long)st_txt.st_size) {
    return false;
  }
  // counts the file cannot hold would wrap `need` below: bound them by the file's size first
  const long long words = (long long)(blob.size() / 8);
  if(n > words || g > words || nb > (long long)blob.size()) {
    return false;
  }
  const size_t need = 40 + (size_t)(3 * n + (n + 1) + 2 * g) * 8 + (size_t)nb;
  if(blob.size() != need) {
    return false;
  }
  try {
  Side s;
  const char *p = blob.data() + 40;
  auto get = [&](std::vector<long long> &v, long long count) {
    v.resize((size_t)count);
    memcpy(v.data(), p, (size_t)count * 8);
    p += (size_t)count * 8;
  };
  get(s.start, n);
  get(s.end, n);
  get(s.length, n);
  get(s.gap_off, n + 1);
  get(s.gap_start, g);
  get(s.gap_end, g);
  if(s.gap_off[0] != 0 || s.gap_off[(size_t)n] != g) {
    return false;
  }
  for(long long r = 0; r < n; ++r) { // the offsets index the gap arrays on the device: a damaged file must not get that far
    if(s.gap_off[(size_t)r] > s.gap_off[(size_t)r + 1]) {
      return false;
    }
  }
  const char *e = blob.data() + blob.size();
  s.major.reserve((size_t)n);
  s.seq_name.reserve((size_t)n);
  for(long long r = 0; r < n; ++r) {
    const char *z = (const char *)memchr(p, 0, (size_t)(e - p));
    if(!z) {
      return false;
    }
    s.major.emplace_back(p, z);
    p = z + 1;
    z = (const char *)memchr(p, 0, (size_t)(e - p));
    if(!z) {
      return false;
    }
    s.seq_name.emplace_back(p, z);
    p = z + 1;
  }
  side = std::move(s);
  return true;
  }
  catch(const std::exception &) { // out of memory while copying: the text file is still there
    return false;
  }
}

int load_side(const std::string &dir, Side &side, const pm_translate_options_t &opt) {
  if(!opt.no_side_file && read_side_soa(dir, side)) {
    return PM_OK;
  }
  return parse_profiles(dir + "/profiles", side);
}

// m_translate.cc:188-207: rows grouped by sequence name, each group sorted by forward start.
void build_side_index(Side &side) {
  side.by_seq.clear();
  for(size_t r = 0; r < side.start.size(); ++r) {
    side.by_seq[side.seq_name[r]].push_back((int)r);
  }
  const std::vector<long long> &s = side.start, &e = side.end;
  for(std::map<std::string, std::vector<int> >::iterator it = side.by_seq.begin(); it != side.by_seq.end(); ++it) {
    std::sort(it->second.begin(), it->second.end(),
              [&](int a, int b) { return std::min(s[a], e[a]) < std::min(s[b], e[b]); });
  }
}

// ------------------------------------------------------------------ deltas

// m_delta.cc:14-68
static void split_offsets(const std::vector<long long> &offsets, DeltaTable &t) {
  size_t k = 0;
  long long column = 0;
  while(k < offsets.size()) {
    long long v = offsets[k];
    bool in_query = v > 0;
    long long first = column + (in_query ? v : -v);
    ++k;
    long long extra = 0;
    while(k < offsets.size() && (offsets[k] == 1 || offsets[k] == -1) && ((offsets[k] > 0) == in_query)) {
      ++extra;
      ++k;
    }
    column = first + extra;
    if(in_query) {
      t.qry_gap_start.push_back(first);
      t.qry_gap_end.push_back(column);
    }
    else {
      t.ref_gap_start.push_back(first);
      t.ref_gap_end.push_back(column);
    }
  }
}

int parse_delta_file(const std::string &path, DeltaTable &t) {
  std::string text;
  if(!read_whole_file(path, text)) {
    return fail(PM_E_PARSE, path + ": cannot read delta file (the reference throws Delta_stream_parse_error, m_delta.cc:72-92)");
  }
  return parse_delta_text(text, path, t);
}

bool read_stream(FILE *f, std::string &out) {
  std::string buf;
  char chunk[1 << 16];
  size_t n;
  while((n = fread(chunk, 1, sizeof chunk, f)) > 0) {
    buf.append(chunk, n);
  }
  out.swap(buf);
  return !ferror(f);
}

int parse_delta_text(const std::string &text, const std::string &path, DeltaTable &t) {
  if(t.ref_gap_off.empty()) {
    t.ref_gap_off.push_back(0);
    t.qry_gap_off.push_back(0);
  }
  Lines lines(text);
  const char *b, *e, *tb, *te;
  if(!lines.next(b, e)) {
    return fail(PM_E_PARSE, path + ": empty delta file");
  }
  const char *p = b;
  if(!next_token(p, e, tb, te) || !next_token(p, e, tb, te)) {
    return fail(PM_E_PARSE, path + ": first line needs two tokens");
  }
  if(!lines.next(b, e)) {
    return fail(PM_E_PARSE, path + ": missing stream type line");
  }
  std::string ref_name, qry_name;
  long long l1 = 0, l2 = 0; // header lengths in force (M_delta_stream::header_lengths_, m_delta.hh:77)
  std::vector<long long> offsets;
  while(lines.next(b, e)) {
    if(b < e && *b == '>') {
      p = b + 1;
      if(!next_token(p, e, tb, te)) {
        return fail(PM_E_PARSE, path + ": bad alignment header");
      }
      ref_name.assign(tb, te);
      if(!next_token(p, e, tb, te)) {
        return fail(PM_E_PARSE, path + ": bad alignment header");
      }
      qry_name.assign(tb, te);
      if(!next_i64(p, e, l1) || !next_i64(p, e, l2)) {
        return fail(PM_E_PARSE, path + ": bad alignment header lengths");
      }
      if(!lines.next(b, e)) {
        return fail(PM_E_PARSE, path + ": header without alignment line");
      }
    }
    p = b;
    long long v[7];
    for(int k = 0; k < 7; ++k) {
      if(!next_i64(p, e, v[k])) {
        return fail(PM_E_PARSE, path + ": alignment line needs 7 integers");
      }
    }
    offsets.clear();
    for(;;) {
      if(!lines.next(b, e)) {
        break;
      }
      if(e - b == 1 && *b == '0') {
        break;
      }
      p = b;
      long long o;
      if(!next_i64(p, e, o) || o < INT_MIN || o > INT_MAX) { // `int gap`, m_delta.cc:189
        return fail(PM_E_PARSE, path + ": bad offset line");
      }
      offsets.push_back(o);
    }
    t.ref_name.push_back(ref_name);
    t.qry_name.push_back(qry_name);
    t.ref_len.push_back(l1);
    t.qry_len.push_back(l2);
    t.ref_start.push_back(v[0]);
    t.ref_end.push_back(v[1]);
    t.qry_start.push_back(v[2]);
    t.qry_end.push_back(v[3]);
    split_offsets(offsets, t);
    t.ref_gap_off.push_back((long long)t.ref_gap_start.size());
    t.qry_gap_off.push_back((long long)t.qry_gap_start.size());
  }
  return PM_OK;
}

// ------------------------------------------------------------------ units

// std::lower_bound with "row ends before the entry starts" (m_translate.cc:175-178,682-695)
static size_t first_candidate(const Side &side, const std::vector<int> &rows, long long rs, long long re) {
  long long key = std::min(rs, re);
  std::vector<int>::const_iterator it =
      std::lower_bound(rows.begin(), rows.end(), key, [&](int r, long long v) { return std::max(side.start[r], side.end[r]) < v; });
  return (size_t)(it - rows.begin());
}

static inline bool ranges_overlap(long long as, long long ae, long long bs, long long be) { // m_range.hh:80-94
  long long s = std::max(std::min(as, ae), std::min(bs, be));
  long long e = std::min(std::max(as, ae), std::max(bs, be));
  return e - s >= 0;
}

// The rows an entry's loops visit (m_translate.cc:682-706): lefts [l0, l0 + nl) of `lr`, rights [r0, r0 + nr) of `rr`, every pair
// of them a unit.  false when one of the entry's sequences has no rows.
struct EntrySpan {
  const std::vector<int> *lr, *rr;
  size_t l0, nl, r0, nr;
};
static bool entry_span(const Side &left, const Side &right, const DeltaTable &t, size_t d, EntrySpan &sp) {
  std::map<std::string, std::vector<int> >::const_iterator li = left.by_seq.find(t.ref_name[d]);
  std::map<std::string, std::vector<int> >::const_iterator ri = right.by_seq.find(t.qry_name[d]);
  if(li == left.by_seq.end() || ri == right.by_seq.end()) {
    return false;
  }
  sp.lr = &li->second;
  sp.rr = &ri->second;
  sp.l0 = first_candidate(left, *sp.lr, t.ref_start[d], t.ref_end[d]);
  sp.r0 = first_candidate(right, *sp.rr, t.qry_start[d], t.qry_end[d]);
  sp.nl = sp.nr = 0;
  for(size_t l = sp.l0; l < sp.lr->size() && ranges_overlap(left.start[(*sp.lr)[l]], left.end[(*sp.lr)[l]], t.ref_start[d], t.ref_end[d]); ++l) {
    ++sp.nl;
  }
  for(size_t r = sp.r0; r < sp.rr->size() && ranges_overlap(right.start[(*sp.rr)[r]], right.end[(*sp.rr)[r]], t.qry_start[d], t.qry_end[d]); ++r) {
    ++sp.nr;
  }
  return true;
}

// m_translate.cc:666-707, for entries [first, t.size()): the units in entry order, left rows outer, right rows inner.  The entries
// are cut into slices, one per thread: a counting pass gives every slice its place in the list, a second pass fills it.
void enumerate_units(const Side &left, const Side &right, const DeltaTable &t, size_t first, UnitList &units) {
  const size_t n = t.ref_start.size() > first ? t.ref_start.size() - first : 0;
  unsigned hw = std::thread::hardware_concurrency();
  size_t n_threads = hw == 0 ? 1 : (hw > 8 ? 8 : hw);
  if(n < 4096) {
    n_threads = 1;
  }
  std::vector<size_t> count(n_threads + 1, 0);
  auto slice = [&](size_t k, size_t &lo, size_t &hi) {
    lo = first + n * k / n_threads;
    hi = first + n * (k + 1) / n_threads;
  };
  auto run = [&](const std::function<void(size_t)> &fn) {
    std::vector<JoinThread> th;
    for(size_t k = 1; k < n_threads; ++k) {
      th.emplace_back([&fn, k]() { fn(k); });
    }
    fn(0);
    for(size_t k = 0; k < th.size(); ++k) {
      th[k].join_and_rethrow();
    }
  };
  run([&](size_t k) {
    size_t lo, hi, c = 0;
    slice(k, lo, hi);
    EntrySpan sp;
    for(size_t d = lo; d < hi; ++d) {
      if(entry_span(left, right, t, d, sp)) {
        c += sp.nl * sp.nr;
      }
    }
    count[k + 1] = c;
  });
  for(size_t k = 0; k < n_threads; ++k) {
    count[k + 1] += count[k];
  }
  const size_t base = units.delta.size();
  units.delta.resize(base + count[n_threads]);
  units.left.resize(base + count[n_threads]);
  units.right.resize(base + count[n_threads]);
  run([&](size_t k) {
    size_t lo, hi, at = base + count[k];
    slice(k, lo, hi);
    EntrySpan sp;
    for(size_t d = lo; d < hi; ++d) {
      if(!entry_span(left, right, t, d, sp)) {
        continue;
      }
      for(size_t l = sp.l0; l < sp.l0 + sp.nl; ++l) {
        for(size_t r = sp.r0; r < sp.r0 + sp.nr; ++r) {
          units.delta[at] = (int)d;
          units.left[at] = (*sp.lr)[l];
          units.right[at] = (*sp.rr)[r];
          ++at;
        }
      }
    }
  });
}

// The same list made on the device (translate_job.hip: enum_count_kernel / enum_fill_kernel): what it needs from here is the row
// index flattened (sequences in the index's order, each one's rows as sorted there) and every entry's sequence on either side.
// Entries under one header share their names, so the look-up is repeated only when the name changes.
struct EnumTables {
  std::vector<int64_t> seq_off[2];
  std::vector<int32_t> seq_rows[2], entry_seq[2];
  EnumInput view() const {
    EnumInput en;
    for(int sd = 0; sd < 2; ++sd) {
      en.n_seq[sd] = (int64_t)seq_off[sd].size() - 1;
      en.seq_off[sd] = seq_off[sd].data();
      en.seq_rows[sd] = seq_rows[sd].data();
      en.entry_seq[sd] = entry_seq[sd].data();
    }
    return en;
  }
};
static void build_enum_tables(const Side &left, const Side &right, const DeltaTable &t, EnumTables &et) {
  const Side *side[2] = {&left, &right};
  const std::vector<std::string> *name[2] = {&t.ref_name, &t.qry_name};
  const size_t n = t.ref_start.size();
  auto one = [&](int sd) {
    std::map<std::string, int> id;
    et.seq_off[sd].assign(1, 0);
    et.seq_rows[sd].clear();
    et.seq_rows[sd].reserve(side[sd]->start.size());
    for(std::map<std::string, std::vector<int> >::const_iterator it = side[sd]->by_seq.begin(); it != side[sd]->by_seq.end(); ++it) {
      id[it->first] = (int)et.seq_off[sd].size() - 1;
      et.seq_rows[sd].insert(et.seq_rows[sd].end(), it->second.begin(), it->second.end());
      et.seq_off[sd].push_back((int64_t)et.seq_rows[sd].size());
    }
    et.entry_seq[sd].resize(n);
    const std::string *last = nullptr;
    int last_id = -1;
    for(size_t d = 0; d < n; ++d) {
      const std::string &nm = (*name[sd])[d];
      if(!last || nm != *last) {
        std::map<std::string, int>::const_iterator f = id.find(nm);
        last_id = f == id.end() ? -1 : f->second;
        last = &nm;
      }
      et.entry_seq[sd][d] = last_id;
    }
  };
  JoinThread other([&]() { one(1); }); // (what it throws -- an allocation failure -- reaches the caller's guard, not std::terminate)
  one(0);
  other.join_and_rethrow();
}

// ------------------------------------------------------------------ text out

// A host array whose elements are left uninitialised (plain malloc), for buffers a copy is about to fill.
template <typename T>
struct HostArray {
  T *p = nullptr;
  size_t n = 0;
  explicit HostArray(size_t count) : n(count) { p = (T *)malloc((count ? count : 1) * sizeof(T)); }
  ~HostArray() { free(p); }
  HostArray(const HostArray &) = delete;
  HostArray &operator=(const HostArray &) = delete;
  bool ok() const { return p != nullptr; }
  T *data() { return p; }
  char *bytes() { return (char *)p; }
  size_t size_bytes() const { return n * sizeof(T); }
};

// First touch of freshly allocated buffers, spread over a few threads (a page fault per 4 KiB is what a 100 MB
// device-to-host copy into new memory otherwise pays for, one at a time).
static void touch_pages(const std::vector<std::pair<char *, size_t> > &bufs) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t n_threads = hw == 0 ? 1 : (hw > 8 ? 8 : hw);
  size_t total = 0;
  for(size_t k = 0; k < bufs.size(); ++k) {
    total += bufs[k].second;
  }
  if(total < ((size_t)8 << 20)) {
    n_threads = 1;
  }
  auto work = [&](size_t t) {
    for(size_t k = 0; k < bufs.size(); ++k) {
      size_t pages = (bufs[k].second + 4095) / 4096;
      for(size_t pg = t; pg < pages; pg += n_threads) {
        ((volatile char *)bufs[k].first)[pg * 4096] = 0;
      }
    }
  };
  std::vector<JoinThread> th;
  for(size_t t = 1; t < n_threads; ++t) {
    th.emplace_back([&work, t]() { work(t); });
  }
  work(0);
  for(size_t k = 0; k < th.size(); ++k) {
    th[k].join_and_rethrow();
  }
}

static pm_rows_t rows_view(const Side &s) {
  pm_rows_t r;
  r.n = (int64_t)s.start.size();
  r.start = (const int64_t *)s.start.data();
  r.end = (const int64_t *)s.end.data();
  r.length = (const int64_t *)s.length.data();
  r.gap_off = (const int64_t *)s.gap_off.data();
  r.gap_start = (const int64_t *)s.gap_start.data();
  r.gap_end = (const int64_t *)s.gap_end.data();
  return r;
}

static pm_deltas_t deltas_view(const DeltaTable &t) {
  pm_deltas_t d;
  d.n = (int64_t)t.ref_start.size();
  d.ref_start = (const int64_t *)t.ref_start.data();
  d.ref_end = (const int64_t *)t.ref_end.data();
  d.qry_start = (const int64_t *)t.qry_start.data();
  d.qry_end = (const int64_t *)t.qry_end.data();
  d.ref_gap_off = (const int64_t *)t.ref_gap_off.data();
  d.ref_gap_start = (const int64_t *)t.ref_gap_start.data();
  d.ref_gap_end = (const int64_t *)t.ref_gap_end.data();
  d.qry_gap_off = (const int64_t *)t.qry_gap_off.data();
  d.qry_gap_start = (const int64_t *)t.qry_gap_start.data();
  d.qry_gap_end = (const int64_t *)t.qry_gap_end.data();
  return d;
}

// Parse both sides and every delta file of a job and list its work units (host only, no device needed).
// `parse_rc`/`parse_msg` keep a delta-file parse failure: the entries read before it are still in the table,
// as the reference would have translated them before throwing (m_translate.cc:722-728).
static double wall_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int load_workload(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, Workload &w,
                  const pm_translate_options_t &opt, bool list_units) {
  const bool timing = opt.timing != 0;
  const double t0 = wall_now();
  PM_TRY(load_side(left_dir, w.left, opt));
  PM_TRY(load_side(right_dir, w.right, opt));
  const double t1 = wall_now();
  parse_deltas(delta_paths, w);
  const double t2 = wall_now();
  if(list_units) {
    index_and_enumerate(w);
  }
  else {
    index_sides(w); // the units are listed on the device, by the job
  }
  if(timing) {
    fprintf(stderr, "[pm]   sides: %.4f s; delta files: %.4f s; %s: %.4f s\n", t1 - t0, t2 - t1, list_units ? "index + enumerate" : "index",
            wall_now() - t2);
  }
  return PM_OK;
}

// The delta files and the unit list of a workload whose two sides are already in place (read from disk, or handed over in
// memory by the make stage: pm_stage_files).
int load_deltas(const std::vector<std::string> &delta_paths, Workload &w) {
  parse_deltas(delta_paths, w);
  index_sides(w);
  return PM_OK;
}

// The delta files into w.table, in list order; stops at the first file that fails to parse (its entries up to the failure stay,
// as the reference translates them before it throws) and records the failure in w.parse_rc / w.parse_msg.  Needs no side.
void parse_deltas(const std::vector<std::string> &delta_paths, Workload &w) {
  w.table = DeltaTable();
  w.table.ref_gap_off.push_back(0);
  w.table.qry_gap_off.push_back(0);
  w.parse_rc = PM_OK;
  w.parse_msg.clear();
  const size_t n = delta_paths.size();
  unsigned hw = std::thread::hardware_concurrency();
  size_t n_threads = std::min<size_t>(n, hw == 0 ? 1 : (hw > 8 ? 8 : hw));
  if(n_threads <= 1) {
    for(size_t k = 0; k < n; ++k) {
      int rc = parse_delta_file(delta_paths[k], w.table);
      if(rc) {
        w.parse_rc = rc;
        w.parse_msg = pm_last_error();
        break;
      }
    }
    return;
  }
  // every file into a table of its own, a few files at a time; the tables are then joined in list order, up to and including
  // the first file that failed (what it held before the failure stays)
  std::vector<DeltaTable> part(n);
  std::vector<int> rc(n, PM_OK);
  std::vector<std::string> msg(n);
  std::atomic<size_t> next(0);
  auto work = [&]() {
    for(size_t k = next.fetch_add(1); k < n; k = next.fetch_add(1)) {
      part[k].ref_gap_off.push_back(0);
      part[k].qry_gap_off.push_back(0);
      rc[k] = parse_delta_file(delta_paths[k], part[k]);
      if(rc[k]) {
        msg[k] = pm_last_error(); // the error slot is per thread
      }
    }
  };
  std::vector<JoinThread> th;
  for(size_t k = 1; k < n_threads; ++k) {
    th.emplace_back([&work]() { work(); });
  }
  work();
  for(size_t k = 0; k < th.size(); ++k) {
    th[k].join_and_rethrow();
  }
  size_t upto = n;
  for(size_t k = 0; k < n; ++k) {
    if(rc[k]) {
      w.parse_rc = rc[k];
      w.parse_msg = msg[k];
      upto = k + 1;
      break;
    }
  }
  DeltaTable &t = w.table;
  size_t entries = 0, gr = 0, gq = 0;
  for(size_t k = 0; k < upto; ++k) {
    entries += part[k].ref_start.size();
    gr += part[k].ref_gap_start.size();
    gq += part[k].qry_gap_start.size();
  }
  for(std::vector<std::string> *v : {&t.ref_name, &t.qry_name}) {
    v->reserve(entries);
  }
  for(std::vector<long long> *v : {&t.ref_len, &t.qry_len, &t.ref_start, &t.ref_end, &t.qry_start, &t.qry_end}) {
    v->reserve(entries);
  }
  t.ref_gap_off.reserve(entries + 1);
  t.qry_gap_off.reserve(entries + 1);
  t.ref_gap_start.reserve(gr);
  t.ref_gap_end.reserve(gr);
  t.qry_gap_start.reserve(gq);
  t.qry_gap_end.reserve(gq);
  for(size_t k = 0; k < upto; ++k) {
    DeltaTable &p = part[k];
    auto cat = [](std::vector<long long> &dst, const std::vector<long long> &src) { dst.insert(dst.end(), src.begin(), src.end()); };
    std::move(p.ref_name.begin(), p.ref_name.end(), std::back_inserter(t.ref_name));
    std::move(p.qry_name.begin(), p.qry_name.end(), std::back_inserter(t.qry_name));
    cat(t.ref_len, p.ref_len);
    cat(t.qry_len, p.qry_len);
    cat(t.ref_start, p.ref_start);
    cat(t.ref_end, p.ref_end);
    cat(t.qry_start, p.qry_start);
    cat(t.qry_end, p.qry_end);
    const long long r0 = (long long)t.ref_gap_start.size(), q0 = (long long)t.qry_gap_start.size();
    for(size_t e = 1; e < p.ref_gap_off.size(); ++e) {
      t.ref_gap_off.push_back(r0 + p.ref_gap_off[e]);
      t.qry_gap_off.push_back(q0 + p.qry_gap_off[e]);
    }
    cat(t.ref_gap_start, p.ref_gap_start);
    cat(t.ref_gap_end, p.ref_gap_end);
    cat(t.qry_gap_start, p.qry_gap_start);
    cat(t.qry_gap_end, p.qry_gap_end);
    p = DeltaTable();
  }
}

// The per-sequence row index of both sides and the unit list of every parsed entry (m_translate.cc:666-707), in entry order.
void index_and_enumerate(Workload &w) {
  index_sides(w);
  enumerate_units(w.left, w.right, w.table, 0, w.units);
  w.units_listed = true;
}

// Only the index: the job lists the units itself (run_workload with w.units_listed false).
void index_sides(Workload &w) {
  JoinThread other([&]() { build_side_index(w.right); });
  build_side_index(w.left);
  other.join_and_rethrow();
  w.units = UnitList();
  w.units_listed = false;
}

void workload_views(const Workload &w, pm_rows_t *left, pm_rows_t *right, pm_deltas_t *deltas, pm_units_t *units) {
  if(left) {
    *left = rows_view(w.left);
  }
  if(right) {
    *right = rows_view(w.right);
  }
  if(deltas) {
    *deltas = deltas_view(w.table);
  }
  if(units) {
    units->n = (int64_t)w.units.delta.size();
    units->delta = w.units.delta.data();
    units->left = w.units.left.data();
    units->right = w.units.right.data();
  }
}

int translate_to_file(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, FILE *out,
                      const pm_translate_options_t &opt, int device) {
  // options.timing: phase times on stderr (where a whole job spends its wall time)
  const bool timing = opt.timing != 0;
  auto now = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  };
  double t0 = now();
  // the HIP runtime takes ~0.25 s to come up in a fresh process: let it do so while the host parses
  int init_rc = PM_OK;
  std::string init_msg;
  double init_s = 0;
  JoinThread init([&]() {
    const double i0 = wall_now();
    init_rc = use_device(device);
    if(init_rc) {
      init_msg = pm_last_error(); // the error slot is per thread
    }
    else {
      const double i1 = wall_now();
      (void)hipFree(nullptr); // the context
      const double i2 = wall_now();
      (void)warm_translate_kernels(); // and the kernels' code object
      warm_text_staging();
      if(timing) {
        fprintf(stderr, "[pm]   start-up: device %.4f s; context %.4f s; code object %.4f s\n", i1 - i0, i2 - i1, wall_now() - i2);
      }
    }
    init_s = wall_now() - i0;
  });
  Workload w;
  int load_rc = load_workload(left_dir, right_dir, delta_paths, w, opt, false);
  std::string load_msg = load_rc ? pm_last_error() : "";
  init.join_and_rethrow();
  if(init_rc) {
    return fail(init_rc, init_msg);
  }
  if(load_rc) {
    return fail(load_rc, load_msg);
  }
  double t1 = now();
  if(timing) {
    fprintf(stderr, "[pm] parse + index: %.3f s; HIP runtime start-up beside it: %.3f s\n", t1 - t0, init_s);
  }
  return run_workload(w, out, opt, device);
}

// Bytes in HBM to a sink, in pieces, through a few pinned staging buffers: while earlier pieces are written (a stream: by writer
// threads, straight to the file descriptor at each piece's place; a string: appended in order), later ones are on their way.
// No big host buffer: nothing to allocate, fault in or pin for a 64 MB text, and the copies run at the link's speed.  (Copying
// pieces into one big pageable buffer while earlier pieces of it are being written fails: the runtime pins the destination of a
// large copy in place, and write(2) from a range that is being pinned or unpinned returns EFAULT.)  The staging buffers are kept
// for the life of the process: a resident worker reuses them, a short-lived tool leaves with them.
struct TextStaging {
  static const int64_t piece = (int64_t)8 << 20;
  static const int n_buf = 4;
  char *p[n_buf] = {nullptr, nullptr, nullptr, nullptr};
  int reserve() {
    for(int k = 0; k < n_buf; ++k) {
      if(!p[k]) {
        PM_HIP(hipHostMalloc((void **)&p[k], (size_t)piece, hipHostMallocPortable));
      }
    }
    return PM_OK;
  }
};
static std::mutex g_staging_lock;
static std::vector<TextStaging> g_staging_free;
static int staging_acquire(TextStaging &st) {
  {
    std::lock_guard<std::mutex> hold(g_staging_lock);
    if(!g_staging_free.empty()) {
      st = g_staging_free.back();
      g_staging_free.pop_back();
      return PM_OK;
    }
  }
  return st.reserve();
}
static void staging_release(const TextStaging &st) {
  std::lock_guard<std::mutex> hold(g_staging_lock);
  g_staging_free.push_back(st);
}
void text_staging_trim() {
  std::vector<TextStaging> all;
  {
    std::lock_guard<std::mutex> hold(g_staging_lock);
    all.swap(g_staging_free);
  }
  for(size_t k = 0; k < all.size(); ++k) {
    for(int b = 0; b < TextStaging::n_buf; ++b) {
      if(all[k].p[b]) {
        (void)hipHostFree(all[k].p[b]);
      }
    }
  }
}
// start-up helper: a set allocated ahead of its use (pinning 32 MB takes a few milliseconds)
void warm_text_staging() {
  TextStaging st;
  if(st.reserve() == PM_OK) {
    staging_release(st);
  }
  else {
    (void)hipGetLastError();
  }
}

// `copied` runs as soon as the last byte has left the device (the caller may then free the device buffer while the last pieces
// are still being written).  The calling thread's current device must be the one `dev` lives on.
int device_bytes_to_sink(const char *dev, int64_t n_bytes, OutSink out, bool timing, const std::function<void()> &copied) {
  if(n_bytes <= 0) {
    copied();
    return PM_OK;
  }
  TextStaging stage;
  const double t0 = wall_now();
  {
    const int rc_stage = staging_acquire(stage);
    if(rc_stage) {
      copied(); // the caller's clean-up hangs on this call whatever happens
      return rc_stage;
    }
  }
  const int64_t piece = TextStaging::piece;
  const int n_buf = TextStaging::n_buf;
  const int64_t n_pieces = (n_bytes + piece - 1) / piece;
  // a stream sink: flush what the caller printed so far, then write the pieces at their places through the descriptor
  long long base = -1;
  int fd = -1;
  if(out.f) {
    if(fflush(out.f) == 0) {
      base = (long long)ftello(out.f);
      fd = fileno(out.f);
    }
    if(base < 0 || fd < 0) {
      base = -1; // not seekable (a pipe): fwrite, piece by piece, in order
    }
  }
  if(out.mem) {
    out.mem->reserve(out.mem->size() + (size_t)n_bytes);
  }
  // pieces at their places can be written by two threads side by side; a sink that only appends has one writer
  const int n_writers = base >= 0 && n_pieces > 2 ? 2 : 1;
  // ready = pieces copied so far; written[k] = piece k is out of its buffer (piece k lives in buffer k % n_buf)
  std::atomic<int64_t> ready(0);
  std::vector<std::atomic<int> > written((size_t)n_pieces);
  for(int64_t k = 0; k < n_pieces; ++k) {
    written[(size_t)k].store(0);
  }
  std::atomic<int> write_failed(0);
  std::vector<JoinThread> writers;
  for(int w = 0; w < n_writers; ++w) {
    writers.emplace_back([&, w]() {
      for(int64_t k = w; k < n_pieces; k += n_writers) {
        while(ready.load(std::memory_order_acquire) <= k) {
          if(ready.load(std::memory_order_acquire) < 0) {
            return;
          }
          std::this_thread::yield();
        }
        if(write_failed.load()) { // the output has failed already (a full disk): no further piece of it is written
          return;
        }
        const int64_t first = k * piece, n = std::min(piece, n_bytes - first);
        const char *src = stage.p[k % n_buf];
        if(base >= 0) {
          int64_t done = 0;
          while(done < n) {
            const ssize_t wr = pwrite(fd, src + done, (size_t)(n - done), (off_t)(base + first + done));
            if(wr <= 0) {
              write_failed.store(errno ? errno : EIO);
              break;
            }
            done += wr;
          }
        }
        else if(!out.write(src, (size_t)n)) {
          write_failed.store(EIO);
        }
        written[(size_t)k].store(1, std::memory_order_release);
      }
    });
  }
  int rc = PM_OK;
  for(int64_t k = 0; k < n_pieces && !rc; ++k) {
    while(k >= n_buf && !written[(size_t)(k - n_buf)].load(std::memory_order_acquire)) { // the buffer still holds piece k - n_buf
      if(write_failed.load()) {
        break;
      }
      std::this_thread::yield();
    }
    if(write_failed.load()) { // no copy into a buffer a writer may still hold, no further copies for an output that has failed
      ready.store(-1, std::memory_order_release);
      break;
    }
    const int64_t first = k * piece, n = std::min(piece, n_bytes - first);
    hipError_t e = hipMemcpy(stage.p[k % n_buf], dev + first, (size_t)n, hipMemcpyDeviceToHost);
    if(e != hipSuccess) {
      rc = fail(PM_E_HIP, std::string("hipMemcpy (device to host): ") + hipGetErrorString(e));
    }
    else {
      ready.store(k + 1, std::memory_order_release);
    }
  }
  if(rc) {
    ready.store(-1, std::memory_order_release);
  }
  copied(); // nothing below needs the device buffer
  const double t1 = wall_now();
  for(size_t w = 0; w < writers.size(); ++w) {
    writers[w].join();
  }
  for(size_t w = 0; w < writers.size(); ++w) {
    writers[w].join_and_rethrow();
  }
  if(!rc && write_failed.load()) {
    rc = fail(PM_E_IO, std::string("write failed: ") + strerror(write_failed.load()));
  }
  if(!rc && base >= 0 && fseeko(out.f, (off_t)(base + n_bytes), SEEK_SET) != 0) {
    rc = fail(PM_E_IO, std::string("write failed: fseeko: ") + strerror(errno));
  }
  staging_release(stage);
  if(timing) {
    fprintf(stderr, "[pm]   %lld bytes to the host, piece by piece beside the writing: %.4f s; the last pieces' writing: %.4f s\n",
            (long long)n_bytes, t1 - t0, wall_now() - t1);
  }
  return rc;
}

// The device part of a translate job and the text of its output: upload + prepare + sizing, one pass, fetch, format + write.
int run_workload(Workload &w, FILE *out, const pm_translate_options_t &opt, int device) {
  return run_tables(w.left, w.right, w.table, w.units_listed ? &w.units : nullptr, w.parse_rc, w.parse_msg, OutSink(out), opt, device);
}

int run_tables(const Side &left, const Side &right, const DeltaTable &table, const UnitList *units, int parse_rc, const std::string &parse_msg,
               OutSink out, const pm_translate_options_t &opt, int device) {
  const bool timing = opt.timing != 0;
  auto now = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  };
  double t1 = now();
  pm_job_t *job = nullptr;
  if(units ? !units->delta.empty() : !table.ref_start.empty()) {
    pm_rows_t lv = rows_view(left), rv = rows_view(right);
    pm_deltas_t dv = deltas_view(table);
    if(units) {
      pm_units_t uv;
      uv.n = (int64_t)units->delta.size();
      uv.delta = units->delta.data();
      uv.left = units->left.data();
      uv.right = units->right.data();
      PM_TRY(pm_job_create_opt(&lv, &rv, &dv, &uv, &opt, device, &job));
    }
    else { // the unit list is made on the device from the sides' row index (left.by_seq, right.by_seq)
      EnumTables et;
      build_enum_tables(left, right, table, et);
      const EnumInput en = et.view();
      PM_TRY(job_create_enumerating(&lv, &rv, &dv, &en, opt, device, &job));
      int64_t n_units = 0;
      (void)pm_job_units(job, &n_units, nullptr, nullptr, nullptr);
      if(n_units == 0) {
        pm_job_destroy(job);
        job = nullptr;
      }
    }
  }
  if(job) {
    double t2 = now();
    int rc = pm_job_run(job, nullptr);
    int64_t ne = 0, no = 0;
    if(!rc) {
      rc = pm_job_sizes(job, &ne, &no);
    }
    double t3 = now();
    // the writer's text is formatted on the device (pm_job_text): what comes back is the bytes of the output, not the
    // entries and offsets they are printed from (half the bytes, and no host formatting)
    std::vector<const char *> ln(left.major.size()), rn(right.major.size());
    for(size_t r = 0; r < ln.size(); ++r) {
      ln[r] = left.major[r].c_str();
    }
    for(size_t r = 0; r < rn.size(); ++r) {
      rn[r] = right.major[r].c_str();
    }
    int64_t n_bytes = 0, failed = -1;
    int32_t failed_status = 0;
    if(!rc) {
      rc = pm_job_text(job, ln.data(), rn.data(), &n_bytes, &failed, &failed_status);
    }
    int32_t failed_at[3] = {-1, -1, -1}; // the failing unit's entry and rows, for the message
    if(!rc && failed >= 0) {
      rc = job_unit_at(job, failed, failed_at);
    }
    double t4 = now();
    // the job's forty device buffers are freed (milliseconds of hipFree) while the last pieces of the text are written
    JoinThread reaper;
    auto reap = [&]() { reaper = JoinThread([job]() { pm_job_destroy(job); }); };
    if(!rc && n_bytes > 0) {
      rc = device_bytes_to_sink(job_text_device(job), n_bytes, out, timing, reap);
    }
    else {
      reap();
    }
    if(reaper.joinable()) {
      reaper.join();
    }
    else {
      pm_job_destroy(job);
    }
    if(rc) {
      return rc;
    }
    if(timing) {
      fprintf(stderr, "[pm] device init + upload + prepare + sizing: %.3f s; run: %.4f s; text on the device: %.4f s; fetch + write: %.3f s (%lld bytes)\n",
              t2 - t1, t3 - t2, t4 - t3, now() - t4, (long long)n_bytes);
    }
    if(failed >= 0) { // the reference died inside this unit: what it had printed is on the stream
      char msg[160];
      snprintf(msg, sizeof msg, "work unit %lld (delta entry %d, left row %d, right row %d) failed with status %d", (long long)failed,
               failed_at[0], failed_at[1], failed_at[2], (int)failed_status);
      return fail(failed_status == PM_ST_MALFORMED_INPUT ? PM_E_MALFORMED : PM_E_UNIT, msg);
    }
  }
  if(parse_rc) {
    return fail(parse_rc, parse_msg);
  }
  return PM_OK;
}

// ------------------------------------------------------------------ several devices (multi.hpp)

// The header names (first two tokens after '>') of a delta header line [b, e).
static void header_names(const char *b, const char *e, std::string &l, std::string &r) {
  const char *p = b + 1, *tb, *te;
  l.clear();
  r.clear();
  if(next_token(p, e, tb, te)) {
    l.assign(tb, te);
    if(next_token(p, e, tb, te)) {
      r.assign(tb, te);
    }
  }
}

// Shard texts, each printed by a writer that started with no header in force (m_delta_stream_writer.hh:55-60), joined into what
// ONE writer prints over the shards' entries in order: a shard's leading `>` line is dropped when the names in force at the end
// of what precedes it are the same (the writer prints a header only when the name pair changes, m_delta_stream_writer.hh:62-67,
// and keeps that state from one delta file to the next).
int merge_shard_texts(const std::vector<std::string> &parts, size_t n_parts, OutSink out) {
  std::string in_l, in_r; // names in force; the writer starts with ("", "")
  bool have = false;
  for(size_t k = 0; k < n_parts; ++k) {
    const std::string &t = parts[k];
    if(t.empty()) {
      continue;
    }
    size_t from = 0;
    if(t[0] == '>') {
      const char *e = (const char *)memchr(t.data(), '\n', t.size());
      const size_t line = e ? (size_t)(e - t.data()) + 1 : t.size();
      std::string l, r;
      header_names(t.data(), t.data() + line - (e ? 1 : 0), l, r);
      if(have && l == in_l && r == in_r) {
        from = line;
      }
    }
    if(!out.write(t.data() + from, t.size() - from)) {
      return fail(PM_E_IO, "write failed");
    }
    // the last header line of this shard is in force for the next
    size_t at = std::string::npos;
    for(size_t q = t.size(); q-- > 0;) {
      if(t[q] == '>' && (q == 0 || t[q - 1] == '\n')) {
        at = q;
        break;
      }
    }
    if(at != std::string::npos) {
      const char *e = (const char *)memchr(t.data() + at, '\n', t.size() - at);
      header_names(t.data() + at, e ? e : t.data() + t.size(), in_l, in_r);
      have = true;
    }
  }
  return PM_OK;
}

// Para_mugsy::translate over several devices: the delta-file list is cut into n_devices contiguous slices, every worker parses
// and translates its slice on its device against the two sides (loaded once, shared read-only) and formats its text in memory;
// the texts are joined in list order.  The reference dies at the first failure with everything before it on the stream: the
// join stops after the first failing shard (its partial output included) and the call returns that shard's error.
int translate_to_file_multi(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, FILE *out,
                            const pm_translate_options_t &opt, const int *devices, int n_devices) {
  Side left, right;
  {
    // the two sides side by side; the devices' runtimes come up meanwhile
    int rc_r = PM_OK;
    std::string msg_r;
    JoinThread other([&]() {
      rc_r = load_side(right_dir, right, opt);
      if(rc_r) {
        msg_r = pm_last_error();
      }
      else {
        build_side_index(right);
      }
    });
    JoinThread warm([&]() {
      for(int k = 0; k < n_devices; ++k) {
        if(use_device(devices[k]) == PM_OK) {
          (void)warm_translate_kernels();
        }
      }
    });
    int rc_l = load_side(left_dir, left, opt);
    if(!rc_l) {
      build_side_index(left);
    }
    std::string msg_l = rc_l ? pm_last_error() : "";
    other.join_and_rethrow();
    warm.join();
    if(rc_l) {
      return fail(rc_l, msg_l);
    }
    if(rc_r) {
      return fail(rc_r, msg_r);
    }
  }
  std::vector<std::string> text((size_t)n_devices);
  std::vector<int> rcs;
  int rc = run_on_devices(
      devices, n_devices,
      [&](int w, int device) {
        int64_t lo, hi;
        partition((int64_t)delta_paths.size(), n_devices, w, lo, hi);
        Workload mine; // its sides stay empty: the shared ones are used
        parse_deltas(std::vector<std::string>(delta_paths.begin() + lo, delta_paths.begin() + hi), mine);
        return run_tables(left, right, mine.table, nullptr, mine.parse_rc, mine.parse_msg, OutSink(&text[(size_t)w]), opt, device);
      },
      &rcs);
  std::string msg = rc ? pm_last_error() : "";
  size_t upto = (size_t)n_devices;
  for(size_t w = 0; w < rcs.size(); ++w) {
    if(rcs[w]) {
      upto = w + 1; // the failing shard's partial output is what the reference had written when it died
      break;
    }
  }
  PM_TRY(merge_shard_texts(text, upto, OutSink(out)));
  return rc ? fail(rc, msg) : PM_OK;
}

} // namespace pm

extern "C" int pm_translate_files(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                                  const char *out_path, int device) {
  return pm_translate_files_as(left_dir, right_dir, delta_paths, n_paths, out_path, left_dir, right_dir, &device, 1);
}

extern "C" int pm_translate_files_as(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                                     const char *out_path, const char *left_name, const char *right_name, const int *devices, int n_devices) {
  return pm_translate_files_opt(left_dir, right_dir, delta_paths, n_paths, out_path, left_name, right_name, devices, n_devices, nullptr);
}

extern "C" int pm_translate_files_opt(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                                      const char *out_path, const char *left_name, const char *right_name, const int *devices, int n_devices,
                                      const pm_translate_options_t *options) {
  if(options && options->coordinate_bits != 0 && options->coordinate_bits != 32 && options->coordinate_bits != 64) {
    return pm::fail(PM_E_INVALID, "pm_translate_files_opt: options.coordinate_bits is 0, 32 or 64");
  }
  const pm_translate_options_t opt = pm::translate_options(options); // resolved once, here: every thread of the job runs under this copy
  if(devices && n_devices > 1) {
    return pm::guarded("pm_translate_files_as", [&]() -> int {
      if(!left_dir || !right_dir || !out_path || !left_name || !right_name || n_paths < 0 || (n_paths > 0 && !delta_paths)) {
        return pm::fail(PM_E_INVALID, "pm_translate_files_as: null argument");
      }
      PM_TRY(pm::check_devices(devices, n_devices, "pm_translate_files_as"));
      std::vector<std::string> paths;
      for(int k = 0; k < n_paths; ++k) {
        if(!delta_paths[k]) {
          return pm::fail(PM_E_INVALID, "pm_translate_files_as: null path");
        }
        paths.push_back(delta_paths[k]);
      }
      FILE *f = fopen(out_path, "wb");
      if(!f) {
        return pm::fail(PM_E_IO, std::string("cannot open ") + out_path);
      }
      fprintf(f, "%s/sequences.fasta %s/sequences.fasta\nNUCMER\n", left_name, right_name); // m_translate_main.cc:35-39
      int rc = pm::translate_to_file_multi(left_dir, right_dir, paths, f, opt, devices, n_devices);
      if(fclose(f) != 0 && !rc) {
        rc = pm::fail(PM_E_IO, "close failed");
      }
      return rc;
    });
  }
  const int device = devices && n_devices == 1 ? devices[0] : 0;
  return pm::guarded("pm_translate_files", [&]() -> int {
  if(!left_dir || !right_dir || !out_path || !left_name || !right_name || n_paths < 0 || (n_paths > 0 && !delta_paths)) {
    return pm::fail(PM_E_INVALID, "pm_translate_files: null argument");
  }
  int rc = PM_OK;
  std::vector<std::string> paths;
  for(int k = 0; k < n_paths; ++k) {
    if(!delta_paths[k]) {
      return pm::fail(PM_E_INVALID, "pm_translate_files: null path");
    }
    paths.push_back(delta_paths[k]);
  }
  FILE *f = fopen(out_path, "wb");
  if(!f) {
    return pm::fail(PM_E_IO, std::string("cannot open ") + out_path);
  }
  // m_translate_main.cc:35-39 (the names as the caller's argv had them, whatever paths the files are opened by)
  fprintf(f, "%s/sequences.fasta %s/sequences.fasta\nNUCMER\n", left_name, right_name);
  rc = pm::translate_to_file(left_dir, right_dir, paths, f, opt, device);
  if(fclose(f) != 0 && !rc) {
    rc = pm::fail(PM_E_IO, "close failed");
  }
  return rc;
  });
}

extern "C" int pm_translate_files_multi(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                                        const char *out_path, const int *devices, int n_devices) {
  return pm::guarded("pm_translate_files_multi", [&]() -> int {
  if(!left_dir || !right_dir || !out_path || n_paths < 0 || (n_paths > 0 && !delta_paths)) {
    return pm::fail(PM_E_INVALID, "pm_translate_files_multi: null argument");
  }
  PM_TRY(pm::check_devices(devices, n_devices, "pm_translate_files_multi"));
  std::vector<std::string> paths;
  for(int k = 0; k < n_paths; ++k) {
    if(!delta_paths[k]) {
      return pm::fail(PM_E_INVALID, "pm_translate_files_multi: null path");
    }
    paths.push_back(delta_paths[k]);
  }
  FILE *f = fopen(out_path, "wb");
  if(!f) {
    return pm::fail(PM_E_IO, std::string("cannot open ") + out_path);
  }
  fprintf(f, "%s/sequences.fasta %s/sequences.fasta\nNUCMER\n", left_dir, right_dir); // m_translate_main.cc:35-39
  int rc = pm::translate_to_file_multi(left_dir, right_dir, paths, f, pm::translate_options(nullptr), devices, n_devices);
  if(fclose(f) != 0 && !rc) {
    rc = pm::fail(PM_E_IO, "close failed");
  }
  return rc;
  });
}

// Host only: the outputs of m_translate runs over consecutive slices of one delta-file list (each a complete file: the two header
// lines of m_translate_main.cc:35-39, then the entries) joined into what one run over the whole list prints.
extern "C" int pm_delta_join_files(const char *const *part_paths, int n_parts, const char *out_path) {
  return pm::guarded("pm_delta_join_files", [&]() -> int {
  if(n_parts < 0 || (n_parts > 0 && !part_paths) || !out_path) {
    return pm::fail(PM_E_INVALID, "pm_delta_join_files: null argument");
  }
  std::vector<std::string> body((size_t)n_parts);
  std::string head;
  for(int k = 0; k < n_parts; ++k) {
    std::string text;
    if(!part_paths[k] || !pm::read_whole_file(part_paths[k], text)) {
      return pm::fail(PM_E_IO, std::string("cannot read ") + (part_paths[k] ? part_paths[k] : "(null)"));
    }
    size_t a = text.find('\n');
    size_t b = a == std::string::npos ? a : text.find('\n', a + 1);
    if(b == std::string::npos) {
      return pm::fail(PM_E_PARSE, std::string(part_paths[k]) + ": not an m_translate output (two header lines expected)");
    }
    if(k == 0) {
      head = text.substr(0, b + 1);
    }
    body[(size_t)k] = text.substr(b + 1);
  }
  FILE *f = fopen(out_path, "wb");
  if(!f) {
    return pm::fail(PM_E_IO, std::string("cannot open ") + out_path);
  }
  int rc = PM_OK;
  if(!head.empty() && fwrite(head.data(), 1, head.size(), f) != head.size()) {
    rc = pm::fail(PM_E_IO, "write failed");
  }
  if(!rc) {
    rc = pm::merge_shard_texts(body, body.size(), pm::OutSink(f));
  }
  if(fclose(f) != 0 && !rc) {
    rc = pm::fail(PM_E_IO, "close failed");
  }
  return rc;
  });
}

extern "C" int pm_partition_weighted(const int64_t *weights, int64_t n_items, int n_parts, int64_t *cuts) {
  return pm::guarded("pm_partition_weighted", [&]() -> int {
    if(n_items < 0 || n_parts < 1 || !cuts || (n_items > 0 && !weights)) {
      return pm::fail(PM_E_INVALID, "pm_partition_weighted: bad argument");
    }
    for(int64_t k = 0; k < n_items; ++k) {
      if(weights[k] < 0) {
        return pm::fail(PM_E_INVALID, "pm_partition_weighted: negative weight");
      }
    }
    std::vector<int64_t> c;
    pm::partition_weighted(weights, n_items, n_parts, c);
    for(int k = 0; k <= n_parts; ++k) {
      cuts[k] = c[(size_t)k];
    }
    return PM_OK;
  });
}

extern "C" int pm_partition(int64_t n_items, int n_parts, int part, int64_t *lo, int64_t *hi) {
  if(n_items < 0 || n_parts < 1 || part < 0 || part >= n_parts || !lo || !hi) {
    return pm::fail(PM_E_INVALID, "pm_partition: bad argument");
  }
  pm::partition(n_items, n_parts, part, *lo, *hi);
  return PM_OK;
}

struct pm_workload {
  pm::Workload w;
};

extern "C" int pm_workload_load(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                                pm_workload_t **out) {
  return pm::guarded("pm_workload_load", [&]() -> int {
  if(!left_dir || !right_dir || !out || n_paths < 0 || (n_paths > 0 && !delta_paths)) {
    return pm::fail(PM_E_INVALID, "pm_workload_load: null argument");
  }
  *out = nullptr;
  std::vector<std::string> paths;
  for(int k = 0; k < n_paths; ++k) {
    if(!delta_paths[k]) {
      return pm::fail(PM_E_INVALID, "pm_workload_load: null path");
    }
    paths.push_back(delta_paths[k]);
  }
  pm_workload *h = new pm_workload();
  int rc = pm::load_workload(left_dir, right_dir, paths, h->w, pm::translate_options(nullptr), true);
  if(rc) {
    delete h;
    return rc;
  }
  *out = h;
  if(h->w.parse_rc) {
    return pm::fail(h->w.parse_rc, h->w.parse_msg); // handle stays valid: the entries read before the failure are in it
  }
  return PM_OK;
  });
}

extern "C" int pm_workload_tables(pm_workload_t *h, pm_rows_t *left, pm_rows_t *right, pm_deltas_t *deltas, pm_units_t *units) {
  if(!h) {
    return pm::fail(PM_E_INVALID, "pm_workload_tables: null workload");
  }
  pm::workload_views(h->w, left, right, deltas, units);
  return PM_OK;
}

extern "C" int pm_workload_row_name(pm_workload_t *h, int side, int64_t row, const char **major_name, const char **seq_name) {
  if(!h || (side != 0 && side != 1)) {
    return pm::fail(PM_E_INVALID, "pm_workload_row_name: bad argument");
  }
  const pm::Side &s = side ? h->w.right : h->w.left;
  if(row < 0 || row >= (int64_t)s.major.size()) {
    return pm::fail(PM_E_INVALID, "pm_workload_row_name: row out of range");
  }
  if(major_name) {
    *major_name = s.major[(size_t)row].c_str();
  }
  if(seq_name) {
    *seq_name = s.seq_name[(size_t)row].c_str();
  }
  return PM_OK;
}

extern "C" int pm_job_create_from_workload(pm_workload_t *h, int device, pm_job_t **out) {
  return pm_job_create_from_workload_opt(h, nullptr, device, out);
}

extern "C" int pm_job_create_from_workload_opt(pm_workload_t *h, const pm_translate_options_t *options, int device, pm_job_t **out) {
  return pm::guarded("pm_job_create_from_workload", [&]() -> int {
    if(!h || !out) {
      return pm::fail(PM_E_INVALID, "pm_job_create_from_workload: null argument");
    }
    *out = nullptr;
    pm_rows_t lv = pm::rows_view(h->w.left), rv = pm::rows_view(h->w.right);
    pm_deltas_t dv = pm::deltas_view(h->w.table);
    pm::EnumTables et;
    pm::build_enum_tables(h->w.left, h->w.right, h->w.table, et);
    const pm::EnumInput en = et.view();
    return pm::job_create_enumerating(&lv, &rv, &dv, &en, pm::translate_options(options), device, out);
  });
}

extern "C" void pm_workload_destroy(pm_workload_t *h) { delete h; }
