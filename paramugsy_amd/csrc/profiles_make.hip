// profiles_make.hip -- `mugsy_profiles make`: MAF -> <out_dir>/profiles + <out_dir>/sequences.fasta  (SURVEY.md 8f.1).
// The producer of the translate path's input format.  Reference (OCaml, cannot be built or run in this image, so
// this is RESTATED FROM SOURCE, NOT EXECUTED -- parity for it is pinned only by hand-derived fixtures and a Python
// transcription of the same source, tests/test_profiles_make_*.py):
//   lib/profiles/m_profile_stream.ml:16-74   MAF lines -> row profiles, block names "%s.%s_%04d"
//   lib/profiles/m_profile.ml:29-47           gaps_of_text: 1-based inclusive runs of '-'
//   lib/profiles/m_profile.ml:122-135         record layout of the `profiles` file
//   lib/profiles/m_make.ml:15-62              combine_text fold -> per-block consensus, FASTA layout
//   lib/profiles/m_range.ml:60-65             of_maf
// Text parsing and printing stay on the host; the two byte kernels (gap runs of every row, consensus of every
// block) run on the GPU: one thread per text byte / per block column, coalesced along the row text.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>

#include <rocprim/rocprim.hpp>

#include <string>
#include <thread>
#include <vector>

#include "pm_internal.hpp"
#include "translate_host.hpp"

namespace pm {

typedef long long i64;

// start[i] = 1 when byte i opens a run of '-' within its row, stop[i] = 1 when it closes one; col[i] = the byte's column in its
// row (the row is found by a search in row_off: rows are a few kilobytes, the table a few thousand entries).
__global__ void gap_flags_kernel(i64 n, const unsigned char *text, const i64 *row_off, int n_rows, int *start, int *stop, int *col) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) {
    return;
  }
  int lo = 0, hi = n_rows; // the last r with row_off[r] <= i (rows without bytes repeat their neighbour's offset)
  while(hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if(row_off[mid] <= i) {
      lo = mid;
    }
    else {
      hi = mid;
    }
  }
  const bool first = i == row_off[lo], last = i + 1 == row_off[lo + 1];
  bool g = text[i] == '-';
  bool prev = !first && text[i - 1] == '-';
  bool next = !last && text[i + 1] == '-';
  start[i] = g && !prev;
  stop[i] = g && !next;
  col[i] = (int)(i - row_off[lo]);
}

// gap_off[r] = runs opened before row r's first byte (start_scan there), gap_off[n_rows] = all of them
__global__ void gap_offsets_kernel(int n_rows, i64 n, const i64 *row_off, const int *start, const int *start_scan, i64 *gap_off) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if(r > n_rows) {
    return;
  }
  const i64 total = (i64)start_scan[n - 1] + start[n - 1];
  gap_off[r] = r < n_rows && row_off[r] < n ? (i64)start_scan[row_off[r]] : total;
}

// k-th opening and k-th closing byte of the whole buffer belong to the same run (runs never cross a row).
__global__ void gap_runs_kernel(i64 n, const int *start, const int *stop, const int *start_scan, const int *stop_scan, const int *col,
                                i64 *gap_start, i64 *gap_end) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) {
    return;
  }
  if(start[i]) {
    gap_start[start_scan[i]] = col[i] + 1; // 1-based column
  }
  if(stop[i]) {
    gap_end[stop_scan[i]] = col[i] + 1;
  }
}

// m_make.ml:15-28 folded over the rows of a block (:35-45): one thread per block column.
__global__ void consensus_kernel(i64 n_cols, int n_blocks, const i64 *cons_off, const int *block_first_row, const i64 *row_off,
                                 const unsigned char *text, unsigned char *cons) {
  i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n_cols) {
    return;
  }
  int lo = 0, hi = n_blocks; // block of this column: last b with cons_off[b] <= g
  while(hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if(cons_off[mid] <= g) {
      lo = mid;
    }
    else {
      hi = mid;
    }
  }
  i64 c = g - cons_off[lo];
  int r0 = block_first_row[lo], r1 = block_first_row[lo + 1];
  unsigned char x = text[row_off[r0] + c];
  for(int r = r0 + 1; r < r1; ++r) {
    unsigned char b = text[row_off[r] + c];
    if(x == b) {
    }
    else if(x != '-' && b != '-') {
      x = 'N';
    }
    else if(x == '-') {
      x = b;
    }
  }
  cons[g] = x;
}

struct MakeRow {
  std::string seq_name;
  long long start, end; // p_range
  long long src_size;
  int block;
  int minor;
};

// m_profile_stream.ml:16-74
static int parse_maf_for_make(const std::string &maf, std::vector<MakeRow> &rows, std::string &text, std::vector<i64> &row_off,
                              std::vector<int> &block_first_row) {
  const char *p = maf.data(), *end = p + maf.size();
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
  row_off.push_back(0);
  int block = 0;
  const char *b, *e;
  for(;;) {
    bool found = false; // drop_until_score, :23-32
    while(getline(b, e)) {
      if(e - b >= 8 && memcmp(b, "a score=", 8) == 0) {
        found = true;
        break;
      }
    }
    if(!found) {
      break;
    }
    block_first_row.push_back((int)rows.size());
    int idx = 0;
    for(;;) { // stream_profiles, :35-58
      if(!getline(b, e)) {
        if(idx == 0) {
          return fail(PM_E_PARSE, "make: Expected alignment, did not get (m_profile_stream.ml:55)");
        }
        break;
      }
      if(e == b) {
        break;
      }
      if(e - b >= 2 && b[0] == 's' && b[1] == ' ') {
        std::vector<std::pair<const char *, const char *> > tok; // split_maf, :16-21
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
          return fail(PM_E_PARSE, "make: Unknown maf line (m_profile_stream.ml:21)");
        }
        auto num = [&](size_t k, long long &v) {
          std::string sx(tok[k].first, tok[k].second);
          char *endp = nullptr;
          v = strtoll(sx.c_str(), &endp, 10);
          return !sx.empty() && endp && *endp == 0;
        };
        MakeRow r;
        long long st, sz;
        if(!num(2, st) || !num(3, sz) || !num(5, r.src_size)) {
          return fail(PM_E_PARSE, "make: int_of_string failure on an `s` line");
        }
        size_t dl = (size_t)(tok[4].second - tok[4].first);
        if(dl != 1 || (*tok[4].first != '+' && *tok[4].first != '-')) {
          return fail(PM_E_PARSE, "make: Invalid direction (m_profile_stream.ml:14)");
        }
        if(*tok[4].first == '+') { // m_range.ml:60-65
          r.start = st + 1;
          r.end = st + sz;
        }
        else {
          r.start = r.src_size - st;
          r.end = r.src_size - st - (sz - 1);
        }
        r.seq_name.assign(tok[1].first, tok[1].second);
        r.block = block;
        r.minor = idx++;
        rows.push_back(r);
        text.append(tok[6].first, tok[6].second);
        row_off.push_back((i64)text.size());
      }
      else if(b[0] == '#') {
        continue;
      }
      else {
        return fail(PM_E_PARSE, "make: Unknown line (m_profile_stream.ml:53)");
      }
    }
    ++block;
  }
  block_first_row.push_back((int)rows.size());
  return PM_OK;
}

} // namespace pm

namespace pm {

// `mugsy_profiles make` (below), handing the rows it wrote back as the flat arrays the translate stage works on (side_out may
// be null); also writes <out_dir>/profiles.soa, the binary form of the same rows (translate_host.cc).
int make_profiles(const char *in_maf, const char *out_dir, const char *basename, int device, Side *side_out) {
  int rc = use_device(device);
  if(rc) {
    return rc;
  }
  const bool timing = pm::timing_on();
  auto wall = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  };
  double lap_t = wall();
  auto lap = [&](const char *what) {
    if(timing) {
      const double t = wall();
      fprintf(stderr, "[pm]   make %s: %-36s %.4f s\n", basename, what, t - lap_t);
      lap_t = t;
    }
  };
  std::string maf;
  {
    FILE *f = fopen(in_maf, "rb");
    if(!f) {
      return fail(PM_E_IO, std::string("cannot open ") + in_maf);
    }
    read_stream(f, maf);
    fclose(f);
  }
  lap("file read");
  std::vector<MakeRow> rows;
  std::string text;
  std::vector<i64> row_off;
  std::vector<int> block_first_row;
  PM_TRY(parse_maf_for_make(maf, rows, text, row_off, block_first_row));
  lap("lines parsed");
  int n_rows = (int)rows.size(), n_blocks = (int)block_first_row.size() - 1;
  i64 n = (i64)text.size();
  // every row of a block must have the block's column count (assert at m_make.ml:16)
  std::vector<i64> cons_off((size_t)n_blocks + 1, 0);
  for(int b = 0; b < n_blocks; ++b) {
    int r0 = block_first_row[b], r1 = block_first_row[b + 1];
    i64 cols = r1 > r0 ? row_off[(size_t)r0 + 1] - row_off[r0] : 0;
    for(int r = r0; r < r1; ++r) {
      if(row_off[(size_t)r + 1] - row_off[r] != cols) {
        return fail(PM_E_PARSE, "make: rows of one block differ in length (assert, m_make.ml:16)");
      }
    }
    cons_off[(size_t)b + 1] = cons_off[b] + cols;
  }
  std::vector<i64> gap_off((size_t)n_rows + 1, 0), gap_start, gap_end;
  std::string cons((size_t)cons_off[n_blocks], '\0');
  if(n > 0) {
    lap("(nothing per byte on the host)");
    // the per-byte buffers come from the pool of kept device buffers (every one of them is written in full before it is read)
    DevBuf d_roff0, d_tmp, d_gs, d_ge, d_goff;
    PooledBuf d_text, d_col, d_start, d_stop, d_sscan, d_escan;
    PM_TRY(d_text.alloc((size_t)n, device));
    PM_HIP(hipMemcpy(d_text.p, text.data(), (size_t)n, hipMemcpyHostToDevice));
    PM_TRY(d_roff0.upload(row_off.data(), ((size_t)n_rows + 1) * 8, nullptr));
    PM_TRY(d_col.alloc((size_t)n * 4, device));
    PM_TRY(d_start.alloc((size_t)n * 4, device));
    PM_TRY(d_stop.alloc((size_t)n * 4, device));
    PM_TRY(d_sscan.alloc((size_t)n * 4, device));
    PM_TRY(d_escan.alloc((size_t)n * 4, device));
    PM_TRY(d_goff.alloc(((size_t)n_rows + 1) * 8));
    unsigned blocks = (unsigned)((n + 255) / 256);
    gap_flags_kernel<<<blocks, 256>>>(n, (const unsigned char *)d_text.p, (const i64 *)d_roff0.p, n_rows, (int *)d_start.p, (int *)d_stop.p,
                                      (int *)d_col.p);
    PM_HIP(hipGetLastError());
    size_t bytes = 0;
    PM_HIP(rocprim::exclusive_scan(nullptr, bytes, (int *)d_start.p, (int *)d_sscan.p, 0, (size_t)n, rocprim::plus<int>()));
    PM_TRY(d_tmp.alloc(bytes ? bytes : 8));
    PM_HIP(rocprim::exclusive_scan(d_tmp.p, bytes, (int *)d_start.p, (int *)d_sscan.p, 0, (size_t)n, rocprim::plus<int>()));
    PM_HIP(rocprim::exclusive_scan(d_tmp.p, bytes, (int *)d_stop.p, (int *)d_escan.p, 0, (size_t)n, rocprim::plus<int>()));
    gap_offsets_kernel<<<(unsigned)((n_rows + 256) / 256), 256>>>(n_rows, n, (const i64 *)d_roff0.p, (const int *)d_start.p, (const int *)d_sscan.p,
                                                                  (i64 *)d_goff.p);
    PM_HIP(hipGetLastError());
    PM_HIP(hipMemcpy(gap_off.data(), d_goff.p, ((size_t)n_rows + 1) * 8, hipMemcpyDeviceToHost));
    i64 total = gap_off[n_rows];
    gap_start.resize((size_t)total);
    gap_end.resize((size_t)total);
    if(total > 0) {
      PM_TRY(d_gs.alloc((size_t)total * 8));
      PM_TRY(d_ge.alloc((size_t)total * 8));
      gap_runs_kernel<<<blocks, 256>>>(n, (const int *)d_start.p, (const int *)d_stop.p, (const int *)d_sscan.p, (const int *)d_escan.p,
                                       (const int *)d_col.p, (i64 *)d_gs.p, (i64 *)d_ge.p);
      PM_HIP(hipGetLastError());
      PM_HIP(hipMemcpy(gap_start.data(), d_gs.p, (size_t)total * 8, hipMemcpyDeviceToHost));
      PM_HIP(hipMemcpy(gap_end.data(), d_ge.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    }
    if(cons_off[n_blocks] > 0) {
      DevBuf d_coff, d_bfr, d_roff, d_cons;
      PM_TRY(d_coff.upload(cons_off.data(), ((size_t)n_blocks + 1) * 8, nullptr));
      PM_TRY(d_bfr.upload(block_first_row.data(), ((size_t)n_blocks + 1) * 4, nullptr));
      PM_TRY(d_roff.upload(row_off.data(), ((size_t)n_rows + 1) * 8, nullptr));
      PM_TRY(d_cons.alloc((size_t)cons_off[n_blocks]));
      i64 nc = cons_off[n_blocks];
      consensus_kernel<<<(unsigned)((nc + 255) / 256), 256>>>(nc, n_blocks, (const i64 *)d_coff.p, (const int *)d_bfr.p, (const i64 *)d_roff.p,
                                                             (const unsigned char *)d_text.p, (unsigned char *)d_cons.p);
      PM_HIP(hipGetLastError());
      PM_HIP(hipMemcpy(&cons[0], d_cons.p, (size_t)nc, hipMemcpyDeviceToHost));
    }
  }
  lap("gap runs + consensus (device)");
  // m_make.ml:48-62: both files are created even when the MAF holds no block
  std::string dir(out_dir);
  FILE *fp = fopen((dir + "/profiles").c_str(), "wb");
  FILE *ff = fopen((dir + "/sequences.fasta").c_str(), "wb");
  if(!fp || !ff) {
    if(fp) fclose(fp);
    if(ff) fclose(ff);
    return fail(PM_E_IO, "cannot create output files in " + dir + " (the directory must exist)");
  }
  std::string buf;
  buf.reserve((size_t)2 << 20);
  auto put = [&buf](long long v) { // decimal, appended in place (a million numbers: no temporaries)
    char tmp[24];
    int k = 0;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do {
      tmp[k++] = (char)('0' + u % 10);
      u /= 10;
    } while(u);
    if(v < 0) {
      tmp[k++] = '-';
    }
    while(k) {
      buf.push_back(tmp[--k]);
    }
  };
  char major[1024];
  int major_block = -1;
  size_t major_len = 0;
  for(int r = 0; r < n_rows; ++r) { // m_profile.ml:122-135
    if(rows[r].block != major_block) {
      major_block = rows[r].block;
      major_len = (size_t)snprintf(major, sizeof major, "%s.%s_%04d", basename, basename, rows[r].block); // m_profile_stream.ml:65
    }
    buf.append(major, major_len);
    buf += ' ';
    put(rows[r].minor);
    buf += ' ';
    buf += rows[r].seq_name;
    buf += ' ';
    put(rows[r].start);
    buf += ' ';
    put(rows[r].end);
    buf += ' ';
    put(row_off[(size_t)r + 1] - row_off[r]);
    buf += ' ';
    put(rows[r].src_size);
    buf += '\n';
    for(i64 g = gap_off[r]; g < gap_off[(size_t)r + 1]; ++g) {
      put(gap_start[(size_t)g]);
      buf += ' ';
      put(gap_end[(size_t)g]);
      buf += '\n';
    }
    buf += "0\n";
    buf.append(text, (size_t)row_off[r], (size_t)(row_off[(size_t)r + 1] - row_off[r]));
    buf += '\n';
    if(buf.size() > (1 << 20)) {
      fwrite(buf.data(), 1, buf.size(), fp);
      buf.clear();
    }
  }
  fwrite(buf.data(), 1, buf.size(), fp);
  buf.clear();
  for(int b = 0; b < n_blocks; ++b) { // m_make.ml:35-45
    if(block_first_row[(size_t)b + 1] == block_first_row[b]) {
      continue;
    }
    snprintf(major, sizeof major, "%s.%s_%04d", basename, basename, b);
    buf += '>';
    buf += major;
    buf += '\n';
    buf.append(cons, (size_t)cons_off[b], (size_t)(cons_off[(size_t)b + 1] - cons_off[b]));
    buf += "\n\n";
  }
  fwrite(buf.data(), 1, buf.size(), ff);
  const long long text_bytes = ftell(fp);
  const bool wrote = fclose(fp) == 0;
  if(fclose(ff) != 0 || !wrote || text_bytes < 0) {
    return fail(PM_E_IO, "cannot write the output files in " + dir);
  }
  lap("profiles + fasta formatted, written");
  // the same rows as flat arrays: what parse_profiles would read back from the file just written
  Side side;
  side.gap_off.assign(gap_off.begin(), gap_off.end());
  side.gap_start.assign(gap_start.begin(), gap_start.end());
  side.gap_end.assign(gap_end.begin(), gap_end.end());
  for(int r = 0; r < n_rows; ++r) {
    snprintf(major, sizeof major, "%s.%s_%04d", basename, basename, rows[r].block);
    side.major.push_back(major);
    side.seq_name.push_back(rows[r].seq_name);
    side.start.push_back(rows[r].start);
    side.end.push_back(rows[r].end);
    side.length.push_back(row_off[(size_t)r + 1] - row_off[r]);
  }
  PM_TRY(write_side_soa(dir, side, text_bytes));
  lap("side file written");
  if(side_out) {
    *side_out = std::move(side);
  }
  return PM_OK;
}

} // namespace pm

using namespace pm;

extern "C" int pm_profiles_make(const char *in_maf, const char *out_dir, const char *basename, int device) {
  return pm::guarded("pm_profiles_make", [&]() -> int {
  if(!in_maf || !out_dir || !basename) {
    return fail(PM_E_INVALID, "pm_profiles_make: null argument");
  }
  return make_profiles(in_maf, out_dir, basename, device, nullptr);
  });
}

// make(left) + make(right) + translate in one process and one HIP context (lib/base/mugsy_profiles_task.ml:40-58 runs them as
// three processes): the two makes and the parsing of the delta files run side by side on host threads, and the rows the makes
// produce go to the translate stage in memory instead of through the `profiles` text.  Same files written, same bytes.
extern "C" int pm_stage_files(const char *left_maf, const char *left_dir, const char *left_basename, const char *right_maf,
                              const char *right_dir, const char *right_basename, const char *const *delta_paths, int n_paths,
                              const char *out_delta, int device) {
  return pm::guarded("pm_stage_files", [&]() -> int {
  if(!left_maf || !left_dir || !left_basename || !right_maf || !right_dir || !right_basename || !out_delta || n_paths < 0 ||
     (n_paths > 0 && !delta_paths)) {
    return fail(PM_E_INVALID, "pm_stage_files: null argument");
  }
  PM_TRY(use_device(device));
  std::vector<std::string> paths;
  for(int k = 0; k < n_paths; ++k) {
    if(!delta_paths[k]) {
      return fail(PM_E_INVALID, "pm_stage_files: null path");
    }
    paths.push_back(delta_paths[k]);
  }
  const bool timing = pm::timing_on();
  auto wall = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  };
  double lap_t = wall();
  auto lap = [&](const char *what) {
    if(timing) {
      const double t = wall();
      fprintf(stderr, "[pm] stage: %-40s %.4f s\n", what, t - lap_t);
      lap_t = t;
    }
  };
  Workload w;
  int rc_l = PM_OK, rc_r = PM_OK;
  std::string msg_l, msg_r;
  JoinThread tl([&]() {
    rc_l = make_profiles(left_maf, left_dir, left_basename, device, &w.left);
    if(rc_l) {
      msg_l = pm_last_error(); // the error slot is per thread
    }
  });
  JoinThread tr([&]() {
    rc_r = make_profiles(right_maf, right_dir, right_basename, device, &w.right);
    if(rc_r) {
      msg_r = pm_last_error();
    }
  });
  JoinThread td([&]() { parse_deltas(paths, w); }); // touches w.table and w.parse_* only
  tl.join_and_rethrow();
  tr.join_and_rethrow();
  td.join_and_rethrow();
  if(rc_l) {
    return fail(rc_l, msg_l);
  }
  if(rc_r) {
    return fail(rc_r, msg_r);
  }
  lap("two makes beside the delta files");
  index_sides(w); // the job lists the units on the device
  lap("rows indexed");
  FILE *f = fopen(out_delta, "wb");
  if(!f) {
    return fail(PM_E_IO, std::string("cannot open ") + out_delta);
  }
  fprintf(f, "%s/sequences.fasta %s/sequences.fasta\nNUCMER\n", left_dir, right_dir); // m_translate_main.cc:35-39
  int rc = run_workload(w, f, translate_options(nullptr), device);
  lap("translate job (device) and its text out");
  if(fclose(f) != 0 && !rc) {
    rc = fail(PM_E_IO, "close failed");
  }
  lap("output closed");
  if(!rc && w.parse_rc) {
    rc = fail(w.parse_rc, w.parse_msg);
  }
  return rc;
  });
}
