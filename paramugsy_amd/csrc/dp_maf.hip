// dp_maf.hip -- MAF blocks into the profile DP and out of it (gfx950).
//
// NO REFERENCE COUNTERPART for the DP itself.  The two byte transforms either side of it are the same
// kind of work the reference does around its own path:
//   pack   rows of a MAF block -> one packed column per block column {nA, nC, nG, nT, nGap, nOther, 0, 0}: the per-column
//          fold over a block's rows that lib/profiles/m_make.ml:15-45 does for the consensus (here it counts instead of
//          voting);
//   emit   two blocks + the DP's path -> the merged block: every row's text expanded along the path with '-' where the
//          path skips its side, as lib/profiles/m_untranslate.ml:38-52 (expand_text) re-inserts gap columns along a
//          coordinate walk.
// Both are byte kernels: one thread per block column (pack), one per (pair, path position) writing that column's byte of every
// row of the merged block (emit), consecutive lanes on consecutive bytes of a row; the path's running column positions come from
// one device scan over all ops.  The file entries (pm_dp_align_maf*) map the input files, send their bytes as they are while the
// lines are indexed on the host, and assemble the bytes of the OUTPUT FILE on the device.
// Symbol policy (stated, since nothing upstream defines it): case-insensitive; A, C, G, T count in bytes 0-3, '-' in byte 4,
// anything else (N, IUPAC codes, '.') in byte 5, which the DP does not score: such a row is neutral in that column.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include "dp_batch.hpp"
#include "multi.hpp"
#include "dp_internal.hpp"
#include "pm_internal.hpp"
#include "translate_host.hpp"

namespace pm {

// last b with off[b] <= g
__device__ __forceinline__ i64 owner_of(const i64 *off, i64 n, i64 g) {
  i64 lo = 0, hi = n;
  while(hi - lo > 1) {
    const i64 mid = (lo + hi) >> 1;
    if(off[mid] <= g) {
      lo = mid;
    }
    else {
      hi = mid;
    }
  }
  return lo;
}

__global__ void dp_pack_kernel(i64 first, i64 n_cols, i64 n_blocks, const i64 *__restrict__ col_off, const i64 *__restrict__ block_row,
                               const i64 *__restrict__ row_off, const unsigned char *__restrict__ text, u64 *__restrict__ cols) {
  i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n_cols) {
    return;
  }
  g += first; // columns [first, first + n_cols) of the list of blocks
  const i64 b = owner_of(col_off, n_blocks, g);
  const i64 c = g - col_off[b];
  u64 acc = 0;
  for(i64 r = block_row[b]; r < block_row[b + 1]; ++r) {
    const unsigned char ch = text[row_off[r] + c] & 0xdf; // upper case ('-' = 0x2d -> 0x0d)
    const int k = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == ('-' & 0xdf) ? 4 : 5;
    acc += 1ull << (8 * k);
  }
  cols[g] = acc;
}

// flags for the scans: a[k] = op k consumes a column of A (M or D), b[k] = of B (M or I).  Bytes between two paths (the unused
// heads of pm_dp_batch_fetch's slots) get flags too; only differences inside one path are ever used.
__global__ void dp_op_flags_kernel(i64 n, const unsigned char *__restrict__ ops, int *__restrict__ fa, int *__restrict__ fb) {
  const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n) {
    return;
  }
  const unsigned char op = ops[g];
  fa[g] = op != 1;
  fb[g] = op != 2;
}

// The merged blocks straight into the bytes they are delivered as: the MAF file they are written to (pm_dp_align_maf), or the blocks'
// lines back to back (pm_dp_emit_maf, pm_dp_align_blocks).  One thread per (pair, path position): it reads
// its op and the two column positions once and writes that column's byte of EVERY row of the merged block -- rows(A) + rows(B)
// stores, each coalesced with the neighbouring threads' (consecutive positions of one output line).  line_text[q] = where the text
// of output line q starts in the file image; pair p's lines are first_line[p] .. (A's rows first).
__global__ void dp_emit_file_kernel(i64 n_pairs, const i64 *__restrict__ ops_off, const int *__restrict__ n_ops, const unsigned char *__restrict__ ops,
                                    const int *__restrict__ pos_a, const int *__restrict__ pos_b, const i64 *__restrict__ block_row_a,
                                    const i64 *__restrict__ row_off_a, const unsigned char *__restrict__ text_a, const i64 *__restrict__ block_row_b,
                                    const i64 *__restrict__ row_off_b, const unsigned char *__restrict__ text_b, const i64 *__restrict__ col_off_a,
                                    const i64 *__restrict__ col_off_b, const i64 *__restrict__ first_line, const i64 *__restrict__ line_text,
                                    char *__restrict__ out, int *bad) {
  for(i64 p = blockIdx.y; p < n_pairs; p += gridDim.y) {
    const i64 len = n_ops[p];
    for(i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < len; k += (i64)gridDim.x * blockDim.x) {
      const i64 o = ops_off[p] + k;
      const unsigned char op = ops[o];
      if(op > 2) {
        atomicOr(bad, 1);
      }
      const i64 ca = pos_a[o] - pos_a[ops_off[p]], cb = pos_b[o] - pos_b[ops_off[p]];
      const i64 ra0 = block_row_a[p], ra = block_row_a[p + 1] - ra0, rb0 = block_row_b[p], rb = block_row_b[p + 1] - rb0;
      const i64 q0 = first_line[p];
      if(ra > 0 && op != 1 && ca >= col_off_a[p + 1] - col_off_a[p]) { // (the rows' starts may lie anywhere in the text: a mapped file)
        atomicOr(bad, 2);
      }
      else {
        for(i64 r = 0; r < ra; ++r) {
          out[line_text[q0 + r] + k] = op != 1 ? (char)text_a[row_off_a[ra0 + r] + ca] : '-';
        }
      }
      if(rb > 0 && op != 2 && cb >= col_off_b[p + 1] - col_off_b[p]) {
        atomicOr(bad, 2);
      }
      else {
        for(i64 r = 0; r < rb; ++r) {
          out[line_text[q0 + ra + r] + k] = op != 2 ? (char)text_b[row_off_b[rb0 + r] + cb] : '-';
        }
      }
    }
  }
}

// Where the lines of the merged blocks go when the output is the blocks themselves (pm_dp_emit_maf, pm_dp_align_blocks: pair p's
// rows(A) + rows(B) lines of n_ops[p] bytes back to back at out_off[p]): pair p's first line is line block_row_a[p] + block_row_b[p].
__global__ void dp_line_table_kernel(i64 n_pairs, const i64 *__restrict__ block_row_a, const i64 *__restrict__ block_row_b,
                                     const i64 *__restrict__ out_off, const int *__restrict__ n_ops, i64 *__restrict__ first_line,
                                     i64 *__restrict__ line_text) {
  const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(p > n_pairs) {
    return;
  }
  const i64 q0 = block_row_a[p] + block_row_b[p];
  first_line[p] = q0;
  if(p < n_pairs) {
    const i64 rows = block_row_a[p + 1] - block_row_a[p] + block_row_b[p + 1] - block_row_b[p];
    for(i64 r = 0; r < rows; ++r) {
      line_text[q0 + r] = out_off[p] + r * (i64)n_ops[p];
    }
  }
}

// Everything of the file that is not a row's text -- `a score=` lines, the fields in front of every text, line ends, blank lines --
// prepared by the host as short pieces of one blob: piece q = blob[src[q] .. src[q + 1]) copied to out + dst[q].
__global__ void dp_pieces_kernel(i64 n_pieces, const i64 *__restrict__ dst, const i64 *__restrict__ src, const char *__restrict__ blob,
                                 char *__restrict__ out) {
  const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if(q >= n_pieces) {
    return;
  }
  char *d = out + dst[q];
  for(i64 c = src[q]; c < src[q + 1]; ++c) {
    *d++ = blob[c];
  }
}

// rows of one block must have one length
static int check_blocks(const int64_t *row_off, int64_t n_rows, const int64_t *block_row, int64_t n_blocks, const char *who) {
  if(!row_off || !block_row || n_rows < 0 || n_blocks < 0 || block_row[0] != 0 || block_row[n_blocks] != n_rows || row_off[0] != 0) {
    return fail(PM_E_INVALID, std::string(who) + ": bad block description");
  }
  for(int64_t b = 0; b < n_blocks; ++b) {
    if(block_row[b + 1] < block_row[b]) {
      return fail(PM_E_INVALID, std::string(who) + ": block_row must not decrease");
    }
    if(block_row[b + 1] - block_row[b] > 255) {
      return fail(PM_E_INVALID, std::string(who) + ": a block has more than 255 rows (a packed column counts rows in a byte)");
    }
    for(int64_t r = block_row[b]; r < block_row[b + 1]; ++r) {
      if(row_off[r + 1] < row_off[r] || row_off[r + 1] - row_off[r] != row_off[block_row[b] + 1] - row_off[block_row[b]]) {
        return fail(PM_E_INVALID, std::string(who) + ": rows of one block must have the same number of columns");
      }
    }
  }
  return PM_OK;
}

int dp_check_blocks(const int64_t *row_off, int64_t n_rows, const int64_t *block_row, int64_t n_blocks, const char *who) {
  return check_blocks(row_off, n_rows, block_row, n_blocks, who);
}

// columns [first, first + n_cols) of a list of blocks whose texts and tables are in device memory, on `stream` (dp_stream.hip packs
// a batch segment by segment with this)
int dp_pack_launch(i64 first, i64 n_cols, i64 n_blocks, const i64 *col_off, const i64 *block_row, const i64 *row_off, const unsigned char *text,
                   u64 *cols, hipStream_t stream) {
  if(n_cols > 0) {
    dp_pack_kernel<<<(unsigned)((n_cols + 255) / 256), 256, 0, stream>>>(first, n_cols, n_blocks, col_off, block_row, row_off, text, cols);
    PM_HIP(hipGetLastError());
  }
  return PM_OK;
}

// One `s` line of a MAF block: the six fields in front of the text verbatim, and the text.
struct MafDpRow {
  std::string head; // "s name start size strand srcSize"
};
// A MAF file as the DP's file entries see it: the file mapped read-only, and where its blocks' rows lie in it.  Nothing of the
// rows' texts is copied on the host: row r is bytes [row_off[r], row_off[r] + row_len[r]) of the mapping, and the device gets the
// file's bytes as they are (the kernels take a start per row; lengths come from the blocks' column counts).
struct MafDpBlocks {
  std::vector<MafDpRow> rows;
  std::vector<int64_t> block_row; // [n_blocks + 1]
  std::vector<int64_t> row_off;   // [n_rows + 1]: where each row's text starts in `bytes`; the last entry is n_bytes
  std::vector<int64_t> row_len;   // [n_rows]
  const char *bytes = nullptr;
  size_t n_bytes = 0;
  void *map = nullptr; // the mapping (or nullptr when the file was read into `held`)
  std::string held;
  MafDpBlocks() = default;
  MafDpBlocks(const MafDpBlocks &) = delete;
  MafDpBlocks &operator=(const MafDpBlocks &) = delete;
  ~MafDpBlocks() {
    if(map) {
      munmap(map, n_bytes);
    }
  }
};

// What one range of the file holds (tables local to the range; offsets are file offsets)
struct MafDpRange {
  std::vector<MafDpRow> rows;
  std::vector<int64_t> block_row, row_off, row_len;
};

// `a` opens a block, `s` lines are its rows, anything else (comments, `##maf`, blank lines, other line types) is skipped:
// the block structure of lib/profiles_lib/maf_read_stream.cc:7-45 without its end-of-file quirks.
// The lines of text[begin, end) into `out`; `open`: the range starts inside a block.
static int parse_maf_range(const char *text, size_t begin, size_t end, const std::string &path, MafDpRange &out, bool open) {
  size_t p = begin;
  while(p < end) {
    const char *nl = (const char *)memchr(text + p, '\n', end - p);
    size_t e = nl ? (size_t)(nl - text) : end;
    size_t le = e;
    if(le > p && text[le - 1] == '\r') {
      --le;
    }
    // a block starts at an `a` line; the score after it is optional in MAF, so a line that is just `a` opens a block too
    if((le == p + 1 && text[p] == 'a') || (le > p + 1 && text[p] == 'a' && (text[p + 1] == ' ' || text[p + 1] == '\t'))) {
      out.block_row.push_back((int64_t)out.rows.size());
      open = true;
    }
    else if(le > p + 1 && text[p] == 's' && (text[p + 1] == ' ' || text[p + 1] == '\t')) {
      if(!open) {
        return fail(PM_E_PARSE, path + ": `s` line outside a block");
      }
      // seven whitespace-separated fields; the text is the last
      size_t q = p, fields = 0, text_at = 0;
      while(q < le) {
        while(q < le && (text[q] == ' ' || text[q] == '\t')) {
          ++q;
        }
        if(q >= le) {
          break;
        }
        ++fields;
        if(fields == 7) {
          text_at = q;
        }
        if(fields >= 7) { // the sequence text: long, and blanks in it are rare -- find them with memchr
          const char *b0 = (const char *)memchr(text + q, ' ', le - q), *b1 = (const char *)memchr(text + q, '\t', le - q);
          const char *stop = b0 && b1 ? (b0 < b1 ? b0 : b1) : (b0 ? b0 : b1);
          q = stop ? (size_t)(stop - text) : le;
          continue;
        }
        while(q < le && text[q] != ' ' && text[q] != '\t') {
          ++q;
        }
      }
      if(fields != 7) {
        return fail(PM_E_PARSE, path + ": an `s` line needs 7 fields");
      }
      size_t head_end = text_at;
      while(head_end > p && (text[head_end - 1] == ' ' || text[head_end - 1] == '\t')) {
        --head_end;
      }
      MafDpRow r;
      r.head.assign(text + p, head_end - p);
      size_t te = text_at;
      while(te < le && text[te] != ' ' && text[te] != '\t') {
        ++te;
      }
      out.row_off.push_back((int64_t)text_at);
      out.row_len.push_back((int64_t)(te - text_at));
      out.rows.push_back(std::move(r));
    }
    p = e + 1;
  }
  return PM_OK;
}

// The file mapped read-only (or read, when it cannot be mapped).
static int maf_map(const std::string &path, MafDpBlocks &out) {
  {
    const int fd = open(path.c_str(), O_RDONLY);
    if(fd < 0) {
      return fail(PM_E_IO, "cannot open " + path);
    }
    struct stat st;
    void *m = MAP_FAILED;
    if(fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    }
    if(m != MAP_FAILED) {
      out.map = m;
      out.bytes = (const char *)m;
      out.n_bytes = (size_t)st.st_size;
    }
    else { // not a regular file (a pipe), or empty: read it
      char buf[1 << 16];
      ssize_t n;
      while((n = read(fd, buf, sizeof buf)) > 0) {
        out.held.append(buf, (size_t)n);
      }
      out.bytes = out.held.data();
      out.n_bytes = out.held.size();
    }
    close(fd);
  }
  return PM_OK;
}

// The mapped file cut at block starts into a few ranges whose lines are indexed side by side.
static int maf_index(const std::string &path, MafDpBlocks &out) {
  const bool timing = pm::timing_on();
  auto wall = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const char *text = out.bytes;
  const size_t size = out.n_bytes;
  const double t_read = wall();
  unsigned hw = std::thread::hardware_concurrency();
  size_t n_ranges = hw == 0 ? 1 : (hw > 16 ? 8 : (hw + 1) / 2); // two files are parsed at once: half the threads each
  if(size < ((size_t)4 << 20)) {
    n_ranges = 1;
  }
  // range k starts at the first block start at or after its share of the bytes
  std::vector<size_t> cut(n_ranges + 1, size);
  cut[0] = 0;
  for(size_t k = 1; k < n_ranges; ++k) {
    size_t p = std::max(cut[k - 1], size * k / n_ranges);
    while(p < size) {
      const char *nl = (const char *)memchr(text + p, '\n', size - p);
      if(!nl) {
        p = size;
        break;
      }
      p = (size_t)(nl - text) + 1;
      if(p < size && text[p] == 'a' && (p + 1 == size || text[p + 1] == ' ' || text[p + 1] == '\t' || text[p + 1] == '\n' || text[p + 1] == '\r')) {
        break;
      }
    }
    cut[k] = p;
  }
  std::vector<MafDpRange> part(n_ranges);
  std::vector<int> rc(n_ranges, PM_OK);
  std::vector<std::string> msg(n_ranges);
  {
    std::vector<JoinThread> th;
    auto work = [&](size_t k) {
      rc[k] = parse_maf_range(text, cut[k], cut[k + 1], path, part[k], k > 0 && cut[k] < size);
      if(rc[k]) {
        msg[k] = pm_last_error();
      }
    };
    for(size_t k = 1; k < n_ranges; ++k) {
      th.emplace_back([&work, k]() { work(k); });
    }
    work(0);
    for(size_t k = 0; k < th.size(); ++k) {
      th[k].join_and_rethrow();
    }
  }
  for(size_t k = 0; k < n_ranges; ++k) {
    if(rc[k]) {
      return fail(rc[k], msg[k]);
    }
  }
  const double t_parsed = wall();
  size_t n_rows = 0, n_blocks = 0;
  for(size_t k = 0; k < n_ranges; ++k) {
    n_rows += part[k].rows.size();
    n_blocks += part[k].block_row.size();
  }
  out.rows.reserve(n_rows);
  out.row_off.reserve(n_rows + 1);
  out.row_len.reserve(n_rows);
  out.block_row.reserve(n_blocks + 1);
  for(size_t k = 0; k < n_ranges; ++k) { // file offsets need no rebasing; the block tables count rows from the range's first
    MafDpRange &p = part[k];
    const int64_t row0 = (int64_t)out.rows.size();
    for(size_t bk = 0; bk < p.block_row.size(); ++bk) {
      out.block_row.push_back(row0 + p.block_row[bk]);
    }
    std::move(p.rows.begin(), p.rows.end(), std::back_inserter(out.rows));
    out.row_off.insert(out.row_off.end(), p.row_off.begin(), p.row_off.end());
    out.row_len.insert(out.row_len.end(), p.row_len.begin(), p.row_len.end());
  }
  out.row_off.push_back((int64_t)size);
  out.block_row.push_back((int64_t)out.rows.size()); // a file without any `a` line: block_row = {0}, no blocks
  // what pm_dp_pack_maf's check refuses: blocks deeper than a byte counts, rows of one block of different lengths
  for(size_t bk = 0; bk + 1 < out.block_row.size(); ++bk) {
    const int64_t r0 = out.block_row[bk], r1 = out.block_row[bk + 1];
    if(r1 - r0 > 255) {
      return fail(PM_E_INVALID, path + ": a block has more than 255 rows (a packed column counts rows in a byte)");
    }
    for(int64_t r = r0 + 1; r < r1; ++r) {
      if(out.row_len[(size_t)r] != out.row_len[(size_t)r0]) {
        return fail(PM_E_INVALID, path + ": rows of one block must have the same number of columns");
      }
    }
  }
  if(timing) {
    fprintf(stderr, "[pm]   %s: %zu ranges indexed in %.4f s, joined in %.4f s\n", path.c_str(), n_ranges, t_parsed - t_read, wall() - t_parsed);
  }
  return PM_OK;
}

static int parse_maf_blocks(const std::string &path, MafDpBlocks &out) {
  PM_TRY(maf_map(path, out));
  return maf_index(path, out);
}

// One side's blocks in device memory: the flat text and its tables, and (after pack()) the packed columns.  The two big
// buffers come from the pool of kept device buffers (pm_internal.hpp).
struct MafSideDev {
  PooledBuf text, cols;
  DevBuf row_off, block_row, col_off;
  i64 n_blocks = 0, n_cols = 0;
  bool text_up = false; // the text was sent ahead (pm_dp_align_maf starts the copies while the files are still being indexed)
  int upload_text(const uint8_t *t, size_t bytes) {
    int device = 0;
    PM_HIP(hipGetDevice(&device));
    PM_TRY(text.alloc(bytes, device));
    if(bytes > 0) {
      PM_HIP(hipMemcpy(text.p, t, bytes, hipMemcpyHostToDevice));
    }
    text_up = true;
    return PM_OK;
  }
  int upload(const uint8_t *t, const int64_t *ro, int64_t n_rows, const int64_t *br, int64_t nb, const int64_t *co) {
    n_blocks = nb;
    n_cols = co[nb];
    if(!text_up) {
      PM_TRY(upload_text(t, (size_t)ro[n_rows]));
    }
    PM_TRY(row_off.upload(ro, (size_t)(n_rows + 1) * 8, nullptr));
    PM_TRY(block_row.upload(br, (size_t)(nb + 1) * 8, nullptr));
    PM_TRY(col_off.upload(co, (size_t)(nb + 1) * 8, nullptr));
    return PM_OK;
  }
  int pack() {
    int device = 0;
    PM_HIP(hipGetDevice(&device));
    PM_TRY(cols.alloc((size_t)std::max<i64>(n_cols, 1) * 8, device));
    if(n_cols > 0) {
      dp_pack_kernel<<<(unsigned)((n_cols + 255) / 256), 256>>>(0, n_cols, n_blocks, (const i64 *)col_off.p, (const i64 *)block_row.p,
                                                                (const i64 *)row_off.p, (const unsigned char *)text.p, (u64 *)cols.p);
      PM_HIP(hipGetLastError());
    }
    return PM_OK;
  }
};

// The merged blocks of n_pairs pairs from texts, tables and paths that are all in device memory (d_ops: ops_end bytes, pair p's
// path at d_ops_off[p], d_n_ops[p] long; d_out_off: where each pair's text goes) into d_out (n_out bytes, allocated here).
static int emit_device(const MafSideDev &A, const MafSideDev &B, i64 n_pairs, i64 n_lines, i64 max_len, const unsigned char *d_ops, i64 ops_end,
                       const i64 *d_ops_off, const int *d_n_ops, const i64 *d_out_off, i64 n_out, DevBuf &d_out, const char *who) {
  int device = 0;
  PM_HIP(hipGetDevice(&device));
  PooledBuf d_fa, d_fb, d_pa, d_pb;
  DevBuf d_bad, d_tmp, d_first_line, d_line_text;
  PM_TRY(d_fa.alloc((size_t)ops_end * 4 + 4, device));
  PM_TRY(d_fb.alloc((size_t)ops_end * 4 + 4, device));
  PM_TRY(d_pa.alloc((size_t)ops_end * 4 + 4, device));
  PM_TRY(d_pb.alloc((size_t)ops_end * 4 + 4, device));
  PM_TRY(d_out.alloc((size_t)n_out));
  PM_TRY(d_bad.alloc(4));
  PM_TRY(d_first_line.alloc((size_t)(n_pairs + 1) * 8));
  PM_TRY(d_line_text.alloc((size_t)(n_lines + 1) * 8));
  PM_HIP(hipMemset(d_bad.p, 0, 4));
  dp_op_flags_kernel<<<(unsigned)((ops_end + 255) / 256), 256>>>(ops_end, d_ops, (int *)d_fa.p, (int *)d_fb.p);
  PM_HIP(hipGetLastError());
  size_t tmp_bytes = 0;
  PM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, (int *)d_fa.p, (int *)d_pa.p, 0, (size_t)ops_end, rocprim::plus<int>()));
  PM_TRY(d_tmp.alloc(tmp_bytes));
  PM_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (int *)d_fa.p, (int *)d_pa.p, 0, (size_t)ops_end, rocprim::plus<int>()));
  PM_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (int *)d_fb.p, (int *)d_pb.p, 0, (size_t)ops_end, rocprim::plus<int>()));
  dp_line_table_kernel<<<(unsigned)((n_pairs + 256) / 256), 256>>>(n_pairs, (const i64 *)A.block_row.p, (const i64 *)B.block_row.p, d_out_off,
                                                                   d_n_ops, (i64 *)d_first_line.p, (i64 *)d_line_text.p);
  PM_HIP(hipGetLastError());
  // one thread per (pair, path position): every row's byte of that column (the kernel of the file path, with the lines back to back)
  const unsigned gx = (unsigned)std::min<i64>((std::max<i64>(max_len, 1) + 255) / 256, 1024), gy = (unsigned)std::min<i64>(std::max<i64>(n_pairs, 1), 65535);
  dp_emit_file_kernel<<<dim3(gx, gy), 256>>>(n_pairs, d_ops_off, d_n_ops, d_ops, (const int *)d_pa.p, (const int *)d_pb.p, (const i64 *)A.block_row.p,
                                             (const i64 *)A.row_off.p, (const unsigned char *)A.text.p, (const i64 *)B.block_row.p,
                                             (const i64 *)B.row_off.p, (const unsigned char *)B.text.p, (const i64 *)A.col_off.p,
                                             (const i64 *)B.col_off.p, (const i64 *)d_first_line.p, (const i64 *)d_line_text.p, (char *)d_out.p,
                                             (int *)d_bad.p);
  PM_HIP(hipGetLastError());
  int bad = 0;
  PM_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
  if(bad) {
    return fail(PM_E_INVALID, std::string(who) + ": an op outside {0, 1, 2} or a path that leaves its block");
  }
  return PM_OK;
}

} // namespace pm

using namespace pm;

extern "C" {

int pm_dp_pack_maf(const uint8_t *text, const int64_t *row_off, int64_t n_rows, const int64_t *block_row, int64_t n_blocks,
                   uint8_t *cols_out, int64_t *col_off_out, int device) {
  PM_TRY(use_device(device));
  PM_TRY(check_blocks(row_off, n_rows, block_row, n_blocks, "pm_dp_pack_maf"));
  if(!col_off_out) {
    return fail(PM_E_INVALID, "pm_dp_pack_maf: null col_off_out");
  }
  col_off_out[0] = 0;
  for(int64_t b = 0; b < n_blocks; ++b) {
    const int64_t first = block_row[b];
    col_off_out[b + 1] = col_off_out[b] + (block_row[b + 1] > first ? row_off[first + 1] - row_off[first] : 0);
  }
  const int64_t n_cols = col_off_out[n_blocks];
  if(!cols_out || n_cols == 0) {
    return PM_OK; // sizes only
  }
  if(!text) {
    return fail(PM_E_INVALID, "pm_dp_pack_maf: null text");
  }
  MafSideDev side;
  PM_TRY(side.upload(text, row_off, n_rows, block_row, n_blocks, col_off_out));
  PM_TRY(side.pack());
  PM_HIP(hipMemcpy(cols_out, side.cols.p, (size_t)n_cols * 8, hipMemcpyDeviceToHost));
  return PM_OK;
}

int pm_dp_emit_maf(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a, const uint8_t *text_b,
                   const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs, const uint8_t *ops,
                   const int64_t *ops_off, const int32_t *n_ops, uint8_t *out_text, int64_t *out_off, int device) {
  PM_TRY(use_device(device));
  PM_TRY(check_blocks(row_off_a, n_rows_a, block_row_a, n_pairs, "pm_dp_emit_maf (A)"));
  PM_TRY(check_blocks(row_off_b, n_rows_b, block_row_b, n_pairs, "pm_dp_emit_maf (B)"));
  if(!ops_off || !n_ops || !out_off) {
    return fail(PM_E_INVALID, "pm_dp_emit_maf: null argument");
  }
  out_off[0] = 0;
  int64_t ops_end = 0;
  for(int64_t p = 0; p < n_pairs; ++p) {
    if(n_ops[p] < 0 || ops_off[p] < 0) {
      return fail(PM_E_INVALID, "pm_dp_emit_maf: bad path description");
    }
    const int64_t rows = (block_row_a[p + 1] - block_row_a[p]) + (block_row_b[p + 1] - block_row_b[p]);
    out_off[p + 1] = out_off[p] + rows * (int64_t)n_ops[p];
    ops_end = std::max(ops_end, ops_off[p] + n_ops[p]);
  }
  const int64_t n_out = out_off[n_pairs];
  if(!out_text || n_out == 0) {
    return PM_OK; // sizes only
  }
  if(ops_end >= ((int64_t)1 << 31)) {
    return fail(PM_E_INVALID, "pm_dp_emit_maf: more than 2^31 ops in one call");
  }
  if(!ops || (!text_a && row_off_a[n_rows_a] > 0) || (!text_b && row_off_b[n_rows_b] > 0)) {
    return fail(PM_E_INVALID, "pm_dp_emit_maf: null buffer");
  }
  // host check: every path consumes exactly its two blocks (a path that does not cannot be expanded)
  for(int64_t p = 0; p < n_pairs; ++p) {
    int64_t na = 0, nb = 0;
    for(int64_t k = 0; k < n_ops[p]; ++k) {
      const uint8_t op = ops[ops_off[p] + k];
      na += op != 1;
      nb += op != 2;
    }
    const int64_t ra = block_row_a[p + 1] - block_row_a[p], rb = block_row_b[p + 1] - block_row_b[p];
    const int64_t la = ra ? row_off_a[block_row_a[p] + 1] - row_off_a[block_row_a[p]] : 0;
    const int64_t lb = rb ? row_off_b[block_row_b[p] + 1] - row_off_b[block_row_b[p]] : 0;
    if((ra && na != la) || (rb && nb != lb)) {
      return fail(PM_E_INVALID, "pm_dp_emit_maf: a path does not span its pair of blocks");
    }
  }
  MafSideDev A, B;
  std::vector<int64_t> cols_a((size_t)n_pairs + 1, 0), cols_b((size_t)n_pairs + 1, 0); // a block has as many columns as its first row has bytes
  int64_t max_len = 1;
  for(int64_t p = 0; p < n_pairs; ++p) {
    const int64_t ra = block_row_a[p], rb = block_row_b[p];
    cols_a[(size_t)p + 1] = cols_a[(size_t)p] + (ra < block_row_a[p + 1] ? row_off_a[ra + 1] - row_off_a[ra] : 0);
    cols_b[(size_t)p + 1] = cols_b[(size_t)p] + (rb < block_row_b[p + 1] ? row_off_b[rb + 1] - row_off_b[rb] : 0);
    max_len = std::max<int64_t>(max_len, n_ops[p]);
  }
  PM_TRY(A.upload(text_a, row_off_a, n_rows_a, block_row_a, n_pairs, cols_a.data()));
  PM_TRY(B.upload(text_b, row_off_b, n_rows_b, block_row_b, n_pairs, cols_b.data()));
  DevBuf d_ops, d_ops_off, d_n_ops, d_out_off, d_out;
  PM_TRY(d_ops.upload(ops, (size_t)ops_end, nullptr));
  PM_TRY(d_ops_off.upload(ops_off, (size_t)n_pairs * 8, nullptr));
  PM_TRY(d_n_ops.upload(n_ops, (size_t)n_pairs * 4, nullptr));
  PM_TRY(d_out_off.upload(out_off, (size_t)(n_pairs + 1) * 8, nullptr));
  PM_TRY(emit_device(A, B, n_pairs, n_rows_a + n_rows_b, max_len, (const unsigned char *)d_ops.p, ops_end, (const i64 *)d_ops_off.p,
                     (const int *)d_n_ops.p, (const i64 *)d_out_off.p, n_out, d_out, "pm_dp_emit_maf"));
  PM_HIP(hipMemcpy(out_text, d_out.p, (size_t)n_out, hipMemcpyDeviceToHost));
  return PM_OK;
}

} // extern "C"

// Both files parsed side by side; each parser reports through its own return value (pm_last_error is per thread).
static int parse_two_mafs(const char *maf_a, const char *maf_b, MafDpBlocks &A, MafDpBlocks &B, const char *who) {
  int rc_b = PM_OK;
  std::string err_b;
  JoinThread other([&]() {
    rc_b = parse_maf_blocks(maf_b, B);
    if(rc_b) {
      err_b = pm_last_error();
    }
  });
  const int rc_a = parse_maf_blocks(maf_a, A);
  other.join_and_rethrow();
  if(rc_a) {
    return rc_a;
  }
  if(rc_b) {
    return fail(rc_b, err_b);
  }
  if(A.block_row.size() != B.block_row.size()) {
    return fail(PM_E_INVALID, std::string(who) + ": the two MAF files must hold the same number of blocks (pair k = block k of each)");
  }
  return PM_OK;
}

// DP batches kept from call to call, at most two per device (their workspace and buffers only grow: dp_batch.hpp): a resident caller's
// next file finds gigabytes of path workspace allocated -- allocating and freeing it costs more than the kernels that use it.  A batch
// whose workspace has grown past 24 GiB is not kept; and whatever is kept is given back when a device allocation fails
// (malloc_trimming, pm_internal.hpp).
static std::mutex g_batch_lock;
static std::vector<std::unique_ptr<pm_dp_batch> > g_batch_cache;
static std::unique_ptr<pm_dp_batch> batch_acquire(int device) {
  {
    std::lock_guard<std::mutex> hold(g_batch_lock);
    for(size_t k = 0; k < g_batch_cache.size(); ++k) {
      if(g_batch_cache[k]->device == device) {
        std::unique_ptr<pm_dp_batch> b = std::move(g_batch_cache[k]);
        g_batch_cache.erase(g_batch_cache.begin() + (long)k);
        return b;
      }
    }
  }
  return std::unique_ptr<pm_dp_batch>(new(std::nothrow) pm_dp_batch());
}
namespace pm {
void dp_batch_cache_trim() {
  std::vector<std::unique_ptr<pm_dp_batch> > all;
  {
    std::lock_guard<std::mutex> hold(g_batch_lock);
    all.swap(g_batch_cache);
  }
  int cur = -1;
  (void)hipGetDevice(&cur);
  for(size_t k = 0; k < all.size(); ++k) {
    (void)hipSetDevice(all[k]->device);
    all[k].reset();
  }
  if(cur >= 0) {
    (void)hipSetDevice(cur);
  }
}
} // namespace pm
static void batch_release(std::unique_ptr<pm_dp_batch> b) {
  if(!b) {
    return;
  }
  if(b->tb.bytes <= ((size_t)24 << 30)) {
    std::lock_guard<std::mutex> hold(g_batch_lock);
    size_t same_device = 0; // at most two kept batches per device (two workers may share one: the device list {0, 0})
    for(size_t k = 0; k < g_batch_cache.size(); ++k) {
      same_device += g_batch_cache[k]->device == b->device;
    }
    if(same_device < 2) {
      g_batch_cache.push_back(std::move(b));
      return;
    }
  }
  b.reset();
}

// Blocks in, merged blocks out, through the device once: texts up, pack, DP, expansion along the paths, merged texts down.
// scores / n_ops: n values each (n_ops[k] = columns of merged block k); out_off: n + 1 byte offsets into `merged`.
static int align_blocks_core(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a, const uint8_t *text_b,
                             const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n, const pm_dp_params_t *params,
                             int device, std::vector<int32_t> &scores, std::vector<int32_t> &n_ops, std::vector<uint8_t> &merged,
                             std::vector<int64_t> &out_off, const std::function<void(const char *)> &lap) {
  PM_TRY(check_blocks(row_off_a, n_rows_a, block_row_a, n, "pm_dp_align (A)"));
  PM_TRY(check_blocks(row_off_b, n_rows_b, block_row_b, n, "pm_dp_align (B)"));
  // a block has as many columns as its first row has bytes
  std::vector<int64_t> coa((size_t)n + 1, 0), cob((size_t)n + 1, 0);
  for(int64_t k = 0; k < n; ++k) {
    const int64_t ra = block_row_a[k], rb = block_row_b[k];
    coa[(size_t)k + 1] = coa[(size_t)k] + (ra < block_row_a[k + 1] ? row_off_a[ra + 1] - row_off_a[ra] : 0);
    cob[(size_t)k + 1] = cob[(size_t)k] + (rb < block_row_b[k + 1] ? row_off_b[rb + 1] - row_off_b[rb] : 0);
  }
  scores.assign((size_t)n, 0);
  n_ops.assign((size_t)n, 0);
  out_off.assign((size_t)n + 1, 0);
  merged.clear();
  if(n == 0) {
    return PM_OK;
  }
  // the texts go to the device once; the packed columns, the paths and the merged texts never leave it before the last copy
  MafSideDev SA, SB;
  PM_TRY(SA.upload(text_a, row_off_a, n_rows_a, block_row_a, n, coa.data()));
  PM_TRY(SB.upload(text_b, row_off_b, n_rows_b, block_row_b, n, cob.data()));
  lap("tables (and texts not sent ahead) on the device");
  PM_TRY(SA.pack());
  PM_TRY(SB.pack());
  lap("pack kernels launched");
  PM_TRY(dp_batch_check_params(params));
  std::unique_ptr<pm_dp_batch> batch = batch_acquire(device);
  if(!batch) {
    return fail(PM_E_INVALID, "out of host memory");
  }
  PM_TRY(dp_batch_init(batch.get(), params, 0, device));
  PM_TRY(dp_batch_load(batch.get(), (const uint8_t *)SA.cols.p, coa.data(), (const uint8_t *)SB.cols.p, cob.data(), n, nullptr));
  PM_HIP(hipStreamSynchronize(nullptr));
  PM_TRY(dp_batch_plan(batch.get(), nullptr));
  PM_TRY(dp_run(batch.get(), nullptr, 1, nullptr, nullptr));
  int perr = 0;
  PM_HIP(hipMemcpy(scores.data(), batch->scores.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(n_ops.data(), batch->n_ops.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(&perr, batch->pipe_error.p, 4, hipMemcpyDeviceToHost));
  if(perr) {
    return fail(PM_E_HIP, "dp_fill_kernel: a stripe timed out waiting for its left neighbour (results invalid)");
  }
  lap("DP (device)");
  // pair k's path is the last n_ops[k] bytes of its slot; every row of its merged block is as long as the path
  std::vector<int64_t> ops_off((size_t)n);
  for(int64_t k = 0; k < n; ++k) {
    ops_off[(size_t)k] = coa[(size_t)k + 1] + cob[(size_t)k + 1] - n_ops[(size_t)k];
    out_off[(size_t)k + 1] = out_off[(size_t)k] + (block_row_a[k + 1] - block_row_a[k] + block_row_b[k + 1] - block_row_b[k]) * (int64_t)n_ops[(size_t)k];
  }
  const int64_t ops_end = coa[(size_t)n] + cob[(size_t)n];
  if(ops_end >= ((int64_t)1 << 31)) {
    return fail(PM_E_INVALID, "pm_dp_align: more than 2^31 columns in one call");
  }
  merged.resize((size_t)out_off[(size_t)n] + 1);
  if(out_off[(size_t)n] > 0) {
    DevBuf d_ops_off, d_n_ops, d_out_off, d_out;
    PM_TRY(d_ops_off.upload(ops_off.data(), (size_t)n * 8, nullptr));
    PM_TRY(d_n_ops.upload(n_ops.data(), (size_t)n * 4, nullptr));
    PM_TRY(d_out_off.upload(out_off.data(), (size_t)(n + 1) * 8, nullptr));
    int64_t max_len = 1;
    for(int64_t k = 0; k < n; ++k) {
      max_len = std::max<int64_t>(max_len, n_ops[(size_t)k]);
    }
    PM_TRY(emit_device(SA, SB, n, n_rows_a + n_rows_b, max_len, (const unsigned char *)batch->ops.p, ops_end, (const i64 *)d_ops_off.p,
                       (const int *)d_n_ops.p, (const i64 *)d_out_off.p, out_off[(size_t)n], d_out, "pm_dp_align"));
    PM_HIP(hipMemcpy(merged.data(), d_out.p, (size_t)out_off[(size_t)n], hipMemcpyDeviceToHost));
  }
  lap("emit (device)");
  batch_release(std::move(batch));
  return PM_OK;
}

// Pairs [lo, hi) of a flat block description as a description of their own (tables rebased to 0; the text is not copied).
struct BlockSlice {
  const uint8_t *text;
  std::vector<int64_t> row_off, block_row;
  int64_t n_rows;
  // absolute: the offsets stay relative to `t` (the whole text is, or will be, in device memory as it is)
  BlockSlice(const uint8_t *t, const int64_t *ro, const int64_t *br, int64_t lo, int64_t hi, bool absolute = false) {
    const int64_t r0 = br[lo], r1 = br[hi];
    n_rows = r1 - r0;
    const int64_t base = absolute ? 0 : ro[r0];
    text = t ? t + base : nullptr;
    row_off.resize((size_t)n_rows + 1);
    for(int64_t r = r0; r <= r1; ++r) {
      row_off[(size_t)(r - r0)] = ro[r] - base;
    }
    block_row.resize((size_t)(hi - lo) + 1);
    for(int64_t k = lo; k <= hi; ++k) {
      block_row[(size_t)(k - lo)] = br[k] - r0;
    }
  }
};

// Blocks [of A and B, pair k = block k] -> the bytes of the MAF file of their merged blocks, into `out`: texts up, pack, DP, then
// the file image assembled ON THE DEVICE (dp_emit_file_kernel for the rows' texts, dp_pieces_kernel for everything around them)
// and brought back through pinned staging pieces beside the writing (device_bytes_to_sink).  with_header: the `##maf` line first.
// What the host does in between is arithmetic on 2 numbers per block and a few dozen bytes per row.
// sent_a / sent_b: the sides' texts already in device memory (the whole files; only with lo = 0 and hi = every block), or null.
static int align_maf_to_sink(const MafDpBlocks &A, const MafDpBlocks &B, int64_t lo, int64_t hi, const pm_dp_params_t *params, int device,
                             bool with_header, OutSink out, const std::function<void(const char *)> &lap,
                             std::unique_ptr<MafSideDev> sent_a = nullptr, std::unique_ptr<MafSideDev> sent_b = nullptr,
                             const std::function<int()> &before_write = nullptr) {
  const int64_t n = hi - lo;
  std::string blob;
  if(with_header) {
    blob = "##maf version=1 scoring=paramugsy_amd\n";
  }
  const bool timing = pm::timing_on();
  if(n == 0) {
    if(before_write) {
      PM_TRY(before_write());
    }
    return out.write(blob.data(), blob.size()) ? (int)PM_OK : fail(PM_E_IO, "write failed");
  }
  // the slice's part of either mapped file: from its first row's text to the start of the row after its last (the lines' other
  // fields in between travel along; the kernels only look at [row start, row start + the block's columns))
  BlockSlice sa((const uint8_t *)A.bytes, A.row_off.data(), A.block_row.data(), lo, hi, sent_a != nullptr);
  BlockSlice sb((const uint8_t *)B.bytes, B.row_off.data(), B.block_row.data(), lo, hi, sent_b != nullptr);
  // a block has as many columns as its rows have bytes (one length per block: checked by the parser)
  std::vector<int64_t> coa((size_t)n + 1, 0), cob((size_t)n + 1, 0);
  for(int64_t k = 0; k < n; ++k) {
    const int64_t ra = A.block_row[(size_t)(lo + k)], rb = B.block_row[(size_t)(lo + k)];
    coa[(size_t)k + 1] = coa[(size_t)k] + (ra < A.block_row[(size_t)(lo + k) + 1] ? A.row_len[(size_t)ra] : 0);
    cob[(size_t)k + 1] = cob[(size_t)k] + (rb < B.block_row[(size_t)(lo + k) + 1] ? B.row_len[(size_t)rb] : 0);
  }
  // (owned through pointers: they are released by a helper thread while the file image is on its way to the host)
  std::unique_ptr<MafSideDev> SAp(sent_a ? sent_a.release() : new MafSideDev()), SBp(sent_b ? sent_b.release() : new MafSideDev());
  MafSideDev &SA = *SAp, &SB = *SBp;
  {
    // the two sides go up side by side (two copies from mapped files, each bound by the host's copy into staging memory)
    int rc_b = PM_OK;
    std::string msg_b;
    JoinThread other([&]() {
      rc_b = use_device(device);
      if(!rc_b) {
        rc_b = SB.upload(sb.text, sb.row_off.data(), sb.n_rows, sb.block_row.data(), n, cob.data());
      }
      if(rc_b) {
        msg_b = pm_last_error();
      }
    });
    const int rc_a = SA.upload(sa.text, sa.row_off.data(), sa.n_rows, sa.block_row.data(), n, coa.data());
    other.join_and_rethrow();
    if(rc_a) {
      return rc_a;
    }
    if(rc_b) {
      return fail(rc_b, msg_b);
    }
  }
  lap("tables (and texts not sent ahead) on the device");
  PM_TRY(SA.pack());
  PM_TRY(SB.pack());
  lap("pack kernels launched");
  PM_TRY(dp_batch_check_params(params));
  std::unique_ptr<pm_dp_batch> batch = batch_acquire(device);
  if(!batch) {
    return fail(PM_E_INVALID, "out of host memory");
  }
  PM_TRY(dp_batch_init(batch.get(), params, 0, device));
  PM_TRY(dp_batch_load(batch.get(), (const uint8_t *)SA.cols.p, coa.data(), (const uint8_t *)SB.cols.p, cob.data(), n, nullptr));
  PM_HIP(hipStreamSynchronize(nullptr));
  PM_TRY(dp_batch_plan(batch.get(), nullptr));
  PM_TRY(dp_run(batch.get(), nullptr, 1, nullptr, nullptr));
  std::vector<int32_t> scores((size_t)n), n_ops((size_t)n);
  int perr = 0;
  PM_HIP(hipMemcpy(scores.data(), batch->scores.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(n_ops.data(), batch->n_ops.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  PM_HIP(hipMemcpy(&perr, batch->pipe_error.p, 4, hipMemcpyDeviceToHost));
  if(perr) {
    return fail(PM_E_HIP, "dp_fill_kernel: a stripe timed out waiting for its left neighbour (results invalid)");
  }
  lap("DP (device)");
  // the file image: where every line's text goes, and the pieces around the texts
  const int64_t ops_end = coa[(size_t)n] + cob[(size_t)n];
  if(ops_end >= ((int64_t)1 << 31)) {
    return fail(PM_E_INVALID, "pm_dp_align_maf: more than 2^31 columns in one call");
  }
  const int64_t n_lines = sa.n_rows + sb.n_rows;
  std::vector<int64_t> ops_off((size_t)n), first_line((size_t)n + 1, 0), line_text((size_t)n_lines + 1, 0), dst, src;
  dst.reserve((size_t)(n_lines + n + 2));
  src.reserve((size_t)(n_lines + n + 3));
  int64_t pos = 0;
  int64_t max_len = 1;
  auto piece = [&](const char *p, size_t len) { // appended to the piece that is open (pieces are cut where a row's text lies between)
    blob.append(p, len);
    pos += (int64_t)len;
  };
  auto cut = [&]() { // the open piece ends here; the next starts at `pos`
    src.push_back((int64_t)blob.size());
    dst.push_back(pos);
  };
  dst.push_back(0);
  src.push_back(0);
  pos = (int64_t)blob.size(); // the header line, if any, opens the first piece
  for(int64_t k = 0; k < n; ++k) {
    const int64_t ra = sa.block_row[(size_t)k + 1] - sa.block_row[(size_t)k], rb = sb.block_row[(size_t)k + 1] - sb.block_row[(size_t)k];
    const int64_t len = n_ops[(size_t)k];
    max_len = std::max(max_len, len);
    ops_off[(size_t)k] = coa[(size_t)k + 1] + cob[(size_t)k + 1] - len; // the path is the last n_ops bytes of the pair's slot
    char a_line[48];
    const int al = snprintf(a_line, sizeof a_line, "a score=%d\n", (int)scores[(size_t)k]);
    piece(a_line, (size_t)al);
    first_line[(size_t)k + 1] = first_line[(size_t)k] + ra + rb;
    for(int64_t r = 0; r < ra + rb; ++r) {
      const MafDpRow &row = r < ra ? A.rows[(size_t)(A.block_row[lo + k] + r)] : B.rows[(size_t)(B.block_row[lo + k] + r - ra)];
      piece(row.head.data(), row.head.size());
      piece(" ", 1);
      line_text[(size_t)(first_line[(size_t)k] + r)] = pos;
      pos += len; // the row's text: written by dp_emit_file_kernel
      cut();
      piece("\n", 1);
    }
    piece("\n", 1);
  }
  src.push_back((int64_t)blob.size()); // the last piece's end
  const int64_t n_pieces = (int64_t)dst.size();
  const int64_t n_out = pos;
  DevBuf d_ops_off, d_n_ops, d_first_line, d_line_text, d_dst, d_src, d_blob, d_bad;
  PooledBuf d_out;
  PM_TRY(d_ops_off.upload(ops_off.data(), (size_t)n * 8, nullptr));
  PM_TRY(d_n_ops.upload(n_ops.data(), (size_t)n * 4, nullptr));
  PM_TRY(d_first_line.upload(first_line.data(), (size_t)(n + 1) * 8, nullptr));
  PM_TRY(d_line_text.upload(line_text.data(), (size_t)(n_lines + 1) * 8, nullptr));
  PM_TRY(d_dst.upload(dst.data(), (size_t)n_pieces * 8, nullptr));
  PM_TRY(d_src.upload(src.data(), (size_t)(n_pieces + 1) * 8, nullptr));
  PM_TRY(d_blob.upload(blob.data(), blob.size(), nullptr));
  PM_TRY(d_out.alloc((size_t)n_out, device));
  PM_TRY(d_bad.alloc(4));
  PM_HIP(hipMemset(d_bad.p, 0, 4));
  {
    // every op's column on either side: one scan each over the ops of the whole batch
    PooledBuf d_fa, d_fb, d_pa, d_pb;
    DevBuf d_tmp;
    PM_TRY(d_fa.alloc((size_t)ops_end * 4 + 4, device));
    PM_TRY(d_fb.alloc((size_t)ops_end * 4 + 4, device));
    PM_TRY(d_pa.alloc((size_t)ops_end * 4 + 4, device));
    PM_TRY(d_pb.alloc((size_t)ops_end * 4 + 4, device));
    if(ops_end > 0) {
      dp_op_flags_kernel<<<(unsigned)((ops_end + 255) / 256), 256>>>(ops_end, (const unsigned char *)batch->ops.p, (int *)d_fa.p, (int *)d_fb.p);
      PM_HIP(hipGetLastError());
      size_t tmp_bytes = 0;
      PM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, (int *)d_fa.p, (int *)d_pa.p, 0, (size_t)ops_end, rocprim::plus<int>()));
      PM_TRY(d_tmp.alloc(tmp_bytes));
      PM_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (int *)d_fa.p, (int *)d_pa.p, 0, (size_t)ops_end, rocprim::plus<int>()));
      PM_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (int *)d_fb.p, (int *)d_pb.p, 0, (size_t)ops_end, rocprim::plus<int>()));
      const unsigned gx = (unsigned)std::min<int64_t>((max_len + 255) / 256, 1024), gy = (unsigned)std::min<int64_t>(n, 65535);
      dp_emit_file_kernel<<<dim3(gx, gy), 256>>>(n, (const i64 *)d_ops_off.p, (const int *)d_n_ops.p, (const unsigned char *)batch->ops.p,
                                                 (const int *)d_pa.p, (const int *)d_pb.p, (const i64 *)SA.block_row.p, (const i64 *)SA.row_off.p,
                                                 (const unsigned char *)SA.text.p, (const i64 *)SB.block_row.p, (const i64 *)SB.row_off.p,
                                                 (const unsigned char *)SB.text.p, (const i64 *)SA.col_off.p, (const i64 *)SB.col_off.p,
                                                 (const i64 *)d_first_line.p, (const i64 *)d_line_text.p,
                                                 (char *)d_out.p, (int *)d_bad.p);
      PM_HIP(hipGetLastError());
    }
    dp_pieces_kernel<<<(unsigned)((n_pieces + 255) / 256), 256>>>(n_pieces, (const i64 *)d_dst.p, (const i64 *)d_src.p, (const char *)d_blob.p,
                                                                  (char *)d_out.p);
    PM_HIP(hipGetLastError());
    int bad = 0;
    PM_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost)); // waits for the kernels
    if(bad) {
      return fail(PM_E_INVALID, "pm_dp_align_maf: an op outside {0, 1, 2} or a path that leaves its block");
    }
  }
  lap("file image (device)");
  // everything but the image is released beside its way to the host (a dozen milliseconds of hipFree for a few GB)
  JoinThread reaper([&]() {
    if(use_device(device) == PM_OK) {
      batch_release(std::move(batch));
      SAp.reset();
      SBp.reset();
    }
  });
  if(before_write) { // the caller's last word before bytes reach its sink (pm_dp_align_maf: the output file is emptied only now)
    const int rc_bw = before_write();
    if(rc_bw) {
      reaper.join();
      return rc_bw;
    }
  }
  const int rc_out = device_bytes_to_sink((const char *)d_out.p, n_out, out, timing, []() {});
  reaper.join();
  PM_TRY(rc_out);
  lap("to the host + write");
  return PM_OK;
}

// The output must not be one of the inputs: they are mapped, and emptying a mapped file pulls its pages from under the readers
// (SIGBUS -- in a resident `serve` process the end of the worker).
static int output_is_no_input(const char *out_maf, const char *maf_a, const char *maf_b, const char *who) {
  struct stat so, si;
  if(stat(out_maf, &so) == 0) {
    for(const char *in : {maf_a, maf_b}) {
      if(stat(in, &si) == 0 && si.st_dev == so.st_dev && si.st_ino == so.st_ino) {
        return fail(PM_E_INVALID, std::string(who) + ": the output " + out_maf + " is one of the input files");
      }
    }
  }
  return PM_OK;
}

extern "C" int pm_dp_align_maf(const char *maf_a, const char *maf_b, const pm_dp_params_t *params, const char *out_maf, int device) {
  return pm::guarded("pm_dp_align_maf", [&]() -> int {
  if(!maf_a || !maf_b || !params || !out_maf) {
    return fail(PM_E_INVALID, "pm_dp_align_maf: null argument");
  }
  PM_TRY(use_device(device));
  const bool timing = pm::timing_on();
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char *what) {
    if(timing) {
      const double t1 = now();
      fprintf(stderr, "pm_dp_align_maf: %-28s %.3f s\n", what, t1 - t0);
      t0 = t1;
    }
  };
  // The inputs first: a call that cannot even map them leaves an output file that is already there as it is.
  MafDpBlocks A, B;
  PM_TRY(maf_map(maf_a, A));
  PM_TRY(maf_map(maf_b, B));
  PM_TRY(output_is_no_input(out_maf, maf_a, maf_b, "pm_dp_align_maf"));
  // The output file is opened beside everything below, WITHOUT truncation: it is emptied only when the inputs have been indexed and
  // found to pair up and the DP has run -- right before the first byte is written (emptying a big file that is already there takes
  // as long as the DP; a failure before that point leaves it untouched).
  FILE *f = nullptr;
  JoinThread opener([&]() {
    const int fd = open(out_maf, O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
    if(fd >= 0) {
      f = fdopen(fd, "wb");
      if(!f) {
        close(fd);
      }
    }
  });
  struct CloseUnlessKept {
    JoinThread &t;
    FILE *&f;
    bool keep = false;
    ~CloseUnlessKept() {
      t.join();
      if(f && !keep) {
        fclose(f);
      }
    }
  } close_unless_kept{opener, f};
  lap("files mapped");
  // the files' bytes go to the device as they are, while their lines are being indexed (four threads: two copies, two indexers,
  // each indexer with its range threads)
  std::unique_ptr<MafSideDev> sent_a(new MafSideDev()), sent_b(new MafSideDev());
  {
    int rc4[4] = {PM_OK, PM_OK, PM_OK, PM_OK};
    std::string msg4[4];
    auto guarded = [&](int k, const std::function<int()> &fn) {
      rc4[k] = fn();
      if(rc4[k]) {
        msg4[k] = pm_last_error();
      }
    };
    JoinThread t1([&]() { guarded(1, [&]() { PM_TRY(use_device(device)); return sent_a->upload_text((const uint8_t *)A.bytes, A.n_bytes); }); });
    JoinThread t2([&]() { guarded(2, [&]() { PM_TRY(use_device(device)); return sent_b->upload_text((const uint8_t *)B.bytes, B.n_bytes); }); });
    JoinThread t3([&]() { guarded(3, [&]() { return maf_index(maf_b, B); }); });
    guarded(0, [&]() { return maf_index(maf_a, A); });
    t1.join_and_rethrow();
    t2.join_and_rethrow();
    t3.join_and_rethrow();
    for(int k : {0, 3, 1, 2}) {
      if(rc4[k]) {
        return fail(rc4[k], msg4[k]);
      }
    }
  }
  if(A.block_row.size() != B.block_row.size()) {
    return fail(PM_E_INVALID, "pm_dp_align_maf: the two MAF files must hold the same number of blocks (pair k = block k of each)");
  }
  const int64_t n = (int64_t)A.block_row.size() - 1;
  lap("indexed, and the bytes on the device");
  opener.join_and_rethrow();
  if(!f) {
    return fail(PM_E_IO, std::string("cannot write ") + out_maf);
  }
  close_unless_kept.keep = true; // closed below, with its error checked
  lap("output file opened");
  int rc = align_maf_to_sink(A, B, 0, n, params, device, true, OutSink(f), lap, std::move(sent_a), std::move(sent_b), [&]() -> int {
    if(ftruncate(fileno(f), 0) != 0 && errno != EINVAL) { // (EINVAL: not a regular file -- a pipe, /dev/stdout: nothing to empty)
      return fail(PM_E_IO, std::string("cannot empty ") + out_maf + ": " + strerror(errno));
    }
    return PM_OK;
  });
  std::string msg = rc ? pm_last_error() : "";
  lap("small device buffers released");
  if(fclose(f) != 0 && !rc) {
    rc = fail(PM_E_IO, std::string("cannot write ") + out_maf);
  }
  else if(rc) {
    rc = fail(rc, msg);
  }
  if(rc) {
    return rc;
  }
  lap("write");
  return PM_OK;
  });
}

extern "C" int pm_dp_align_blocks(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a, const uint8_t *text_b,
                                  const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs,
                                  const pm_dp_params_t *params, int32_t *scores, int32_t *merged_columns, uint8_t *out_text, int64_t out_capacity,
                                  int64_t *out_off, int device) {
  return pm::guarded("pm_dp_align_blocks", [&]() -> int {
  if(!row_off_a || !row_off_b || !block_row_a || !block_row_b || !params || !scores || !merged_columns || !out_off || n_pairs < 0 ||
     (!out_text && out_capacity > 0)) {
    return fail(PM_E_INVALID, "pm_dp_align_blocks: null argument");
  }
  PM_TRY(use_device(device));
  std::vector<int32_t> s, n_ops;
  std::vector<uint8_t> merged;
  std::vector<int64_t> off;
  PM_TRY(align_blocks_core(text_a, row_off_a, n_rows_a, block_row_a, text_b, row_off_b, n_rows_b, block_row_b, n_pairs, params, device, s, n_ops, merged,
                           off, [](const char *) {}));
  memcpy(out_off, off.data(), (size_t)(n_pairs + 1) * 8);
  if(n_pairs > 0) {
    memcpy(scores, s.data(), (size_t)n_pairs * 4);
    memcpy(merged_columns, n_ops.data(), (size_t)n_pairs * 4);
  }
  if(off[(size_t)n_pairs] > out_capacity) {
    return fail(PM_E_INVALID, "pm_dp_align_blocks: out_text holds " + std::to_string(out_capacity) + " bytes, the merged blocks need " +
                                  std::to_string(off[(size_t)n_pairs]) + " (at most (rows of A + rows of B) x (columns of A + columns of B) per pair)");
  }
  if(off[(size_t)n_pairs] > 0) {
    memcpy(out_text, merged.data(), (size_t)off[(size_t)n_pairs]);
  }
  return PM_OK;
  });
}

// ------------------------------------------------------------------ several devices (multi.hpp)

// The slices of a list of block pairs over n_devices workers, cut by cells (multi.hpp): a block has as many columns as its first row
// has bytes.
static void block_pair_cuts(const int64_t *row_off_a, const int64_t *block_row_a, const int64_t *row_off_b, const int64_t *block_row_b, int64_t n,
                            int n_devices, std::vector<int64_t> &cuts) {
  std::vector<int64_t> weight((size_t)n);
  for(int64_t k = 0; k < n; ++k) {
    const int64_t ra = block_row_a[k], rb = block_row_b[k];
    const int64_t la = ra < block_row_a[k + 1] ? row_off_a[ra + 1] - row_off_a[ra] : 0, lb = rb < block_row_b[k + 1] ? row_off_b[rb + 1] - row_off_b[rb] : 0;
    weight[(size_t)k] = pair_weight(la, lb);
  }
  partition_weighted(weight.data(), n, n_devices, cuts);
}

// pack -> DP -> expansion of the pairs' contiguous slices on their devices; worker w's merged texts stay in merged[w], its scores
// and merged widths go to their places in the whole job's arrays, part_off[w] = the slice's own text offsets (n_w + 1 values).
static int align_blocks_multi_core(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a,
                                   const uint8_t *text_b, const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n,
                                   const pm_dp_params_t *params, const int *devices, int n_devices, std::vector<int32_t> &scores,
                                   std::vector<int32_t> &n_ops, std::vector<std::vector<uint8_t> > &merged,
                                   std::vector<std::vector<int64_t> > &part_off, std::vector<int64_t> &cuts) {
  PM_TRY(check_blocks(row_off_a, n_rows_a, block_row_a, n, "pm_dp_align (A)"));
  PM_TRY(check_blocks(row_off_b, n_rows_b, block_row_b, n, "pm_dp_align (B)"));
  scores.assign((size_t)n, 0);
  n_ops.assign((size_t)n, 0);
  merged.assign((size_t)n_devices, std::vector<uint8_t>());
  part_off.assign((size_t)n_devices, std::vector<int64_t>(1, 0));
  block_pair_cuts(row_off_a, block_row_a, row_off_b, block_row_b, n, n_devices, cuts);
  return run_on_devices(devices, n_devices, [&](int w, int device) {
    const int64_t lo = cuts[(size_t)w], hi = cuts[(size_t)w + 1];
    if(hi <= lo) {
      return (int)PM_OK;
    }
    BlockSlice sa(text_a, row_off_a, block_row_a, lo, hi), sb(text_b, row_off_b, block_row_b, lo, hi);
    std::vector<int32_t> s, m;
    PM_TRY(align_blocks_core(sa.text, sa.row_off.data(), sa.n_rows, sa.block_row.data(), sb.text, sb.row_off.data(), sb.n_rows, sb.block_row.data(),
                             hi - lo, params, device, s, m, merged[(size_t)w], part_off[(size_t)w], [](const char *) {}));
    memcpy(scores.data() + lo, s.data(), (size_t)(hi - lo) * 4);
    memcpy(n_ops.data() + lo, m.data(), (size_t)(hi - lo) * 4);
    return (int)PM_OK;
  });
}

extern "C" int pm_dp_align_blocks_multi(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a,
                                        const uint8_t *text_b, const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b,
                                        int64_t n_pairs, const pm_dp_params_t *params, const int *devices, int n_devices, int32_t *scores,
                                        int32_t *merged_columns, uint8_t *out_text, int64_t out_capacity, int64_t *out_off) {
  return pm::guarded("pm_dp_align_blocks_multi", [&]() -> int {
  if(!row_off_a || !row_off_b || !block_row_a || !block_row_b || !params || !scores || !merged_columns || !out_off || n_pairs < 0 ||
     (!out_text && out_capacity > 0)) {
    return fail(PM_E_INVALID, "pm_dp_align_blocks_multi: null argument");
  }
  PM_TRY(check_devices(devices, n_devices, "pm_dp_align_blocks_multi"));
  std::vector<int32_t> s, m;
  std::vector<std::vector<uint8_t> > merged;
  std::vector<std::vector<int64_t> > part_off;
  std::vector<int64_t> cuts;
  PM_TRY(align_blocks_multi_core(text_a, row_off_a, n_rows_a, block_row_a, text_b, row_off_b, n_rows_b, block_row_b, n_pairs, params, devices,
                                 n_devices, s, m, merged, part_off, cuts));
  // the host-side gather: the slices' texts back to back, in pair order
  out_off[0] = 0;
  for(int w = 0; w < n_devices; ++w) {
    const int64_t lo = cuts[(size_t)w], hi = cuts[(size_t)w + 1];
    for(int64_t k = lo; k < hi; ++k) {
      out_off[k + 1] = out_off[lo] + part_off[(size_t)w][(size_t)(k - lo) + 1];
    }
  }
  if(n_pairs > 0) {
    memcpy(scores, s.data(), (size_t)n_pairs * 4);
    memcpy(merged_columns, m.data(), (size_t)n_pairs * 4);
  }
  if(out_off[n_pairs] > out_capacity) {
    return fail(PM_E_INVALID, "pm_dp_align_blocks_multi: out_text holds " + std::to_string(out_capacity) + " bytes, the merged blocks need " +
                                  std::to_string(out_off[n_pairs]));
  }
  for(int w = 0; w < n_devices; ++w) {
    const int64_t lo = cuts[(size_t)w], hi = cuts[(size_t)w + 1];
    const int64_t bytes = out_off[hi] - out_off[lo];
    if(bytes > 0) {
      memcpy(out_text + out_off[lo], merged[(size_t)w].data(), (size_t)bytes);
    }
  }
  return PM_OK;
  });
}

extern "C" int pm_dp_align_maf_multi(const char *maf_a, const char *maf_b, const pm_dp_params_t *params, const char *out_maf, const int *devices,
                                     int n_devices) {
  return pm::guarded("pm_dp_align_maf_multi", [&]() -> int {
  if(!maf_a || !maf_b || !params || !out_maf) {
    return fail(PM_E_INVALID, "pm_dp_align_maf_multi: null argument");
  }
  PM_TRY(check_devices(devices, n_devices, "pm_dp_align_maf_multi"));
  MafDpBlocks A, B;
  PM_TRY(parse_two_mafs(maf_a, maf_b, A, B, "pm_dp_align_maf_multi"));
  PM_TRY(output_is_no_input(out_maf, maf_a, maf_b, "pm_dp_align_maf_multi"));
  const int64_t n = (int64_t)A.block_row.size() - 1;
  // every worker assembles the file bytes of its slice's merged blocks on its device and brings them to a buffer of its own; the
  // host-side gather is writing the buffers in pair order
  std::vector<std::string> part((size_t)n_devices);
  std::vector<int64_t> cuts; // by cells: a block has as many columns as its first row has bytes (row_len)
  {
    std::vector<int64_t> weight((size_t)n);
    for(int64_t k = 0; k < n; ++k) {
      const int64_t ra = A.block_row[(size_t)k], rb = B.block_row[(size_t)k];
      weight[(size_t)k] = pair_weight(ra < A.block_row[(size_t)k + 1] ? A.row_len[(size_t)ra] : 0, rb < B.block_row[(size_t)k + 1] ? B.row_len[(size_t)rb] : 0);
    }
    partition_weighted(weight.data(), n, n_devices, cuts);
  }
  PM_TRY(run_on_devices(devices, n_devices, [&](int w, int device) {
    const int64_t lo = cuts[(size_t)w], hi = cuts[(size_t)w + 1];
    return align_maf_to_sink(A, B, lo, hi, params, device, w == 0, OutSink(&part[(size_t)w]), [](const char *) {});
  }));
  FILE *f = fopen(out_maf, "wb");
  if(!f) {
    return fail(PM_E_IO, std::string("cannot write ") + out_maf);
  }
  bool ok = true;
  for(int w = 0; w < n_devices; ++w) {
    ok = ok && (part[(size_t)w].empty() || fwrite(part[(size_t)w].data(), 1, part[(size_t)w].size(), f) == part[(size_t)w].size());
  }
  if(fclose(f) != 0 || !ok) {
    return fail(PM_E_IO, std::string("cannot write ") + out_maf);
  }
  return PM_OK;
  });
}
