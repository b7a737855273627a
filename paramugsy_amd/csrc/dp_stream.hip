// dp_stream.hip -- the profile DP fed from host memory: the columns of a batch go up in segments on one HIP stream while the
// fill kernel already runs on the segments that have arrived, and the results come back on a third stream (gfx950).
//
// NO REFERENCE COUNTERPART.  pm_dp_batch_* keeps a batch resident in HBM (that is what bench.py's `value`
// measures); a caller that starts from host buffers pays upload -> kernels -> download one after the other there.  Here one
// reusable batch is loaded in `segments` pieces of consecutive pairs:
//     upload stream    offsets, then per segment its columns + the column-statistics kernel + an event
//     compute stream   dp_fill_kernel per segment, each behind its segment's event; then ONE path kernel per workspace chunk
//                      (its latency-bound chain is paid once per chunk, not once per segment)
//     download stream  scores, paths and path lengths into the caller's arrays (same layout as pm_dp_batch_fetch)
// The kernel variant (int8 or int16 column weights) depends on the largest counts in the data, which only the device sees:
// it is chosen from the first segment's statistics and checked against the whole batch's once the last segment is up; a batch
// whose later segments break the choice is simply run again (rare: the counts of a batch's profiles are alike).
// Copies are asynchronous only from / to pinned host memory (pm_dp_host_alloc).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <functional>
#include <new>
#include <vector>

#include "dp_batch.hpp"
#include "multi.hpp"

using namespace pm;

struct pm_dp_stream {
  int device = 0;
  int segments = 4;
  pm_dp_batch *b = nullptr;
  hipStream_t up = nullptr, comp = nullptr, down = nullptr;
  hipEvent_t ev_comp = nullptr;
  int *host_words = nullptr; // pinned: [0..7] statistics of the first segment, [8] the fill kernel's pipe error
  // pm_dp_stream_align_text: the two sides' row texts and tables in device memory (grow only)
  DevBuf t_text[2], t_row_off[2], t_block_row[2], t_col_off[2];
  ~pm_dp_stream() {
    delete b;
    if(ev_comp) {
      (void)hipEventDestroy(ev_comp);
    }
    for(hipStream_t s : {up, comp, down}) {
      if(s) {
        (void)hipStreamDestroy(s);
      }
    }
    if(host_words) {
      (void)hipHostFree(host_words);
    }
  }
};

extern "C" {

int pm_dp_host_alloc(void **ptr, int64_t bytes) {
  if(!ptr || bytes < 0) {
    return fail(PM_E_INVALID, "pm_dp_host_alloc: bad argument");
  }
  *ptr = nullptr;
  PM_HIP(hipHostMalloc(ptr, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault));
  return PM_OK;
}

void pm_dp_host_free(void *ptr) {
  if(ptr) {
    (void)hipHostFree(ptr);
  }
}

int pm_dp_stream_create(const pm_dp_params_t *params, int32_t segments, int64_t workspace_bytes, int device, pm_dp_stream_t **out) {
  return pm_dp_stream_create_opt(params, nullptr, segments, workspace_bytes, device, out);
}

int pm_dp_stream_create_opt(const pm_dp_params_t *params, const pm_dp_options_t *options, int32_t segments, int64_t workspace_bytes, int device,
                            pm_dp_stream_t **out) {
  if(!out) {
    return fail(PM_E_INVALID, "pm_dp_stream_create: null out");
  }
  *out = nullptr;
  if(!params || segments < 1) {
    return fail(PM_E_INVALID, "pm_dp_stream_create: bad argument");
  }
  PM_TRY(use_device(device));
  PM_TRY(dp_batch_check_params(params));
  pm_dp_stream *s = new(std::nothrow) pm_dp_stream();
  if(!s) {
    return fail(PM_E_INVALID, "out of host memory");
  }
  s->device = device;
  s->segments = segments;
  int rc = PM_OK;
  auto hipok = [&](hipError_t e, const char *what) {
    if(e != hipSuccess && !rc) {
      rc = fail(PM_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
  };
  // The copy streams at priorities of their own: the runtime multiplexes a process's streams onto a few hardware queues per
  // priority level, in creation order, and a stream whose queue it shares waits behind its packets.  With the batch's two extra
  // fill streams in the normal pool, the third chunk's fill kernel sat behind the upload stream's last packet -- 133 ms into a call
  // whose columns it needed had arrived at 62 (profiles/r04_stream.txt).  The upload stream at the HIGHEST: its small kernels (the
  // column statistics, the packing of row texts) stand between a segment's copies and its event, and at the lowest they waited tens
  // of milliseconds for a free slot beside the fill kernels, the uploads behind them.  The download stream carries copies only.
  int prio_least = 0, prio_greatest = 0;
  hipok(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest), "hipDeviceGetStreamPriorityRange");
  hipok(hipStreamCreateWithPriority(&s->up, hipStreamNonBlocking, prio_greatest), "hipStreamCreate");
  hipok(hipStreamCreateWithFlags(&s->comp, hipStreamNonBlocking), "hipStreamCreate");
  hipok(hipStreamCreateWithPriority(&s->down, hipStreamNonBlocking, prio_least), "hipStreamCreate");
  hipok(hipEventCreateWithFlags(&s->ev_comp, hipEventDisableTiming), "hipEventCreate");
  hipok(hipHostMalloc((void **)&s->host_words, 16 * sizeof(int), hipHostMallocDefault), "hipHostMalloc");
  s->b = new(std::nothrow) pm_dp_batch();
  if(!s->b && !rc) {
    rc = fail(PM_E_INVALID, "out of host memory");
  }
  if(!rc) {
    rc = dp_batch_init(s->b, params, workspace_bytes, device, options);
  }
  if(rc) {
    delete s;
    return rc;
  }
  *out = s;
  return PM_OK;
}

} // extern "C"

// The engine behind pm_dp_stream_align and pm_dp_stream_align_text: `load` enqueues the batch's segments on the upload stream
// (dp_batch_load_segments*), the rest is the same.
static int stream_align_core(pm_dp_stream_t *s, const std::function<int()> &load, int64_t n_pairs, int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  pm_dp_batch *b = s->b;
  const int traceback = ops != nullptr;
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
      fprintf(stderr, "[pm]   stream align: %-44s %.4f s\n", what, t - lap_t);
      lap_t = t;
    }
  };
  int rc = load();
  lap("segments enqueued");
  auto drain = [&]() {
    for(hipStream_t st : {s->up, s->comp, s->down}) {
      hipError_t e = hipStreamSynchronize(st);
      if(e != hipSuccess && !rc) {
        rc = fail(PM_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
      }
    }
    if(b->path_stream) {
      (void)hipStreamSynchronize(b->path_stream);
    }
  };
  auto run_and_fetch = [&]() {
    int r = dp_run(b, s->comp, traceback, nullptr, nullptr);
    if(r) {
      return r;
    }
    PM_HIP(hipEventRecord(s->ev_comp, s->comp));
    // A batch of several chunks that came in segments keeps the caller's order, so a chunk's results are one range of each array:
    // they leave behind the chunk's path kernel, beside the kernels of the chunks after it (the headline batch: 820 MB, 14 ms at the
    // end of the call otherwise).
    const size_t nc = b->chunk_tb.size();
    if(traceback && nc > 1 && b->path_stream && !b->seg_first.empty() && b->ev_path.size() >= nc && b->chunk_first.size() == nc + 1) {
      for(size_t c = 0; c < nc; ++c) {
        const i64 lo = b->chunk_first[c], hi = b->chunk_first[c + 1];
        if(hi <= lo) {
          continue;
        }
        PM_HIP(hipStreamWaitEvent(s->down, b->ev_path[c], 0));
        PM_HIP(hipMemcpyAsync(scores + lo, (const int *)b->scores.p + lo, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, s->down));
        PM_HIP(hipMemcpyAsync(n_ops + lo, (const int *)b->n_ops.p + lo, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, s->down));
        const i64 o_lo = b->off_a[(size_t)lo] + b->off_b[(size_t)lo], o_hi = b->off_a[(size_t)hi] + b->off_b[(size_t)hi];
        if(o_hi > o_lo) {
          PM_HIP(hipMemcpyAsync(ops + o_lo, (const unsigned char *)b->ops.p + o_lo, (size_t)(o_hi - o_lo), hipMemcpyDeviceToHost, s->down));
        }
      }
      PM_HIP(hipStreamWaitEvent(s->down, s->ev_comp, 0));
    }
    else {
      PM_HIP(hipStreamWaitEvent(s->down, s->ev_comp, 0));
      PM_HIP(hipMemcpyAsync(scores, b->scores.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, s->down));
      if(traceback) {
        PM_HIP(hipMemcpyAsync(n_ops, b->n_ops.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, s->down));
        if(b->total_a + b->total_b > 0) {
          PM_HIP(hipMemcpyAsync(ops, b->ops.p, (size_t)(b->total_a + b->total_b), hipMemcpyDeviceToHost, s->down));
        }
      }
    }
    PM_HIP(hipMemcpyAsync(&s->host_words[8], b->pipe_error.p, 4, hipMemcpyDeviceToHost, s->down));
    return (int)PM_OK;
  };
  if(!rc) {
    // the layout -- order, chunks, tiers: from the lengths alone -- while the first segment is on its way (8 ms of host time for the
    // 100 000 ragged pairs of bench.py's c2, which stood between the segment's arrival and the first launch)
    rc = dp_batch_plan_layout(b, s->comp);
  }
  lap("layout");
  if(!rc) {
    // the first segment's statistics choose the kernel variant; its fill kernel starts while the other segments are on their way
    hipError_t e = hipEventSynchronize(b->ev_seg[0]);
    if(e != hipSuccess) {
      rc = fail(PM_E_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    }
  }
  lap("first segment arrived");
  if(!rc) {
    rc = dp_batch_plan_variant(b, s->host_words);
  }
  lap("variant (first segment's statistics)");
  bool first_dot4 = false, first_uni = false;
  int first_rows = 0;
  if(!rc) {
    first_dot4 = b->dot4;
    first_uni = b->uni;
    first_rows = b->params.rows_a;
    rc = run_and_fetch();
  }
  lap("kernels and downloads enqueued");
  if(!rc) {
    // the whole batch's statistics: refuse what pm_dp_batch_create refuses, and run again if they change the variant
    hipError_t e = hipStreamSynchronize(s->up);
    if(e != hipSuccess) {
      rc = fail(PM_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
  }
  if(!rc) {
    rc = dp_batch_plan_variant(b, dp_batch_final_stats(b));
    if(!rc && (b->dot4 != first_dot4 || b->uni != first_uni || b->params.rows_a != first_rows)) {
      drain();
      if(!rc) {
        rc = run_and_fetch();
      }
    }
  }
  lap("uploads done, variant checked (whole batch's statistics)");
  drain();
  lap("drained");
  if(!rc && s->host_words[8]) {
    (void)dp_clear_pipe_error(b); // reported: the stream's batch is reused by the next call
    s->host_words[8] = 0;
    rc = fail(PM_E_HIP, "dp_fill_kernel: a stripe timed out waiting for its left neighbour (results invalid)");
  }
  return rc;
}

extern "C" {

int pm_dp_stream_align(pm_dp_stream_t *s, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                       int64_t n_pairs, int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  if(!s || n_pairs < 0 || !off_a || !off_b || !scores) {
    return fail(PM_E_INVALID, "pm_dp_stream_align: null argument");
  }
  if((ops == nullptr) != (n_ops == nullptr)) {
    return fail(PM_E_INVALID, "pm_dp_stream_align: ops and n_ops go together");
  }
  PM_TRY(use_device(s->device));
  if(n_pairs == 0) {
    return PM_OK;
  }
  return stream_align_core(s, [&]() { return dp_batch_load_segments(s->b, cols_a, off_a, cols_b, off_b, n_pairs, s->segments, s->host_words, s->up); },
                           n_pairs, scores, ops, n_ops);
}

// Row texts in instead of packed columns: pair k = block k of each side (the flat description of pm_dp_pack_maf).  The texts go up
// in the same segments, each packed on the device by pm_dp_pack_maf's kernel as soon as it has arrived -- the packed columns
// (8 bytes per column, whatever the number of rows) never cross the link: a 2-row profile moves a quarter of the bytes.
int pm_dp_stream_align_text(pm_dp_stream_t *s, const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a,
                            const uint8_t *text_b, const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs,
                            int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  return guarded("pm_dp_stream_align_text", [&]() -> int {
  if(!s || n_pairs < 0 || !row_off_a || !row_off_b || !block_row_a || !block_row_b || !scores) {
    return fail(PM_E_INVALID, "pm_dp_stream_align_text: null argument");
  }
  if((ops == nullptr) != (n_ops == nullptr)) {
    return fail(PM_E_INVALID, "pm_dp_stream_align_text: ops and n_ops go together");
  }
  PM_TRY(use_device(s->device));
  PM_TRY(dp_check_blocks(row_off_a, n_rows_a, block_row_a, n_pairs, "pm_dp_stream_align_text (A)"));
  PM_TRY(dp_check_blocks(row_off_b, n_rows_b, block_row_b, n_pairs, "pm_dp_stream_align_text (B)"));
  if(n_pairs == 0) {
    return PM_OK;
  }
  if((row_off_a[n_rows_a] > 0 && !text_a) || (row_off_b[n_rows_b] > 0 && !text_b)) {
    return fail(PM_E_INVALID, "pm_dp_stream_align_text: null text");
  }
  const uint8_t *text[2] = {text_a, text_b};
  const int64_t *row_off[2] = {row_off_a, row_off_b}, *block_row[2] = {block_row_a, block_row_b};
  const int64_t n_rows[2] = {n_rows_a, n_rows_b};
  // a block has as many columns as its first row has bytes
  std::vector<int64_t> col_off[2];
  for(int sd = 0; sd < 2; ++sd) {
    col_off[sd].assign((size_t)n_pairs + 1, 0);
    for(int64_t k = 0; k < n_pairs; ++k) {
      const int64_t r = block_row[sd][k];
      col_off[sd][(size_t)k + 1] = col_off[sd][(size_t)k] + (r < block_row[sd][k + 1] ? row_off[sd][r + 1] - row_off[sd][r] : 0);
    }
  }
  auto grow = [](DevBuf &buf, size_t bytes) { return buf.bytes >= bytes ? (int)PM_OK : buf.alloc(bytes + bytes / 8); };
  auto load = [&]() {
    for(int sd = 0; sd < 2; ++sd) { // the tables first (small): blocking copies from the caller's pageable arrays
      PM_TRY(grow(s->t_text[sd], (size_t)row_off[sd][n_rows[sd]] + 16));
      PM_TRY(grow(s->t_row_off[sd], (size_t)(n_rows[sd] + 1) * 8));
      PM_TRY(grow(s->t_block_row[sd], (size_t)(n_pairs + 1) * 8));
      PM_TRY(grow(s->t_col_off[sd], (size_t)(n_pairs + 1) * 8));
      PM_HIP(hipMemcpyAsync(s->t_row_off[sd].p, row_off[sd], (size_t)(n_rows[sd] + 1) * 8, hipMemcpyHostToDevice, s->up));
      PM_HIP(hipMemcpyAsync(s->t_block_row[sd].p, block_row[sd], (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, s->up));
      PM_HIP(hipMemcpyAsync(s->t_col_off[sd].p, col_off[sd].data(), (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, s->up));
    }
    PM_HIP(hipStreamSynchronize(s->up)); // col_off's vectors and the caller's tables may be pageable
    return dp_batch_load_segments_from(
        s->b, col_off[0].data(), col_off[1].data(), n_pairs, s->segments, s->host_words, s->up,
        [&](int sd, i64 lo, i64 hi, i64 c0, i64 c1, hipStream_t st) {
          // the texts of the segment's rows (contiguous), then one thread per column of the segment
          const int64_t t0 = row_off[sd][block_row[sd][lo]], t1 = row_off[sd][block_row[sd][hi]];
          if(t1 > t0) {
            PM_HIP(hipMemcpyAsync((char *)s->t_text[sd].p + t0, text[sd] + t0, (size_t)(t1 - t0), hipMemcpyHostToDevice, st));
          }
          return dp_pack_launch(c0, c1 - c0, n_pairs, (const i64 *)s->t_col_off[sd].p, (const i64 *)s->t_block_row[sd].p,
                                (const i64 *)s->t_row_off[sd].p, (const unsigned char *)s->t_text[sd].p,
                                (u64 *)(sd == 0 ? s->b->cols_a.p : s->b->cols_b.p), st);
        },
        true, true);
  };
  return stream_align_core(s, load, n_pairs, scores, ops, n_ops);
  });
}

void pm_dp_stream_destroy(pm_dp_stream_t *s) {
  if(!s) {
    return;
  }
  (void)hipSetDevice(s->device);
  delete s;
}

// The same over several devices of one node (multi.hpp): the pairs are cut into n_devices contiguous slices, every worker feeds its
// slice to its device through a pm_dp_stream of its own and writes its results straight into the caller's arrays at the slice's
// place -- pair k's scores[k], n_ops[k] and ops slot are where pm_dp_stream_align / pm_dp_batch_fetch put them, so the "gather"
// is the layout itself.  Workers that share a device (the same index named twice) share its workspace budget.
int pm_dp_align_multi(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                      const pm_dp_params_t *params, const int *devices, int n_devices, int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  return guarded("pm_dp_align_multi", [&]() -> int {
  if(n_pairs < 0 || !off_a || !off_b || !params || !scores || (n_pairs > 0 && (!cols_a || !cols_b))) {
    return fail(PM_E_INVALID, "pm_dp_align_multi: null argument");
  }
  if((ops == nullptr) != (n_ops == nullptr)) {
    return fail(PM_E_INVALID, "pm_dp_align_multi: ops and n_ops go together");
  }
  if(off_a[0] != 0 || off_b[0] != 0) {
    return fail(PM_E_INVALID, "pm_dp_align_multi: offsets must start at 0");
  }
  PM_TRY(check_devices(devices, n_devices, "pm_dp_align_multi"));
  PM_TRY(dp_batch_check_params(params));
  std::vector<int64_t> cuts; // slices of equal cells, not of equal counts (multi.hpp)
  {
    std::vector<int64_t> weight((size_t)n_pairs);
    for(int64_t k = 0; k < n_pairs; ++k) {
      if(off_a[k + 1] < off_a[k] || off_b[k + 1] < off_b[k]) {
        return fail(PM_E_INVALID, "pm_dp_align_multi: bad profile length");
      }
      weight[(size_t)k] = pair_weight(off_a[k + 1] - off_a[k], off_b[k + 1] - off_b[k]);
    }
    partition_weighted(weight.data(), n_pairs, n_devices, cuts);
  }
  return run_on_devices(devices, n_devices, [&](int w, int device) {
    const int64_t lo = cuts[(size_t)w], hi = cuts[(size_t)w + 1];
    if(hi <= lo) {
      return (int)PM_OK;
    }
    int sharing = 0;
    for(int k = 0; k < n_devices; ++k) {
      sharing += devices[k] == device;
    }
    int64_t workspace = 0; // the library's default (dp_default_budget_bytes)
    if(sharing > 1) {
      workspace = dp_default_budget_bytes() / sharing;
    }
    std::vector<int64_t> oa((size_t)(hi - lo) + 1), ob((size_t)(hi - lo) + 1);
    for(int64_t k = lo; k <= hi; ++k) {
      oa[(size_t)(k - lo)] = off_a[k] - off_a[lo];
      ob[(size_t)(k - lo)] = off_b[k] - off_b[lo];
    }
    pm_dp_stream_t *st = nullptr;
    PM_TRY(pm_dp_stream_create(params, 4, workspace, device, &st));
    int rc = pm_dp_stream_align(st, cols_a + off_a[lo] * 8, oa.data(), cols_b + off_b[lo] * 8, ob.data(), hi - lo, scores + lo,
                                ops ? ops + off_a[lo] + off_b[lo] : nullptr, n_ops ? n_ops + lo : nullptr);
    std::string msg = rc ? pm_last_error() : "";
    pm_dp_stream_destroy(st);
    return rc ? fail(rc, msg) : (int)PM_OK;
  });
  });
}

} // extern "C"
