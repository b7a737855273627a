// dp_batch.hpp -- the host-side state of a batch of profile pairs (pm_dp_batch_t), shared by dp_kernels.hip (create / run /
// fetch) and dp_stream.hip (the same state reused slice after slice with uploads, kernels and downloads on three streams).
#pragma once

#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <vector>

#include "dp_internal.hpp"
#include "pm_internal.hpp"

// A fill launch that takes its work from a queue of tiles (dp_fill_tiles_kernel): the list, made once per layout for the launch of
// the pairs at positions [first, first + n), and what its wavefronts share while they run
struct DpTilePlan {
  pm::i64 first = 0, n = 0;
  int tile_steps = 0;
  pm::i64 n_tiles = 0, total_stripes = 0;
  pm::DevBuf tiles, sync_off, sync_words, state;
};

struct pm_dp_batch {
  int device = 0;
  pm_dp_options_t opt = {}; // how this batch is run (all zero: chosen per batch); copied at dp_batch_init
  pm::i64 n_pairs = 0, total_a = 0, total_b = 0;
  std::vector<pm::i64> off_a, off_b;
  pm::DevBuf cols_a, cols_b, d_off_a, d_off_b, bnd, scores, ops, n_ops, tb, d_tb_off, d_order, stats;
  std::vector<int> order;                      // processing order of the pairs (longest first), a permutation of 0 .. n_pairs - 1
  std::vector<pm::i64> chunk_first;            // first POSITION in `order` of each chunk, plus n_pairs
  std::vector<std::vector<pm::i64> > chunk_tb; // per chunk: word offsets of the pairs at its positions
  pm::i64 tb_words_cap = 0;
  pm::i64 tb_budget_bytes = 0;
  pm::DpParamsD params;
  int max_sub_acgt = 0, max_sub_all = 0;
  pm::i64 cells = 0;
  int cols_per_lane = 16; // columns of B a lane owns per stripe: 16, 8 for small batches (dp_batch_plan); opt.cols_per_lane fixes it
  bool cols_forced = false;
  bool dot4 = false;      // all counts and ACGT weights fit int8 (opt.int16_weights forces the int16 path)
  bool uni = false;       // every column of A holds the same number of symbols: gap row folded into the weights (opt.no_uniform_depth)
  int waves_override = 0; // opt.waves_per_pair
  bool tail = false;      // the narrow last stripes of dp_internal.hpp (checkpoint mode, 16 columns per lane; opt.full_stripes disables)
  bool ckpt = true;       // paths from checkpoints + block recomputation (dp_walk.hip), or 4 stored decision bits per cell
  bool mode_auto = true;  // ckpt chosen per batch in dp_batch_plan (opt.path_mode fixes it)
  int walk_lanes = 0;     // lanes per pair of the checkpoint walk; 0 = chosen per launch; opt.walk_lanes overrides
  std::vector<std::unique_ptr<DpTilePlan> > tile_plans; // of this layout's launches (dp_batch_plan_layout forgets them)
  pm::DevBuf pipe_error;
  // progress words of pairs whose stripes run on several workgroups (dp_fill_kernel, NG): an arena sized by dp_run for all the
  // launches of a pass, every launch taking a piece of its own (launches of one pass may overlap on their streams)
  pm::DevBuf gprog;
  size_t gprog_used = 0;
  // the band of the checkpoint walk (dp_internal.hpp), set up by dp_batch_plan for one-chunk batches of few pairs
  pm::DevBuf d_band_work, d_band_off, band_bits;
  pm::i64 band_work_items = 0; // 0: no band
  int band_lanes = 0;          // the lanes per pair the band's blocks are laid out for
  hipStream_t last_stream = nullptr;
  // chunk pipeline (more than one chunk): the workspace is n_slots equal parts, chunk c uses part c % n_slots; the path kernel of
  // chunk c runs on `path_stream` beside the fill kernels of the chunks after it
  pm::i64 tb_half_words = 0;
  // where every chunk's part of the workspace starts (words).  A batch whose workspace fits the budget but is cut into chunks all
  // the same (dp_batch_plan: so that the path kernel of one chunk runs beside the fill kernel of the next) gives every chunk a
  // part of its own (slot_reuse false); a batch that does not fit reuses n_slots parts in turn
  std::vector<pm::i64> chunk_base;
  bool slot_reuse = false;
  // The longest pairs of a chunk in launches of their own (dp_batch_plan): a wavefront works through its pair's stripes one after the
  // other, so a launch lasts at least as long as its longest pair -- in a ragged launch of 12 500 pairs that pair alone took as long
  // as all the others together.  chunk_tiers[c]: positions (in `order`) where the chunk's launch is cut: the first few hundred pairs
  // run in small launches, for which dp_launch_fill chooses several wavefronts per pair, on side streams beside the launch of the rest
  std::vector<std::vector<pm::i64> > chunk_tiers;
  std::vector<hipStream_t> tier_streams;
  std::vector<hipEvent_t> tier_events;
  std::vector<pm::i64> chunk_groups; // workgroups of every chunk's fill launch (filled in by dp_run as it launches)
  pm::DevBuf fill_started;           // per chunk: workgroups of its fill kernel that have started (the gate of the next chunk's)
  pm::DevBuf tier_started;           // per chunk: workgroups of its tiers' fill kernels that have started (the gate of the launch of the rest)
  hipStream_t path_stream = nullptr;
  // the walk beside the fill kernel of its own launch (DpEarly, dp_internal.hpp): a stream for the early walkers, per workspace slot
  // the lists the fill kernel publishes into, and two events (lists zeroed; early walkers done)
  hipStream_t early_stream = nullptr;
  pm::DevBuf early_buf;
  size_t early_slot_ints = 0;
  std::vector<hipEvent_t> ev_early_ready, ev_early_done;
  // the fill kernels of every slot run on a stream of their own, so that chunk c + 1's first wavefronts take the SIMDs chunk c's
  // last ones leave (a launch ends with the chip draining: its last round of pairs fills only part of it); ev_begin orders that
  // stream behind whatever the caller's stream held when dp_run was called
  float last_fill_busy_ms = 0;           // profiled run: the time during which some fill kernel ran (launches may overlap)
  int n_slots = 3;                       // parts of the workspace that consecutive chunks use in turn (opt.slots)
  std::vector<hipStream_t> fill_streams; // for the chunks of slots 1 .. n_slots - 1 (slot 0: the caller's stream)
  hipEvent_t ev_begin = nullptr;
  std::vector<hipEvent_t> ev_fill, ev_path;                       // per chunk: fill done / path done
  std::vector<hipEvent_t> tv_fill0, tv_fill1, tv_path0, tv_path1; // timing events of the profiled run
  // ... and of the tiers' fill kernels (three pairs per chunk, on the tiers' own streams): the launches the step waits for -- without
  // them the profiled run's fill time, its fill-busy union and bench.py's roofline left the longest pairs' kernels out
  std::vector<hipEvent_t> tv_tier0, tv_tier1;
  std::vector<int> tv_tier_chunk; // which chunk each recorded pair of this pass belongs to (size = pairs recorded)
  // dp_stream.hip: the columns arrive in segments of consecutive pairs (seg_first: first pair of each, plus n_pairs); ev_seg[k]
  // fires on the upload stream when segment k is in HBM; dp_batch_plan cuts such a batch into chunks that end where segments end
  // (a small one: one chunk, which dp_run launches segment by segment), dp_run holds a chunk's launch back until its segment is up
  std::vector<pm::i64> seg_first;
  std::vector<hipEvent_t> ev_seg;
  bool seg_events_armed = false;
  // pinned host staging of a reusable batch (dp_stream.hip): offsets, workspace offsets and the column statistics
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  int host_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ~pm_dp_batch() {
    for(std::vector<hipEvent_t> *v : {&ev_fill, &ev_path, &tv_fill0, &tv_fill1, &tv_path0, &tv_path1, &ev_seg, &tv_tier0, &tv_tier1}) {
      for(hipEvent_t e : *v) {
        if(e) {
          (void)hipEventDestroy(e);
        }
      }
    }
    if(path_stream) {
      (void)hipStreamDestroy(path_stream);
    }
    if(early_stream) {
      (void)hipStreamDestroy(early_stream);
    }
    for(std::vector<hipEvent_t> *v : {&ev_early_ready, &ev_early_done}) {
      for(hipEvent_t e : *v) {
        if(e) {
          (void)hipEventDestroy(e);
        }
      }
    }
    for(hipStream_t st : fill_streams) {
      (void)hipStreamDestroy(st);
    }
    for(hipStream_t st : tier_streams) {
      (void)hipStreamDestroy(st);
    }
    for(hipEvent_t e : tier_events) {
      (void)hipEventDestroy(e);
    }
    if(ev_begin) {
      (void)hipEventDestroy(ev_begin);
    }
    if(pinned) {
      (void)hipHostFree(pinned);
    }
  }
};

namespace pm {

// The steps pm_dp_batch_create is made of (dp_kernels.hip).  A reusable batch goes reserve once, then load / plan / run per slice.
int64_t dp_default_budget_bytes();
int dp_batch_check_params(const pm_dp_params_t *params);
// options == nullptr: the process's defaults (pm_dp_set_default_options)
int dp_batch_init(pm_dp_batch *h, const pm_dp_params_t *params, int64_t tb_budget_bytes, int device, const pm_dp_options_t *options = nullptr);
int dp_clear_pipe_error(pm_dp_batch *h); // synchronous (see dp_kernels.hip)
// device buffers for up to cap_pairs pairs with cap_a / cap_b columns in all; only grows
int dp_batch_reserve(pm_dp_batch *h, i64 cap_pairs, i64 cap_a, i64 cap_b);
// offsets (rebased to 0), columns and the column statistics kernels on `stream`; the statistics land in h->host_stats once the
// stream has run that far (the copies are asynchronous when the host buffers are pinned).  off_a / off_b point at the slice's first offset.
int dp_batch_load(pm_dp_batch *h, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                  int64_t n_pairs, hipStream_t stream);
// the same in `segments` pieces of about equal column counts: after piece k, ev_seg[k] is recorded on `stream`; the statistics of
// piece 0 alone land in stats_first (pinned) behind ev_seg[0], those of the whole batch in the usual place behind the last event
int dp_batch_load_segments(pm_dp_batch *h, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                           int64_t n_pairs, int segments, int *stats_first, hipStream_t stream);
int dp_batch_load_segments_from(pm_dp_batch *h, const int64_t *off_a, const int64_t *off_b, int64_t n_pairs, int segments, int *stats_first,
                                hipStream_t stream, const std::function<int(int, i64, i64, i64, i64, hipStream_t)> &fill, bool have_a,
                                bool have_b);
// dp_batch_plan with the statistics given (dp_batch_plan itself reads the batch's own)
int dp_batch_plan_with(pm_dp_batch *h, const int *stats, hipStream_t stream);
// its two halves: what the statistics decide (validity, the kernel's arithmetic), and the layout (from the lengths and options alone)
int dp_batch_plan_variant(pm_dp_batch *h, const int *stats);
int dp_batch_plan_layout(pm_dp_batch *h, hipStream_t stream);
const int *dp_batch_final_stats(const pm_dp_batch *h);
// after the load has completed: validate against the statistics, choose the kernel variant, cut the batch into chunks of the
// workspace, send the chunks' offset table on `stream`
int dp_batch_plan(pm_dp_batch *h, hipStream_t stream);
int dp_run(pm_dp_batch *h, hipStream_t stream, int traceback, float *ms_fill, float *ms_path);
// dp_maf.hip: the block-description check of pm_dp_pack_maf, and its kernel over a range of columns on a stream
int dp_check_blocks(const int64_t *row_off, int64_t n_rows, const int64_t *block_row, int64_t n_blocks, const char *who);
int dp_pack_launch(i64 first, i64 n_cols, i64 n_blocks, const i64 *col_off, const i64 *block_row, const i64 *row_off, const unsigned char *text,
                   u64 *cols, hipStream_t stream);

} // namespace pm
