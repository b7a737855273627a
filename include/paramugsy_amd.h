/*
 * paramugsy_amd.h -- C ABI of libparamugsy_amd.so (MI355X / gfx950).
 *
 * What this boundary replaces.  In the reference the "profiles" translate stage is reached only as a
 * child process: `m_translate <left_dir> <right_dir> <nucmer_list> <out_delta>`
 * (lib/m_translate/m_translate_main.cc:10-13,19-46), whose one library entry point is
 *     void Para_mugsy::translate(left_dir, right_dir, nucmer_list, out_stream)   lib/m_translate/m_translate.hh:9-14
 * There is no OCaml `external`, ctypes stub or extern "C" anywhere in the reference, so this header is NEW
 * surface: it is what a ctypes/cgo/JNI binding of that entry point would bind.  The byte-compatible
 * drop-in is the executable bin/m_translate built on top of it (same argv, same output bytes).
 * INTEGRATION.md shows the OCaml ctypes stub and the one-line change to the task script.
 *
 * Three levels, all plain C (pointers, sizes, no C++ or torch types):
 *   1. pm_translate_files()     == Para_mugsy::translate + the two header lines of m_translate_main.cc:35-39
 *   2. pm_job_*()               a batch of (delta entry x left row x right row) work units kept resident in
 *                               HBM; one pm_job_run() is one pass of the hot path
 *                               (_translate_delta_with_profiles, lib/m_translate/m_translate.cc:625-647, over
 *                               every unit of the batch)
 *   3. pm_rows_*_batch()        batched coordinate conversions of lib/profiles_lib/m_profile.cc:91-149
 *
 * Every function returns PM_OK (0) or a negative PM_E_* code; pm_last_error() gives a message.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with PM_E_NO_DEVICE.
 */
#ifndef PARAMUGSY_AMD_H
#define PARAMUGSY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library-level return codes (negative) ---- */
#define PM_OK 0
#define PM_E_INVALID (-1)   /* bad argument (null pointer, negative size, index out of range) */
#define PM_E_NO_DEVICE (-2) /* no HIP device / HIP runtime error at init */
#define PM_E_HIP (-3)       /* a HIP call failed; see pm_last_error() */
#define PM_E_IO (-4)        /* file could not be opened / written */
#define PM_E_PARSE (-5)     /* Profile_read_error (m_profile.hh:13) / Delta_stream_parse_error (m_delta.hh:13) */
#define PM_E_UNIT (-6)      /* at least one work unit ended in a PM_ST_* failure (the reference aborts there) */
#define PM_E_MALFORMED (-7) /* a gap list is not ascending and disjoint: outside the domain of this library */

/* ---- per-unit status: 0 or the failure class the reference would have died with ---- */
#define PM_ST_OK 0
#define PM_ST_SEQ_IDX_OUT_OF_RANGE 1     /* Seq_idx_out_of_range, m_profile.cc:110 */
#define PM_ST_PROFILE_IDX_OUT_OF_RANGE 2 /* Profile_idx_out_of_range, m_profile.cc:147,162 */
#define PM_ST_IS_NONE 3                  /* Is_none_error from .value(), m_option.hh:28 */
#define PM_ST_ASSERT_GAP_BEHIND 4        /* assert(r_diff >= 0), m_translate.cc:42-43 */
#define PM_ST_ASSERT_SUB_LENGTHS 5       /* assert(...length() == ...), m_translate.cc:550-551 */
#define PM_ST_ALREADY_UNNEXT 6           /* Already_unnext_gap, m_translate.cc:74 */
#define PM_ST_STEP_LIMIT 7               /* merge did not terminate within its step budget */
#define PM_ST_OFFSET_ORDER 8             /* reserved: no longer produced (such units are merged as the reference's writer does) */
#define PM_ST_MALFORMED_INPUT 9          /* unit touches a row / entry whose gap list is not ascending+disjoint */
#define PM_ST_TEXT_RANGE 10              /* untranslate: String.sub / expand_text index outside a row's text (Invalid_argument) */

/* ---- batch description (host pointers; copied to the device by pm_job_create) ---- */

/* Row profiles of one side.  One row == one M_profile (m_profile.hh:26-100) minus its names and text. */
typedef struct pm_rows {
  int64_t n;
  const int64_t *start;     /* [n]   p_range start (1-based, may be > end: reverse strand) */
  const int64_t *end;       /* [n]   p_range end */
  const int64_t *length;    /* [n]   p_length (columns) */
  const int64_t *gap_off;   /* [n+1] CSR offsets into gap_start/gap_end */
  const int64_t *gap_start; /* [gap_off[n]] gap runs in column coordinates, ascending and disjoint per row */
  const int64_t *gap_end;
} pm_rows_t;

/* Parsed delta entries.  One == one M_delta_entry (m_delta.hh:17-62) minus its header strings. */
typedef struct pm_deltas {
  int64_t n;
  const int64_t *ref_start; /* [n] */
  const int64_t *ref_end;
  const int64_t *qry_start;
  const int64_t *qry_end;
  const int64_t *ref_gap_off; /* [n+1] */
  const int64_t *ref_gap_start;
  const int64_t *ref_gap_end;
  const int64_t *qry_gap_off; /* [n+1] */
  const int64_t *qry_gap_start;
  const int64_t *qry_gap_end;
} pm_deltas_t;

/* Work units: the (entry, left row, right row) triples the loops of m_translate.cc:698-706 visit. */
typedef struct pm_units {
  int64_t n;
  const int32_t *delta; /* [n] index into pm_deltas */
  const int32_t *left;  /* [n] index into the left pm_rows */
  const int32_t *right; /* [n] index into the right pm_rows */
} pm_units_t;

/* One emitted delta entry (what M_delta_builder::to_delta returns, m_delta_builder.cc:7-22), with its gap
 * lists already merged into the signed-offset form M_delta_stream_writer prints
 * (deltas_of_gaps, m_delta_stream_writer.hh:14-53; the terminating 0 is included). */
typedef struct pm_entry {
  int64_t ref_start;
  int64_t ref_end;
  int64_t qry_start;
  int64_t qry_end;
  int64_t offset_begin; /* index of this entry's first value in the offsets array */
  int64_t n_offsets;    /* values including the terminating 0 */
} pm_entry_t;

typedef struct pm_job pm_job_t; /* opaque; owns device memory */

const char *pm_last_error(void);
/* The file-level entries keep some buffers from call to call for a resident caller's sake: a few pinned staging pieces (32 MB a
 * set), device scratch buffers (at most 4 GiB in all, none above 1 GiB) and the DP batches of the MAF entries with their path
 * workspace (batches that have grown past 24 GiB are not kept).  This frees them all. */
int pm_release_caches(void);
int pm_device_count(void);
/* name/CU count of device `dev`; name buffer of `cap` bytes */
int pm_device_info(int dev, char *name, int cap, int *compute_units, int64_t *hbm_bytes);

/* How a translate job is run -- every field 0: chosen by the library (what production callers pass, or NULL).  Given when a job is
 * created (pm_job_create_opt, pm_job_create_from_workload_opt) or a file-level job is run (pm_translate_files_opt); the entries
 * without an options argument take the process's defaults, pm_translate_set_default_options (copied under a lock).  The library
 * reads no environment variable for any of this (round 5; rounds 1-4 read PM_TRANSLATE_WIDE, PM_TRANSLATE_LIBRARY_SCANS, PM_NO_SOA
 * and PM_TIMING inside entries that pm_translate_files_multi runs on one thread per device); the executables under bin/ and the
 * Python binding keep those names as a SPELLING and hand the library a struct. */
typedef struct pm_translate_options {
  int32_t coordinate_bits; /* 64: the int64 tables and kernels (the reference's `long`, lib/profiles_lib/m_range.hh:8) also where int would do */
  int32_t library_scans;   /* 1: the step's prefix sums through rocPRIM, as jobs above 8.4 M units take them (for tests) */
  int32_t no_side_file;    /* 1: <dir>/profiles is parsed also where a matching <dir>/profiles.soa lies beside it */
  int32_t timing;          /* 1: phase times of the file-level entries on stderr (as a process default: of every file-level entry) */
  int32_t reserved[4];
} pm_translate_options_t;
int pm_translate_set_default_options(const pm_translate_options_t *options); /* NULL: all zero again */

/* Upload a batch, build the per-gap prefix tables and validate the gap lists on the device. */
int pm_job_create(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                  int device, pm_job_t **out);
int pm_job_create_opt(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units,
                      const pm_translate_options_t *options, int device, pm_job_t **out);
/* One pass of the hot path over every unit: filter, count, scan, emit.  Asynchronous on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  Inputs and outputs stay in HBM. */
int pm_job_run(pm_job_t *job, void *hip_stream);
/* The same pass with HIP events recorded on `hip_stream` between its four phases (filter + compaction of the units
 * that survive the first overlap test, count pass, offset scans, emit pass); waits for completion and reports each
 * phase's device time in milliseconds (measurement aid for bench.py; any pointer may be NULL). */
int pm_job_run_profiled(pm_job_t *job, void *hip_stream, float *ms_filter, float *ms_count, float *ms_scan, float *ms_emit);
/* Wait for the last run and report the output sizes. */
int pm_job_sizes(pm_job_t *job, int64_t *n_entries, int64_t *n_offsets);
/* Copy results to host arrays sized from pm_job_sizes.  unit_entry_off has units.n + 1 elements: unit u owns
 * entries [unit_entry_off[u], unit_entry_off[u+1]).  Any pointer may be NULL to skip that array.
 * Returns PM_OK, or PM_E_UNIT when some unit_status is non-zero (arrays are still filled). */
int pm_job_fetch(pm_job_t *job, int32_t *unit_status, int64_t *unit_entry_off, pm_entry_t *entries, int64_t *offsets);
/* The text M_delta_stream_writer::write prints for the last run's results (lib/profiles_lib/m_delta_stream_writer.hh:55-82),
 * formatted ON THE DEVICE: a `>left right lenL lenR` line whenever the pair of major names changes, `rs re qs qe 1 2 3`, then one
 * signed offset per line.  left_major / right_major: p_major_name of every row of the two sides (NUL-terminated).  The text ends
 * with the first failing unit's partial output, as the reference's stream does when it dies there: *failed_unit = that unit's
 * index or -1, *failed_status its PM_ST_* (either pointer may be NULL).  pm_job_text sizes and formats (the bytes stay in HBM);
 * pm_job_text_fetch copies *n_bytes bytes to `out`. */
int pm_job_text(pm_job_t *job, const char *const *left_major, const char *const *right_major, int64_t *n_bytes, int64_t *failed_unit,
                int32_t *failed_status);
int pm_job_text_fetch(pm_job_t *job, char *out);
/* bytes [first, first + n) of the text (a caller that writes the text out while the rest is still on its way) */
int pm_job_text_fetch_range(pm_job_t *job, char *out, int64_t first, int64_t n);
/* Algorithmic bytes one pm_job_run moves (inputs read + outputs written), for roofline accounting. */
int pm_job_algorithmic_bytes(pm_job_t *job, int64_t *bytes);
/* Algorithmic bytes of the two heavy kernels separately (count pass, emit pass) and the number of live units (those
 * that pass the first overlap test), for per-kernel roofline accounting. */
int pm_job_kernel_bytes(pm_job_t *job, int64_t *count_bytes, int64_t *emit_bytes, int64_t *n_live);
/* Width of the coordinate arithmetic the job's kernels run in: 32 when every number in the job's tables is below 2^25
 * in magnitude (the kernels then use int tables and int registers: same results, checked, about twice the resident
 * wavefronts), else 64 (the reference's `long`, lib/profiles_lib/m_range.hh:8).  pm_translate_options_t.coordinate_bits = 64
 * forces 64. */
int pm_job_coordinate_bits(pm_job_t *job, int *bits);
/* Width the job holds SEQUENCE POSITIONS in (round 5).  A job whose positions pass 2^25 -- a chromosome, a concatenated assembly --
 * while every row's and entry's span, length and gap column stays below it keeps its positions in 64 bits and runs everything else
 * as the 32-bit job does: a position enters the arithmetic only as its difference from the start of a row or entry that contains it
 * (lib/profiles_lib/m_profile.cc:93-99), and that difference is bounded by the row's length.  Such a job reports coordinate_bits 32,
 * position_bits 64; its results are those of the reference's `long` (lib/profiles_lib/m_range.hh:8), checked like the 32-bit job's. */
int pm_job_position_bits(pm_job_t *job, int *bits);
void pm_job_destroy(pm_job_t *job);

/* Batched coordinate conversions on one side's rows (a3/a4).  `row` selects the row per query; results and
 * per-query status (PM_ST_*; IS_NONE = column is a gap) are written to host arrays. */
int pm_rows_profile_idx_of_seq_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *seq_idx,
                                         int64_t *profile_idx, int32_t *status, int device);
int pm_rows_seq_idx_of_profile_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *profile_idx,
                                         int64_t *seq_idx, int32_t *status, int device);

/* Host-only loading of a job's files (no device needed): both `profiles` files (read_profile_file,
 * m_profile.cc:15-85), every delta file (M_delta_stream, m_delta.cc:72-92,148-220), the per-sequence sorted
 * row index (_profile_map_of_dir, m_translate.cc:188-207) and the unit list of the loops at
 * m_translate.cc:666-707.  pm_workload_tables() fills views into memory owned by the workload.
 * pm_workload_load returns PM_E_PARSE with a VALID handle when a delta file is malformed part-way: the
 * entries before the failure are loaded, as the reference translates them before it throws. */
typedef struct pm_workload pm_workload_t;
int pm_workload_load(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                     pm_workload_t **out);
int pm_workload_tables(pm_workload_t *w, pm_rows_t *left, pm_rows_t *right, pm_deltas_t *deltas, pm_units_t *units);
/* A job over a loaded workload whose unit list is made ON THE DEVICE (the loops of _translate_delta, m_translate.cc:666-707: per
 * entry a binary search and an overlap scan in each side's sorted rows, a scan over the per-entry counts, one thread per unit):
 * the same list, in the same order, as pm_workload_tables hands out -- which is what the file-level entries below run on.
 * pm_job_units reports the job's unit count and copies the list out (any pointer may be NULL). */
int pm_job_create_from_workload(pm_workload_t *w, int device, pm_job_t **out);
int pm_job_create_from_workload_opt(pm_workload_t *w, const pm_translate_options_t *options, int device, pm_job_t **out);
int pm_job_units(pm_job_t *job, int64_t *n_units, int32_t *delta, int32_t *left, int32_t *right);
/* side 0 = left, 1 = right; pointers stay valid until pm_workload_destroy */
int pm_workload_row_name(pm_workload_t *w, int side, int64_t row, const char **major_name, const char **seq_name);
void pm_workload_destroy(pm_workload_t *w);

/* File level: the whole of Para_mugsy::translate plus m_translate_main.cc's two header lines.
 * delta_paths: n_paths NUL-terminated strings.  Output bytes equal the reference's. */
int pm_translate_files(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                       const char *out_path, int device);
/* The same with the two directory names of the output's first line (m_translate_main.cc:35-39 prints its argv strings) given apart
 * from the paths the files are opened by, and a device list (n_devices > 1: pm_translate_files_multi's split).  What the resident
 * worker (`mugsy_profiles serve -socket`) runs for an m_translate client in another working directory: the client's relative paths
 * are resolved against ITS directory, the output holds the strings it was started with.  New surface. */
int pm_translate_files_as(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths, const char *out_path,
                          const char *left_name, const char *right_name, const int *devices, int n_devices);
/* ... and with explicit options (NULL: the process's defaults), for this job alone, on whatever threads and devices it runs */
int pm_translate_files_opt(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths, const char *out_path,
                           const char *left_name, const char *right_name, const int *devices, int n_devices,
                           const pm_translate_options_t *options);

/* The two tools of lib/profiles_cpp that compile upstream (nothing in the reference invokes them).
 * pm_sort_delta   == m_sort_delta (lib/profiles_cpp/m_sort_delta.cc:58-91): delta text in, the same entries sorted by
 *                    (header pair, ref start, query start, ref end, query end) out; like upstream no file header is
 *                    written.  in_path/out_path NULL = stdin/stdout.
 * pm_maf_analyzer == maf_analyzer <maf> (lib/profiles_cpp/maf_analyzer.cc:12-38): per genome the ranges NOT covered
 *                    by the MAF's rows, upstream's arithmetic quirks included (maf_analyzer_missing.cc:115,119-126). */
int pm_sort_delta(const char *in_path, const char *out_path, int device);
int pm_maf_analyzer(const char *maf_path, const char *out_path, int device);

/* `mugsy_profiles make -in_maf <maf> -out_dir <dir> -basename <b>` (lib/profiles/m_make.ml:90-93): writes <dir>/profiles
 * (records of lib/profiles/m_profile.ml:122-135, the translate path's input) and <dir>/sequences.fasta (one consensus
 * per block, m_make.ml:15-45).  <dir> must exist.  Restated from the OCaml source, which cannot be run in this build
 * image: see the header of csrc/profiles_make.hip. */
int pm_profiles_make(const char *in_maf, const char *out_dir, const char *basename, int device);
/* make(left) + make(right) + translate of one Mugsy_profile node in ONE process and one HIP context: what
 * lib/base/mugsy_profiles_task.ml:40-58 runs as three processes.  Writes the same files with the same bytes
 * (<dir>/profiles, <dir>/sequences.fasta for both sides, out_delta); the rows go from the make stage to the translate stage in
 * memory.  pm_profiles_make (and this) also write <dir>/profiles.soa, the rows of <dir>/profiles as flat binary arrays: the
 * translate stage reads it instead of parsing the text when it matches the text file (PM_NO_SOA=1 disables). */
int pm_stage_files(const char *left_maf, const char *left_dir, const char *left_basename, const char *right_maf, const char *right_dir,
                   const char *right_basename, const char *const *delta_paths, int n_paths, const char *out_delta, int device);
/* `mugsy_profiles untranslate -profile_paths_list <file of dirs> -in_maf <maf> -out_maf <maf>`
 * (lib/profiles/m_untranslate.ml:206-221): rewrites a MAF whose `s` lines name profile blocks into one over the real
 * genomes.  profile_dirs: the directories the list file names, in order.  Restated from the OCaml source (see
 * csrc/untranslate.hip). */
int pm_untranslate(const char *const *profile_dirs, int n_dirs, const char *in_maf, const char *out_maf, int device);

/* ------------------------------------------------------------------------------------------------------
 * Profile x profile DP (BASELINE.json's GCUPS metric).  NO REFERENCE COUNTERPART: the reference has no DP, no
 * scores, no traceback; this interface and the computation behind it are specified by this
 * repo (oracle/dp_oracle.h) and checked against its own scalar oracle only.
 *
 * A profile is a run of 8-byte columns {nA, nC, nG, nT, nGap, 0, 0, 0} (how many rows hold each symbol).
 * Pair k aligns columns [off_a[k], off_a[k+1]) of cols_a with columns [off_b[k], off_b[k+1]) of cols_b,
 * globally, with affine gaps, int32 scores; recurrence and tie-breaking: oracle/dp_oracle.h.
 * Bytes 5-7 of a column are not scored (byte 5 is where pm_dp_pack_maf counts symbols that are neither ACGT nor a gap).
 * Limits, all enforced by pm_dp_batch_create (PM_E_INVALID): |sub| <= 127, gap penalties >= 0 with gap_open + gap_extend <= 32767,
 * profile length <= 2^24 columns, (largest row total of a column of B) x max|sub| <= 32767, and scores within +-2^28:
 * rows(A) x rows(B) x max|sub| x min(La, Lb) + (La + Lb) x gap_extend + 2 x gap_open < 2^28 for every pair. */
typedef struct pm_dp_params {
  int32_t sub[25]; /* sub[a*5+b], symbols A, C, G, T, gap */
  int32_t gap_open;
  int32_t gap_extend;
} pm_dp_params_t;

typedef struct pm_dp_batch pm_dp_batch_t; /* opaque; owns device memory */

/* How a batch is run -- every field 0: chosen per batch by the library (what production callers pass, or NULL).  Tests, the fuzzer and
 * the timing tools set fields to force one variant or another; the library reads no environment variable for any of this.  A batch
 * takes its options when it is created (pm_dp_batch_create_opt, pm_dp_stream_create_opt); the entry points that make their batches
 * themselves (pm_dp_align_*, pm_dp_stream_create, pm_dp_batch_create) take the process's defaults, pm_dp_set_default_options. */
typedef struct pm_dp_options {
  int32_t path_mode;       /* 1: paths from stored decision bits; 2: from checkpoints and recomputed blocks */
  int32_t cols_per_lane;   /* 8 or 16 columns of B per lane and stripe */
  int32_t waves_per_pair;  /* 1, 2, 4, 8, 16 wavefronts per pair */
  int32_t groups_per_pair; /* 1: one workgroup per pair, always; 2, 4, ...: that many (with waves_per_pair >= 2) */
  int32_t band;            /* 1: never; 2: whenever it fits (the walk's precomputed blocks around the diagonal) */
  int32_t walk_lanes;      /* lanes per pair of the checkpoint walk (a power of two) */
  int32_t int16_weights;   /* 1: the int16 column weights also where int8 would do */
  int32_t no_uniform_depth;/* 1: the general column score also where every column of A holds the same number of symbols */
  int32_t keep_order;      /* 1: pairs are processed in input order (default: longest first) */
  int32_t no_tiers;        /* 1: a chunk's longest pairs are not put into launches of their own */
  int32_t tier_min_pairs;  /* chunks of at least this many pairs may get tiers (default 4 096) */
  int32_t full_stripes;    /* 1: the last stripe of a pair is as wide as the others (default: narrower last stripes, 16 columns per lane) */
  int32_t slots;           /* 2..8 parts of the path workspace that the chunks of a batch that does not fit use in turn (default 3) */
  int32_t split;           /* N >= 1: a batch that fits is cut into N chunks all the same (default: 1 or 2, by the size of its walk) */
  int32_t no_gate;         /* 1: a chunk's fill kernel is not held back until the chunk before has nothing left to dispatch */
  int32_t tile_steps;      /* 1: no launch takes its work from a queue of tiles; a multiple of 64: every checkpoint / score launch does, in
                            * tiles of that many steps (default: launches of few long pairs, 1 024 steps) */
  int32_t early_walk;      /* the walk BESIDE the fill kernel of its own launch (finished pairs taken from per-XCD lists while the fill goes on):
                            * 1: never; 2: wherever a launch allows it (one wavefront per pair); default: launches of long pairs whose walk
                            * nothing else hides */
  int32_t reserved;
  int64_t segment_cells;   /* host-fed engine: no more upload segments than leave each this many cells (default 5e9) */
} pm_dp_options_t;
/* The defaults batches are created with when no options are given (NULL: all zero again).  Copied under a lock. */
int pm_dp_set_default_options(const pm_dp_options_t *options);

/* Upload a batch (host pointers).  tb_budget_bytes bounds the path workspace (<= 0: 60 % of the device's memory; only what the
 * batch needs is allocated); a batch that needs more is processed in chunks of a third of it, the fill kernels of consecutive
 * chunks on streams of their own (one takes the SIMDs the other leaves as it drains), the path kernel of a chunk beside the
 * fill kernels of the next. */
int pm_dp_batch_create(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                       const pm_dp_params_t *params, int64_t tb_budget_bytes, int device, pm_dp_batch_t **out);
int pm_dp_batch_create_opt(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                           const pm_dp_params_t *params, const pm_dp_options_t *options, int64_t tb_budget_bytes, int device,
                           pm_dp_batch_t **out);
/* One pass over every pair: fill (scores, and what the path walk needs) and, when traceback != 0, the path walk.
 * Asynchronous on hip_stream. */
int pm_dp_batch_run(pm_dp_batch_t *batch, int traceback, void *hip_stream);
/* Same, timed with HIP events on the stream (waits): device milliseconds of the fill and traceback kernels. */
int pm_dp_batch_run_profiled(pm_dp_batch_t *batch, int traceback, void *hip_stream, float *ms_fill, float *ms_traceback);
/* After pm_dp_batch_run_profiled: the time during which SOME fill kernel was running.  ms_fill above is the sum over the step's
 * launches (what a kernel trace adds up); the fill launches of a batch of several chunks overlap, so the two differ there. */
int pm_dp_batch_fill_busy_ms(pm_dp_batch_t *batch, float *ms);
/* scores[n_pairs]; n_ops[n_pairs]; ops: pair k owns bytes [off_a[k]+off_b[k], off_a[k+1]+off_b[k+1]) and its path is
 * the LAST n_ops[k] bytes of that slot, first op first (0 = M, 1 = I: column of B against a gap, 2 = D). */
int pm_dp_batch_fetch(pm_dp_batch_t *batch, int32_t *scores, uint8_t *ops, int32_t *n_ops);
int pm_dp_batch_info(pm_dp_batch_t *batch, int64_t *cells, int64_t *traceback_bytes_per_run, int64_t *input_bytes, int32_t *n_chunks);
/* The batch's processing order and its chunks of the path workspace.  The pairs are processed longest first (a launch lasts at
 * least as long as its longest pair; results are returned in the caller's order all the same): order[q] = the pair at position q
 * (n_pairs values, may be NULL); chunk c covers positions [first_position[c], first_position[c+1]), first_position[n_chunks] =
 * n_pairs (at most `capacity` values are written). */
int pm_dp_batch_chunks(pm_dp_batch_t *batch, int64_t *first_position, int32_t capacity, int32_t *order);
/* Which kernel variant the batch runs: columns of B per lane (8/16); *dot4 bit 0 = the int8 dot4 path applies, bit 1 = uniform
 * depth (every column of A holds the same number of symbols, so the gap row of the score is folded into the base weights and a
 * per-column constant); and the VALU instructions per DP cell of that variant (for roofline accounting). */
/* What a pass computes and launches: the cells the fill kernel's stripes cover (>= the pairs' La x Lb: a stripe is 64 lanes x 16, 8
 * or 4 columns wide whatever is left of B), whether the last stripes are the narrow ones, and the fill launches of a pass (one per
 * workspace chunk and one per tier of long pairs).  Any pointer may be NULL. */
int pm_dp_batch_geometry(pm_dp_batch_t *batch, int64_t *padded_cells, int32_t *narrow_last_stripes, int64_t *fill_launches);
int pm_dp_batch_variant(pm_dp_batch_t *batch, int32_t *cols_per_lane, int32_t *dot4, int32_t *valu_ops_per_cell);
/* How the batch gets its paths: checkpoints != 0 -> the fill kernel computes scores only and leaves row/column checkpoints,
 * and the walk re-runs the recurrence inside the block_rows x block_columns blocks the path crosses (the default);
 * checkpoints == 0 -> the fill kernel stores 4 decision bits per cell.  Same results; chosen per batch (a batch of a few short
 * pairs is better off storing the bits: the walk's chain of blocks and its extra launch have a latency that does not shrink
 * with the batch), or fixed by the environment PM_DP_MODE=bits|ckpt. */
int pm_dp_batch_path_mode(pm_dp_batch_t *batch, int32_t *checkpoints, int32_t *block_rows, int32_t *block_columns);
void pm_dp_batch_destroy(pm_dp_batch_t *batch);

/* The DP fed from host memory (csrc/dp_stream.hip): pm_dp_stream_align loads a batch in `segments` pieces of consecutive pairs
 * on an upload stream, runs the fill kernels behind the pieces as they arrive (a workspace chunk per piece, its pairs ordered longest
 * first; a small batch: one chunk with a launch per piece), one path kernel per workspace chunk, and brings the results back on a
 * download stream, chunk by chunk.  `segments` is an upper limit: the engine takes fewer when a piece would be too small a launch.
 * Same inputs, same outputs and same output layout as
 * pm_dp_batch_create + run + fetch (ops == n_ops == NULL: scores only); the device buffers are kept from call to call.
 * Copies are asynchronous only from / to pinned host memory: pm_dp_host_alloc / pm_dp_host_free hand it out (pageable buffers
 * work, their copies just serialise on the host).  workspace_bytes: path workspace (<= 0: as pm_dp_batch_create).  Blocking: returns when
 * every result is in the caller's arrays. */
typedef struct pm_dp_stream pm_dp_stream_t;
int pm_dp_host_alloc(void **ptr, int64_t bytes);
void pm_dp_host_free(void *ptr);
int pm_dp_stream_create(const pm_dp_params_t *params, int32_t segments, int64_t workspace_bytes, int device, pm_dp_stream_t **out);
int pm_dp_stream_create_opt(const pm_dp_params_t *params, const pm_dp_options_t *options, int32_t segments, int64_t workspace_bytes, int device,
                            pm_dp_stream_t **out);
int pm_dp_stream_align(pm_dp_stream_t *stream, const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b,
                       int64_t n_pairs, int32_t *scores, uint8_t *ops, int32_t *n_ops);
/* The same fed with ROW TEXTS instead of packed columns: pair k = block k of each side in the flat description of pm_dp_pack_maf
 * below (text, row_off, block_row).  The texts go up in the same segments and every segment is packed on the device as soon as it
 * has arrived: the packed columns (8 bytes per column whatever the number of rows) never cross the link, so a 2-row profile moves
 * a quarter of the bytes.  Results as pm_dp_stream_align's. */
int pm_dp_stream_align_text(pm_dp_stream_t *stream, const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a,
                            const int64_t *block_row_a, const uint8_t *text_b, const int64_t *row_off_b, int64_t n_rows_b,
                            const int64_t *block_row_b, int64_t n_pairs, int32_t *scores, uint8_t *ops, int32_t *n_ops);
void pm_dp_stream_destroy(pm_dp_stream_t *stream);

/* MAF blocks into the DP and out of it (csrc/dp_maf.hip).  A list of blocks is described flat: `text` holds the gapped texts of
 * every row of every block back to back (no separators), row r is bytes [row_off[r], row_off[r+1]), block b is rows
 * [block_row[b], block_row[b+1]) (block_row[0] = 0, block_row[n_blocks] = n_rows); the rows of one block have the same length.
 * Symbols: case-insensitive A, C, G, T -> bytes 0-3 of the packed column, '-' -> byte 4, anything else (N, IUPAC) -> byte 5
 * (carried, not scored).  At most 255 rows per block.
 *
 * pm_dp_pack_maf: rows of every block -> one packed 8-byte column per block column (the per-column fold over a block's rows of
 *   lib/profiles/m_make.ml:15-45, counting instead of voting).  col_off_out[n_blocks + 1] is always written; with cols_out == NULL
 *   only the sizes are computed (cols_out holds col_off_out[n_blocks] * 8 bytes).
 * pm_dp_emit_maf: pair k = block k of A and block k of B plus its path ops[ops_off[k] .. ops_off[k] + n_ops[k]) (first op first,
 *   0 = M, 1 = I, 2 = D) -> the merged block: rows(A) + rows(B) rows of n_ops[k] bytes, A's rows first, '-' where the path skips
 *   a row's side (the expansion of lib/profiles/m_untranslate.ml:38-52 along the DP's path).  out_off[n_pairs + 1] (byte offsets of
 *   the pairs' texts) is always written; with out_text == NULL only the sizes are computed.  A path that does not span its pair
 *   of blocks is refused (PM_E_INVALID).
 * pm_dp_align_maf: two MAF files with the same number of blocks -> a MAF file of the merged blocks, `a score=<DP score>`, every
 *   `s` line keeping its name / start / size / strand / srcSize fields.  New surface (the reference has no such stage). */
int pm_dp_pack_maf(const uint8_t *text, const int64_t *row_off, int64_t n_rows, const int64_t *block_row, int64_t n_blocks,
                   uint8_t *cols_out, int64_t *col_off_out, int device);
int pm_dp_emit_maf(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a, const uint8_t *text_b,
                   const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs, const uint8_t *ops,
                   const int64_t *ops_off, const int32_t *n_ops, uint8_t *out_text, int64_t *out_off, int device);
int pm_dp_align_maf(const char *maf_a, const char *maf_b, const pm_dp_params_t *params, const char *out_maf, int device);
/* The same for blocks that are already in memory (the flat description above): pack -> DP -> expansion in one call, the texts going
 * to the device once and the packed columns and paths never leaving it.  scores[n_pairs]; merged_columns[n_pairs] = columns of
 * merged block k (= length of its path); merged block k = rows(A_k) + rows(B_k) rows of merged_columns[k] bytes at
 * out_text + out_off[k], A's rows first; out_off[n_pairs + 1] is always written.  out_capacity: bytes out_text holds -- a merged
 * block has at most columns(A_k) + columns(B_k) columns, so (rows of A + rows of B) x (columns of A + columns of B) summed over the
 * pairs always suffices; too small is PM_E_INVALID with out_off filled in (call again with out_off[n_pairs] bytes).  New surface. */
int pm_dp_align_blocks(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a, const uint8_t *text_b,
                       const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs, const pm_dp_params_t *params,
                       int32_t *scores, int32_t *merged_columns, uint8_t *out_text, int64_t out_capacity, int64_t *out_off, int device);

/* ------------------------------------------------------------------------------------------------------
 * Several devices of one node behind one call (csrc/multi.hpp).  The host north_star names (OCaml behind a C ABI, or the CLI)
 * cannot run one torch.distributed process per GPU; these entries take a device list instead: the job's independent items --
 * pairs, blocks, delta files -- are cut into n_devices contiguous slices (delta files by count, pm_partition: the first n % parts
 * slices hold one item more; pairs and blocks by cells, pm_partition_weighted), one host thread and one HIP context per device run the single-device path on their slice, there is NO collective, and
 * the outputs are gathered on the host in input order.  What this replaces in the reference: the chunked pair lists of
 * lib/base/pm_job.ml:43-57,83-91 run as `run_size` concurrent OS processes (lib/base/queued_task_server.ml:57-66), whose outputs
 * the order-sensitive writer concatenates (lib/profiles_lib/m_delta_stream_writer.hh:62-67).  The same device may be named more
 * than once (its workers share it).  Results are byte for byte those of the single-device call.  If a worker fails the call
 * fails with the error of the first failing slice (pm_last_error names the device).  New surface. */
int pm_partition(int64_t n_items, int n_parts, int part, int64_t *lo, int64_t *hi);
/* The same for items of unequal cost: contiguous slices cut where the running sum of `weights` (>= 0) comes closest to k / n_parts
 * of the total; cuts[0 .. n_parts] = the slices' bounds (cuts[0] = 0, cuts[n_parts] = n_items).  Equal weights: pm_partition's
 * slices.  The pm_dp_align_*_multi entries cut their pairs so, by cells: weight = La x Lb + La + Lb + 1 (a pair of MAF blocks: the
 * columns of its blocks' first rows), so that the slices of a ragged batch hold equal work, not equal counts.  Host only. */
int pm_partition_weighted(const int64_t *weights, int64_t n_items, int n_parts, int64_t *cuts);
/* pm_translate_files over a device list: every worker parses and translates its slice of the delta-file list against the two
 * sides (loaded once); the texts are joined in list order with the writer's header rule re-applied at the seams (a `>` line is
 * printed only when the name pair changes).  On a failure the output holds what the reference had written when it died: the
 * slices before the failing one and the failing slice's partial output. */
int pm_translate_files_multi(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths,
                             const char *out_path, const int *devices, int n_devices);
/* Host only (no device): complete m_translate outputs over consecutive slices of one delta-file list (two header lines each) ->
 * the file one run over the whole list prints; the join pm_translate_files_multi does in memory, for callers that shard across
 * processes or nodes themselves (the reference's SGE driver, lib/base/sge_interface.ml:55-74). */
int pm_delta_join_files(const char *const *part_paths, int n_parts, const char *out_path);
/* pm_dp_stream_align over a device list: host columns in, scores / ops / n_ops out in pm_dp_batch_fetch's layout (every slice
 * writes its results at its own place of the caller's arrays; ops == n_ops == NULL: scores only). */
int pm_dp_align_multi(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                      const pm_dp_params_t *params, const int *devices, int n_devices, int32_t *scores, uint8_t *ops, int32_t *n_ops);
/* pm_dp_align_blocks / pm_dp_align_maf over a device list: MAF blocks in, merged MAF blocks out in pair order -- north_star's
 * "host-side gather of MAF blocks". */
int pm_dp_align_blocks_multi(const uint8_t *text_a, const int64_t *row_off_a, int64_t n_rows_a, const int64_t *block_row_a,
                             const uint8_t *text_b, const int64_t *row_off_b, int64_t n_rows_b, const int64_t *block_row_b, int64_t n_pairs,
                             const pm_dp_params_t *params, const int *devices, int n_devices, int32_t *scores, int32_t *merged_columns,
                             uint8_t *out_text, int64_t out_capacity, int64_t *out_off);
int pm_dp_align_maf_multi(const char *maf_a, const char *maf_b, const pm_dp_params_t *params, const char *out_maf, const int *devices,
                          int n_devices);

#ifdef __cplusplus
}
#endif

#endif
