// translate_host.hpp -- host-side tables of the translate path (see translate_host.cc).
#pragma once

#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/paramugsy_amd.h"

namespace pm {

// One side's rows in file order, as flat arrays (the layout pm_rows_t points into).
struct Side {
  std::vector<std::string> major;    // p_major_name
  std::vector<std::string> seq_name; // p_seq_name
  std::vector<long long> start, end, length;
  std::vector<long long> gap_off, gap_start, gap_end;
  std::map<std::string, std::vector<int> > by_seq; // rows of a sequence, sorted by forward start
};

// Parsed delta entries of all files of a job, flat (the layout pm_deltas_t points into).
struct DeltaTable {
  std::vector<std::string> ref_name, qry_name; // header names in force for each entry
  std::vector<long long> ref_len, qry_len;     // header lengths in force for each entry
  std::vector<long long> ref_start, ref_end, qry_start, qry_end;
  std::vector<long long> ref_gap_off, ref_gap_start, ref_gap_end;
  std::vector<long long> qry_gap_off, qry_gap_start, qry_gap_end;
};

struct UnitList {
  std::vector<int32_t> delta, left, right;
};

// Everything a translate job needs on the host: both sides, all delta entries, the unit list.
struct Workload {
  Side left, right;
  DeltaTable table;
  UnitList units;
  bool units_listed = false; // false: the job lists them on the device (index_sides), units stays empty
  int parse_rc = 0;
  std::string parse_msg;
};

int parse_profiles(const std::string &path, Side &side);
void build_side_index(Side &side);
int parse_delta_file(const std::string &path, DeltaTable &table);
int parse_delta_text(const std::string &text, const std::string &label, DeltaTable &table);
bool read_stream(FILE *f, std::string &out);
void enumerate_units(const Side &left, const Side &right, const DeltaTable &table, size_t first_entry, UnitList &units);
// Where a job's delta text goes: a stream, or a string in memory (a shard of a multi-device job, gathered by the caller).
struct OutSink {
  FILE *f = nullptr;
  std::string *mem = nullptr;
  OutSink(FILE *f_) : f(f_) {}
  OutSink(std::string *m) : mem(m) {}
  bool write(const char *p, size_t n) {
    if(mem) {
      mem->append(p, n);
      return true;
    }
    return fwrite(p, 1, n, f) == n;
  }
};
// <dir>/profiles.soa when it matches <dir>/profiles, else the text file (parse_profiles)
int load_side(const std::string &dir, Side &side, const pm_translate_options_t &opt);
int write_side_soa(const std::string &dir, const Side &side, long long profiles_text_bytes);
int load_deltas(const std::vector<std::string> &delta_paths, Workload &w);
void parse_deltas(const std::vector<std::string> &delta_paths, Workload &w);
void index_and_enumerate(Workload &w);
void index_sides(Workload &w);
int run_workload(Workload &w, FILE *out, const pm_translate_options_t &opt, int device);
// The device part of a translate job over tables that are in place (the two sides may be shared, read-only, by several callers):
// upload + prepare + sizing, one pass, fetch, format.  units: the unit list, or null for the job to list the units on the device from
// the sides' row index (build_side_index must have run on both).  parse_rc / parse_msg: a delta-file parse failure to report after the
// output of the entries read before it.
int run_tables(const Side &left, const Side &right, const DeltaTable &table, const UnitList *units, int parse_rc, const std::string &parse_msg,
               OutSink out, const pm_translate_options_t &opt, int device);
int load_workload(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, Workload &w,
                  const pm_translate_options_t &opt, bool list_units = true);
void workload_views(const Workload &w, pm_rows_t *left, pm_rows_t *right, pm_deltas_t *deltas, pm_units_t *units);
// Bytes in device memory to a sink through pinned staging pieces, the writing beside the copying; `copied` runs once the last byte
// has left the device.  The calling thread's current device must be the buffer's.
int device_bytes_to_sink(const char *dev, int64_t n_bytes, OutSink out, bool timing, const std::function<void()> &copied);
// multi.hpp: the job's delta-file list over several devices, texts joined in list order (header rule re-applied at the seams)
int merge_shard_texts(const std::vector<std::string> &parts, size_t n_parts, OutSink out);
int translate_to_file_multi(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, FILE *out,
                            const pm_translate_options_t &opt, const int *devices, int n_devices);
int translate_to_file(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, FILE *out,
                      const pm_translate_options_t &opt, int device);

} // namespace pm
