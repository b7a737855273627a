// translate_host.hpp -- host-side tables of the translate path (see translate_host.cc).
#pragma once

#include <cstdint>
#include <cstdio>
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
  int parse_rc = 0;
  std::string parse_msg;
};

int parse_profiles(const std::string &path, Side &side);
void build_side_index(Side &side);
int parse_delta_file(const std::string &path, DeltaTable &table);
int parse_delta_text(const std::string &text, const std::string &label, DeltaTable &table);
bool read_stream(FILE *f, std::string &out);
void enumerate_units(const Side &left, const Side &right, const DeltaTable &table, size_t first_entry, UnitList &units);
int write_results(FILE *f, const Side &left, const Side &right, const UnitList &units, const int32_t *status,
                  const int64_t *unit_entry_off, const pm_entry_t *entries, const int64_t *offsets, std::string &last_left,
                  std::string &last_right);
// <dir>/profiles.soa when it matches <dir>/profiles, else the text file (parse_profiles)
int load_side(const std::string &dir, Side &side);
int write_side_soa(const std::string &dir, const Side &side, long long profiles_text_bytes);
int load_deltas(const std::vector<std::string> &delta_paths, Workload &w);
void parse_deltas(const std::vector<std::string> &delta_paths, Workload &w);
void index_and_enumerate(Workload &w);
int run_workload(Workload &w, FILE *out, int device);
int load_workload(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, Workload &w);
void workload_views(const Workload &w, pm_rows_t *left, pm_rows_t *right, pm_deltas_t *deltas, pm_units_t *units);
int translate_to_file(const std::string &left_dir, const std::string &right_dir, const std::vector<std::string> &delta_paths, FILE *out,
                      int device);

} // namespace pm
