/*
 * oracle/pm_oracle_capi.cc -- TEST INFRASTRUCTURE ONLY.
 * C entry points of the CPU oracle, taking the same flat tables as the product's C ABI
 * (struct layouts from include/paramugsy_amd.h), so a test can hand identical inputs to both and compare the
 * outputs element by element.  Loaded with ctypes by tests/ and by bench.py's cpu_baseline leg only.
 */
#include "pm_oracle.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "../include/paramugsy_amd.h"

namespace {

pmo::Profile row_profile(const pm_rows_t *rows, int64_t r) {
  pmo::Profile p;
  p.range = pmo::Range{rows->start[r], rows->end[r]};
  p.length = rows->length[r];
  p.src_size = 0;
  for(int64_t k = rows->gap_off[r]; k < rows->gap_off[r + 1]; ++k) {
    p.gaps.push_back(pmo::Range{rows->gap_start[k], rows->gap_end[k]});
  }
  return p;
}

pmo::DeltaEntry delta_entry(const pm_deltas_t *d, int64_t k) {
  pmo::DeltaEntry e;
  e.ref = pmo::Range{d->ref_start[k], d->ref_end[k]};
  e.query = pmo::Range{d->qry_start[k], d->qry_end[k]};
  for(int64_t g = d->ref_gap_off[k]; g < d->ref_gap_off[k + 1]; ++g) {
    e.ref_gaps.push_back(pmo::Range{d->ref_gap_start[g], d->ref_gap_end[g]});
  }
  for(int64_t g = d->qry_gap_off[k]; g < d->qry_gap_off[k + 1]; ++g) {
    e.query_gaps.push_back(pmo::Range{d->qry_gap_start[g], d->qry_gap_end[g]});
  }
  return e;
}

struct Result {
  std::vector<int32_t> status;
  std::vector<int64_t> unit_entry_off;
  std::vector<pm_entry_t> entries;
  std::vector<int64_t> offsets;
};

}  // namespace

extern "C" {

/* Runs every unit on the oracle.  Returns an opaque result (free with pmo_result_free). */
void *pmo_translate_units(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const pm_units_t *units) {
  Result *res = new Result();
  res->unit_entry_off.push_back(0);
  for(int64_t u = 0; u < units->n; ++u) {
    pmo::Profile lp = row_profile(left, units->left[u]);
    pmo::Profile rp = row_profile(right, units->right[u]);
    pmo::DeltaEntry de = delta_entry(deltas, units->delta[u]);
    std::vector<pmo::DeltaEntry> emitted;
    int32_t st = 0;
    try {
      pmo::translate_unit(de, lp, rp, &emitted);
    }
    catch(pmo::Failure const &f) {
      st = (int32_t)f.code;
    }
    for(size_t k = 0; k < emitted.size(); ++k) {
      std::vector<long> offs = pmo::offsets_of_gaps(emitted[k]);
      pm_entry_t e;
      e.ref_start = emitted[k].ref.s;
      e.ref_end = emitted[k].ref.e;
      e.qry_start = emitted[k].query.s;
      e.qry_end = emitted[k].query.e;
      e.offset_begin = (int64_t)res->offsets.size();
      e.n_offsets = (int64_t)offs.size();
      res->entries.push_back(e);
      res->offsets.insert(res->offsets.end(), offs.begin(), offs.end());
    }
    res->status.push_back(st);
    res->unit_entry_off.push_back((int64_t)res->entries.size());
  }
  return res;
}

void pmo_result_sizes(void *h, int64_t *n_entries, int64_t *n_offsets) {
  Result *res = (Result *)h;
  *n_entries = (int64_t)res->entries.size();
  *n_offsets = (int64_t)res->offsets.size();
}

void pmo_result_fetch(void *h, int32_t *status, int64_t *unit_entry_off, pm_entry_t *entries, int64_t *offsets) {
  Result *res = (Result *)h;
  if(status && !res->status.empty()) {
    memcpy(status, res->status.data(), res->status.size() * sizeof(int32_t));
  }
  if(unit_entry_off) {
    memcpy(unit_entry_off, res->unit_entry_off.data(), res->unit_entry_off.size() * sizeof(int64_t));
  }
  if(entries && !res->entries.empty()) {
    memcpy(entries, res->entries.data(), res->entries.size() * sizeof(pm_entry_t));
  }
  if(offsets && !res->offsets.empty()) {
    memcpy(offsets, res->offsets.data(), res->offsets.size() * sizeof(int64_t));
  }
}

void pmo_result_free(void *h) { delete(Result *)h; }

/* a3 / a4 over a batch of (row, index) queries; status uses the PM_ST_* numbering */
void pmo_profile_idx_of_seq_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *si, int64_t *out,
                                      int32_t *status) {
  for(int64_t q = 0; q < n; ++q) {
    pmo::Profile p = row_profile(rows, row[q]);
    out[q] = 0;
    status[q] = 0;
    try {
      out[q] = pmo::profile_idx_of_seq_idx(p, si[q]);
    }
    catch(pmo::Failure const &f) {
      status[q] = (int32_t)f.code;
    }
  }
}

void pmo_seq_idx_of_profile_idx_batch(const pm_rows_t *rows, int64_t n, const int32_t *row, const int64_t *pi, int64_t *out,
                                      int32_t *status) {
  for(int64_t q = 0; q < n; ++q) {
    pmo::Profile p = row_profile(rows, row[q]);
    out[q] = 0;
    status[q] = 0;
    try {
      long v = 0;
      if(pmo::seq_idx_of_profile_idx(p, pi[q], &v)) {
        out[q] = v;
      }
      else {
        status[q] = (int32_t)pmo::IS_NONE;
      }
    }
    catch(pmo::Failure const &f) {
      status[q] = (int32_t)f.code;
    }
  }
}

/* The unit list the oracle's restatement of m_translate.cc:666-707 visits, for a job given as files.
 * Rows are numbered in file order per side, entries in stream order over all files; returns the number of
 * units and writes up to `cap` triples. */
int64_t pmo_enumerate_units(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths, int64_t cap,
                            int32_t *u_delta, int32_t *u_left, int32_t *u_right) {
  /* file-order row ids: re-read each side and tag rows through the text field, which load_profile_map ignores (lite) */
  pmo::ProfileMap sides[2];
  const char *dirs[2] = {left_dir, right_dir};
  for(int s = 0; s < 2; ++s) {
    std::ifstream in((std::string(dirs[s]) + "/profiles").c_str());
    pmo::Profile p;
    long id = 0;
    pmo::ProfileMap map;
    while(pmo::read_profile(in, true, &p)) {
      std::ostringstream tag;
      tag << id++;
      p.text = tag.str();
      map[p.seq_name].push_back(p);
    }
    for(pmo::ProfileMap::iterator it = map.begin(); it != map.end(); ++it) {
      std::sort(it->second.begin(), it->second.end(), [](pmo::Profile const &a, pmo::Profile const &b) {
        return pmo::forward_of(a.range).s < pmo::forward_of(b.range).s;
      });
    }
    sides[s] = map;
  }
  int64_t n = 0;
  int32_t d = 0;
  for(int k = 0; k < n_paths; ++k) {
    std::ifstream in(delta_paths[k]);
    pmo::DeltaReader reader(in);
    pmo::DeltaEntry de;
    while(reader.next(&de)) {
      pmo::ProfileMap::const_iterator l = sides[0].find(de.names.first);
      pmo::ProfileMap::const_iterator r = sides[1].find(de.names.second);
      if(l != sides[0].end() && r != sides[1].end()) {
        std::vector<std::pair<size_t, size_t> > pairs;
        pmo::units_for_entry(de, l->second, r->second, &pairs);
        for(size_t j = 0; j < pairs.size(); ++j) {
          if(n < cap) {
            u_delta[n] = d;
            u_left[n] = (int32_t)atol(l->second[pairs[j].first].text.c_str());
            u_right[n] = (int32_t)atol(r->second[pairs[j].second].text.c_str());
          }
          ++n;
        }
      }
      ++d;
    }
  }
  return n;
}

/* whole-job oracle run, for timing as cpu_baseline kind "port" */
int pmo_translate_files(const char *left_dir, const char *right_dir, const char *const *delta_paths, int n_paths, const char *out_path) {
  std::vector<std::string> paths;
  for(int k = 0; k < n_paths; ++k) {
    paths.push_back(delta_paths[k]);
  }
  std::ofstream out(out_path);
  out << (std::string(left_dir) + "/sequences.fasta") << " " << (std::string(right_dir) + "/sequences.fasta") << std::endl;
  out << "NUCMER\n";
  try {
    pmo::translate(left_dir, right_dir, paths, out);
  }
  catch(pmo::Failure const &f) {
    return (int)f.code;
  }
  return 0;
}

}  // extern "C"
