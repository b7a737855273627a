/*
 * oracle/pm_oracle.cc -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See pm_oracle.hh for scope, parity status and who may use this file.
 *
 * Written from the behaviour of the reference, not from its text: flat structs,
 * bool+out-parameter instead of a heap-allocating option type, one cursor
 * struct for the paired gap lists, one struct for the merge state.
 */
#include "pm_oracle.hh"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace pmo {

/* ---------------------------------------------------------------- a1 ranges */

/* m_range.hh:80-94: both forced forward, (max starts, min ends), kept if end-start >= 0 */
bool overlap(Range a, Range b, Range *out) {
  Range fa = forward_of(a);
  Range fb = forward_of(b);
  long s = std::max(fa.s, fb.s);
  long e = std::min(fa.e, fb.e);
  if(e - s >= 0) {
    if(out) {
      out->s = s;
      out->e = e;
    }
    return true;
  }
  return false;
}

/* m_range.hh:49-52 */
bool contains(Range r, long v) {
  Range f = forward_of(r);
  return f.s <= v && v <= f.e;
}

/* m_range.hh:106-115: 0-based MAF (start,size,strand) -> 1-based inclusive, reverse counted from src end */
Range range_of_maf(long start, long size, long src_size, bool forward) {
  if(forward) {
    return Range{start + 1, start + size};
  }
  return Range{src_size - start, src_size - start - (size - 1)};
}

/* -------------------------------------------------------------- a2 profiles */

static long total_gap_length(Gaps const &gaps) {
  long n = 0;
  for(size_t k = 0; k < gaps.size(); ++k) {
    n += range_length(gaps[k]);
  }
  return n;
}

/* m_profile.hh:46-63 */
Profile make_derived_profile(std::string const &major_name, std::string const &minor_name,
                             std::string const &seq_name, Range range, Gaps const &gaps) {
  Profile p;
  p.major_name = major_name;
  p.minor_name = minor_name;
  p.seq_name = seq_name;
  p.range = range;
  p.length = range_length(range) + total_gap_length(gaps);
  p.src_size = 0;
  p.gaps = gaps;
  return p;
}

/* m_profile.cc:15-85.  Header `major minor seq start end length src_size` (the last two are read as
 * unsigned int there, :27-28), gap lines until a line that is exactly "0", then one text line. */
bool read_profile(std::istream &in, bool lite, Profile *out) {
  std::string line;
  if(!std::getline(in, line)) {
    return false;
  }
  std::istringstream head(line);
  Profile p;
  long start = 0;
  long end = 0;
  unsigned int length = 0;
  unsigned int src_size = 0;
  if(!(head >> p.major_name >> p.minor_name >> p.seq_name >> start >> end >> length >> src_size)) {
    throw Failure(PARSE_ERROR);
  }
  p.range = Range{start, end};
  p.length = length;
  p.src_size = src_size;
  while(std::getline(in, line) && line != "0") {
    std::istringstream gap_line(line);
    long gs = 0;
    long ge = 0;
    if(!(gap_line >> gs >> ge)) {
      throw Failure(PARSE_ERROR);
    }
    p.gaps.push_back(Range{gs, ge});
  }
  std::string text;
  std::getline(in, text);
  if(!lite) {
    p.text = text;
  }
  *out = p;
  return true;
}

/* a3: m_profile.cc:91-112 */
long profile_idx_of_seq_idx(Profile const &p, long si) {
  long offset = std::labs(p.range.s - si) + 1;
  if(!contains(p.range, si)) {
    throw Failure(SEQ_IDX_OUT_OF_RANGE);
  }
  long skipped = 0;
  for(size_t k = 0; k < p.gaps.size(); ++k) {
    if(p.gaps[k].s <= offset + skipped) {
      skipped += range_length(p.gaps[k]);
    }
    else {
      break;
    }
  }
  return skipped + offset;
}

/* a4: m_profile.cc:114-149 */
bool seq_idx_of_profile_idx(Profile const &p, long pi, long *out) {
  if(!(pi < p.length + 1)) {
    throw Failure(PROFILE_IDX_OUT_OF_RANGE);
  }
  long skipped = 0;
  for(size_t k = 0; k < p.gaps.size(); ++k) {
    if(p.gaps[k].e < pi) {
      skipped += range_length(p.gaps[k]);
    }
    else if(p.gaps[k].s <= pi) {
      return false; /* the column is a gap in this row */
    }
    else {
      break;
    }
  }
  long offset = pi - skipped - 1;
  *out = is_forward(p.range) ? p.range.s + offset : p.range.s - offset;
  return true;
}

/* a5: m_profile.cc:160-206.  Gaps stay in the parent's column coordinates (:152-159). */
bool subset_profile(Profile const &p, long s, long e, Profile *out) {
  if(s <= 0 || p.length < s || e <= 0 || p.length < e) {
    throw Failure(PROFILE_IDX_OUT_OF_RANGE);
  }
  if(s > e) {
    std::swap(s, e);
  }
  Range window{s, e};
  Gaps kept;
  for(size_t k = 0; k < p.gaps.size(); ++k) {
    Range clipped;
    if(overlap(p.gaps[k], window, &clipped)) {
      kept.push_back(clipped);
    }
  }
  if(!kept.empty()) {
    if(kept.size() == 1 && kept[0].s == s && kept[0].e == e) {
      return false;
    }
    if(kept.front().s == s) {
      s = kept.front().e + 1;
    }
    if(kept.back().e == e) {
      e = kept.back().s - 1;
    }
  }
  long seq_s = 0;
  long seq_e = 0;
  if(!seq_idx_of_profile_idx(p, s, &seq_s)) {
    throw Failure(IS_NONE);
  }
  if(!seq_idx_of_profile_idx(p, e, &seq_e)) {
    throw Failure(IS_NONE);
  }
  *out = make_derived_profile(p.major_name, p.minor_name, p.seq_name, Range{seq_s, seq_e}, kept);
  return true;
}

/* a5: m_profile.cc:208-212 */
Profile subset_seq(Profile const &p, long s, long e) {
  /* argument evaluation order is unspecified in the reference; both calls can only fail with the same class */
  long ps = profile_idx_of_seq_idx(p, s);
  long pe = profile_idx_of_seq_idx(p, e);
  Profile sub;
  if(!subset_profile(p, ps, pe, &sub)) {
    throw Failure(IS_NONE);
  }
  return sub;
}

/* ---------------------------------------------------------------- a7 deltas */

/* m_delta.cc:14-68.  A signed offset opens a gap |v| columns after the previous gap's end; following +-1
 * of the same sign extend it.  Negative -> gap in the reference row, positive -> gap in the query row. */
void split_gaps(std::vector<long> const &offsets, Gaps *ref_gaps, Gaps *query_gaps) {
  size_t k = 0;
  long column = 0;
  while(k < offsets.size()) {
    long v = offsets[k];
    bool in_query = v > 0;
    long first = column + (in_query ? v : -v);
    ++k;
    long extra = 0;
    while(k < offsets.size() && (offsets[k] == 1 || offsets[k] == -1)) {
      if((offsets[k] > 0) != in_query) {
        break;
      }
      ++extra;
      ++k;
    }
    Range gap{first, first + extra};
    column = gap.e;
    (in_query ? query_gaps : ref_gaps)->push_back(gap);
  }
}

/* m_delta.cc:72-92: two tokens on line 1 (the second overwrites the first, both stored), line 2 = type */
DeltaReader::DeltaReader(std::istream &in) : in_(in) {
  std::string line;
  if(!std::getline(in_, line)) {
    throw Failure(PARSE_ERROR);
  }
  std::istringstream iss(line);
  std::string tok;
  if(!(iss >> tok >> tok)) {
    throw Failure(PARSE_ERROR);
  }
  files = std::make_pair(tok, tok);
  if(!std::getline(in_, kind)) {
    throw Failure(PARSE_ERROR);
  }
}

/* m_delta.cc:148-220 */
bool DeltaReader::next(DeltaEntry *out) {
  std::string line;
  if(!std::getline(in_, line)) {
    return false;
  }
  if(line[0] == '>') {
    std::istringstream head(line);
    char marker;
    head >> marker;
    if(!(head >> names_.first >> names_.second >> lengths_.first >> lengths_.second)) {
      throw Failure(PARSE_ERROR);
    }
    if(!std::getline(in_, line)) {
      throw Failure(PARSE_ERROR);
    }
  }
  std::istringstream coords(line);
  long rs, re, qs, qe, e1, e2, e3;
  if(!(coords >> rs >> re >> qs >> qe >> e1 >> e2 >> e3)) {
    throw Failure(PARSE_ERROR);
  }
  std::vector<long> offsets;
  while(std::getline(in_, line) && line != "0") {
    std::istringstream one(line);
    int v; /* int there (:189) */
    if(!(one >> v)) {
      throw Failure(PARSE_ERROR);
    }
    offsets.push_back(v);
  }
  DeltaEntry de;
  de.names = names_;
  de.lengths = lengths_;
  de.ref = Range{rs, re};
  de.query = Range{qs, qe};
  split_gaps(offsets, &de.ref_gaps, &de.query_gaps);
  *out = de;
  return true;
}

/* a8: m_delta.cc:94-146.  Both ranges flipped; each gap list reversed and mirrored about that row's
 * own column count (|range| + its own gaps). */
static Gaps mirror_gaps(Gaps const &gaps, long columns) {
  Gaps out;
  for(size_t k = gaps.size(); k-- > 0;) {
    out.push_back(Range{columns - gaps[k].e + 1, columns - gaps[k].s + 1});
  }
  return out;
}

DeltaEntry reverse_entry(DeltaEntry const &de) {
  DeltaEntry r;
  r.names = de.names;
  r.lengths = de.lengths;
  r.ref = Range{de.ref.e, de.ref.s};
  r.query = Range{de.query.e, de.query.s};
  r.ref_gaps = mirror_gaps(de.ref_gaps, range_length(de.ref) + total_gap_length(de.ref_gaps));
  r.query_gaps = mirror_gaps(de.query_gaps, range_length(de.query) + total_gap_length(de.query_gaps));
  return r;
}

/* a10: m_delta_stream_writer.hh:14-53.  Two-list merge by gap start, ties to the query list. */
std::vector<long> offsets_of_gaps(DeltaEntry const &de) {
  std::vector<long> out;
  size_t r = 0;
  size_t q = 0;
  long column = 0;
  for(;;) {
    bool have_r = r < de.ref_gaps.size();
    bool have_q = q < de.query_gaps.size();
    if(!have_r && !have_q) {
      out.push_back(0);
      return out;
    }
    bool take_ref = have_r && (!have_q || de.ref_gaps[r].s < de.query_gaps[q].s);
    Range g = take_ref ? de.ref_gaps[r] : de.query_gaps[q];
    long sign = take_ref ? -1 : 1;
    out.push_back(sign * (g.s - column));
    for(long n = range_length(g) - 1; n > 0; --n) {
      out.push_back(sign);
    }
    column = g.e;
    if(take_ref) {
      ++r;
    }
    else {
      ++q;
    }
  }
}

/* m_delta_stream_writer.hh:55-82: header only when the name pair changes; error fields are the literal 1 2 3 */
void DeltaWriter::write(DeltaEntry const &de) {
  if(de.names != last_names_) {
    out_ << '>' << de.names.first << ' ' << de.names.second << ' ' << de.lengths.first << ' ' << de.lengths.second << '\n';
    last_names_ = de.names;
  }
  std::vector<long> offsets = offsets_of_gaps(de);
  out_ << de.ref.s << ' ' << de.ref.e << ' ' << de.query.s << ' ' << de.query.e << " 1 2 3\n";
  for(size_t k = 0; k < offsets.size(); ++k) {
    out_ << offsets[k] << '\n';
  }
}

/* -------------------------------------------------- a11 paired gap cursor */

namespace {

enum Row { ROW_REF = 0, ROW_QUERY = 1 };

/* m_translate.cc:24-139: a cursor over two gap lists with one push-back slot per row */
struct PairCursor {
  Gaps const *list[2];
  size_t at[2];
  bool held[2];
  Range hold[2];

  PairCursor(Gaps const &ref, Gaps const &query) {
    list[ROW_REF] = &ref;
    list[ROW_QUERY] = &query;
    at[0] = at[1] = 0;
    held[0] = held[1] = false;
    hold[0] = hold[1] = Range{0, 0};
  }

  bool has(Row r) const { return held[r] || at[r] < list[r]->size(); }          /* :78-94 peek */
  Range front(Row r) const { return held[r] ? hold[r] : (*list[r])[at[r]]; }
  bool done() const { return !has(ROW_REF) && !has(ROW_QUERY); }                /* :96-98 */

  /* :34-62: the gap whose start is nearer to its row's position; ties to the reference row */
  bool pick(long ref_pos, long query_pos, Row *row, Range *gap) const {
    if(has(ROW_REF) && has(ROW_QUERY)) {
      long rd = front(ROW_REF).s - ref_pos;
      long qd = front(ROW_QUERY).s - query_pos;
      if(rd < 0 || qd < 0) {
        throw Failure(ASSERT_GAP_BEHIND); /* :42-43 */
      }
      *row = rd <= qd ? ROW_REF : ROW_QUERY;
    }
    else if(has(ROW_REF)) {
      *row = ROW_REF;
    }
    else if(has(ROW_QUERY)) {
      *row = ROW_QUERY;
    }
    else {
      return false;
    }
    *gap = front(*row);
    return true;
  }

  void pop(Row r) { /* :100-126 */
    if(held[r]) {
      held[r] = false;
    }
    else {
      ++at[r];
    }
  }

  void push_back(Row r, Range gap) { /* :64-76 */
    if(held[r]) {
      throw Failure(ALREADY_UNNEXT);
    }
    hold[r] = gap;
    held[r] = true;
  }
};

/* a9: m_delta_builder.hh:9-87, m_delta_builder.cc:7-22 */
struct SegmentBuilder {
  long ref_start, ref_pos, query_start, query_pos;
  Gaps gaps[2];
  bool query_mirrored;
  long query_columns;
  std::pair<std::string, std::string> names;
  std::pair<long, long> lengths;

  void restart(long r, long q) { /* :22-30 */
    gaps[0].clear();
    gaps[1].clear();
    ref_start = ref_pos = r;
    query_start = query_pos = q;
  }

  void add_gap(Row row, Range diff) { /* :32-63 */
    long walked = (row == ROW_REF ? ref_pos - ref_start : query_pos - query_start) + total_gap_length(gaps[row]);
    gaps[row].push_back(Range{diff.s + walked + 1, diff.e + walked + 1});
    if(row == ROW_REF) {
      ref_pos += diff.s;
      query_pos += diff.e + 1;
    }
    else {
      ref_pos += diff.e + 1;
      query_pos += diff.s;
    }
  }

  void add_offset(long n) { /* :65-68 */
    ref_pos += n;
    query_pos += n;
  }

  long query_column(long pi) const { return query_mirrored ? query_columns - pi + 1 : pi; } /* m_metaprofile.hh:20-27 */

  bool finish(DeltaEntry *out) const { /* m_delta_builder.cc:7-22 */
    if(ref_start == ref_pos || query_start == query_pos) {
      return false;
    }
    out->names = names;
    out->lengths = lengths;
    out->ref = Range{ref_start, ref_pos - 1};
    out->query = Range{query_column(query_start), query_column(query_pos - 1)};
    out->ref_gaps = gaps[0];
    out->query_gaps = gaps[1];
    return true;
  }
};

/* a12: the 4-list merge, m_translate.cc:141-168 (state) and :220-472 (one step) */
struct Merge {
  PairCursor rows;   /* gaps of the two row profiles (query side possibly mirrored) */
  PairCursor delta;  /* gaps of the delta entry's own two rows, in its column coordinates */
  long ref_pos;
  long query_pos;
  long column;
  long last_column;
  SegmentBuilder seg;
  std::vector<DeltaEntry> *sink;

  Merge(Gaps const &pr, Gaps const &pq, Gaps const &dr, Gaps const &dq) : rows(pr, pq), delta(dr, dq) {}

  void consume_delta_piece(Row row, Range d) { /* :220-231 */
    if(row == ROW_REF) {
      ref_pos += d.s;
      query_pos += d.e + 1;
    }
    else {
      ref_pos += d.e + 1;
      query_pos += d.s;
    }
    column += d.e + 1;
  }

  void consume_row_gap(Row row, Range g) { /* :233-244 */
    if(row == ROW_REF) {
      ref_pos += g.e + 1;
      query_pos += g.s;
    }
    else {
      ref_pos += g.s;
      query_pos += g.e + 1;
    }
    column += g.s;
  }

  void close_segment(Row row, Range g) { /* :309-316 and :436-443 */
    seg.add_offset(g.s);
    consume_row_gap(row, g);
    rows.pop(row);
    DeltaEntry e;
    bool emit = seg.finish(&e);
    seg.restart(ref_pos, query_pos);
    if(emit) {
      sink->push_back(e);
    }
  }

  void take_delta_gap(Row row, Range d) { /* :320-322 and :450-452 */
    seg.add_gap(row, d);
    consume_delta_piece(row, d);
    delta.pop(row);
  }

  void split_delta_gap(Row row, Range whole, Range d, long keep) { /* :379-385 and :395-401 */
    Range piece{d.s, d.s + keep - 1};
    Range rest{whole.s + keep, whole.e};
    seg.add_gap(row, piece);
    consume_delta_piece(row, piece);
    delta.pop(row);
    delta.push_back(row, rest);
  }

  Range relative_to_row(Row row, Range g) const {
    long base = row == ROW_REF ? ref_pos : query_pos;
    return Range{g.s - base, g.e - base};
  }

  /* :246-268 */
  bool other_row_gap_within(Row row, Range d) const {
    Row other = row == ROW_REF ? ROW_QUERY : ROW_REF;
    if(!rows.has(other)) {
      return false;
    }
    return relative_to_row(other, rows.front(other)).s <= d.e;
  }

  void step() { /* :279-472 */
    Row prow, drow;
    Range pgap, dgap;
    bool have_p = rows.pick(ref_pos, query_pos, &prow, &pgap);
    bool have_d = delta.pick(column, column, &drow, &dgap);
    if(have_p && have_d) {
      Range g = relative_to_row(prow, pgap);
      Range d{dgap.s - column, dgap.e - column};
      if(g.s <= d.s) {
        close_segment(prow, g);
      }
      else if(d.e < g.s || (prow == drow && !other_row_gap_within(prow, d))) {
        take_delta_gap(drow, d);
      }
      else if(prow == drow) {
        Row other = prow == ROW_REF ? ROW_QUERY : ROW_REF;
        if(rows.has(other)) {
          Range o = relative_to_row(other, rows.front(other));
          split_delta_gap(drow, dgap, d, o.s - d.s);
        }
      }
      else {
        split_delta_gap(drow, dgap, d, g.s - d.s);
      }
    }
    else if(have_p) {
      close_segment(prow, relative_to_row(prow, pgap));
    }
    else if(have_d) {
      take_delta_gap(drow, Range{dgap.s - column, dgap.e - column});
    }
    else if(column <= last_column) { /* :464-470 */
      seg.add_offset(last_column - column + 1);
      DeltaEntry e;
      if(seg.finish(&e)) {
        sink->push_back(e);
      }
    }
  }
};

}  // namespace

/* a13 + the pair filter of :625-647 */
void translate_unit(DeltaEntry const &de_in, Profile const &left, Profile const &right,
                    std::vector<DeltaEntry> *out) {
  Range ref_seq, query_seq;
  if(!overlap(de_in.ref, left.range, &ref_seq) || !overlap(de_in.query, right.range, &query_seq)) {
    return; /* :636-639 */
  }
  /* :210-217 */
  DeltaEntry de = is_forward(de_in.ref) != is_forward(left.range) ? reverse_entry(de_in) : de_in;

  /* :496-511: the entry's two rows as profiles over its own columns */
  Profile d_ref = make_derived_profile("", "", "", de.ref, de.ref_gaps);
  Profile d_query = make_derived_profile("", "", "", de.query, de.query_gaps);
  Range d_ref_cols{profile_idx_of_seq_idx(d_ref, ref_seq.s), profile_idx_of_seq_idx(d_ref, ref_seq.e)};
  Range d_query_cols{profile_idx_of_seq_idx(d_query, query_seq.s), profile_idx_of_seq_idx(d_query, query_seq.e)};
  Range cols;
  if(!overlap(d_ref_cols, d_query_cols, &cols)) {
    return; /* :513 */
  }
  Profile d_ref_sub, d_query_sub;
  bool have_ref_sub = subset_profile(d_ref, cols.s, cols.e, &d_ref_sub);       /* :527-529 */
  bool have_query_sub = subset_profile(d_query, cols.s, cols.e, &d_query_sub); /* :531-533 */
  if(!have_ref_sub || !have_query_sub) {
    return; /* :535 */
  }
  Profile left_sub = subset_seq(left, d_ref_sub.range.s, d_ref_sub.range.e);       /* :539-541 */
  Profile right_sub = subset_seq(right, d_query_sub.range.s, d_query_sub.range.e); /* :543-545 */
  if(range_length(d_ref_sub.range) != range_length(left_sub.range) ||
     range_length(d_query_sub.range) != range_length(right_sub.range)) {
    throw Failure(ASSERT_SUB_LENGTHS); /* :550-551 */
  }

  /* :557-570: walk the right profile backwards when its direction differs from the entry's query row */
  bool mirrored = is_forward(right.range) != is_forward(d_query.range);
  Gaps right_gaps;
  if(mirrored) {
    for(size_t k = right_sub.gaps.size(); k-- > 0;) {
      right_gaps.push_back(Range{right.length - right_sub.gaps[k].e + 1, right.length - right_sub.gaps[k].s + 1});
    }
  }
  else {
    right_gaps = right_sub.gaps;
  }

  long ref_start = profile_idx_of_seq_idx(left, left_sub.range.s); /* :572 */
  long query_start = mirrored ? right.length - profile_idx_of_seq_idx(right, right_sub.range.e) + 1
                              : profile_idx_of_seq_idx(right, right_sub.range.s); /* :575-581 */

  Merge m(left_sub.gaps, right_gaps, d_ref_sub.gaps, d_query_sub.gaps); /* :586-603 */
  m.ref_pos = ref_start;
  m.query_pos = query_start;
  m.column = cols.s;
  m.last_column = cols.e;
  m.sink = out;
  m.seg.query_mirrored = mirrored; /* :605-610 */
  m.seg.query_columns = right.length;
  m.seg.names = std::make_pair(left.major_name, right.major_name);
  m.seg.lengths = std::make_pair(left.length, right.length);
  m.seg.restart(ref_start, query_start);

  /* :612-618.  The reference has no bound here; a bound far above any terminating run turns a
   * would-be endless loop into a reportable failure. */
  long budget = 4 * (long)(left_sub.gaps.size() + right_gaps.size() + d_ref_sub.gaps.size() + d_query_sub.gaps.size()) + 64;
  while(!m.rows.done() || !m.delta.done()) {
    if(budget-- <= 0) {
      throw Failure(STEP_LIMIT);
    }
    m.step();
  }
  m.step();
}

/* --------------------------------------------------------------- a14 driver */

static bool starts_before(Profile const &a, Profile const &b) { /* :170-173,180-182 */
  return forward_of(a.range).s < forward_of(b.range).s;
}

ProfileMap load_profile_map(std::string const &dir) { /* :188-207 */
  std::ifstream in((dir + "/profiles").c_str());
  ProfileMap map;
  Profile p;
  while(read_profile(in, true, &p)) {
    map[p.seq_name].push_back(p);
  }
  for(ProfileMap::iterator it = map.begin(); it != map.end(); ++it) {
    std::sort(it->second.begin(), it->second.end(), starts_before);
  }
  return map;
}

/* :682-695: lower_bound with "profile ends before the entry starts" (:175-178,184-186) */
static size_t first_candidate(std::vector<Profile> const &profiles, Range entry_range) {
  long key = forward_of(entry_range).s;
  std::vector<Profile>::const_iterator it = std::lower_bound(
      profiles.begin(), profiles.end(), key,
      [](Profile const &p, long v) { return forward_of(p.range).e < v; });
  return (size_t)(it - profiles.begin());
}

void units_for_entry(DeltaEntry const &de, std::vector<Profile> const &left, std::vector<Profile> const &right,
                     std::vector<std::pair<size_t, size_t> > *pairs) { /* :698-706 */
  size_t l0 = first_candidate(left, de.ref);
  size_t r0 = first_candidate(right, de.query);
  for(size_t l = l0; l < left.size() && overlap(left[l].range, de.ref, 0); ++l) {
    for(size_t r = r0; r < right.size() && overlap(right[r].range, de.query, 0); ++r) {
      pairs->push_back(std::make_pair(l, r));
    }
  }
}

void translate_stream(ProfileMap const &left, ProfileMap const &right, DeltaReader &reader, DeltaWriter &writer) { /* :650-709 */
  DeltaEntry de;
  while(reader.next(&de)) {
    ProfileMap::const_iterator l = left.find(de.names.first);
    ProfileMap::const_iterator r = right.find(de.names.second);
    if(l == left.end() || r == right.end()) {
      continue;
    }
    std::vector<std::pair<size_t, size_t> > pairs;
    units_for_entry(de, l->second, r->second, &pairs);
    for(size_t k = 0; k < pairs.size(); ++k) {
      std::vector<DeltaEntry> emitted;
      try {
        translate_unit(de, l->second[pairs[k].first], r->second[pairs[k].second], &emitted);
      }
      catch(Failure const &) {
        /* entries written before the failure were already on the stream in the reference */
        for(size_t j = 0; j < emitted.size(); ++j) {
          writer.write(emitted[j]);
        }
        throw;
      }
      for(size_t j = 0; j < emitted.size(); ++j) {
        writer.write(emitted[j]);
      }
    }
  }
}

void translate(std::string const &left_dir, std::string const &right_dir,
               std::vector<std::string> const &delta_paths, std::ostream &out) { /* :713-730 */
  ProfileMap left = load_profile_map(left_dir);
  ProfileMap right = load_profile_map(right_dir);
  DeltaWriter writer(out);
  for(size_t k = 0; k < delta_paths.size(); ++k) {
    std::ifstream in(delta_paths[k].c_str());
    DeltaReader reader(in);
    translate_stream(left, right, reader, writer);
  }
}

int m_translate_main(int argc, char **argv) { /* m_translate_main.cc:19-46 */
  if(argc < 5) {
    std::cerr << "Usage: m_translate <left_profile_dir> <right_profile_dir> <nucmer_file_list> <output_delta_path>" << std::endl;
    return 1;
  }
  std::vector<std::string> paths;
  std::ifstream list(argv[3]);
  std::string line;
  while(std::getline(list, line)) {
    paths.push_back(line);
  }
  std::ofstream out(argv[4]);
  out << (std::string(argv[1]) + "/sequences.fasta") << " " << (std::string(argv[2]) + "/sequences.fasta") << std::endl;
  out << "NUCMER\n";
  try {
    translate(argv[1], argv[2], paths, out);
  }
  catch(Failure const &f) {
    out.flush();
    std::cerr << "oracle m_translate: failure class " << (int)f.code << " (the reference aborts here)" << std::endl;
    return 134;
  }
  return 0;
}

/* ------------------------------------------------------------ a15 sorter */

/* m_sort_delta.cc:19-35 instantiated for the two key shapes; false on full equality */
static bool names_before(DeltaEntry const &l, DeltaEntry const &r) { /* :37-45 */
  if(l.names.first != r.names.first) {
    return l.names.first < r.names.first;
  }
  return l.names.second < r.names.second;
}

static bool coords_before(DeltaEntry const &l, DeltaEntry const &r) { /* :47-55: (rs, qs, re, qe) */
  long a[4] = {l.ref.s, l.query.s, l.ref.e, l.query.e};
  long b[4] = {r.ref.s, r.query.s, r.ref.e, r.query.e};
  for(int k = 0; k < 4; ++k) {
    if(a[k] != b[k]) {
      return a[k] < b[k];
    }
  }
  return false;
}

void sort_delta_entries(std::vector<DeltaEntry> *entries) { /* :58-71 */
  std::vector<DeltaEntry> &v = *entries;
  std::sort(v.begin(), v.end(), names_before);
  size_t group = 0;
  for(size_t k = 0; k < v.size(); ++k) {
    if(v[group].names != v[k].names) {
      std::sort(v.begin() + group, v.begin() + k, coords_before);
      group = k;
    }
  }
  std::sort(v.begin() + group, v.end(), coords_before);
}

int m_sort_delta_main(std::istream &in, std::ostream &out) { /* :73-91; no file header is written */
  DeltaReader reader(in);
  DeltaWriter writer(out);
  std::vector<DeltaEntry> all;
  DeltaEntry de;
  while(reader.next(&de)) {
    all.push_back(de);
  }
  sort_delta_entries(&all);
  for(size_t k = 0; k < all.size(); ++k) {
    writer.write(all[k]);
  }
  return 0;
}

/* --------------------------------------------------------------- a16 MAF */

/* maf_read_stream.hh:23-47: `s genome start size strand src_size text` */
static MafRow parse_maf_row(std::string const &line) {
  std::istringstream iss(line);
  std::string marker, strand;
  MafRow row;
  iss >> marker;
  if(!(iss >> row.genome >> row.start >> row.size >> strand >> row.src_size >> row.text)) {
    throw Failure(PARSE_ERROR);
  }
  row.range = range_of_maf(row.start, row.size, row.src_size, strand == "+");
  return row;
}

/* maf_read_stream.cc:7-45, including: a last line without '\n' leaves eof set and ends the stream; the line
 * that ends a block's `s` rows is consumed */
bool read_maf_block(std::istream &in, MafBlock *out) {
  std::string line;
  while(std::getline(in, line) && ('#' == line[0] || line.empty())) {
  }
  if(in.eof()) {
    return false;
  }
  if('a' != line[0]) {
    return false;
  }
  std::istringstream iss(line);
  char marker;
  iss >> marker;
  MafBlock block;
  if(!(iss >> block.score >> block.label)) {
    throw Failure(PARSE_ERROR);
  }
  while(std::getline(in, line) && 's' == line[0]) {
    block.rows.push_back(parse_maf_row(line));
  }
  *out = block;
  return true;
}

/* --------------------------------------------------------------- a17 coverage */

void MafCoverage::add(MafBlock const &block) { /* maf_analyzer_missing.cc:143-150 with _insert :38-104 */
  for(size_t k = 0; k < block.rows.size(); ++k) {
    MafRow const &row = block.rows[k];
    sizes_[row.genome] = row.src_size;
    std::vector<Range> &v = covered_[row.genome];
    Range r = forward_of(row.range);
    size_t at = 0; /* :25-36: first stored range that starts after r ends */
    while(at < v.size() && !(r.e < v[at].s)) {
      ++at;
    }
    if(v.empty()) {
      v.push_back(r);
      continue;
    }
    bool touches_prev = at != 0 && v[at - 1].e + 1 == r.s;
    bool touches_next = at != v.size() && r.e + 1 == v[at].s;
    if(touches_prev && touches_next) {
      v[at] = Range{v[at - 1].s, v[at].e};
      v.erase(v.begin() + (at - 1));
    }
    else if(touches_next) {
      v[at] = Range{r.s, v[at].e};
    }
    else if(touches_prev) {
      v[at - 1] = Range{v[at - 1].s, r.e};
    }
    else {
      v.insert(v.begin() + at, r);
    }
  }
}

std::map<std::string, std::vector<Range> > MafCoverage::report() const { /* :152-160 with _add_missing :106-135 */
  std::map<std::string, std::vector<Range> > out;
  for(std::map<std::string, std::vector<Range> >::const_iterator it = covered_.begin(); it != covered_.end(); ++it) {
    std::vector<Range> const &v = it->second;
    long size = sizes_.find(it->first)->second;
    std::vector<Range> &missing = out[it->first];
    if(v.empty()) {
      missing.push_back(Range{1, size});
      continue;
    }
    if(1 < v[0].s) {
      missing.push_back(Range{1, v[0].e - 1}); /* :115 uses the END of the first covered range */
    }
    if(v.size() > 1) {
      for(size_t k = 1; k + 1 < v.size(); ++k) { /* :119-126 stops one short of the last range */
        missing.push_back(Range{v[k - 1].e + 1, v[k].s - 1});
      }
    }
    if(v.back().e < size) {
      missing.push_back(Range{v.back().e + 1, size});
    }
  }
  return out;
}

int maf_analyzer_main(int argc, char **argv, std::ostream &out) { /* maf_analyzer.cc:12-38 */
  if(argc < 2) {
    return 1;
  }
  std::ifstream in(argv[1]);
  MafCoverage cov;
  MafBlock block;
  while(read_maf_block(in, &block)) {
    cov.add(block);
  }
  std::map<std::string, std::vector<Range> > rep = cov.report();
  for(std::map<std::string, std::vector<Range> >::const_iterator it = rep.begin(); it != rep.end(); ++it) {
    out << "--------\n";
    for(size_t k = 0; k < it->second.size(); ++k) {
      out << it->first << "\t" << it->second[k].s << "\t" << it->second[k].e << std::endl;
    }
  }
  return 0;
}

}  // namespace pmo
