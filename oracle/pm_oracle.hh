/*
 * oracle/pm_oracle.hh -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain, scalar C++ restatement of the algorithm of paramugsy's "profiles"
 * translate path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or
 * call anything under oracle/.  The product (paramugsy_amd/) never does.
 *
 * PARITY STATUS: pinned.  The restatement is checked byte for byte against the
 * upstream reference itself, compiled from its own sources into oracle/_ref/
 * (oracle/Makefile `ref`), on the committed fixtures under tests/golden/ and on
 * seeded fuzz inputs (tests/test_oracle_vs_ref.py), plus the one known-answer
 * vector the reference carries (m_delta.cc:43-49).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/).  Where the reference throws an
 * exception or trips an assert (which ends the process there), the oracle throws
 * pmo::Failure carrying a code, so a test can check the failure class too.
 */
#ifndef PM_ORACLE_HH
#define PM_ORACLE_HH

#include <cstdint>
#include <iosfwd>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace pmo {

/* Failure classes; numeric values are shared with include/paramugsy_amd.h (PM_ST_*) */
enum Code {
  OK = 0,
  SEQ_IDX_OUT_OF_RANGE = 1,      /* m_profile.hh:15  */
  PROFILE_IDX_OUT_OF_RANGE = 2,  /* m_profile.hh:16  */
  IS_NONE = 3,                   /* m_option.hh:14   */
  ASSERT_GAP_BEHIND = 4,         /* m_translate.cc:42-43 */
  ASSERT_SUB_LENGTHS = 5,        /* m_translate.cc:550-551 */
  ALREADY_UNNEXT = 6,            /* m_translate.cc:22,74 */
  STEP_LIMIT = 7,                /* no reference counterpart: the reference would not terminate */
  PARSE_ERROR = 20               /* Profile_read_error / Delta_stream_parse_error / Maf_parse_error */
};

struct Failure {
  Code code;
  explicit Failure(Code c) : code(c) {}
};

/* a1: m_range.hh:11-58.  1-indexed, inclusive, direction = order of the two ends. */
struct Range {
  long s;
  long e;
};

inline bool is_forward(Range r) { return r.s <= r.e; }                        /* m_range.hh:36 */
inline long range_length(Range r) { return (r.s <= r.e ? r.e - r.s : r.s - r.e) + 1; } /* m_range.hh:34 */
inline Range forward_of(Range r) { return is_forward(r) ? r : Range{r.e, r.s}; }       /* m_range.hh:40-47,67-78 */
bool overlap(Range a, Range b, Range *out);                                   /* m_range.hh:80-94 */
bool contains(Range r, long v);                                               /* m_range.hh:49-52 */
Range range_of_maf(long start, long size, long src_size, bool forward);       /* m_range.hh:106-115 */

typedef std::vector<Range> Gaps;

/* a2: m_profile.hh:26-100 */
struct Profile {
  std::string major_name;
  std::string minor_name;
  std::string seq_name;
  Range range;
  long length;   /* p_length: columns */
  long src_size;
  Gaps gaps;     /* in profile (column) coordinates */
  std::string text;
};

/* the 5-argument constructor of m_profile.hh:46-63: length = |range| + sum |gap| */
Profile make_derived_profile(std::string const &major_name, std::string const &minor_name,
                             std::string const &seq_name, Range range, Gaps const &gaps);

bool read_profile(std::istream &in, bool lite, Profile *out);      /* m_profile.cc:15-85 */
long profile_idx_of_seq_idx(Profile const &p, long si);            /* m_profile.cc:91-112 */
bool seq_idx_of_profile_idx(Profile const &p, long pi, long *out); /* m_profile.cc:114-149 */
bool subset_profile(Profile const &p, long s, long e, Profile *out); /* m_profile.cc:160-206 */
Profile subset_seq(Profile const &p, long s, long e);              /* m_profile.cc:208-212 */

/* a7/a8: m_delta.hh:17-62 */
struct DeltaEntry {
  std::pair<std::string, std::string> names;
  std::pair<long, long> lengths;
  Range ref;
  Range query;
  Gaps ref_gaps;
  Gaps query_gaps;
};

void split_gaps(std::vector<long> const &offsets, Gaps *ref_gaps, Gaps *query_gaps); /* m_delta.cc:14-68 */
DeltaEntry reverse_entry(DeltaEntry const &de);                                        /* m_delta.cc:94-146 */
std::vector<long> offsets_of_gaps(DeltaEntry const &de);                               /* m_delta_stream_writer.hh:14-53 */

/* a7: m_delta.hh:64-79, m_delta.cc:72-92,148-220 */
class DeltaReader {
public:
  explicit DeltaReader(std::istream &in);
  bool next(DeltaEntry *out);
  std::pair<std::string, std::string> files;
  std::string kind;

private:
  std::istream &in_;
  std::pair<std::string, std::string> names_;
  std::pair<long, long> lengths_;
};

/* a10: m_delta_stream_writer.hh:55-82 */
class DeltaWriter {
public:
  explicit DeltaWriter(std::ostream &out) : out_(out) {}
  void write(DeltaEntry const &de);

private:
  std::ostream &out_;
  std::pair<std::string, std::string> last_names_;
};

/* a11-a13: one (delta entry x left row x right row) work unit.
 * m_translate.cc:625-647 (+ :474-621, :279-472).  Appends emitted entries to *out. */
void translate_unit(DeltaEntry const &de, Profile const &left, Profile const &right,
                    std::vector<DeltaEntry> *out);

/* a14 */
typedef std::map<std::string, std::vector<Profile> > ProfileMap;
ProfileMap load_profile_map(std::string const &dir);                     /* m_translate.cc:188-207 */
/* which (left index, right index) pairs the reference visits for one entry, in order;
 * m_translate.cc:666-707 */
void units_for_entry(DeltaEntry const &de, std::vector<Profile> const &left, std::vector<Profile> const &right,
                     std::vector<std::pair<size_t, size_t> > *pairs);
void translate_stream(ProfileMap const &left, ProfileMap const &right, DeltaReader &reader, DeltaWriter &writer); /* m_translate.cc:650-709 */
void translate(std::string const &left_dir, std::string const &right_dir,
               std::vector<std::string> const &delta_paths, std::ostream &out);  /* m_translate.cc:713-730 */
int m_translate_main(int argc, char **argv);                                     /* m_translate_main.cc:19-46 */

/* a15: lib/profiles_cpp/m_sort_delta.cc:58-91 */
void sort_delta_entries(std::vector<DeltaEntry> *entries);
int m_sort_delta_main(std::istream &in, std::ostream &out);

/* a16: lib/profiles_lib/maf_read_stream.{hh,cc} */
struct MafRow {
  std::string genome;
  long start;
  long size;
  long src_size;
  std::string text;
  Range range;
};
struct MafBlock {
  std::string score;
  std::string label;
  std::vector<MafRow> rows;
};
bool read_maf_block(std::istream &in, MafBlock *out);  /* maf_read_stream.cc:7-45 */

/* a17: lib/profiles_cpp/maf_analyzer_missing.cc */
class MafCoverage {
public:
  void add(MafBlock const &block);                                        /* :143-150 */
  std::map<std::string, std::vector<Range> > report() const;              /* :152-160, :106-135 */
  std::map<std::string, std::vector<Range> > const &covered() const { return covered_; }

private:
  std::map<std::string, std::vector<Range> > covered_;
  std::map<std::string, long> sizes_;
};
int maf_analyzer_main(int argc, char **argv, std::ostream &out);          /* maf_analyzer.cc:12-38 */

}  // namespace pmo

#endif
