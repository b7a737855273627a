/*
 * oracle/oracle_cli.cc -- TEST INFRASTRUCTURE ONLY.
 * Command-line twins of the reference binaries, running on the restatement in pm_oracle.cc, so that
 * their outputs can be diffed byte for byte against oracle/_ref/{m_translate,m_sort_delta,
 * maf_analyzer,ref_units}.  One binary per -DORACLE_CLI_* macro (oracle/Makefile).
 */
#include "pm_oracle.hh"

#include <fstream>
#include <iostream>
#include <sstream>

#if defined(ORACLE_CLI_M_TRANSLATE)

int main(int argc, char **argv) {
  std::ios_base::sync_with_stdio(false);
  return pmo::m_translate_main(argc, argv);
}

#elif defined(ORACLE_CLI_M_SORT_DELTA)

int main() {
  std::ios_base::sync_with_stdio(false);
  try {
    return pmo::m_sort_delta_main(std::cin, std::cout);
  }
  catch(pmo::Failure const &f) {
    std::cout.flush();
    std::cerr << "oracle m_sort_delta: failure class " << (int)f.code << std::endl;
    return 134;
  }
}

#elif defined(ORACLE_CLI_MAF_ANALYZER)

int main(int argc, char **argv) {
  try {
    return pmo::maf_analyzer_main(argc, argv, std::cout);
  }
  catch(pmo::Failure const &f) {
    std::cout.flush();
    std::cerr << "oracle maf_analyzer: failure class " << (int)f.code << std::endl;
    return 134;
  }
}

#elif defined(ORACLE_CLI_UNITS)

/* Same command language as oracle/ref_units_driver.cc (documented there). */
static void print_profile(pmo::Profile const &p) {
  std::cout << "PROFILE " << p.range.s << ' ' << p.range.e << ' ' << p.length << ' ' << p.gaps.size();
  for(size_t k = 0; k < p.gaps.size(); ++k) {
    std::cout << ' ' << p.gaps[k].s << ' ' << p.gaps[k].e;
  }
  std::cout << '\n';
}

static pmo::Gaps read_gaps(std::istringstream &iss) {
  pmo::Gaps g;
  long n = 0;
  iss >> n;
  for(long k = 0; k < n; ++k) {
    long s, e;
    iss >> s >> e;
    g.push_back(pmo::Range{s, e});
  }
  return g;
}

static void print_record(pmo::Profile const &p) {
  std::cout << "PREC " << p.major_name << ' ' << p.minor_name << ' ' << p.seq_name << ' ' << p.range.s << ' ' << p.range.e << ' ' << p.length << ' '
            << p.src_size << ' ' << p.gaps.size();
  for(size_t k = 0; k < p.gaps.size(); ++k) {
    std::cout << ' ' << p.gaps[k].s << ' ' << p.gaps[k].e;
  }
  std::cout << ' ' << (p.text.empty() ? std::string("-") : p.text) << '\n';
}

static void print_gaps(pmo::Gaps const &g) {
  std::cout << ' ' << g.size();
  for(size_t k = 0; k < g.size(); ++k) {
    std::cout << ' ' << g[k].s << ' ' << g[k].e;
  }
}

int main() {
  std::ios_base::sync_with_stdio(false);
  pmo::Profile cur;
  cur.range = pmo::Range{1, 1};
  cur.length = 1;
  cur.src_size = 0;
  pmo::DeltaEntry cur_d;
  cur_d.ref = cur_d.query = pmo::Range{1, 1};
  const char *parse_error_name = "exception";
  std::string line;
  while(std::getline(std::cin, line)) {
    std::istringstream iss(line);
    std::string cmd;
    if(!(iss >> cmd)) {
      continue;
    }
    try {
      parse_error_name = cmd == "mafread" ? "Maf_parse_error" : "Profile_read_error";
      if(cmd == "profile") {
        long s, e, len;
        iss >> s >> e >> len;
        cur = pmo::Profile();
        cur.range = pmo::Range{s, e};
        cur.length = len;
        cur.src_size = 0;
        cur.gaps = read_gaps(iss);
        std::cout << "OK\n";
      }
      else if(cmd == "p2s") {
        long si;
        iss >> si;
        std::cout << "IDX " << pmo::profile_idx_of_seq_idx(cur, si) << '\n';
      }
      else if(cmd == "s2p") {
        long pi, v;
        iss >> pi;
        if(pmo::seq_idx_of_profile_idx(cur, pi, &v)) {
          std::cout << "IDX " << v << '\n';
        }
        else {
          std::cout << "NONE\n";
        }
      }
      else if(cmd == "sub") {
        long s, e;
        iss >> s >> e;
        pmo::Profile sub;
        if(pmo::subset_profile(cur, s, e, &sub)) {
          print_profile(sub);
        }
        else {
          std::cout << "NONE\n";
        }
      }
      else if(cmd == "subseq") {
        long s, e;
        iss >> s >> e;
        print_profile(pmo::subset_seq(cur, s, e));
      }
      else if(cmd == "delta") {
        long rs, re, qs, qe;
        iss >> rs >> re >> qs >> qe;
        cur_d = pmo::DeltaEntry();
        cur_d.ref = pmo::Range{rs, re};
        cur_d.query = pmo::Range{qs, qe};
        cur_d.ref_gaps = read_gaps(iss);
        cur_d.query_gaps = read_gaps(iss);
        std::cout << "OK\n";
      }
      else if(cmd == "dparse") {
        long rs, re, qs, qe, v;
        iss >> rs >> re >> qs >> qe;
        std::ostringstream text;
        text << "a b\nNUCMER\n>r q 1 1\n" << rs << ' ' << re << ' ' << qs << ' ' << qe << " 0 0 0\n";
        while(iss >> v) {
          text << v << '\n';
        }
        text << "0\n";
        std::istringstream in(text.str());
        pmo::DeltaReader reader(in);
        if(!reader.next(&cur_d)) {
          throw pmo::Failure(pmo::IS_NONE);
        }
        std::cout << "DELTA " << cur_d.ref.s << ' ' << cur_d.ref.e << ' ' << cur_d.query.s << ' ' << cur_d.query.e;
        print_gaps(cur_d.ref_gaps);
        print_gaps(cur_d.query_gaps);
        std::cout << '\n';
      }
      else if(cmd == "drev") {
        pmo::DeltaEntry r = pmo::reverse_entry(cur_d);
        std::cout << "DELTA " << r.ref.s << ' ' << r.ref.e << ' ' << r.query.s << ' ' << r.query.e;
        print_gaps(r.ref_gaps);
        print_gaps(r.query_gaps);
        std::cout << '\n';
      }
      else if(cmd == "ofmaf") {
        long start, size, src_size;
        std::string strand;
        iss >> start >> size >> src_size >> strand;
        pmo::Range r = pmo::range_of_maf(start, size, src_size, strand == "+");
        std::cout << "RANGE " << r.s << ' ' << r.e << '\n';
      }
      else if(cmd == "mafread") {
        std::string path;
        iss >> path;
        std::ifstream in(path.c_str());
        pmo::MafBlock b;
        while(pmo::read_maf_block(in, &b)) {
          std::cout << "ENTRY " << b.score << ' ' << b.label << ' ' << b.rows.size() << '\n';
          for(size_t r = 0; r < b.rows.size(); ++r) {
            pmo::MafRow const &a = b.rows[r];
            std::cout << "ALN " << a.genome << ' ' << a.start << ' ' << a.size << ' ' << a.src_size << ' ' << a.range.s << ' ' << a.range.e << ' '
                      << a.text << '\n';
          }
        }
        std::cout << "END\n";
      }
      else if(cmd == "profread" || cmd == "profpick") {
        std::string path;
        long arg = 0;
        iss >> path >> arg;
        std::ifstream in(path.c_str());
        pmo::Profile p;
        long k = 0;
        bool picked = false;
        for(; pmo::read_profile(in, cmd == "profread" && arg != 0, &p); ++k) {
          if(cmd == "profread") {
            cur = p;
            print_record(cur);
          }
          else if(k == arg) {
            cur = p;
            print_record(cur);
            picked = true;
            break;
          }
        }
        if(cmd == "profread") {
          std::cout << "END\n";
        }
        else if(!picked) {
          std::cout << "NONE\n";
        }
      }
      else if(cmd == "d2o") {
        std::vector<long> o = pmo::offsets_of_gaps(cur_d);
        std::cout << "OFFSETS";
        for(size_t k = 0; k < o.size(); ++k) {
          std::cout << ' ' << o[k];
        }
        std::cout << '\n';
      }
      else {
        std::cout << "BADCMD\n";
      }
    }
    catch(pmo::Failure const &f) {
      switch(f.code) {
      case pmo::SEQ_IDX_OUT_OF_RANGE:
        std::cout << "EXC Seq_idx_out_of_range\n";
        break;
      case pmo::PROFILE_IDX_OUT_OF_RANGE:
        std::cout << "EXC Profile_idx_out_of_range\n";
        break;
      case pmo::IS_NONE:
        std::cout << "EXC Is_none_error\n";
        break;
      case pmo::PARSE_ERROR:
        std::cout << "EXC " << parse_error_name << "\n";
        break;
      default:
        std::cout << "EXC exception\n";
      }
    }
  }
  return 0;
}

#else
#error "define one ORACLE_CLI_* macro"
#endif
