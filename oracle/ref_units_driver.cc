/*
 * oracle/ref_units_driver.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * A small command interpreter, written for this repo, that is linked against
 * the upstream reference's own lib/profiles_lib objects (compiled from
 * /root/reference where they lie; see oracle/Makefile target `ref`).  It lets
 * the tests probe the reference's unit-level functions
 * one call at a time.  oracle_cli.cc (-DORACLE_CLI_UNITS) is the twin that runs
 * the same command language on this repo's restatement; the two outputs are
 * diffed byte for byte (tests/test_oracle_vs_ref.py, tests/golden/units_*.txt).
 *
 * Command language (one command per line, integers are decimal longs):
 *   profile <start> <end> <p_length> <ngaps> {<gs> <ge>}*   set current profile
 *   p2s <seq_idx>          profile_idx_of_seq_idx      (m_profile.cc:91-112)
 *   s2p <profile_idx>      seq_idx_of_profile_idx      (m_profile.cc:114-149)
 *   sub <s> <e>            subset_profile              (m_profile.cc:160-206)
 *   subseq <s> <e>         subset_seq                  (m_profile.cc:208-212)
 *   delta <rs> <re> <qs> <qe> <nr> {<gs> <ge>}* <nq> {<gs> <ge>}*   set current delta entry
 *   drev                   M_delta_entry::reverse      (m_delta.cc:94-146)
 *   d2o                    deltas_of_gaps              (m_delta_stream_writer.hh:14-53)
 *   dparse <rs> <re> <qs> <qe> {<offset>}*   M_delta_stream::next + _split_gaps (m_delta.cc:148-220,14-68)
 *                          on a one-entry delta text built from the arguments; sets the current delta entry
 *   ofmaf <start> <size> <src_size> <+|->   of_maf  (m_range.hh:106-115): the range of a MAF `s` line
 *   mafread <path>         Maf_read_stream::next until it returns none (maf_read_stream.cc:7-45, the `s` lines through
 *                          Maf_entry_alignment's constructor, maf_read_stream.hh:20-45): one line ENTRY <score> <label> <rows> per
 *                          block, one line ALN <genome> <start> <size> <src_size> <range start> <range end> <text> per row, then END
 *   profread <path> <lite 0|1>   read_profile_file until none (m_profile.cc:15-85): one line
 *                          PREC <major> <minor> <seq> <start> <end> <length> <src_size> <ngaps> {<gs> <ge>}* <text or -> per record
 *                          (the record becomes the current profile: p2s / s2p / sub work on the last one read), then END
 *   profpick <path> <k>    the same file again, the k-th record (0-based) becomes the current profile; prints its PREC line
 * Every command but mafread / profread prints exactly one line.
 */
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <m_option.hh>
#include <m_range.hh>
#include <m_profile.hh>
#include <m_delta.hh>
#include <m_delta_stream_writer.hh>
#include <maf_read_stream.hh>

using namespace Para_mugsy;

typedef std::vector<M_range<M_profile_idx> > gaps_t;

static void print_profile(M_profile const &p) {
  std::cout << "PROFILE " << p.p_range.get_start() << ' ' << p.p_range.get_end() << ' ' << p.p_length << ' ' << p.p_gaps.size();
  for(gaps_t::const_iterator i = p.p_gaps.begin(); i != p.p_gaps.end(); ++i) {
    std::cout << ' ' << i->get_start() << ' ' << i->get_end();
  }
  std::cout << '\n';
}

static void print_record(M_profile const &p) {
  std::cout << "PREC " << p.p_major_name << ' ' << p.p_minor_name << ' ' << p.p_seq_name << ' ' << p.p_range.get_start() << ' '
            << p.p_range.get_end() << ' ' << p.p_length << ' ' << p.p_src_size << ' ' << p.p_gaps.size();
  for(gaps_t::const_iterator i = p.p_gaps.begin(); i != p.p_gaps.end(); ++i) {
    std::cout << ' ' << i->get_start() << ' ' << i->get_end();
  }
  std::cout << ' ' << (p.p_seq_text.empty() ? std::string("-") : p.p_seq_text) << '\n';
}

static gaps_t read_gaps(std::istringstream &iss) {
  gaps_t g;
  long n = 0;
  iss >> n;
  for(long k = 0; k < n; ++k) {
    long s, e;
    iss >> s >> e;
    g.push_back(M_range<M_profile_idx>(s, e));
  }
  return g;
}

int main() {
  std::ios_base::sync_with_stdio(false);
  M_profile cur("", "", "", M_range<M_seq_idx>(1, 1), 1, 0, gaps_t(), "");
  M_delta_entry cur_d(std::make_pair(std::string(), std::string()), std::make_pair(0L, 0L),
                      M_range<M_seq_idx>(1, 1), M_range<M_seq_idx>(1, 1), gaps_t(), gaps_t());
  std::string line;
  while(std::getline(std::cin, line)) {
    std::istringstream iss(line);
    std::string cmd;
    if(!(iss >> cmd)) {
      continue;
    }
    try {
      if(cmd == "profile") {
        long s, e, len;
        iss >> s >> e >> len;
        gaps_t g = read_gaps(iss);
        cur = M_profile("", "", "", M_range<M_seq_idx>(s, e), len, 0, g, "");
        std::cout << "OK\n";
      }
      else if(cmd == "p2s") {
        long si;
        iss >> si;
        std::cout << "IDX " << profile_idx_of_seq_idx(cur, si) << '\n';
      }
      else if(cmd == "s2p") {
        long pi;
        iss >> pi;
        M_option<M_seq_idx> o = seq_idx_of_profile_idx(cur, pi);
        if(o) {
          std::cout << "IDX " << o.value() << '\n';
        }
        else {
          std::cout << "NONE\n";
        }
      }
      else if(cmd == "sub") {
        long s, e;
        iss >> s >> e;
        M_option<M_profile> o = subset_profile(cur, s, e);
        if(o) {
          print_profile(o.value());
        }
        else {
          std::cout << "NONE\n";
        }
      }
      else if(cmd == "subseq") {
        long s, e;
        iss >> s >> e;
        print_profile(subset_seq(cur, s, e));
      }
      else if(cmd == "delta") {
        long rs, re, qs, qe;
        iss >> rs >> re >> qs >> qe;
        gaps_t rg = read_gaps(iss);
        gaps_t qg = read_gaps(iss);
        cur_d = M_delta_entry(std::make_pair(std::string(), std::string()), std::make_pair(0L, 0L),
                              M_range<M_seq_idx>(rs, re), M_range<M_seq_idx>(qs, qe), rg, qg);
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
        M_delta_stream ds(in);
        cur_d = ds.next().value();
        M_delta_entry const &r = cur_d;
        std::cout << "DELTA " << r.ref_range.get_start() << ' ' << r.ref_range.get_end() << ' '
                  << r.query_range.get_start() << ' ' << r.query_range.get_end() << ' ' << r.ref_gaps.size();
        for(gaps_t::const_iterator i = r.ref_gaps.begin(); i != r.ref_gaps.end(); ++i) {
          std::cout << ' ' << i->get_start() << ' ' << i->get_end();
        }
        std::cout << ' ' << r.query_gaps.size();
        for(gaps_t::const_iterator i = r.query_gaps.begin(); i != r.query_gaps.end(); ++i) {
          std::cout << ' ' << i->get_start() << ' ' << i->get_end();
        }
        std::cout << '\n';
      }
      else if(cmd == "drev") {
        M_delta_entry r = cur_d.reverse();
        std::cout << "DELTA " << r.ref_range.get_start() << ' ' << r.ref_range.get_end() << ' '
                  << r.query_range.get_start() << ' ' << r.query_range.get_end() << ' ' << r.ref_gaps.size();
        for(gaps_t::const_iterator i = r.ref_gaps.begin(); i != r.ref_gaps.end(); ++i) {
          std::cout << ' ' << i->get_start() << ' ' << i->get_end();
        }
        std::cout << ' ' << r.query_gaps.size();
        for(gaps_t::const_iterator i = r.query_gaps.begin(); i != r.query_gaps.end(); ++i) {
          std::cout << ' ' << i->get_start() << ' ' << i->get_end();
        }
        std::cout << '\n';
      }
      else if(cmd == "ofmaf") {
        long start, size, src_size;
        std::string strand;
        iss >> start >> size >> src_size >> strand;
        M_range<long> r = of_maf(start, size, src_size, strand == "+" ? D_FORWARD : D_REVERSE);
        std::cout << "RANGE " << r.get_start() << ' ' << r.get_end() << '\n';
      }
      else if(cmd == "mafread") {
        std::string path;
        iss >> path;
        std::ifstream in(path.c_str());
        Maf_read_stream mrs(in);
        for(;;) { // (M_option has no assignment)
          M_option<Maf_entry> e = mrs.next();
          if(!e) {
            break;
          }
          Maf_entry const &me = e.value();
          std::cout << "ENTRY " << me.score() << ' ' << me.label() << ' ' << (me.alignments_end() - me.alignments_begin()) << '\n';
          for(std::vector<Maf_entry_alignment>::const_iterator a = me.alignments_begin(); a != me.alignments_end(); ++a) {
            std::cout << "ALN " << a->genome_name() << ' ' << a->start() << ' ' << a->size() << ' ' << a->src_size() << ' '
                      << a->range().get_start() << ' ' << a->range().get_end() << ' ' << a->text() << '\n';
          }
        }
        std::cout << "END\n";
      }
      else if(cmd == "profread" || cmd == "profpick") {
        std::string path;
        long arg = 0;
        iss >> path >> arg;
        std::ifstream in(path.c_str());
        long k = 0;
        bool picked = false;
        for(;; ++k) {
          M_option<M_profile> p = read_profile_file(cmd == "profread" && arg != 0, in);
          if(!p) {
            break;
          }
          if(cmd == "profread") {
            cur = p.value();
            print_record(cur);
          }
          else if(k == arg) {
            cur = p.value();
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
        std::vector<long> o = deltas_of_gaps(cur_d);
        std::cout << "OFFSETS";
        for(std::vector<long>::const_iterator i = o.begin(); i != o.end(); ++i) {
          std::cout << ' ' << *i;
        }
        std::cout << '\n';
      }
      else {
        std::cout << "BADCMD\n";
      }
    }
    catch(Maf_parse_error const &) {
      std::cout << "EXC Maf_parse_error\n";
    }
    catch(Profile_read_error const &) {
      std::cout << "EXC Profile_read_error\n";
    }
    catch(Seq_idx_out_of_range const &) {
      std::cout << "EXC Seq_idx_out_of_range\n";
    }
    catch(Profile_idx_out_of_range const &) {
      std::cout << "EXC Profile_idx_out_of_range\n";
    }
    catch(Is_none_error const &) {
      std::cout << "EXC Is_none_error\n";
    }
    catch(std::exception const &) {
      std::cout << "EXC exception\n";
    }
  }
  return 0;
}
