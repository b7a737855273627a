// mugsy_profiles_main.cc -- the multi-command executable of lib/profiles (m_profiles_cli.ml:6-21), for the commands
// this repo implements on the GPU.  Flags are OCaml Arg style, as the task script passes them
// (lib/base/mugsy_profiles_task.ml:46-50):
//   mugsy_profiles make -in_maf <maf> -out_dir <dir> -basename <name>            (lib/profiles/m_make.ml:66-93)
//   mugsy_profiles translate -profiles_left <dir> -profiles_right <dir> -nucmer_list <file> -out_delta <file>
//                                                                                (lib/profiles/m_translate.ml:780-851)
//   mugsy_profiles untranslate -profile_paths_list <file> -in_maf <maf> -out_maf <maf>   (lib/profiles/m_untranslate.ml:168-221)
// The other commands (maf_to_xmfa, fasta_to_maf) are format converters outside the path; they exit 2 here.
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <fcntl.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>

#include "serve_common.hpp"

#include "../../include/paramugsy_amd.h"

static void mkdir_p(const std::string &path) { // Shell.mkdir ~p:(), m_make.ml:92
  std::string cur;
  for(size_t k = 0; k <= path.size(); ++k) {
    if(k == path.size() || path[k] == '/') {
      if(!cur.empty()) {
        mkdir(cur.c_str(), 0777);
      }
    }
    if(k < path.size()) {
      cur.push_back(path[k]);
    }
  }
}

static std::vector<std::string> read_list(const std::string &path) {
  std::vector<std::string> out;
  std::ifstream list(path.c_str());
  std::string line;
  while(std::getline(list, line)) {
    out.push_back(line);
  }
  return out;
}

// "0,1,2" -> {0, 1, 2}; empty when the text is empty or malformed
static std::vector<int> parse_devices(const std::string &text) {
  std::vector<int> out;
  size_t at = 0;
  while(at < text.size()) {
    size_t e = text.find(',', at);
    if(e == std::string::npos) {
      e = text.size();
    }
    const std::string tok = text.substr(at, e - at);
    if(tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) {
      return std::vector<int>();
    }
    out.push_back(atoi(tok.c_str()));
    at = e + 1;
  }
  return out;
}

// One command; returns the process exit code it stands for (0 ok, 2 = uncaught OCaml exception / bad flags).
static int run_command(const std::string &cmd, std::map<std::string, std::string> &flag, int device) {
  int rc;
  if(cmd == "make") {
    if(flag["-basename"].empty() || flag["-out_dir"].empty() || flag["-in_maf"].empty()) {
      fprintf(stderr, "Must provide -basename, -out_dir and -in_maf\n"); // m_make.ml:78-83 raise Failure
      return 2;
    }
    mkdir_p(flag["-out_dir"]);
    rc = pm_profiles_make(flag["-in_maf"].c_str(), flag["-out_dir"].c_str(), flag["-basename"].c_str(), device);
  }
  else if(cmd == "translate") {
    if(flag["-profiles_left"].empty() || flag["-profiles_right"].empty() || flag["-nucmer_list"].empty() || flag["-out_delta"].empty()) {
      fprintf(stderr, "Must provide -profiles_left, -profiles_right, -nucmer_list and -out_delta\n");
      return 2;
    }
    std::vector<std::string> paths = read_list(flag["-nucmer_list"]);
    std::vector<const char *> cpaths;
    for(size_t k = 0; k < paths.size(); ++k) {
      cpaths.push_back(paths[k].c_str());
    }
    // new in this build: -devices 0,1,... (or PARAMUGSY_DEVICES) spreads the delta-file list over several GPUs of the node, one
    // host thread per device, output identical to the one-device run (pm_translate_files_multi)
    const char *devs_env = getenv("PARAMUGSY_DEVICES");
    const std::string devs_text = !flag["-devices"].empty() ? flag["-devices"] : (devs_env ? devs_env : "");
    std::vector<int> devs = parse_devices(devs_text);
    if(!devs_text.empty() && devs.empty()) {
      fprintf(stderr, "-devices takes a comma-separated list of device indices\n");
      return 2;
    }
    if(devs.size() == 1) { // a list of one is the one-device path on that device (as in m_translate)
      device = devs[0];
    }
    if(devs.size() > 1) {
      rc = pm_translate_files_multi(flag["-profiles_left"].c_str(), flag["-profiles_right"].c_str(), cpaths.data(), (int)cpaths.size(),
                                    flag["-out_delta"].c_str(), devs.data(), (int)devs.size());
    }
    else {
      rc = pm_translate_files(flag["-profiles_left"].c_str(), flag["-profiles_right"].c_str(), cpaths.data(), (int)cpaths.size(),
                              flag["-out_delta"].c_str(), device);
    }
  }
  else if(cmd == "align") {
    // new in this build (the DP has no reference counterpart): two MAF files with the same number of blocks -> a MAF
    // file of merged blocks; -devices 0,1,... spreads the pairs over several GPUs (pm_dp_align_maf_multi)
    if(flag["-left_maf"].empty() || flag["-right_maf"].empty() || flag["-out_maf"].empty()) {
      fprintf(stderr, "Must provide -left_maf, -right_maf and -out_maf\n");
      return 2;
    }
    pm_dp_params_t prm;
    const int rows = flag["-rows"].empty() ? 1 : atoi(flag["-rows"].c_str()); // penalties scale with the row pairs of a column pair
    for(int a = 0; a < 5; ++a) {
      for(int b = 0; b < 5; ++b) {
        prm.sub[a * 5 + b] = (a == 4 && b == 4) ? 0 : (a == 4 || b == 4) ? -3 : (a == b ? 5 : -4);
      }
    }
    prm.gap_open = (flag["-gap_open"].empty() ? 8 : atoi(flag["-gap_open"].c_str())) * rows * rows;
    prm.gap_extend = (flag["-gap_extend"].empty() ? 2 : atoi(flag["-gap_extend"].c_str())) * rows * rows;
    const char *devs_env = getenv("PARAMUGSY_DEVICES");
    const std::string devs_text = !flag["-devices"].empty() ? flag["-devices"] : (devs_env ? devs_env : "");
    std::vector<int> devs = parse_devices(devs_text);
    if(!devs_text.empty() && devs.empty()) {
      fprintf(stderr, "-devices takes a comma-separated list of device indices\n");
      return 2;
    }
    if(devs.size() == 1) {
      device = devs[0];
    }
    if(devs.size() > 1) {
      rc = pm_dp_align_maf_multi(flag["-left_maf"].c_str(), flag["-right_maf"].c_str(), &prm, flag["-out_maf"].c_str(), devs.data(), (int)devs.size());
    }
    else {
      rc = pm_dp_align_maf(flag["-left_maf"].c_str(), flag["-right_maf"].c_str(), &prm, flag["-out_maf"].c_str(), device);
    }
  }
  else if(cmd == "stage") {
    // new in this build: the make + make + translate prefix of lib/base/mugsy_profiles_task.ml:40-58 in one process
    const char *need[] = {"-left_maf", "-left_dir", "-left_basename", "-right_maf", "-right_dir", "-right_basename", "-nucmer_list", "-out_delta"};
    for(const char *n : need) {
      if(flag[n].empty()) {
        fprintf(stderr, "Must provide -left_maf, -left_dir, -left_basename, -right_maf, -right_dir, -right_basename, -nucmer_list and -out_delta\n");
        return 2;
      }
    }
    mkdir_p(flag["-left_dir"]);
    mkdir_p(flag["-right_dir"]);
    std::vector<std::string> paths = read_list(flag["-nucmer_list"]);
    std::vector<const char *> cpaths;
    for(size_t k = 0; k < paths.size(); ++k) {
      cpaths.push_back(paths[k].c_str());
    }
    rc = pm_stage_files(flag["-left_maf"].c_str(), flag["-left_dir"].c_str(), flag["-left_basename"].c_str(), flag["-right_maf"].c_str(),
                        flag["-right_dir"].c_str(), flag["-right_basename"].c_str(), cpaths.data(), (int)cpaths.size(),
                        flag["-out_delta"].c_str(), device);
  }
  else if(cmd == "untranslate") {
    if(flag["-profile_paths_list"].empty() || flag["-in_maf"].empty() || flag["-out_maf"].empty()) {
      fprintf(stderr, "Must provide -profile_paths_list, -in_maf and -out_maf\n"); // m_untranslate.ml:183-188
      return 2;
    }
    std::vector<std::string> dirs = read_list(flag["-profile_paths_list"]);
    std::vector<const char *> cdirs;
    for(size_t k = 0; k < dirs.size(); ++k) {
      cdirs.push_back(dirs[k].c_str());
    }
    rc = pm_untranslate(cdirs.data(), (int)cdirs.size(), flag["-in_maf"].c_str(), flag["-out_maf"].c_str(), device);
  }
  else {
    fprintf(stderr, "mugsy_profiles: command '%s' is not implemented by this build\n", cmd.c_str());
    return 2;
  }
  if(rc != PM_OK) {
    fprintf(stderr, "mugsy_profiles %s: error %d: %s\n", cmd.c_str(), rc, pm_last_error());
    return 2; // an uncaught OCaml exception exits 2
  }
  return 0;
}

// ---- `serve -socket <path>`: the resident worker behind the drop-in m_translate (bin/m_translate asks it before it would bring a HIP
// runtime of its own up, csrc/m_translate_main.cc).  One request per connection, one line, TAB-separated:
//   translate <cwd> <left_dir> <right_dir> <out_path> <devices or -> <n> <delta path> x n       (paths as the client's argv / list had them)
//   stats                                      -> `done 0\njobs <translate requests served so far>\n` (bench.py: was a run served?)
//   quit
// answered by `done <exit code>\n` and, after a failure, the message the client prints.
// `-socket default` listens where the client looks when nobody says (serve_common.hpp: a directory that is the caller's alone).
// Only a peer of the worker's own uid is listened to (SO_PEERCRED; the socket is 0600 besides).  A client gets
// PARAMUGSY_SERVE_REQUEST_TIMEOUT seconds (default 5) to put its one line on the socket and as long to take the answer: one that
// connects and then says nothing (stopped, or killed with the descriptor still held) is dropped and the next one served.  Two
// workers started at once do not unlink each other's socket: the probe-unlink-bind sequence runs under a lock on `<path>.lock`,
// held for the worker's life.  Relative paths are the CLIENT's: they are
// resolved against its working directory; the output's first line holds the strings it was started with (pm_translate_files_as).
// Requests are served one after the other (a second client waits in the listen queue): a node's job is 0.05 s of a GPU.
static std::string client_path(const std::string &cwd, const std::string &p) { return !p.empty() && p[0] == '/' ? p : cwd + "/" + p; }

static bool read_line(int fd, std::string &line, size_t limit = (size_t)64 << 20) {
  line.clear();
  char buf[4096];
  for(;;) {
    const ssize_t n = read(fd, buf, sizeof buf);
    if(n < 0 && errno == EINTR) {
      continue;
    }
    if(n <= 0) { // end of stream, or SO_RCVTIMEO ran out (EAGAIN): no request
      return false;
    }
    for(ssize_t k = 0; k < n; ++k) {
      if(buf[k] == '\n') {
        line.append(buf, (size_t)k);
        return true;
      }
    }
    line.append(buf, (size_t)n);
    if(line.size() > limit) {
      return false;
    }
  }
}

static void write_all(int fd, const std::string &text) {
  size_t at = 0;
  while(at < text.size()) {
    const ssize_t n = write(fd, text.data() + at, text.size() - at);
    if(n <= 0) {
      return;
    }
    at += (size_t)n;
  }
}

static int serve_socket(const std::string &path_arg, int device) {
  signal(SIGPIPE, SIG_IGN); // a client that went away must not take the worker with it
  const std::string path = path_arg == "default" ? pm_serve::default_socket_path(true) : path_arg;
  if(path.empty()) {
    fprintf(stderr, "mugsy_profiles serve: no directory of this user's own for the default socket (XDG_RUNTIME_DIR, /tmp/paramugsy-<uid>)\n");
    return 2;
  }
  // one worker per path: the lock is taken before anything is probed or unlinked, and kept
  const std::string lock_path = path + ".lock";
  const int lock_fd = open(lock_path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0600);
  if(lock_fd < 0 || flock(lock_fd, LOCK_EX | LOCK_NB) != 0) {
    fprintf(stderr, "mugsy_profiles serve: %s: %s\n", lock_path.c_str(),
            lock_fd < 0 ? strerror(errno) : "another worker holds the lock (it is starting or listening)");
    return 2;
  }
  const int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  sockaddr_un addr;
  memset(&addr, 0, sizeof addr);
  addr.sun_family = AF_UNIX;
  if(ls < 0 || path.size() >= sizeof addr.sun_path) {
    fprintf(stderr, "mugsy_profiles serve: cannot make a socket at %s\n", path.c_str());
    return 2;
  }
  memcpy(addr.sun_path, path.c_str(), path.size() + 1);
  // a socket file left behind by a worker that is gone is replaced; one that answers is somebody else's worker
  {
    const int probe = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if(probe >= 0 && connect(probe, (sockaddr *)&addr, sizeof addr) == 0) {
      close(probe);
      fprintf(stderr, "mugsy_profiles serve: a worker is already listening at %s\n", path.c_str());
      return 2;
    }
    if(probe >= 0) {
      close(probe);
    }
    unlink(path.c_str());
  }
  const mode_t old = umask(0077); // the socket is its owner's alone
  const int rc_bind = bind(ls, (sockaddr *)&addr, sizeof addr);
  umask(old);
  if(rc_bind != 0 || listen(ls, 64) != 0) {
    fprintf(stderr, "mugsy_profiles serve: cannot listen at %s: %s\n", path.c_str(), strerror(errno));
    return 2;
  }
  (void)pm_device_count(); // bring the runtime up before the first request
  const double request_timeout = pm_serve::env_seconds("PARAMUGSY_SERVE_REQUEST_TIMEOUT", 5.0);
  long jobs = 0;
  bool running = true;
  while(running) {
    const int fd = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
    if(fd < 0) {
      if(errno == EINTR) {
        continue;
      }
      break;
    }
    if(!pm_serve::peer_is_me(fd)) { // not this user's process: not a word
      close(fd);
      continue;
    }
    pm_serve::set_timeouts(fd, request_timeout, request_timeout);
    std::string line;
    if(read_line(fd, line)) {
      std::vector<std::string> tok;
      size_t at = 0;
      while(at <= line.size()) {
        size_t e = line.find('\t', at);
        if(e == std::string::npos) {
          e = line.size();
        }
        tok.push_back(line.substr(at, e - at));
        at = e + 1;
      }
      if(tok[0] == "quit") {
        write_all(fd, "done 0\n");
        running = false;
      }
      else if(tok[0] == "stats") {
        write_all(fd, "done 0\njobs " + std::to_string(jobs) + "\n");
      }
      else if(tok[0] == "translate" && tok.size() >= 7 && (size_t)atol(tok[6].c_str()) + 7 == tok.size()) {
        const std::string &cwd = tok[1];
        std::vector<std::string> paths;
        for(size_t k = 7; k < tok.size(); ++k) {
          paths.push_back(client_path(cwd, tok[k]));
        }
        std::vector<const char *> cpaths;
        for(size_t k = 0; k < paths.size(); ++k) {
          cpaths.push_back(paths[k].c_str());
        }
        std::vector<int> devs = tok[5] == "-" ? std::vector<int>() : parse_devices(tok[5]);
        if(devs.empty()) {
          devs.push_back(device);
        }
        const int rc = pm_translate_files_as(client_path(cwd, tok[2]).c_str(), client_path(cwd, tok[3]).c_str(), cpaths.data(), (int)cpaths.size(),
                                             client_path(cwd, tok[4]).c_str(), tok[2].c_str(), tok[3].c_str(), devs.data(), (int)devs.size());
        ++jobs;
        char head[64];
        snprintf(head, sizeof head, "done %d\n", rc);
        write_all(fd, std::string(head) + (rc ? std::string(pm_last_error()) + "\n" : std::string()));
      }
      else {
        write_all(fd, "done -1\nmugsy_profiles serve: bad request\n");
      }
    }
    close(fd);
  }
  close(ls);
  unlink(path.c_str());
  unlink(lock_path.c_str());
  close(lock_fd);
  return 0;
}

int main(int argc, char **argv) {
  if(argc < 2) {
    fprintf(stderr, "usage: mugsy_profiles {make|translate|untranslate|stage|align|serve} <flags>\n");
    return 1;
  }
  std::string cmd = argv[1];
  const char *dev_env = getenv("PARAMUGSY_DEVICE");
  int device = dev_env ? atoi(dev_env) : 0;
  {
    // the executable's switches: the library reads no environment variable, it is handed a struct (pm_translate_options_t)
    pm_translate_options_t opt;
    memset(&opt, 0, sizeof opt);
    const char *e;
    opt.coordinate_bits = (e = getenv("PM_TRANSLATE_WIDE")) && e[0] == '1' ? 64 : 0;
    opt.library_scans = (e = getenv("PM_TRANSLATE_LIBRARY_SCANS")) && e[0] == '1';
    opt.no_side_file = getenv("PM_NO_SOA") != nullptr;
    opt.timing = getenv("PM_TIMING") != nullptr;
    if(opt.coordinate_bits || opt.library_scans || opt.no_side_file || opt.timing) {
      pm_translate_set_default_options(&opt);
    }
  }
  int code;
  if(cmd == "serve" && argc >= 4 && std::string(argv[2]) == "-socket") {
    code = serve_socket(argv[3], device);
  }
  else if(cmd == "serve") {
    // new in this build: a resident worker.  The orchestrator starts one short process per tree node
    // (lib/base/job_processor.ml:184-211) and each pays the HIP runtime's start-up; a worker pays it once.  Protocol: one
    // command per line on stdin, TAB-separated (`stage<TAB>-left_maf<TAB>path<TAB>...`), answered by `done <exit code>` on stdout;
    // `quit` or end of input ends the worker.
    (void)pm_device_count(); // bring the runtime up before the first command
    std::string line;
    code = 0;
    while(std::getline(std::cin, line)) {
      std::vector<std::string> tok;
      size_t at = 0;
      while(at <= line.size()) {
        size_t e = line.find('\t', at);
        if(e == std::string::npos) {
          e = line.size();
        }
        tok.push_back(line.substr(at, e - at));
        at = e + 1;
      }
      if(tok.empty() || tok[0].empty()) {
        continue;
      }
      if(tok[0] == "quit") {
        break;
      }
      std::map<std::string, std::string> flag;
      for(size_t k = 1; k + 1 < tok.size(); k += 2) {
        flag[tok[k]] = tok[k + 1];
      }
      int c = run_command(tok[0], flag, device);
      code = c ? c : code;
      printf("done %d\n", c);
      fflush(stdout);
    }
  }
  else {
    std::map<std::string, std::string> flag;
    for(int k = 2; k + 1 < argc; k += 2) {
      flag[argv[k]] = argv[k + 1];
    }
    code = run_command(cmd, flag, device);
  }
  if(code) {
    return code;
  }
  // done: everything this process wrote is flushed below; leave without tearing the HIP runtime down (tens of
  // milliseconds that a short-lived tool has no use for) -- unless a tool library rides along in this process
  // (rocprofv3 and friends write their results from exit handlers)
  fflush(stdout);
  fflush(stderr);
  if(!getenv("LD_PRELOAD") && !getenv("ROCP_TOOL_LIBRARIES") && !getenv("HSA_TOOLS_LIB")) {
    _exit(0);
  }
  return 0;
}
