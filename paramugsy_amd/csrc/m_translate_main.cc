// m_translate_main.cc -- the drop-in executable.  Same argv, same exit behaviour and same output bytes as
// the reference's lib/m_translate/m_translate_main.cc:19-46; the work runs on the GPU through the C ABI.
//
// A fresh process pays 0.05-0.19 s for a HIP runtime of its own before the 0.05 s job (round 3: the drop-in lost to the reference
// run as sixteen processes).  So this executable links NOTHING of HIP: it first asks a resident worker -- `mugsy_profiles serve -socket
// <path>`, one per GPU, started once per node by whoever owns the node -- over a UNIX socket (PARAMUGSY_SERVE_SOCKET; default
// $XDG_RUNTIME_DIR/paramugsy/serve.sock or /tmp/paramugsy-<uid>/serve.sock, a directory that is the caller's alone:
// serve_common.hpp) and prints what the worker says; only when nobody listens does it load libparamugsy_amd.so
// (dlopen, from ../paramugsy_amd beside this file) and run the job in this process, as before.  The orchestrator's task script
// (lib/base/mugsy_profiles_task.ml:53-58) needs no change either way.
// Whom it believes: only a peer of its own uid (SO_PEERCRED); anybody else on that socket is treated as nobody listening.  How long
// it waits: PARAMUGSY_SERVE_TIMEOUT seconds (default 600) for the verdict -- once the request is out the job is the worker's, so a
// worker that hangs or goes away ends this process with the reference's failure exit, never with a second run of the job; and a
// verdict of success is believed only when the output file is there.
// Optional: PARAMUGSY_DEVICE=<n> selects the HIP device (default 0; with a worker: the worker's); PARAMUGSY_DEVICES=0,1,... spreads the
// delta-file list over several devices of the node (pm_translate_files_multi; the argv stays the reference's).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include "serve_common.hpp"

typedef int (*translate_as_fn)(const char *, const char *, const char *const *, int, const char *, const char *, const char *, const int *, int);
typedef const char *(*last_error_fn)(void);
typedef int (*set_defaults_fn)(const void *);
// pm_translate_options_t (include/paramugsy_amd.h), spelled out because this file includes nothing of the library
struct translate_options {
  int coordinate_bits, library_scans, no_side_file, timing, reserved[4];
};

static bool plain(const std::string &s) { return s.find('\t') == std::string::npos && s.find('\n') == std::string::npos; }

// The job through a resident worker: 1 when it ran there (rc and message set), 0 when nobody listens (or the request cannot be put
// on one line): then it runs here.
static int ask_worker(char **argv, const std::vector<std::string> &paths, const char *devices, int &rc, std::string &message) {
  std::string sock;
  if(const char *e = getenv("PARAMUGSY_SERVE_SOCKET")) {
    sock = e;
  }
  else {
    sock = pm_serve::default_socket_path(false); // "" when no directory of the caller's own holds one
  }
  if(sock.empty() || sock == "none") {
    return 0;
  }
  sockaddr_un addr;
  memset(&addr, 0, sizeof addr);
  addr.sun_family = AF_UNIX;
  if(sock.size() >= sizeof addr.sun_path) {
    return 0;
  }
  memcpy(addr.sun_path, sock.c_str(), sock.size() + 1);
  char cwd[4096];
  if(!getcwd(cwd, sizeof cwd)) {
    return 0;
  }
  std::string req = std::string("translate\t") + cwd + "\t" + argv[1] + "\t" + argv[2] + "\t" + argv[4] + "\t" + (devices && *devices ? devices : "-") +
                    "\t" + std::to_string(paths.size());
  if(!plain(cwd) || !plain(argv[1]) || !plain(argv[2]) || !plain(argv[4])) {
    return 0;
  }
  for(size_t k = 0; k < paths.size(); ++k) {
    if(!plain(paths[k])) {
      return 0;
    }
    req += "\t" + paths[k];
  }
  req += "\n";
  const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  if(fd < 0) {
    return 0;
  }
  if(connect(fd, (sockaddr *)&addr, sizeof addr) != 0) {
    close(fd);
    return 0;
  }
  if(!pm_serve::peer_is_me(fd)) { // somebody else's process answers at that path: not a worker of ours, and told nothing
    close(fd);
    fprintf(stderr, "m_translate: the process listening at %s is not this user's; running the job here\n", sock.c_str());
    return 0;
  }
  pm_serve::set_timeouts(fd, pm_serve::env_seconds("PARAMUGSY_SERVE_TIMEOUT", 600.0), 10.0);
  size_t at = 0;
  while(at < req.size()) {
    const ssize_t n = send(fd, req.data() + at, req.size() - at, MSG_NOSIGNAL);
    if(n <= 0) {
      close(fd);
      if(at == 0) {
        return 0; // not a byte is out: nothing can have run
      }
      rc = -1; // a worker that has part of a request may not be raced by a second run of the job
      message = "the resident worker stopped taking the request";
      return 1;
    }
    at += (size_t)n;
  }
  std::string reply;
  char buf[4096];
  bool timed_out = false;
  for(;;) {
    const ssize_t n = read(fd, buf, sizeof buf);
    if(n < 0 && errno == EINTR) {
      continue;
    }
    if(n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
      timed_out = true;
    }
    if(n <= 0) {
      break;
    }
    reply.append(buf, (size_t)n);
  }
  close(fd);
  if(reply.compare(0, 5, "done ") != 0) { // the worker died on the job: what it wrote is on the stream, as the reference's would be
    rc = -1;
    message = timed_out ? "no verdict from the resident worker within PARAMUGSY_SERVE_TIMEOUT" : "the resident worker went away";
    return 1;
  }
  rc = atoi(reply.c_str() + 5);
  if(rc == 0) { // success is the output file, not the word
    struct stat st;
    if(stat(argv[4], &st) != 0) {
      rc = -1;
      message = std::string("the resident worker reported success but ") + argv[4] + " does not exist";
      return 1;
    }
  }
  const size_t nl = reply.find('\n');
  message = nl == std::string::npos ? "" : reply.substr(nl + 1);
  while(!message.empty() && message[message.size() - 1] == '\n') {
    message.erase(message.size() - 1);
  }
  return 1;
}

int main(int argc, char **argv) {
  if(argc < 5) {
    fprintf(stderr, "Usage: m_translate <left_profile_dir> <right_profile_dir> <nucmer_file_list> <output_delta_path>\n");
    return 1;
  }
  std::vector<std::string> paths;
  {
    std::ifstream list(argv[3]);
    std::string line;
    while(std::getline(list, line)) {
      paths.push_back(line);
    }
  }
  const char *devices_env = getenv("PARAMUGSY_DEVICES");
  int rc = 0;
  std::string message;
  if(!ask_worker(argv, paths, devices_env, rc, message)) {
    // nobody listens: the library, and the job in this process
    std::string lib;
    {
      char exe[4096];
      const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
      if(n > 0) {
        exe[n] = 0;
        lib = exe;
        lib = lib.substr(0, lib.rfind('/')); // bin
        lib = lib.substr(0, lib.rfind('/')) + "/paramugsy_amd/libparamugsy_amd.so";
      }
    }
    void *h = lib.empty() ? nullptr : dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if(!h) {
      h = dlopen("libparamugsy_amd.so", RTLD_NOW | RTLD_GLOBAL);
    }
    translate_as_fn run = h ? (translate_as_fn)dlsym(h, "pm_translate_files_as") : nullptr;
    last_error_fn last_error = h ? (last_error_fn)dlsym(h, "pm_last_error") : nullptr;
    if(!run || !last_error) {
      fprintf(stderr, "m_translate: cannot load libparamugsy_amd.so (%s): %s\n", lib.c_str(), dlerror());
      return 134;
    }
    {
      // this executable's switches (the library itself reads no environment variable): handed over as the process's defaults
      translate_options opt;
      memset(&opt, 0, sizeof opt);
      const char *e;
      opt.coordinate_bits = (e = getenv("PM_TRANSLATE_WIDE")) && e[0] == '1' ? 64 : 0;
      opt.library_scans = (e = getenv("PM_TRANSLATE_LIBRARY_SCANS")) && e[0] == '1';
      opt.no_side_file = getenv("PM_NO_SOA") != nullptr;
      opt.timing = getenv("PM_TIMING") != nullptr;
      set_defaults_fn set_defaults = (set_defaults_fn)dlsym(h, "pm_translate_set_default_options");
      if(set_defaults && (opt.coordinate_bits || opt.library_scans || opt.no_side_file || opt.timing)) {
        set_defaults(&opt);
      }
    }
    std::vector<const char *> cpaths;
    for(size_t k = 0; k < paths.size(); ++k) {
      cpaths.push_back(paths[k].c_str());
    }
    const char *dev_env = getenv("PARAMUGSY_DEVICE");
    std::vector<int> devs;
    if(devices_env) {
      for(const char *p = devices_env; *p;) {
        char *end = nullptr;
        long v = strtol(p, &end, 10);
        if(end == p) {
          break;
        }
        devs.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
      }
    }
    if(devs.empty()) {
      devs.push_back(dev_env ? atoi(dev_env) : 0);
    }
    rc = run(argv[1], argv[2], cpaths.data(), (int)cpaths.size(), argv[4], argv[1], argv[2], devs.data(), (int)devs.size());
    if(rc) {
      message = last_error();
    }
  }
  if(rc != 0) {
    fprintf(stderr, "m_translate: error %d: %s\n", rc, message.c_str());
    // the reference ends in SIGABRT (uncaught exception / assert) on every failure past argument checking
    return 134;
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
