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
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include <sys/stat.h>

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

int main(int argc, char **argv) {
  if(argc < 2) {
    fprintf(stderr, "usage: mugsy_profiles {make|translate|untranslate} <flags>\n");
    return 1;
  }
  std::string cmd = argv[1];
  std::map<std::string, std::string> flag;
  for(int k = 2; k + 1 < argc; k += 2) {
    flag[argv[k]] = argv[k + 1];
  }
  const char *dev_env = getenv("PARAMUGSY_DEVICE");
  int device = dev_env ? atoi(dev_env) : 0;
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
    std::vector<std::string> paths;
    std::ifstream list(flag["-nucmer_list"].c_str());
    std::string line;
    while(std::getline(list, line)) {
      paths.push_back(line);
    }
    std::vector<const char *> cpaths;
    for(size_t k = 0; k < paths.size(); ++k) {
      cpaths.push_back(paths[k].c_str());
    }
    rc = pm_translate_files(flag["-profiles_left"].c_str(), flag["-profiles_right"].c_str(), cpaths.data(), (int)cpaths.size(),
                            flag["-out_delta"].c_str(), device);
  }
  else if(cmd == "untranslate") {
    if(flag["-profile_paths_list"].empty() || flag["-in_maf"].empty() || flag["-out_maf"].empty()) {
      fprintf(stderr, "Must provide -profile_paths_list, -in_maf and -out_maf\n"); // m_untranslate.ml:183-188
      return 2;
    }
    std::vector<std::string> dirs;
    std::ifstream list(flag["-profile_paths_list"].c_str());
    std::string line;
    while(std::getline(list, line)) {
      dirs.push_back(line);
    }
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
