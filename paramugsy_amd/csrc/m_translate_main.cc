// m_translate_main.cc -- the drop-in executable.  Same argv, same exit behaviour and same output bytes as
// the reference's lib/m_translate/m_translate_main.cc:19-46; the work runs on the GPU through the C ABI.
// Optional: PARAMUGSY_DEVICE=<n> selects the HIP device (default 0); PARAMUGSY_DEVICES=0,1,... spreads the delta-file list over
// several devices of the node (pm_translate_files_multi; the argv stays the reference's).
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/paramugsy_amd.h"

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
  std::vector<const char *> cpaths;
  for(size_t k = 0; k < paths.size(); ++k) {
    cpaths.push_back(paths[k].c_str());
  }
  const char *dev_env = getenv("PARAMUGSY_DEVICE");
  int device = dev_env ? atoi(dev_env) : 0;
  std::vector<int> devs;
  if(const char *list = getenv("PARAMUGSY_DEVICES")) {
    for(const char *p = list; *p;) {
      char *end = nullptr;
      long v = strtol(p, &end, 10);
      if(end == p) {
        break;
      }
      devs.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
    }
  }
  int rc = devs.size() > 1 ? pm_translate_files_multi(argv[1], argv[2], cpaths.data(), (int)cpaths.size(), argv[4], devs.data(), (int)devs.size())
                           : pm_translate_files(argv[1], argv[2], cpaths.data(), (int)cpaths.size(), argv[4], devs.size() == 1 ? devs[0] : device);
  if(rc != PM_OK) {
    fprintf(stderr, "m_translate: error %d: %s\n", rc, pm_last_error());
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
