// side_tools_main.cc -- drop-in executables for the two lib/profiles_cpp tools that compile upstream:
//   m_sort_delta  < in.delta > out.delta      (lib/profiles_cpp/m_sort_delta.cc:73-91)
//   maf_analyzer  <maf>       > report        (lib/profiles_cpp/maf_analyzer.cc:12-38)
// Same command lines and output bytes; the work runs on the GPU through the C ABI.  PARAMUGSY_DEVICE selects the device.
#include <cstdio>
#include <unistd.h>
#include <cstdlib>

#include "../../include/paramugsy_amd.h"

int main(int argc, char **argv) {
  const char *dev_env = getenv("PARAMUGSY_DEVICE");
  int device = dev_env ? atoi(dev_env) : 0;
#if defined(PM_TOOL_SORT_DELTA)
  (void)argc;
  (void)argv;
  int rc = pm_sort_delta(nullptr, nullptr, device);
#elif defined(PM_TOOL_MAF_ANALYZER)
  if(argc < 2) {
    fprintf(stderr, "Usage: maf_analyzer <maf>\n"); // upstream dereferences argv[1] unchecked
    return 1;
  }
  int rc = pm_maf_analyzer(argv[1], nullptr, device);
#else
#error "define PM_TOOL_SORT_DELTA or PM_TOOL_MAF_ANALYZER"
#endif
  if(rc != PM_OK) {
    fprintf(stderr, "error %d: %s\n", rc, pm_last_error());
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
