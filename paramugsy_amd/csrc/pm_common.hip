// pm_common.hip -- error slot and device queries of libparamugsy_amd.so.
#include "pm_internal.hpp"

#include <cstring>

namespace pm {

static thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess || n <= 0) {
    return fail(PM_E_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                                    " (libparamugsy_amd has no CPU path)");
  }
  if(device < 0 || device >= n) {
    return fail(PM_E_INVALID, "device index out of range");
  }
  e = hipSetDevice(device);
  if(e != hipSuccess) {
    return fail(PM_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  return PM_OK;
}

} // namespace pm

extern "C" {

const char *pm_last_error(void) { return pm::g_last_error.c_str(); }

int pm_device_count(void) {
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess) {
    return 0;
  }
  return n;
}

int pm_device_info(int dev, char *name, int cap, int *compute_units, int64_t *hbm_bytes) {
  int rc = pm::use_device(dev);
  if(rc) {
    return rc;
  }
  hipDeviceProp_t prop;
  PM_HIP(hipGetDeviceProperties(&prop, dev));
  if(name && cap > 0) {
    std::strncpy(name, prop.gcnArchName, (size_t)cap - 1);
    name[cap - 1] = 0;
  }
  if(compute_units) {
    *compute_units = prop.multiProcessorCount;
  }
  if(hbm_bytes) {
    *hbm_bytes = (int64_t)prop.totalGlobalMem;
  }
  return PM_OK;
}

} // extern "C"
