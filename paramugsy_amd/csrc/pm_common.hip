// pm_common.hip -- error slot and device queries of libparamugsy_amd.so.
#include "pm_internal.hpp"

#include <cstring>
#include <mutex>
#include <vector>

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

// ---- the process's default translate options (pm_translate_set_default_options)
namespace {
std::mutex g_translate_defaults_lock;
pm_translate_options_t g_translate_defaults = {};
} // namespace

pm_translate_options_t translate_options(const pm_translate_options_t *given) {
  if(given) {
    return *given;
  }
  std::lock_guard<std::mutex> hold(g_translate_defaults_lock);
  return g_translate_defaults;
}

// ---- DevPool (pm_internal.hpp)
namespace {
struct PoolEntry {
  int device;
  void *p;
  size_t bytes;
};
std::mutex g_pool_lock;
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
const size_t POOL_MAX_ONE = (size_t)1 << 30, POOL_MAX_ALL = (size_t)4 << 30;
} // namespace

void *DevPool::take(int device, size_t bytes, size_t *got) {
  std::lock_guard<std::mutex> hold(g_pool_lock);
  size_t best = g_pool.size();
  for(size_t k = 0; k < g_pool.size(); ++k) { // the smallest kept buffer that is big enough (and not wastefully so)
    if(g_pool[k].device == device && g_pool[k].bytes >= bytes && g_pool[k].bytes <= 2 * bytes + (1 << 20) &&
       (best == g_pool.size() || g_pool[k].bytes < g_pool[best].bytes)) {
      best = k;
    }
  }
  if(best == g_pool.size()) {
    return nullptr;
  }
  void *p = g_pool[best].p;
  *got = g_pool[best].bytes;
  g_pool_bytes -= g_pool[best].bytes;
  g_pool.erase(g_pool.begin() + (long)best);
  return p;
}

void DevPool::give(int device, void *p, size_t bytes) {
  {
    std::lock_guard<std::mutex> hold(g_pool_lock);
    if(bytes <= POOL_MAX_ONE && g_pool_bytes + bytes <= POOL_MAX_ALL) {
      g_pool.push_back(PoolEntry{device, p, bytes});
      g_pool_bytes += bytes;
      return;
    }
  }
  int cur = -1;
  (void)hipGetDevice(&cur);
  if(cur != device) {
    (void)hipSetDevice(device);
  }
  (void)hipFree(p);
  if(cur != device && cur >= 0) {
    (void)hipSetDevice(cur);
  }
}

void DevPool::trim() {
  std::vector<PoolEntry> all;
  {
    std::lock_guard<std::mutex> hold(g_pool_lock);
    all.swap(g_pool);
    g_pool_bytes = 0;
  }
  int cur = -1;
  (void)hipGetDevice(&cur);
  for(size_t k = 0; k < all.size(); ++k) {
    (void)hipSetDevice(all[k].device);
    (void)hipFree(all[k].p);
  }
  if(cur >= 0) {
    (void)hipSetDevice(cur);
  }
}

hipError_t malloc_trimming(void **p, size_t n) {
  hipError_t e = hipMalloc(p, n);
  if(e == hipErrorOutOfMemory) {
    (void)hipGetLastError(); // the failed allocation's sticky error
    // only what nobody is using at the moment is kept in these caches, so giving it back cannot pull a buffer from under a call
    dp_batch_cache_trim();
    text_staging_trim();
    DevPool::trim();
    e = hipMalloc(p, n);
  }
  return e;
}

} // namespace pm

extern "C" {

int pm_release_caches(void) {
  pm::dp_batch_cache_trim();
  pm::text_staging_trim();
  pm::DevPool::trim();
  return PM_OK;
}


const char *pm_last_error(void) { return pm::g_last_error.c_str(); }

int pm_translate_set_default_options(const pm_translate_options_t *options) {
  if(options && options->coordinate_bits != 0 && options->coordinate_bits != 32 && options->coordinate_bits != 64) {
    return pm::fail(PM_E_INVALID, "pm_translate_set_default_options: coordinate_bits is 0, 32 or 64");
  }
  std::lock_guard<std::mutex> hold(pm::g_translate_defaults_lock);
  pm::g_translate_defaults = options ? *options : pm_translate_options_t{};
  return PM_OK;
}

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
