// pm_internal.hpp -- shared host-side plumbing of libparamugsy_amd.so (error slot, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <utility>

#include "../../include/paramugsy_amd.h"

namespace pm {

// Records the message returned by pm_last_error() and hands `code` back.
int fail(int code, const std::string &msg);
// hipSetDevice with a loud failure when there is no usable device (no CPU fallback exists).
int use_device(int device);

#define PM_HIP(call)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if(e_ != hipSuccess) {                                                                                 \
      return ::pm::fail(PM_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                      \
    }                                                                                                      \
  } while(0)

#define PM_TRY(call)  \
  do {                \
    int rc_ = (call); \
    if(rc_) {         \
      return rc_;     \
    }                 \
  } while(0)

// An extern "C" entry point's body run so that no C++ exception (std::bad_alloc from a table sized by the input, a std::system_error
// from a thread that cannot be started) crosses the C boundary: it becomes a return code and a message.
template <typename F>
int guarded(const char *who, F body) {
  try {
    return body();
  }
  catch(const std::bad_alloc &) {
    return fail(PM_E_INVALID, std::string(who) + ": out of host memory");
  }
  catch(const std::exception &e) {
    return fail(PM_E_INVALID, std::string(who) + ": " + e.what());
  }
}

// A helper thread that is joined when it goes out of scope: an exception (or an early return) that unwinds past a joinable
// std::thread ends the process in std::terminate instead of reaching guarded() above.  The body runs inside a catch-all: what it
// throws is kept and thrown again by join_and_rethrow() in the thread that started it (plain join() and the destructor drop it).
class JoinThread {
public:
  JoinThread() = default;
  template <typename F>
  explicit JoinThread(F body) : thrown_(std::make_shared<std::exception_ptr>()) {
    std::shared_ptr<std::exception_ptr> slot = thrown_;
    t_ = std::thread([body, slot]() mutable {
      try {
        body();
      }
      catch(...) {
        *slot = std::current_exception();
      }
    });
  }
  JoinThread(JoinThread &&) = default;
  JoinThread &operator=(JoinThread &&o) {
    join();
    t_ = std::move(o.t_);
    thrown_ = std::move(o.thrown_);
    return *this;
  }
  JoinThread(const JoinThread &) = delete;
  JoinThread &operator=(const JoinThread &) = delete;
  ~JoinThread() { join(); }
  bool joinable() const { return t_.joinable(); }
  void join() {
    if(t_.joinable()) {
      t_.join();
    }
  }
  void join_and_rethrow() {
    join();
    if(thrown_ && *thrown_) {
      std::exception_ptr e = *thrown_;
      *thrown_ = nullptr;
      std::rethrow_exception(e);
    }
  }

private:
  std::thread t_;
  std::shared_ptr<std::exception_ptr> thrown_;
};

// hipMalloc that, when the device is out of memory, gives back what this library keeps from call to call (the buffer pool, the
// kept DP batches with their gigabytes of workspace, the pinned staging pieces) and tries once more: a resident process must not
// fail -- or quietly run a DP in more chunks -- while gigabytes sit unused in its own caches.
hipError_t malloc_trimming(void **p, size_t n);

// Owning device allocation.
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if(p) {
      (void)hipFree(p);
      p = nullptr;
      bytes = 0;
    }
  }
  int alloc(size_t n) {
    release();
    if(n == 0) {
      n = 16; // keep pointers non-null so kernels can take them unconditionally
    }
    hipError_t e = malloc_trimming(&p, n);
    if(e != hipSuccess) {
      p = nullptr;
      return fail(PM_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    bytes = n;
    return PM_OK;
  }
  int upload(const void *src, size_t n, hipStream_t stream) {
    int rc = alloc(n);
    if(rc) {
      return rc;
    }
    if(n > 0) {
      // blocking copy: the caller's (pageable) host buffer may be reused as soon as this returns
      (void)stream;
      hipError_t e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
      if(e != hipSuccess) {
        return fail(PM_E_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
      }
    }
    return PM_OK;
  }
};

// Device buffers kept from call to call (per device, at most 4 GiB in all, none above 1 GiB): a resident caller's second
// file-level call finds its hundred-megabyte scratch buffers allocated -- hipMalloc + hipFree of them cost more than the
// kernels that use them.  PooledBuf behaves like DevBuf; what does not fit the pool's limits is simply freed.
class DevPool {
public:
  static void *take(int device, size_t bytes, size_t *got);
  static void give(int device, void *p, size_t bytes);
  static void trim(); // frees everything kept
};

struct PooledBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int device = 0;
  PooledBuf() = default;
  PooledBuf(const PooledBuf &) = delete;
  PooledBuf &operator=(const PooledBuf &) = delete;
  ~PooledBuf() { release(); }
  void release() {
    if(p) {
      DevPool::give(device, p, bytes);
      p = nullptr;
      bytes = 0;
    }
  }
  int alloc(size_t n, int dev) {
    release();
    if(n == 0) {
      n = 16;
    }
    device = dev;
    p = DevPool::take(dev, n, &bytes);
    if(!p) {
      hipError_t e = malloc_trimming(&p, n);
      if(e != hipSuccess) {
        p = nullptr;
        return fail(PM_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
      }
      bytes = n;
    }
    return PM_OK;
  }
};

// what pm_release_caches() frees besides the pool: the pinned staging pieces (translate_host.cc), the kept DP batches (dp_maf.hip)
void text_staging_trim();
void dp_batch_cache_trim();

// The options a translate job runs under (pm_translate_options_t): the caller's, or -- a null pointer -- a copy of the process's
// defaults taken under their lock.  Resolved once at the entry point and handed down by value: nothing below reads the environment
// or a global, so the workers of pm_translate_files_multi (one thread per device) all see what their job was started with.
pm_translate_options_t translate_options(const pm_translate_options_t *given);
inline bool timing_on() { return translate_options(nullptr).timing != 0; } // the file-level entries outside the translate path

// What the device needs to list a job's work units itself (translate_job.hip): per side (0 left, 1 right) the rows of every sequence
// sorted by forward start, and per delta entry the index of its sequence on that side (-1: the side has no such sequence).
struct EnumInput {
  int64_t n_seq[2];
  const int64_t *seq_off[2]; // [n_seq + 1]
  const int32_t *seq_rows[2];
  const int32_t *entry_seq[2]; // [entries]
};
int job_create_enumerating(const pm_rows_t *left, const pm_rows_t *right, const pm_deltas_t *deltas, const EnumInput *en,
                           const pm_translate_options_t &opt, int device, pm_job_t **out);
int job_unit_at(pm_job_t *job, int64_t unit, int32_t out[3]); // a unit's delta entry, left row, right row

} // namespace pm