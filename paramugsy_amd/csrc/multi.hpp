// multi.hpp -- several devices of one node behind one C-ABI call: a static, contiguous partition of the job's independent items
// (pairs, blocks, delta files), one host thread and one HIP context per device, no collective; the outputs are gathered on
// the host in input order.  The reference's analogue is the chunked pair list (lib/base/pm_job.ml:43-57,83-91) run as
// `run_size` concurrent OS processes (lib/base/queued_task_server.ml:57-66).
#pragma once

#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "pm_internal.hpp"

namespace pm {

// Contiguous and balanced: the first n % parts parts get one item more (the rule of paramugsy_amd/shard.py::partition).
inline void partition(int64_t n, int parts, int k, int64_t &lo, int64_t &hi) {
  const int64_t base = n / parts, extra = n % parts;
  lo = k * base + (k < extra ? k : extra);
  hi = lo + base + (k < extra ? 1 : 0);
}

// The same for items of unequal cost (a pair's cells, La x Lb): contiguous slices cut where the running sum of the weights comes
// closest to k / parts of the total -- a slice of a ragged batch cut by COUNT is bounded by its longest pair and the slices differ
// in cells (round 3: north_star says "statically pair-partitioned", not "by count").  cuts: parts + 1 positions, cuts[0] = 0,
// cuts[parts] = n, never decreasing.  Items of one and the same weight (and a total of 0): exactly partition()'s slices.  Still
// static -- a function of the lengths alone -- and contiguous, so the ordered host-side gather is unchanged.
inline void partition_weighted(const int64_t *weight, int64_t n, int parts, std::vector<int64_t> &cuts) {
  cuts.assign((size_t)parts + 1, 0);
  bool uniform = true;
  __int128 total = 0;
  for(int64_t k = 0; k < n; ++k) {
    uniform = uniform && weight[k] == weight[0];
    total += weight[k] > 0 ? weight[k] : 0;
  }
  if(uniform || total <= 0) {
    for(int k = 0; k < parts; ++k) {
      int64_t lo, hi;
      partition(n, parts, k, lo, hi);
      cuts[(size_t)k] = lo;
      cuts[(size_t)k + 1] = hi;
    }
    return;
  }
  __int128 run = 0; // exact: (run + w) / total against k / parts, cross-multiplied
  int64_t at = 0;
  for(int k = 1; k < parts; ++k) {
    while(at < n) { // take item `at` while that brings the running sum no further from the target than it is
      const __int128 w = weight[at] > 0 ? weight[at] : 0;
      if((run + w) * parts - total * k > total * k - run * parts) {
        break;
      }
      run += w;
      ++at;
    }
    cuts[(size_t)k] = at;
  }
  cuts[(size_t)parts] = n;
}

// a pair's weight: its cells, plus its columns so that empty profiles still count for something
inline int64_t pair_weight(int64_t la, int64_t lb) { return la * lb + la + lb + 1; }

inline int check_devices(const int *devices, int n_devices, const char *who) {
  if(!devices || n_devices < 1) {
    return fail(PM_E_INVALID, std::string(who) + ": needs at least one device");
  }
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    return fail(PM_E_NO_DEVICE, "no HIP device (libparamugsy_amd has no CPU path)");
  }
  for(int k = 0; k < n_devices; ++k) {
    if(devices[k] < 0 || devices[k] >= n) {
      return fail(PM_E_INVALID, std::string(who) + ": device index out of range");
    }
  }
  return PM_OK;
}

// fn(worker, device) on one thread per entry of `devices` (the same device may be named more than once: its workers then share
// it).  Every worker's return code and message are kept; the call returns the code of the FIRST worker (in partition order) that
// failed and puts its message into this thread's error slot -- the error slot is per thread, so a worker's message would
// otherwise be lost with its thread.  rc_out (optional): every worker's code.
template <typename F>
int run_on_devices(const int *devices, int n_devices, F fn, std::vector<int> *rc_out = nullptr) {
  std::vector<int> rc((size_t)n_devices, PM_OK);
  std::vector<std::string> msg((size_t)n_devices);
  std::vector<JoinThread> th; // (joined also when starting a later worker throws)
  th.reserve((size_t)n_devices);
  for(int w = 0; w < n_devices; ++w) {
    th.emplace_back([&, w]() {
      int r = use_device(devices[w]);
      if(!r) {
        try {
          r = fn(w, devices[w]);
        }
        catch(const std::exception &e) {
          r = fail(PM_E_INVALID, std::string("worker failed: ") + e.what());
        }
        catch(...) {
          r = fail(PM_E_INVALID, "worker failed");
        }
      }
      rc[(size_t)w] = r;
      if(r) {
        msg[(size_t)w] = pm_last_error();
      }
    });
  }
  for(size_t k = 0; k < th.size(); ++k) {
    th[k].join();
  }
  if(rc_out) {
    *rc_out = rc;
  }
  for(int w = 0; w < n_devices; ++w) {
    if(rc[(size_t)w]) {
      return fail(rc[(size_t)w], "device " + std::to_string(devices[w]) + " (worker " + std::to_string(w) + "): " + msg[(size_t)w]);
    }
  }
  return PM_OK;
}

} // namespace pm
