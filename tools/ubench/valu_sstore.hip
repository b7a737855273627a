// tools/ubench/valu_sstore.hip -- can a VALU-bound wave also push lane masks out through scalar stores?
// Each wave runs `iters` steps of 224 dependent-ish VALU ops (the DP step's size), one LDS read with a full
// lgkmcnt wait (as the DP step has), and NST s_store_dwordx4 of v_cmp results (16 bytes each).  5 waves per SIMD.
// Reports the time per variant: NST = 0, 8 (128 B/step), 16 (256 B/step).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NST>
__global__ void __launch_bounds__(64) k(unsigned *out, int iters, int *sink) {
  __shared__ int lds[64];
  lds[threadIdx.x] = threadIdx.x;
  unsigned long long base = (unsigned long long)out + (unsigned long long)blockIdx.x * (unsigned long long)iters * 256ull;
  int a = threadIdx.x, b = blockIdx.x, c = 3, d = 7;
  for(int it = 0; it < iters; ++it) {
    int l = lds[(threadIdx.x + it) & 63];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a += l;
#pragma unroll
    for(int g = 0; g < 16; ++g) {
      // 14 VALU ops per group; one group = one "cell"
      asm volatile("v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      if(NST > 0 && (g % (16 / (NST > 0 ? NST : 1))) == 0) {
        unsigned long long p = base + (unsigned long long)it * 256ull + (unsigned long long)(g * 16);
        asm volatile("v_cmp_gt_i32 s[20:21], %1, %2\n\tv_cmp_gt_i32 s[22:23], %2, %1\n\t"
                     "s_store_dwordx4 s[20:23], %0, 0x0"
                     :: "s"(p), "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "memory");
      }
      else {
        asm volatile("v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(a), "+v"(b));
      }
    }
  }
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if(a + b + c + d == 0x12345678) sink[0] = a;
}
// as k<16> but the same 256 B per step leave as 32 s_store_dwordx2 whose data operands the compiler allocates
__global__ void __launch_bounds__(64) k2(unsigned *out, int iters, int *sink) {
  __shared__ int lds[64];
  lds[threadIdx.x] = threadIdx.x;
  unsigned long long base = (unsigned long long)out + (unsigned long long)blockIdx.x * (unsigned long long)iters * 256ull;
  int a = threadIdx.x, b = blockIdx.x, c = 3, d = 7;
  for(int it = 0; it < iters; ++it) {
    int l = lds[(threadIdx.x + it) & 63];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a += l;
    unsigned long long p = base + (unsigned long long)it * 256ull;
#pragma unroll
    for(int g = 0; g < 16; ++g) {
      unsigned long long m0, m1;
      asm volatile("v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_max_i32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_max_i32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_cmp_gt_i32 %4, %0, %1\n\tv_cmp_gt_i32 %5, %1, %0\n\t"
                   "s_store_dwordx2 %4, %6, %7\n\ts_store_dwordx2 %5, %6, %8"
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1)
                   : "s"(p), "i"(g * 8), "i"(128 + g * 8)
                   : "memory");
    }
  }
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if(a + b + c + d == 0x12345678) sink[0] = a;
}
template <int NST> void run(unsigned *d, int *sink, int blocks, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NST><<<blocks, 64>>>(d, 10, sink);
  (void)hipEventRecord(e0);
  k<NST><<<blocks, 64>>>(d, iters, sink);
  (void)hipEventRecord(e1);
  hipError_t err = hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  double valu = (double)blocks * iters * (16.0 * 14 + 1) * 64;
  printf("stores/step=%2d (%3d B): err=%d %.3f ms  %.2f T lane-ops/s  scalar-store %.1f GB/s\n", NST, NST * 16, (int)err, ms, valu / (ms * 1e-3) / 1e12,
         (double)blocks * iters * NST * 16 / (ms * 1e-3) / 1e9);
}
int main() {
  const int blocks = 256 * 4 * 5, iters = 2000;
  unsigned *d; (void)hipMalloc(&d, (size_t)blocks * iters * 256); int *sink; (void)hipMalloc(&sink, 4);
  run<0>(d, sink, blocks, iters);
  run<8>(d, sink, blocks, iters);
  run<16>(d, sink, blocks, iters);
  run<4>(d, sink, blocks, iters);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k2<<<blocks, 64>>>(d, 10, sink);
    (void)hipEventRecord(e0);
    k2<<<blocks, 64>>>(d, iters, sink);
    (void)hipEventRecord(e1);
    hipError_t err = hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("32 x s_store_dwordx2 per step (256 B): err=%d %.3f ms  %.2f T lane-ops/s  scalar-store %.1f GB/s\n", (int)err, ms,
           (double)blocks * iters * (16.0 * 14 + 1) * 64 / (ms * 1e-3) / 1e12, (double)blocks * iters * 256 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
