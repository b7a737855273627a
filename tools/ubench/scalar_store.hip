// tools/ubench/scalar_store.hip -- does gfx950 execute scalar stores, and how fast?  Each wave writes `iters` x 64 bytes
// (4 x s_store_dwordx4) of SGPR data to its own region; verified on the host; reports GB/s with 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) sstore_kernel(unsigned *out, int iters) {
  // wave-uniform base pointer and data
  unsigned long long base = (unsigned long long)(out) + (unsigned long long)blockIdx.x * (unsigned long long)iters * 64ull;
  unsigned v0 = blockIdx.x * 4u + 1u;
  for(int it = 0; it < iters; ++it) {
    unsigned long long p = base + (unsigned long long)it * 64ull;
    unsigned a = v0 + it, b = a + 1, c = a + 2, d = a + 3;
    asm volatile(
        "s_mov_b32 s20, %2\n\ts_mov_b32 s21, %3\n\ts_mov_b32 s22, %4\n\ts_mov_b32 s23, %5\n\t"
        "s_store_dwordx4 s[20:23], %0, 0x0\n\t"
        "s_store_dwordx4 s[20:23], %0, 0x10\n\t"
        "s_store_dwordx4 s[20:23], %0, 0x20\n\t"
        "s_store_dwordx4 s[20:23], %0, 0x30\n\t"
        "s_waitcnt lgkmcnt(8)"
        :: "s"(p), "s"(0), "s"(a), "s"(b), "s"(c), "s"(d)
        : "s20", "s21", "s22", "s23", "memory");
  }
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void __launch_bounds__(64) vstore_kernel(unsigned *out, int iters) {
  unsigned *base = out + (size_t)blockIdx.x * iters * 16;
  for(int it = 0; it < iters; ++it) {
    if(threadIdx.x < 16) base[(size_t)it * 16 + threadIdx.x] = blockIdx.x * 4u + 1u + it + (threadIdx.x & 3);
  }
}
int main() {
  const int blocks = 256 * 32, iters = 1000;
  size_t words = (size_t)blocks * iters * 16;
  unsigned *d; (void)hipMalloc(&d, words * 4); (void)hipMemset(d, 0, words * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for(int variant = 0; variant < 2; ++variant) {
    (void)hipMemset(d, 0, words * 4);
    (void)hipEventRecord(e0);
    if(variant == 0) sstore_kernel<<<blocks, 64>>>(d, iters); else vstore_kernel<<<blocks, 64>>>(d, iters);
    (void)hipEventRecord(e1);
    hipError_t err = hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(words);
    (void)hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for(size_t b = 0; b < (size_t)blocks && bad < 5; ++b)
      for(int it = 0; it < iters; ++it)
        for(int k = 0; k < 16; ++k)
          if(h[(b * iters + it) * 16 + k] != (unsigned)(b * 4u + 1u + it + (k & 3))) { ++bad; break; }
    printf("%s: err=%d %.3f ms  %.1f GB/s  mismatching 64-B records: %zu\n", variant == 0 ? "s_store_dwordx4" : "global_store (16 lanes)", (int)err, ms,
           words * 4 / (ms * 1e-3) / 1e9, bad);
  }
  return 0;
}
