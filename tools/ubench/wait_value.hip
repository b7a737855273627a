// GPU box: does hipStreamWaitValue32 hold a stream back until a kernel on ANOTHER stream has counted its workgroups in, and what
// does the wait cost?  hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_value tools/ubench/wait_value.hip && timeout -k 5 60 /tmp/wait_value
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
__global__ void count_in_and_spin(int *started, int iters, float *sink) {
  if(threadIdx.x == 0) {
    __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  float v = threadIdx.x;
  for(int i = 0; i < iters; ++i) {
    v = v * 1.0001f + 0.5f;
  }
  if(v == 12345.f) {
    *sink = v;
  }
}
__global__ void mark(long long *t) { *t = wall_clock64(); }
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if(!can) {
    return 0;
  }
  int *started = nullptr;
  float *sink = nullptr;
  long long *t = nullptr;
  CK(hipMalloc(&started, 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&t, 16));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for(int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(started, 0, 4));
    CK(hipDeviceSynchronize());
    const int grid = 20000; // more workgroups than the chip holds: the last ones start late
    hipLaunchKernelGGL(count_in_and_spin, dim3(grid), dim3(64), 0, a, started, 200000, sink);
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, t); // when kernel A has ENDED
    CK(hipStreamWaitValue32(b, started, grid, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, t + 1); // when every workgroup of A has STARTED
    CK(hipDeviceSynchronize());
    long long h[2];
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    printf("rep %d: the waiting stream went on %.3f ms before kernel A ended (100 MHz clock)\n", rep, (h[0] - h[1]) / 1e5);
  }
  printf("ok\n");
  return 0;
}
