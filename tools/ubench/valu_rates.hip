// tools/ubench/valu_rates.hip -- measured issue rates of the VALU instructions the DP kernel is made of.
// Each kernel runs N_ITER x 64 independent-ish instructions of one kind per wave, 8 waves per SIMD on every CU;
// reports wave-instructions per cycle per SIMD (clock from s_memtime vs wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short2_t __attribute__((ext_vector_type(2)));
#define N_ITER 2000
template <int KIND> __global__ void __launch_bounds__(256) rate_kernel(int *out, int seed) {
  int v[8];
  for(int k = 0; k < 8; ++k) v[k] = seed + threadIdx.x * (k + 1);
  int a = seed * 3 + 1, b = seed * 5 + 2;
  for(int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for(int r = 0; r < 8; ++r) {
#pragma unroll
      for(int k = 0; k < 8; ++k) {
        if(KIND == 0) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
        if(KIND == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[k]) : "v"(b));
        if(KIND == 2) v[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b + k), v[k], false);
        if(KIND == 3) v[k] = __builtin_amdgcn_alignbit(v[k], a + k, 31);
        if(KIND == 4) v[k] = __builtin_amdgcn_update_dpp(a, v[k], 0x138, 0xf, 0xf, false); // wave_shr:1
        if(KIND == 5) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
        if(KIND == 7) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[k]) : "v"(a));
        if(KIND == 8) asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(v[k]), "v"(a) : "vcc");
        if(KIND == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
        if(KIND == 6) v[k] = __builtin_amdgcn_sdot4(a, b + k, v[k], false);               // v_dot4_i32_i8
      }
    }
    a += it;
  }
  int s = 0;
  for(int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, int *d_out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU = 8 waves per SIMD
  rate_kernel<KIND><<<blocks, 256>>>(d_out, 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  rate_kernel<KIND><<<blocks, 256>>>(d_out, 2);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * 4 * N_ITER * 64;   // wave-instructions of the measured kind
  double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
  printf("%-28s %8.3f ms  %.3f wave-instr/ns/SIMD  -> %.2f cycles per wave-instr at 2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
  int *d_out; (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  run<0>("v_max_i32", d_out);
  run<1>("v_sub_u32", d_out);
  run<2>("v_dot2_i32_i16", d_out);
  run<3>("v_alignbit_b32", d_out);
  run<4>("v_mov_b32_dpp wave_shr:1", d_out);
  run<5>("v_max3_i32", d_out);
  run<6>("v_dot4_i32_i8", d_out);
  run<7>("v_pk_max_i16", d_out);
  run<8>("v_cmp_gt_i32 (vcc)", d_out);
  run<9>("v_add3_u32", d_out);
  return 0;
}
