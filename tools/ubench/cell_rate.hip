// tools/ubench/cell_rate.hip -- what the DP's score-only cell sequence (dp_internal.hpp, UNI + DOT4: v_max, v_dot4c, v_max,
// v_add, v_max3, v_subrev per cell) can issue at by itself: C cells per step as in dp_fill_kernel, no loads, no stores, no
// per-step bookkeeping; W waves per SIMD on every CU.  Prints cells per second of the whole chip, to hold against the fill
// kernel's own rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int C, int NOPS>
__global__ void __launch_bounds__(64) k_cells(int *out, int seed, int steps) {
  int w0[C], w2[C], hop[C], f[C];
  for(int c = 0; c < C; ++c) {
    w0[c] = seed * (c + 3) + threadIdx.x;
    w2[c] = seed + c;
    hop[c] = -c;
    f[c] = -1000 - c;
  }
  int e = -1000, a = seed * 7 + threadIdx.x, gop = seed & 15;
  int diag = seed;
  for(int t = 0; t < steps; ++t) {
    int d = diag + w2[0], dn;
    int hl = hop[C - 1] ^ t; // stands for the value handed over by the left lane
#pragma unroll
    for(int c = 0; c < C; ++c) {
      int h;
      if(NOPS) {
        asm volatile("v_max_i32 %[e], %[e], %[hl]\n\t"
                     "v_dot4c_i32_i8 %[d], %[a], %[w0]\n\t"
                     "v_max_i32 %[f], %[f], %[hop]\n\t"
                     "v_add_u32 %[dn], %[hop], %[w2n]\n\t"
                     "s_nop 0\n\t"
                     "v_max3_i32 %[h], %[d], %[e], %[f]\n\t"
                     "v_subrev_u32 %[hop], %[gop], %[h]\n\t"
                     : [e] "+v"(e), [d] "+v"(d), [f] "+v"(f[c]), [dn] "=&v"(dn), [h] "=&v"(h), [hop] "+v"(hop[c])
                     : [hl] "v"(hl), [a] "v"(a), [w0] "v"(w0[c]), [w2n] "v"(w2[(c + 1) % C]), [gop] "s"(gop));
      }
      else { // the same six without the s_nop, the dot's result read one instruction earlier than the hardware allows: timing only
        asm volatile("v_max_i32 %[e], %[e], %[hl]\n\t"
                     "v_dot4c_i32_i8 %[d], %[a], %[w0]\n\t"
                     "v_max_i32 %[f], %[f], %[hop]\n\t"
                     "v_add_u32 %[dn], %[hop], %[w2n]\n\t"
                     "v_max3_i32 %[h], %[d], %[e], %[f]\n\t"
                     "v_subrev_u32 %[hop], %[gop], %[h]\n\t"
                     : [e] "+v"(e), [d] "+v"(d), [f] "+v"(f[c]), [dn] "=&v"(dn), [h] "=&v"(h), [hop] "+v"(hop[c])
                     : [hl] "v"(hl), [a] "v"(a), [w0] "v"(w0[c]), [w2n] "v"(w2[(c + 1) % C]), [gop] "s"(gop));
      }
      hl = hop[c];
      d = dn;
    }
    diag = hl;
    a += t;
  }
  int s = e + diag;
  for(int c = 0; c < C; ++c) {
    s += hop[c] + f[c];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the walk's cell (with the four decision bits, dp_internal.hpp PM_CELL_*_T; DOT4, not UNI), C2 cells per step, two DPP hand-offs,
// one LDS read and one LDS write per step as in dp_walk_kernel's block loop
template <int C2>
__global__ void __launch_bounds__(64) k_trace(int *out, int seed, int steps) {
  __shared__ int2 rows[64];
  __shared__ unsigned short bits[64][64];
  int w0[C2], w2[C2], hop[C2], f[C2];
  for(int c = 0; c < C2; ++c) {
    w0[c] = seed * (c + 3) + threadIdx.x;
    w2[c] = seed + c;
    hop[c] = -c;
    f[c] = -1000 - c;
  }
  rows[threadIdx.x] = make_int2(seed + threadIdx.x, seed * 3);
  __syncthreads();
  int e = -1000, gop = seed & 15, diag = seed;
  for(int t = 0; t < steps; ++t) {
    const int2 a = rows[(t + threadIdx.x) & 63];
    int hl = __builtin_amdgcn_update_dpp(hop[C2 - 1], hop[C2 - 1], 0x138, 0xf, 0xf, false);
    e = __builtin_amdgcn_update_dpp(e, e, 0x138, 0xf, 0xf, false);
    int d, dn;
    unsigned acc = 0;
    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a.y), "v"(w2[0]), "v"(diag));
    const int hl0 = hl;
#pragma unroll
    for(int c = 0; c < C2; ++c) {
      int h, tt;
      asm volatile("v_sub_u32 %[t], %[hl], %[e]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\tv_max_i32 %[e], %[e], %[hl]\n\t"
                   "v_dot4c_i32_i8 %[d], %[ax], %[w0]\n\t"
                   "v_sub_u32 %[t], %[hop], %[f]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\tv_max_i32 %[f], %[f], %[hop]\n\t"
                   "v_dot2_i32_i16 %[dn], %[ay], %[w2n], %[hop]\n\t"
                   "v_max3_i32 %[h], %[d], %[e], %[f]\n\tv_sub_u32 %[t], %[d], %[h]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\t"
                   "v_sub_u32 %[t], %[e], %[f]\n\tv_alignbit_b32 %[acc], %[acc], %[t], 31\n\t"
                   "v_subrev_u32 %[hop], %[gop], %[h]"
                   : [e] "+v"(e), [d] "+v"(d), [f] "+v"(f[c]), [dn] "=&v"(dn), [h] "=&v"(h), [hop] "+v"(hop[c]), [acc] "+v"(acc), [t] "=&v"(tt)
                   : [hl] "v"(hl), [ax] "v"(a.x), [ay] "v"(a.y), [w0] "v"(w0[c]), [w2n] "v"(w2[(c + 1) % C2]), [gop] "s"(gop));
      hl = hop[c];
      d = dn;
    }
    diag = hl0;
    bits[t & 63][threadIdx.x] = (unsigned short)acc;
  }
  int s = e + diag + bits[3][threadIdx.x];
  for(int c = 0; c < C2; ++c) {
    s += hop[c] + f[c];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C2> static void run_trace(int *out, int waves_per_simd, int steps) {
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k_trace<C2><<<blocks, 64>>>(out, 3, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_trace<C2><<<blocks, 64>>>(out, 3, steps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double)steps * (C2 * 14 + 8); // per wave: cells + hand-off, LDS, loop
  printf("trace C2=%d waves/SIMD=%d  %.3f ms  %.1f cycles per step of one SIMD, ~%.2f cycles per instruction per SIMD\n", C2, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * steps), ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * instr));
}

template <int C, int NOPS> static void run(int *out, int waves_per_simd, int steps) {
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k_cells<C, NOPS><<<blocks, 64>>>(out, 3, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_cells<C, NOPS><<<blocks, 64>>>(out, 3, steps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double cells = (double)blocks * 64 * C * steps;
  printf("C=%2d nops=%d waves/SIMD=%d  %.3f ms  %.2f T cells/s  (%.1f cycles per cell-instruction-row of one wave at 2.4 GHz)\n", C, NOPS, waves_per_simd,
         ms, cells / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * C * steps));
}

int main() {
  int *out;
  hipMalloc(&out, 256 * 4 * 16 * 64 * 4);
  for(int w : {1, 2, 3, 4, 5, 8}) {
    run<16, 1>(out, w, 20000);
  }
  for(int w : {1, 2, 4, 5, 8}) {
    run<16, 0>(out, w, 20000);
  }
  for(int w : {3, 4}) {
    run<32, 1>(out, w, 10000);
  }
  for(int w : {1, 2, 3, 4, 6}) {
    run_trace<4>(out, w, 20000);
  }
  for(int w : {1, 3}) {
    run_trace<8>(out, w, 10000);
  }
  return 0;
}
