// tools/ubench/valu_rates2.hip -- issue rates of a broad set of gfx950 VALU instructions (round 2).
// Round 1 found v_sub_u32 at ~2.5 cycles per wave64 instruction but v_max_i32 / v_alignbit / v_dot* at ~4.3: some
// instruction classes issue at full rate, others at half.  Which ones decides how the DP cell should be written, so
// every candidate is measured here, at 8 and at 4 waves per SIMD on every CU.
// Every kernel runs N_ITER x 64 instructions of one kind per wave on 8 independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 1000

#define BODY(ASM, ...)                                                                                     \
  int v[8];                                                                                                \
  for(int k = 0; k < 8; ++k) v[k] = seed + threadIdx.x * (k + 1);                                          \
  int a = seed * 3 + 1 + threadIdx.x, b = seed * 5 + 2;                                                    \
  for(int it = 0; it < N_ITER; ++it) {                                                                     \
    _Pragma("unroll") for(int r = 0; r < 8; ++r) {                                                         \
      _Pragma("unroll") for(int k = 0; k < 8; ++k) { asm volatile(ASM : "+v"(v[k]) : "v"(a), "v"(b) : __VA_ARGS__); } \
    }                                                                                                      \
    a += it;                                                                                               \
  }                                                                                                        \
  int s = 0;                                                                                               \
  for(int k = 0; k < 8; ++k) s += v[k];                                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;

#define DEF(NAME, ASM, ...) \
  __global__ void __launch_bounds__(256) k_##NAME(int *out, int seed) { BODY(ASM, __VA_ARGS__) }

// two-operand (dst, dst, a)
DEF(add_u32, "v_add_u32 %0, %0, %1", "memory")
DEF(sub_u32, "v_sub_u32 %0, %0, %1", "memory")
DEF(max_i32, "v_max_i32 %0, %0, %1", "memory")
DEF(min_i32, "v_min_i32 %0, %0, %1", "memory")
DEF(max_u32, "v_max_u32 %0, %0, %1", "memory")
DEF(and_b32, "v_and_b32 %0, %0, %1", "memory")
DEF(or_b32, "v_or_b32 %0, %0, %1", "memory")
DEF(xor_b32, "v_xor_b32 %0, %0, %1", "memory")
DEF(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0", "memory")
DEF(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0", "memory")
DEF(ashrrev_i32, "v_ashrrev_i32 %0, 31, %0", "memory")
DEF(mov_b32, "v_mov_b32 %0, %1", "memory")
DEF(cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "memory")
DEF(add_co_u32, "v_add_co_u32 %0, vcc, %0, %1", "vcc", "memory")
DEF(sub_co_u32, "v_sub_co_u32 %0, vcc, %0, %1", "vcc", "memory")
DEF(addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %0, vcc", "vcc", "memory")
DEF(subb_co_u32, "v_subb_co_u32 %0, vcc, %0, %1, vcc", "vcc", "memory")
DEF(cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %1", "vcc", "memory")
DEF(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1", "vcc", "memory")
DEF(cmp_lt_i32_s, "v_cmp_lt_i32 s[20:21], %0, %1", "s20", "s21", "memory")
DEF(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1", "vcc", "memory")
DEF(mul_i32_i24, "v_mul_i32_i24 %0, %0, %1", "memory")
DEF(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2", "memory")
DEF(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", "memory")
DEF(add3_u32, "v_add3_u32 %0, %0, %1, %2", "memory")
DEF(lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1", "memory")
DEF(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 1", "memory")
DEF(lshl_or_b32, "v_lshl_or_b32 %0, %0, 1, %1", "memory")
DEF(and_or_b32, "v_and_or_b32 %0, %0, %1, %2", "memory")
DEF(or3_b32, "v_or3_b32 %0, %0, %1, %2", "memory")
DEF(bfi_b32, "v_bfi_b32 %0, %1, %0, %2", "memory")
DEF(bfe_u32, "v_bfe_u32 %0, %0, 3, 8", "memory")
DEF(bfe_i32, "v_bfe_i32 %0, %0, 3, 8", "memory")
DEF(perm_b32, "v_perm_b32 %0, %0, %1, %2", "memory")
DEF(alignbit, "v_alignbit_b32 %0, %0, %1, 31", "memory")
DEF(alignbyte, "v_alignbyte_b32 %0, %0, %1, 3", "memory")
DEF(max3_i32, "v_max3_i32 %0, %0, %1, %2", "memory")
DEF(min3_i32, "v_min3_i32 %0, %0, %1, %2", "memory")
DEF(med3_i32, "v_med3_i32 %0, %0, %1, %2", "memory")
DEF(max3_u32, "v_max3_u32 %0, %0, %1, %2", "memory")
DEF(sad_u32, "v_sad_u32 %0, %0, %1, %2", "memory")
DEF(xad_u32, "v_xad_u32 %0, %0, %1, %2", "memory")
DEF(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", "memory")
// fp32
DEF(add_f32, "v_add_f32 %0, %0, %1", "memory")
DEF(sub_f32, "v_sub_f32 %0, %0, %1", "memory")
DEF(mul_f32, "v_mul_f32 %0, %0, %1", "memory")
DEF(max_f32, "v_max_f32 %0, %0, %1", "memory")
DEF(min_f32, "v_min_f32 %0, %0, %1", "memory")
DEF(fma_f32, "v_fma_f32 %0, %0, %1, %2", "memory")
DEF(fmac_f32, "v_fmac_f32 %0, %1, %2", "memory")
DEF(max3_f32, "v_max3_f32 %0, %0, %1, %2", "memory")
DEF(med3_f32, "v_med3_f32 %0, %0, %1, %2", "memory")
DEF(cvt_f32_i32, "v_cvt_f32_i32 %0, %0", "memory")
DEF(cvt_i32_f32, "v_cvt_i32_f32 %0, %0", "memory")
// 16-bit packed / scalar 16
DEF(pk_max_i16, "v_pk_max_i16 %0, %0, %1", "memory")
DEF(pk_min_i16, "v_pk_min_i16 %0, %0, %1", "memory")
DEF(pk_add_i16, "v_pk_add_i16 %0, %0, %1", "memory")
DEF(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1", "memory")
DEF(pk_add_u16, "v_pk_add_u16 %0, %0, %1", "memory")
DEF(pk_max_u16, "v_pk_max_u16 %0, %0, %1", "memory")
DEF(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2", "memory")
DEF(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1", "memory")
DEF(pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0", "memory")
DEF(pk_ashrrev_i16, "v_pk_ashrrev_i16 %0, 15, %0", "memory")
DEF(pk_max_f16, "v_pk_max_f16 %0, %0, %1", "memory")
DEF(pk_add_f16, "v_pk_add_f16 %0, %0, %1", "memory")
DEF(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2", "memory")
DEF(max_i16, "v_max_i16 %0, %0, %1", "memory")
DEF(add_u16, "v_add_u16 %0, %0, %1", "memory")
// dots
DEF(dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0", "memory")
DEF(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0", "memory")
DEF(dot4c_i32_i8, "v_dot4c_i32_i8 %0, %1, %2", "memory")
DEF(dot8_i32_i4, "v_dot8_i32_i4 %0, %1, %2, %0", "memory")
DEF(dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0", "memory")
DEF(dot2c_f32_bf16, "v_dot2c_f32_bf16 %0, %1, %2", "memory")
// cross-lane
DEF(mov_dpp_shr1, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "memory")
DEF(mov_dpp_rowshr1, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "memory")
DEF(max_dpp_rowshr1, "v_max_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "memory")
DEF(add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "memory")
DEF(max_sdwa, "v_max_i32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "memory")

// 64-bit packed fp32 ops: operate on register pairs
#define DEF64(NAME, ASM)                                                                                               \
  __global__ void __launch_bounds__(256) k_##NAME(int *out, int seed) {                                               \
    double v[4];                                                                                                       \
    for(int k = 0; k < 4; ++k) v[k] = seed + threadIdx.x * (k + 1);                                                    \
    double a = seed * 3 + 1 + threadIdx.x, b = seed * 5 + 2;                                                           \
    for(int it = 0; it < N_ITER; ++it) {                                                                               \
      _Pragma("unroll") for(int r = 0; r < 16; ++r) {                                                                  \
        _Pragma("unroll") for(int k = 0; k < 4; ++k) { asm volatile(ASM : "+v"(v[k]) : "v"(a), "v"(b) : "memory"); }   \
      }                                                                                                                \
      a += it;                                                                                                         \
    }                                                                                                                  \
    double s = 0;                                                                                                      \
    for(int k = 0; k < 4; ++k) s += v[k];                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;                                                               \
  }
DEF64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEF64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEF64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEF64(pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
DEF64(lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
DEF64(add_f64, "v_add_f64 %0, %0, %1")

typedef void (*kern_t)(int *, int);
static void run(const char *name, kern_t k, int *d_out, int blocks_per_cu, FILE *f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu; // blocks of 256 threads per CU = waves per SIMD
  k<<<blocks, 256>>>(d_out, 1);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for(int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k<<<blocks, 256>>>(d_out, 2 + rep);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  double winstr = (double)blocks * 4 * N_ITER * 64;
  double per_simd_per_s = winstr / (best * 1e-3) / 1024.0;
  fprintf(f, "%-18s waves/SIMD %d  %8.3f ms  -> %.2f cycles per wave-instr at 2.4 GHz\n", name, blocks_per_cu, best, 2.4e9 / per_simd_per_s);
  fflush(f);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

#define RUN(NAME)                          \
  run(#NAME, k_##NAME, d_out, 8, stdout);  \
  run(#NAME, k_##NAME, d_out, 4, stdout);

int main() {
  int *d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  RUN(add_u32) RUN(sub_u32) RUN(max_i32) RUN(min_i32) RUN(max_u32) RUN(and_b32) RUN(or_b32) RUN(xor_b32) RUN(lshlrev_b32)
  RUN(lshrrev_b32) RUN(ashrrev_i32) RUN(mov_b32) RUN(cndmask) RUN(add_co_u32) RUN(sub_co_u32) RUN(addc_co_u32) RUN(subb_co_u32)
  RUN(cmp_lt_i32) RUN(cmp_lt_u32) RUN(cmp_lt_i32_s) RUN(cmp_lt_f32) RUN(mul_i32_i24) RUN(mad_i32_i24) RUN(mad_u32_u24) RUN(add3_u32)
  RUN(lshl_add_u32) RUN(add_lshl_u32) RUN(lshl_or_b32) RUN(and_or_b32) RUN(or3_b32) RUN(bfi_b32) RUN(bfe_u32) RUN(bfe_i32) RUN(perm_b32)
  RUN(alignbit) RUN(alignbyte) RUN(max3_i32) RUN(min3_i32) RUN(med3_i32) RUN(max3_u32) RUN(sad_u32) RUN(xad_u32) RUN(mul_lo_u32)
  RUN(add_f32) RUN(sub_f32) RUN(mul_f32) RUN(max_f32) RUN(min_f32) RUN(fma_f32) RUN(fmac_f32) RUN(max3_f32) RUN(med3_f32)
  RUN(cvt_f32_i32) RUN(cvt_i32_f32)
  RUN(pk_max_i16) RUN(pk_min_i16) RUN(pk_add_i16) RUN(pk_sub_i16) RUN(pk_add_u16) RUN(pk_max_u16) RUN(pk_mad_i16) RUN(pk_mul_lo_u16)
  RUN(pk_lshlrev_b16) RUN(pk_ashrrev_i16) RUN(pk_max_f16) RUN(pk_add_f16) RUN(pk_fma_f16) RUN(max_i16) RUN(add_u16)
  RUN(dot2_i32_i16) RUN(dot4_i32_i8) RUN(dot4c_i32_i8) RUN(dot8_i32_i4) RUN(dot2_f32_f16) RUN(dot2c_f32_bf16)
  RUN(mov_dpp_shr1) RUN(mov_dpp_rowshr1) RUN(max_dpp_rowshr1) RUN(add_sdwa) RUN(max_sdwa)
  RUN(pk_add_f32) RUN(pk_mul_f32) RUN(pk_fma_f32) RUN(pk_mov_b32) RUN(lshlrev_b64) RUN(add_f64)
  return 0;
}
