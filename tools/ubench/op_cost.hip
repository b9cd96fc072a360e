// Saturated issue cost (cycles per wave64 instruction and SIMD, at 1 / 2 / 4 waves per SIMD) of the individual VALU instructions the
// epilogue chains of p2vit_{gemm,ln,attn,misc}.hip are made of (gfx950).  Eight independent destination registers per instruction kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 1024
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, ASM)                                                                                     \
  __global__ void k_##NAME(unsigned* out, unsigned a, unsigned b, int n) {                                    \
    unsigned x[8];                                                                                            \
    unsigned long long d[8];                                                                                  \
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i * 40503u + a; d[i] = ((unsigned long long)x[i] << 20) | b; } \
    long long t0 = clock64();                                                                                 \
    for (int i = 0; i < n; ++i) {                                                                             \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile(ASM : "+v"(x[j]), "+v"(d[j]) : "v"(a), "v"(b) : "vcc"); \
    }                                                                                                         \
    long long t1 = clock64();                                                                                 \
    unsigned s = 0;                                                                                           \
    for (int i = 0; i < 8; ++i) s += x[i] + (unsigned)d[i];                                                   \
    if (s == 0x12345678u) out[2] = s;                                                                         \
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned)((t1 - t0) * 16 / (n * 8)));                        \
  }
// %0: 32-bit VGPR (in/out), %1: 64-bit VGPR pair (in/out), %2 / %3: 32-bit VGPR inputs
KERNEL(add_f32, "v_add_f32 %0, %0, %2")
KERNEL(mul_f32, "v_mul_f32 %0, %0, %2")
KERNEL(fma_f32, "v_fma_f32 %0, %0, %2, %3")
KERNEL(rndne, "v_rndne_f32 %0, %0")
KERNEL(ldexp, "v_ldexp_f32 %0, %0, %2")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %2, 1, %0")
KERNEL(med3_f32, "v_med3_f32 %0, %0, %2, %3")
KERNEL(max3_f32, "v_max3_f32 %0, %0, %2, %3")
KERNEL(and_b32, "v_and_b32 %0, %0, %2")
KERNEL(bfe_i32, "v_bfe_i32 %0, %0, 8, 8")
KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 23, 8")
KERNEL(add_u32, "v_add_u32 %0, %0, %2")
KERNEL(sub_u32, "v_sub_u32 %0, %2, %0")
KERNEL(add3_u32, "v_add3_u32 %0, %0, %2, %3")
KERNEL(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %2")
KERNEL(ashrrev, "v_ashrrev_i32 %0, 9, %0")
KERNEL(med3_i32, "v_med3_i32 %0, %0, %2, %3")
KERNEL(mul_i24, "v_mul_i32_i24 %0, %0, %2")
KERNEL(mul_i24_sdwa, "v_mul_i32_i24_sdwa %0, %2, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL(mad_i24, "v_mad_i32_i24 %0, %0, %2, %3")
KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %0, %3")
KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %2")
KERNEL(perm, "v_perm_b32 %0, %0, %2, %3")
KERNEL(xor, "v_xor_b32 %0, 0x80808080, %0")
KERNEL(cmp_cndmask, "v_cmp_ge_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc")
KERNEL(cmp_cndmask_sdwa, "v_cmp_ge_f32 vcc, %2, %3\n\tv_cndmask_b32_sdwa %0, %2, %2, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1")
KERNEL(add64, "v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %0, vcc, %0, %3, vcc")
KERNEL(lshl_add_u64, "v_lshl_add_u64 %1, %1, 0, %1")
KERNEL(mul_f64, "v_mul_f64 %1, %1, %1")
KERNEL(cvt_f32_f64, "v_cvt_f32_f64 %0, %1")
KERNEL(cvt_f64_i32, "v_cvt_f64_i32 %1, %0")
KERNEL(fma_f64, "v_fma_f64 %1, %1, %1, %1")
KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %2")
KERNEL(pk_mul_f32, "v_pk_mul_f32 %1, %1, %1")
KERNEL(pk_fma_f32, "v_pk_fma_f32 %1, %1, %1, %1")
KERNEL(add_dpp, "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(permlane32_swap, "v_permlane32_swap_b32 %0, %0")
KERNEL(rcp_f32, "v_rcp_f32 %0, %0")
KERNEL(sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL(div_scale, "v_div_scale_f32 %0, vcc, %0, %2, %0")
KERNEL(div_fmas, "v_div_fmas_f32 %0, %0, %2, %3")
KERNEL(div_fixup, "v_div_fixup_f32 %0, %0, %2, %3")
#define RUN(NAME, NINSTR)                                                                                             \
  do {                                                                                                                \
    float r[3];                                                                                                       \
    int w = 0;                                                                                                        \
    for (int wps = 1; wps <= 4; wps *= 2) {                                                                           \
      hipMemset(d, 0, 64);                                                                                            \
      hipLaunchKernelGGL(k_##NAME, dim3(256), dim3(256 * wps), 0, 0, d, 0x3f9e3779u, 0x40490fdbu, ITER);              \
      hipDeviceSynchronize();                                                                                         \
      unsigned h;                                                                                                     \
      hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);                                                                     \
      r[w++] = (float)h / 16.0f / wps / NINSTR;                                                                       \
    }                                                                                                                 \
    printf("%-18s %6.2f %6.2f %6.2f   cycles per instruction and SIMD at 1 / 2 / 4 waves per SIMD\n", #NAME, r[0], r[1], r[2]); \
  } while (0)
int main() {
  unsigned* d; hipMalloc(&d, 64);
  RUN(add_f32, 1); RUN(mul_f32, 1); RUN(fma_f32, 1); RUN(rndne, 1); RUN(ldexp, 1); RUN(cvt_f32_i32, 1); RUN(cvt_i32_f32, 1); RUN(cvt_u32_f32, 1);
  RUN(cvt_f32_ubyte0, 1); RUN(cvt_pk_u8, 1); RUN(med3_f32, 1); RUN(max3_f32, 1); RUN(and_b32, 1); RUN(bfe_i32, 1); RUN(bfe_u32, 1); RUN(add_u32, 1);
  RUN(sub_u32, 1); RUN(add3_u32, 1); RUN(lshl_add_u32, 1); RUN(ashrrev, 1); RUN(med3_i32, 1); RUN(mul_i24, 1); RUN(mul_i24_sdwa, 1); RUN(mad_i24, 1);
  RUN(mad_u24, 1); RUN(mul_lo_u32, 1); RUN(perm, 1); RUN(xor, 1); RUN(cmp_cndmask, 2); RUN(cmp_cndmask_sdwa, 2); RUN(add64, 2); RUN(lshl_add_u64, 1);
  RUN(mul_f64, 1); RUN(cvt_f32_f64, 1); RUN(cvt_f64_i32, 1); RUN(fma_f64, 1); RUN(pk_add_u16, 1); RUN(pk_mul_f32, 1); RUN(pk_fma_f32, 1);
  RUN(add_dpp, 1); RUN(permlane32_swap, 1); RUN(rcp_f32, 1); RUN(sqrt_f32, 1); RUN(div_scale, 1); RUN(div_fmas, 1); RUN(div_fixup, 1);
  return 0;
}
