// p2vit_device.h -- device helpers shared by the kernel translation units of libp2vit_hip.so (CDNA4 / gfx950).
//
// Built with -ffp-contract=off: every fp32 epilogue reproduces the reference's eager fp32 operation order (one IEEE rounding per
// torch op), so no mul+add may be fused behind our back.  `/` and sqrtf are the correctly-rounded forms (hipcc default), rintf is
// v_rndne_f32 (half-to-even, = torch.round).
// Reference citations are relative to /root/reference (LeSN-Lab/diff-ViT).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "p2vit_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------
// clamp to the int8 grid in ONE v_med3_f32: fminf(fmaxf(r, lo), hi) compiles to a canonicalising v_max + v_med3 (the IEEE min/max of
// a possibly signalling NaN); for a NaN both forms return -128 (v_med3 falls back to min3)
__device__ __forceinline__ float clamp8f(float r) { return __builtin_amdgcn_fmed3f(r, -128.f, 127.f); }
// clamp(round(v), -128, 127)  == UniformQuantizer.quant for int8 (quantizer/uniform.py:85-87)
__device__ __forceinline__ int sat8(float v) { return (int)clamp8f(rintf(v)); }
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
  return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)d << 24);
}
__device__ __forceinline__ int sx8(unsigned w, int i) { return (int)(int8_t)(w >> (8 * i)); }
// clamp(r, -128, 127) of four INTEGRAL floats (already rounded with rintf) as four int8 bytes.  v_cvt_pk_u8_f32 saturates to
// [0,255] (measured on gfx950: 256, 1e9, +inf -> 255; -1, -1e9, -inf, NaN -> 0) and r + 128 is exact for |r| < 2^24 (beyond that
// the value saturates anyway), so  clamp(r,-128,127) == (sat_u8(r + 128)) ^ 0x80  byte-wise: 2 instructions per value + 1 per dword.
__device__ __forceinline__ unsigned pack4_sat(float r0, float r1, float r2, float r3) {
  unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(r0 + 128.f, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(r1 + 128.f, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(r2 + 128.f, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(r3 + 128.f, 3, w);
  return w ^ 0x80808080u;
}
__device__ __forceinline__ float sat8f(float v) { return clamp8f(rintf(v)); }
// clamp(rint(o), -128, 127) of four fp32 values as four int8 bytes WITHOUT v_rndne / v_cvt_pk: the clamp commutes with the rounding (its
// bounds are integers, rint is monotone), and adding 1.5 * 2^23 rounds the clamped value to an integer (round-half-even: the addition's own
// rounding on the unit grid) whose two's-complement code is the low byte of the sum; the SDWA form writes that byte into its place of the
// packed dword.  v_med3 + v_add_f32_sdwa = 10.4 cycles per value against 14.4 for rndne, +128, cvt_pk_u8 (profiles/r03_op_cost.txt).
// Finite inputs only (a NaN would not give the -128 that v_cvt_pk_u8_f32 does).
__device__ __forceinline__ float pre_pack(float o) { return clamp8f(o); }
__device__ __forceinline__ unsigned pack4_pre(float c0, float c1, float c2, float c3) {       // c = pre_pack(o)
  const float magic = 12582912.f;
  unsigned w = __float_as_uint(c0 + magic);                           // byte 0; bytes 1-3 are overwritten below
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c1), "v"(magic));
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c2), "v"(magic));
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c3), "v"(magic));
  return w;
}
__device__ __forceinline__ unsigned pack4_rne_sat(float o0, float o1, float o2, float o3) {
  return pack4_pre(pre_pack(o0), pre_pack(o1), pre_pack(o2), pre_pack(o3));
}

// Exchange between the two 32-lane halves so that each lane ends with 16 CONTIGUOUS bytes of an MFMA
// 32x32 accumulator column block.  In: d[g] = bytes [8g+4h, 8g+4h+4) (h = lane>>5).
// Out (as uint4 x,y,z,w): bytes [16h, 16h+16).
__device__ __forceinline__ uint4 halves_to_row16(unsigned d0, unsigned d1, unsigned d2, unsigned d3) {
  auto r02 = __builtin_amdgcn_permlane32_swap(d0, d2, false, false);
  auto r13 = __builtin_amdgcn_permlane32_swap(d1, d3, false, false);
  return make_uint4(r02[0], r02[1], r13[0], r13[1]);
}
// Inverse: in = 16 contiguous bytes [16h,16h+16) as uint4; out g[i] = bytes [8i+4h, 8i+4h+4).
__device__ __forceinline__ void row16_to_halves(uint4 e, unsigned& g0, unsigned& g1, unsigned& g2, unsigned& g3) {
  auto r01 = __builtin_amdgcn_permlane32_swap(e.x, e.y, false, false);
  auto r23 = __builtin_amdgcn_permlane32_swap(e.z, e.w, false, false);
  g0 = r01[0]; g2 = r01[1]; g1 = r23[0]; g3 = r23[1];
}

// ---------------------------------------------------------------------------------------------------
// Packed int4 weights (p2v_linear.packed4, include/p2vit.h): two codes per byte.  gfx950 has no int4 MFMA, so a fragment is
// widened in registers to the int8 operand of v_mfma_i32_32x32x32_i8 -- as (code << 4), i.e. 16 x code, which is exact in int8
// ([-128, 112]) and costs 3 VALU per dword (shift, and, and) instead of a sign extension per nibble; the accumulator then holds
// 16 x the true sum, its conversion to fp32 is still exact (a 24-bit integer shifted by 4), and the 1/16 is folded into the
// power-of-two column scale when the epilogue constants are staged.
//   8 bytes of a lane = its 16 consecutive k: byte j of dword 0 = code[j] | code[4+j] << 4, of dword 1 = code[8+j] | code[12+j] << 4,
//   so the even / odd nibble planes come out as the dwords [0..3], [4..7], [8..11], [12..15] of the int8 fragment.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ v4i unpack_w4(unsigned lo, unsigned hi) {
  const unsigned m = 0xF0F0F0F0u;
  return (v4i){(int)((lo << 4) & m), (int)(lo & m), (int)((hi << 4) & m), (int)(hi & m)};
}
// byte offset of the 8-byte chunk c (k = 16c .. 16c+15) of row `row` inside a packed [128][32 B] tile: chunk index XOR-swizzled by
// (row>>3)&3 so that a ds_read_b64 of 32 rows x one chunk touches every bank once
__device__ __forceinline__ int lds_off_w4(int row, int c) { return row * 32 + ((c ^ ((row >> 3) & 3)) << 3); }

// ---------------------------------------------------------------------------------------------------
// GELU -> PoT requant.  Canonical value: q = clamp(rne(RN32(gelu(y)) / s)), gelu(y) = 0.5*y*erfc(-y/sqrt2)
// (reference: float nn.GELU then QAct, layers_quant.py:331-333).  Fast path: A&S 7.1.26 erfc (|err| <=
// 1.5e-7) in fp32; its total error is far below GELU_EPS, so whenever the scaled value is further than
// GELU_EPS/s from a rounding boundary the code is already decided.  Otherwise (about 1e-4 of the
// elements) the lane takes the fp64 path.  tests/test_engine_gpu.py::test_gelu_fast_path_bound_and_exactness sweeps the fp32 line
// to check the bound.  (Frozen plans use the exact threshold table below instead; this path serves scales without a table.)
// ---------------------------------------------------------------------------------------------------
#define GELU_EPS 1.2e-6f   // measured max |gelu_fast - RN32(gelu)| over all fp32 in +-[2^-20,32): 4.8e-7 (tools/gelu_stats.py)
// approximation only (its error is bounded by the exhaustive sweep in tests): fused multiply-adds are fine here
__device__ __forceinline__ float gelu_fast(float y) {
  const float z = fabsf(y) * 0.70710678f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);       // 0.5 * A&S 7.1.26 polynomial
  p = __builtin_fmaf(t, p, 0.5f * 1.421413741f);
  p = __builtin_fmaf(t, p, 0.5f * -0.284496736f);
  p = __builtin_fmaf(t, p, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504f);
  const float hc = (p * t) * e;                     // 0.5*erfc(|y|/sqrt2) = Phi(-|y|)
  return fmaxf(y, 0.f) - fabsf(y) * hc;             // y*Phi(y) = relu(y) - |y|*Phi(-|y|)
}
__device__ __noinline__ inline float gelu_exact(float y) {
  double yd = (double)y;
  return (float)(0.5 * yd * erfc(-yd * 0.70710678118654752440));
}
// clamp(rne(x), -128, 127) of an fp32 value that is within `eps` (absolute) of the exact pre-rounding value:
// decided iff x is further than eps from a rounding boundary.  (|x| >= 2^23 has no fraction: always decided.)
__device__ __forceinline__ bool rne_decided(float x, float r, float eps) { return fabsf(x - r) < 0.5f - eps; }

__device__ __forceinline__ int gelu_q8(float y, float inv_s, bool force_slow, bool* took_slow) {
  const float t = gelu_fast(y) * inv_s;
  float r = rintf(t);
  const bool slow = force_slow || !rne_decided(t, r, GELU_EPS * inv_s) || !(GELU_EPS * inv_s < 0.25f);
  if (__builtin_amdgcn_ballot_w64(slow) != 0) {           // wave-uniform branch: ~1e-4 of the lanes need the fp64 value
    if (slow) {
      if (took_slow) *took_slow = true;
      r = rintf(gelu_exact(y) * inv_s);
    }
  }
  return (int)clamp8f(r);
}

// Four at a time: the fast values are computed branch-free (instruction-level parallelism across the four
// dependent chains), ONE wave-uniform branch covers the rare lanes that need the fp64 value.
__device__ __forceinline__ void gelu_q8x4(const float (&y)[4], float inv_s, float (&q)[4]) {
  float r[4];
  bool slow[4], any = !(GELU_EPS * inv_s < 0.25f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = gelu_fast(y[i]) * inv_s;
    r[i] = rintf(t);
    slow[i] = !rne_decided(t, r[i], GELU_EPS * inv_s);
    any |= slow[i];
  }
  if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (slow[i] || !(GELU_EPS * inv_s < 0.25f)) r[i] = rintf(gelu_exact(y[i]) * inv_s);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = r[i];          // integral, NOT clamped: the byte packing saturates
}

// ---------------------------------------------------------------------------------------------------
// GELU -> PoT requant as an EXACT threshold table (p2v_gelu_tab, include/p2vit.h).
//   code(y) = clamp(rne(RN32(gelu(y)) * 2^e)) is a step function of the fp32 pre-activation y with < 256 steps.  The y axis is
//   cut into cells of width s/2 (k = 2/s, a power of two, so u = y*k is exact):  i = clamp(floor(u) + off, 0, cells-1).
//   Steps of the monotone branch are >= s/1.13 apart, so a cell holds at most ONE step (the builder verifies this for every
//   cell, also around the minimum of GELU at y = -0.7518 where a down- and an up-step can come close); the entry is
//   { thr * k, lo | hi << 8 } and code = u >= thr * k ? hi : lo  (thr = +inf for a cell without a step).
//   Epilogue cost per output: the fma that forms u from the accumulator, cvt_flr, med3, shift-add, one ds_read_b64, v_cmp, v_cndmask
//   (SDWA: selects the byte AND packs it into the output dword) -- 5 VALU after that fma, against ~22 for the A&S polynomial + margin test.
//   The table is built on the device by an exhaustive sweep over EVERY finite fp32 in real-line order with the fp64 erfc
//   (k_gelu_tab_sweep): nothing about monotonicity or step spacing is assumed, it is checked.
// ---------------------------------------------------------------------------------------------------
// u = y * k (exact: k is a power of two).  The epilogues get u straight from the accumulator - fma(acc, colscale * k, bias * k) IS
// RN(acc * colscale + bias) * k - and never form y itself (round 4: one instruction per output less); cell = clamp(floor(u) + off), the
// thresholds are stored times k.  Returns the byte offset of the entry RELATIVE TO ENTRY `off` (the callers fold off * 8 into the base);
// lo = -off, hi = cells - 1 - off as floats: u is clamped to [lo, hi + 0.5] first (one v_med3_f32; a NaN becomes lo), then floored.
__device__ __forceinline__ int gelu_tab_offset(float u, float lo, float hi_half) {
  int f;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(f) : "v"(__builtin_amdgcn_fmed3f(u, lo, hi_half)));
  return f << 3;
}
__device__ __forceinline__ int gelu_code_exact(float y, float inv_s) {
  const float r = rintf(gelu_exact(y) * inv_s);
  return (int)clamp8f(r);
}
// byte B of d := (u >= thr) ? hi : lo   with e = {bits of thr * k, lo | hi << 8}; the other bytes of d are kept (B > 0) / zeroed (B = 0)
#define P2V_GELU_SEL(B, UNUSED, DST, YV, ENT)                                                                             \
  asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:BYTE_" #B " dst_unused:" UNUSED              \
      " src0_sel:BYTE_0 src1_sel:BYTE_1"                                                                                   \
      : "+v"(DST) : "v"(YV), "v"(__uint_as_float(ENT.x)), "v"(ENT.y) : "vcc")
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// eight outputs of one lane (two groups of four) -> two dwords of int8 codes: all eight table reads are requested before the first
// select waits for one.  u0 / u1: the scaled pre-activations y * k;  tabz: entry `off` of the table (cell of u in [0, 1))
__device__ __forceinline__ void gelu_tab_q8x8(const float (&u0)[4], const float (&u1)[4], const unsigned char* tabz, float lo, float hi,
                                              unsigned& d0, unsigned& d1) {
  uint2 e0[4], e1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) e0[i] = *reinterpret_cast<const uint2*>(tabz + gelu_tab_offset(u0[i], lo, hi));
#pragma unroll
  for (int i = 0; i < 4; ++i) e1[i] = *reinterpret_cast<const uint2*>(tabz + gelu_tab_offset(u1[i], lo, hi));
  d0 = 0;
  d1 = 0;
  P2V_GELU_SEL(0, "UNUSED_PAD", d0, u0[0], e0[0]);
  P2V_GELU_SEL(1, "UNUSED_PRESERVE", d0, u0[1], e0[1]);
  P2V_GELU_SEL(2, "UNUSED_PRESERVE", d0, u0[2], e0[2]);
  P2V_GELU_SEL(3, "UNUSED_PRESERVE", d0, u0[3], e0[3]);
  P2V_GELU_SEL(0, "UNUSED_PAD", d1, u1[0], e1[0]);
  P2V_GELU_SEL(1, "UNUSED_PRESERVE", d1, u1[1], e1[1]);
  P2V_GELU_SEL(2, "UNUSED_PRESERVE", d1, u1[2], e1[2]);
  P2V_GELU_SEL(3, "UNUSED_PRESERVE", d1, u1[3], e1[3]);
}

// n-th finite fp32 in real-line order: n in [0, 2F), F = 0x7F800000 (negative values by falling magnitude, -0, +0, positives)
#define P2V_F32_FINITE 0x7F800000ull
__device__ __forceinline__ float f32_in_order(unsigned long long n) {
  return __uint_as_float(n < P2V_F32_FINITE ? 0x80000000u | (unsigned)(P2V_F32_FINITE - 1 - n) : (unsigned)(n - P2V_F32_FINITE));
}

// clamp(rne(x / s), -128, 127) with IEEE-division semantics (the reference divides by the non-power-of-two
// PTF scales, ptf.py:133) at the price of one multiply: t = x * fl(1/s) is within 2^-23 |t| of the true
// quotient and fl(x/s) within 2^-24 |t|; below |t| = 256 that is < 5e-5, so when t is further than 1e-4 from a
// rounding boundary both round to the same integer; above 256 both clamp.  Otherwise (2e-4 of the lanes) divide.
__device__ __forceinline__ float div_q8f(float x, float s, float rs) {
  const float t = x * rs;
  float r = rintf(t);
  const bool slow = !rne_decided(t, r, 1.0e-4f);
  if (__builtin_amdgcn_ballot_w64(slow) != 0) {
    if (slow) r = rintf(x / s);
  }
  return clamp8f(r);
}
template <bool CLAMP>
__device__ __forceinline__ void div_q8fx4(const float (&x)[4], const float (&s)[4], const float (&rs)[4], float (&out)[4]) {
  float r[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = x[i] * rs[i];
    r[i] = rintf(t);
    dv[i] = t - r[i];
  }
  // ONE test for the four values: max |t - r| (two v_max3 with |.| modifiers) against the margin
  const float dmax = fmaxf(fmaxf(fmaxf(fabsf(dv[0]), fabsf(dv[1])), fabsf(dv[2])), fabsf(dv[3]));
  if (__builtin_amdgcn_ballot_w64(!(dmax < 0.5f - 1.0e-4f)) != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (!(fabsf(dv[i]) < 0.5f - 1.0e-4f)) r[i] = rintf(x[i] / s[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = CLAMP ? clamp8f(r[i]) : r[i];   // unclamped when the caller packs (saturating)
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef short v2i16 __attribute__((ext_vector_type(2)));

#define CHECK_LAUNCH()                                     \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return (int)e_;                  \
  } while (0)
