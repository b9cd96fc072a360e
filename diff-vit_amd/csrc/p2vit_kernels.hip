// p2vit_kernels.hip -- CDNA4 (gfx950) kernels of the PoT-PTQ quantized ViT forward.
//
// Built with -ffp-contract=off: every fp32 epilogue below reproduces the reference's eager fp32
// operation order (one IEEE rounding per torch op), so no mul+add may be fused behind our back.
// `/` and sqrtf are the correctly-rounded forms (hipcc default), rintf is v_rndne_f32 (half-to-even,
// = torch.round).
//
// Data layout in HBM: activations are int8 codes, row-major [batch*tokens][channels]; weights are int8
// codes [n_pad][k_pad] (K contiguous) -- both GEMM operands are K-contiguous, which is exactly the
// v_mfma_i32_32x32x32_i8 fragment shape (lane l: row l&31, 16 consecutive k-bytes at 16*(l>>5)).
//
// Reference citations are relative to /root/reference (LeSN-Lab/diff-ViT).
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
#ifdef P2V_EXP_NOTRIM   /* A/B baseline builds only (tools/exp): the round-2 forms of the round-3 instruction trims */
__device__ __forceinline__ float clamp8f(float r) { return fminf(fmaxf(r, -128.f), 127.f); }
#else
__device__ __forceinline__ float clamp8f(float r) { return __builtin_amdgcn_fmed3f(r, -128.f, 127.f); }
#endif
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
#if defined(P2V_EXP_NOTRIM2) || defined(P2V_EXP_OLDPACK)   /* A/B baseline builds only (tools/exp) */
__device__ __forceinline__ float pre_pack(float o) { return rintf(o); }
__device__ __forceinline__ unsigned pack4_pre(float r0, float r1, float r2, float r3) { return pack4_sat(r0, r1, r2, r3); }
#else
__device__ __forceinline__ float pre_pack(float o) { return clamp8f(o); }
__device__ __forceinline__ unsigned pack4_pre(float c0, float c1, float c2, float c3) {       // c = pre_pack(o)
  const float magic = 12582912.f;
  unsigned w = __float_as_uint(c0 + magic);                           // byte 0; bytes 1-3 are overwritten below
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c1), "v"(magic));
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c2), "v"(magic));
  asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(c3), "v"(magic));
  return w;
}
#endif
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
__device__ __noinline__ float gelu_exact(float y) {
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
//   cut into cells of width s/2 (k = 2/s, a power of two, so y*k is exact):  i = clamp(floor(fma(y, k, off)), 0, cells-1).
//   Steps of the monotone branch are >= s/1.13 apart, so a cell holds at most ONE step (the builder verifies this for every
//   cell, also around the minimum of GELU at y = -0.7518 where a down- and an up-step can come close); the entry is
//   { thr, lo | hi << 8 } and code = y >= thr ? hi : lo  (thr = +inf for a cell without a step).
//   Epilogue cost per output: fma, med3, cvt, shift, one ds_read_b64, v_cmp, v_cndmask (SDWA: selects the byte AND packs it
//   into the output dword) -- 6 VALU after the bias fma, against ~22 for the A&S polynomial + margin test.
//   The table is built on the device by an exhaustive sweep over EVERY finite fp32 in real-line order with the fp64 erfc
//   (k_gelu_tab_sweep): nothing about monotonicity or step spacing is assumed, it is checked.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned gelu_tab_offset(float y, float k, float off, float tmax) {
  const float t = __builtin_amdgcn_fmed3f(__builtin_fmaf(y, k, off), 0.f, tmax);    // y*k exact; NaN -> cell 0
  return (unsigned)t << 3;                                                          // v_cvt_u32_f32 truncates: floor for t >= 0
}
__device__ __forceinline__ int gelu_code_exact(float y, float inv_s) {
  const float r = rintf(gelu_exact(y) * inv_s);
  return (int)clamp8f(r);
}
// byte B of d := (y >= thr) ? hi : lo   with e = {thr bits, lo | hi << 8}; the other bytes of d are kept (B > 0) / zeroed (B = 0)
#define P2V_GELU_SEL(B, UNUSED, DST, YV, ENT)                                                                             \
  asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:BYTE_" #B " dst_unused:" UNUSED              \
      " src0_sel:BYTE_0 src1_sel:BYTE_1"                                                                                   \
      : "+v"(DST) : "v"(YV), "v"(__uint_as_float(ENT.x)), "v"(ENT.y) : "vcc")
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// eight outputs of one lane (two groups of four) -> two dwords of int8 codes: all eight table reads are requested before the first
// select waits for one
__device__ __forceinline__ void gelu_tab_q8x8(const float (&y0)[4], const float (&y1)[4], const unsigned char* tab, float k, float off, float tmax,
                                              unsigned& d0, unsigned& d1) {
  uint2 e0[4], e1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) e0[i] = *reinterpret_cast<const uint2*>(tab + gelu_tab_offset(y0[i], k, off, tmax));
#pragma unroll
  for (int i = 0; i < 4; ++i) e1[i] = *reinterpret_cast<const uint2*>(tab + gelu_tab_offset(y1[i], k, off, tmax));
  d0 = 0;
  d1 = 0;
  P2V_GELU_SEL(0, "UNUSED_PAD", d0, y0[0], e0[0]);
  P2V_GELU_SEL(1, "UNUSED_PRESERVE", d0, y0[1], e0[1]);
  P2V_GELU_SEL(2, "UNUSED_PRESERVE", d0, y0[2], e0[2]);
  P2V_GELU_SEL(3, "UNUSED_PRESERVE", d0, y0[3], e0[3]);
  P2V_GELU_SEL(0, "UNUSED_PAD", d1, y1[0], e1[0]);
  P2V_GELU_SEL(1, "UNUSED_PRESERVE", d1, y1[1], e1[1]);
  P2V_GELU_SEL(2, "UNUSED_PRESERVE", d1, y1[2], e1[2]);
  P2V_GELU_SEL(3, "UNUSED_PRESERVE", d1, y1[3], e1[3]);
}

// n-th finite fp32 in real-line order: n in [0, 2F), F = 0x7F800000 (negative values by falling magnitude, -0, +0, positives)
#define P2V_F32_FINITE 0x7F800000ull
__device__ __forceinline__ float f32_in_order(unsigned long long n) {
  return __uint_as_float(n < P2V_F32_FINITE ? 0x80000000u | (unsigned)(P2V_F32_FINITE - 1 - n) : (unsigned)(n - P2V_F32_FINITE));
}
// scratch: cnt[cells] | thr[cells] | lohi[cells] | first[cells]; first[] preset to 0xFFFFFFFF, cnt[] to 0
__global__ __launch_bounds__(256) void k_gelu_tab_sweep(float inv_s, float k, float off, float tmax, int cells, unsigned* scratch, int per_thread) {
  unsigned* cnt = scratch;
  unsigned* thr = scratch + cells;
  unsigned* lohi = scratch + 2 * cells;
  unsigned* first = scratch + 3 * cells;
  const unsigned long long n0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * per_thread;
  if (n0 >= 2 * P2V_F32_FINITE) return;
  int pc = 0;
  unsigned pi = 0xFFFFFFFFu;
  if (n0 > 0) {
    const float yp = f32_in_order(n0 - 1);
    pc = gelu_code_exact(yp, inv_s);
    pi = gelu_tab_offset(yp, k, off, tmax) >> 3;
  }
  for (int j = 0; j < per_thread; ++j) {
    const unsigned long long n = n0 + j;
    if (n >= 2 * P2V_F32_FINITE) break;
    const float y = f32_in_order(n);
    const int c = gelu_code_exact(y, inv_s);
    const unsigned i = gelu_tab_offset(y, k, off, tmax) >> 3;
    if (i != pi) first[i] = (unsigned)c & 255u;                 // first value of a cell: its code when the cell has no step
    if (n > 0 && c != pc) {
      atomicAdd(&cnt[i], 1u);
      thr[i] = __float_as_uint(y);
      lohi[i] = ((unsigned)pc & 255u) | (((unsigned)c & 255u) << 8);
    }
    pc = c;
    pi = i;
  }
}
// status: 0 ok, bit 0 = a cell with two steps, bit 1 = a cell no fp32 value maps to
__global__ void k_gelu_tab_finish(int cells, const unsigned* scratch, uint2* table, unsigned* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const unsigned c = scratch[i], f = scratch[3 * cells + i];
  if (c > 1) atomicOr(status, 1u);
  if (f > 255u) atomicOr(status, 2u);
  table[i] = c == 0 ? make_uint2(0x7F800000u, f | (f << 8)) : make_uint2(scratch[cells + i], scratch[2 * cells + i]);
}
// independent check: every finite fp32 through the epilogue's lookup against the fp64 evaluation
__global__ __launch_bounds__(256) void k_gelu_tab_check(float inv_s, float k, float off, float tmax, const unsigned char* table,
                                                        unsigned long long* mismatches) {
  unsigned long long bad = 0;
  for (unsigned long long n = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; n < 2 * P2V_F32_FINITE;
       n += (unsigned long long)gridDim.x * blockDim.x * 8) {
    float y[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i >> 2][i & 3] = f32_in_order(n + i < 2 * P2V_F32_FINITE ? n + i : n);
    unsigned d[2];
    gelu_tab_q8x8(y[0], y[1], table, k, off, tmax, d[0], d[1]);          // the lookup of the GEMM epilogues
#pragma unroll
    for (int i = 0; i < 8; ++i) bad += (sx8(d[i >> 2], i & 3) != gelu_code_exact(y[i >> 2][i & 3], inv_s)) ? 1 : 0;
  }
  if (bad) atomicAdd(mismatches, bad);
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

// ---------------------------------------------------------------------------------------------------
// K0: qact_input + im2col   (vit_fquant.py:705-706; layers_quant.py:467; layers.py:82-88)
// one thread = 4 consecutive pixels of one patch row -> one dword of the patch matrix.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quantize_patchify(const float* __restrict__ img, int B, int C, int H, int W,
                                                           int P, float inv_s, int8_t* __restrict__ out, int k_pad, int rows_per_block) {
  const int gw = W / P, gh = H / P;
  const int kq = k_pad >> 2;  // dwords per output row
  // the (channel, patch row, 4-pixel group) of a thread is fixed: decomposed once, not per element (the per-element 64-bit
  // div/mod chain of the first version cost ~60 instructions per pixel)
  const long long rows = (long long)B * gh * gw;
  for (int d = threadIdx.x; d < kq; d += (int)blockDim.x) {
    const int col = d * 4;
    const bool live = col < C * P * P;
    const int c = col / (P * P), rem = col % (P * P), i = rem / P, j = rem % P;
    const long long chan_off = ((long long)c * H + i) * W + j;
    long long row = (long long)blockIdx.x * rows_per_block;
    const long long row_end = row + rows_per_block < rows ? row + rows_per_block : rows;
    for (; row < row_end; ++row) {            // row -> (image, patch y, patch x): wave-uniform, scalar arithmetic
      const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
      unsigned v = 0;
      if (live) {
        const float4 f = *reinterpret_cast<const float4*>(img + (long long)b * C * H * W + chan_off + (long long)py * P * W + px * P);
        v = pack4_rne_sat(f.x * inv_s, f.y * inv_s, f.z * inv_s, f.w * inv_s);
      }
      *reinterpret_cast<unsigned*>(out + row * k_pad + col) = v;
    }
  }
}

// cls rows of the residual stream: constant per model (vit_fquant.py:718-733 applied to cls_token)
__global__ void k_fill_cls(int8_t* __restrict__ x, int B, int T, int D, const int8_t* __restrict__ cls) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * D) x[(long long)(i / D) * T * D + (i % D)] = cls[i % D];
}

// ---------------------------------------------------------------------------------------------------
// K1: int8 MFMA GEMM with fused fp32 epilogue.
//   D^T tile = W_tile (A operand, rows n) x X_tile^T (B operand, rows m): accumulator rows (registers)
//   run over output channels n, accumulator columns (lanes) over activation rows m, so every lane owns
//   4-byte groups of one output row and per-channel constants are plain float4 loads.
//   Block 128(m) x 128(n), 4 waves as 2x2 of 64x64, BK = 64 bytes, double-buffered LDS with the 16-byte
//   chunk index XOR-swizzled by (row>>2)&3 so ds_read_b128 of 32 rows x same chunk is conflict free.
// ---------------------------------------------------------------------------------------------------
#define GBM 128
#define GBN 128
#define GBK 64
#define P2V_EPI_GELU_TAB 5   // internal: P2V_EPI_GELU with a threshold table in LDS (p2v_epilogue.gelu.table != NULL)

__device__ __forceinline__ int lds_off64(int row, int chunk) { return row * GBK + ((chunk ^ ((row >> 2) & 3)) << 4); }

// per-block staging of the per-channel epilogue constants (read by every lane of the block)
struct EpiLds {
  float colscale[GBN], bias[GBN], s_mid[GBN], s_res[GBN], s_next[GBN], r_mid[GBN], r_next[GBN], m128_sres[GBN];   // m128_sres = -128 * s_res (exact)
};

template <int EPI>
__device__ __forceinline__ void gemm_stage_epilogue(EpiLds* e, int n0, int tid, const GemmArgs& g) {
  if (tid < GBN) {
    const int n = n0 + tid;
    // REQUANT: (acc*cs + b) * 2^e == acc*(cs*2^e) + b*2^e with the same single rounding (power-of-two scaling commutes with
    // rounding; the plan checks that 1/s_out is a power of two), so the multiply leaves the per-output chain
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : 1.0f;
    e->colscale[tid] = g.colscale[n] * fold * (g.w4 ? 0.0625f : 1.0f);   // arrays are padded to n_pad; packed int4: acc = 16 x sum
    e->bias[tid] = g.bias[n] * fold;
    const bool ok = n < g.N;
    if (EPI == P2V_EPI_RESID) {
      const float sm = ok ? g.ep.s_mid[n] : 1.f;
      e->s_mid[tid] = sm;
      e->r_mid[tid] = 1.0f / sm;
      const float srs = ok ? g.ep.s_res[n] : 1.f;
      e->s_res[tid] = srs;
      e->m128_sres[tid] = -128.f * srs;
    }
    if (EPI == P2V_EPI_RESID || EPI == P2V_EPI_EMBED) {
      const float sn = ok ? g.ep.s_next[n] : 1.f;
      e->s_next[tid] = sn;
      e->r_next[tid] = 1.0f / sn;
    }
  }
}

// EMBED and HEAD epilogues (one launch each per forward, k_gemm_i8); the per-block epilogues of the layer GEMMs are gemm_epilogue_tile2
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_tile(const v16i& acc, int m, int n_tile, int nl, int h, const GemmArgs& g, const EpiLds* e) {
  static_assert(EPI == P2V_EPI_EMBED || EPI == P2V_EPI_HEAD, "stem / head epilogue");
  // lane owns output row m, channels n_tile + 8*gq + 4*h + {0..3}, gq = 0..3   (C/D map of 32x32 MFMA);
  // nl = n_tile - n0 (column offset inside the block tile, for the LDS constants)
  const bool row_ok = m < g.M;
  unsigned d[4];
  long long out_row = m;
  int tok = 0;
  if (EPI == P2V_EPI_EMBED) {
    int b = m / g.ep.patches, p = m % g.ep.patches;
    tok = p + 1;
    out_row = (long long)b * (g.ep.patches + 1) + tok;
  }
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = n_tile + 8 * gq + 4 * h, c = nl + 8 * gq + 4 * h;
    const float4 cs = *reinterpret_cast<const float4*>(e->colscale + c);
    const float4 bs = *reinterpret_cast<const float4*>(e->bias + c);
    float y[4];
    // F.linear / F.conv2d on fake-quantised operands: exact integer sum * (s_x*s_w[n]), then ONE rounding
    // for the fp32 bias (layers.py:87,178).  The product int * 2^k is exact, so the fused multiply-add rounds
    // exactly once, like mul-then-add does.
    y[0] = __builtin_fmaf((float)acc[4 * gq + 0], cs.x, bs.x);
    y[1] = __builtin_fmaf((float)acc[4 * gq + 1], cs.y, bs.y);
    y[2] = __builtin_fmaf((float)acc[4 * gq + 2], cs.z, bs.z);
    y[3] = __builtin_fmaf((float)acc[4 * gq + 3], cs.w, bs.w);
    float q[4];                 // integral floats; the byte packing below saturates to [-128,127]
    if (EPI == P2V_EPI_EMBED) {
      const float4 sn = *reinterpret_cast<const float4*>(e->s_next + c);
      const float4 rn = *reinterpret_cast<const float4*>(e->r_next + c);
      float4 pe = make_float4(0, 0, 0, 0);
      if (n < g.N) pe = *reinterpret_cast<const float4*>(g.ep.pos_deq + (long long)tok * g.N + n);
      const float snv[4] = {sn.x, sn.y, sn.z, sn.w}, pev[4] = {pe.x, pe.y, pe.z, pe.w}, rnv[4] = {rn.x, rn.y, rn.z, rn.w};
      float xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float q1 = sat8f(y[i] * g.ep.inv_s_pe);               // PatchEmbed.qact
        const float q2 = sat8f(q1 * g.ep.pe_to_embed);              // qact_embed (both PoT: exact ratio)
        xv[i] = __builtin_fmaf(q2, g.ep.s_embed, pev[i]);           // + qact_pos(pos_embed); int*2^k exact -> one rounding
      }
      div_q8fx4<false>(xv, snv, rnv, q);                            // qact1 (PTF)
    } else {  // HEAD: logits fp32 on the act_out grid
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q[i] = sat8f(y[i] * g.ep.inv_s_out);
        if (row_ok && n + i < g.N) {
          reinterpret_cast<float*>(g.out)[(long long)m * g.ldo + n + i] = q[i] * g.ep.s_out;
          if (g.out_codes) g.out_codes[(long long)m * g.ldo + n + i] = (int8_t)(int)q[i];
        }
      }
    }
    d[gq] = pack4_sat(q[0], q[1], q[2], q[3]);
    if (EPI == P2V_EPI_EMBED) __builtin_amdgcn_sched_barrier(0);   // keep the constant reads of the next group from being hoisted (register pressure)
  }
  if (EPI != P2V_EPI_HEAD) {
    uint4 o = halves_to_row16(d[0], d[1], d[2], d[3]);
    if (row_ok && n_tile + 16 * h < g.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(g.out) + out_row * g.ldo + n_tile + 16 * h) = o;
  }
}

// The same epilogue for the TWO 32-row blocks a wave owns under one 32-column group (rows m_first + l31 and m_first + 32 + l31): the
// per-channel constants depend on the columns only, so they are read from LDS once per 4-channel group and used for both blocks
// (half the LDS reads and half the exposed read latencies of two gemm_epilogue_tile calls).  REQUANT / GELU / GELU_TAB / RESID.
// LEAN: no look-ahead of the per-channel constants (24 registers in the RESID form): the 8-wave 256-row tile must stay within 128
// VGPRs and has four waves per SIMD to cover the LDS round trip instead
template <int EPI, bool LEAN = false>
__device__ __forceinline__ void gemm_epilogue_tile2(const v16i (&acc)[2], int m_first, int n_tile, int nl, int h, const GemmArgs& g,
                                                    const EpiLds* e, const uint4 (&resv)[2], const unsigned char* gtab = nullptr) {
  static_assert(EPI == P2V_EPI_REQUANT || EPI == P2V_EPI_GELU || EPI == P2V_EPI_GELU_TAB || EPI == P2V_EPI_RESID, "row-pair epilogue");
  unsigned d[2][4], res[2][4];
  const bool row_ok[2] = {m_first < g.M, m_first + 32 < g.M};
  if (EPI == P2V_EPI_RESID) {
    row16_to_halves(resv[0], res[0][0], res[0][1], res[0][2], res[0][3]);
    row16_to_halves(resv[1], res[1][0], res[1][1], res[1][2], res[1][3]);
  }
  // the constants of group gq + 1 are requested before group gq is computed (their LDS latency hides behind ~130 VALU instructions)
  struct Consts { float4 cs, bs, sm, sr, sn, rm, rn, mr; };
  auto load_consts = [&](int gq) {
    const int c = nl + 8 * gq + 4 * h;
    Consts k;
    k.cs = *reinterpret_cast<const float4*>(e->colscale + c);
    k.bs = *reinterpret_cast<const float4*>(e->bias + c);
    if (EPI == P2V_EPI_RESID) {
      k.sm = *reinterpret_cast<const float4*>(e->s_mid + c); k.sr = *reinterpret_cast<const float4*>(e->s_res + c);
      k.sn = *reinterpret_cast<const float4*>(e->s_next + c); k.rm = *reinterpret_cast<const float4*>(e->r_mid + c);
      k.rn = *reinterpret_cast<const float4*>(e->r_next + c);
      if (!LEAN) k.mr = *reinterpret_cast<const float4*>(e->m128_sres + c);
    }
    return k;
  };
  Consts knext;
  if (!LEAN) knext = load_consts(0);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = n_tile + 8 * gq + 4 * h;
    const Consts k = LEAN ? load_consts(gq) : knext;
    if (!LEAN && gq < 3) knext = load_consts(gq + 1);
    const float4 cs = k.cs, bs = k.bs;
    float smv[4], srv[4], snv[4], rmv[4], rnv[4], mrv[4];
    if (EPI == P2V_EPI_RESID) {
      smv[0] = k.sm.x; smv[1] = k.sm.y; smv[2] = k.sm.z; smv[3] = k.sm.w;
      srv[0] = k.sr.x; srv[1] = k.sr.y; srv[2] = k.sr.z; srv[3] = k.sr.w;
      snv[0] = k.sn.x; snv[1] = k.sn.y; snv[2] = k.sn.z; snv[3] = k.sn.w;
      rmv[0] = k.rm.x; rmv[1] = k.rm.y; rmv[2] = k.rm.z; rmv[3] = k.rm.w;
      rnv[0] = k.rn.x; rnv[1] = k.rn.y; rnv[2] = k.rn.z; rnv[3] = k.rn.w;
      if (LEAN) {           // -128 * s_res on the fly (exact): four registers less
#pragma unroll
        for (int i = 0; i < 4; ++i) mrv[i] = -128.f * srv[i];
      } else {
        mrv[0] = k.mr.x; mrv[1] = k.mr.y; mrv[2] = k.mr.z; mrv[3] = k.mr.w;
      }
    }
    float yy[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      yy[b][0] = __builtin_fmaf((float)acc[b][4 * gq + 0], cs.x, bs.x);      // one rounding, see gemm_epilogue_tile
      yy[b][1] = __builtin_fmaf((float)acc[b][4 * gq + 1], cs.y, bs.y);
      yy[b][2] = __builtin_fmaf((float)acc[b][4 * gq + 2], cs.z, bs.z);
      yy[b][3] = __builtin_fmaf((float)acc[b][4 * gq + 3], cs.w, bs.w);
      if (EPI != P2V_EPI_RESID && g.ep.tap_out && row_ok[b] && n < g.N) {
        const float un = EPI == P2V_EPI_REQUANT ? 1.0f / g.ep.inv_s_out : 1.0f;
        *reinterpret_cast<float4*>(g.ep.tap_out + (long long)(m_first + 32 * b) * g.N + n) =
            make_float4(yy[b][0] * un, yy[b][1] * un, yy[b][2] * un, yy[b][3] * un);
      }
    }
    if (EPI == P2V_EPI_GELU_TAB) {
#ifdef P2V_EXP_DUMMY_MFMA   /* experiment only: P2V_EXP_DUMMY_MFMA extra v_mfma_i32_32x32x32_i8 on live (random) data per 8 outputs */
      {
        v16i dacc_ = acc[0];
        const v4i da_ = {__float_as_int(yy[0][0]), __float_as_int(yy[0][1]), __float_as_int(yy[0][2]), __float_as_int(yy[0][3])};
        const v4i db_ = {__float_as_int(yy[1][0]), __float_as_int(yy[1][1]), __float_as_int(yy[1][2]), __float_as_int(yy[1][3])};
#pragma unroll
        for (int q_ = 0; q_ < P2V_EXP_DUMMY_MFMA; ++q_) dacc_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(da_, db_, dacc_, 0, 0, 0);
        asm volatile("" :: "v"(dacc_));
      }
#endif
#ifdef P2V_EXP_DUMMY_LDS    /* experiment only: P2V_EXP_DUMMY_LDS extra ds_read_b128 (conflict-free, consecutive lanes) per 8 outputs */
      {
        v4i dl_ = {0, 0, 0, 0};
#pragma unroll
        for (int q_ = 0; q_ < P2V_EXP_DUMMY_LDS; ++q_) {
          v4i t_;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t_) : "v"((unsigned)((threadIdx.x & 63) * 16)), "i"(q_ * 1024));
          dl_ ^= t_;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(dl_));
      }
#endif
#ifdef P2V_EXP_DUMMY   /* experiment only (never in the product build): P2V_EXP_DUMMY independent VALU instructions per 8 outputs */
      {
#ifdef P2V_EXP_DUMMY_PK   /* the same number of instructions, packed: twice the lane operations */
        v2f dp_ = {yy[0][0], yy[0][0]};
#pragma unroll
        for (int q_ = 0; q_ < P2V_EXP_DUMMY; ++q_) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(dp_) : "v"((v2f){yy[0][q_ & 3], yy[1][q_ & 3]}));
        asm volatile("" :: "v"(dp_));
#else
        float dm_ = yy[0][0];
#pragma unroll
        for (int q_ = 0; q_ < P2V_EXP_DUMMY; ++q_) asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(dm_) : "v"(yy[0][q_ & 3]));
        asm volatile("" :: "v"(dm_));
#endif
      }
#endif
      gelu_tab_q8x8(yy[0], yy[1], gtab, g.ep.gelu.k, g.ep.gelu.off, (float)(g.ep.gelu.cells - 1), d[0][gq], d[1][gq]);
      continue;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float (&y)[4] = yy[b];
      float q[4];
      if (EPI == P2V_EPI_REQUANT) {
        d[b][gq] = pack4_rne_sat(y[0], y[1], y[2], y[3]);
        continue;
      } else if (EPI == P2V_EPI_GELU) {
        gelu_q8x4(y, g.ep.inv_s_out, q);
      } else {   // RESID, see gemm_epilogue_tile
        float q3[4], xs[4];
        div_q8fx4<true>(y, smv, rmv, q3);
        const unsigned ru = res[b][gq] ^ 0x80808080u;
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = __builtin_fmaf((float)((ru >> (8 * i)) & 255u), srv[i], mrv[i]) + q3[i] * smv[i];
        div_q8fx4<false>(xs, snv, rnv, q);
      }
      d[b][gq] = pack4_sat(q[0], q[1], q[2], q[3]);
    }
    if (EPI == P2V_EPI_RESID) __builtin_amdgcn_sched_barrier(0);   // keep the constant reads of the next group from being hoisted (register pressure)
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const uint4 o = halves_to_row16(d[b][0], d[b][1], d[b][2], d[b][3]);
    if (row_ok[b] && n_tile + 16 * h < g.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(g.out) + (long long)(m_first + 32 * b) * g.ldo + n_tile + 16 * h) = o;
  }
}

// one k-tile of MFMA work for a wave: 2 k-steps x (1 weight frag, 2 activation frags, 2 MFMAs)
template <bool W4>
__device__ __forceinline__ void gemm_compute_tile(const int8_t* cx, const int8_t* cw, int wm, int wn, int l31, int h, v16i (&acc)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    v4i fw;
    if (W4) {
      const uint2 p = *reinterpret_cast<const uint2*>(cw + lds_off_w4(wn * 32 + l31, 2 * ks + h));
      fw = unpack_w4(p.x, p.y);
    } else {
      fw = *reinterpret_cast<const v4i*>(cw + lds_off64(wn * 32 + l31, 2 * ks + h));
    }
    const v4i f0 = *reinterpret_cast<const v4i*>(cx + lds_off64(wm * 64 + l31, 2 * ks + h));
    const v4i f1 = *reinterpret_cast<const v4i*>(cx + lds_off64(wm * 64 + 32 + l31, 2 * ks + h));
    acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw, f0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw, f1, acc[1], 0, 0, 0);
  }
}

// Stem and head GEMM (EMBED / HEAD epilogues; one launch each per forward): 128x128 block tile, 8 waves (2 along m x 4 along n,
// 64x32 each), <= 128 VGPRs -> 2 workgroups (16 waves) per CU.  Global->LDS staging goes through a 3-deep ring of NAMED registers
// (an indexed array of prefetch registers is placed in scratch by hipcc: measured), one barrier per k-tile.  The layer GEMMs run
// k_gemm_dma / k_ln_gemm.
template <int EPI, bool W4>
__global__ __launch_bounds__(512, 4) void k_gemm_i8(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) int8_t lds[2 * (GBM + GBN) * GBK + sizeof(EpiLds)];
  int8_t* sX = lds;                    // [2][GBM][GBK] activation rows
  int8_t* sW = lds + 2 * GBM * GBK;    // [2][GBN][GBK] weight rows
  EpiLds* sE = reinterpret_cast<EpiLds*>(lds + 2 * (GBM + GBN) * GBK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware tile order: each XCD walks a contiguous range of tiles, n fastest, so the tiles that share
  // an activation panel hit the same L2.
  int bid = blockIdx.x, nt = gridDim.x, xcd = bid & 7, qd = nt >> 3, rm = nt & 7;
  int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tn = t % g.tiles_n, tm = t / g.tiles_n;
  const int m0 = tm * GBM, n0 = tn * GBN;

  const int lrow = tid >> 2, lchunk = tid & 3;      // 512 threads: one 16-byte chunk of each operand per k-tile
  int mr0 = m0 + lrow;
  mr0 = mr0 < g.M ? mr0 : g.M - 1;
  const int8_t* gx0 = g.A + (long long)mr0 * g.lda + lchunk * 16;
  // packed int4: the W tile of k-tile T is the contiguous 4 KB block (tn * nk + T), an LDS image: threads 0..255 copy 16 bytes each
  const int8_t* gw0 = W4 ? g.W + (long long)tn * (g.K / GBK) * 4096 + (tid & 255) * 16 : g.W + (long long)(n0 + lrow) * g.K + lchunk * 16;
  const int o0 = lds_off64(lrow, lchunk);

  v16i acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0;

  const int nk = g.K / GBK;
  uint4 ax0, aw0, bx0, bw0, cx0, cw0;
#define G_LOAD(P, T)                                                        \
  do {                                                                      \
    P##x0 = *reinterpret_cast<const uint4*>(gx0 + (T) * GBK);               \
    P##w0 = *reinterpret_cast<const uint4*>(gw0 + (T) * (W4 ? 4096 : GBK));  \
  } while (0)
#define G_STEP(P, T)                                                        \
  do {                                                                      \
    int8_t* bx_ = sX + ((T) & 1) * GBM * GBK;                               \
    int8_t* bw_ = sW + ((T) & 1) * GBN * GBK;                               \
    *reinterpret_cast<uint4*>(bx_ + o0) = P##x0;                            \
    if (!W4) *reinterpret_cast<uint4*>(bw_ + o0) = P##w0;                   \
    else if (tid < 256) *reinterpret_cast<uint4*>(bw_ + tid * 16) = P##w0;  \
    __syncthreads();                                                        \
    if ((T) + 3 < nk) G_LOAD(P, (T) + 3);                                   \
    gemm_compute_tile<W4>(bx_, bw_, wm, wn, l31, h, acc);                   \
  } while (0)
  G_LOAD(a, 0);
  if (nk > 1) G_LOAD(b, 1);
  if (nk > 2) G_LOAD(c, 2);
  gemm_stage_epilogue<EPI>(sE, n0, tid, g);        // visible after the first barrier of the k loop
  for (int kt = 0; kt < nk; kt += 3) {
    G_STEP(a, kt);
    if (kt + 1 < nk) G_STEP(b, kt + 1);
    if (kt + 2 < nk) G_STEP(c, kt + 2);
  }
#undef G_LOAD
#undef G_STEP
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
    gemm_epilogue_tile<EPI>(acc[mi], m0 + wm * 64 + mi * 32 + l31, n0 + wn * 32, wn * 32, h, g, sE);
}

// ---------------------------------------------------------------------------------------------------
// K1f: patch embedding of an UN-quantised image (VisionTransformer(input_quant=False): the reference's vit_large factory,
//   vit_fquant.py:925, 705-706): the fp32 pixels go straight into the QConv2d, whose weights are fake-quantised (layers.py:82-88):
//   y = F.conv2d(x, code_w * s_w, bias).  Not an integer contraction - canonical reading: the sum of the
//   products x * code_w in fp64 (every product is exact there; 24 + 8 bits), times the power-of-two s_w, plus the bias, rounded to fp32
//   ONCE; then the EMBED chain of gemm_epilogue_tile.  One launch per forward (39.5 G fp64 FMAs per 256 ViT-L images, ~2 % of the
//   step); 64 x 64 output tile, 4 x 4 outputs per thread, k-tiles of 16 straight from the image (im2col folded into the addressing).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_fp32(const float* __restrict__ img, int B, int Cin, int H, int Wd, int P, GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sX[16][68], sW[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int gw = Wd / P, gh = H / P, patches = gw * gh;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int K = Cin * P * P;
  // loader role: row lr of the tile, 4 consecutive k
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  int mrow = m0 + lr;
  mrow = mrow < g.M ? mrow : g.M - 1;
  const int b_ = mrow / patches, pr = mrow % patches, py = pr / gw, px = pr % gw;
  const float* ibase = img + (long long)b_ * Cin * H * Wd + (long long)py * P * Wd + px * P;
  const int8_t* wbase = g.W + (long long)(n0 + lr) * g.K + lk;           // rows padded to n_pad, zero beyond N
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int k = k0 + lk;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned wv = 0;
    if (k < K) {                                                          // K is a multiple of 4 (patch_size % 4 == 0)
      const int c = k / (P * P), rem = k % (P * P), i = rem / P, j = rem % P;
      xv = *reinterpret_cast<const float4*>(ibase + ((long long)c * H + i) * Wd + j);
      wv = *reinterpret_cast<const unsigned*>(wbase + k0);
    }
    __syncthreads();
    sX[lk + 0][lr] = xv.x; sX[lk + 1][lr] = xv.y; sX[lk + 2][lr] = xv.z; sX[lk + 3][lr] = xv.w;
    sW[lk + 0][lr] = (float)sx8(wv, 0); sW[lk + 1][lr] = (float)sx8(wv, 1); sW[lk + 2][lr] = (float)sx8(wv, 2); sW[lk + 3][lr] = (float)sx8(wv, 3);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 xa = *reinterpret_cast<const float4*>(&sX[kk][ty * 4]);
      const float4 wa = *reinterpret_cast<const float4*>(&sW[kk][tx * 4]);
      const double xd[4] = {(double)xa.x, (double)xa.y, (double)xa.z, (double)xa.w};
      const double wd[4] = {(double)wa.x, (double)wa.y, (double)wa.z, (double)wa.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(xd[i], wd[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
    const int bb = m / patches, tok = m % patches + 1;
    const long long out_row = (long long)bb * (patches + 1) + tok;
    const int n = n0 + tx * 4;
    if (n >= g.N) continue;                                               // N is a multiple of 4
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float y = (float)__builtin_fma(acc[i][j], (double)g.colscale[n + j], (double)g.bias[n + j]);   // ONE rounding
      const float q1 = sat8f(y * g.ep.inv_s_pe);                          // PatchEmbed.qact
      const float q2 = sat8f(q1 * g.ep.pe_to_embed);                      // qact_embed (both PoT: exact ratio)
      const float xs = __builtin_fmaf(q2, g.ep.s_embed, g.ep.pos_deq[(long long)tok * g.N + n + j]);   // + qact_pos(pos_embed)
      q[j] = rintf(xs / g.ep.s_next[n + j]);                              // qact1 (PTF): IEEE division like the reference
    }
    *reinterpret_cast<unsigned*>(reinterpret_cast<int8_t*>(g.out) + out_row * g.ldo + n) = pack4_sat(q[0], q[1], q[2], q[3]);
  }
}

// ---------------------------------------------------------------------------------------------------
// K1d: the tiled GEMM with LDS-DMA staging (global_load_lds_dwordx4, gfx950).
//   Why: in the register-staged round-1 kernel (removed) every k-tile moved 16 KB global -> VGPR -> ds_write_b128 -> LDS.  ds_write_b128 sustains ~79 B/clk per
//   CU (13 cycles per wave-instruction), i.e. ~207 cycles of the CU's one LDS store path per workgroup and k-tile; with three
//   workgroups per CU that is ~620 cycles per round of k-tiles beside 768 cycles of MFMA and ~380 cycles of fragment reads on the
//   same LDS: the k-loop was bound by LDS, not by the matrix pipe (measured 1.1 k cycles per k-tile).  The DMA writes LDS without
//   passing through registers: no ds_write at all, and the 48 staging VGPRs of the 3-deep register ring are gone (one more wave
//   per SIMD).
//   Layout per stage: X tile [128][64] at +0, W tile [128][64] at +8192, both with the 16-byte chunk XOR swizzle of lds_off64.
//   A DMA wave-instruction fills 1 KB = 16 rows x 64 B linearly (lane l -> row l>>2, slot l&3), so the swizzle goes on the
//   per-lane SOURCE address: slot s of row r receives logical chunk s ^ ((r>>2)&3).
//   Synchronisation (NST = 3 stages, one barrier per k-tile): tile t+2 is requested right after the barrier of tile t, into the
//   stage tile t-1 was read from (every wave has passed barrier t only after finishing tile t-1).  A wave waits for ITS OWN
//   pieces of tile t with a counted s_waitcnt vmcnt(4) (the 4 younger requests of tile t+1 stay in flight), then joins the
//   barrier.  NST = 2: a second barrier after the reads of tile t guards the refill of its stage.
//   The fragment reads are inline-asm ds_read_b128: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see while a
//   DMA is pending (it cannot prove they do not alias), which would serialise the pipeline; __syncthreads() likewise drains vmcnt,
//   hence the raw s_barrier.
// ---------------------------------------------------------------------------------------------------
#define DMA_STAGE_BYTES (2 * GBM * GBK)     // 16 KB: X tile + W tile of the 128 x 128 form (a packed int4 W tile fills half of its 8 KB)
// packed int4 weights: the W fragments are 8-byte reads of the [128][32 B] tile image, widened in registers (unpack_w4)
template <int OFF, bool FIRST = false>      // FIRST: the tile's first k-tile starts the sums (C operand = the literal 0: no accumulator clearing)
__device__ __forceinline__ void gemm_compute_tile_dma_w4(unsigned aX0, unsigned aX1, unsigned aW0, unsigned aW1, v16i (&acc)[2][2]) {
  v4i x0a, x1a, x0b, x1b;
  v2u p0a, p1a, p0b, p1b;
  const unsigned bX0 = aX0 ^ 32u, bX1 = aX1 ^ 32u, bW0 = aW0 ^ 16u, bW1 = aW1 ^ 16u;     // k-step 1: chunk ^ 2
  asm volatile(
      "ds_read_b64 %0, %8 offset:%16\n\tds_read_b64 %1, %9 offset:%16\n\tds_read_b128 %2, %10 offset:%16\n\tds_read_b128 %3, %11 offset:%16\n\t"
      "ds_read_b64 %4, %12 offset:%16\n\tds_read_b64 %5, %13 offset:%16\n\tds_read_b128 %6, %14 offset:%16\n\tds_read_b128 %7, %15 offset:%16\n\t"
      "s_waitcnt lgkmcnt(4)"
      : "=&v"(p0a), "=&v"(p1a), "=&v"(x0a), "=&v"(x1a), "=&v"(p0b), "=&v"(p1b), "=&v"(x0b), "=&v"(x1b)
      : "v"(aW0), "v"(aW1), "v"(aX0), "v"(aX1), "v"(bW0), "v"(bW1), "v"(bX0), "v"(bX1), "i"(OFF)
      : "memory");
  const v4i w0a = unpack_w4(p0a[0], p0a[1]), w1a = unpack_w4(p1a[0], p1a[1]);
  acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0a, x0a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[0][0], 0, 0, 0);
  acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0a, x1a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[0][1], 0, 0, 0);
  acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1a, x0a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[1][0], 0, 0, 0);
  acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1a, x1a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[1][1], 0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0b), "+v"(p1b), "+v"(x0b), "+v"(x1b));
  const v4i w0b = unpack_w4(p0b[0], p0b[1]), w1b = unpack_w4(p1b[0], p1b[1]);
  acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0b, x0b, acc[0][0], 0, 0, 0);
  acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0b, x1b, acc[0][1], 0, 0, 0);
  acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1b, x0b, acc[1][0], 0, 0, 0);
  acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1b, x1b, acc[1][1], 0, 0, 0);
}
template <int OFF, bool FIRST = false>      // FIRST: the tile's first k-tile starts the sums (C operand = the literal 0: no accumulator clearing)
__device__ __forceinline__ void gemm_compute_tile_dma(unsigned aX0, unsigned aX1, unsigned aW0, unsigned aW1, v16i (&acc)[2][2]) {
  // a*: LDS byte addresses of this lane's fragment rows at k-step 0; k-step 1 is the same address with bit 5 flipped (chunk ^ 2)
  v4i x0a, x1a, w0a, w1a, x0b, x1b, w0b, w1b;
  const unsigned bX0 = aX0 ^ 32u, bX1 = aX1 ^ 32u, bW0 = aW0 ^ 32u, bW1 = aW1 ^ 32u;
  asm volatile(
      "ds_read_b128 %0, %8 offset:%16\n\tds_read_b128 %1, %9 offset:%16\n\tds_read_b128 %2, %10 offset:%16\n\tds_read_b128 %3, %11 offset:%16\n\t"
      "ds_read_b128 %4, %12 offset:%16\n\tds_read_b128 %5, %13 offset:%16\n\tds_read_b128 %6, %14 offset:%16\n\tds_read_b128 %7, %15 offset:%16\n\t"
      "s_waitcnt lgkmcnt(4)"
      : "=&v"(w0a), "=&v"(w1a), "=&v"(x0a), "=&v"(x1a), "=&v"(w0b), "=&v"(w1b), "=&v"(x0b), "=&v"(x1b)
      : "v"(aW0), "v"(aW1), "v"(aX0), "v"(aX1), "v"(bW0), "v"(bW1), "v"(bX0), "v"(bX1), "i"(OFF)
      : "memory");
  acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0a, x0a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[0][0], 0, 0, 0);
  acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0a, x1a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[0][1], 0, 0, 0);
  acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1a, x0a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[1][0], 0, 0, 0);
  acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1a, x1a, FIRST ? (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[1][1], 0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0b), "+v"(w1b), "+v"(x0b), "+v"(x1b));   // the k-step-1 fragments are ordered behind this wait
  acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0b, x0b, acc[0][0], 0, 0, 0);
  acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0b, x1b, acc[0][1], 0, 0, 0);
  acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1b, x0b, acc[1][0], 0, 0, 0);
  acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1b, x1b, acc[1][1], 0, 0, 0);
}

// MT = waves along m: 2 -> 128 x 128 tile, 4 waves (3 workgroups per CU); 4 -> 256 x 128 tile, 8 waves, 2 workgroups per CU (round 3):
// per k-tile 24 KB of operands feed 64 MFMAs instead of 16 KB feeding 32, i.e. 47 instead of 64 B/clk/CU of operand fetch at full
// MFMA rate against the ~49 B/clk the CU's L1 delivers - the launcher picks it when the grid still fills the chip.
#ifdef P2V_DIAG
#define GD_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (g.stamps && threadIdx.x == 0) g.stamps[(long long)blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter();  \
  } while (0)
#else
#define GD_STAMP(slot) do { } while (0)
#endif
template <int EPI, int NST, bool W4, int MT>
__global__ __launch_bounds__(128 * MT, MT == 4 ? 4 : (NST == 2 ? 4 : 3)) void k_gemm_dma(GemmArgs g) {
  constexpr int TBM = 64 * MT;                                   // tile rows
  constexpr int NTH = 128 * MT;                                  // threads
  constexpr int STAGE = (TBM + GBN) * GBK;                       // X tile + W tile (a packed int4 W tile fills half of its 8 KB)
  constexpr int EPI_BYTES = (EPI == P2V_EPI_RESID) ? (int)sizeof(EpiLds) : 2 * GBN * (int)sizeof(float);   // colscale + bias only
  __shared__ __attribute__((aligned(1024))) int8_t lds[NST * STAGE + EPI_BYTES];
  EpiLds* sE = reinterpret_cast<EpiLds*>(lds + NST * STAGE);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];   // GELU threshold table (cells * 8 bytes)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x, nt = gridDim.x, xcd = bid & 7, qd = nt >> 3, rm = nt & 7;
  int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tn = t % g.tiles_n, tm = t / g.tiles_n;
  const int m0 = tm * TBM, n0 = tn * GBN;

  // ---- DMA source addresses: wave w moves rows [32w, 32w+32) of the X tile (two 16-row pieces) and, of the 128-row W tile,
  //      rows [32w, 32w+32) (MT = 2: two pieces) / [16w, 16w+16) (MT = 4: one piece)
  const int lr = lane >> 2, pc = lane & 3;
  const int ra = 32 * wave + lr, rb = ra + 16;
  int mra = m0 + ra, mrb = m0 + rb;
  mra = mra < g.M ? mra : g.M - 1;
  mrb = mrb < g.M ? mrb : g.M - 1;
  // source = wave-uniform base (the matrix + the k offset: scalar registers) + this lane's 32-bit byte offset, the form the LDS-DMA load
  // takes as  saddr + zext(voffset): no 64-bit address arithmetic per k-tile (eight v_lshl_add_u64 per wave and k-tile before; the launcher
  // checks that both matrices stay below 4 GB)
  const unsigned gxa = (unsigned)mra * (unsigned)g.lda + ((pc ^ ((ra >> 2) & 3)) << 4);
  const unsigned gxb = (unsigned)mrb * (unsigned)g.lda + ((pc ^ ((rb >> 2) & 3)) << 4);
  const int wa = (MT == 4 ? 16 * wave : 32 * wave) + lr, wb = wa + 16;
  const unsigned gwa = (unsigned)(n0 + wa) * (unsigned)g.K + ((pc ^ ((wa >> 2) & 3)) << 4);
  const unsigned gwb = (unsigned)(n0 + wb) * (unsigned)g.K + ((pc ^ ((wb >> 2) & 3)) << 4);
  // packed int4: the W tile of k-tile kt is the contiguous 4 KB LDS image (tn * nk + kt): one coalesced 1 KB piece per wave (MT = 2);
  // with 8 waves each wave moves 512 bytes (its lower 32 lanes)
  const unsigned gw4 = (unsigned)tn * (unsigned)(g.K / GBK) * 4096u + (MT == 4 ? wave * 512 + (lane & 31) * 16 : wave * 1024 + lane * 16);
  auto dma = [&](int stage, int kt) {
    int8_t* dst = lds + stage * STAGE + wave * (32 * GBK);
    const int ko = kt * GBK;
    // written as assembly: hipcc folds  uniform + zext(lane offset)  back into 64-bit vector additions (two v_lshl_add_u64 per request);
    // the LDS destination of a request is M0 + 16 * lane
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define P2V_DMA16_(SBASE, VOFF, DST)                                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"   /* (one wait state between the M0 write and its use) */ \
                 :: "v"(VOFF), "s"(SBASE), "s"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(DST)) : "memory", "m0")
    const int8_t* xk = g.A + ko;                              // uniform
    const int8_t* wk = g.W + (W4 ? kt * 4096 : ko);
    P2V_DMA16_(xk, gxa, dst);
    P2V_DMA16_(xk, gxb, dst + 16 * GBK);
    if (W4) {
      if (MT == 4) {
        if (lane < 32) P2V_DMA16_(wk, gw4, lds + stage * STAGE + TBM * GBK + wave * 512);
      } else {
        P2V_DMA16_(wk, gw4, lds + stage * STAGE + TBM * GBK + wave * 1024);
      }
    } else if (MT == 4) {
      P2V_DMA16_(wk, gwa, lds + stage * STAGE + TBM * GBK + wave * (16 * GBK));
    } else {
      P2V_DMA16_(wk, gwa, dst + TBM * GBK);
      P2V_DMA16_(wk, gwb, dst + TBM * GBK + 16 * GBK);
    }
#undef P2V_DMA16_
#pragma clang diagnostic pop
  };
  const int nk = g.K / GBK;
  GD_STAMP(0);
  dma(0, 0);
  if (nk > 1) dma(1, 1);

  // ---- epilogue constants / GELU table (compiler-visible LDS stores: they may wait for the requests above, which the first
  //      k-tile needs anyway); residual codes requested early
  gemm_stage_epilogue<EPI>(sE, n0, tid, g);
  if (EPI == P2V_EPI_GELU_TAB)
    for (int i = tid; i < g.ep.gelu.cells; i += NTH)
      reinterpret_cast<uint2*>(dyn_lds)[i] = reinterpret_cast<const uint2*>(g.ep.gelu.table)[i];
  // MT = 2: all four 16-byte pieces a lane needs are requested before the k-loop; MT = 4 (128-VGPR budget, four waves per SIMD to
  // cover the latency): two pieces at a time, right before the column group that consumes them
  uint4 resv[2][2];
  auto load_resid = [&](int ni) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31, n = n0 + wn * 64 + ni * 32 + 16 * h;
      resv[ni][mi] = make_uint4(0, 0, 0, 0);
      if (m < g.M && n < g.N) resv[ni][mi] = *reinterpret_cast<const uint4*>(g.ep.residual + (long long)m * g.ldo + n);
    }
  };
  if (EPI == P2V_EPI_RESID && MT == 2) {
    load_resid(0);
    load_resid(1);
  }

  v16i acc[2][2];                                        // started by the first k-tile

  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int8_t*)lds;
  const unsigned aX0 = lbase + lds_off64(wm * 64 + l31, h), aX1 = lbase + lds_off64(wm * 64 + 32 + l31, h);
  const unsigned aW0 = lbase + TBM * GBK + (W4 ? lds_off_w4(wn * 64 + l31, h) : lds_off64(wn * 64 + l31, h));
  const unsigned aW1 = lbase + TBM * GBK + (W4 ? lds_off_w4(wn * 64 + 32 + l31, h) : lds_off64(wn * 64 + 32 + l31, h));
  constexpr int PCS = (W4 || MT == 4) ? 3 : 4;        // LDS-DMA requests of one wave per k-tile (exec-masked ones count as well)
  GD_STAMP(1);

  // one k-tile: own pieces landed (younger requests stay in flight) -> barrier -> refill the freed stage -> MFMAs
#define P2V_KTILE(S, KT) P2V_KTILE_(S, KT, false)
#define P2V_KTILE_(S, KT, FIRST)                                                                                             \
  do {                                                                                                               \
    if (NST == 3) {                                                                                                  \
      /* in flight behind tile KT: tile KT+1 (PCS requests of this wave) */                                          \
      if ((KT) + 1 < nk) { if (PCS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
      asm volatile("s_barrier" ::: "memory");    /* tile KT landed for everyone; everyone is done reading tile KT-1 */ \
      if ((KT) + 2 < nk) dma(((S) + 2) % 3, (KT) + 2);                                                               \
      if (W4) gemm_compute_tile_dma_w4<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                        \
      else gemm_compute_tile_dma<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                              \
    } else {                                                                                                         \
      if ((KT) + 1 < nk) { if (PCS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
      asm volatile("s_barrier" ::: "memory");                                                                        \
      if (W4) gemm_compute_tile_dma_w4<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                        \
      else gemm_compute_tile_dma<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                              \
      if ((KT) + 2 < nk) {                                                                                           \
        asm volatile("s_barrier" ::: "memory");                                                                      \
        dma((S), (KT) + 2);                                                                                          \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)
  // the first k-tile is peeled: its MFMAs start the sums from the literal 0 (64 accumulator registers are never cleared)
  if (NST == 3) {
    P2V_KTILE_(0, 0, true);
    if (1 < nk) P2V_KTILE(1, 1);
    if (2 < nk) P2V_KTILE(2, 2);
    for (int kt = 3; kt < nk; kt += 3) {
      P2V_KTILE(0, kt);
      if (kt + 1 < nk) P2V_KTILE(1, kt + 1);
      if (kt + 2 < nk) P2V_KTILE(2, kt + 2);
    }
  } else {
    P2V_KTILE_(0, 0, true);
    if (1 < nk) P2V_KTILE(1, 1);
    for (int kt = 2; kt < nk; kt += 2) {
      P2V_KTILE(0, kt);
      if (kt + 1 < nk) P2V_KTILE(1, kt + 1);
    }
  }
#undef P2V_KTILE_
#undef P2V_KTILE
#ifdef P2V_DIAG
  asm volatile("s_nop 0" :: "v"(acc[1][1][0]));
#endif
  GD_STAMP(2);
  __syncthreads();        // nothing is in flight any more; orders the constant stores before the epilogue reads for every wave
  GD_STAMP(3);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    if (EPI == P2V_EPI_RESID && MT == 4) load_resid(ni);              // (held across the other group's epilogue they would spill)
    gemm_epilogue_tile2<EPI, (MT == 4 && EPI == P2V_EPI_RESID)>(acc[ni], m0 + wm * 64 + l31, n0 + wn * 64 + ni * 32, wn * 64 + ni * 32, h, g, sE,
                                                                 resv[ni], dyn_lds);
  }
  GD_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------
// K2: integer LayerNorm (QIntLayerNorm mode 'int', layers.py:255-289) + /channel_scale + qact0 clamp
// (vit_fquant.py:284-289).  One row per 32-lane half wave (12 bytes/lane at C=384), LN_ROWS rows per half
// wave so the five per-channel constant vectors stay in registers.  sum x and sum x^2 are exact integers;
// everything after mirrors the reference's fp32 operation order.
// ---------------------------------------------------------------------------------------------------
// sum over the 32 lanes of a half wave, result in every lane: four DPP butterflies (no address registers, VALU rate) and
// one ds_swizzle for the distance-16 step.  After the xor-1/xor-2 steps the four lanes of a quad agree, so the mirrors
// of 8 and 16 lanes act as xor-4 and xor-8.
__device__ __forceinline__ int half_wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
  v += __builtin_amdgcn_ds_swizzle(v, 0x401F);                      // bitmask mode: lane ^ 16
  return v;
}

// generic per-element chain: every step of get_MN / the 'int' forward as written in the reference
// os > 0: the output scale itself - the two quotients are IEEE divisions like the reference's; os == 0: multiply by io = 1/scale
// (identical for powers of two)
__device__ __forceinline__ float ln_elem_generic(float xq, float g, float bta, float io, float pm, float rs, float mos, float os) {
  const float A = os > 0.f ? (rs * g) / os : (rs * g) * io;        // (s1/std)*gamma / out_scale
  const float absA = fabsf(A);
  int N = 134 - (int)(__float_as_uint(absA) >> 23);                // 7 - floor(log2|A|)   (get_MN, layers.py:234-238)
  N = N < 0 ? 0 : (N > 31 ? 31 : N);
  const float M = fminf(floorf(ldexpf(absA, N)), 255.f);           // floor(|A| * 2^N), clamped
  const float sM = copysignf(M, A);                                // A.sign() * M  (M == 0 when A == 0)
  const float tb = bta - mos * g;
  const float Bv = rintf(ldexpf(os > 0.f ? tb / os : tb * io, N)); // layers.py:283-286
  const float o = rintf(ldexpf(sM * xq + Bv, -N));                 // layers.py:288
  return rintf(o * pm);                                            // * out_scale / cs_next / s_next (clamped by the packing)
}

// Fast chain (used when 1/out_scale is a power of two for every channel, which is the P2-ViT case, and the row's
// multipliers are inside the unclamped range of get_MN).  With io = 2^e:  A = (rs*g)*io = rs*(g*io)  and
// (b - mos*g)*io = b*io - mos*(g*io)  with the same roundings, so g*io and b*io are folded once per workgroup (shared
// through LDS).  For 2^-24 <= |A| < 2^8:  N = 134 - exp(A) is unclamped and M = floor(|A| 2^N) in [128,255] is the top
// 8 significant bits of A, i.e.  sign*M*2^-N == A with the low 16 mantissa bits cleared =: T;  and
// rint(((sM*xq + Bv) rounded) * 2^-N) == rint(fma(T, xq, Bv*2^-N))  because T*xq is exact (8 x 11 bits) and scaling by
// 2^-N commutes with the rounding.  Bit-identical to the generic chain (tests drive both through P2V_LN_GENERIC=1).
// LANES = 32: one row per half wave (C <= 1024);  LANES = 64: one row per wave (PatchMerging rows of up to 2048 channels)
// per-lane view of the folded per-channel constants of a LayerNorm (held in registers across rows)
// LDSC: post_mul and the PTF mask are re-read from the workgroup's LDS copy where they are used (one ds_read_b128 per four channels and
// row) instead of living in 8 * NCH registers - the stand-alone kernel, whose scratch stays valid, then fits two rows per batch (ln_rows)
// in the register budget of three waves per SIMD
template <int NCH, bool LDSC = false>
struct LnLane {
  bool on[NCH];
  float4 gm[NCH], bt[NCH];            // gamma*io, beta*io
  float4 pm[LDSC ? 1 : NCH];          // post_mul
  float4 mkf[LDSC ? 1 : NCH];         // PTF mask (in_scale / s1): 1, 2, 4 or 8
  const float* sPl;                   // LDSC: this lane's first four channels in the LDS copies, and the chunk stride in floats
  const float* sMl;
  int cstride;
  float gmin, gmax;                   // extreme |gamma*io| over all channels
  float bmax;                         // max |beta*io| over all channels (bound of the LayerNorm offset, see ln_row)
  bool pot;                           // 1/out_scale is a power of two for every channel and the fold is exact
  bool pm_one;                        // post_mul == 1 for every channel (norm1 of P2-ViT: out_scale / channel_scale / qact0 scale): no second requant
  __device__ __forceinline__ float4 post4(int i) const { return LDSC ? *reinterpret_cast<const float4*>(sPl + i * cstride) : pm[LDSC ? 0 : i]; }
  __device__ __forceinline__ float4 mask4(int i) const { return LDSC ? *reinterpret_cast<const float4*>(sMl + i * cstride) : mkf[LDSC ? 0 : i]; }
};

// Fold, test and publish the per-channel constants once per workgroup (every thread calls it; contains a barrier), then load this
// lane's channels: lane l of a row group owns channels (l + LANES*i)*4 .. +3.
template <int NCH, int LANES, class LL>
__device__ __forceinline__ void ln_prepare(const p2v_ln& ln, int C, bool force_generic, float* sG, float* sB, float* sP, int* sM,
                                           int tid, int nthreads, LL& L) {
  const int l32 = tid & (LANES - 1);
  int potf = force_generic ? 0 : 1, pm1 = 1;
  for (int t4 = tid; t4 < NCH * LANES; t4 += nthreads) {   // one thread per 4 channels: fold, test, and publish
    const int c = t4 * 4;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), b = g, io = make_float4(1.f, 1.f, 1.f, 1.f), pmv = g, mk = g;
    if (c < C) {
      g = *reinterpret_cast<const float4*>(ln.gamma + c);
      b = *reinterpret_cast<const float4*>(ln.beta + c);
      io = *reinterpret_cast<const float4*>(ln.inv_out + c);
      pmv = *reinterpret_cast<const float4*>(ln.post_mul + c);
      mk = *reinterpret_cast<const float4*>(ln.mask + c);
    }
    const float g4[4] = {g.x, g.y, g.z, g.w}, b4[4] = {b.x, b.y, b.z, b.w}, i4[4] = {io.x, io.y, io.z, io.w};
    if (c < C) pm1 &= (int)(pmv.x == 1.f) & (int)(pmv.y == 1.f) & (int)(pmv.z == 1.f) & (int)(pmv.w == 1.f);
    float go[4], bo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ib = __float_as_uint(i4[j]);
      const int p2 = (int)((ib & 0x807FFFFFu) == 0u) & (int)((ib >> 23) - 32u <= 190u);    // +2^e, far from under/overflow
      go[j] = g4[j] * i4[j];
      bo[j] = b4[j] * i4[j];
      const float ga = fabsf(go[j]), ba = fabsf(bo[j]);
      // the fold must be exact: no product may leave the normal range
      const int gok = (int)(g4[j] == 0.f) | ((int)(ga >= 1.0e-30f) & (int)(ga <= 1.0e30f));
      const int bok = (int)(b4[j] == 0.f) | ((int)(ba >= 1.0e-30f) & (int)(ba <= 1.0e30f));
      potf &= p2 & gok & bok;
    }
    *reinterpret_cast<float4*>(sG + c) = make_float4(go[0], go[1], go[2], go[3]);
    *reinterpret_cast<float4*>(sB + c) = make_float4(bo[0], bo[1], bo[2], bo[3]);
    *reinterpret_cast<float4*>(sP + c) = pmv;
    *reinterpret_cast<float4*>(sM + c) = mk;
  }
  L.pot = __syncthreads_and(potf) != 0;
  L.pm_one = __syncthreads_and(pm1) != 0;
#ifdef P2V_EXP_NOTRIM
  L.pm_one = false;
#endif
  // extreme |g io| over all channels (every row group covers all of them)
  float gmin = 3.0e38f, gmax = 0.f, bmax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (l32 + LANES * i) * 4;
    L.on[i] = c < C;
    const float4 gv = *reinterpret_cast<const float4*>(sG + c);
    L.gm[i] = gv;
    L.bt[i] = *reinterpret_cast<const float4*>(sB + c);
    if (L.on[i]) bmax = fmaxf(bmax, fmaxf(fmaxf(fabsf(L.bt[i].x), fabsf(L.bt[i].y)), fmaxf(fabsf(L.bt[i].z), fabsf(L.bt[i].w))));
    if constexpr (sizeof(L.pm) == sizeof(float4) * NCH) {          // (the register-resident form)
      L.pm[i] = *reinterpret_cast<const float4*>(sP + c);
      L.mkf[i] = *reinterpret_cast<const float4*>(sM + c);
    }
    const float lo = fminf(fminf(fabsf(gv.x), fabsf(gv.y)), fminf(fabsf(gv.z), fabsf(gv.w)));
    const float hi = fmaxf(fmaxf(fabsf(gv.x), fabsf(gv.y)), fmaxf(fabsf(gv.z), fabsf(gv.w)));
    gmin = fminf(gmin, L.on[i] ? lo : 3.0e38f);
    gmax = fmaxf(gmax, L.on[i] ? hi : 0.f);
  }
  L.sPl = sP + l32 * 4;
  L.sMl = reinterpret_cast<const float*>(sM) + l32 * 4;
  L.cstride = LANES * 4;
  {   // positive floats order like their bit patterns: integer min/max butterflies inside the half wave
    int lo = (int)__float_as_uint(gmin), hi = (int)__float_as_uint(gmax), bh = (int)__float_as_uint(bmax);
#define LN_MM(ctrl) lo = min(lo, __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xF, 0xF, false)); hi = max(hi, __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xF, 0xF, false)); \
                    bh = max(bh, __builtin_amdgcn_update_dpp(bh, bh, ctrl, 0xF, 0xF, false));
    LN_MM(0xB1) LN_MM(0x4E) LN_MM(0x141) LN_MM(0x140)
#undef LN_MM
    lo = min(lo, __builtin_amdgcn_ds_swizzle(lo, 0x401F));
    hi = max(hi, __builtin_amdgcn_ds_swizzle(hi, 0x401F));
    bh = max(bh, __builtin_amdgcn_ds_swizzle(bh, 0x401F));
    if (LANES == 64) {
      lo = min(lo, __shfl_xor(lo, 32));
      hi = max(hi, __shfl_xor(hi, 32));
      bh = max(bh, __shfl_xor(bh, 32));
    }
    L.gmin = __uint_as_float((unsigned)lo);
    L.gmax = __uint_as_float((unsigned)hi);
    L.bmax = __uint_as_float((unsigned)bh);
  }
}

// The LayerNorm of a row in four steps, shared by the one-row and the batched form below:
//   ln_sums   the lane's x_q = code * mask and its part of sum x_q, sum x_q^2
//   ln_reduce the sums over the row group (32 or 64 lanes), result in every lane
//   ln_scalars mean / std -> the two row scalars rs = s1 / std, mos = mean / std and the fast-chain test
//   ln_apply  the per-element chain with those scalars -> packed output codes
template <int NCH, class LL>
__device__ __forceinline__ void ln_sums(const unsigned (&wcur)[NCH], const LL& L, float (&xq)[NCH][4], int& S1, unsigned& S2) {
  S2 = 0;                                       // C * (128*8)^2 <= 2^31 for C <= 2048: exact in 32 unsigned bits
#if defined(P2V_EXP_NOTRIM2) || defined(P2V_EXP_OLDSUMS)
  S1 = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const unsigned w = wcur[i];
    const float4 mk_ = L.mask4(i);
    const int m4[4] = {(int)mk_.x, (int)mk_.y, (int)mk_.z, (int)mk_.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = __mul24(sx8(w, j), m4[j]);           // x_q * in_scale_mask  (layers.py:269-273); w == 0 past C
      xq[i][j] = (float)v;
      S1 += v;
      S2 += (unsigned)__mul24(v, v);
    }
  }
#else
  // the lane's partial sums in fp32: |x_q| <= 1024, so sum x_q of 32 values and sum x_q^2 of 16 values (<= 2^24) are exact - full-rate
  // add / fma instead of 24-bit multiplies and three-operand adds (profiles/r03_op_cost.txt); the cross-lane sums stay integers
  float S1p = 0.f, S2p = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const unsigned w = wcur[i];
    const float4 mk_ = L.mask4(i);
    const float m4[4] = {mk_.x, mk_.y, mk_.z, mk_.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[i][j] = (float)sx8(w, j) * m4[j];       // x_q * in_scale_mask  (layers.py:269-273), exact; w == 0 past C
    S1p += (xq[i][0] + xq[i][1]) + (xq[i][2] + xq[i][3]);                  // (short dependency chains: a wave may be alone on its SIMD)
    S2p += __builtin_fmaf(xq[i][1], xq[i][1], xq[i][0] * xq[i][0]) + __builtin_fmaf(xq[i][3], xq[i][3], xq[i][2] * xq[i][2]);
    if ((i & 3) == 3 || i == NCH - 1) {
      S2 += (unsigned)S2p;
      S2p = 0.f;
    }
  }
  S1 = (int)S1p;
#endif
}
template <int LANES>
__device__ __forceinline__ void ln_reduce(int& S1, unsigned& S2) {
  S1 = half_wave_sum(S1);
  S2 = (unsigned)half_wave_sum((int)S2);        // two's-complement adds: the unsigned total is exact
  if (LANES == 64) {
    S1 += __shfl_xor(S1, 32);
    S2 += (unsigned)__shfl_xor((int)S2, 32);
  }
}
template <int NCH, class LL>
__device__ __forceinline__ void ln_scalars(int S1, unsigned S2, const LL& L, const p2v_ln& ln, int C, float& rs, float& mos, bool& fast) {
  const float s1 = ln.s1;
  const float Cf = (float)C;
  const float s1oC = s1 / Cf;
  const float S1f = (float)S1, S2f = (float)S2;
  const float mean = (S1f / Cf) * s1;                                  // x_q.mean(-1) * in_scale1
  const float stdv = s1oC * sqrtf(Cf * S2f - S1f * S1f);               // layers.py:276-277
  rs = s1 / stdv;
  mos = mean / stdv;
  // |A| = RN(rs*|g io|) is monotone in |g io|: the two extreme channels bound every channel exactly
  // ... and the offset Bv = rint(t * 2^N) (t = beta*io - mos*gamma*io) is rounded by adding and subtracting 1.5 * 2^(23-N), which is
  // rint on the 2^-N grid (ties to even included) as long as |t| * 2^N < 2^22: |t| <= bmax + |mos| gmax and N <= 134 - exp(rs * gmin),
  // so one comparison per row against 2^(exp(rs*gmin) - 112) bounds every channel (1 % margin for the roundings of t itself)
  const float amin = rs * L.gmin;
  const float tlim = __uint_as_float((((__float_as_uint(amin) >> 23) & 255u) + 15u) << 23);       // 2^22 * 2^-(134 - e_min)
  fast = L.pot && amin >= 0x1p-24f && rs * L.gmax < 256.f && (L.bmax + fabsf(mos) * L.gmax) * 1.01f < tlim;
}
template <int NCH, int LANES, class LL>
__device__ __forceinline__ void ln_apply(const float (&xq)[NCH][4], const LL& L, const p2v_ln& ln, int l32, float rs, float mos, bool fast,
                                         unsigned (&outw)[NCH]) {
  if (fast) {
    auto chain = [&](auto PM1c) {
      constexpr bool PM1 = decltype(PM1c)::value;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const float g4[4] = {L.gm[i].x, L.gm[i].y, L.gm[i].z, L.gm[i].w}, b4[4] = {L.bt[i].x, L.bt[i].y, L.bt[i].z, L.bt[i].w};
        float p4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (!PM1) {
          const float4 pm_ = L.post4(i);
          p4[0] = pm_.x; p4[1] = pm_.y; p4[2] = pm_.z; p4[3] = pm_.w;
        }
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; j += 2) {   // two channels at a time: only the 3-source fma is packed (measured on gfx950, tools/ubench/valu_rate:
          // a wave alone on its SIMD issues a 2-source fp32 op every ~4.9 cycles, a v_pk_mul/add_f32 every ~13, a 3-source v_fma_f32 every ~8, v_pk_fma_f32 ~13)
          v2f T2, Bq2;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float A = rs * g4[j + e];
            const float t = b4[j + e] - mos * g4[j + e];
            const unsigned Ab = __float_as_uint(A);
            T2[e] = __uint_as_float(Ab & 0xFFFF0000u);                              // sign * M * 2^-N
#ifdef P2V_EXP_NOTRIM
            const int N = 134 - (int)((Ab >> 23) & 255u);                           // in [0, 31] by the range test
            Bq2[e] = ldexpf(rintf(ldexpf(t, N)), -N);                               // Bv * 2^-N
#else
            // Bv * 2^-N = t rounded to the 2^-N grid, N = 134 - exp(A): C = 1.5 * 2^(23-N) has the exponent field exp(A) + 16 (4 full-rate
            // instructions instead of bfe, add, sub, ldexp, rndne, ldexp: tools/ubench/op_cost.hip, profiles/r03_op_cost.txt)
            const float Cm = __uint_as_float((Ab & 0x7F800000u) + 0x08400000u);
            Bq2[e] = (t + Cm) - Cm;
#endif
          }
          const v2f x2 = {xq[i][j], xq[i][j + 1]};
          const v2f o2 = __builtin_elementwise_fma(T2, x2, Bq2);
          if (PM1) {                                                                // out * 1: the LayerNorm output IS the code
            q[j] = o2[0];
            q[j + 1] = o2[1];
          } else {
            q[j] = rintf(o2[0]) * p4[j];
            q[j + 1] = rintf(o2[1]) * p4[j + 1];
          }
        }
        outw[i] = pack4_rne_sat(q[0], q[1], q[2], q[3]);                            // the (last) rounding is the packing's
      }
    };
    if (L.pm_one) chain(std::integral_constant<bool, true>{});      // wave-uniform
    else chain(std::integral_constant<bool, false>{});
  } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = L.on[i] ? (l32 + LANES * i) * 4 : 0;
      const float4 gv = *reinterpret_cast<const float4*>(ln.gamma + cc), bv = *reinterpret_cast<const float4*>(ln.beta + cc);
      const float4 iv = *reinterpret_cast<const float4*>(ln.inv_out + cc), pv = *reinterpret_cast<const float4*>(ln.post_mul + cc);
      const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
      const float i4[4] = {iv.x, iv.y, iv.z, iv.w}, p4[4] = {pv.x, pv.y, pv.z, pv.w};
      float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ln.out_scale) ov = *reinterpret_cast<const float4*>(ln.out_scale + cc);
      const float o4[4] = {ov.x, ov.y, ov.z, ov.w};
      float q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = ln_elem_generic(xq[i][j], g4[j], b4[j], i4[j], p4[j], rs, mos, o4[j]);
      outw[i] = pack4_sat(q[0], q[1], q[2], q[3]);
    }
  }
}

// One row: packed input codes wcur[i] (0 where the lane's channels lie past C) -> packed output codes outw[i].  Every lane of the
// row group (32 or 64 lanes) must call it: the sums are cross-lane reductions.
template <int NCH, int LANES, class LL>
__device__ __forceinline__ void ln_row(const unsigned (&wcur)[NCH], const LL& L, const p2v_ln& ln, int C, int l32, unsigned (&outw)[NCH]) {
  float xq[NCH][4], rs, mos;
  int S1;
  unsigned S2;
  bool fast;
  ln_sums<NCH>(wcur, L, xq, S1, S2);
  ln_reduce<LANES>(S1, S2);
  ln_scalars<NCH>(S1, S2, L, ln, C, rs, mos, fast);
  ln_apply<NCH, LANES>(xq, L, ln, l32, rs, mos, fast, outw);
}

// R rows of a row group at once (round 3).  The row scalars - three IEEE divisions, a square root and the range tests, ~55 instructions
// that every lane of the group would repeat per row - are computed ONCE for the R rows: lane l keeps the sums of row (l mod R), runs the
// scalar chain on them, and row r's results are read back from lane r of the group (ds_bpermute).  Same operations on the same values
// as ln_row, only in other lanes: bit-identical.
template <int NCH, int LANES, int R, class LL>
__device__ __forceinline__ void ln_rows(const unsigned (&wcur)[R][NCH], const LL& L, const p2v_ln& ln, int C, int l32, unsigned (&outw)[R][NCH]) {
  static_assert((R & (R - 1)) == 0 && R <= 8, "rows per batch");
  float xq[R][NCH][4];
  int S1k = 0;
  unsigned S2k = 0;
  const int mine = l32 & (R - 1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int S1;
    unsigned S2;
    ln_sums<NCH>(wcur[r], L, xq[r], S1, S2);
    ln_reduce<LANES>(S1, S2);
    S1k = mine == r ? S1 : S1k;
    S2k = mine == r ? S2 : S2k;
  }
  float rs, mos;
  bool fast;
  ln_scalars<NCH>(S1k, S2k, L, ln, C, rs, mos, fast);
  const int fasti = fast ? 1 : 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float rs_r = __shfl(rs, r, LANES), mos_r = __shfl(mos, r, LANES);
    const bool fast_r = __shfl(fasti, r, LANES) != 0;
    ln_apply<NCH, LANES>(xq[r], L, ln, l32, rs_r, mos_r, fast_r, outw[r]);
  }
}

// rows per batch in the stand-alone kernel: with every constant in registers two rows cost 23 more VGPRs, i.e. the third wave per SIMD at
// C = 384 and 768 (measured 9 % / 6 % slower); with post_mul and the mask left in LDS (LN_LDSC) the pair fits
#ifndef LN_BATCH
#define LN_BATCH 2
#endif
#ifndef LN_LDSC
#define LN_LDSC true
#endif
template <int NCH, int LANES, bool PRE>        // PRE: the constants were folded when the plan was created (LnPre)
__global__ __launch_bounds__(256) void k_int_layernorm(LnArgs a) {
  // per-channel constants are folded once per workgroup, shared through LDS, then held in registers (re-reading them from
  // LDS per row frees 46 VGPRs but measured 10 % slower: the kernel is bound by VALU issue, not by occupancy)
  __shared__ __attribute__((aligned(16))) float sG[NCH * LANES * 4], sB[NCH * LANES * 4], sP[NCH * LANES * 4];
  __shared__ __attribute__((aligned(16))) int sM[NCH * LANES * 4];
  const int tid = threadIdx.x, l32 = tid & (LANES - 1), hw = tid / LANES;   // l32: lane within the row group
  LnLane<NCH, LN_LDSC> L;
  if constexpr (PRE) {      // only post_mul and the mask go through LDS (LN_LDSC), one barrier
    for (int t4 = tid; t4 < NCH * LANES; t4 += 256) {
      const int c = t4 * 4;
      float4 pmv = make_float4(0.f, 0.f, 0.f, 0.f), mk = pmv;
      if (c < a.C) {
        pmv = *reinterpret_cast<const float4*>(a.ln.post_mul + c);
        mk = *reinterpret_cast<const float4*>(a.ln.mask + c);
      }
      *reinterpret_cast<float4*>(sP + c) = pmv;
      *reinterpret_cast<float4*>(sM + c) = mk;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (l32 + LANES * i) * 4;
      L.on[i] = c < a.C;
      L.gm[i] = *reinterpret_cast<const float4*>(a.pre.gm + c);
      L.bt[i] = *reinterpret_cast<const float4*>(a.pre.bt + c);
      if constexpr (sizeof(L.pm) == sizeof(float4) * NCH) {
        L.pm[i] = *reinterpret_cast<const float4*>(sP + c);
        L.mkf[i] = *reinterpret_cast<const float4*>(sM + c);
      }
    }
    L.sPl = sP + l32 * 4;
    L.sMl = reinterpret_cast<const float*>(sM) + l32 * 4;
    L.cstride = LANES * 4;
    L.gmin = a.pre.gmin;
    L.gmax = a.pre.gmax;
    L.bmax = a.pre.bmax;
    L.pot = a.pre.pot != 0 && a.force_generic == 0;
    L.pm_one = a.pre.pm_one != 0;
#ifdef P2V_EXP_NOTRIM
    L.pm_one = false;
#endif
  } else {
    ln_prepare<NCH, LANES>(a.ln, a.C, a.force_generic != 0, sG, sB, sP, sM, tid, 256, L);
  }
  const int LN_ROWS = a.rows_per_half;
  const long long row0 = ((long long)blockIdx.x * (256 / LANES) + hw) * LN_ROWS;
  // Row r+1 is requested at the top of the iteration of row r and first touched just before the stores of row r.  The two
  // empty asm statements pin that placement: left alone, hipcc sinks the loads of a loop-carried value to the loop end,
  // behind the stores, and waits vmcnt(0) there - two exposed memory round trips per row (measured: 3 us per row).
  int colofs[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) colofs[i] = L.on[i] ? (l32 + LANES * i) * 4 : 0;     // clamped: loads are unconditional
  const long long last_row = a.rows - 1;
  constexpr int R = LN_BATCH;                                      // rows per batch of ln_rows (the row scalars are computed once per batch)
  unsigned wnext[R][NCH];
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      wnext[u][i] = *reinterpret_cast<const unsigned*>(a.x + (row0 + u < a.rows ? row0 + u : last_row) * a.row_stride + colofs[i]);
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(wnext[u][i]));   // first rows landed: no wait is merged into the loop head
#pragma unroll 1
  for (int rr = 0; rr < LN_ROWS; rr += R) {
    const long long row = row0 + rr;
    if (row >= a.rows) break;   // uniform within the row group; the reductions below stay inside it
    unsigned wcur[R][NCH];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) wcur[u][i] = L.on[i] ? wnext[u][i] : 0u;
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long long nrow = row + R + u < a.rows ? row + R + u : last_row;
#pragma unroll
      for (int i = 0; i < NCH; ++i) wnext[u][i] = *reinterpret_cast<const unsigned*>(a.x + nrow * a.row_stride + colofs[i]);
    }
    asm volatile("" ::: "memory");                 // the loads stay above this line
    unsigned outw[R][NCH];
    if constexpr (R == 1) ln_row<NCH, LANES>(wcur[0], L, a.ln, a.C, l32, outw[0]);
    else ln_rows<NCH, LANES, R>(wcur, L, a.ln, a.C, l32, outw);
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(wnext[u][i]));   // the wait for the next rows lands here, ahead of the stores
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (rr + u >= LN_ROWS || row + u >= a.rows) continue;
      int8_t* dst = a.out + (row + u) * a.out_stride;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (L.on[i]) *reinterpret_cast<unsigned*>(dst + (l32 + LANES * i) * 4) = outw[u][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K2b: LayerNorm fused into the GEMM that consumes it (norm1 -> qkv, norm2 -> fc1):  QIntLayerNorm 'int' -> /channel_scale ->
//   qact0 -> QLinear -> (GELU ->) QAct   (vit_fquant.py:431-434,284-293,307; layers_quant.py:305-316,331-333).
//   X-stationary: a workgroup owns 64 rows.  Prologue = the LayerNorm kernel's row code (ln_prepare / ln_row); its output codes
//   go to an LDS panel [K/64][64 rows][64 B] instead of HBM.  Main loop over the 128-column tiles of the layer: wave w computes
//   all 64 rows x columns [32w, 32w+32), so the W rows it needs are its own.  The plan stores these weights a second time in
//   MFMA-FRAGMENT ORDER (p2v_linear.w_frag: [column tile][wave][k-step][lane][16 B]), so the A operand of every MFMA is one fully
//   coalesced 1 KB global load straight into registers: no LDS staging for W, no barrier in the main loop, and all waits are
//   the compiler's own exact scoreboard (an LDS-DMA ring version of this kernel spent ~700 of 890 cycles per k-step on manual
//   wait counting, M0 set-up and DMA issue; profiles/r02_ln_gemm_timeline.txt).  The W fragments of column tile j+1 are requested
//   into the registers tile j has just consumed, one k-step behind the MFMAs: a full tile of lead.
//   What it removes per block: two LayerNorm launches, 2 x (read + write of the residual-sized tensor), and every re-read of the
//   activation panel by the 9 / 12 column-tile workgroups of the tiled kernel.
// ---------------------------------------------------------------------------------------------------
#define LG_BM 64
// W fragments of the fused kernels: 16 bytes per lane (one code per byte) or, for packed int4 weights (p2v_linear.packed4), 8 bytes
// per lane widened in registers like the tiled kernel's (unpack_w4: codes << 4, the 1/16 goes into the column scale)
template <bool W4> struct LgW { typedef uint4 raw; };
template <> struct LgW<true> { typedef uint2 raw; };
__device__ __forceinline__ v4i lg_wfrag(uint4 w) { return __builtin_bit_cast(v4i, w); }
__device__ __forceinline__ v4i lg_wfrag(uint2 w) { return unpack_w4(w.x, w.y); }
struct LnGemmLds {                        // byte offsets inside the dynamic LDS allocation
  int panel, consts, fold, table, total;
};
__host__ __device__ inline LnGemmLds ln_gemm_lds(int K, int N, int nch, int table_cells) {
  LnGemmLds o;
  const int kt = (K + GBK - 1) / GBK, tiles_n = (N + GBN - 1) / GBN;
  o.panel = 0;
  o.consts = o.panel + kt * LG_BM * GBK;
  o.fold = o.panel;        // the LayerNorm fold scratch (4 * nch * 512 B <= the panel) is dead before the first panel row is written
  o.table = o.consts + tiles_n * GBN * 2 * (int)sizeof(float);
  o.total = o.table + table_cells * 8;
  return o;
}

#ifdef P2V_DIAG
#define LG_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (g.stamps && threadIdx.x == 0 && (slot) < 62) g.stamps[(long long)blockIdx.x * 64 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define LG_STAMP(slot) do { } while (0)
#endif
template <int EPI, int KT, bool W4>      // KT = k-tiles of 64 channels (C <= 64*KT); W4: packed int4 fragment copy
__global__ __launch_bounds__(256, 2) void k_ln_gemm(LnArgs a, GemmArgs g) {
  typedef typename LgW<W4>::raw wraw;
  constexpr int NCH = (KT + 1) / 2;      // 128-channel groups of a LayerNorm row
  constexpr int NI = 2 * KT;             // k-steps of 32
  extern __shared__ __attribute__((aligned(1024))) unsigned char lg_smem[];
  LG_STAMP(0);
  const int C = a.C;
  const int cells = EPI == P2V_EPI_GELU_TAB ? g.ep.gelu.cells : 0;
  const LnGemmLds lay = ln_gemm_lds(C, g.N, NCH, cells);
  int8_t* panel = reinterpret_cast<int8_t*>(lg_smem + lay.panel);
  float* consts = reinterpret_cast<float*>(lg_smem + lay.consts);      // per column tile: colscale[128] | bias[128]
  const unsigned char* gtab = lg_smem + lay.table;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * LG_BM;
  const int tiles_n = g.tiles_n;

  // ---- W fragments of column tile 0 (registers): element ((j*4 + wave)*NI + i)*64 + lane of 16 bytes
  const wraw* wsrc = reinterpret_cast<const wraw*>(g.W) + (long long)wave * NI * 64 + lane;
  wraw wf[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) wf[i] = wsrc[i * 64];

  // ---- the 64 rows of the residual stream: one row per half wave, 8 rows each, two rows in flight ahead of the two being normalised
  constexpr int RPH = LG_BM / 8;                                       // rows per half wave
  static_assert(RPH % 2 == 0, "rows are processed in pairs");
  const int hw = tid >> 5;
  auto load_row = [&](int r, unsigned (&w)[NCH]) {
    long long row = (long long)m0 + hw * RPH + r;
    row = row < a.rows ? row : a.rows - 1;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (l31 + 32 * i) * 4;
      w[i] = *reinterpret_cast<const unsigned*>(a.x + row * a.row_stride + (c < C ? c : 0));
    }
  };
  unsigned win[2][NCH];
  load_row(0, win[0]);
  load_row(1, win[1]);
  // ---- per-column constants of the whole layer (REQUANT: 2^e folded in, see gemm_stage_epilogue) and the GELU table
  {
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : 1.0f;
    const float cfold = fold * (W4 ? 0.0625f : 1.0f);                   // packed int4: the accumulator holds 16 x the sum
    for (int n4 = tid; n4 < tiles_n * (GBN / 4); n4 += 256) {           // four columns per thread and turn
      const int j_ = n4 >> 5, c_ = (n4 & 31) * 4;
      const float4 cv = *reinterpret_cast<const float4*>(g.colscale + n4 * 4);   // arrays are padded to n_pad
      const float4 bv = *reinterpret_cast<const float4*>(g.bias + n4 * 4);
      *reinterpret_cast<float4*>(consts + j_ * 2 * GBN + c_) = make_float4(cv.x * cfold, cv.y * cfold, cv.z * cfold, cv.w * cfold);
      *reinterpret_cast<float4*>(consts + j_ * 2 * GBN + GBN + c_) = make_float4(bv.x * fold, bv.y * fold, bv.z * fold, bv.w * fold);
    }
    if (EPI == P2V_EPI_GELU_TAB)
      for (int i = tid; i < cells; i += 256)
        reinterpret_cast<uint2*>(lg_smem + lay.table)[i] = reinterpret_cast<const uint2*>(g.ep.gelu.table)[i];
  }
  LG_STAMP(1);
  // ---- LayerNorm -> LDS panel (and, on request, HBM)
  {
    float* sG = reinterpret_cast<float*>(lg_smem + lay.fold);
    float* sB = sG + NCH * 128;
    float* sP = sB + NCH * 128;
    int* sM = reinterpret_cast<int*>(sP + NCH * 128);
    LnLane<NCH> L;
    ln_prepare<NCH, 32>(a.ln, C, a.force_generic != 0, sG, sB, sP, sM, tid, 256, L);
    __syncthreads();        // every lane holds its folded constants in registers: the scratch (= the panel) may be overwritten
#pragma unroll 1
    for (int r = 0; r < RPH; r += 2) {
      unsigned wnext[2][NCH];
      if (r + 2 < RPH) {
        load_row(r + 2, wnext[0]);
        load_row(r + 3, wnext[1]);
      }
      unsigned wcur[2][NCH], outw2[2][NCH];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < NCH; ++i) wcur[u][i] = L.on[i] ? win[u][i] : 0u;
      ln_rows<NCH, 32, 2>(wcur, L, a.ln, C, l31, outw2);              // the pair shares one pass of the row-scalar arithmetic
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int lrow = hw * RPH + r + u;
        const unsigned (&outw)[NCH] = outw2[u];
        const long long row = (long long)m0 + lrow;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = (l31 + 32 * i) * 4;
          if (c < KT * GBK)      // channels past C inside the last k-tile are zero
            *reinterpret_cast<unsigned*>(panel + (c >> 6) * (LG_BM * GBK) + lrow * GBK + ((((c & 63) >> 4) ^ ((lrow >> 2) & 3)) << 4) + (c & 15)) =
                L.on[i] ? outw[i] : 0u;
          if (a.out && L.on[i] && row < a.rows) *reinterpret_cast<unsigned*>(a.out + row * a.out_stride + c) = outw[i];
        }
      }
      if (r + 2 < RPH) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          win[0][i] = wnext[0][i];
          win[1][i] = wnext[1][i];
        }
      }
    }
  }
  LG_STAMP(2);
  __syncthreads();        // panel, constants and table are complete
  LG_STAMP(3);

  // X fragment addresses: rows l31 / 32+l31 of panel k-tile kt, chunk 2*ks + h
  const int8_t* pXa = panel + lds_off64(l31, h);
  const int8_t* pXb = panel + lds_off64(32 + l31, h);
  const int xks = (lds_off64(l31, 2 + h) - lds_off64(l31, h));           // +-32: the k-step-1 chunk of the same row
  v16i acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0;

  for (int j = 0; j < tiles_n; ++j) {
    const bool more = j + 1 < tiles_n;                                   // wave-uniform
    // the fragment loads are unconditional (the last tile re-requests itself): with a branch around them hipcc cannot count the
    // outstanding requests and waits vmcnt(0) at the top of every tile - i.e. for the output STORES of the tile before
    const wraw* wnext = wsrc + (long long)(more ? j + 1 : j) * 4 * NI * 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int off = (i >> 1) * (LG_BM * GBK) + (i & 1) * xks;
      const v4i xa = *reinterpret_cast<const v4i*>(pXa + off);
      const v4i xb = *reinterpret_cast<const v4i*>(pXb + off);
      const v4i wfi = lg_wfrag(wf[i]);
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfi, xa, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfi, xb, acc[1], 0, 0, 0);
      wf[i] = wnext[i * 64];                                             // the fragment of the next column tile, a tile ahead of its use
    }
    LG_STAMP(4 + 2 * j);
    const EpiLds* e = reinterpret_cast<const EpiLds*>(consts + j * 2 * GBN);
    {
      const uint4 nores[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
      gemm_epilogue_tile2<EPI>(acc, m0 + l31, j * GBN + 32 * wave, 32 * wave, h, g, e, nores, gtab);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0;
    }
    LG_STAMP(5 + 2 * j);
  }
}

// ---------------------------------------------------------------------------------------------------
// K2c: the fused LayerNorm + GEMM kernel, dense form (round 3).  Same data flow as k_ln_gemm (64 rows per workgroup, LayerNorm
//   output in an LDS panel, W fragments straight from the fragment-order copy), reorganised around the two facts measured in
//   round 2 (tools/ubench/valu_rate.hip, profiles/r02_ln_gemm_timeline.txt): a wave alone on its SIMD issues one VALU
//   instruction per ~4.9 cycles, two waves one per ~2.7; and the matrix pipe sat idle through every epilogue.
//     * 8 waves (512 threads, one workgroup per CU, up to 256 registers per wave): two waves per SIMD in every phase.  The
//       LayerNorm phase runs on 16 half waves (4 rows each); in the GEMM phase wave group g = wave >> 2 owns the column tiles
//       g, g + 2, ..., so both groups stream different W tiles and run the same program half a tile apart.
//     * software pipeline across column tiles inside a wave: the MFMAs of the group's NEXT tile (second accumulator set) are
//       issued one at a time between the pieces of the CURRENT tile's epilogue - 24 half-pieces of 4-16 VALU instructions, in
//       program order, pinned with sched_barrier so hipcc cannot re-cluster them.  An MFMA holds the matrix pipe for 32 cycles
//       while the wave goes on issuing the epilogue's VALU work: the k-loop disappears under the epilogue.  Three copies of the
//       tile body: steady state (MFMAs + W prefetch for the tile after), next-to-last (MFMAs only), last (epilogue only).
//   Results are bit-identical to k_ln_gemm (same ln_prepare / ln_row, same epilogue arithmetic); k_ln_gemm stays for the
//   arithmetic GELU epilogue (scales without a table) and for launches with activation taps.
// ---------------------------------------------------------------------------------------------------
// MFMA q (0 .. 2*NI-1; k-step q>>1, row block q&1) of the next tile goes into half-piece (q*12)/NI of the 24 half-pieces
template <int NI>
__host__ __device__ constexpr int lg2_mfma_at(int hp) {
  for (int q = 0; q < 2 * NI; ++q)
    if ((q * 12) / NI == hp) return q;
  return -1;
}

// EPI: P2V_EPI_REQUANT or P2V_EPI_GELU_TAB;  KT = k-tiles of 64 channels (C <= 64*KT);  NG = wave groups (1: 4 waves, every wave all
// column tiles, two workgroups per CU; 2: 8 waves, the groups alternate column tiles, one workgroup per CU)
#ifndef LG2_LN_ROWS
#define LG2_LN_ROWS 2   /* 4 spills at C = 384 (196 B of scratch): 89.4 k against 98.6 k img/s */
#endif
template <int EPI, int KT, int NG, bool W4>
__global__ __launch_bounds__(256 * NG, 2) void k_ln_gemm2(LnArgs a, GemmArgs g) {
  typedef typename LgW<W4>::raw wraw;
  constexpr int NT = 256 * NG;           // threads
  static_assert(EPI == P2V_EPI_REQUANT || EPI == P2V_EPI_GELU_TAB, "pipelined epilogues");
  constexpr int NCH = (KT + 1) / 2;      // 128-channel groups of a LayerNorm row
  constexpr int NI = 2 * KT;             // k-steps of 32
  extern __shared__ __attribute__((aligned(1024))) unsigned char lg_smem[];
  LG_STAMP(0);
  const int C = a.C;
  const int cells = EPI == P2V_EPI_GELU_TAB ? g.ep.gelu.cells : 0;
  const LnGemmLds lay = ln_gemm_lds(C, g.N, NCH, cells);
  int8_t* panel = reinterpret_cast<int8_t*>(lg_smem + lay.panel);
  float* consts = reinterpret_cast<float*>(lg_smem + lay.consts);      // per column tile: colscale[128] | bias[128]
  const unsigned char* gtab = lg_smem + lay.table;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = NG == 1 ? 0 : (wave >> 2), cw = wave & 3;            // wave group (column-tile parity), 32-column block of a tile
  const int h = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * LG_BM;
  const int tiles_n = g.tiles_n;
  const int n_g = (tiles_n - grp + NG - 1) / NG;                       // column tiles of this group: j = grp + NG*it

  // ---- W fragments of the group's first tile (registers): element ((j*4 + cw)*NI + i)*64 + lane of 16 bytes
  const wraw* wsrc = reinterpret_cast<const wraw*>(g.W) + (long long)cw * NI * 64 + lane;
  auto wtile = [&](int it) {                                           // fragments of the group's it-th tile (clamped: loads are unconditional)
    int j = grp + NG * it;
    j = j < tiles_n ? j : tiles_n - 1;
    return wsrc + (long long)j * 4 * NI * 64;
  };
  wraw wf[NI];
  {
    const wraw* w0 = wtile(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = w0[i * 64];
  }

  // ---- the 64 rows of the residual stream: one row per half wave, 4 rows each, the second pair in flight while the first is normalised
  constexpr int RPH = LG_BM / (8 * NG);                                // rows per half wave
  static_assert(RPH % 2 == 0, "rows are processed in pairs");
  const int hw = tid >> 5;
  // 32-bit offsets from the workgroup's first row (64 rows x row stride < 2^31: checked by the launcher): the 64-bit row * stride
  // products of the first version cost ~20 VALU instructions per row pair
  const int8_t* xblk = a.x + (long long)m0 * a.row_stride;
  const int rstride = (int)a.row_stride;
  const int last_lr = (int)(a.rows - 1 - m0);                           // rows past the end re-read the last row (never stored)
  auto load_row = [&](int r, unsigned (&w)[NCH]) {
    int lr = hw * RPH + r;
    lr = lr < last_lr ? lr : last_lr;
    const int8_t* rp = xblk + lr * rstride;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (l31 + 32 * i) * 4;
      w[i] = *reinterpret_cast<const unsigned*>(rp + (c < C ? c : 0));
    }
  };
  constexpr int LR = LG2_LN_ROWS < RPH ? LG2_LN_ROWS : RPH;             // rows per LayerNorm batch (ln_rows)
  static_assert(RPH % LR == 0, "rows are processed in batches");
  unsigned win[LR][NCH];
#pragma unroll
  for (int u = 0; u < LR; ++u) load_row(u, win[u]);
  // ---- per-column constants of the whole layer (REQUANT: 2^e folded in, see gemm_stage_epilogue) and the GELU table
  {
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : 1.0f;
    const float cfold = fold * (W4 ? 0.0625f : 1.0f);                   // packed int4: the accumulator holds 16 x the sum
    for (int n4 = tid; n4 < tiles_n * (GBN / 4); n4 += NT) {           // four columns per thread and turn
      const int j_ = n4 >> 5, c_ = (n4 & 31) * 4;
      const float4 cv = *reinterpret_cast<const float4*>(g.colscale + n4 * 4);   // arrays are padded to n_pad
      const float4 bv = *reinterpret_cast<const float4*>(g.bias + n4 * 4);
      *reinterpret_cast<float4*>(consts + j_ * 2 * GBN + c_) = make_float4(cv.x * cfold, cv.y * cfold, cv.z * cfold, cv.w * cfold);
      *reinterpret_cast<float4*>(consts + j_ * 2 * GBN + GBN + c_) = make_float4(bv.x * fold, bv.y * fold, bv.z * fold, bv.w * fold);
    }
    if (EPI == P2V_EPI_GELU_TAB)
      for (int i = tid; i < cells; i += NT)
        reinterpret_cast<uint2*>(lg_smem + lay.table)[i] = reinterpret_cast<const uint2*>(g.ep.gelu.table)[i];
  }
  LG_STAMP(1);
  // ---- LayerNorm -> LDS panel (and, on request, HBM)
  {
    float* sG = reinterpret_cast<float*>(lg_smem + lay.fold);
    float* sB = sG + NCH * 128;
    float* sP = sB + NCH * 128;
    int* sM = reinterpret_cast<int*>(sP + NCH * 128);
    LnLane<NCH> L;
    if (a.pre.gm) {         // folded when the plan was created (LnPre): straight into registers - no scratch, no barriers
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (l31 + 32 * i) * 4;
        L.on[i] = c < C;
        L.gm[i] = *reinterpret_cast<const float4*>(a.pre.gm + c);
        L.bt[i] = *reinterpret_cast<const float4*>(a.pre.bt + c);
        L.pm[i] = L.on[i] ? *reinterpret_cast<const float4*>(a.ln.post_mul + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        L.mkf[i] = L.on[i] ? *reinterpret_cast<const float4*>(a.ln.mask + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      L.gmin = a.pre.gmin;
      L.gmax = a.pre.gmax;
      L.bmax = a.pre.bmax;
      L.pot = a.pre.pot != 0 && a.force_generic == 0;
      L.pm_one = a.pre.pm_one != 0;
#ifdef P2V_EXP_NOTRIM
      L.pm_one = false;
#endif
    } else {
      ln_prepare<NCH, 32>(a.ln, C, a.force_generic != 0, sG, sB, sP, sM, tid, NT, L);
      __syncthreads();      // every lane holds its folded constants in registers: the scratch (= the panel) may be overwritten
    }
#pragma unroll 1
    for (int r = 0; r < RPH; r += LR) {
      unsigned wnext[LR][NCH];
      if (r + LR < RPH) {
#pragma unroll
        for (int u = 0; u < LR; ++u) load_row(r + LR + u, wnext[u]);
      }
      unsigned wcur[LR][NCH], outw2[LR][NCH];
#pragma unroll
      for (int u = 0; u < LR; ++u)
#pragma unroll
        for (int i = 0; i < NCH; ++i) wcur[u][i] = L.on[i] ? win[u][i] : 0u;
      ln_rows<NCH, 32, LR>(wcur, L, a.ln, C, l31, outw2);             // the batch shares one pass of the row-scalar arithmetic
#pragma unroll
      for (int u = 0; u < LR; ++u) {
        const int lrow = hw * RPH + r + u;
        const unsigned (&outw)[NCH] = outw2[u];
        const long long row = (long long)m0 + lrow;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = (l31 + 32 * i) * 4;
          if (c < KT * GBK)      // channels past C inside the last k-tile are zero
            *reinterpret_cast<unsigned*>(panel + (c >> 6) * (LG_BM * GBK) + lrow * GBK + ((((c & 63) >> 4) ^ ((lrow >> 2) & 3)) << 4) + (c & 15)) =
                L.on[i] ? outw[i] : 0u;
          if (a.out && L.on[i] && row < a.rows) *reinterpret_cast<unsigned*>(a.out + row * a.out_stride + c) = outw[i];
        }
      }
      if (r + LR < RPH) {
#pragma unroll
        for (int u = 0; u < LR; ++u)
#pragma unroll
          for (int i = 0; i < NCH; ++i) win[u][i] = wnext[u][i];
      }
    }
  }
  LG_STAMP(2);
  __syncthreads();        // panel, constants and table are complete
  LG_STAMP(3);
  if (n_g <= 0) return;   // wave-uniform (a layer with a single column tile: the second group has nothing to do)

  // X fragment addresses: rows l31 / 32+l31 of panel k-tile kt, chunk 2*ks + h
  const int8_t* pXa = panel + lds_off64(l31, h);
  const int8_t* pXb = panel + lds_off64(32 + l31, h);
  const int xks = (lds_off64(l31, 2 + h) - lds_off64(l31, h));           // +-32: the k-step-1 chunk of the same row
#define LG2_XOFF(i) ((((i) >> 1) * (LG_BM * GBK)) + (((i) & 1) ? xks : 0))
  v16i acc[2], accn[2];                                                  // current tile (epilogue) / next tile (MFMAs)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0;
  // ---- the group's first tile: plain k-loop (nothing to overlap with), W fragments of its second tile requested behind the MFMAs
  {
    const wraw* wn = wtile(1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const v4i xa = *reinterpret_cast<const v4i*>(pXa + LG2_XOFF(i));
      const v4i xb = *reinterpret_cast<const v4i*>(pXb + LG2_XOFF(i));
      const v4i wfi = lg_wfrag(wf[i]);
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfi, xa, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfi, xb, acc[1], 0, 0, 0);
      wf[i] = wn[i * 64];
    }
  }
  LG_STAMP(4);
  // X fragments of k-step 0 for the first interleaved MFMAs (double-buffered by k-step parity; the panel is the same for every tile)
  v4i XA[2], XB[2];
  XA[0] = *reinterpret_cast<const v4i*>(pXa + LG2_XOFF(0));
  XB[0] = *reinterpret_cast<const v4i*>(pXb + LG2_XOFF(0));
  XA[1] = XA[0];
  XB[1] = XB[0];
  const float gk = g.ep.gelu.k, goff = g.ep.gelu.off, gtmax = (float)(cells - 1);

  // One column tile: the epilogue of acc (tile j) in 24 half-pieces; with MF the 2*NI MFMAs of the group's next tile are issued
  // one per half-piece into accn (X fragments one k-step ahead); with LD the W fragment of the tile after next replaces the one
  // an MFMA pair has just consumed.
  v4i wcur = {0, 0, 0, 0};                                               // the widened fragment between the two MFMAs of a k-step
  // acc_ / accn_: the accumulators of this tile / of the next one.  The steady-state loop runs the body twice per turn with the two sets
  // exchanged instead of copying 32 registers per tile (COPY = false); the odd tile and the two tail forms copy the next set into the first
  auto tile_body = [&](auto MFc, auto LDc, auto COPYc, int j, const wraw* wnn, v16i (&acc)[2], v16i (&accn)[2]) {
    constexpr bool MF = decltype(MFc)::value, LD = decltype(LDc)::value, COPY = decltype(COPYc)::value;
    const float* cst = consts + j * 2 * GBN + 32 * cw + 4 * h;           // colscale of this wave's columns; bias at + GBN
    const int n_tile = j * GBN + 32 * cw;
    unsigned d[2][4];
    float4 cs = *reinterpret_cast<const float4*>(cst), bs = *reinterpret_cast<const float4*>(cst + GBN);
#define LG2_MFMA(HP)                                                                                                     \
    do {                                                                                                                 \
      constexpr int q_ = lg2_mfma_at<NI>(HP);                                                                            \
      if constexpr (MF && q_ >= 0) {                                                                                     \
        constexpr int i_ = q_ >> 1, nx_ = (i_ + 1) % NI;                                                                 \
        if constexpr ((q_ & 1) == 0) {                                                                                   \
          XA[nx_ & 1] = *reinterpret_cast<const v4i*>(pXa + LG2_XOFF(nx_));                                              \
          wcur = lg_wfrag(wf[i_]);                                                                                       \
          if constexpr (i_ == 0) accn[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wcur, XA[i_ & 1], (v16i){0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, 0, 0, 0); \
          else accn[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wcur, XA[i_ & 1], accn[0], 0, 0, 0);                      \
        } else {                                                                                                         \
          XB[nx_ & 1] = *reinterpret_cast<const v4i*>(pXb + LG2_XOFF(nx_));                                              \
          if constexpr (i_ == 0) accn[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wcur, XB[i_ & 1], (v16i){0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, 0, 0, 0); \
          else accn[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wcur, XB[i_ & 1], accn[1], 0, 0, 0);                      \
          if constexpr (LD) wf[i_] = wnn[i_ * 64];                                                                       \
        }                                                                                                                \
      }                                                                                                                  \
    } while (0)
#define LG2_FENCE() __builtin_amdgcn_sched_barrier(0)
    auto group = [&](auto GQc) {
      constexpr int gq = decltype(GQc)::value;
      const float4 csc = cs, bsc = bs;
      if (gq < 3) {                                                      // constants of the next group: an LDS round trip ahead
        cs = *reinterpret_cast<const float4*>(cst + 8 * (gq + 1));
        bs = *reinterpret_cast<const float4*>(cst + GBN + 8 * (gq + 1));
      }
      float y0[4], y1[4];
      // F.linear on fake-quantised operands: exact integer sum * (s_x*s_w[n]), then ONE rounding for the fp32 bias (layers.py:178)
      LG2_MFMA(6 * gq + 0);
      y0[0] = __builtin_fmaf((float)acc[0][4 * gq + 0], csc.x, bsc.x);
      y0[1] = __builtin_fmaf((float)acc[0][4 * gq + 1], csc.y, bsc.y);
      y0[2] = __builtin_fmaf((float)acc[0][4 * gq + 2], csc.z, bsc.z);
      y0[3] = __builtin_fmaf((float)acc[0][4 * gq + 3], csc.w, bsc.w);
      LG2_FENCE();
      LG2_MFMA(6 * gq + 1);
      y1[0] = __builtin_fmaf((float)acc[1][4 * gq + 0], csc.x, bsc.x);
      y1[1] = __builtin_fmaf((float)acc[1][4 * gq + 1], csc.y, bsc.y);
      y1[2] = __builtin_fmaf((float)acc[1][4 * gq + 2], csc.z, bsc.z);
      y1[3] = __builtin_fmaf((float)acc[1][4 * gq + 3], csc.w, bsc.w);
      LG2_FENCE();
      if constexpr (EPI == P2V_EPI_GELU_TAB) {
        uint2 e0[4], e1[4];
        LG2_MFMA(6 * gq + 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) e0[i] = *reinterpret_cast<const uint2*>(gtab + gelu_tab_offset(y0[i], gk, goff, gtmax));
        LG2_FENCE();
        LG2_MFMA(6 * gq + 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) e1[i] = *reinterpret_cast<const uint2*>(gtab + gelu_tab_offset(y1[i], gk, goff, gtmax));
        LG2_FENCE();
        LG2_MFMA(6 * gq + 4);
        d[0][gq] = 0;
        P2V_GELU_SEL(0, "UNUSED_PAD", d[0][gq], y0[0], e0[0]);
        P2V_GELU_SEL(1, "UNUSED_PRESERVE", d[0][gq], y0[1], e0[1]);
        P2V_GELU_SEL(2, "UNUSED_PRESERVE", d[0][gq], y0[2], e0[2]);
        P2V_GELU_SEL(3, "UNUSED_PRESERVE", d[0][gq], y0[3], e0[3]);
        LG2_FENCE();
        LG2_MFMA(6 * gq + 5);
        d[1][gq] = 0;
        P2V_GELU_SEL(0, "UNUSED_PAD", d[1][gq], y1[0], e1[0]);
        P2V_GELU_SEL(1, "UNUSED_PRESERVE", d[1][gq], y1[1], e1[1]);
        P2V_GELU_SEL(2, "UNUSED_PRESERVE", d[1][gq], y1[2], e1[2]);
        P2V_GELU_SEL(3, "UNUSED_PRESERVE", d[1][gq], y1[3], e1[3]);
        LG2_FENCE();
      } else {       // REQUANT: the 2^e of the following QAct is folded into the constants; the byte packing saturates
        float r0[4], r1[4];
        LG2_MFMA(6 * gq + 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) r0[i] = pre_pack(y0[i]);
        LG2_FENCE();
        LG2_MFMA(6 * gq + 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) r1[i] = pre_pack(y1[i]);
        LG2_FENCE();
        LG2_MFMA(6 * gq + 4);
        d[0][gq] = pack4_pre(r0[0], r0[1], r0[2], r0[3]);
        LG2_FENCE();
        LG2_MFMA(6 * gq + 5);
        d[1][gq] = pack4_pre(r1[0], r1[1], r1[2], r1[3]);
        LG2_FENCE();
      }
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
#undef LG2_MFMA
#undef LG2_FENCE
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint4 o = halves_to_row16(d[b][0], d[b][1], d[b][2], d[b][3]);
      const int m = m0 + 32 * b + l31;
      if (m < g.M && n_tile + 16 * h < g.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(g.out) + (long long)m * g.ldo + n_tile + 16 * h) = o;
    }
    if constexpr (MF && COPY) {
      acc[0] = accn[0];
      acc[1] = accn[1];
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  int it = 0;
  for (; it + 3 < n_g; it += 2) {
    tile_body(T_{}, T_{}, F_{}, grp + NG * it, wtile(it + 2), acc, accn);
    LG_STAMP(5 + it);
    tile_body(T_{}, T_{}, F_{}, grp + NG * (it + 1), wtile(it + 3), accn, acc);
    LG_STAMP(6 + it);
  }
  if (it + 2 < n_g) {
    tile_body(T_{}, T_{}, T_{}, grp + NG * it, wtile(it + 2), acc, accn);
    LG_STAMP(5 + it);
    ++it;
  }
  if (it + 1 < n_g) {
    tile_body(T_{}, F_{}, T_{}, grp + NG * it, wsrc, acc, accn);
    LG_STAMP(5 + it);
    ++it;
  }
  tile_body(F_{}, F_{}, T_{}, grp + NG * it, wsrc, acc, accn);
  LG_STAMP(5 + it);
#undef LG2_XOFF
}

// ---------------------------------------------------------------------------------------------------
// K3: fused attention core  (vit_fquant.py:309-326; QIntSoftmax layers.py:323-376)
//   one workgroup per (image, head); K (int8) and V^T (bf16) staged in LDS; each wave owns 16-query
//   blocks.  S^T = K . Q^T on v_mfma_i32_16x16x64_i8 (one instruction covers head_dim 64) puts a score row
//   on the 4 lanes {q, q+16, q+32, q+48}: 4 keys per 16-key block per lane, so the row max and the exact
//   int64 sum of exp_int = z * 2^(32-q) are in-lane plus two cross-lane steps.  exp_int depends only on
//   (max - score) in [0,255]: a 256-entry LDS table.  P = 2^-k is exact in bf16 and V codes are exact in
//   bf16, so P.V on v_mfma_f32_16x16x32_bf16 is exact in its fp32 accumulator (|sum| < 2^24 units of 2^-15).
//   The accumulator of S^T is already the B operand of the P.V product (k index = key): formal k = 8g+j of
//   a 32-key step is key 4g+j (j<4) / 16+4g+(j-4) (j>=4); the V^T fragment is read with the same map.
//   ~100 VGPRs -> 4 waves/SIMD, 3 workgroups (47 KB LDS each) per CU.
// ---------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef short v2i16 __attribute__((ext_vector_type(2)));

// ISH: the score multiplier qk_scale * s_q1^2 / s_attn is 2^-pshift with pshift >= 1 (head_dim 64: qk_scale = 1/8): the qact_attn1
// codes come from an integer round-half-even shift instead of the fp32 cvt / mul / rndne / med3 / cvt chain (2.5 VALU per score less)
#ifdef P2V_DIAG
extern unsigned long long* g_gemm_stamps;
#define AT_STAMP(slot)                                                                                             \
  do {                                                                                                             \
    if (a.stamps && threadIdx.x == 0 && (slot) < 16) a.stamps[(long long)blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define AT_STAMP(slot) do { } while (0)
#endif
// NKP = 32-key pairs covering the tokens (7 for 197; 19 for the 577 tokens of 384^2 / 16); TAP: also write probs_k.  The score slots of a
// query block live in registers (8 per 32-key pair): up to NKP = 7 the kernel fits 128 VGPRs (four waves per SIMD), up to 10 it takes 168,
// beyond that 256 (one 8-wave workgroup per CU; K / V^T of 608 keys x head_dim 64 need 121 KB of LDS).
template <int HD, int NKP, bool TAP, bool ISH>
__global__ __launch_bounds__(512, NKP <= 7 ? 4 : (NKP <= 10 ? 3 : 2)) void k_lis_attention(AttnArgs a) {
  constexpr int KROWS = NKP * 32;
  constexpr int NKB = NKP * 2;                  // 16-key blocks
  constexpr int VSTRIDE = KROWS + 4;            // bf16 elements; dword stride = 2*odd -> conflict-free b64 reads
  constexpr int CH = HD / 16;                   // 16-byte chunks per K row
  constexpr int NDT = HD / 16;                  // 16-wide output-channel tiles
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int8_t* sK = reinterpret_cast<int8_t*>(smem);                                  // [KROWS][HD] swizzled
  unsigned short* sVt = reinterpret_cast<unsigned short*>(smem + KROWS * HD);    // [HD][VSTRIDE] bf16
  // two 8-byte-stride tables addressed by the same byte offset 8*d: exp_int (int64) and the fp64 reciprocal of float(exp_int)
  unsigned char* lutE = smem + KROWS * HD + HD * VSTRIDE * 2;                      // [257] long long
  unsigned char* lutFR = lutE + 258 * 8;                                           // [257] double
  // the score slots hold ABSOLUTE LDS byte addresses of their exp_int entry (table base folded into the per-row constant of the
  // subtraction): the gathers need no address arithmetic (hipcc otherwise adds the zero base of the dynamic LDS block per element)
  typedef __attribute__((address_space(3))) const long long* lds_i64p;
  typedef __attribute__((address_space(3))) const double* lds_f64p;
  const int ebase = (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lutE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x / a.H, head = blockIdx.x % a.H;
  const int N = a.N, D = a.H * HD, ld = 3 * D;
  const int8_t* base = a.qkv + (long long)b * N * ld + head * HD;
  AT_STAMP(0);

  // exp table: d = max - score -> exp_int = z * 2^(32-q)       (int_exp / int_polynomial, layers.py:334-358)
  // entry 256 is the sentinel of padded keys: contributes 0 to the sum and maps to probability 0.
  if (tid < 256) {
    int xi = -tid;
    const int lim = 32 * a.at.x0_int;
    xi = xi < lim ? lim : xi;
    const int q = xi / a.at.x0_int;              // both <= 0: trunc == floor
    const int r = xi - a.at.x0_int * q;
    const long long z = (long long)r * (r + a.at.b_int) + a.at.c_int;
    long long e = z << (32 - q);
    e = e < 0 ? 0 : e;
    const float ef = (float)e;                   // exact: z < 2^24; 1 <= e <= 2^56 (z > 0 on (x0, 0], shift >= 0)
    reinterpret_cast<long long*>(lutE)[tid] = e;
    // the fp64 reciprocal (IEEE division, correctly rounded): the per-score quotient is one fp64 multiply by it, see below
    reinterpret_cast<double*>(lutFR)[tid] = 1.0 / (double)ef;
    if (tid == 0) {
      reinterpret_cast<long long*>(lutE)[256] = 0;
      reinterpret_cast<double*>(lutFR)[256] = 1.0;                          // sum / 1 >= 2^32 -> k clamps to 16 -> probability 0
    }
  }
  // stage K rows (swizzled so that a 16-row x 16-byte-chunk fragment read is conflict free) and V^T (bf16)
  AT_STAMP(1);
  // two chunks per thread and turn: all four global loads are requested before the first LDS store waits for one
  for (int i0 = tid; i0 < KROWS * CH; i0 += 2 * (int)blockDim.x) {
    uint4 kv[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      const int row = i / CH, c = i % CH;
      kv[u] = make_uint4(0, 0, 0, 0);
      vv[u] = make_uint4(0, 0, 0, 0);
      if (i < KROWS * CH && row < N) {
        kv[u] = *reinterpret_cast<const uint4*>(base + (long long)row * ld + D + c * 16);
        vv[u] = *reinterpret_cast<const uint4*>(base + (long long)row * ld + 2 * D + c * 16);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i >= KROWS * CH) break;
      const int row = i / CH, c = i % CH;
      const int sw = (HD == 64) ? (c ^ (((row >> 3) & 1) << 1)) : c;
      *reinterpret_cast<uint4*>(sK + row * HD + sw * 16) = kv[u];
      const unsigned w4[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float f = (float)sx8(w4[j >> 2], j & 3);
        sVt[(c * 16 + j) * VSTRIDE + row] = (unsigned short)(__float_as_uint(f) >> 16);   // exact bf16
      }
    }
  }

  // (q@k^T)*scale / s_attn  ==  (acc * qk_scale) * (s_q1^2 / s_attn): the power-of-two factors commute with
  // the single rounding of the *scale product (vit_fquant.py:316-317)
  // s_q1^2 / s_attn is a power of two (checked by the launcher), so ((acc * qk_scale) * 2^e) == acc * (qk_scale * 2^e)
  // with the same single rounding; the NEGATED code is produced (round-half-even and the clamp are symmetric).
  const float nmm = -(a.at.qk_scale * (a.at.s_qkv_sq * a.at.inv_s_attn));
  const int nqb = (N + 15) >> 4;
  const int nwaves = (int)(blockDim.x >> 6);
  // the Q fragment of a wave's first query block is requested before the barrier and the one of its next block a block ahead: its
  // global-memory latency overlaps the staging wait / the arithmetic of the current block
  AT_STAMP(2);
  v4i fq_next = {0, 0, 0, 0};
  if (g < CH && wave < nqb) {
    const int qr0 = wave * 16 + l15;
    fq_next = *reinterpret_cast<const v4i*>(base + (long long)(qr0 < N ? qr0 : N - 1) * ld + g * 16);
  }
  __syncthreads();
  AT_STAMP(3);
  [[maybe_unused]] int stamp_base = 4;
  for (int qb = wave; qb < nqb; qb += nwaves) {
    const int qrow = qb * 16 + l15;
    const v4i fq = fq_next;
    if (g < CH && qb + nwaves < nqb) {
      const int qn = (qb + nwaves) * 16 + l15;
      fq_next = *reinterpret_cast<const v4i*>(base + (long long)(qn < N ? qn : N - 1) * ld + g * 16);
    }
    v4i s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {          // all score MFMAs first: no dependent use behind an MFMA
      const int row = kb * 16 + l15, c = g & (CH - 1);
      const int sw = (HD == 64) ? (c ^ (((row >> 3) & 1) << 1)) : c;
      v4i fk = *reinterpret_cast<const v4i*>(sK + row * HD + sw * 16);
      if (g >= CH) fk = (v4i){0, 0, 0, 0};
      s[kb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq, (v4i){0, 0, 0, 0}, 0, 0, 0);
    }
    const bool tail_empty = (NKB - 1) * 16 >= N;     // last 16-key block holds only padding (e.g. N = 197: keys 208..223)
#ifdef P2V_DIAG
    asm volatile("s_nop 0" :: "v"(s[NKB - 1][0]));       // the stamp below waits for the last score MFMA
#endif
    AT_STAMP(stamp_base);
    long long S = 0;
    if (ISH) {
      // codes = clamp(rne(score * 2^-p)): (float)score * 2^-p is exact, so torch.round of it is the integer round-half-even shift
      // (s + 2^(p-1) - 1 + bit p of s) >> p.  Row max of the codes; d = max - code; padded keys get a code far below every real one.
      const int p = a.pshift, hm1 = (1 << (p - 1)) - 1;
      int mx = -100000;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb == NKB - 1 && tail_empty) continue;       // wave-uniform: no arithmetic for a block of padding
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int sv = s[kb][r];
          int code = (int)((unsigned)sv + (unsigned)hm1 + (((unsigned)sv >> p) & 1u)) >> p;    // |sv| < 2^21: no overflow
          code = code < -128 ? -128 : (code > 127 ? 127 : code);
          if (kb >= NKB - 2) code = (kb * 16 + 4 * g + r) < N ? code : -100000;
          s[kb][r] = code;
          mx = code > mx ? code : mx;
        }
      }
      {
        int o = __shfl_xor(mx, 16);
        mx = o > mx ? o : mx;
        o = __shfl_xor(mx, 32);
        mx = o > mx ? o : mx;
      }
      const int mx8 = (mx << 3) + ebase;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb == NKB - 1 && tail_empty) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int d8;                                          // 8 * (max - code), the byte offset into both tables, in ONE instruction
          asm("v_mad_i32_i24 %0, %1, -8, %2" : "=v"(d8) : "v"(s[kb][r]), "v"(mx8));      // (hipcc splits mul24(x,-8)+y into shift + sub)
          if (kb >= NKB - 2) d8 = d8 > ebase + 2048 ? ebase + 2048 : d8;   // padding -> the sentinel entry
          s[kb][r] = d8;
          S += *(lds_i64p)(uintptr_t)(unsigned)d8;
        }
        __builtin_amdgcn_sched_barrier(0);                         // keep live ranges short
      }
    } else {
    // scores -> NEGATED int8 codes of qact_attn1 (nc = -code) ; row min of nc = -(row max).  Padded keys get +1000.
    int mn = 1000;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb == NKB - 1 && tail_empty) continue;       // wave-uniform: no arithmetic for a block of padding
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int nc = (int)__builtin_amdgcn_fmed3f(rintf((float)s[kb][r] * nmm), -127.f, 128.f);
        if (kb >= NKB - 2) nc = (kb * 16 + 4 * g + r) < N ? nc : 1000;
        s[kb][r] = nc;
        mn = nc < mn ? nc : mn;
      }
    }
    {
      int o = __shfl_xor(mn, 16);
      mn = o < mn ? o : mn;
      o = __shfl_xor(mn, 32);
      mn = o < mn ? o : mn;
    }
    // d = max - code = nc - mn in [0, 255]; s[][] := 8*d, the byte offset into both tables (256 = sentinel of padding)
    const int neg8mn = -8 * mn + ebase;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb == NKB - 1 && tail_empty) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int d8 = (s[kb][r] << 3) + neg8mn;
        if (kb >= NKB - 2) d8 = d8 > ebase + 2048 ? ebase + 2048 : d8;
        s[kb][r] = d8;
        S += *(lds_i64p)(uintptr_t)(unsigned)d8;
      }
      __builtin_amdgcn_sched_barrier(0);                         // keep live ranges short
    }
    }
    S += __shfl_xor(S, 16);
    S += __shfl_xor(S, 32);
    const float Sf = (float)S;                                  // exp_int.sum(-1): exact, then one rounding
    const double Sd = (double)Sf;
    AT_STAMP(stamp_base + 1);

    v4f o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NKP; ++p) {
      unsigned pk[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float ratio[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * e2 + e;                              // element of the 8-wide B fragment
          const int kb = 2 * p + (j >> 2), r = j & 3;
          if (kb == NKB - 1 && tail_empty) {                     // (compile-time kb, wave-uniform flag)
            ratio[e] = 4.0e9f;                                   // -> probability 0
            continue;
          }
          // round(sum / exp_int), layers.py:370: the correctly rounded fp32 quotient from ONE fp64 multiply and one conversion.
          // Sf = A 2^a and exp_int = B 2^b with integers A, B < 2^24, so A/B lies at least 2^-49 (relative) away from every
          // fp32 rounding boundary (|A - mB| is a non-zero multiple of the boundary's unit, B < 2^24) and is never one itself
          // (a 25-bit odd m times B has more than 24 bits); Sd * RN64(1/exp_int) is within 2^-52 of A/B, so converting it to
          // fp32 rounds to the same side.  (v_mul_f64 + v_cvt_f32_f64 replace v_mul_f32 + four 3-source v_fma_f32.)
          const double rd = ((lds_f64p)(uintptr_t)(unsigned)s[kb][r])[258];          // the reciprocal table starts 258 entries behind exp_int
          ratio[e] = rintf((float)(Sd * rd));
          if (TAP && s[kb][r] < ebase + 2048 && qrow < N) {
            int k = (int)((__float_as_uint(ratio[e]) + 0x00400000u) >> 23) - 127;   // log_round, layers.py:323-329
            a.probs_k[(((long long)b * a.H + head) * N + qrow) * N + kb * 16 + 4 * g + r] = (int8_t)(k > 16 ? 16 : k);
          }
        }
        // log_round on the two high halves at once: E = (bits + 0x00400000) >> 23 is the biased exponent of 2^k
        // (ratio >= 1 so k >= 0); 2^-k as bf16 is (254 - E) << 7, and k >= 16 (E >= 143) -> 0   (layers.py:372-375)
        const unsigned hi2 = __builtin_amdgcn_perm(__float_as_uint(ratio[1]), __float_as_uint(ratio[0]), 0x07060302u);
        const v2u16 eb = __builtin_bit_cast(v2u16, (__builtin_bit_cast(unsigned, __builtin_bit_cast(v2u16, hi2) + (v2u16){0x40, 0x40})) & 0x7F807F80u);
        const v2u16 hb = (v2u16){0x7F00, 0x7F00} - eb;                                  // (254 - E) << 7
        const v2i16 neg = __builtin_bit_cast(v2i16, (v2u16)(hb - (v2u16){0x3800, 0x3800})) >> (v2i16){15, 15};   // all ones where k >= 16
        pk[e2] = __builtin_bit_cast(unsigned, hb) & ~__builtin_bit_cast(unsigned, neg);
      }
      v4i pb = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
      const v8bf fb = __builtin_bit_cast(v8bf, pb);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const unsigned short* vp = sVt + (dt * 16 + l15) * VSTRIDE + p * 32 + 4 * g;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
        v4i va = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, va), fb, o[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // qact2: (attn @ v) / s  with attn@v = O * s_q1   (vit_fquant.py:325-326); lane owns channels 16dt+4g..+3
#ifdef P2V_DIAG
    asm volatile("s_nop 0" :: "v"(o[NDT - 1][0]));
#endif
    AT_STAMP(stamp_base + 2);
    {
      // unconditional stores: a padding query row (qrow >= N) was computed from the Q fragment of row N-1, so its values ARE row N-1's
      // and it may store them there.  With the stores behind a branch hipcc cannot count them and waits vmcnt(0) - for these
      // stores - before the next block may use its prefetched Q fragment
      const int qs = qrow < N ? qrow : N - 1;
      int8_t* dst = a.out + ((long long)b * N + qs) * D + head * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        *reinterpret_cast<unsigned*>(dst + dt * 16) =
            pack4_rne_sat(o[dt][0] * a.at.av_mul, o[dt][1] * a.at.av_mul, o[dt][2] * a.at.av_mul, o[dt][3] * a.at.av_mul);
    }
    AT_STAMP(stamp_base + 3);
    stamp_base += 4;
  }
}

// ---------------------------------------------------------------------------------------------------
// K4: Swin window attention core (swin_quant.py:186-217, 366-391), head_dim 32, windows of ws*ws <= 64 tokens.
//   One wave per (image, window, head), four heads per workgroup; structure of k_lis_attention (S^T = K.Q^T on
//   v_mfma_i32_16x16x64_i8, a score row on 4 lanes, exp_int table, exact int64 sum, P.V on v_mfma_f32_16x16x32_bf16).
//   The reference multiplies the dequantised q by head_dim^-0.5 (not a power of two at head_dim 32) BEFORE the dot product:
//   v_c = RN32(code_c * sigma), sigma = s_q1 * scale.  The rounding error of that product is a small integer number of
//   units u = ulp(sigma):  v_c = code_c*sigma + eta_c*u with |eta_c| <= 64, eta_c = fma(code_c, sigma, -v_c)/u exactly.  So
//       sum_c v_c k_c  =  sigma * (sum_c code_c k_c)  +  u * (sum_c eta_c k_c)
//   is two int8 dot products: the K operand is duplicated into both halves of the 64-deep MFMA and the Q operand holds the
//   codes in the lower half for the first product and the eta plane in the upper half for the second.  The two integers are
//   combined in fp64 (exact: < 2^53) and rounded once to fp32 - the canonical reading of the reference's fp32 matmul.
//   Window partition, cyclic shift and their inverses are a row-index table; the shifted-window mask is a region-id table
//   (different regions -> -100, i.e. the clamp entry 256 of the exp table after max subtraction; padding keys use the zero
//   entry 257); the relative-position index is linear in the token coordinates: lin_i - lin_j + const.
// ---------------------------------------------------------------------------------------------------
#define WA_HD 32
#define WA_KEYS 64
// NT = ws*ws when known at compile time (49 for the 7x7 windows of every Swin variant), 0 = generic.  A score slot (kb, r) holds
// key kb*16 + 4g + r: when kb*16 + r >= NT it is padding in every lane and its arithmetic is dropped (3 of 16 slots at NT = 49).
#define WA_DEAD(kb, r) (NT > 0 && (kb) * 16 + (r) >= NT)
// ... and when kb*16 + 12 + r < NT it is a real key in every lane: no padding test
#define WA_PAD(kb, r) (!(NT > 0 && (kb) * 16 + 12 + (r) < NT))
template <int NT, bool TAP>
__global__ __launch_bounds__(256, 3) void k_window_attention(WinAttnArgs a) {
  constexpr int VSTRIDE = WA_KEYS + 4;                                         // bf16 elements
  __shared__ __attribute__((aligned(16))) int8_t sK[4][WA_KEYS * WA_HD];
  __shared__ __attribute__((aligned(16))) unsigned short sVt[4][WA_HD * VSTRIDE];
  __shared__ float sT[4][232];                                                // bias-table column of the head ((2*8-1)^2 = 225 max), times s_table / s_q2
  __shared__ __attribute__((aligned(16))) unsigned short sP[4][4][16];          // NT = 49: probabilities of the lone 49th query, see the tail block
  __shared__ unsigned short sMeta[WA_KEYS + 16];                               // per token: lin (y*(2ws-1)+x) | region << 10
  __shared__ long long lutE[258];
  __shared__ double lutFR[258];                                                // fp64 reciprocal of float(exp_int)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int ws = a.wa.ws, N = ws * ws, nW = a.wa.n_windows;
  const int hgroups = (a.H + 3) >> 2;
  const int blk = blockIdx.x;
  const int hg = blk % hgroups, w = (blk / hgroups) % nW, b = blk / (hgroups * nW);
  const int head = hg * 4 + wave;
  const int C = a.H * WA_HD;
  const long long ldq = a.wa.qkv_stride ? a.wa.qkv_stride : 3 * C, ldo = a.wa.out_stride ? a.wa.out_stride : C;
  // The kernel is short (four query blocks per wave) and its global loads form chains (row table -> K / V / Q rows): every load is
  // requested as early as its address is known and the arithmetic of the tables runs under the latency - row table, region ids and
  // the bias column first, then the exp table, then K / V and the first Q block, then the LDS stores
  const bool hok = head < a.H;
  const bool live = lane < N;
  const int rowj = a.wa.win_index[w * N + (live ? lane : 0)];                  // row of token `lane` of this window
  const int tsz = (2 * ws - 1) * (2 * ws - 1);
  int8_t tcode[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (hok && lane + 64 * i < tsz) tcode[i] = a.wa.table_codes[(lane + 64 * i) * a.H + head];
  int reg_t = 0;
  if (a.wa.region && tid < N) reg_t = (int)a.wa.region[w * N + tid];
  // exp table of the log-int-softmax (as in k_lis_attention); entry 256 = clamp value (masked pairs), 257 = padding
  for (int t = tid; t < 258; t += (int)blockDim.x) {
    int xi = -t;
    const int lim = 32 * a.wa.x0_int;
    xi = (xi < lim || t >= 256) ? lim : xi;
    const int q = xi / a.wa.x0_int;
    const int r = xi - a.wa.x0_int * q;
    const long long z = (long long)r * (r + a.wa.b_int) + a.wa.c_int;
    long long e = z << (32 - q);
    e = e < 0 ? 0 : e;
    if (t == 257) e = 0;
    const float ef = t == 257 ? 1.0f : (float)e;
    lutE[t] = e;
    lutFR[t] = 1.0 / (double)ef;
  }
  if (tid < WA_KEYS + 16) {
    const int t = tid < N ? tid : 0;
    sMeta[tid] = (unsigned short)(((t / ws) * (2 * ws - 1) + (t % ws)) | (reg_t << 10));
  }
  // K rows (int8) and V rows of the window's tokens (rows >= N are zero) and the Q fragment of the first query block
  const int8_t* hbase = a.qkv + (long long)b * a.T * ldq + head * WA_HD;
  uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0, v0 = k0, v1 = k0;
  if (hok && live) {
    const int8_t* base = hbase + (long long)rowj * ldq;
    k0 = *reinterpret_cast<const uint4*>(base + C);
    k1 = *reinterpret_cast<const uint4*>(base + C + 16);
    v0 = *reinterpret_cast<const uint4*>(base + 2 * C);
    v1 = *reinterpret_cast<const uint4*>(base + 2 * C + 16);
  }
  int rowq_next = __shfl(rowj, l15 < N ? l15 : N - 1);
  v4i qc_next = {0, 0, 0, 0};
  if (hok) qc_next = *reinterpret_cast<const v4i*>(hbase + (long long)rowq_next * ldq + (g & 1) * 16);
  const float inv_sa = 1.0f / a.wa.s_attn, inv_s2 = 1.0f / a.wa.s_q2;          // powers of two: exact
  // qact2((a1 * s_attn + code * s_table)) = clamp(rint(fma(a1, s_attn / s_q2, code * s_table / s_q2))): the products are exact (powers of
  // two) and the one rounding of the sum is the reference's, so the bias column is staged already scaled
  const float tb_mul = a.wa.s_table * inv_s2, a1_mul = a.wa.s_attn * inv_s2;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (lane + 64 * i < tsz) sT[wave][lane + 64 * i] = (float)tcode[i] * tb_mul;
  {
    *reinterpret_cast<uint4*>(&sK[wave][lane * WA_HD]) = k0;
    *reinterpret_cast<uint4*>(&sK[wave][lane * WA_HD + 16]) = k1;
    const unsigned vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int c = 0; c < WA_HD; ++c) {
      const float f = (float)sx8(vw[c >> 2], c & 3);
      sVt[wave][c * VSTRIDE + lane] = (unsigned short)(__float_as_uint(f) >> 16);   // exact bf16
    }
  }
  __syncthreads();
  if (!hok) return;
  const float sigma = a.wa.s_q1 * a.wa.qk_scale;                               // exact (s_q1 = 2^e)
  const float inv_u = __uint_as_float((unsigned)(254 - (int)(__float_as_uint(sigma) >> 23) + 23) << 23);   // 1 / ulp(sigma)
  // score = u * X,  X = (sigma / u) * S1 + S2 an integer below 2^53 (u = ulp(sigma) * s_q1 of the keys, a power of two): RN32(u * X) = u * RN32(X)
  const float sig_m = sigma * inv_u;                                           // the 24-bit significand of sigma as an integer
  const double sig_int = (double)sig_m;
  const float x_mul = ((1.0f / inv_u) * a.wa.s_q1) * inv_sa;                   // u / s_attn: a power of two
  const float m100 = (float)(int)(100.0f * inv_s2);                            // 100 / sf as an integer
  const float av_mul = a.wa.s_q1 / a.wa.s_q3;
  const int c0 = (ws - 1) * (2 * ws - 1) + (ws - 1);
  const int nqb = (N + 15) >> 4;
  // per score slot of this lane: relative-position term and region of its key (the same for every query block)
  int linj[4][4];
  unsigned regj[4][4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned mj = sMeta[kb * 16 + 4 * g + r];
      linj[kb][r] = (int)(mj & 1023u);
      regj[kb][r] = mj >> 10;
    }
  for (int qb = 0; qb < nqb; ++qb) {
    const int qi = qb * 16 + l15;
    const int qr = qi < N ? qi : N - 1;
    const int rowq = rowq_next;
    const v4i qc = qc_next;
    if (qb + 1 < nqb) {                                                        // the next block's Q fragment, a block ahead
      const int qn = qi + 16 < N ? qi + 16 : N - 1;
      rowq_next = __shfl(rowj, qn);
      qc_next = *reinterpret_cast<const v4i*>(hbase + (long long)rowq_next * ldq + (g & 1) * 16);
    }
    // eta plane: the lanes of g >= 2 need it, and the lane 32 below holds the same sixteen codes - each computes eight (g < 2: dwords 0-1,
    // g >= 2: dwords 2-3) and the lower half hands its two dwords up (v_permlane32_swap).  In units of u: eta = RN32(code * m) - code * m
    // with m = sigma / u, the significand of sigma as an integer (scaling by a power of two commutes with the rounding); the integral
    // float goes into its byte by the magic addition of pack4_pre
    unsigned eh[2];
    {
      const unsigned qh[2] = {(unsigned)(g < 2 ? qc[0] : qc[2]), (unsigned)(g < 2 ? qc[1] : qc[3])};
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        float et[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float cf = (float)sx8(qh[d], e);
          const float v = cf * sig_m;                                           // RN32(code * m)
          et[e] = __builtin_fmaf(-cf, sig_m, v);                                // exact, an integer in [-64, 64]
        }
        eh[d] = pack4_pre(et[0], et[1], et[2], et[3]);
      }
    }
    const auto up0 = __builtin_amdgcn_permlane32_swap(0u, eh[0], false, false);   // [0]: upper half := eh of the lower half
    const auto up1 = __builtin_amdgcn_permlane32_swap(0u, eh[1], false, false);
    const v4i qeta = {(int)up0[0], (int)up1[0], (int)eh[0], (int)eh[1]};
    const v4i fq1 = g < 2 ? qc : (v4i){0, 0, 0, 0};
    const v4i fq2 = g < 2 ? (v4i){0, 0, 0, 0} : qeta;
    v4i s1[4], s2[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const v4i fk = *reinterpret_cast<const v4i*>(&sK[wave][(kb * 16 + l15) * WA_HD + (g & 1) * 16]);
      s1[kb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq1, (v4i){0, 0, 0, 0}, 0, 0, 0);
      s2[kb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq2, (v4i){0, 0, 0, 0}, 0, 0, 0);
    }
    const unsigned mi = sMeta[qr];
    const float* trow = &sT[wave][(int)(mi & 1023u) + c0];                     // bias entry of key j: trow[-lin_j]
    const unsigned reg_i = mi >> 10;
    v4i pb[2];
    if (NT == 49 && qb == 3) {
      // The last query block of a 7 x 7 window holds ONE query (the 49th): all sixteen query columns of the score tile are that query, so the
      // thirteen live score slots of a lane are shared out over its sixteen columns - lane (l15, g) finishes slot l15 only (one chain instead
      // of thirteen), the row max and the sum run over the 16-lane rows as well, and the probabilities return to the operand layout of the
      // P.V product through 128 bytes of LDS per wave
      const int t = l15 < 13 ? l15 : 12;
      int v1 = s1[0][0], v2 = s2[0][0];
#pragma unroll
      for (int u = 1; u < 13; ++u) {
        const int kb = u < 12 ? u >> 2 : 3, r = u < 12 ? u & 3 : 0;
        v1 = t == u ? s1[kb][r] : v1;
        v2 = t == u ? s2[kb][r] : v2;
      }
      const int j = t < 12 ? (t >> 2) * 16 + 4 * g + (t & 3) : 48 + 4 * g;
      const bool valid = l15 < 13 && j < N;
      const unsigned mj = sMeta[j];
      const double X = __builtin_fma(sig_int, (double)v1, (double)v2);
      const float a1 = __builtin_amdgcn_fmed3f(rintf((float)X * x_mul), -128.f, 127.f);
      const float a2 = __builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(a1, a1_mul, trow[-(int)(mj & 1023u)])), -128.f, 127.f);
      float xi = a2;
      if (a.wa.region) xi -= (mj >> 10) != reg_i ? m100 : 0.f;
      xi = valid ? xi : -3.0e9f;
      float mx = xi;
#define WA_ROWMAX(ctrl) mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), ctrl, 0xF, 0xF, false)));
      WA_ROWMAX(0xB1) WA_ROWMAX(0x4E) WA_ROWMAX(0x141) WA_ROWMAX(0x140)          // as half_wave_sum: the 16 lanes of a row
#undef WA_ROWMAX
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int d = valid ? (int)fminf(mx - xi, 256.f) : 257;
      long long S = lutE[d];
#pragma unroll
      for (int o_ = 1; o_ < 64; o_ <<= 1) S += __shfl_xor(S, o_);
      const double Sd = (double)(float)S;
      const float ratio = rintf((float)(Sd * lutFR[d]));
      const int E = (int)((__float_as_uint(ratio) + 0x00400000u) >> 23);         // biased exponent of 2^k
      if (TAP && valid)
        a.probs_k[((((long long)b * nW + w) * a.H + head) * N + (N - 1)) * N + j] = (int8_t)(E - 127 > 16 ? 16 : E - 127);
      sP[wave][g][l15] = (unsigned short)((E < 143 && l15 < 13) ? (254 - E) << 7 : 0);     // 2^-k as bf16, 0 from k = 16 (and for padding: sum / 1 >= 2^32)
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      pb[0] = *reinterpret_cast<const v4i*>(&sP[wave][g][0]);                  // slots 0-7 = key blocks 0, 1;  8-15 = key blocks 2, 3 (13-15: zero)
      pb[1] = *reinterpret_cast<const v4i*>(&sP[wave][g][8]);
    } else {
      float xs[4][4];
      float mx = -3.0e9f;
      auto scores = [&](auto MASKc) {
        constexpr bool MASK = decltype(MASKc)::value;                            // shifted windows: pairs from different regions get -100
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (WA_DEAD(kb, r)) continue;
            const double X = __builtin_fma(sig_int, (double)s1[kb][r], (double)s2[kb][r]);         // exact
            const float a1 = __builtin_amdgcn_fmed3f(rintf((float)X * x_mul), -128.f, 127.f);     // ONE rounding, then qact_attn1
            const float a2 = __builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(a1, a1_mul, trow[-linj[kb][r]])), -128.f, 127.f);   // qact2
            float xi = a2;
            if (MASK) xi -= regj[kb][r] != reg_i ? m100 : 0.f;
            if (WA_PAD(kb, r)) xi = kb * 16 + 4 * g + r < N ? xi : -3.0e9f;
            xs[kb][r] = xi;
            mx = fmaxf(mx, xi);
          }
        }
      };
      if (a.wa.region) scores(std::integral_constant<bool, true>{});             // wave-uniform
      else scores(std::integral_constant<bool, false>{});
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      long long S = 0;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (WA_DEAD(kb, r)) continue;
          int d = (int)fminf(mx - xs[kb][r], 256.f);                             // integral values: exact
          if (WA_PAD(kb, r)) d = kb * 16 + 4 * g + r < N ? d : 257;
          s1[kb][r] = d;
          S += lutE[d];
        }
      S += __shfl_xor(S, 16);
      S += __shfl_xor(S, 32);
      const float Sf = (float)S;
      const double Sd = (double)Sf;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        unsigned pk[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          float ratio[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int jj = 2 * e2 + e;
            const int kb = 2 * p + (jj >> 2), r = jj & 3;
            if (WA_DEAD(kb, r)) {
              ratio[e] = 4.0e9f;                                             // -> probability 0
              continue;
            }
            // correctly rounded fp32 quotient, as in k_lis_attention; a padding key (entry 257: reciprocal 1) gives sum / 1 >= 2^32 -> k clamps -> 0
            ratio[e] = rintf((float)(Sd * lutFR[s1[kb][r]]));
            if (TAP && qi < N && kb * 16 + 4 * g + r < N) {
              const int k = (int)((__float_as_uint(ratio[e]) + 0x00400000u) >> 23) - 127;
              a.probs_k[((((long long)b * nW + w) * a.H + head) * N + qi) * N + kb * 16 + 4 * g + r] = (int8_t)(k > 16 ? 16 : k);
            }
          }
          // log_round and the clamp at 16 on the two high halves at once, see k_lis_attention
          const unsigned hi2 = __builtin_amdgcn_perm(__float_as_uint(ratio[1]), __float_as_uint(ratio[0]), 0x07060302u);
          const v2u16 eb = __builtin_bit_cast(v2u16, (__builtin_bit_cast(unsigned, __builtin_bit_cast(v2u16, hi2) + (v2u16){0x40, 0x40})) & 0x7F807F80u);
          const v2u16 hb = (v2u16){0x7F00, 0x7F00} - eb;                                  // (254 - E) << 7
          const v2i16 neg = __builtin_bit_cast(v2i16, (v2u16)(hb - (v2u16){0x3800, 0x3800})) >> (v2i16){15, 15};   // all ones where k >= 16
          pk[e2] = __builtin_bit_cast(unsigned, hb) & ~__builtin_bit_cast(unsigned, neg);
        }
        pb[p] = (v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
      }
    }
    v4f o[2] = {(v4f){0.f, 0.f, 0.f, 0.f}, (v4f){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const v8bf fb = __builtin_bit_cast(v8bf, pb[p]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned short* vp = &sVt[wave][(dt * 16 + l15) * VSTRIDE + p * 32 + 4 * g];
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
        v4i va = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, va), fb, o[dt], 0, 0, 0);
      }
    }
    if (qi < N) {
      int8_t* dst = a.out + ((long long)b * a.T + rowq) * ldo + head * WA_HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<unsigned*>(dst + dt * 16) =
            pack4_rne_sat(o[dt][0] * av_mul, o[dt][1] * av_mul, o[dt][2] * av_mul, o[dt][3] * av_mul);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// module-level helpers
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fake_quant_f32(const float* __restrict__ x, long long n, const float* __restrict__ scale,
                                                        int n_scale, long long inner, float lo, float hi,
                                                        float* __restrict__ out, int8_t* __restrict__ codes) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float s = scale[n_scale == 1 ? 0 : (i / inner) % n_scale];
    float q = rintf(x[i] / s);
    q = fminf(fmaxf(q, lo), hi);
    if (out) out[i] = q * s;
    if (codes) codes[i] = (int8_t)(int)q;
  }
}

__global__ __launch_bounds__(256) void k_gelu_quant_f32(const float* __restrict__ y, long long n, float inv_s,
                                                        int8_t* __restrict__ codes, unsigned long long* flags, int force_slow) {
  unsigned long long cnt = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    bool slow = false;
    codes[i] = (int8_t)gelu_q8(y[i], inv_s, force_slow != 0, &slow);
    cnt += slow ? 1 : 0;
  }
  if (flags && cnt) atomicAdd(flags, cnt);
}

// max |gelu_fast - gelu_exact| over a bit-pattern range of fp32 inputs (bound check of GELU_EPS)
__global__ __launch_bounds__(256) void k_gelu_err_sweep(unsigned first_bits, unsigned count, float* max_err) {
  float m = 0.f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float y = __uint_as_float(first_bits + i);
    const float e = fabsf(gelu_fast(y) - gelu_exact(y));
    m = e > m ? e : m;
  }
  for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(max_err), __float_as_uint(m));
}

// ---------------------------------------------------------------------------------------------------
// host launchers (called from the C ABI in p2vit_capi.cpp)
// ---------------------------------------------------------------------------------------------------
int g_attn_waves = 8;     // P2V_ATTN_WAVES
int g_gemm_stages = 3;    // P2V_GEMM_STAGES: 2 / 3 = LDS-DMA ring depth of k_gemm_dma
#ifdef P2V_DIAG
unsigned long long* g_gemm_stamps = nullptr;
#endif
#define CHECK_LAUNCH()                                     \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return (int)e_;                  \
  } while (0)

int p2v_launch_patchify(const float* img, int B, int C, int H, int W, int P, float inv_s, int8_t* out, int k_pad, hipStream_t st) {
  const long long rows = (long long)B * (H / P) * (W / P);
  const int rows_per_block = 8;
  hipLaunchKernelGGL(k_quantize_patchify, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, img, B, C, H, W, P,
                     inv_s, out, k_pad, rows_per_block);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_embed_fp32(const float* img, int B, int C, int H, int W, int P, const GemmArgs& g, hipStream_t st) {
  if (g.w4 || g.N % 4 || P % 4) return -1;
  const dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64));
  hipLaunchKernelGGL(k_embed_fp32, grid, dim3(256), 0, st, img, B, C, H, W, P, g);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_fill_cls(int8_t* x, int B, int T, int D, const int8_t* cls, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_cls, dim3((B * D + 255) / 256), dim3(256), 0, st, x, B, T, D, cls);
  CHECK_LAUNCH();
  return 0;
}

int g_gemm_tile = 0;      // P2V_GEMM_TILE: 0 = by grid size, 128 / 256 = force the tile height of the layer GEMMs
static int device_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t pr;
    cus[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return cus[dev];
}
// dynamic LDS (the GELU table) of a tiled-GEMM instantiation beyond what it has been granted so far on this device
template <typename K>
static bool grant_dynamic_lds(K kernel, int slot, int bytes) {
  static int granted[16][8] = {{0}};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  if (dev >= 0 && bytes <= granted[dev][slot]) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (dev >= 0) granted[dev][slot] = bytes;
  return true;
}

int p2v_launch_gemm(int epi, const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
#ifdef P2V_DIAG
  g.stamps = g_gemm_stamps;
#endif
  g.tiles_n = (g.N + GBN - 1) / GBN;
  if (epi != P2V_EPI_HEAD && epi != P2V_EPI_EMBED) {
    // the tiled kernel addresses both matrices with 32-bit lane offsets
    if ((long long)g.M * g.lda + g.K >= (1LL << 32) || (long long)g.tiles_n * GBN * g.K >= (1LL << 32)) return -1;
    // 256-row tiles (8 waves, two workgroups per CU) when the grid still gives every CU its two workgroups; else 128-row tiles
    const long long tiles256 = (long long)((g.M + 255) / 256) * g.tiles_n;
    // (packed int4 weights keep 128 rows: the 8-wave form moves a 4 KB W tile as eight half-wave pieces and measured 3 % slower on DeiT-B W4)
    const bool big = g_gemm_stages == 3 && (g_gemm_tile == 256 || (g_gemm_tile == 0 && !g.w4 && tiles256 >= 2LL * device_cus()));
    const int tiles_m = big ? (g.M + 255) / 256 : (g.M + GBM - 1) / GBM;
    dim3 grid4(g.tiles_n * tiles_m), block4(big ? 512 : 256);
    unsigned tab_bytes = (epi == P2V_EPI_GELU && g.ep.gelu.table) ? (unsigned)g.ep.gelu.cells * 8u : 0u;
    // static LDS of the GELU_TAB instantiations (ring + column constants) plus the table can pass the 64 KB a kernel gets by default
    // (1/scale = 256: 2111 cells = 16.5 KB): ask for the larger dynamic block once per process and device, or use the arithmetic epilogue
    if (tab_bytes) {
      const int stat = (big ? 3 * (256 + GBN) * GBK : g_gemm_stages * DMA_STAGE_BYTES) + 2 * GBN * (int)sizeof(float);
      if (stat + (int)tab_bytes > 64 * 1024) {
        bool ok;
        if (big) ok = g.w4 ? grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, 3, true, 4>, 0, (int)tab_bytes)
                           : grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, 3, false, 4>, 1, (int)tab_bytes);
        else if (g.w4) ok = grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, 3, true, 2>, 2, (int)tab_bytes);
        else if (g_gemm_stages == 2) ok = grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, 2, false, 2>, 3, (int)tab_bytes);
        else ok = grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, 3, false, 2>, 4, (int)tab_bytes);
        if (!ok) tab_bytes = 0;                   // arithmetic P2V_EPI_GELU kernel: same codes, no table
      }
    }
#define P2V_LAUNCH_TILED(KERNEL)                                                                                          \
    switch (epi) {                                                                                                        \
      case P2V_EPI_REQUANT: hipLaunchKernelGGL(KERNEL(P2V_EPI_REQUANT), grid4, block4, 0, st, g); break;                  \
      case P2V_EPI_GELU:                                                                                                  \
        if (tab_bytes) hipLaunchKernelGGL(KERNEL(P2V_EPI_GELU_TAB), grid4, block4, tab_bytes, st, g);                     \
        else hipLaunchKernelGGL(KERNEL(P2V_EPI_GELU), grid4, block4, 0, st, g);                                           \
        break;                                                                                                            \
      case P2V_EPI_RESID: hipLaunchKernelGGL(KERNEL(P2V_EPI_RESID), grid4, block4, 0, st, g); break;                      \
      default: return -1;                                                                                                 \
    }
#define P2V_K_DMA2(E) (k_gemm_dma<E, 2, false, 2>)
#define P2V_K_DMA3(E) (k_gemm_dma<E, 3, false, 2>)
#define P2V_K_DMA3P(E) (k_gemm_dma<E, 3, true, 2>)
#define P2V_K_DMA3L(E) (k_gemm_dma<E, 3, false, 4>)
#define P2V_K_DMA3PL(E) (k_gemm_dma<E, 3, true, 4>)
    if (big) {
      if (g.w4) { P2V_LAUNCH_TILED(P2V_K_DMA3PL) }
      else { P2V_LAUNCH_TILED(P2V_K_DMA3L) }
    } else if (g.w4) { P2V_LAUNCH_TILED(P2V_K_DMA3P) }          // packed int4 weights: the LDS-DMA kernel only
    else if (g_gemm_stages == 2) { P2V_LAUNCH_TILED(P2V_K_DMA2) }
    else { P2V_LAUNCH_TILED(P2V_K_DMA3) }
#undef P2V_LAUNCH_TILED
    CHECK_LAUNCH();
    return 0;
  }
  // EMBED / HEAD: one launch each per forward; 8-wave shape (64x32 wave tiles, <= 128 VGPRs)
  const int tiles_m = (g.M + GBM - 1) / GBM;
  dim3 grid(g.tiles_n * tiles_m), block(512);
  if (epi == P2V_EPI_EMBED) {
    if (g.w4) hipLaunchKernelGGL((k_gemm_i8<P2V_EPI_EMBED, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((k_gemm_i8<P2V_EPI_EMBED, false>), grid, block, 0, st, g);
  } else {
    if (g.w4) hipLaunchKernelGGL((k_gemm_i8<P2V_EPI_HEAD, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((k_gemm_i8<P2V_EPI_HEAD, false>), grid, block, 0, st, g);
  }
  CHECK_LAUNCH();
  return 0;
}

#ifdef P2V_DIAG
extern unsigned long long* g_gemm_stamps;
#endif
int g_ln_generic = 0;     // P2V_LN_GENERIC=1
// LayerNorm + GEMM in one launch.  Returns -3 when the shape is outside what the fused kernel is instantiated for (callers then
// run p2v_launch_layernorm + p2v_launch_gemm).
int g_ln_gemm = 1;        // P2V_LN_GEMM=0: never fuse (A/B runs)
bool p2v_ln_gemm_supported(int epi, int C, int N, int table_cells) {
  if (!g_ln_gemm || (epi != P2V_EPI_REQUANT && epi != P2V_EPI_GELU)) return false;
  if (C % 4 || C > 384 || N % 16) return false;
  return ln_gemm_lds(C, N, ((C + GBK - 1) / GBK + 1) / 2, table_cells).total <= 80 * 1024;     // two workgroups per CU
}
template <int EPI, int KT, int VER, bool W4>
static int launch_ln_gemm_t2(const LnArgs& a, const GemmArgs& g, int cells, hipStream_t st) {
  const int smem = ln_gemm_lds(a.C, g.N, (KT + 1) / 2, cells).total;
  static int granted[16] = {0};                 // per device: the dynamic LDS size this instantiation has been allowed so far
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  const void* fn;
  if constexpr (VER == 3) fn = reinterpret_cast<const void*>(&k_ln_gemm2<EPI, KT, 2, W4>);
  else if constexpr (VER == 2) fn = reinterpret_cast<const void*>(&k_ln_gemm2<EPI, KT, 1, W4>);
  else fn = reinterpret_cast<const void*>(&k_ln_gemm<EPI, KT, W4>);
  if (dev < 0 || smem > granted[dev]) {         // (a racing second thread only repeats the call)
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) granted[dev] = smem;
  }
  const dim3 grid((unsigned)((g.M + LG_BM - 1) / LG_BM));
  if constexpr (VER == 3) hipLaunchKernelGGL((k_ln_gemm2<EPI, KT, 2, W4>), grid, dim3(512), (unsigned)smem, st, a, g);
  else if constexpr (VER == 2) hipLaunchKernelGGL((k_ln_gemm2<EPI, KT, 1, W4>), grid, dim3(256), (unsigned)smem, st, a, g);
  else hipLaunchKernelGGL((k_ln_gemm<EPI, KT, W4>), grid, dim3(256), (unsigned)smem, st, a, g);
  CHECK_LAUNCH();
  return 0;
}
template <int EPI, int KT, int VER>
static int launch_ln_gemm_t(const LnArgs& a, const GemmArgs& g, int cells, hipStream_t st) {
  return g.w4 ? launch_ln_gemm_t2<EPI, KT, VER, true>(a, g, cells, st) : launch_ln_gemm_t2<EPI, KT, VER, false>(a, g, cells, st);
}
int g_ln_gemm_ver = 2;    // P2V_LN_GEMM_V=1: the 4-wave kernel of round 2 for every launch (A/B runs; same results)
// g0.W must point to the FRAGMENT-ORDER copy of the weights (p2v_linear.w_frag; packed two codes per byte when g0.w4)
int p2v_launch_ln_gemm(int epi, const LnArgs& a_, const GemmArgs& g0, hipStream_t st) {
  const int cells = (epi == P2V_EPI_GELU && g0.ep.gelu.table) ? g0.ep.gelu.cells : 0;
  if (!p2v_ln_gemm_supported(epi, a_.C, g0.N, cells)) return -3;
  if (a_.row_stride < 0 || a_.row_stride > (1 << 24)) return -3;       // the kernels address a workgroup's 64 rows with 32-bit offsets
  LnArgs a = a_;
  a.force_generic = g_ln_generic;
  GemmArgs g = g0;
  g.tiles_n = (g.N + GBN - 1) / GBN;
#ifdef P2V_DIAG
  g.stamps = g_gemm_stamps;
#endif
  const int kt = (a.C + GBK - 1) / GBK;
#define P2V_LG(EPI_, VER_)                                                        \
  switch (kt) {                                                                   \
    case 1: return launch_ln_gemm_t<EPI_, 1, VER_>(a, g, cells, st);              \
    case 2: return launch_ln_gemm_t<EPI_, 2, VER_>(a, g, cells, st);              \
    case 3: return launch_ln_gemm_t<EPI_, 3, VER_>(a, g, cells, st);              \
    case 4: return launch_ln_gemm_t<EPI_, 4, VER_>(a, g, cells, st);              \
    case 5: return launch_ln_gemm_t<EPI_, 5, VER_>(a, g, cells, st);              \
    default: return launch_ln_gemm_t<EPI_, 6, VER_>(a, g, cells, st);             \
  }
  // the dense 8-wave kernel: REQUANT and table GELU without activation taps; everything else runs the 4-wave kernel
  if (g_ln_gemm_ver == 3 && !g.ep.tap_out) {
    if (epi == P2V_EPI_REQUANT) { P2V_LG(P2V_EPI_REQUANT, 3) }
    if (cells) { P2V_LG(P2V_EPI_GELU_TAB, 3) }
  }
  if (g_ln_gemm_ver == 2 && !g.ep.tap_out) {
    if (epi == P2V_EPI_REQUANT) { P2V_LG(P2V_EPI_REQUANT, 2) }
    if (cells) { P2V_LG(P2V_EPI_GELU_TAB, 2) }
  }
  if (epi == P2V_EPI_REQUANT) { P2V_LG(P2V_EPI_REQUANT, 1) }
  if (cells) { P2V_LG(P2V_EPI_GELU_TAB, 1) }
  P2V_LG(P2V_EPI_GELU, 1)
#undef P2V_LG
}

int g_ln_rows = 4;        // P2V_LN_ROWS: consecutive rows per half wave
int p2v_launch_layernorm(const LnArgs& a_, hipStream_t st) {
  LnArgs a = a_;
  a.force_generic = g_ln_generic;
  a.rows_per_half = g_ln_rows;
  const int LN_ROWS = g_ln_rows;
  // one row per WAVE above 384 channels (64 lanes x 4 channels x up to 8 groups = 2048 channels): the per-lane constants of a half-wave
  // row cost ~45 VGPRs per 128 channels (C = 768: 256 VGPRs, one wave per SIMD; a wave per row: 161, three)
  const bool wide = a.C > 384;
  const int nch = wide ? (a.C + 255) / 256 : (a.C + 127) / 128;
  const int rows_per_block = (wide ? 4 : 8) * LN_ROWS;
  dim3 grid((unsigned)((a.rows + rows_per_block - 1) / rows_per_block)), block(256);
  if (wide) {
    switch (nch) {
      case 2: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<2, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<2, 64, false>), grid, block, 0, st, a); break;
      case 3: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<3, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<3, 64, false>), grid, block, 0, st, a); break;
      case 4: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<4, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<4, 64, false>), grid, block, 0, st, a); break;
      case 5: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<5, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<5, 64, false>), grid, block, 0, st, a); break;
      case 6: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<6, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<6, 64, false>), grid, block, 0, st, a); break;
      case 7: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<7, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<7, 64, false>), grid, block, 0, st, a); break;
      case 8: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<8, 64, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<8, 64, false>), grid, block, 0, st, a); break;
      default: return -1;
    }
  } else {
    switch (nch) {
      case 1: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<1, 32, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<1, 32, false>), grid, block, 0, st, a); break;
      case 2: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<2, 32, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<2, 32, false>), grid, block, 0, st, a); break;
      case 3: if (a.pre.gm) hipLaunchKernelGGL((k_int_layernorm<3, 32, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_int_layernorm<3, 32, false>), grid, block, 0, st, a); break;
      default: return -1;
    }
  }
  CHECK_LAUNCH();
  return 0;
}

template <int HD, int NKB>
static int launch_attn_t(const AttnArgs& a_, hipStream_t st) {
  constexpr int KROWS = NKB * 32;   // NKB here = 32-key pairs
  constexpr size_t smem = (size_t)KROWS * HD + (size_t)HD * (KROWS + 4) * 2 + 258 * 8 + 258 * 8;
  AttnArgs a = a_;
  {   // score multiplier 2^-p with p >= 1: the integer round-half-even path (|score| <= 64 * 128 * 128 < 2^21, p <= 24)
    int ex;
    const float m = a.at.qk_scale * (a.at.s_qkv_sq * a.at.inv_s_attn);
    a.pshift = (m > 0.f && frexpf(m, &ex) == 0.5f && ex <= 0 && ex >= -23) ? 1 - ex : 0;
  }
#ifdef P2V_DIAG
  a.stamps = g_gemm_stamps;
#endif
  const dim3 grid(a.B * a.H), block(64 * g_attn_waves);
#define P2V_ATTN_LAUNCH(TAP_, ISH_)                                                                                          \
  do {                                                                                                                       \
    if (smem > 64 * 1024) {          /* K and V^T of more than ~300 keys: beyond the default dynamic LDS limit */            \
      static bool granted[16] = {false};                                                                                     \
      int dev = 0;                                                                                                           \
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;                                                \
      if (dev < 0 || !granted[dev]) {                                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lis_attention<HD, NKB, TAP_, ISH_>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                           \
        if (e != hipSuccess) return (int)e;                                                                                  \
        if (dev >= 0) granted[dev] = true;                                                                                   \
      }                                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((k_lis_attention<HD, NKB, TAP_, ISH_>), grid, block, smem, st, a);                                    \
  } while (0)
  if (a.probs_k) {
    if (a.pshift) P2V_ATTN_LAUNCH(true, true);
    else P2V_ATTN_LAUNCH(true, false);
  } else {
    if (a.pshift) P2V_ATTN_LAUNCH(false, true);
    else P2V_ATTN_LAUNCH(false, false);
  }
#undef P2V_ATTN_LAUNCH
  CHECK_LAUNCH();
  return 0;
}

// PatchMerging gather: 16-byte chunks, one thread each (C % 16 == 0)
__global__ __launch_bounds__(256) void k_patch_merge_gather(const int8_t* __restrict__ x, int B, int H, int W, int C, int8_t* __restrict__ out) {
  const int cpr = C / 16, H2 = H / 2, W2 = W / 2;
  const long long total = (long long)B * H2 * W2 * 4 * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    long long r = i / cpr;
    const int q = (int)(r % 4); r /= 4;               // x0..x3: q = 0 (dy0,dx0), 1 (dy1,dx0), 2 (dy0,dx1), 3 (dy1,dx1)
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2);
    const int b = (int)(r / H2);
    const int dy = q & 1, dx = q >> 1;
    const uint4 v = *reinterpret_cast<const uint4*>(x + (((long long)b * H + 2 * h2 + dy) * W + 2 * w2 + dx) * C + ch * 16);
    *reinterpret_cast<uint4*>(out + (((long long)b * H2 + h2) * W2 + w2) * 4 * C + q * C + ch * 16) = v;
  }
}

// AdaptiveAvgPool1d over tokens + qact3: one thread per (image, 4 channels)
__global__ __launch_bounds__(256) void k_avgpool_quant(const int8_t* __restrict__ x, int B, int T, int C, float s_in, float inv_s_out,
                                                       int8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  if (i >= B * c4) return;
  const int b = i / c4, c = (i % c4) * 4;
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int t = 0; t < T; ++t) {
    const unsigned w = *reinterpret_cast<const unsigned*>(x + ((long long)b * T + t) * C + c);
    s0 += sx8(w, 0); s1 += sx8(w, 1); s2 += sx8(w, 2); s3 += sx8(w, 3);
  }
  const float Tf = (float)T;
  *reinterpret_cast<unsigned*>(out + (long long)b * C + c) =
      pack4_sat(rintf((((float)s0 * s_in) / Tf) * inv_s_out), rintf((((float)s1 * s_in) / Tf) * inv_s_out),
                rintf((((float)s2 * s_in) / Tf) * inv_s_out), rintf((((float)s3 * s_in) / Tf) * inv_s_out));
}

int p2v_launch_patch_merge_gather(const int8_t* x, int B, int H, int W, int C, int8_t* out, hipStream_t st) {
  const long long total = (long long)B * (H / 2) * (W / 2) * 4 * (C / 16);
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 256 * 32 ? 256 * 32 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(k_patch_merge_gather, dim3(blocks), dim3(256), 0, st, x, B, H, W, C, out);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_avgpool_quant(const int8_t* x, int B, int T, int C, float s_in, float inv_s_out, int8_t* out, hipStream_t st) {
  hipLaunchKernelGGL(k_avgpool_quant, dim3((B * (C / 4) + 255) / 256), dim3(256), 0, st, x, B, T, C, s_in, inv_s_out, out);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_window_attention(const WinAttnArgs& a, hipStream_t st) {
  const int hgroups = (a.H + 3) / 4;
  const dim3 grid((unsigned)(a.B * a.wa.n_windows * hgroups));
  if (a.wa.ws == 7) {
    if (a.probs_k) hipLaunchKernelGGL((k_window_attention<49, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_window_attention<49, false>), grid, dim3(256), 0, st, a);
  } else {
    if (a.probs_k) hipLaunchKernelGGL((k_window_attention<0, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_window_attention<0, false>), grid, dim3(256), 0, st, a);
  }
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_attention(const AttnArgs& a, int head_dim, hipStream_t st) {
  const int nkb = (a.N + 31) / 32;
  {   // the kernel folds s_q1^2 / s_attn into qk_scale: exact only for a power of two (both are PoT scales in the reference)
    int ex;
    const float m2 = a.at.s_qkv_sq * a.at.inv_s_attn;
    if (!(m2 > 0.f) || frexpf(m2, &ex) != 0.5f) return -2;
  }
  // every ceil(tokens / 32) up to 19 (608 tokens) is instantiated: the padding of a launch is always less than one 32-key pair
#define P2V_ATTN_CASES(HD_)                                                                                                  \
  switch (nkb) {                                                                                                             \
    case 1: return launch_attn_t<HD_, 1>(a, st);   case 2: return launch_attn_t<HD_, 2>(a, st);                              \
    case 3: return launch_attn_t<HD_, 3>(a, st);   case 4: return launch_attn_t<HD_, 4>(a, st);                              \
    case 5: return launch_attn_t<HD_, 5>(a, st);   case 6: return launch_attn_t<HD_, 6>(a, st);                              \
    case 7: return launch_attn_t<HD_, 7>(a, st);   case 8: return launch_attn_t<HD_, 8>(a, st);                              \
    case 9: return launch_attn_t<HD_, 9>(a, st);   case 10: return launch_attn_t<HD_, 10>(a, st);                            \
    case 11: return launch_attn_t<HD_, 11>(a, st); case 12: return launch_attn_t<HD_, 12>(a, st);                            \
    case 13: return launch_attn_t<HD_, 13>(a, st); case 14: return launch_attn_t<HD_, 14>(a, st);                            \
    case 15: return launch_attn_t<HD_, 15>(a, st); case 16: return launch_attn_t<HD_, 16>(a, st);                            \
    case 17: return launch_attn_t<HD_, 17>(a, st); case 18: return launch_attn_t<HD_, 18>(a, st);                            \
    case 19: return launch_attn_t<HD_, 19>(a, st);                                                                           \
    default: return -1;                                                                                                      \
  }
  if (head_dim == 64) { P2V_ATTN_CASES(64) }
  if (head_dim == 32) { P2V_ATTN_CASES(32) }
#undef P2V_ATTN_CASES
  return -1;
}

int p2v_launch_fake_quant(const float* x, long long n, const float* scale, int n_scale, long long inner, int lo, int hi,
                          float* out, int8_t* codes, hipStream_t st) {
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_fake_quant_f32, dim3((unsigned)blocks), dim3(256), 0, st, x, n, scale, n_scale, inner, (float)lo,
                     (float)hi, out, codes);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gelu_quant(const float* y, long long n, float inv_s, int8_t* codes, unsigned long long* flags, int force_slow,
                          hipStream_t st) {
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_gelu_quant_f32, dim3((unsigned)blocks), dim3(256), 0, st, y, n, inv_s, codes, flags, force_slow);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gelu_sweep(unsigned first_bits, unsigned count, float* max_err, hipStream_t st) {
  hipLaunchKernelGGL(k_gelu_err_sweep, dim3(2048), dim3(256), 0, st, first_bits, count, max_err);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gelu_table_build(float inv_s, const p2v_gelu_tab& t, unsigned* scratch, hipStream_t st) {
  const int cells = t.cells;
  // scratch: cnt | thr | lohi | first | status
  hipError_t e = hipMemsetAsync(scratch, 0, (size_t)3 * cells * 4, st);
  if (e == hipSuccess) e = hipMemsetAsync(scratch + 3 * cells, 0xFF, (size_t)cells * 4, st);
  if (e == hipSuccess) e = hipMemsetAsync(scratch + 4 * cells, 0, 4, st);
  if (e != hipSuccess) return (int)e;
  const int per_thread = 2048;
  const unsigned long long threads = (2 * P2V_F32_FINITE + per_thread - 1) / per_thread;
  hipLaunchKernelGGL(k_gelu_tab_sweep, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, inv_s, t.k, t.off, (float)(cells - 1), cells,
                     scratch, per_thread);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_gelu_tab_finish, dim3((cells + 255) / 256), dim3(256), 0, st, cells, scratch,
                     reinterpret_cast<uint2*>(const_cast<void*>(t.table)), scratch + 4 * cells);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gelu_table_check(float inv_s, const p2v_gelu_tab& t, unsigned long long* mismatches, hipStream_t st) {
  hipLaunchKernelGGL(k_gelu_tab_check, dim3(8192), dim3(256), 0, st, inv_s, t.k, t.off, (float)(t.cells - 1),
                     reinterpret_cast<const unsigned char*>(t.table), mismatches);
  CHECK_LAUNCH();
  return 0;
}
