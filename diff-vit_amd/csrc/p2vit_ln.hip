// p2vit_ln.hip -- integer LayerNorm (stand-alone) and the fused LayerNorm + GEMM kernels, with their launchers.
#include "p2vit_epilogue.h"

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
// The lane-resident PTF mask is kept TIMES 2^-16 (LN_XS), so x_q = code * mask arrives scaled by 2^-16 for free: the fast chain multiplies it by
// T * 2^16 (below), the partial sums are scaled back once per lane and row (powers of two: every product, sum and rounding is unchanged).
#define LN_XS 0x1p-16f
#define LN_XS_INV 65536.f
__device__ __forceinline__ float4 ln_scale_mask(float4 m) { return make_float4(m.x * LN_XS, m.y * LN_XS, m.z * LN_XS, m.w * LN_XS); }
template <int NCH, bool LDSC = false>
struct LnLane {
  bool on[NCH];
  float4 gm[NCH], bt[NCH];            // gamma*io, beta*io
  float4 pm[LDSC ? 1 : NCH];          // post_mul
  float4 mkf[LDSC ? 1 : NCH];         // PTF mask (in_scale / s1): 1, 2, 4 or 8 - times LN_XS
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
    *reinterpret_cast<float4*>(sM + c) = ln_scale_mask(mk);
  }
  L.pot = __syncthreads_and(potf) != 0;
  L.pm_one = __syncthreads_and(pm1) != 0;
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
  // the lane's partial sums in fp32: |x_q| <= 1024, so sum x_q of 32 values and sum x_q^2 of 16 values (<= 2^24) are exact - full-rate
  // add / fma instead of 24-bit multiplies and three-operand adds (profiles/r03_op_cost.txt); the cross-lane sums stay integers.
  // xq[][] holds x_q * 2^-16 (the mask is stored scaled, LN_XS): the sums come out times 2^-16 / 2^-32 and are scaled back per lane
  float S1p = 0.f, S2p = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const unsigned w = wcur[i];
    const float4 mk_ = L.mask4(i);
    const float m4[4] = {mk_.x, mk_.y, mk_.z, mk_.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[i][j] = (float)sx8(w, j) * m4[j];       // x_q * in_scale_mask  (layers.py:269-273) * 2^-16, exact; w == 0 past C
    S1p += (xq[i][0] + xq[i][1]) + (xq[i][2] + xq[i][3]);                  // (short dependency chains: a wave may be alone on its SIMD)
    S2p += __builtin_fmaf(xq[i][1], xq[i][1], xq[i][0] * xq[i][0]) + __builtin_fmaf(xq[i][3], xq[i][3], xq[i][2] * xq[i][2]);
    if ((i & 3) == 3 || i == NCH - 1) {
      S2 += (unsigned)(S2p * (LN_XS_INV * LN_XS_INV));
      S2p = 0.f;
    }
  }
  S1 = (int)(S1p * LN_XS_INV);
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
    // A2 = A * 2^16 (rs scaled once per row): sign * M * 2^-N * 2^16 = A2 with its low 16 mantissa bits cleared multiplies the SCALED x_q
    // (xq[][] = x_q * 2^-16, LN_XS) - the same exact product as before; t = beta*io - mos*gamma*io is unchanged; and the magic constant of the
    // offset rounding, 1.5 * 2^(23-N), has exactly A2's exponent field: ONE v_and_or_b32 (round 3: and + add of 16 to the field)
    const float rs2 = rs * LN_XS_INV;
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
            const float A2 = rs2 * g4[j + e];                                       // A * 2^16 (exact scaling)
            const float t = b4[j + e] - mos * g4[j + e];
            const unsigned Ab = __float_as_uint(A2);
            T2[e] = __uint_as_float(Ab & 0xFFFF0000u);                              // sign * M * 2^-N * 2^16
            // Bv * 2^-N = t rounded to the 2^-N grid, N = 134 - exp(A): C = 1.5 * 2^(23-N) has the exponent field exp(A) + 16 = exp(A2)
            // (3 full-rate instructions instead of bfe, add, sub, ldexp, rndne, ldexp: tools/ubench/op_cost.hip, profiles/r03_op_cost.txt)
            unsigned Cmb;       // (Ab & 0x7F800000) | 0x00400000 as ONE instruction (hipcc emits and + or: a VOP3 takes no literal, so the two
                                // constants sit in a scalar and a vector register)
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(Cmb) : "v"(Ab), "s"(0x7F800000u), "v"(0x00400000u));
            const float Cm = __uint_as_float(Cmb);
            Bq2[e] = (t + Cm) - Cm;
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
      for (int j = 0; j < 4; ++j) q[j] = ln_elem_generic(xq[i][j] * LN_XS_INV, g4[j], b4[j], i4[j], p4[j], rs, mos, o4[j]);
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
  // A workgroup lives for only a few rows (16 - 32): its first rows are requested BEFORE the constants are staged, so that one memory round
  // trip covers both (round 4; before, the row loads were issued behind the prologue's barrier: two exposed round trips per workgroup)
  const int LN_ROWS = a.rows_per_half;
  const long long row0 = ((long long)blockIdx.x * (256 / LANES) + hw) * LN_ROWS;
  const long long last_row = a.rows - 1;
  constexpr int R = LN_BATCH;                                      // rows per batch of ln_rows (the row scalars are computed once per batch)
  int colofs[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) colofs[i] = (l32 + LANES * i) * 4 < a.C ? (l32 + LANES * i) * 4 : 0;     // clamped: loads are unconditional
  unsigned wnext[R][NCH];
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      wnext[u][i] = *reinterpret_cast<const unsigned*>(a.x + (row0 + u < a.rows ? row0 + u : last_row) * a.row_stride + colofs[i]);
  if constexpr (PRE) {      // only post_mul and the mask go through LDS (LN_LDSC), one barrier; the folded constants are requested first
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (l32 + LANES * i) * 4;
      L.on[i] = c < a.C;
      L.gm[i] = *reinterpret_cast<const float4*>(a.pre.gm + c);
      L.bt[i] = *reinterpret_cast<const float4*>(a.pre.bt + c);
    }
    for (int t4 = tid; t4 < NCH * LANES; t4 += 256) {
      const int c = t4 * 4;
      float4 pmv = make_float4(0.f, 0.f, 0.f, 0.f), mk = pmv;
      if (c < a.C) {
        pmv = *reinterpret_cast<const float4*>(a.ln.post_mul + c);
        mk = *reinterpret_cast<const float4*>(a.ln.mask + c);
      }
      *reinterpret_cast<float4*>(sP + c) = pmv;
      *reinterpret_cast<float4*>(sM + c) = ln_scale_mask(mk);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (l32 + LANES * i) * 4;
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
  } else {
    ln_prepare<NCH, LANES>(a.ln, a.C, a.force_generic != 0, sG, sB, sP, sM, tid, 256, L);
  }
  // Row r+1 is requested at the top of the iteration of row r and first touched just before the stores of row r.  The two
  // empty asm statements pin that placement: left alone, hipcc sinks the loads of a loop-carried value to the loop end,
  // behind the stores, and waits vmcnt(0) there - two exposed memory round trips per row (measured: 3 us per row).
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
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : (EPI == P2V_EPI_GELU_TAB ? g.ep.gelu.k : 1.0f);   // GELU_TAB: u = y * k, see gelu_tab_offset
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
  // at C = 384 the LayerNorm phase (48 registers of per-channel constants, two rows in flight) and a whole tile of W fragments (48) do not fit
  // 256 registers together: hipcc spilled one fragment to scratch and reloaded it after the phase.  The LAST fragment of the first tile - used
  // eleven k-steps after the phase ends - is therefore requested after the LayerNorm phase
  constexpr int NI_EARLY = KT >= 6 ? NI - 1 : NI;
  wraw wf[NI];
  {
    const wraw* w0 = wtile(0);
#pragma unroll
    for (int i = 0; i < NI_EARLY; ++i) wf[i] = w0[i * 64];
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
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : (EPI == P2V_EPI_GELU_TAB ? g.ep.gelu.k : 1.0f);   // GELU_TAB: u = y * k, see gelu_tab_offset
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
        L.mkf[i] = L.on[i] ? ln_scale_mask(*reinterpret_cast<const float4*>(a.ln.mask + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      L.gmin = a.pre.gmin;
      L.gmax = a.pre.gmax;
      L.bmax = a.pre.bmax;
      L.pot = a.pre.pot != 0 && a.force_generic == 0;
      L.pm_one = a.pre.pm_one != 0;
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
  if constexpr (NI_EARLY < NI) wf[NI - 1] = wtile(0)[(NI - 1) * 64];
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
  const int goff = (int)g.ep.gelu.off;
  const float glo = (float)-goff, ghi = (float)(cells - 1 - goff) + 0.5f;     // clamp bounds of u ahead of the floor
  const unsigned char* gtabz = gtab + goff * 8;                           // entry `off`: the cell of u in [0, 1)

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
        for (int i = 0; i < 4; ++i) e0[i] = *reinterpret_cast<const uint2*>(gtabz + gelu_tab_offset(y0[i], glo, ghi));
        LG2_FENCE();
        LG2_MFMA(6 * gq + 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) e1[i] = *reinterpret_cast<const uint2*>(gtabz + gelu_tab_offset(y1[i], glo, ghi));
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
// host launchers (called from the C ABI in p2vit_capi.cpp)
// ---------------------------------------------------------------------------------------------------
#ifdef P2V_DIAG
extern unsigned long long* g_gemm_stamps;
#endif
int g_ln_pre = 1;         // P2V_LN_PRE=0: ignore p2v_ln.pre, every workgroup folds the constants itself (A/B and parity runs; same codes)
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
  if (!g_ln_pre) a.pre.gm = nullptr;
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
  if (!g_ln_pre) a.pre.gm = nullptr;
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

