// p2vit_epilogue.h -- the fp32 epilogues of the int8 MFMA GEMMs, shared by the tiled kernels (p2vit_gemm.hip) and the fused
// LayerNorm + GEMM kernels (p2vit_ln.hip).
#pragma once
#include "p2vit_device.h"

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
#define P2V_EPI_RESID_PRE 6  // internal: P2V_EPI_RESID with the constants of p2v_resid_prefold (p2v_epilogue.resid_tab != NULL)

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
    // GELU_TAB: the table is indexed and compared on u = y * k (k = 2 / s_out, a power of two), see gelu_tab_offset
    const float fold = EPI == P2V_EPI_REQUANT ? g.ep.inv_s_out : (EPI == P2V_EPI_GELU_TAB ? g.ep.gelu.k : 1.0f);
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
        const float un = EPI == P2V_EPI_REQUANT ? 1.0f / g.ep.inv_s_out : (EPI == P2V_EPI_GELU_TAB ? 1.0f / g.ep.gelu.k : 1.0f);
        *reinterpret_cast<float4*>(g.ep.tap_out + (long long)(m_first + 32 * b) * g.N + n) =
            make_float4(yy[b][0] * un, yy[b][1] * un, yy[b][2] * un, yy[b][3] * un);
      }
    }
    if (EPI == P2V_EPI_GELU_TAB) {
      const int goff = (int)g.ep.gelu.off;
      gelu_tab_q8x8(yy[0], yy[1], gtab + goff * 8, (float)-goff, (float)(g.ep.gelu.cells - 1 - goff) + 0.5f, d[0][gq], d[1][gq]);
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


// ---------------------------------------------------------------------------------------------------
// RESID epilogue on pre-folded constants (p2v_resid_prefold, include/p2vit.h): QLinear -> QAct(PTF) -> + residual -> QAct(PTF)
// (vit_fquant.py:334-338,431; layers_quant.py:342-346, vit_fquant.py:468).  Per output, after the accumulator:
//   t  = fma(acc, colscale * fl(1/s_mid), bias * fl(1/s_mid))       the first quotient in one instruction; margin test as in div_q8fx4
//                                                                     (one test per EIGHT outputs; the rare undecided lane divides exactly)
//   q3 = clamp(rint(t));  xs = RN(res * s_res) + RN(q3 * s_mid)      the reference's three roundings
//   q  = clamp(rint(fma(xs, rh, xs * rl)))                           second quotient by a 48-bit reciprocal, no test: every numerator
//                                                                     this channel can produce was checked against the IEEE division
//                                                                     when the table was built (k_resid_prefold)
// 14 VALU instructions per output against 17.5 of the generic form, six instead of eight ds_read_b128 of constants per four channels,
// and one wave-uniform branch per eight outputs instead of four.
// ---------------------------------------------------------------------------------------------------
#define P2V_RESID_TAB_ARRAYS 6
struct ResidLds {
  float c1[GBN], b1[GBN], s_mid[GBN], s_res[GBN], rh[GBN], rl[GBN];
};
// the second quotient exactly as the epilogue computes it (also used by the table builder's exhaustive check)
__device__ __forceinline__ float resid_q2(float xs, float rh, float rl) { return __builtin_fmaf(xs, rh, xs * rl); }

template <bool LEAN>
__device__ __forceinline__ void gemm_epilogue_resid_pre(const v16i (&acc)[2], int m_first, int n_tile, int nl, int h, const GemmArgs& g,
                                                        const ResidLds* e, const uint4 (&resv)[2]) {
  unsigned d[2][4], res[2][4];
  const bool row_ok[2] = {m_first < g.M, m_first + 32 < g.M};
  row16_to_halves(resv[0], res[0][0], res[0][1], res[0][2], res[0][3]);
  row16_to_halves(resv[1], res[1][0], res[1][1], res[1][2], res[1][3]);
  struct Consts { float4 c1, b1, sm, sr, rh, rl; };
  auto load_consts = [&](int gq) {
    const int c = nl + 8 * gq + 4 * h;
    Consts k;
    k.c1 = *reinterpret_cast<const float4*>(e->c1 + c); k.b1 = *reinterpret_cast<const float4*>(e->b1 + c);
    k.sm = *reinterpret_cast<const float4*>(e->s_mid + c); k.sr = *reinterpret_cast<const float4*>(e->s_res + c);
    k.rh = *reinterpret_cast<const float4*>(e->rh + c); k.rl = *reinterpret_cast<const float4*>(e->rl + c);
    return k;
  };
  Consts knext;
  if (!LEAN) knext = load_consts(0);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const Consts k = LEAN ? load_consts(gq) : knext;
    if (!LEAN && gq < 3) knext = load_consts(gq + 1);     // an LDS round trip ahead of its use
    const float c1[4] = {k.c1.x, k.c1.y, k.c1.z, k.c1.w}, b1[4] = {k.b1.x, k.b1.y, k.b1.z, k.b1.w};
    const float sm[4] = {k.sm.x, k.sm.y, k.sm.z, k.sm.w}, sr[4] = {k.sr.x, k.sr.y, k.sr.z, k.sr.w};
    const float rh[4] = {k.rh.x, k.rh.y, k.rh.z, k.rh.w}, rl[4] = {k.rl.x, k.rl.y, k.rl.z, k.rl.w};
    float r[2][4], dv[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = __builtin_fmaf((float)acc[b][4 * gq + i], c1[i], b1[i]);
        r[b][i] = rintf(t);
        dv[b][i] = t - r[b][i];
      }
    const float dmax = fmaxf(fmaxf(fmaxf(fmaxf(fabsf(dv[0][0]), fabsf(dv[0][1])), fabsf(dv[0][2])), fmaxf(fabsf(dv[0][3]), fabsf(dv[1][0]))),
                             fmaxf(fmaxf(fabsf(dv[1][1]), fabsf(dv[1][2])), fabsf(dv[1][3])));
    if (__builtin_amdgcn_ballot_w64(!(dmax < 0.5f - 1.0e-4f)) != 0) {      // ~1e-3 of the waves: the reference's own operations for the undecided lanes
      const int n = n_tile + 8 * gq + 4 * h;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!(fabsf(dv[b][i]) < 0.5f - 1.0e-4f) && n + i < g.N) {
            const float y = __builtin_fmaf((float)acc[b][4 * gq + i], g.colscale[n + i] * (g.w4 ? 0.0625f : 1.0f), g.bias[n + i]);
            r[b][i] = rintf(y / g.ep.s_mid[n + i]);
          }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float q3 = clamp8f(r[b][i]);
        const float xs = (float)sx8(res[b][gq], i) * sr[i] + q3 * sm[i];
        c[i] = pre_pack(resid_q2(xs, rh[i], rl[i]));
      }
      d[b][gq] = pack4_pre(c[0], c[1], c[2], c[3]);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the constant reads of the next group from being hoisted (register pressure)
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const uint4 o = halves_to_row16(d[b][0], d[b][1], d[b][2], d[b][3]);
    if (row_ok[b] && n_tile + 16 * h < g.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(g.out) + (long long)(m_first + 32 * b) * g.ldo + n_tile + 16 * h) = o;
  }
}
