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

#include "p2vit_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------
// clamp(round(v), -128, 127)  == UniformQuantizer.quant for int8 (quantizer/uniform.py:85-87)
__device__ __forceinline__ int sat8(float v) {
  float r = rintf(v);
  r = fminf(fmaxf(r, -128.f), 127.f);
  return (int)r;
}
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
  return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)d << 24);
}
__device__ __forceinline__ int sx8(unsigned w, int i) { return (int)(int8_t)(w >> (8 * i)); }

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
// GELU -> PoT requant.  Canonical value: q = clamp(rne(RN32(gelu(y)) / s)), gelu(y) = 0.5*y*erfc(-y/sqrt2)
// (reference: float nn.GELU then QAct, layers_quant.py:331-333).  Fast path: A&S 7.1.26 erfc (|err| <=
// 1.5e-7) in fp32; its total error is far below GELU_EPS, so whenever the scaled value is further than
// GELU_EPS/s from a rounding boundary the code is already decided.  Otherwise (about 1e-4 of the
// elements) the lane takes the fp64 path.  tests/test_gelu_gpu.py sweeps the fp32 line to check the bound.
// ---------------------------------------------------------------------------------------------------
#define GELU_EPS 4.0e-6f
__device__ __forceinline__ float gelu_fast(float y) {
  float z = fabsf(y) * 0.70710678f;
  float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504f);
  float hc = 0.5f * p * e;  // 0.5*erfc(|y|/sqrt2)
  return y < 0.f ? y * hc : y - y * hc;
}
__device__ __noinline__ float gelu_exact(float y) {
  double yd = (double)y;
  return (float)(0.5 * yd * erfc(-yd * 0.70710678118654752440));
}
__device__ __forceinline__ int gelu_q8(float y, float inv_s, bool force_slow, bool* took_slow) {
  float g = gelu_fast(y);
  float t = g * inv_s;
  float r = rintf(t);
  bool safe = (fabsf(t - r) < 0.5f - GELU_EPS * inv_s) || (fabsf(t) > 129.f);
  if (force_slow || !safe || !(GELU_EPS * inv_s < 0.25f)) {
    if (took_slow) *took_slow = true;
    r = rintf(gelu_exact(y) * inv_s);
  }
  r = fminf(fmaxf(r, -128.f), 127.f);
  return (int)r;
}

// ---------------------------------------------------------------------------------------------------
// K0: qact_input + im2col   (vit_fquant.py:705-706; layers_quant.py:467; layers.py:82-88)
// one thread = 4 consecutive pixels of one patch row -> one dword of the patch matrix.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quantize_patchify(const float* __restrict__ img, int B, int C, int H, int W,
                                                           int P, float inv_s, int8_t* __restrict__ out, int k_pad) {
  const int gw = W / P, gh = H / P;
  const int kq = k_pad >> 2;  // dwords per output row
  const long long total = (long long)B * gh * gw * kq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int col = (int)(idx % kq) * 4;
    long long row = idx / kq;
    unsigned v = 0;
    if (col < C * P * P) {
      int c = col / (P * P), rem = col % (P * P), i = rem / P, j = rem % P;
      int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
      const float* src = img + (((long long)b * C + c) * H + (py * P + i)) * W + px * P + j;
      float4 f = *reinterpret_cast<const float4*>(src);
      v = pack4(sat8(f.x * inv_s), sat8(f.y * inv_s), sat8(f.z * inv_s), sat8(f.w * inv_s));
    }
    *reinterpret_cast<unsigned*>(out + row * k_pad + col) = v;
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

__device__ __forceinline__ int lds_off64(int row, int chunk) { return row * GBK + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_tile(const v16i& acc, int m, int n_tile, int h, const GemmArgs& g) {
  // lane owns output row m, channels n_tile + 8*gq + 4*h + {0..3}, gq = 0..3   (C/D map of 32x32 MFMA)
  const bool row_ok = m < g.M;
  unsigned d[4];
  unsigned res[4];
  if (EPI == P2V_EPI_RESID) {
    uint4 e = make_uint4(0, 0, 0, 0);
    if (row_ok && n_tile + 16 * h < g.N)
      e = *reinterpret_cast<const uint4*>(g.ep.residual + (long long)m * g.ldo + n_tile + 16 * h);
    row16_to_halves(e, res[0], res[1], res[2], res[3]);
  }
  long long out_row = m;
  int tok = 0;
  if (EPI == P2V_EPI_EMBED) {
    int b = m / g.ep.patches, p = m % g.ep.patches;
    tok = p + 1;
    out_row = (long long)b * (g.ep.patches + 1) + tok;
  }
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = n_tile + 8 * gq + 4 * h;
    const float4 cs = *reinterpret_cast<const float4*>(g.colscale + n);
    const float4 bs = *reinterpret_cast<const float4*>(g.bias + n);
    float y[4];
    // F.linear / F.conv2d on fake-quantised operands: exact integer sum * (s_x*s_w[n]), then ONE rounding
    // for the fp32 bias (layers.py:87,178)
    y[0] = (float)acc[4 * gq + 0] * cs.x + bs.x;
    y[1] = (float)acc[4 * gq + 1] * cs.y + bs.y;
    y[2] = (float)acc[4 * gq + 2] * cs.z + bs.z;
    y[3] = (float)acc[4 * gq + 3] * cs.w + bs.w;
    int q[4];
    if (EPI == P2V_EPI_REQUANT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = sat8(y[i] * g.ep.inv_s_out);
    } else if (EPI == P2V_EPI_GELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = gelu_q8(y[i], g.ep.inv_s_out, false, nullptr);
    } else if (EPI == P2V_EPI_RESID) {
      // QAct(PTF) -> x + . -> QAct(PTF): non power-of-two per-channel scales, true fp32 divisions
      float4 sm = make_float4(1, 1, 1, 1), sr = sm, sn = sm;
      if (n < g.N) {
        sm = *reinterpret_cast<const float4*>(g.ep.s_mid + n);
        sr = *reinterpret_cast<const float4*>(g.ep.s_res + n);
        sn = *reinterpret_cast<const float4*>(g.ep.s_next + n);
      }
      const float smv[4] = {sm.x, sm.y, sm.z, sm.w}, srv[4] = {sr.x, sr.y, sr.z, sr.w}, snv[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x3 = (float)sat8(y[i] / smv[i]) * smv[i];
        float xr = (float)sx8(res[gq], i) * srv[i];
        q[i] = sat8((xr + x3) / snv[i]);
      }
    } else if (EPI == P2V_EPI_EMBED) {
      float4 sn = make_float4(1, 1, 1, 1), pe = make_float4(0, 0, 0, 0);
      if (n < g.N) {
        sn = *reinterpret_cast<const float4*>(g.ep.s_next + n);
        pe = *reinterpret_cast<const float4*>(g.ep.pos_deq + (long long)tok * g.N + n);
      }
      const float snv[4] = {sn.x, sn.y, sn.z, sn.w}, pev[4] = {pe.x, pe.y, pe.z, pe.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q1 = sat8(y[i] * g.ep.inv_s_pe);                  // PatchEmbed.qact
        int q2 = sat8((float)q1 * g.ep.pe_to_embed);          // qact_embed (both PoT: exact ratio)
        float xv = (float)q2 * g.ep.s_embed + pev[i];         // + qact_pos(pos_embed)
        q[i] = sat8(xv / snv[i]);                             // qact1 (PTF)
      }
    } else {  // HEAD: logits fp32 on the act_out grid
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q[i] = sat8(y[i] * g.ep.inv_s_out);
        if (row_ok && n + i < g.N) {
          reinterpret_cast<float*>(g.out)[(long long)m * g.ldo + n + i] = (float)q[i] * g.ep.s_out;
          if (g.out_codes) g.out_codes[(long long)m * g.ldo + n + i] = (int8_t)q[i];
        }
      }
    }
    d[gq] = pack4(q[0], q[1], q[2], q[3]);
  }
  if (EPI != P2V_EPI_HEAD) {
    uint4 o = halves_to_row16(d[0], d[1], d[2], d[3]);
    if (row_ok && n_tile + 16 * h < g.N)
      *reinterpret_cast<uint4*>(reinterpret_cast<int8_t*>(g.out) + out_row * g.ldo + n_tile + 16 * h) = o;
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_i8(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) int8_t lds[2 * (GBM + GBN) * GBK];
  int8_t* sX = lds;                    // [2][GBM][GBK] activation rows
  int8_t* sW = lds + 2 * GBM * GBK;    // [2][GBN][GBK] weight rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: each XCD walks a contiguous range of tiles, n fastest, so the tiles that share
  // an activation panel hit the same L2.
  int bid = blockIdx.x, nt = gridDim.x, xcd = bid & 7, qd = nt >> 3, rm = nt & 7;
  int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tn = t % g.tiles_n, tm = t / g.tiles_n;
  const int m0 = tm * GBM, n0 = tn * GBN;

  const int lrow = tid >> 2, lchunk = tid & 3;
  const int8_t* gx[2];
  const int8_t* gwp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = lrow + 64 * i;
    int mr = m0 + r;
    mr = mr < g.M ? mr : g.M - 1;
    gx[i] = g.A + (long long)mr * g.lda + lchunk * 16;
    gwp[i] = g.W + (long long)(n0 + r) * g.K + lchunk * 16;
  }
  v16i acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

  const int nk = g.K / GBK;
  uint4 rx[2], rw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rx[i] = *reinterpret_cast<const uint4*>(gx[i]);
    rw[i] = *reinterpret_cast<const uint4*>(gwp[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = lrow + 64 * i;
    *reinterpret_cast<uint4*>(sX + lds_off64(r, lchunk)) = rx[i];
    *reinterpret_cast<uint4*>(sW + lds_off64(r, lchunk)) = rw[i];
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rx[i] = *reinterpret_cast<const uint4*>(gx[i] + (kt + 1) * GBK);
        rw[i] = *reinterpret_cast<const uint4*>(gwp[i] + (kt + 1) * GBK);
      }
    }
    const int8_t* cx = sX + cur * GBM * GBK;
    const int8_t* cw = sW + cur * GBN * GBK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fw[i] = *reinterpret_cast<const v4i*>(cw + lds_off64(wn * 64 + i * 32 + l31, 2 * ks + h));
        fx[i] = *reinterpret_cast<const v4i*>(cx + lds_off64(wm * 64 + i * 32 + l31, 2 * ks + h));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[ni], fx[mi], acc[ni][mi], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      int8_t* nx = sX + (cur ^ 1) * GBM * GBK;
      int8_t* nw = sW + (cur ^ 1) * GBN * GBK;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int r = lrow + 64 * i;
        *reinterpret_cast<uint4*>(nx + lds_off64(r, lchunk)) = rx[i];
        *reinterpret_cast<uint4*>(nw + lds_off64(r, lchunk)) = rw[i];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      gemm_epilogue_tile<EPI>(acc[ni][mi], m0 + wm * 64 + mi * 32 + l31, n0 + wn * 64 + ni * 32, h, g);
}

// ---------------------------------------------------------------------------------------------------
// K2: integer LayerNorm (QIntLayerNorm mode 'int', layers.py:255-289) + /channel_scale + qact0 clamp
// (vit_fquant.py:284-289).  One row per 32-lane half wave (12 bytes/lane at C=384), LN_ROWS rows per half
// wave so the five per-channel constant vectors stay in registers.  sum x and sum x^2 are exact integers;
// everything after mirrors the reference's fp32 operation order.
// ---------------------------------------------------------------------------------------------------
#define LN_ROWS 8
template <int NCH>
__global__ __launch_bounds__(256) void k_int_layernorm(LnArgs a) {
  const int tid = threadIdx.x, l32 = tid & 31, hw = tid >> 5;
  float4 mk[NCH], gm[NCH], bt[NCH], io[NCH], pm[NCH];
  bool on[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (l32 + 32 * i) * 4;
    on[i] = c < a.C;
    int cc = on[i] ? c : 0;
    mk[i] = *reinterpret_cast<const float4*>(a.ln.mask + cc);
    gm[i] = *reinterpret_cast<const float4*>(a.ln.gamma + cc);
    bt[i] = *reinterpret_cast<const float4*>(a.ln.beta + cc);
    io[i] = *reinterpret_cast<const float4*>(a.ln.inv_out + cc);
    pm[i] = *reinterpret_cast<const float4*>(a.ln.post_mul + cc);
  }
  const float s1 = a.ln.s1;
  const float Cf = (float)a.C;
  const long long row0 = ((long long)blockIdx.x * 8 + hw) * LN_ROWS;
  for (int rr = 0; rr < LN_ROWS; ++rr) {
    const long long row = row0 + rr;
    if (row >= a.rows) break;   // uniform within the half wave; shuffles below use width 32
    const int8_t* src = a.x + row * a.row_stride;
    float xq[NCH][4];
    int S1 = 0;
    long long S2 = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      unsigned w = on[i] ? *reinterpret_cast<const unsigned*>(src + (l32 + 32 * i) * 4) : 0u;
      const float m4[4] = {mk[i].x, mk[i].y, mk[i].z, mk[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int v = on[i] ? sx8(w, j) * (int)m4[j] : 0;   // x_q * in_scale_mask  (layers.py:269-273)
        xq[i][j] = (float)v;
        S1 += v;
        S2 += (long long)(v * v);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      S1 += __shfl_xor(S1, o, 32);
      S2 += __shfl_xor(S2, o, 32);
    }
    const float S1f = (float)S1, S2f = (float)S2;
    const float mean = (S1f / Cf) * s1;                                  // x_q.mean(-1) * in_scale1
    const float stdv = (s1 / Cf) * sqrtf(Cf * S2f - S1f * S1f);          // layers.py:276-277
    const float rs = s1 / stdv;
    const float mos = mean / stdv;
    unsigned outw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float g4[4] = {gm[i].x, gm[i].y, gm[i].z, gm[i].w}, b4[4] = {bt[i].x, bt[i].y, bt[i].z, bt[i].w};
      const float i4[4] = {io[i].x, io[i].y, io[i].z, io[i].w}, p4[4] = {pm[i].x, pm[i].y, pm[i].z, pm[i].w};
      int q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float A = (rs * g4[j]) * i4[j];                            // (s1/std)*gamma / out_scale
        const float absA = fabsf(A);
        const int eA = (int)((__float_as_uint(absA) >> 23) & 255u) - 127; // floor(log2|A|)
        int N = 7 - eA;                                                  // get_MN, layers.py:234-238
        N = N < 0 ? 0 : (N > 31 ? 31 : N);
        const float pN = __uint_as_float((unsigned)(127 + N) << 23);
        const float inN = __uint_as_float((unsigned)(127 - N) << 23);
        float M = floorf(absA * pN);
        M = fminf(M, 255.f);
        const float sM = A < 0.f ? -M : (A > 0.f ? M : 0.f);             // A.sign() * M
        const float Bv = rintf(((b4[j] - mos * g4[j]) * i4[j]) * pN);    // layers.py:283-286
        const float o = rintf((sM * xq[i][j] + Bv) * inN);               // layers.py:288
        q[j] = sat8(o * p4[j]);                                          // * out_scale / cs_next / s_next
      }
      outw[i] = pack4(q[0], q[1], q[2], q[3]);
    }
    int8_t* dst = a.out + row * a.out_stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (on[i]) *reinterpret_cast<unsigned*>(dst + (l32 + 32 * i) * 4) = outw[i];
  }
}

// ---------------------------------------------------------------------------------------------------
// K3: fused attention core  (vit_fquant.py:309-326; QIntSoftmax layers.py:323-376)
//   one workgroup per (image, head); K (int8) and V^T (bf16) staged in LDS; each wave owns 32-query
//   blocks.  S^T = K . Q^T on the int8 MFMA puts a whole score row on one lane pair, so the row max and
//   the exact int64 sum of exp_int = z * 2^(32-q) are in-lane plus one cross-half shuffle.  exp_int
//   depends only on (max - score) in [0,255]: a 256-entry LDS table.  P = 2^-k is exact in bf16 and V
//   codes are exact in bf16, so P.V on the bf16 MFMA is exact in its fp32 accumulator (|sum| < 2^24 units
//   of 2^-15).
// ---------------------------------------------------------------------------------------------------
template <int HD, int NKB>
__global__ __launch_bounds__(256) void k_lis_attention(AttnArgs a) {
  constexpr int KROWS = NKB * 32;
  constexpr int VSTRIDE = KROWS + 4;            // bf16 elements; dword stride = 2*odd -> conflict-free b64 reads
  constexpr int CH = HD / 16;                   // 16-byte chunks per K row
  constexpr int NDT = HD / 32;                  // 32-wide output-channel tiles
  constexpr int NKS = HD / 32;                  // int8 MFMA k-steps over head_dim
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int8_t* sK = reinterpret_cast<int8_t*>(smem);                                  // [KROWS][HD] swizzled
  unsigned short* sVt = reinterpret_cast<unsigned short*>(smem + KROWS * HD);    // [HD][VSTRIDE] bf16
  long long* lutE = reinterpret_cast<long long*>(smem + KROWS * HD + HD * VSTRIDE * 2);  // [256]
  float* lutF = reinterpret_cast<float*>(lutE + 256);                             // [256]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / a.H, head = blockIdx.x % a.H;
  const int N = a.N, D = a.H * HD, ld = 3 * D;
  const int8_t* base = a.qkv + (long long)b * N * ld + head * HD;

  // exp table: d = max - score -> exp_int = z * 2^(32-q)       (int_exp / int_polynomial, layers.py:334-358)
  {
    int xi = -tid;
    const int lim = 32 * a.at.x0_int;
    xi = xi < lim ? lim : xi;
    const int q = xi / a.at.x0_int;              // both <= 0: trunc == floor
    const int r = xi - a.at.x0_int * q;
    const long long z = (long long)r * (r + a.at.b_int) + a.at.c_int;
    long long e = z << (32 - q);
    e = e < 0 ? 0 : e;
    lutE[tid] = e;
    lutF[tid] = (float)e;                        // exact: z < 2^24
  }
  // stage K rows and V^T
  for (int i = tid; i < KROWS * CH; i += 256) {
    const int row = i / CH, c = i % CH;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (row < N) {
      kv = *reinterpret_cast<const uint4*>(base + (long long)row * ld + D + c * 16);
      vv = *reinterpret_cast<const uint4*>(base + (long long)row * ld + 2 * D + c * 16);
    }
    const int sw = (HD == 64) ? (c ^ ((row >> 2) & 3)) : (c ^ ((row >> 3) & 1));
    *reinterpret_cast<uint4*>(sK + row * HD + sw * 16) = kv;
    const unsigned w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float f = (float)sx8(w4[j >> 2], j & 3);
      sVt[(c * 16 + j) * VSTRIDE + row] = (unsigned short)(__float_as_uint(f) >> 16);   // exact bf16
    }
  }
  __syncthreads();

  const int nqb = (N + 31) >> 5;
  for (int qb = wave; qb < nqb; qb += 4) {
    const int qrow = qb * 32 + l31;
    const int qr = qrow < N ? qrow : N - 1;
    v4i fq[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      fq[ks] = *reinterpret_cast<const v4i*>(base + (long long)qr * ld + ks * 32 + h * 16);
    v16i s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int row = kb * 32 + l31, c = 2 * ks + h;
        const int sw = (HD == 64) ? (c ^ ((row >> 2) & 3)) : (c ^ ((row >> 3) & 1));
        const v4i fk = *reinterpret_cast<const v4i*>(sK + row * HD + sw * 16);
        s[kb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fk, fq[ks], s[kb], 0, 0, 0);
      }
    }
    // scores -> int8 codes of qact_attn1, row max           ((q@k^T)*scale -> QAct, vit_fquant.py:316-317)
    int mx = -1000;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float f = (((float)s[kb][r] * a.at.s_qkv_sq) * a.at.qk_scale) * a.at.inv_s_attn;
        const int c = key < N ? sat8(f) : -1000;
        s[kb][r] = c;
        mx = c > mx ? c : mx;
      }
    {
      const int o = __shfl_xor(mx, 32);
      mx = o > mx ? o : mx;
    }
    long long S = 0;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = s[kb][r];
        const int d = c == -1000 ? -1 : mx - c;
        s[kb][r] = d;
        S += d >= 0 ? lutE[d] : 0ll;
      }
    S += __shfl_xor(S, 32);
    const float Sf = (float)S;                                  // exp_int.sum(-1): exact, then one rounding

    v16f o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      unsigned pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        unsigned hw2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int d = s[kb][r + e];
          unsigned bits = 0;
          int k = 16;
          if (d >= 0) {
            const float ratio = rintf(Sf / lutF[d]);            // round(sum / exp_int), layers.py:370
            k = (int)((__float_as_uint(ratio) + 0x00400000u) >> 23) - 127;   // log_round, layers.py:323-329
            k = k < 0 ? 0 : (k > 16 ? 16 : k);
            bits = k < 16 ? (unsigned)(127 - k) << 7 : 0u;      // 2^-k as bf16; k>=16 -> 0 (layers.py:372-375)
          }
          hw2[e] = bits;
          if (a.probs_k && d >= 0 && qrow < N) {
            const int key = kb * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * h;
            a.probs_k[(((long long)b * a.H + head) * N + qrow) * N + key] = (int8_t)k;
          }
        }
        pk[r >> 1] = hw2[0] | (hw2[1] << 16);
      }
      // O^T += V^T . P^T : A = V^T fragment (rows = channel), B = P^T (k = key, permuted as the accumulator
      // rows come: element j of half h is key 16s + 8(j>>2) + 4h + (j&3)).
#pragma unroll
      for (int sst = 0; sst < 2; ++sst) {
        v4i pb = {(int)pk[4 * sst + 0], (int)pk[4 * sst + 1], (int)pk[4 * sst + 2], (int)pk[4 * sst + 3]};
        const v8bf fb = __builtin_bit_cast(v8bf, pb);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const unsigned short* vp = sVt + (dt * 32 + l31) * VSTRIDE + kb * 32 + 16 * sst + 4 * h;
          const uint2 lo = *reinterpret_cast<const uint2*>(vp);
          const uint2 hi = *reinterpret_cast<const uint2*>(vp + 8);
          v4i va = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, va), fb, o[dt], 0, 0, 0);
        }
      }
    }
    // qact2: (attn @ v) / s  with attn@v = O * s_q1   (vit_fquant.py:325-326)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      unsigned dw[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        dw[gq] = pack4(sat8(o[dt][4 * gq + 0] * a.at.av_mul), sat8(o[dt][4 * gq + 1] * a.at.av_mul),
                       sat8(o[dt][4 * gq + 2] * a.at.av_mul), sat8(o[dt][4 * gq + 3] * a.at.av_mul));
      const uint4 ov = halves_to_row16(dw[0], dw[1], dw[2], dw[3]);
      if (qrow < N)
        *reinterpret_cast<uint4*>(a.out + ((long long)b * N + qrow) * D + head * HD + dt * 32 + 16 * h) = ov;
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
#define CHECK_LAUNCH()                                     \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return (int)e_;                  \
  } while (0)

int p2v_launch_patchify(const float* img, int B, int C, int H, int W, int P, float inv_s, int8_t* out, int k_pad, hipStream_t st) {
  long long total = (long long)B * (H / P) * (W / P) * (k_pad / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_quantize_patchify, dim3(blocks), dim3(256), 0, st, img, B, C, H, W, P, inv_s, out, k_pad);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_fill_cls(int8_t* x, int B, int T, int D, const int8_t* cls, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_cls, dim3((B * D + 255) / 256), dim3(256), 0, st, x, B, T, D, cls);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gemm(int epi, const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.tiles_n = (g.N + GBN - 1) / GBN;
  const int tiles_m = (g.M + GBM - 1) / GBM;
  dim3 grid(g.tiles_n * tiles_m), block(256);
  switch (epi) {
    case P2V_EPI_REQUANT: hipLaunchKernelGGL(k_gemm_i8<P2V_EPI_REQUANT>, grid, block, 0, st, g); break;
    case P2V_EPI_GELU: hipLaunchKernelGGL(k_gemm_i8<P2V_EPI_GELU>, grid, block, 0, st, g); break;
    case P2V_EPI_RESID: hipLaunchKernelGGL(k_gemm_i8<P2V_EPI_RESID>, grid, block, 0, st, g); break;
    case P2V_EPI_EMBED: hipLaunchKernelGGL(k_gemm_i8<P2V_EPI_EMBED>, grid, block, 0, st, g); break;
    case P2V_EPI_HEAD: hipLaunchKernelGGL(k_gemm_i8<P2V_EPI_HEAD>, grid, block, 0, st, g); break;
    default: return -1;
  }
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_layernorm(const LnArgs& a, hipStream_t st) {
  const int nch = (a.C + 127) / 128;
  const int rows_per_block = 8 * LN_ROWS;
  dim3 grid((unsigned)((a.rows + rows_per_block - 1) / rows_per_block)), block(256);
  switch (nch) {
    case 1: hipLaunchKernelGGL(k_int_layernorm<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_int_layernorm<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(k_int_layernorm<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_int_layernorm<4>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(k_int_layernorm<6>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(k_int_layernorm<8>, grid, block, 0, st, a); break;
    default: return -1;
  }
  CHECK_LAUNCH();
  return 0;
}

template <int HD, int NKB>
static int launch_attn_t(const AttnArgs& a, hipStream_t st) {
  constexpr int KROWS = NKB * 32;
  constexpr size_t smem = (size_t)KROWS * HD + (size_t)HD * (KROWS + 4) * 2 + 256 * 8 + 256 * 4;
  hipLaunchKernelGGL((k_lis_attention<HD, NKB>), dim3(a.B * a.H), dim3(256), smem, st, a);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_attention(const AttnArgs& a, int head_dim, hipStream_t st) {
  const int nkb = (a.N + 31) / 32;
  if (head_dim == 64) {
    switch (nkb) {
      case 1: return launch_attn_t<64, 1>(a, st);
      case 2: return launch_attn_t<64, 2>(a, st);
      case 7: return launch_attn_t<64, 7>(a, st);
      default: return -1;
    }
  } else if (head_dim == 32) {
    switch (nkb) {
      case 1: return launch_attn_t<32, 1>(a, st);
      case 2: return launch_attn_t<32, 2>(a, st);
      case 7: return launch_attn_t<32, 7>(a, st);
      default: return -1;
    }
  }
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
