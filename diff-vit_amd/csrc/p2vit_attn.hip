// p2vit_attn.hip -- fused attention cores: ViT log-int-softmax attention and Swin window attention, with their launchers.
#include "p2vit_attn_lis.h"

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
  const float av_mul = (a.wa.s_q1 / a.wa.s_q3) * P2V_PROB_SCALE;            // the probabilities are scaled by 2^-111 (lis_prob_pair)
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
      sP[wave][g][l15] = (unsigned short)((E < 143 && l15 < 13) ? (143 - E) << 7 : 0);     // 2^-k * 2^-111 as bf16 (lis_prob_pair), 0 from k = 16 (and for padding: sum / 1 >= 2^32)
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
          pk[e2] = lis_prob_pair(ratio[0], ratio[1]);            // 2^-k * 2^-111 as bf16, 0 from k = 16 on (see k_lis_attention)
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
        *reinterpret_cast<unsigned*>(dst + dt * 16) = pack4_rne_sat(o[dt][0] * av_mul, o[dt][1] * av_mul, o[dt][2] * av_mul, o[dt][3] * av_mul);
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// host launchers (called from the C ABI in p2vit_capi.cpp)
// ---------------------------------------------------------------------------------------------------
int g_attn_waves = 8;     // P2V_ATTN_WAVES
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

int p2v_launch_attention_wide(const AttnArgs& a, int head_dim, int nkb, hipStream_t st);      // p2vit_attn_wide.hip

int g_attn_stream = 0;    // P2V_ATTN_STREAM=1: every launch takes the streaming kernel (parity runs against the resident one; same codes)
int p2v_launch_attention(const AttnArgs& a, int head_dim, hipStream_t st) {
  const int nkb = (a.N + 31) / 32;
  {   // the kernel folds s_q1^2 / s_attn into qk_scale: exact only for a power of two (both are PoT scales in the reference)
    int ex;
    const float m2 = a.at.s_qkv_sq * a.at.inv_s_attn;
    if (!(m2 > 0.f) || frexpf(m2, &ex) != 0.5f) return -2;
  }
  if (g_attn_stream || a.N > p2v_resident_tokens_of(head_dim)) return p2v_launch_attention_stream(a, head_dim, st);
  if (head_dim == 64) { P2V_ATTN_CASES(64, 19, true) }
  if (head_dim == 32) { P2V_ATTN_CASES(32, 19, false) }
  return p2v_launch_attention_wide(a, head_dim, nkb, st);
}
