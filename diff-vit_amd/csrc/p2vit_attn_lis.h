// p2vit_attn_lis.h -- the ViT log-int-softmax attention kernel and its per-shape launcher, shared by the two translation units that
// instantiate it (p2vit_attn.hip: head_dim 32 / 64; p2vit_attn_wide.hip: 48 / 80 / 96 / 128), so that they compile side by side.
#pragma once
#include "p2vit_device.h"

extern int g_attn_waves;     // P2V_ATTN_WAVES

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

// log_round + the uint4 clamp of two ratios at once (layers.py:323-329, 372-375): E = (bits + 0x00400000) >> 23 is the biased exponent of 2^k
// (ratio >= 1, so k = E - 127 >= 0) and the probability is 2^-k, or 0 from k = 16 on.  It goes into the P.V product as bf16 SCALED BY
// 2^-111: exponent field 16 - k, i.e. (143 - E) << 7 with an UNSIGNED-SATURATING subtraction (v_pk_sub_u16 clamp) - from k = 16 on the field
// saturates to the all-zero pattern, which IS +0.0, so the clamp costs nothing (round 3: (254 - E) << 7, a second subtraction, a shift and an
// and-not to zero the small ones).  The scaled probabilities are normal bf16 numbers >= 2^-126, the products with the integer V codes and their
// fp32 sums are multiples of 2^-126 below 2^-102: still exact in 24 bits, none denormal; the 2^111 is folded into av_mul (a power of two).
#define P2V_PROB_SCALE 0x1p111f
__device__ __forceinline__ unsigned lis_prob_pair(float r0, float r1) {
  const unsigned hi2 = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  const unsigned eb = __builtin_bit_cast(unsigned, __builtin_bit_cast(v2u16, hi2) + (v2u16){0x40, 0x40}) & 0x7F807F80u;      // E << 7, twice
  unsigned out;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(out) : "s"(0x47804780u), "v"(eb));                                          // sat((143 - E) << 7)
  return out;
}

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
// HD: any multiple of 16 up to 128 (round 4; instantiated for 32, 48, 64, 80, 96, 128): the score of a key block is the sum of NQ = ceil(HD / 64)
// 64-deep MFMAs (lane group g holds chunk 4 j + g of the row in step j, zero past the row), the P.V product runs over HD / 16 output tiles.
template <int HD, int NKP, bool TAP, bool ISH>
__global__ __launch_bounds__(512, HD > 64 ? 2 : (NKP <= 7 ? 4 : (NKP <= 10 ? 3 : 2))) void k_lis_attention(AttnArgs a) {
  static_assert(HD % 16 == 0 && HD >= 16 && HD <= 128, "head_dim");
  constexpr int KROWS = NKP * 32;
  constexpr int NKB = NKP * 2;                  // 16-key blocks
  constexpr int VSTRIDE = KROWS + 4;            // bf16 elements; dword stride = 2*odd -> conflict-free b64 reads
  constexpr int CH = HD / 16;                   // 16-byte chunks per K row
  constexpr int NQ = (CH + 3) / 4;              // 64-deep MFMA steps of a score
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
    // the fp64 reciprocal (IEEE division, correctly rounded): the per-score quotient is one fp64 multiply by it, see below
    reinterpret_cast<long long*>(lutE)[tid] = e;
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
  const float avm = a.at.av_mul * P2V_PROB_SCALE;        // the probabilities enter the P.V product scaled by 2^-111 (lis_prob_pair); exact, checked by the launcher
  const int nqb = (N + 15) >> 4;
  const int nwaves = (int)(blockDim.x >> 6);
  // the Q fragment of a wave's first query block is requested before the barrier and the one of its next block a block ahead: its
  // global-memory latency overlaps the staging wait / the arithmetic of the current block
  AT_STAMP(2);
  v4i fq_next[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    fq_next[j] = (v4i){0, 0, 0, 0};
    if (4 * j + g < CH && wave < nqb) {
      const int qr0 = wave * 16 + l15;
      fq_next[j] = *reinterpret_cast<const v4i*>(base + (long long)(qr0 < N ? qr0 : N - 1) * ld + (4 * j + g) * 16);
    }
  }
  __syncthreads();
  AT_STAMP(3);
  [[maybe_unused]] int stamp_base = 4;
  for (int qb = wave; qb < nqb; qb += nwaves) {
    const int qrow = qb * 16 + l15;
    v4i fq[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      fq[j] = fq_next[j];
      if (4 * j + g < CH && qb + nwaves < nqb) {
        const int qn = (qb + nwaves) * 16 + l15;
        fq_next[j] = *reinterpret_cast<const v4i*>(base + (long long)(qn < N ? qn : N - 1) * ld + (4 * j + g) * 16);
      }
    }
    v4i s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {          // all score MFMAs first: no dependent use behind an MFMA
      const int row = kb * 16 + l15;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const bool live = 4 * j + g < CH;        // (the chunk of this lane group exists in the row)
        const int c = live ? 4 * j + g : 0;
        const int sw = (HD == 64) ? (c ^ (((row >> 3) & 1) << 1)) : c;
        v4i fk = *reinterpret_cast<const v4i*>(sK + row * HD + sw * 16);
        if (!live) fk = (v4i){0, 0, 0, 0};
        if (j == 0) s[kb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq[0], (v4i){0, 0, 0, 0}, 0, 0, 0);
        else s[kb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq[j], s[kb], 0, 0, 0);
      }
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
          int code = (int)((unsigned)sv + (unsigned)hm1 + (((unsigned)sv >> p) & 1u)) >> p;    // |sv| <= 128 * 128 * 128 = 2^21: no overflow
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
        pk[e2] = lis_prob_pair(ratio[0], ratio[1]);              // 2^-k * 2^-111 as bf16, 0 from k = 16 on
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
        *reinterpret_cast<unsigned*>(dst + dt * 16) = pack4_rne_sat(o[dt][0] * avm, o[dt][1] * avm, o[dt][2] * avm, o[dt][3] * avm);
    }
    AT_STAMP(stamp_base + 3);
    stamp_base += 4;
  }
}

// ISH_OK: the integer score requant is instantiated for this head_dim (64, where qk_scale is a power of two); otherwise every launch takes the
// fp32 chain, which is exact for any multiplier
template <int HD, int NKB, bool ISH_OK = true>
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
  if constexpr (!ISH_OK) a.pshift = 0;
  if (a.probs_k) {
    if constexpr (ISH_OK) { if (a.pshift) { P2V_ATTN_LAUNCH(true, true); CHECK_LAUNCH(); return 0; } }
    P2V_ATTN_LAUNCH(true, false);
  } else {
    if constexpr (ISH_OK) { if (a.pshift) { P2V_ATTN_LAUNCH(false, true); CHECK_LAUNCH(); return 0; } }
    P2V_ATTN_LAUNCH(false, false);
  }
#undef P2V_ATTN_LAUNCH
  CHECK_LAUNCH();
  return 0;
}

// every ceil(tokens / 32) up to MAXP is instantiated: the padding of a launch is always less than one 32-key pair
#define P2V_ATTN_CASE(HD_, P_, MAXP_, ISH_) case P_: if constexpr (P_ <= MAXP_) return launch_attn_t<HD_, P_, ISH_>(a, st); else return -1;
#define P2V_ATTN_CASES(HD_, MAXP_, ISH_)                                                                                              \
  switch (nkb) {                                                                                                                      \
    P2V_ATTN_CASE(HD_, 1, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 2, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 3, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 4, MAXP_, ISH_)     \
    P2V_ATTN_CASE(HD_, 5, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 6, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 7, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 8, MAXP_, ISH_)     \
    P2V_ATTN_CASE(HD_, 9, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 10, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 11, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 12, MAXP_, ISH_)  \
    P2V_ATTN_CASE(HD_, 13, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 14, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 15, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 16, MAXP_, ISH_) \
    P2V_ATTN_CASE(HD_, 17, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 18, MAXP_, ISH_) P2V_ATTN_CASE(HD_, 19, MAXP_, ISH_)                                   \
    default: return -1;                                                                                                               \
  }
