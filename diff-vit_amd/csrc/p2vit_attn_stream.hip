// p2vit_attn_stream.hip -- the ViT log-int-softmax attention for token counts beyond what k_lis_attention keeps resident (round 4).
//
// k_lis_attention holds K (int8) and V^T (bf16) of an image's head in LDS and a query block's scores in registers: 608 tokens at most
// (544 / 384 at head_dim 96 / 128, p2v_max_tokens).  The reference's VisionTransformer takes any img_size (vit_fquant.py:494,535-540), e.g.
// the 448^2 / 512^2 fine-tunes of the ViT family (785 / 1025 tokens).  For those this kernel STREAMS the keys: one wave = one 16-query block
// of one (image, head); three passes over the key blocks -
//   1. scores (K fragments straight from global / L2, the same v_mfma_i32_16x16x64_i8 products) -> qact_attn1 codes, packed four to a dword
//      into LDS (256 B per key block), row minimum of the negated codes;
//   2. exp_int of every code from the 257-entry table, exact int64 sum (layers.py:334-358);
//   3. per pair of key blocks: V rows staged transposed as bf16 in LDS (32 keys at a time), probabilities 2^-k (lis_prob_pair), P.V on
//      v_mfma_f32_16x16x32_bf16.
// Same arithmetic, operand layouts and helper code as k_lis_attention (fp32 score requantisation: exact for any multiplier), so the codes are
// identical where both kernels apply (tests/test_engine_gpu.py::test_lis_attention_streamed).  Throughput is not the point of this path: every
// query block re-reads K and V of its head (L2 hits) - it exists so that no geometry the reference accepts is refused.
#include "p2vit_attn_lis.h"

#define AS_VSTRIDE 36          // bf16 elements per channel row of the staged V^T pair (32 keys + 4: conflict-free b64 reads)

template <int HD, bool TAP>
__global__ __launch_bounds__(64) void k_lis_attention_stream(AttnArgs a) {
  static_assert(HD % 16 == 0 && HD >= 16 && HD <= 128, "head_dim");
  constexpr int CH = HD / 16, NQ = (CH + 3) / 4, NDT = HD / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* lutE = reinterpret_cast<long long*>(smem);                              // [258] exp_int
  double* lutFR = reinterpret_cast<double*>(smem + 258 * 8);                         // [258] fp64 reciprocal of float(exp_int)
  unsigned short* sVt = reinterpret_cast<unsigned short*>(smem + 2 * 258 * 8);       // [HD][AS_VSTRIDE] bf16: V^T of one 32-key pair
  unsigned* sCodes = reinterpret_cast<unsigned*>(smem + 2 * 258 * 8 + HD * AS_VSTRIDE * 2);   // [key blocks (even)][64 lanes] packed negated codes + 127

  const int lane = threadIdx.x, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x / a.H, head = blockIdx.x % a.H, qb = blockIdx.y;
  const int N = a.N, D = a.H * HD, ld = 3 * D;
  const int nkb = (N + 15) >> 4, nkp = (nkb + 1) >> 1;
  const int8_t* base = a.qkv + (long long)b * N * ld + head * HD;

  // exp table (as in k_lis_attention): entry 256 = sentinel of padded keys
  for (int t = lane; t < 256; t += 64) {
    int xi = -t;
    const int lim = 32 * a.at.x0_int;
    xi = xi < lim ? lim : xi;
    const int q = xi / a.at.x0_int;
    const int r = xi - a.at.x0_int * q;
    const long long z = (long long)r * (r + a.at.b_int) + a.at.c_int;
    long long e = z << (32 - q);
    e = e < 0 ? 0 : e;
    lutE[t] = e;
    lutFR[t] = 1.0 / (double)(float)e;
  }
  if (lane == 0) {
    lutE[256] = 0;
    lutFR[256] = 1.0;                                                                // sum / 1 >= 2^32 -> k clamps to 16 -> probability 0
  }
  const int qrow = qb * 16 + l15;
  const int qload = qrow < N ? qrow : N - 1;
  v4i fq[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    fq[j] = (v4i){0, 0, 0, 0};
    if (4 * j + g < CH) fq[j] = *reinterpret_cast<const v4i*>(base + (long long)qload * ld + (4 * j + g) * 16);
  }
  const float nmm = -(a.at.qk_scale * (a.at.s_qkv_sq * a.at.inv_s_attn));            // the NEGATED code is produced, as in k_lis_attention
  const float avm = a.at.av_mul * P2V_PROB_SCALE;

  // ---- pass 1: scores -> negated qact_attn1 codes nc in [-127, 128], stored as nc + 127 in a byte; row minimum of nc over the real keys
  int mn = 1000;
  for (int kb = 0; kb < 2 * nkp; ++kb) {
    const int krow = kb * 16 + l15;
    const int kload = krow < N ? krow : N - 1;
    v4i s = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      v4i fk = {0, 0, 0, 0};
      if (4 * j + g < CH && kb < nkb) fk = *reinterpret_cast<const v4i*>(base + (long long)kload * ld + D + (4 * j + g) * 16);
      s = __builtin_amdgcn_mfma_i32_16x16x64_i8(fk, fq[j], s, 0, 0, 0);
    }
    unsigned packed = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nc = (int)__builtin_amdgcn_fmed3f(rintf((float)s[r] * nmm), -127.f, 128.f);
      const bool real = kb * 16 + 4 * g + r < N;
      mn = (real && nc < mn) ? nc : mn;
      packed |= (unsigned)(nc + 127) << (8 * r);
    }
    sCodes[kb * 64 + lane] = packed;
  }
  {
    int o = __shfl_xor(mn, 16);
    mn = o < mn ? o : mn;
    o = __shfl_xor(mn, 32);
    mn = o < mn ? o : mn;
  }
  __syncthreads();                                                                    // tables and codes visible (one wave: cheap)

  // ---- pass 2: d = nc - mn in [0, 255] (256 for padding), exact int64 sum of exp_int
  long long S = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    const unsigned packed = sCodes[kb * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nc = (int)((packed >> (8 * r)) & 255u) - 127;
      const int d = kb * 16 + 4 * g + r < N ? nc - mn : 256;
      S += lutE[d];
    }
  }
  S += __shfl_xor(S, 16);
  S += __shfl_xor(S, 32);
  const double Sd = (double)(float)S;                                                 // exp_int.sum(-1): exact, then one rounding

  // ---- pass 3: probabilities and P.V, one pair of key blocks (32 keys) at a time
  v4f o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < nkp; ++p) {
    __syncthreads();                                                                  // the previous pair's fragments have been read
    for (int i = lane; i < 32 * CH; i += 64) {                                        // V rows of the pair, transposed to bf16 (exact)
      const int row = i / CH, c = i % CH, key = p * 32 + row;
      uint4 vv = make_uint4(0, 0, 0, 0);
      if (key < N) vv = *reinterpret_cast<const uint4*>(base + (long long)key * ld + 2 * D + c * 16);
      const unsigned w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float f = (float)sx8(w4[j >> 2], j & 3);
        sVt[(c * 16 + j) * AS_VSTRIDE + row] = (unsigned short)(__float_as_uint(f) >> 16);
      }
    }
    __syncthreads();
    unsigned pk[4];
    const unsigned pc0 = sCodes[(2 * p) * 64 + lane], pc1 = sCodes[(2 * p + 1) * 64 + lane];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float ratio[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * e2 + e, half = j >> 2, r = j & 3;
        const int key = p * 32 + 16 * half + 4 * g + r;
        const int nc = (int)(((half ? pc1 : pc0) >> (8 * r)) & 255u) - 127;
        const int d = key < N ? nc - mn : 256;
        ratio[e] = rintf((float)(Sd * lutFR[d]));                                     // correctly rounded fp32 quotient, see k_lis_attention
        if (TAP && key < N && qrow < N) {
          const int k = (int)((__float_as_uint(ratio[e]) + 0x00400000u) >> 23) - 127; // log_round, layers.py:323-329
          a.probs_k[(((long long)b * a.H + head) * N + qrow) * N + key] = (int8_t)(k > 16 ? 16 : k);
        }
      }
      pk[e2] = lis_prob_pair(ratio[0], ratio[1]);
    }
    const v4i pb = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
    const v8bf fb = __builtin_bit_cast(v8bf, pb);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const unsigned short* vp = sVt + (dt * 16 + l15) * AS_VSTRIDE + 4 * g;
      const uint2 lo = *reinterpret_cast<const uint2*>(vp);
      const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
      const v4i va = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, va), fb, o[dt], 0, 0, 0);
    }
  }
  if (qrow < N) {
    int8_t* dst = a.out + ((long long)b * N + qrow) * D + head * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      *reinterpret_cast<unsigned*>(dst + dt * 16) = pack4_rne_sat(o[dt][0] * avm, o[dt][1] * avm, o[dt][2] * avm, o[dt][3] * avm);
  }
}

template <int HD>
static int launch_stream_t(const AttnArgs& a, hipStream_t st) {
  const int nkb = (a.N + 15) / 16, nkp = (nkb + 1) / 2;
  const size_t smem = 2 * 258 * 8 + (size_t)HD * AS_VSTRIDE * 2 + (size_t)2 * nkp * 64 * 4;
  const dim3 grid((unsigned)(a.B * a.H), (unsigned)nkb), block(64);
  auto launch = [&](auto kernel) -> int {
    if (smem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kernel, grid, block, smem, st, a);
    CHECK_LAUNCH();
    return 0;
  };
  return a.probs_k ? launch(&k_lis_attention_stream<HD, true>) : launch(&k_lis_attention_stream<HD, false>);
}

// tokens beyond p2v_max_tokens(head_dim), up to P2V_MAX_TOKENS_STREAMED
int p2v_launch_attention_stream(const AttnArgs& a, int head_dim, hipStream_t st) {
  if (a.N > P2V_MAX_TOKENS_STREAMED || a.B * a.H <= 0) return -1;
  switch (head_dim) {
    case 32: return launch_stream_t<32>(a, st);
    case 48: return launch_stream_t<48>(a, st);
    case 64: return launch_stream_t<64>(a, st);
    case 80: return launch_stream_t<80>(a, st);
    case 96: return launch_stream_t<96>(a, st);
    case 128: return launch_stream_t<128>(a, st);
    default: return -1;
  }
}
