// p2vit_gemm.hip -- tiled int8 MFMA GEMMs (stem / head, fp32-image patch embedding, the LDS-DMA layer GEMM) and their launcher.
#include "p2vit_epilogue.h"

#ifdef P2V_DIAG
unsigned long long* g_gemm_stamps = nullptr;
#endif

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
//   barrier.
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
template <int EPI, bool W4, int MT>
__global__ __launch_bounds__(128 * MT, MT == 4 ? 4 : 3) void k_gemm_dma(GemmArgs g) {
  constexpr int NST = 3;                                         // stages of the LDS-DMA ring
  constexpr int TBM = 64 * MT;                                   // tile rows
  constexpr int NTH = 128 * MT;                                  // threads
  constexpr int STAGE = (TBM + GBN) * GBK;                       // X tile + W tile (a packed int4 W tile fills half of its 8 KB)
  constexpr bool RES = EPI == P2V_EPI_RESID || EPI == P2V_EPI_RESID_PRE;
  constexpr int EPI_BYTES = EPI == P2V_EPI_RESID ? (int)sizeof(EpiLds) : (EPI == P2V_EPI_RESID_PRE ? (int)sizeof(ResidLds) : 2 * GBN * (int)sizeof(float));   // else colscale + bias only
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
  if constexpr (EPI == P2V_EPI_RESID_PRE) {      // the tile's six pre-folded arrays: one contiguous 3 KB block of the table
    if (tid < P2V_RESID_TAB_ARRAYS * GBN / 4)
      reinterpret_cast<float4*>(sE)[tid] = reinterpret_cast<const float4*>(g.ep.resid_tab + (long long)tn * (P2V_RESID_TAB_ARRAYS * GBN))[tid];
  } else {
    gemm_stage_epilogue<EPI>(sE, n0, tid, g);
  }
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
  if (RES && MT == 2) {
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
    /* in flight behind tile KT: tile KT+1 (PCS requests of this wave) */                                            \
    if ((KT) + 1 < nk) { if (PCS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
    asm volatile("s_barrier" ::: "memory");    /* tile KT landed for everyone; everyone is done reading tile KT-1 */  \
    if ((KT) + 2 < nk) dma(((S) + 2) % 3, (KT) + 2);                                                                 \
    if (W4) gemm_compute_tile_dma_w4<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                   \
    else gemm_compute_tile_dma<(S) * STAGE, FIRST>(aX0, aX1, aW0, aW1, acc);                                         \
  } while (0)
  // the first k-tile is peeled: its MFMAs start the sums from the literal 0 (64 accumulator registers are never cleared)
  P2V_KTILE_(0, 0, true);
  if (1 < nk) P2V_KTILE(1, 1);
  if (2 < nk) P2V_KTILE(2, 2);
  for (int kt = 3; kt < nk; kt += 3) {
    P2V_KTILE(0, kt);
    if (kt + 1 < nk) P2V_KTILE(1, kt + 1);
    if (kt + 2 < nk) P2V_KTILE(2, kt + 2);
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
    if (RES && MT == 4) load_resid(ni);                                // (held across the other group's epilogue they would spill)
    if constexpr (EPI == P2V_EPI_RESID_PRE)
      gemm_epilogue_resid_pre<MT == 4>(acc[ni], m0 + wm * 64 + l31, n0 + wn * 64 + ni * 32, wn * 64 + ni * 32, h, g,
                                       reinterpret_cast<const ResidLds*>(sE), resv[ni]);
    else
      gemm_epilogue_tile2<EPI, (MT == 4 && EPI == P2V_EPI_RESID)>(acc[ni], m0 + wm * 64 + l31, n0 + wn * 64 + ni * 32, wn * 64 + ni * 32, h, g, sE,
                                                                   resv[ni], dyn_lds);
  }
  GD_STAMP(4);
}


// ---------------------------------------------------------------------------------------------------
// host launchers (called from the C ABI in p2vit_capi.cpp)
// ---------------------------------------------------------------------------------------------------
int p2v_launch_embed_fp32(const float* img, int B, int C, int H, int W, int P, const GemmArgs& g, hipStream_t st) {
  if (g.w4 || g.N % 4 || P % 4) return -1;
  const dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64));
  hipLaunchKernelGGL(k_embed_fp32, grid, dim3(256), 0, st, img, B, C, H, W, P, g);
  CHECK_LAUNCH();
  return 0;
}

int g_resid_pre = 1;      // P2V_RESID_PRE=0: ignore p2v_epilogue.resid_tab (A/B and parity runs of the generic RESID epilogue)
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
    const bool big = (g_gemm_tile == 256 || (g_gemm_tile == 0 && !g.w4 && tiles256 >= 2LL * device_cus()));
    const int tiles_m = big ? (g.M + 255) / 256 : (g.M + GBM - 1) / GBM;
    dim3 grid4(g.tiles_n * tiles_m), block4(big ? 512 : 256);
    unsigned tab_bytes = (epi == P2V_EPI_GELU && g.ep.gelu.table) ? (unsigned)g.ep.gelu.cells * 8u : 0u;
    // static LDS of the GELU_TAB instantiations (ring + column constants) plus the table can pass the 64 KB a kernel gets by default
    // (1/scale = 256: 2111 cells = 16.5 KB): ask for the larger dynamic block once per process and device, or use the arithmetic epilogue
    if (tab_bytes) {
      const int stat = (big ? 3 * (256 + GBN) * GBK : 3 * DMA_STAGE_BYTES) + 2 * GBN * (int)sizeof(float);
      if (stat + (int)tab_bytes > 64 * 1024) {
        bool ok;
        if (big) ok = g.w4 ? grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, true, 4>, 0, (int)tab_bytes)
                           : grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, false, 4>, 1, (int)tab_bytes);
        else if (g.w4) ok = grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, true, 2>, 2, (int)tab_bytes);
        else ok = grant_dynamic_lds(&k_gemm_dma<P2V_EPI_GELU_TAB, false, 2>, 4, (int)tab_bytes);
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
      case P2V_EPI_RESID:                                                                                                 \
        if (g.ep.resid_tab && g_resid_pre) hipLaunchKernelGGL(KERNEL(P2V_EPI_RESID_PRE), grid4, block4, 0, st, g);        \
        else hipLaunchKernelGGL(KERNEL(P2V_EPI_RESID), grid4, block4, 0, st, g);                                          \
        break;                                                                                                            \
      default: return -1;                                                                                                 \
    }
#define P2V_K_DMA3(E) (k_gemm_dma<E, false, 2>)
#define P2V_K_DMA3P(E) (k_gemm_dma<E, true, 2>)
#define P2V_K_DMA3L(E) (k_gemm_dma<E, false, 4>)
#define P2V_K_DMA3PL(E) (k_gemm_dma<E, true, 4>)
    if (big) {
      if (g.w4) { P2V_LAUNCH_TILED(P2V_K_DMA3PL) }
      else { P2V_LAUNCH_TILED(P2V_K_DMA3L) }
    } else if (g.w4) { P2V_LAUNCH_TILED(P2V_K_DMA3P) }          // packed int4 weights: the LDS-DMA kernel only
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


// ---------------------------------------------------------------------------------------------------
// Table builder of the pre-folded RESID epilogue (p2v_resid_prefold): one workgroup per output channel.  Writes the channel's six
// constants and runs the exhaustive check of the test-free second quotient: all 256 x 256 (residual code, q3 code) numerators through
// the reference's own operations (three roundings, IEEE division, round-half-even, clamp) and through the epilogue's form.
// flags[0] &= every channel usable;  flags[1] += mismatching numerators (diagnostic).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resid_prefold(const float* __restrict__ colscale, const float* __restrict__ bias, int w4,
                                                       const float* __restrict__ s_mid, const float* __restrict__ s_res,
                                                       const float* __restrict__ s_next, int N, float* __restrict__ tab, unsigned* flags) {
  const int n = blockIdx.x, tid = threadIdx.x;
  float* t6 = tab + (long long)(n / GBN) * (P2V_RESID_TAB_ARRAYS * GBN) + (n % GBN);
  if (n >= N) {                              // padding columns of the last tile: harmless values, never stored
    if (tid == 0) { t6[0] = 0.f; t6[GBN] = 0.f; t6[2 * GBN] = 1.f; t6[3 * GBN] = 1.f; t6[4 * GBN] = 1.f; t6[5 * GBN] = 0.f; }
    return;
  }
  const float cs = colscale[n] * (w4 ? 0.0625f : 1.0f), bs = bias[n], sm = s_mid[n], sr = s_res[n], sn = s_next[n];
  const float rm = 1.0f / sm;
  const float rh = 1.0f / sn;
  const float rl = (float)(1.0 / (double)sn - (double)rh);
  bool ok = sm > 0.f && sr > 0.f && sn > 0.f && sm < 1.0e30f && sr < 1.0e30f && sn < 1.0e30f && sm > 1.0e-30f && sr > 1.0e-30f && sn > 1.0e-30f;
  {   // (i) the folded first quotient: colscale a power of two well inside the normal range (its product with fl(1/s_mid) is then exact)
      //     and |bias / s_mid| <= 512 (error bound of the folded form, include/p2vit.h)
    const unsigned cb = __float_as_uint(cs);
    ok = ok && (cb & 0x807FFFFFu) == 0u && (cb >> 23) - 32u <= 190u && fabsf(bs * rm) <= 512.f;
  }
  unsigned bad = 0;
  if (ok) {
    const float q3 = (float)(tid - 128);
    const float a = q3 * sm;
#pragma unroll 4
    for (int rc = -128; rc < 128; ++rc) {
      const float xs = (float)rc * sr + a;
      const float ref = clamp8f(rintf(xs / sn));                                                    // the reference: IEEE division
      const float c = pre_pack(resid_q2(xs, rh, rl));
      const int got = sx8(pack4_pre(c, c, c, c), 0);                                                // the epilogue: bytes as packed
      bad += (int)ref != got ? 1u : 0u;
    }
  }
  __shared__ unsigned s_bad;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  if (bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (tid == 0) {
    if (!ok || s_bad) atomicAnd(&flags[0], 0u);
    if (s_bad) atomicAdd(&flags[1], s_bad);
    t6[0] = cs * rm; t6[GBN] = bs * rm; t6[2 * GBN] = sm; t6[3 * GBN] = sr; t6[4 * GBN] = rh; t6[5 * GBN] = rl;
  }
}

// flags: dev [2], preset by the caller to {1, 0}
int p2v_launch_resid_prefold(const p2v_linear& lin, const p2v_epilogue& ep, int N, float* tab, unsigned* flags, hipStream_t st) {
  const int n_pad = (N + GBN - 1) / GBN * GBN;
  hipLaunchKernelGGL(k_resid_prefold, dim3(n_pad), dim3(256), 0, st, lin.colscale, lin.bias, lin.packed4 ? 1 : 0, ep.s_mid, ep.s_res, ep.s_next, N, tab,
                     flags);
  CHECK_LAUNCH();
  return 0;
}
