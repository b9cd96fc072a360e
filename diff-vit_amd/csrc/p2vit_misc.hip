// p2vit_misc.hip -- element-wise kernels: input quantisation + im2col, cls rows, GELU threshold-table build / check, module-level
// fake-quant helpers, Swin patch-merge gather and average pool; with their launchers.
#include "p2vit_device.h"

// cell of the pre-activation y: the epilogues' index function on u = y * k (gelu_tab_offset), as an entry number
__device__ __forceinline__ unsigned gelu_tab_cell(float y, float k, int off, int cells) {
  return (unsigned)((gelu_tab_offset(y * k, (float)-off, (float)(cells - 1 - off) + 0.5f) >> 3) + off);
}
// scratch: cnt[cells] | thr[cells] | lohi[cells] | first[cells]; first[] preset to 0xFFFFFFFF, cnt[] to 0.  thr[] receives the threshold
// TIMES k (exact), the form the epilogues compare against
__global__ __launch_bounds__(256) void k_gelu_tab_sweep(float inv_s, float k, int off, int cells, unsigned* scratch, int per_thread) {
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
    pi = gelu_tab_cell(yp, k, off, cells);
  }
  for (int j = 0; j < per_thread; ++j) {
    const unsigned long long n = n0 + j;
    if (n >= 2 * P2V_F32_FINITE) break;
    const float y = f32_in_order(n);
    const int c = gelu_code_exact(y, inv_s);
    const unsigned i = gelu_tab_cell(y, k, off, cells);
    if (i != pi) first[i] = (unsigned)c & 255u;                 // first value of a cell: its code when the cell has no step
    if (n > 0 && c != pc) {
      atomicAdd(&cnt[i], 1u);
      thr[i] = __float_as_uint(y * k);
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
__global__ __launch_bounds__(256) void k_gelu_tab_check(float inv_s, float k, int off, int cells, const unsigned char* table,
                                                        unsigned long long* mismatches) {
  unsigned long long bad = 0;
  for (unsigned long long n = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; n < 2 * P2V_F32_FINITE;
       n += (unsigned long long)gridDim.x * blockDim.x * 8) {
    float y[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i >> 2][i & 3] = f32_in_order(n + i < 2 * P2V_F32_FINITE ? n + i : n);
    unsigned d[2];
    float u[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i >> 2][i & 3] = y[i >> 2][i & 3] * k;
    gelu_tab_q8x8(u[0], u[1], table + off * 8, (float)-off, (float)(cells - 1 - off) + 0.5f, d[0], d[1]);          // the lookup of the GEMM epilogues
#pragma unroll
    for (int i = 0; i < 8; ++i) bad += (sx8(d[i >> 2], i & 3) != gelu_code_exact(y[i >> 2][i & 3], inv_s)) ? 1 : 0;
  }
  if (bad) atomicAdd(mismatches, bad);
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

int p2v_launch_patchify(const float* img, int B, int C, int H, int W, int P, float inv_s, int8_t* out, int k_pad, hipStream_t st) {
  const long long rows = (long long)B * (H / P) * (W / P);
  const int rows_per_block = 8;
  hipLaunchKernelGGL(k_quantize_patchify, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, img, B, C, H, W, P,
                     inv_s, out, k_pad, rows_per_block);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_fill_cls(int8_t* x, int B, int T, int D, const int8_t* cls, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_cls, dim3((B * D + 255) / 256), dim3(256), 0, st, x, B, T, D, cls);
  CHECK_LAUNCH();
  return 0;
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
  hipLaunchKernelGGL(k_gelu_tab_sweep, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, inv_s, t.k, (int)t.off, cells,
                     scratch, per_thread);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_gelu_tab_finish, dim3((cells + 255) / 256), dim3(256), 0, st, cells, scratch,
                     reinterpret_cast<uint2*>(const_cast<void*>(t.table)), scratch + 4 * cells);
  CHECK_LAUNCH();
  return 0;
}

int p2v_launch_gelu_table_check(float inv_s, const p2v_gelu_tab& t, unsigned long long* mismatches, hipStream_t st) {
  hipLaunchKernelGGL(k_gelu_tab_check, dim3(8192), dim3(256), 0, st, inv_s, t.k, (int)t.off, t.cells,
                     reinterpret_cast<const unsigned char*>(t.table), mismatches);
  CHECK_LAUNCH();
  return 0;
}

// ---- stream probe: `kernels` launches of one wave that waits `usec` microseconds on the constant-rate counter ------------------------------
// No memory traffic, one wave: two streams whose hardware queues are serviced concurrently finish a train of these in the time of one train;
// queues that share a dispatch pipe (or streams that share a queue) take the sum.  The host side (engine.side_streams) uses it to pick side
// streams for the batch slicing that really run beside the caller's stream.  Every wave leaves the loop: the counter advances on its own.
__global__ __launch_bounds__(256) void k_stream_probe(long long ticks) {
  extern __shared__ unsigned char probe_lds[];
  if (ticks < 0) probe_lds[threadIdx.x] = 0;                  // (keeps the dynamic LDS allocation)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

int p2v_launch_stream_probe(int kernels, int usec, int workgroups, int lds_bytes, hipStream_t st) {
  int dev = 0, khz = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (e != hipSuccess) return (int)e;
  if (khz <= 0) khz = 100000;
  const long long ticks = (long long)usec * khz / 1000;
  for (int i = 0; i < kernels; ++i) hipLaunchKernelGGL(k_stream_probe, dim3(workgroups), dim3(256), lds_bytes, st, ticks);
  CHECK_LAUNCH();
  return 0;
}
