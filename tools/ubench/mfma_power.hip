// What the int8 matrix pipe sustains under the board's power limit (gfx950): v_mfma_i32_32x32x32_i8 back to back on every SIMD for
// ~1 s per variant, wall-clock TOP/s and the in-kernel clock (s_memtime / s_memrealtime), with
//   (a) constant operands (every lane the same small integers: the issue-rate micro-benchmark of round 2, 4.96 POP/s),
//   (b) random int8 operands held in registers,
//   (c) random operands re-read from LDS before every MFMA pair (2 ds_read_b128 per 2 MFMAs, the register blocking of the layer GEMMs).
// usage: ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int n, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) int lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = (int)hash(i * 2654435761u + blockIdx.x);
  __syncthreads();
  v4i a, b;
  if (MODE == 0) { a = (v4i){3, 4, 5, 6}; b = (v4i){9, 15, 21, 27}; }
  else {
    unsigned s = hash(tid + 977 * blockIdx.x);
    a = (v4i){(int)hash(s), (int)hash(s + 1), (int)hash(s + 2), (int)hash(s + 3)};
    b = (v4i){(int)hash(s + 4), (int)hash(s + 5), (int)hash(s + 6), (int)hash(s + 7)};
  }
  v16i c[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const v4i* lp = reinterpret_cast<const v4i*>(lds) + (tid & 63);
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (MODE == 2) { a = lp[((i * 2 + u) & 7) * 64]; b = lp[(((i * 2 + u) & 7) + 8) * 64]; }
      c[2 * u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[2 * u], 0, 0, 0);
      c[2 * u + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c[2 * u + 1], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
  if (s == 123456789) out[0] = s;
  if (tid == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE> void run(const char* name, int* d, unsigned long long* clk) {
  const int n = 1 << 16, reps = 6;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256 * 4), dim3(256), 0, 0, d, n, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256 * 4), dim3(256), 0, 0, d, n, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  const double ops = (double)reps * 256 * 4 * 4 * n * 4 * 65536.0;
  printf("%-46s %.1f ms  %.0f TOP/s  in-kernel clock %.0f MHz\n", name, ms, ops / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0);
}
int main() {
  int* d; unsigned long long* clk; hipMalloc(&d, 64); hipMalloc(&clk, 64);
  run<0>("constant operands in registers", d, clk);
  run<1>("random operands in registers", d, clk);
  run<2>("random operands, ds_read_b128 x2 per MFMA pair", d, clk);
  run<1>("random operands in registers (again)", d, clk);
  return 0;
}
