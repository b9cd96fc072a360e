// MFMA issue-rate micro-benchmark for gfx950: cycles per v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x32_bf16 of ONE
// wave with 1, 2 or 4 independent accumulators, at 1, 2 and 4 waves per SIMD; plus the wall-clock int8 rate of the whole GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define ITER 1024
template <int OP, int NACC>
__global__ void k(float* out, int n, int seed) {
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 9};
  v16i c[4]; v4i d[4]; v4f f[4];
  for (int j = 0; j < 4; ++j) { for (int r = 0; r < 16; ++r) c[j][r] = 0; d[j] = (v4i){0, 0, 0, 0}; f[j] = (v4f){0, 0, 0, 0}; }
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if (OP == 0) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[j], 0, 0, 0);
        if (OP == 1) d[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d[j], 0, 0, 0);
        if (OP == 2) f[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), f[j], 0, 0, 0);
      }
  }
  long long t1 = clock64();
  int s = 0; float fs = 0;
  for (int j = 0; j < 4; ++j) { for (int r = 0; r < 16; ++r) s += c[j][r]; s += d[j][0] + d[j][3]; fs += f[j][0] + f[j][3]; }
  if (s + (int)fs == 123456789) out[0] = s;
  // the slowest wave of the grid (waves of one SIMD are served oldest first)
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out) + 1, __float_as_uint((float)(t1 - t0) / (n * 4)));
}
template <int OP, int NACC> void run(const char* name, float* d) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL((k<OP, NACC>), dim3(256), dim3(256 * wps), 0, 0, d, ITER, 3);
    hipError_t er = hipDeviceSynchronize(); if (er != hipSuccess || hipGetLastError() != hipSuccess) printf("launch error %s\n", hipGetErrorString(er));
    float h[2]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s accumulators=%d waves/SIMD=%d: %.1f ticks per MFMA of one wave, %.1f per SIMD\n", name, NACC, wps, h[1], h[1] / wps);
  }
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  run<0, 1>("v_mfma_i32_32x32x32_i8", d); run<0, 2>("v_mfma_i32_32x32x32_i8", d); run<0, 4>("v_mfma_i32_32x32x32_i8", d);
  run<1, 1>("v_mfma_i32_16x16x64_i8", d); run<1, 4>("v_mfma_i32_16x16x64_i8", d);
  run<2, 1>("v_mfma_f32_16x16x32_bf16", d); run<2, 4>("v_mfma_f32_16x16x32_bf16", d);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<0, 4>), dim3(256 * 4), dim3(256 * wps), 0, 0, d, ITER * 8, 3);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 10.0 * 256 * 4 * (4 * wps) * ITER * 8 * 4 * 65536.0;
    printf("wall: 32x32x32 i8, waves/SIMD=%d: %.3f ms -> %.0f TOP/s\n", wps, ms, ops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
