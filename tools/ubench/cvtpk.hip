// semantics probe: v_cvt_pk_u8_f32 on out-of-range, fractional, negative, NaN, inf inputs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* x, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
  float h[] = {0.f, 1.f, 254.f, 255.f, 256.f, 300.f, 1e9f, -1.f, -0.4f, -1e9f, 0.5f, 1.5f, 2.5f, 0.49f, 0.51f, 254.5f, 255.4f, 255.6f, NAN, INFINITY, -INFINITY, 127.999f};
  int n = sizeof h / sizeof h[0];
  float* d; unsigned* o; hipMalloc(&d, sizeof h); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  unsigned r[64]; hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], r[i] & 255u);
  return 0;
}
