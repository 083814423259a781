// Sustained fp32 MFMA ceiling of the machine: waves that do nothing but independent
// v_mfma_f32_32x32x2_f32 (no memory traffic), 1..4 waves per SIMD, for long enough to settle clocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 1.f; a2[e] = 2.f; a3[e] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD
    const int iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double flops = 5.0 * blocks * 4.0 * iters * 16.0 * (2.0 * 32 * 32 * 2);
    printf("%d wave(s)/SIMD: %.1f ms  %.1f TFLOP/s\n", wps, ms, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
