// Sustained ceiling of the six-product (fp32-class) arithmetic on the bf16 matrix pipe: waves that do nothing
// but independent v_mfma_f32_32x32x16_bf16 on register-resident operands (no memory traffic in the loop),
// 1..2 waves per SIMD, long enough for the clocks to settle under the power limit.  Two data patterns: zeros
// (what a spec-sheet peak is measured on) and random normal-like bf16 values (what a GEMM feeds the pipe).
// Prints TFLOP/s of bf16 MFMA, the fp32-class equivalent (/ 6) and the clock the rate implies for a fully
// busy pipe (a 32x32x16 bf16 MFMA occupies its SIMD's matrix core for 32 cycles: 256 CUs x 4 SIMDs x
// 32768 FLOP / 32 cycles = 1048.6 kFLOP per clock; 2.4 GHz -> 2516 TFLOP/s).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ src, float* out, int iters) {
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // six operand pairs per lane, loaded once (the three pieces of A and of B)
  bf16x8 a[3], b[3];
  for (int j = 0; j < 3; ++j) {
    a[j] = src[(threadIdx.x * 6 + j) & 4095];
    b[j] = src[(threadIdx.x * 6 + 3 + j) & 4095];
  }
  for (int i = 0; i < iters; ++i) {
    // the six products with i + j <= 2, smallest terms first, on SIX accumulators (no dependent chain)
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[5], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 6; ++j)
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 12345.678f) out[0] = s;
}

static unsigned short f2bf(float f) {
  union { float f; unsigned u; } cv; cv.f = f; unsigned u = cv.u;
  return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main() {
  const int N = 4096 * 8;
  unsigned short* h = (unsigned short*)malloc(N * 2);
  void* d; float* o;
  hipMalloc(&d, N * 2); hipMalloc((void**)&o, 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int pattern = 0; pattern < 2; ++pattern) {
    srand(1);
    for (int i = 0; i < N; ++i) {
      float v = 0.f;
      if (pattern) { for (int r = 0; r < 12; ++r) v += rand() / (float)RAND_MAX; v -= 6.f; }   // ~N(0,1)
      h[i] = f2bf(v);
    }
    hipMemcpy(d, h, N * 2, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps, iters = 400000;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, 4000);
      hipDeviceSynchronize();
      hipEventRecord(s);
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, iters);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      const double flops = 20.0 * blocks * 4.0 * iters * 6.0 * (2.0 * 32 * 32 * 16);
      const double tf = flops / (ms * 1e-3) / 1e12;
      printf("%-6s operands, %d wave(s)/SIMD: %8.1f ms  %7.1f TFLOP/s bf16 MFMA = %6.1f TFLOP/s fp32-class (six products)"
             "  implied clock %.2f GHz\n", pattern ? "random" : "zero", wps, ms, tf, tf / 6.0, tf * 1e12 / 1048576.0 / 1e9);
    }
  }
  return 0;
}
