// Do VALU instructions hide behind MFMAs on one SIMD?  Per iteration and wave: 48 v_mfma_f32_32x32x16_bf16
// (4 accumulators) and 192 integer VALU instructions (v_alignbit_b32 + v_add_u32, 8 independent chains), 1 or 2 waves per SIMD, three schedules:
//   phased       48 MFMAs, then 192 VALU (what a "split, barrier, MFMA" slab loop does; overlap only ACROSS waves)
//   interleaved  every MFMA followed by its 4 VALU instructions in the SAME wave's stream
//   mfma / valu  each alone
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f2g_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 f2g_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  const f2g_f32x2 x = {x0, x1};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f2g_bf16x2));
  const f2g_f32x2 r1 = x - f2g_f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f2g_bf16x2));
  const f2g_f32x2 r2 = r1 - f2g_f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, f2g_bf16x2));
}

// the split of PAIRS pairs of floats (9 VALU each) as the payload: 48 MFMAs + PAIRS splits per iteration;
// MODE 4 = splits only, 5 = phased, 6 = one pair after every (48 / PAIRS)-th MFMA
template <int MODE, int PAIRS>
__global__ __launch_bounds__(256) void ks(const bf16x8* __restrict__ src, float* out, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  bf16x8 a[3], b[3];
  for (int j = 0; j < 3; ++j) {
    a[j] = src[(threadIdx.x * 6 + j) & 4095];
    b[j] = src[(threadIdx.x * 6 + 3 + j) & 4095];
  }
  float x[PAIRS][2];
  unsigned sink = 0;
  for (int j = 0; j < PAIRS; ++j) { x[j][0] = threadIdx.x * 0.37f + j; x[j][1] = threadIdx.x * 1.13f - j; }
  for (int i = 0; i < iters; ++i) {
    if (MODE == 5) {
#pragma unroll
      for (int m = 0; m < 48; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4 || MODE == 5) {
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) {
        unsigned p0, p1, p2;
        split3_pair(x[j][0], x[j][1], p0, p1, p2);
        sink ^= p0 ^ p1;
        x[j][0] = __uint_as_float((p2 << 16) | 0x3f000000u); x[j][1] = __uint_as_float((p2 & 0xffff0000u) | 0x3f00u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 6) {
      constexpr int EVERY = 48 / PAIRS;
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[m & 3], 0, 0, 0);
        if (m % EVERY == 0) {
          const int j = m / EVERY;
          unsigned p0, p1, p2;
          split3_pair(x[j][0], x[j][1], p0, p1, p2);
          sink ^= p0 ^ p1;
          x[j][0] = __uint_as_float((p2 << 16) | 0x3f000000u); x[j][1] = __uint_as_float((p2 & 0xffff0000u) | 0x3f00u);
        }
      }
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (PAIRS * 13 + 47) / 48, 0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int j = 0; j < PAIRS; ++j) s += x[j][0] + x[j][1];
  if (s == 12345.678f || sink == 0x12345u) out[0] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ src, float* out, int iters, unsigned c1, unsigned c2) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  bf16x8 a[3], b[3];
  for (int j = 0; j < 3; ++j) {
    a[j] = src[(threadIdx.x * 6 + j) & 4095];
    b[j] = src[(threadIdx.x * 6 + 3 + j) & 4095];
  }
  unsigned v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 77u + j;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 48; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[m & 3], 0, 0, 0);
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 96; ++m) v[m & 7] = __builtin_amdgcn_alignbit(v[m & 7], v[m & 7], 7) + c1;
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) v[(2 * m + q) & 7] = __builtin_amdgcn_alignbit(v[(2 * m + q) & 7], v[(2 * m + q) & 7], 7) + c1;
      }
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int j = 0; j < 8; ++j) s += (float)v[j];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, const void* d, float* o) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, 200, 12345u, 7u);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, iters, 12345u, 7u);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    // per SIMD: wps waves x iters x 5 launches iterations
    const double ns_per_iter = ms * 1e6 / (5.0 * iters * wps);
    printf("%-12s %d wave(s)/SIMD: %8.2f ms   %7.1f ns per (48 MFMA + 192 VALU) wave-iteration on its SIMD\n", name, wps, ms, ns_per_iter);
  }
}

template <int MODE, int PAIRS>
static void runs(const char* name, const void* d, float* o) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 20000;
    hipLaunchKernelGGL((ks<MODE, PAIRS>), dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, 200);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((ks<MODE, PAIRS>), dim3(blocks), dim3(256), 0, 0, (const bf16x8*)d, o, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-22s %2d pairs, %d wave(s)/SIMD: %8.2f ms   %7.1f ns per wave-iteration on its SIMD\n", name, PAIRS, wps, ms, ms * 1e6 / (5.0 * iters * wps));
  }
}

int main() {
  void* d; float* o;
  hipMalloc(&d, 4096 * 16 * 2); hipMemset(d, 0x3c, 4096 * 16 * 2); hipMalloc((void**)&o, 4);
  run<0>("mfma only", d, o);
  run<1>("valu only", d, o);
  run<2>("phased", d, o);
  run<3>("interleaved", d, o);
  runs<4, 8>("split only", d, o);
  runs<5, 8>("split phased", d, o);
  runs<6, 8>("split interleaved", d, o);
  runs<4, 16>("split only", d, o);
  runs<5, 16>("split phased", d, o);
  runs<6, 16>("split interleaved", d, o);
  return 0;
}
