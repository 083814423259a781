// GEMM mainloop laboratory (not part of the library): C[M,N] = A[M,K] * W[N,K]^T, fp32 MFMA
// (v_mfma_f32_32x32x2_f32), 128x128x32 tiles, 4 waves of 64x64.  One kernel template with
// ablation / candidate variants, timed interleaved in one process on random data:
//   0 BASE     register-staged double buffering, one barrier per slab (the library's structure)
//   1 NOLOAD   no global loads / LDS stores inside the K loop (MFMA + ds_read + barrier)
//   2 NOSTORE  global loads issued and kept alive, no LDS stores
//   3 MFMAONLY no LDS reads either (operands stay in registers)
//   4 GLDS     global_load_lds_dwordx4 straight into an XOR-swizzled LDS image, 2 buffers
//   5 GLDS3    same with 3 buffers and counted vmcnt (loads span a barrier)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_lab.hip -o tools/micro/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string.h>
#include <dlfcn.h>
#include <string>
#include "../../include/flow2gan_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, BM = 128, BN = 128, LDR = BK + 4;

__device__ __forceinline__ void tile_of_block(int& m0, int& n0) {
  const int tiles_n = gridDim.y, tiles_m = gridDim.x;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.y * tiles_m + blockIdx.x;
  const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  m0 = tm * BM;
  n0 = tn * BN;
}

template <int VAR>
__global__ __launch_bounds__(256, 2) void lab(const float* __restrict__ A, const float* __restrict__ W,
                                              float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool GL = VAR >= 4;
  constexpr int NBUF = VAR == 5 ? 3 : 2;
  constexpr int ASZ = GL ? BM * BK : BM * LDR;
  constexpr int BSZ = GL ? BN * BK : BN * LDR;
  float* As = smem;
  float* Bs = smem + NBUF * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(m0, n0);
  const int nt = K / BK;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- register-staged loader (VAR 0..3): thread -> 4 chunks of A, 4 of B per slab
  const int ch = tid & 7, rr = tid >> 3;  // 8 chunks per row, 32 rows per pass
  const float* pa[4];
  const float* pb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int ra = m0 + rr + 32 * q; ra = ra < M ? ra : M - 1;
    int rb = n0 + rr + 32 * q; rb = rb < N ? rb : N - 1;
    pa[q] = A + (long long)ra * K + ch * 4;
    pb[q] = W + (long long)rb * K + ch * 4;
  }
  // ---- direct-to-LDS loader (VAR 4, 5): LDS chunk p = row*8 + c', holds source chunk
  // c = c' ^ ((row>>1)&7); a wave instruction fills 64 consecutive chunks (8 rows)
  const float* ga[4];
  const float* gb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = (q * 4 + wave) * 64 + lane;
    const int row = p >> 3, cs = (p & 7) ^ ((row >> 1) & 7);
    int ra = m0 + row; ra = ra < M ? ra : M - 1;
    int rb = n0 + row; rb = rb < N ? rb : N - 1;
    ga[q] = A + (long long)ra * K + cs * 4;
    gb[q] = W + (long long)rb * K + cs * 4;
  }
  auto glds = [&](int t, int buf) {
    float* ad = As + buf * ASZ;
    float* bd = Bs + buf * BSZ;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds(ga[q] + t * BK, (__attribute__((address_space(3))) void*)(ad + (q * 4 + wave) * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb[q] + t * BK, (__attribute__((address_space(3))) void*)(bd + (q * 4 + wave) * 256), 16, 0, 0);
    }
  };
  auto mfma_slab = [&](const float* Ab, const float* Bb) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float a[2][4], b[2][4];
      const int kk = h * 16 + s4 * 4;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = (wm * 2 + mi) * 32 + li;
        float4 tv;
        if (VAR == 3) tv = make_float4(Ab[0], Ab[1], Ab[2], Ab[3]);
        else if (GL) tv = *reinterpret_cast<const float4*>(Ab + row * BK + (((kk >> 2) ^ ((row >> 1) & 7)) << 2));
        else tv = *reinterpret_cast<const float4*>(Ab + row * LDR + kk);
        a[mi][0] = tv.x; a[mi][1] = tv.y; a[mi][2] = tv.z; a[mi][3] = tv.w;
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = (wn * 2 + ni) * 32 + li;
        float4 tv;
        if (VAR == 3) tv = make_float4(Bb[0], Bb[1], Bb[2], Bb[3]);
        else if (GL) tv = *reinterpret_cast<const float4*>(Bb + col * BK + (((kk >> 2) ^ ((col >> 1) & 7)) << 2));
        else tv = *reinterpret_cast<const float4*>(Bb + col * LDR + kk);
        b[ni][0] = tv.x; b[ni][1] = tv.y; b[ni][2] = tv.z; b[ni][3] = tv.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni], 0, 0, 0);
    }
  };

  float extra[16];
  if (VAR == 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) extra[i] = A[tid + i * 64];
  }
  if (!GL) {
    float4 ra[4], rb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const float4*>(pa[q]);
      rb[q] = *reinterpret_cast<const float4*>(pb[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(As + (rr + 32 * q) * LDR + ch * 4) = ra[q];
      *reinterpret_cast<float4*>(Bs + (rr + 32 * q) * LDR + ch * 4) = rb[q];
    }
    __syncthreads();
    float regA[4] = {As[tid], As[tid + 1], As[tid + 2], As[tid + 3]};
    float regB[4] = {Bs[tid], Bs[tid + 1], Bs[tid + 2], Bs[tid + 3]};
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      float4 la[4], lb[4];
      if (VAR == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(extra[i]));
      }
      if (VAR == 0 || VAR == 2) {
        const int kn = (t + 1 < nt) ? (t + 1) * BK : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          la[q] = *reinterpret_cast<const float4*>(pa[q] + kn);
          lb[q] = *reinterpret_cast<const float4*>(pb[q] + kn);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (VAR == 3) mfma_slab(regA, regB);
      else if (VAR == 0 || VAR == 2) mfma_slab(As + cur * ASZ, Bs + cur * BSZ);
      else mfma_slab(As, Bs);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR == 0 || VAR == 2) {
        if (t + 1 < nt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<float4*>(As + (cur ^ 1) * ASZ + (rr + 32 * q) * LDR + ch * 4) = la[q];
            *reinterpret_cast<float4*>(Bs + (cur ^ 1) * BSZ + (rr + 32 * q) * LDR + ch * 4) = lb[q];
          }
        }
      }
      __syncthreads();
    }
  } else if (VAR == 4) {
    glds(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      if (t + 1 < nt) glds(t + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_slab(As + cur * ASZ, Bs + cur * BSZ);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // 3 buffers: slab t+2 is issued while slab t computes; at the end of iteration t only slab
    // t+1 must have landed (vmcnt(8) = the 8 loads of slab t+2 may stay in flight)
    glds(0, 0);
    if (nt > 1) glds(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
      int nxt2 = cur + 2; nxt2 = nxt2 >= 3 ? nxt2 - 3 : nxt2;
      if (t + 2 < nt) glds(t + 2, nxt2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_slab(As + cur * ASZ, Bs + cur * BSZ);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 >= 3 ? 0 : cur + 1;
    }
  }
  if (VAR == 2) {
    float sx = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sx += extra[i];
    if (sx == 1234.5f) C[0] = sx;
  }
  // epilogue
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + (wn * 2 + ni) * 32 + li;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + (wm * 2 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (col < N && row < M) C[(long long)row * N + col] = acc[mi][ni][e];
      }
  }
}


// ---- LEAN: the same tile / pipeline with (almost) no VALU instruction inside the K loop.
// Measured (PMC, this file): with two waves per SIMD saturating the fp32 MFMA pipe, every other
// VALU instruction issued on that SIMD costs ~37 cycles of pipe time, so address arithmetic,
// masks and on-load transforms are what separates the library's 108 TFLOP/s from the MFMA-only
// 143.  Here: uniform (SGPR) tile base + per-lane 32-bit offsets (saddr addressing, advanced by
// SALU), LDS addresses that are per-thread constants + immediates (K loop unrolled by two for
// static buffer offsets), bias folded into the accumulator initialisation.
template <int PRE>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(4, 4))) void lean(const float* __restrict__ A, const float* __restrict__ W,
                                               float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TSZ = BM * LDR;  // floats per operand buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(m0, n0);
  const int nt = K / BK;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int ch = tid & 7, rr = tid >> 3;
  unsigned offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int ra = rr + 32 * q; ra = (m0 + ra < M) ? ra : M - 1 - m0;
    int rb = rr + 32 * q; rb = (n0 + rb < N) ? rb : N - 1 - n0;
    offA[q] = (unsigned)(ra * K + ch * 4) * 4u;
    offB[q] = (unsigned)(rb * K + ch * 4) * 4u;
  }
  const char* baseA = reinterpret_cast<const char*>(A + (long long)m0 * K);
  const char* baseB = reinterpret_cast<const char*>(W + (long long)n0 * K);
  float* wA = smem + rr * LDR + ch * 4;               // + buf*TSZ + q*32*LDR
  float* wB = smem + 2 * TSZ + rr * LDR + ch * 4;
  const float* rA = smem + (wm * 64 + li) * LDR + h * 16;            // + buf*TSZ + mi*32*LDR + s4*4
  const float* rB = smem + 2 * TSZ + (wn * 64 + li) * LDR + h * 16;

  auto gload = [&](int t, float4 (&la)[4], float4 (&lb)[4]) {
    const char* a = baseA + (long long)t * (BK * 4);
    const char* b = baseB + (long long)t * (BK * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      la[q] = *reinterpret_cast<const float4*>(a + offA[q]);
      lb[q] = *reinterpret_cast<const float4*>(b + offB[q]);
    }
  };
  auto lstore = [&](int buf, const float4 (&la)[4], const float4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(wA + buf * TSZ + q * 32 * LDR) = la[q];
      *reinterpret_cast<float4*>(wB + buf * TSZ + q * 32 * LDR) = lb[q];
    }
  };
  auto mfma_slab = [&](int buf) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const float4*>(rA + buf * TSZ + mi * 32 * LDR + s4 * 4);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const float4*>(rB + buf * TSZ + ni * 32 * LDR + s4 * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const float av = q == 0 ? a[mi].x : q == 1 ? a[mi].y : q == 2 ? a[mi].z : a[mi].w;
            const float bv = q == 0 ? b[ni].x : q == 1 ? b[ni].y : q == 2 ? b[ni].z : b[ni].w;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
          }
    }
  };
  {
    float4 la[4], lb[4];
    gload(0, la, lb);
    lstore(0, la, lb);
  }
  __syncthreads();
  // K loop unrolled by two: buffer indices are compile-time constants
  auto step = [&](int t, int cur) {
    float4 la[4], lb[4];
    if (PRE) {
      gload(t + 1 < nt ? t + 1 : 0, la, lb);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_slab(cur);
    __builtin_amdgcn_sched_barrier(0);
    if (PRE) {
      lstore(cur ^ 1, la, lb);   // unconditional (the last iteration stages slab 0 again, unused)
    } else if (t + 1 < nt) {
      gload(t + 1, la, lb);
      lstore(cur ^ 1, la, lb);
    }
    __syncthreads();
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step(t, 0);
    step(t + 1, 1);
  }
  if (t < nt) step(t, 0);
  // epilogue: uniform row bases, per-lane constant offset
  const unsigned coff = (unsigned)((4 * h) * N + li) * 4u;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col0 = n0 + (wn * 2 + ni) * 32;
      const int row0 = m0 + (wm * 2 + mi) * 32;
      char* cb = reinterpret_cast<char*>(C + (long long)row0 * N + col0);
      const bool full = row0 + 32 <= M && col0 + 32 <= N;
      if (full) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          *reinterpret_cast<float*>(cb + (long long)((e & 3) + 8 * (e >> 2)) * N * 4 + coff) = acc[mi][ni][e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < M && col0 + li < N) C[(long long)row * N + col0 + li] = acc[mi][ni][e];
        }
      }
    }
}

template <int PRE>
float run_lean(const float* A, const float* W, float* C, int M, int N, int K, int iters) {
  const size_t smem = (size_t)4 * BM * LDR * 4;
  auto kern = lean<PRE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), smem, 0, A, W, C, M, N, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  return ms / iters;
}

// ---- LEANB: buffer loads (resource in SGPRs, per-thread constant VGPR offset, K advance in an
// SGPR soffset: no address VALU, out-of-range rows read zeros), loads issued before the MFMA phase.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL2>
__global__ __launch_bounds__(256, 2) void leanb(const float* __restrict__ A, const float* __restrict__ W,
                                                float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TSZ = BM * LDR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(m0, n0);
  const int nt = K / BK;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int ch = tid & 7, rr = tid >> 3;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((long long)M * K * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((long long)N * K * 4), 0x00020000);
  unsigned offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rowa = m0 + rr + 32 * q, rowb = n0 + rr + 32 * q;
    offA[q] = rowa < M ? (unsigned)(rowa * K + ch * 4) * 4u : 0x80000000u;
    offB[q] = rowb < N ? (unsigned)(rowb * K + ch * 4) * 4u : 0x80000000u;
  }
  float* wA = smem + rr * LDR + ch * 4;
  float* wB = smem + 2 * TSZ + rr * LDR + ch * 4;
  const float* rA = smem + (wm * 64 + li) * LDR + h * 16;
  const float* rB = smem + 2 * TSZ + (wn * 64 + li) * LDR + h * 16;

  auto gload = [&](int t, u32x4 (&la)[4], u32x4 (&lb)[4]) {
    const int so = t * (BK * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      la[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, offA[q], so, 0);
      lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offB[q], so, 0);
    }
  };
  auto lstore = [&](int bufoff, const u32x4 (&la)[4], const u32x4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(wA + bufoff + q * 32 * LDR) = la[q];
      *reinterpret_cast<u32x4*>(wB + bufoff + q * 32 * LDR) = lb[q];
    }
  };
  auto mfma_slab = [&](int bufoff) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const float4*>(rA + bufoff + mi * 32 * LDR + s4 * 4);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const float4*>(rB + bufoff + ni * 32 * LDR + s4 * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const float av = q == 0 ? a[mi].x : q == 1 ? a[mi].y : q == 2 ? a[mi].z : a[mi].w;
            const float bv = q == 0 ? b[ni].x : q == 1 ? b[ni].y : q == 2 ? b[ni].z : b[ni].w;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
          }
    }
  };
  {
    u32x4 la[4], lb[4];
    gload(0, la, lb);
    lstore(0, la, lb);
  }
  __syncthreads();
  auto step = [&](int t, int curoff, int nxtoff) {
    u32x4 la[4], lb[4];
    gload(t + 1 < nt ? t + 1 : 0, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    mfma_slab(curoff);
    __builtin_amdgcn_sched_barrier(0);
    lstore(nxtoff, la, lb);
    __syncthreads();
  };
  if (UNROLL2) {
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      step(t, 0, TSZ);
      step(t + 1, TSZ, 0);
    }
    if (t < nt) step(t, 0, TSZ);
  } else {
    for (int t = 0; t < nt; ++t) {
      const int cur = (t & 1) * TSZ;
      step(t, cur, TSZ - cur);
    }
  }
  const unsigned coff = (unsigned)((4 * h) * N + li) * 4u;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col0 = n0 + (wn * 2 + ni) * 32;
      const int row0 = m0 + (wm * 2 + mi) * 32;
      char* cb = reinterpret_cast<char*>(C + (long long)row0 * N + col0);
      const bool full = row0 + 32 <= M && col0 + 32 <= N;
      if (full) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          *reinterpret_cast<float*>(cb + (long long)((e & 3) + 8 * (e >> 2)) * N * 4 + coff) = acc[mi][ni][e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < M && col0 + li < N) C[(long long)row * N + col0 + li] = acc[mi][ni][e];
        }
      }
    }
}

template <int U>
float run_leanb(const float* A, const float* W, float* C, int M, int N, int K, int iters) {
  const size_t smem = (size_t)4 * BM * LDR * 4;
  auto kern = leanb<U>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), smem, 0, A, W, C, M, N, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  return ms / iters;
}

template <int VAR>
float run(const float* A, const float* W, float* C, int M, int N, int K, int iters) {
  constexpr bool GL = VAR >= 4;
  constexpr int NBUF = VAR == 5 ? 3 : 2;
  const size_t smem = (size_t)NBUF * ((GL ? BM * BK : BM * LDR) + (GL ? BN * BK : BN * LDR)) * 4;
  auto kern = lab<VAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), smem, 0, A, W, C, M, N, K);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  return ms / iters;
}

static f2g_operand plain(const float* p, int rows, int cols) {
  f2g_operand o; memset(&o, 0, sizeof(o));
  o.base = p; o.rows = rows; o.cols = cols; o.P0 = 1; o.P1 = 1; o.seglen = cols; o.L1 = 1;
  o.L0u = cols; o.unit = 1; o.step0 = 1; o.step1 = 1; o.seq_stride = cols; o.line_stride = cols;
  return o;
}
typedef int (*gemm_fn)(const f2g_gemm_desc*, f2g_stream_t);
static const float *gA, *gW; static float* gC; static int gM, gN, gK;
extern "C" int lean_pre(const f2g_gemm_desc* d, f2g_stream_t) { return 0; }
extern "C" int f2g_gemm_copy(const f2g_gemm_desc*, f2g_stream_t);
float run_lib(gemm_fn fn, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int iters) {
  f2g_gemm_desc d; memset(&d, 0, sizeof(d));
  d.A = plain(A, M, K); d.B = plain(W, N, K);
  d.E.C = C; d.E.ldc = N; d.E.bias = bias; d.form = 0; d.split_k = 1;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) if (fn(&d, 0) != 0) { printf("f2g_gemm failed\n"); exit(1); }
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  return ms / iters;
}

int main(int argc, char** argv) {
  std::vector<gemm_fn> libs; std::vector<std::string> libnames;
  libs.push_back((gemm_fn)1); libnames.push_back("LEAN loads after MFMA");
  libs.push_back((gemm_fn)2); libnames.push_back("LEAN loads before MFMA");
  libs.push_back((gemm_fn)3); libnames.push_back("LEANB buffer loads");
  libs.push_back((gemm_fn)4); libnames.push_back("LEANB buffer loads unroll2");
  libs.push_back(f2g_gemm_copy); libnames.push_back("in-executable copy (var7)");
  for (int i = 1; i < argc; ++i) {
    void* hnd = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
    if (!hnd) { printf("dlopen %s: %s\n", argv[i], dlerror()); return 1; }
    libs.push_back((gemm_fn)dlsym(hnd, "f2g_gemm")); libnames.push_back(argv[i]);
  }
  const int shapes[][3] = {{38016, 1024, 5120}, {6016, 768, 2304}, {24064, 384, 1152}, {24064, 1152, 384},
                           {12032, 512, 1536}, {4096, 4096, 4096}};
  const char* names[] = {"BASE", "NOLOAD", "BASE+16V", "MFMAONLY", "GLDS", "GLDS3"};
  const int nshapes = getenv("LAB_SHAPES") ? atoi(getenv("LAB_SHAPES")) : 6;
  const unsigned legmask = getenv("LAB_LEGS") ? (unsigned)strtoul(getenv("LAB_LEGS"), 0, 0) : 0xffffffffu;
  int shape_i = 0;
  for (auto& sh : shapes) {
    if (shape_i++ >= nshapes) break;
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    float *A, *W, *C, *C0;
    hipMalloc(&A, ha.size() * 4); hipMalloc(&W, hw.size() * 4);
    hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C0, (size_t)M * N * 4);
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    // correctness of the candidates against BASE
    run<0>(A, W, C0, M, N, K, 1);
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost);
    for (int v = 4; v <= 5; ++v) {
      if (!((legmask >> v) & 1)) continue;
      hipMemset(C, 0, (size_t)M * N * 4);
      if (v == 4) run<4>(A, W, C, M, N, K, 1); else run<5>(A, W, C, M, N, K, 1);
      hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost);
      double md = 0;
      for (size_t i = 0; i < h0.size(); ++i) md = std::max(md, (double)fabsf(h0[i] - h1[i]));
      printf("  check %s vs BASE max|d| = %.3g\n", names[v], md);
    }
    const double flop = 2.0 * M * N * (double)K;
    const int iters = std::max(3, (int)(4e12 / flop));
    const int rounds = 6;
    std::vector<double> libsum(libs.size(), 0.0);
    double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9}, sum[6] = {0};
    const int nleg = 6 + (int)libs.size();
    std::vector<int> order(nleg);
    for (int i = 0; i < nleg; ++i) order[i] = i;
    for (int r = 0; r < rounds; ++r) {
      // a different order every round (rotation + reversal): DVFS / thermal state carries over
      // from leg to leg, so a fixed order would bias the later legs
      std::rotate(order.begin(), order.begin() + 1, order.end());
      if (r & 1) std::reverse(order.begin(), order.end());
      for (int leg : order) {
        float t = 0;
        if (!((legmask >> leg) & 1)) continue;
        switch (leg) {
          case 0: t = run<0>(A, W, C, M, N, K, iters); break;
          case 1: t = run<1>(A, W, C, M, N, K, iters); break;
          case 2: t = run<2>(A, W, C, M, N, K, iters); break;
          case 3: t = run<3>(A, W, C, M, N, K, iters); break;
          case 4: t = run<4>(A, W, C, M, N, K, iters); break;
          case 5: t = run<5>(A, W, C, M, N, K, iters); break;
          default:
            if (libs[leg - 6] == (gemm_fn)1) t = run_lean<0>(A, W, C, M, N, K, iters);
            else if (libs[leg - 6] == (gemm_fn)2) t = run_lean<1>(A, W, C, M, N, K, iters);
            else if (libs[leg - 6] == (gemm_fn)3) t = run_leanb<0>(A, W, C, M, N, K, iters);
            else if (libs[leg - 6] == (gemm_fn)4) t = run_leanb<1>(A, W, C, M, N, K, iters);
            else t = run_lib(libs[leg - 6], A, W, C, nullptr, M, N, K, iters);
            break;
        }
        if (leg < 6) { best[leg] = std::min(best[leg], (double)t); sum[leg] += t; }
        else libsum[leg - 6] += t;
      }
      if (r & 1) std::reverse(order.begin(), order.end());
    }
    printf("M=%d N=%d K=%d (%d iters x %d rounds)\n", M, N, K, iters, rounds);
    for (int v = 0; v < 6; ++v)
      printf("  %-9s mean %8.3f ms  %6.1f TFLOP/s   best %6.1f TFLOP/s\n", names[v], sum[v] / rounds,
             flop / (sum[v] / rounds * 1e-3) / 1e12, flop / (best[v] * 1e-3) / 1e12);
    for (size_t l = 0; l < libs.size(); ++l) {
      hipMemset(C, 0, (size_t)M * N * 4);
      if (libs[l] == (gemm_fn)1) run_lean<0>(A, W, C, M, N, K, 1);
      else if (libs[l] == (gemm_fn)2) run_lean<1>(A, W, C, M, N, K, 1);
      else if (libs[l] == (gemm_fn)3) run_leanb<0>(A, W, C, M, N, K, 1);
      else if (libs[l] == (gemm_fn)4) run_leanb<1>(A, W, C, M, N, K, 1);
      else run_lib(libs[l], A, W, C, nullptr, M, N, K, 1);
      hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost);
      double md = 0;
      for (size_t i = 0; i < h0.size(); ++i) md = std::max(md, (double)fabsf(h0[i] - h1[i]));
      printf("  %-28s mean %8.3f ms  %6.1f TFLOP/s  (max|d| vs BASE %.2g)\n", libnames[l].c_str(), libsum[l] / rounds,
             flop / (libsum[l] / rounds * 1e-3) / 1e12, md);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(C0);
  }
  return 0;
}
