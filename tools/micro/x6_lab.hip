// Laboratory (not part of the library): fp32-class GEMM on the bf16 matrix pipe.
//   C[M,N] = A[M,K] * W[N,K]^T with every fp32 operand split into THREE bf16 pieces
//   x = p0 + p1 + p2 (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1): 24 mantissa bits) and the
//   six products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the dropped terms are
//   <= 2^-24 relative, i.e. the error class of fp32 rounding itself (the library's two-piece split-bf16
//   mode stops at 2^-16).  Question asked here: what rate does a plain 128 x 128 x 32 tiling reach, and
//   what error against float64, next to the three-product variant on the same data path?
// Operand planes are separate row-major bf16 matrices [piece][rows][K]; a block stages 3 x (128 + 128)
// rows x 64 bytes per slab (LDS rows 80 bytes apart: conflict-free ds_read_b128), double-buffered.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/x6_lab.hip -o tools/micro/x6_lab
//   -DLAB_IMG: operands as a slab-interleaved image (the three pieces of a row's 32-element slab side by
//   side, 192 contiguous bytes: whole cache lines per row instead of three half lines); configurations
//   8 / 9 / 12 / 13 only.  LAB_ONLY=<configuration>, LAB_SHAPES=<n>, LAB_ABL=<bits> (timing ablations of
//   configurations 8 / 9: 2 no global loads, 4 no LDS stores, 8 no barrier, 16 no fragment reads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;

__global__ void split3_kernel(const float* __restrict__ x, __bf16* __restrict__ p, long long n, long long K) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const __bf16 a = (__bf16)v;
    const float r1 = v - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
#ifdef LAB_IMG
    // slab-interleaved image: the three pieces of a row's 32-element slab lie side by side (192 bytes)
    const long long r = i / K, k = i - r * K;
    const long long o = (r * (K / 32) + k / 32) * 96 + (k & 31);
    p[o] = a;
    p[o + 32] = b;
    p[o + 64] = c;
#else
    p[i] = a;
    p[n + i] = b;
    p[2 * n + i] = c;
#endif
  }
}

// NP: pieces used (2 -> three products a0b0 + a0b1 + a1b0, 3 -> six products with i + j <= 2)
// BK: k per slab: 32 (LDS rows 80 bytes apart, 120 KB: one block per CU) or 16 (48 bytes, 74 KB: two)
// PIPE: two register stages -- the loads of slab t + 2 fly while slab t is multiplied and slab t + 1
// (loaded during the previous iteration) is written to LDS between the MFMAs
//       2: the library's split-bf16 schedule on top -- fragments of the NEXT k-step are read under the
//       MFMAs of the current one (two fragment sets), the barrier sits between the two k-steps of a slab
template <int NP, int BK, int PIPE = 0>
__global__ __launch_bounds__(256, BK == 16 ? 2 : 1) void x6_kernel(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Wp,
                                                    float* __restrict__ C, int M, int N, int K, int bare) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: a staged row holds its three pieces side by side, rows 3 * 2 BK + 16 + (BK == 32 ? 64 : 32)
  // bytes apart = 4 dwords mod 64: ds_read_b128 serves lanes {0-3, 12-15, 20-27} etc. together, and only
  // this residue keeps such a group on 64 distinct banks (80-byte rows: 2-way conflicts, 134 -> ... )
  constexpr int PSTEP = BK * 2;                              // bytes of one piece of a row
  constexpr int PITCH = BK == 32 ? 272 : 144;                // (3 * 64 + 80 / 3 * 32 + 48)
  constexpr int OPER = 128 * PITCH;                          // one operand of a slab
  constexpr int STAGE = 2 * OPER;
  constexpr int CPR = BK / 8;                                // 16-byte chunks per row and plane
  constexpr int NQ = 128 * CPR / 256;                        // chunks per thread and plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  // XCD-contiguous tile order, N fastest
  const int tiles_n = gridDim.y, tiles_m = gridDim.x, nblk = tiles_m * tiles_n;
  int bid = blockIdx.y * tiles_m + blockIdx.x;
  {
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const long long planeA = (long long)M * K, planeW = (long long)N * K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // loader: a plane of a slab = 128 rows x CPR chunks of 16 bytes = NQ per thread
  constexpr int RSTEP = 256 / CPR;                           // rows covered by one pass of the block
  // (staged row of a thread: quads of lanes read one 64-byte line each; the two quads of an 8-lane
  // LDS store group take rows 8 apart -- one apart, their 16-byte stores share 12 of 16 banks)
  const int ch = tid % CPR, q4 = tid / CPR;
  const int r0 = CPR == 4 ? ((q4 & ~15) | ((q4 & 1) << 3) | ((q4 >> 1) & 7)) : q4;
  const __bf16* ga[NQ];
  const __bf16* gw[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int ra = m0 + r0 + RSTEP * q, rw = n0 + r0 + RSTEP * q;
    ra = ra < M ? ra : M - 1;
    rw = rw < N ? rw : N - 1;
    ga[q] = Ap + (long long)ra * K + ch * 8;
    gw[q] = Wp + (long long)rw * K + ch * 8;
  }
  u32x4 la[NP][NQ], lw[NP][NQ], ya[NP][NQ], yw[NP][NQ], za3[NP][NQ], zw3[NP][NQ];
#ifdef LAB_IMG
  // slab-interleaved image (BK = 32): a row's slab = 4 NP chunks of 16 contiguous bytes (the first NP
  // pieces of its 192); chunk id = tid + 256 j -> (row, chunk); LDS rows hold the same bytes in order
  constexpr int CPRI = 4 * NP;
  // buffer loads: the resource in SGPRs, one per-thread CONSTANT byte offset per chunk (decoded once),
  // the slab advance in a scalar register -- no vector address arithmetic in the loop
  const unsigned rowbytes = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (unsigned)M * rowbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (unsigned)N * rowbytes, 0x00020000);
  unsigned voA[NP * NQ], voW[NP * NQ];
#pragma unroll
  for (int j = 0; j < NP * NQ; ++j) {
    const int id = tid + 256 * j, row = id / CPRI, c = id - row * CPRI;
    voA[j] = (unsigned)(m0 + row) * rowbytes + c * 16;      // rows past the end: out of range = zeros
    voW[j] = (unsigned)(n0 + row) * rowbytes + c * 16;
  }
  auto gload2 = [&](int k0, u32x4 (&xa)[NP][NQ], u32x4 (&xw)[NP][NQ]) {
    const int so = (k0 / 32) * 192;
#pragma unroll
    for (int j = 0; j < NP * NQ; ++j) {
      xa[j / NQ][j % NQ] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], so, 0);
      xw[j / NQ][j % NQ] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], so, 0);
    }
  };
  auto lstore2 = [&](int buf, const u32x4 (&xa)[NP][NQ], const u32x4 (&xw)[NP][NQ]) {
#pragma unroll
    for (int j = 0; j < NP * NQ; ++j) {
      const int id = tid + 256 * j, row = id / CPRI, c = id - row * CPRI;
      unsigned char* base = smem + buf * STAGE + row * PITCH + c * 16;
      *reinterpret_cast<u32x4*>(base) = xa[j / NQ][j % NQ];
      *reinterpret_cast<u32x4*>(base + OPER) = xw[j / NQ][j % NQ];
    }
  };
#else
  auto gload2 = [&](int k0, u32x4 (&xa)[NP][NQ], u32x4 (&xw)[NP][NQ]) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        xa[p][q] = *reinterpret_cast<const u32x4*>(ga[q] + p * planeA + k0);
        xw[p][q] = *reinterpret_cast<const u32x4*>(gw[q] + p * planeW + k0);
      }
  };
  auto lstore2 = [&](int buf, const u32x4 (&xa)[NP][NQ], const u32x4 (&xw)[NP][NQ]) {
    unsigned char* base = smem + buf * STAGE + (r0 * PITCH + ch * 16);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        *reinterpret_cast<u32x4*>(base + p * PSTEP + q * RSTEP * PITCH) = xa[p][q];
        *reinterpret_cast<u32x4*>(base + OPER + p * PSTEP + q * RSTEP * PITCH) = xw[p][q];
      }
  };
#endif
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        la[p][q] = *reinterpret_cast<const u32x4*>(ga[q] + p * planeA + k0);
        lw[p][q] = *reinterpret_cast<const u32x4*>(gw[q] + p * planeW + k0);
      }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * STAGE + (r0 * PITCH + ch * 16);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        *reinterpret_cast<u32x4*>(base + p * PSTEP + q * RSTEP * PITCH) = la[p][q];
        *reinterpret_cast<u32x4*>(base + OPER + p * PSTEP + q * RSTEP * PITCH) = lw[p][q];
      }
  };
  const unsigned char* rA = smem + (wm * 64 + li) * PITCH + h * 16;
  const unsigned char* rB = smem + OPER + (wn * 64 + li) * PITCH + h * 16;
  const int nt = K / BK;
  auto frag_mfma = [&](int cur, int ks) {
    bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[p][i] = *reinterpret_cast<const bf16x8*>(rA + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
        fb[p][i] = *reinterpret_cast<const bf16x8*>(rB + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
      }
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int j = s - i;
        if (j < 0 || j >= NP) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[j][ni], acc[mi][ni], 0, 0, 0);
      }
  };
  if constexpr (PIPE == 2 || PIPE == 4) {
    bf16x8 f0a[NP][2], f0b[NP][2], f1a[NP][2], f1b[NP][2];
    auto frags = [&](int cur, int ks, bf16x8 (&fa)[NP][2], bf16x8 (&fb)[NP][2]) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[p][i] = *reinterpret_cast<const bf16x8*>(rA + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
          fb[p][i] = *reinterpret_cast<const bf16x8*>(rB + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
        }
    };
    auto mfmas = [&](const bf16x8 (&fa)[NP][2], const bf16x8 (&fb)[NP][2]) {
#pragma unroll
      for (int s = NP - 1; s >= 0; --s)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int j = s - i;
          if (j < 0 || j >= NP) continue;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[j][ni], acc[mi][ni], 0, 0, 0);
        }
    };
    gload2(0, la, lw);
    lstore2(0, la, lw);
    gload2(nt > 1 ? BK : 0, la, lw);
    __syncthreads();
    frags(0, 0, f0a, f0b);
    if (bare == 1) {  // (lab: the bare MFMA stream of this schedule -- no LDS, no loads in the loop)
      frags(0, 1, f1a, f1b);
      for (int t = 0; t < nt; ++t) {
        mfmas(f0a, f0b);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1a, f1b);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    auto step = [&](int t, u32x4 (&xa)[NP][NQ], u32x4 (&xw)[NP][NQ], u32x4 (&za)[NP][NQ], u32x4 (&zw)[NP][NQ]) {
      const int cur = (t & 1) * STAGE, nxt = ((t + 1) & 1) * STAGE;
      // (ablation bits of `bare`, timing only: 2 no global loads, 4 no LDS stores, 8 no barrier,
      // 16 no fragment reads)
      if (!(bare & 16)) frags(cur, 1, f1a, f1b);
      if (!(bare & 2)) gload2(t + 2 < nt ? (t + 2) * BK : 0, za, zw);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(f0a, f0b);
      if (!(bare & 4)) lstore2((t + 1) & 1, xa, xw);
      // issue order: one LDS store behind each of the first MFMAs
#pragma unroll
      for (int i = 0; i < NP * NP * 4 - (NP == 3 ? 12 : 4); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 2 * NP * NQ) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(bare & 8)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (!(bare & 16)) frags(nxt, 0, f0a, f0b);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(f1a, f1b);
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (PIPE == 4) {
      // three register stages: the loads of slab t + 3 are requested in pass t (two passes of cover)
      auto step3 = [&](int t, u32x4 (&sa)[NP][NQ], u32x4 (&sw)[NP][NQ], u32x4 (&na)[NP][NQ], u32x4 (&nw)[NP][NQ]) {
        const int cur = (t & 1) * STAGE, nxt = ((t + 1) & 1) * STAGE;
        gload2(t + 3 < nt ? (t + 3) * BK : 0, na, nw);      // into the set slab t occupied
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 1, f1a, f1b);
        mfmas(f0a, f0b);
        lstore2((t + 1) & 1, sa, sw);
        // issue order: ONE MFMA first (its fragments are a pass old: the wait in front of it must not
        // cover the reads below), then the next k-step's fragment reads, then an LDS store behind
        // each of the following MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);
#pragma unroll
        for (int i = 1; i < NP * NP * 4 - (NP == 3 ? 12 : 4); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i <= 2 * NP * NQ) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        frags(nxt, 0, f0a, f0b);
        mfmas(f1a, f1b);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NP * NP * 4, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      // sets: slab s lives in set s % 3 = (la, ya, za3); slab 1 is in la after the prologue above ->
      // rename: la = slab 1, then load slab 2 into ya
      gload2(nt > 2 ? 2 * BK : 0, ya, yw);
      int t = 0;
      for (; t + 2 < nt; t += 3) {
        step3(t, la, lw, za3, zw3);          // stores slab t+1 (la), loads t+3 -> za3
        step3(t + 1, ya, yw, la, lw);        // stores slab t+2 (ya), loads t+4 -> la
        step3(t + 2, za3, zw3, ya, yw);      // stores slab t+3 (za3), loads t+5 -> ya
      }
      if (t < nt) { step3(t, la, lw, za3, zw3); ++t; }
      if (t < nt) { step3(t, ya, yw, la, lw); ++t; }
    } else {
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      step(t, la, lw, ya, yw);
      step(t + 1, ya, yw, la, lw);
    }
    if (t < nt) step(t, la, lw, ya, yw);
    }
    }
  } else if constexpr (PIPE == 1) {
    gload2(0, la, lw);
    lstore2(0, la, lw);
    gload2(nt > 1 ? BK : 0, la, lw);
    __syncthreads();
    auto step = [&](int t, u32x4 (&xa)[NP][NQ], u32x4 (&xw)[NP][NQ], u32x4 (&za)[NP][NQ], u32x4 (&zw)[NP][NQ]) {
      const int cur = (t & 1) * STAGE;
      gload2(t + 2 < nt ? (t + 2) * BK : 0, za, zw);       // (past the end: re-read, never used)
      __builtin_amdgcn_sched_barrier(0);
      frag_mfma(cur, 0);
      lstore2((t + 1) & 1, xa, xw);                         // slab t + 1, in registers since last pass
      if (BK == 32) frag_mfma(cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      step(t, la, lw, ya, yw);
      step(t + 1, ya, yw, la, lw);
    }
    if (t < nt) step(t, la, lw, ya, yw);
  } else {
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = (t & 1) * STAGE;
    if (t + 1 < nt) gload((t + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[p][i] = *reinterpret_cast<const bf16x8*>(rA + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
          fb[p][i] = *reinterpret_cast<const bf16x8*>(rB + cur + p * PSTEP + i * 32 * PITCH + ks * 32);
        }
      // smallest terms first
#pragma unroll
      for (int s = NP - 1; s >= 0; --s)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int j = s - i;
          if (j < 0 || j >= NP) continue;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[j][ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (t + 1 < nt) lstore((t + 1) & 1);
    __syncthreads();
  }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + (wn * 2 + ni) * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + (wm * 2 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < M && col < N) C[(long long)row * N + col] = acc[mi][ni][e];
      }
    }
}

#ifdef LAB_IMG
// x6c: the structure that leans on TWO blocks per CU instead of intra-wave pipelining -- single LDS
// buffer (rows 208 bytes apart = 52 dwords: conflict-free like every odd multiple of 4), the whole slab's
// fragments in registers, next slab's operands requested one pass ahead by buffer loads:
//   store slab t -> barrier -> read its 24 fragments -> barrier -> 48 MFMAs   (53 KB of LDS, <= 256 registers)
template <int NP>
__global__ __launch_bounds__(256, 2) void x6c_kernel(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Wp,
                                                     float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PITCH = 208, OPER = 128 * PITCH, CPRI = 4 * NP, NJ = 2 * NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  const int tiles_n = gridDim.y, tiles_m = gridDim.x, nblk = tiles_m * tiles_n;
  int bid = blockIdx.y * tiles_m + blockIdx.x;
  {
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned rowbytes = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (unsigned)M * rowbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (unsigned)N * rowbytes, 0x00020000);
  unsigned voA[NJ], voW[NJ];
  int lo[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = tid + 256 * j, row = id / CPRI, c = id - row * CPRI;
    voA[j] = (unsigned)(m0 + row) * rowbytes + c * 16;
    voW[j] = (unsigned)(n0 + row) * rowbytes + c * 16;
    lo[j] = row * PITCH + c * 16;
  }
  u32x4 xa[NJ], xw[NJ];
  auto gload = [&](int t) {
    const int so = t * 192;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], so, 0);
      xw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], so, 0);
    }
  };
  const unsigned char* rA = smem + (wm * 64 + li) * PITCH + h * 16;
  const unsigned char* rB = smem + OPER + (wn * 64 + li) * PITCH + h * 16;
  const int nt = K / 32;
  gload(0);
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      *reinterpret_cast<u32x4*>(smem + lo[j]) = xa[j];
      *reinterpret_cast<u32x4*>(smem + OPER + lo[j]) = xw[j];
    }
    gload(t + 1 < nt ? t + 1 : 0);
    __syncthreads();
    bf16x8 fa[2][NP][2], fb[2][NP][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[ks][p][i] = *reinterpret_cast<const bf16x8*>(rA + p * 64 + i * 32 * PITCH + ks * 32);
          fb[ks][p][i] = *reinterpret_cast<const bf16x8*>(rB + p * 64 + i * 32 * PITCH + ks * 32);
        }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int s = NP - 1; s >= 0; --s)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int j = s - i;
          if (j < 0 || j >= NP) continue;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][mi], fb[ks][j][ni], acc[mi][ni], 0, 0, 0);
        }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + (wn * 2 + ni) * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + (wm * 2 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < M && col < N) C[(long long)row * N + col] = acc[mi][ni][e];
      }
    }
}

template <int NP>
float run_c(const __bf16* Ap, const __bf16* Wp, float* C, int M, int N, int K, int iters) {
  const size_t smem = 2 * 128 * 208;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((x6c_kernel<NP>), grid, dim3(256), smem, 0, Ap, Wp, C, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((x6c_kernel<NP>), grid, dim3(256), smem, 0, Ap, Wp, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}
#endif

template <int NP, int BK, int PIPE = 0>
float run(const __bf16* Ap, const __bf16* Wp, float* C, int M, int N, int K, int iters, int bare = 0) {
  const size_t smem = 2 * 2 * 128 * (BK == 32 ? 272 : 144);
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(x6_kernel<NP, BK, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    done = true;
  }
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((x6_kernel<NP, BK, PIPE>), grid, dim3(256), smem, 0, Ap, Wp, C, M, N, K, bare);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((x6_kernel<NP, BK, PIPE>), grid, dim3(256), smem, 0, Ap, Wp, C, M, N, K, bare);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int shapes[][3] = {{38016, 1024, 5120}, {38016, 1024, 2560}, {19008, 1024, 5120}, {113920, 512, 640},
                           {6016, 768, 2304}, {24064, 1152, 384}};
  int shape_i = 0;
  const int nshapes = getenv("LAB_SHAPES") ? atoi(getenv("LAB_SHAPES")) : 6;
  for (auto& sh : shapes) {
    if (shape_i++ >= nshapes) break;
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * (1.0f + 3.0f * fabsf(rnd()));
    for (auto& v : hw) v = rnd() * 0.05f;
    float *A, *W, *C;
    __bf16 *Ap, *Wp;
    hipMalloc(&A, ha.size() * 4); hipMalloc(&W, hw.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&Ap, ha.size() * 6); hipMalloc(&Wp, hw.size() * 6);
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split3_kernel, dim3(4096), dim3(256), 0, 0, A, Ap, (long long)ha.size(), (long long)K);
    hipLaunchKernelGGL(split3_kernel, dim3(4096), dim3(256), 0, 0, W, Wp, (long long)hw.size(), (long long)K);
    const double flop = 2.0 * M * N * (double)K;
    const int iters = 10;
    std::vector<float> hc((size_t)M * N);
    // float64 reference and the plain fp32 fmaf chain (what the fp32 MFMA computes) on sampled entries
    const int NS = 96;
    const int abl = getenv("LAB_ABL") ? atoi(getenv("LAB_ABL")) : 0;
    const int only = getenv("LAB_ONLY") ? atoi(getenv("LAB_ONLY")) : -1;
    for (int cfg = 0; cfg < 16; ++cfg) {
      if (only >= 0 && cfg != only) continue;
      const int np = 2 + (cfg & 1), bk = (cfg & 2) && cfg < 8 ? 16 : 32;
      const int pipe = cfg >= 14 ? 5 : cfg >= 12 ? 4 : (cfg >= 10 ? 3 : (cfg >= 8 ? 2 : (cfg >= 4 ? 1 : 0)));
      float ms = 0;
      switch (cfg) {
        case 0: ms = run<2, 32>(Ap, Wp, C, M, N, K, iters); break;
        case 1: ms = run<3, 32>(Ap, Wp, C, M, N, K, iters); break;
        case 2: ms = run<2, 16>(Ap, Wp, C, M, N, K, iters); break;
        case 3: ms = run<3, 16>(Ap, Wp, C, M, N, K, iters); break;
        case 4: ms = run<2, 32, 1>(Ap, Wp, C, M, N, K, iters); break;
        case 5: ms = run<3, 32, 1>(Ap, Wp, C, M, N, K, iters); break;
        case 6: ms = run<2, 16, 1>(Ap, Wp, C, M, N, K, iters); break;
        case 7: ms = run<3, 16, 1>(Ap, Wp, C, M, N, K, iters); break;
        case 8: ms = run<2, 32, 2>(Ap, Wp, C, M, N, K, iters, abl); break;
        case 9: ms = run<3, 32, 2>(Ap, Wp, C, M, N, K, iters, abl); break;
        case 10: ms = run<2, 32, 2>(Ap, Wp, C, M, N, K, iters, 1); break;
        case 11: ms = run<3, 32, 2>(Ap, Wp, C, M, N, K, iters, 1); break;
        case 12: ms = run<2, 32, 4>(Ap, Wp, C, M, N, K, iters); break;
        case 13: ms = run<3, 32, 4>(Ap, Wp, C, M, N, K, iters); break;
#ifdef LAB_IMG
        case 14: ms = run_c<2>(Ap, Wp, C, M, N, K, iters); break;
        default: ms = run_c<3>(Ap, Wp, C, M, N, K, iters); break;
#else
        default: continue;
#endif
      }
      hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
      double worst = 0, worst32 = 0, scale = 0;
      unsigned q = 777;
      for (int i = 0; i < NS; ++i) {
        q = q * 1664525u + 1013904223u; const int m = (q >> 8) % M;
        q = q * 1664525u + 1013904223u; const int n = (q >> 8) % N;
        double ref = 0, mag = 0; float f32 = 0.f;
        for (int k = 0; k < K; ++k) {
          const double a = ha[(size_t)m * K + k], w = hw[(size_t)n * K + k];
          ref += a * w; mag += fabs(a * w);
          f32 = fmaf(ha[(size_t)m * K + k], hw[(size_t)n * K + k], f32);
        }
        worst = fmax(worst, fabs(hc[(size_t)m * N + n] - ref) / mag);
        worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
        scale = fmax(scale, mag);
      }
      printf("M=%6d N=%5d K=%5d  slab %2d %s %s: %8.1f us = %6.1f TFLOP/s fp32-equivalent (%d bf16 MFMAs per product: %.2f of the bf16 peak) | "
             "max |err| / sum|a w| over %d entries: %.2e  (fp32 fmaf chain: %.2e)\n",
             M, N, K, bk, pipe == 5 ? "two blocks per CU, single LDS buffer" : pipe == 4 ? "3-stage + fragment sets" : pipe == 3 ? "BARE MFMA stream (results garbage)" : pipe == 2 ? "2-stage + fragment sets" : (pipe ? "2-stage" : "simple "), np == 2 ? "3 products (2 pieces)" : "6 products (3 pieces)", ms * 1e3, flop / ms / 1e9,
             np == 2 ? 3 : 6, flop * (np == 2 ? 3 : 6) / ms / 1e9 / 2500.0, NS, worst, worst32);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(Ap); hipFree(Wp);
  }
  return 0;
}
