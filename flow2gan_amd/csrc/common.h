// Shared helpers for libflow2gan_hip.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/flow2gan_hip.h"

int f2g_check_launch();
// narrow.hip: VALU path for <= 4 output columns / gradient rows; 1 = handled, 0 = not applicable
int f2g_gemm_narrow(const f2g_gemm_desc& d, hipStream_t st);
void f2g_set_error(const char* msg);
// gemm_x6p.hip: ping-pong tap-walking fp32-class GEMM (precision 3 over halo-map images); ok = 1 if taken
int f2g_x6p_ok(const f2g_gemm_desc& d, int taps);
int f2g_launch_x6p(const f2g_gemm_desc& d, int taps, long long a_extent, hipStream_t st);
// tap-walking fp32-class weight gradient of a stride-1 five-tap conv over halo maps (gemm_x6p.hip)
int f2g_leanw6t_ok(const f2g_gemm_desc& d, int split);
int f2g_launch_leanw6t(const f2g_gemm_desc& d, int split, hipStream_t st);
// the same schedule over row operands (plain matrices / strided single-segment windows): split = 0 images, 1 = fp32
int f2g_x6pr_ok(const f2g_gemm_desc& d);
int f2g_launch_x6pr(const f2g_gemm_desc& d, int split, int P0, unsigned seq, unsigned step, unsigned off,
                    unsigned bytes, hipStream_t st);
// elementwise.hip: out[k][c - begin[k]] += sum_r a[r, c] for up to 3 column ranges (null = skip)
struct f2g_colsegs {
  float* out[3];
  int begin[3];
  int count[3];
};
int f2g_colsum_segments(const float* a, long long lda, int rows, int ncols,
                        const f2g_colsegs& segs, hipStream_t st);

#define F2G_WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// n / d for 0 <= n < 2^31 with a per-kernel magic number (gfx950 has no integer divide: the
// compiler's expansion costs ~35 VALU instructions, and the windowed loaders divide once per
// K slab / per chunk): magic = ceil(2^32 / d) over-estimates the quotient by at most one.
__device__ __forceinline__ unsigned magic_of(int d) {
  return d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1u) / (unsigned)d) : 0u;
}
__device__ __forceinline__ int fast_div(int n, int d, unsigned mg) {
  if (d == 1) return n;
  int q = (int)__umulhi((unsigned)n, mg);
  q -= (q * d > n) ? 1 : 0;
  return q;
}

static inline int f2g_grid_for(int64_t n, int block, int cap = 2048 * 4) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
