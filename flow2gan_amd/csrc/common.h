// Shared helpers for libflow2gan_hip.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/flow2gan_hip.h"

int f2g_check_launch();
// Library options (capi.hip): every tunable the dispatch looks at is one int of a table that is initialised
// ONCE -- defaults, then `F2G_OPTS="name=value,..."` from the environment -- and changed afterwards only through
// f2g_set_option (tests, lab tools).  No getenv on a launch path.
enum f2g_opt_id {
  F2G_OPT_LEAN,             // 1: the lean GEMM kernels (0: generic kernels everywhere they can stand in)
  F2G_OPT_LEAN_TALL,        // bf16 lean instances on 256 x 128 tiles: 0 never, 1 K >= 640 on >= 400 tall tiles, 2 always
  F2G_OPT_LEAN_TAP,         // 1: tap-reusing split-bf16 instance for stride-1 conv windows
  F2G_OPT_LEAN_WGRAD,       // exact-fp32 K-major weight gradient: 0 off, 1 reductions >= 4096 rows per block, 2 always
  F2G_OPT_X6_TAP,           // 1: tap-walking precision-3 kernel for stride-1 conv windows
  F2G_OPT_X6_WIDE,          // 1: wide (LDS-turned) epilogue of the precision-3 kernels
  F2G_OPT_X6P,              // gemm_x6p_kernel: 0 off, 1 chip-filling grids, 2 whatever the grid (tests)
  F2G_OPT_W6T,              // 1: tap-walking precision-3 weight gradients
  F2G_OPT_DETERMINISTIC,    // 1: no split / stream-K on the library's own initiative (bit-reproducible forward)
  F2G_OPT_STREAMK,          // lean kernel stream-K: 0 off, 1 latency regime, 2 every under-filled grid
  F2G_OPT_CONV2CH_V2,       // 1: persistent first-MRD-layer kernels
  F2G_OPT_CONV32_V2,        // 1: persistent exact-fp32 band-conv forward / data gradient
  F2G_OPT_CONV32_WGRAD_V2,  // 1: persistent exact-fp32 band-conv weight gradient
  F2G_OPT_MLP_RT,           // fused block kernel: rows / 32 per tile (0: the launch decides)
  F2G_OPT_MLP_SPLIT,        // fused MLP: parts of the hidden dimension (0 / 1: never split)
  F2G_OPT_MULTI_RT384,      // multi-branch launch: rows / 32 per tile of the 384-channel entries
  F2G_OPT_MULTI_RT512,      // ... of the 512-channel entries
  F2G_OPT_STREAMK_MIN,      // lean stream-K: fewest K slabs a block takes (prologue / epilogue amortisation)
  F2G_OPT_COUNT
};
int f2g_opt(int id);
// narrow.hip: VALU path for <= 4 output columns / gradient rows; 1 = handled, 0 = not applicable
int f2g_gemm_narrow(const f2g_gemm_desc& d, hipStream_t st);
void f2g_set_error(const char* msg);
// gemm_x6p.hip: ping-pong tap-walking fp32-class GEMM (precision 3 over halo-map images); ok = 1 if taken
int f2g_x6p_ok(const f2g_gemm_desc& d, int taps);
int f2g_launch_x6p(const f2g_gemm_desc& d, int taps, long long a_extent, hipStream_t st);
// tap-walking fp32-class weight gradient of a stride-1 five-tap conv over halo maps (gemm_x6p.hip)
int f2g_leanw6t_ok(const f2g_gemm_desc& d, int split);
int f2g_launch_leanw6t(const f2g_gemm_desc& d, int split, hipStream_t st);
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

// Three bf16 pieces of TWO floats, round to nearest even at every step (as f2g_split_bf16x3): x = p0 + p1 + p2 to
// ~2^-25 relative.  On pairs one v_cvt_pk_bf16_f32 yields the packed piece (low half = x0), two bit operations
// widen it again and one v_pk_add_f32 takes the remainder: 9 VALU instructions per pair (the element-wise
// formulation the in-kernel splits used until round 6 compiled to 14.5).
typedef float f2g_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 f2g_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f2g_split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  const f2g_f32x2 x = {x0, x1};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f2g_bf16x2));
  const f2g_f32x2 r1 = x - f2g_f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f2g_bf16x2));
  const f2g_f32x2 r2 = r1 - f2g_f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, f2g_bf16x2));
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
