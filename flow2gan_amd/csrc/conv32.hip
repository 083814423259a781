// Direct LDS-tiled convolution for the 32 -> 32 channel band layers of the multi-resolution STFT
// discriminator (reference discriminators.py:171-181: Conv2d(32, 32, (3, 9), stride (1, 2),
// padding (1, 4)) x 3 per band, 15 bands per step and pass).
//
// As an implicit GEMM these layers have N = 32 output columns: 16 FLOP per byte of im2col operand,
// and the operand (27 overlapping taps) is re-gathered from L2 for every tap -- measured 70-80
// TFLOP/s, L2-bandwidth bound.  Here a block stages the input patch of its 8 x 16 output pixels ONCE
// in LDS (10 rows x 39 columns x 32 channels, split into even / odd columns so that a wave's
// 16 consecutive output columns read a stride-1, bank-conflict-free run for every tap), streams the
// 27 (32 x 32) weight tiles through a double buffer, and feeds the fp32 MFMAs straight from the
// patch at tap-shifted addresses: global traffic per output tile drops from 27 x 16 KB to 50 KB.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 16;          // output tile (rows x columns) = 128 pixels = 4 x 32 (x 2 tap halves = 8 waves)
constexpr int C = 32;                   // channels in = out
constexpr int PITCH = 36;               // floats per staged pixel (conflict-free ds_read_b128)
constexpr int KH = 3, KW = 9;
constexpr int IW = TW + (KW - 1) / 2;   // staged columns per parity: 16 + 4 = 20
constexpr int IH = TH + KH - 1;         // staged rows: 10
constexpr int SUB = IH * IW * PITCH + 16;   // floats of one parity sub-tile (+64 B: the 8-byte stores of
                                         // the split-bf16 staging put pixels x and x+1 = the two parities
                                         // into one 16-lane group; without the skew they share every bank)
constexpr int WB = C * PITCH;           // floats of one weight tile
constexpr int TG = 2;                   // taps per barrier (weights double-buffered per group)

__device__ __attribute__((aligned(16))) float c32_zero[4] = {0.f, 0.f, 0.f, 0.f};

// ---- split-bf16 variants (P3; f2g_conv32_desc.precision = 1) -----------------------------------
// Same tiling; every product is lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 (6 MFMAs of 32
// cycles per tap and wave instead of 16 of 64).  A staged pixel keeps its 144-byte pitch as
// [32 hi | 32 lo | pad]: the patch is split ONCE per block while it is staged (each element then
// feeds 27 taps x 32 outputs), the weight tiles arrive pre-split (f2g_split_bf16 image of the same
// packed matrix, same addressing) and go to LDS as two 8-byte halves.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b, float& ra, float& rb) {
  const __bf16 ha = (__bf16)a, hb = (__bf16)b;
  ra = a - (float)ha;
  rb = b - (float)hb;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}

// write the split image of four floats: hi half at p, lo half 16 floats (64 bytes) further
__device__ __forceinline__ void store_split4(float* p, const float4 v) {
  float r0, r1, r2, r3, z0, z1;
  u32x2 hi, lo;
  hi.x = pack_bf16(v.x, v.y, r0, r1);
  hi.y = pack_bf16(v.z, v.w, r2, r3);
  lo.x = pack_bf16(r0, r1, z0, z1);
  lo.y = pack_bf16(r2, r3, z0, z1);
  *reinterpret_cast<u32x2*>(p) = hi;
  *reinterpret_cast<u32x2*>(p + 16) = lo;
}

__device__ __forceinline__ void store_presplit4(float* p, const float4 v) {
  *reinterpret_cast<u32x2*>(p) = u32x2{__float_as_uint(v.x), __float_as_uint(v.y)};
  *reinterpret_cast<u32x2*>(p + 16) = u32x2{__float_as_uint(v.z), __float_as_uint(v.w)};
}

// one tap: A = pixel row (pitch 144 B, this lane's pixel), B = weight row (this lane's channel)
__device__ __forceinline__ void tap_mfma3(const float* Ab, const float* Bb, int hh, f32x16& acc, f32x16& acc2) {
  bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    ah[ks] = *reinterpret_cast<const bf16x8*>(Ab + (2 * ks + hh) * 4);
    al[ks] = *reinterpret_cast<const bf16x8*>(Ab + (2 * ks + hh) * 4 + 16);
    bh[ks] = *reinterpret_cast<const bf16x8*>(Bb + (2 * ks + hh) * 4);
    bl[ks] = *reinterpret_cast<const bf16x8*>(Bb + (2 * ks + hh) * 4 + 16);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh[0], acc, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], bh[1], acc2, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bl[0], acc, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bl[1], acc2, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh[0], acc, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bh[1], acc2, 0, 0, 0);
}

template <bool P3>
__global__ __launch_bounds__(512, 2) void conv32_s2_fwd_kernel(const f2g_conv32_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* At = sm;                 // [2 parities][IH][IW][PITCH]
  float* Bt = sm + 2 * SUB;       // [2 buffers][TG taps][C][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int tiles_w = (d.Wout + TW - 1) / TW;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int s = blockIdx.y;
  const int h0 = th * TH, w0 = tw * TW;
  const float* xs = d.x + (long long)s * d.x_seq;

  // ---- stage the input patch: rows h0-1 .. h0+8, columns 2*w0-4 .. 2*w0+34 (39), zero outside
  const int x0 = 2 * w0 - (KW - 1) / 2;
  for (int i = tid; i < IH * (2 * IW - 1) * (C / 4); i += 512) {
    const int c4 = i & 7;
    const int px = i >> 3;
    const int r = px / (2 * IW - 1), xr = px - r * (2 * IW - 1);
    const int h = h0 - 1 + r, x = x0 + xr;
    const bool ok = h >= 0 && h < d.H && x >= 0 && x < d.Win;
    const float* p = ok ? xs + (long long)h * d.x_line + (long long)x * C + c4 * 4 : c32_zero;
    const float4 v = *reinterpret_cast<const float4*>(p);
    if constexpr (P3) store_split4(At + (xr & 1) * SUB + (r * IW + (xr >> 1)) * PITCH + c4 * 2, v);
    else *reinterpret_cast<float4*>(At + (xr & 1) * SUB + (r * IW + (xr >> 1)) * PITCH + c4 * 4) = v;
  }
  // weights: thread = (tap of the group, co, 4 input channels); group 0 -> buffer 0
  const int wu = tid >> 8, wco = (tid & 255) >> 3, wc4 = tid & 7;
  const float* wrow = d.w + (long long)wco * (KH * KW * C) + wc4 * 4;
  if constexpr (P3) store_presplit4(Bt + wu * WB + wco * PITCH + wc4 * 2,
                                    *reinterpret_cast<const float4*>(wrow + wu * C));
  else *reinterpret_cast<float4*>(Bt + wu * WB + wco * PITCH + wc4 * 4) =
      *reinterpret_cast<const float4*>(wrow + wu * C);
  __syncthreads();

  f32x16 acc, acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
  // 8 waves: waves 0-3 take the first tap of every group, waves 4-7 the second; pixel group = wave & 3
  const int pg = wave & 3, myu = wave >> 2;
  const int p = pg * 32 + li;             // this lane's output pixel inside the tile
  const int ph = p >> 4, pw = p & 15;
  const int kk0 = hh * 16;                // lane halves split the 32 input channels

  // taps are processed in groups of TG per barrier (weights double-buffered per group); two
  // accumulators alternate so that consecutive MFMAs never wait for each other
  constexpr int NG = (KH * KW + TG - 1) / TG;
  for (int gidx = 0; gidx < NG; ++gidx) {
    const int cur = gidx & 1;
    int tn = (gidx + 1) * TG + wu;
    tn = tn < KH * KW ? tn : KH * KW - 1;
    const float4 wn = *reinterpret_cast<const float4*>(wrow + tn * C);  // next group, in flight
    {
      const int u = myu;
      const int t = gidx * TG + u;
      if (t < KH * KW) {
        const int dh = t / KW, j = t - dh * KW;
        const float* Ab = At + (j & 1) * SUB + ((ph + dh) * IW + pw + (j >> 1)) * PITCH + (P3 ? 0 : kk0);
        const float* Bb = Bt + (cur * TG + u) * WB + li * PITCH + (P3 ? 0 : kk0);
        if constexpr (P3) tap_mfma3(Ab, Bb, hh, acc, acc2);
        else
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const float4 a = *reinterpret_cast<const float4*>(Ab + s4 * 4);
          const float4 b = *reinterpret_cast<const float4*>(Bb + s4 * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
        }
      }
    }
    if constexpr (P3) store_presplit4(Bt + ((cur ^ 1) * TG + wu) * WB + wco * PITCH + wc4 * 2, wn);
    else *reinterpret_cast<float4*>(Bt + ((cur ^ 1) * TG + wu) * WB + wco * PITCH + wc4 * 4) = wn;
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
  // combine the two tap halves: waves 4-7 hand their partial tile over through LDS (the patch is dead)
  float* red = At;  // [4 pixel groups][16][64 lanes]
  if (myu == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(pg * 16 + e) * 64 + lane] = acc[e];
  }
  __syncthreads();
  if (myu == 1) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += red[(pg * 16 + e) * 64 + lane];

  // ---- epilogue: bias + leaky ReLU; accumulator element e sits on pixel row (e&3)+8*(e>>2)+4*hh
  const float bias = d.bias ? d.bias[li] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int q = pg * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
    const int oh = h0 + (q >> 4), ow = w0 + (q & 15);
    if (oh < d.H && ow < d.Wout) {
      float v = acc[e] + bias;
      if (d.lrelu_slope != 0.f) v = v > 0.f ? v : d.lrelu_slope * v;
      d.y[(long long)s * d.y_seq + (long long)oh * d.y_line + (long long)ow * C + li] = v;
    }
  }
}

// ---- persistent forward kernel (round 3; exact fp32) -------------------------------------------
// The kernel above spends 30 % of a block's life outside the MFMAs (70 % of the fp32 rate on the
// largest band, less on bands whose width wastes tile columns): every block stages its patch with
// nothing to overlap it but the CU's second block -- which was launched in the same phase --, half
// of the waves idle through a cross-wave reduction at the end, and a barrier separates every 16
// MFMAs of a wave.  Here
//   * a block is PERSISTENT: it walks over tiles b, b + G, ... and requests the next tile's patch
//     (13 x 16 bytes per thread, kept in registers) before it computes the current one, so the
//     global latency of the patch hides under 432 MFMAs per wave;
//   * 4 waves per block, every wave runs ALL 27 taps of its 32 pixels (no tap halves, no
//     reduction; 32 MFMAs per wave between barriers), 2 blocks per CU;
//   * the weight double buffer keeps cycling across tiles (the last tap group of a tile fetches
//     taps 0, 1 for the next);
//   * two tile shapes, 8 x 16 and 16 x 8 output pixels: the launcher takes the one that wastes fewer
//     columns of the band (widths 13 ... 129: 20 -> 24 instead of 32, 51 -> 56 instead of 64, ...).
// (register arrays of native vectors: arrays of HIP's float4 STRUCT are filled through memcpy
// intrinsics, which kept them in scratch memory)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TH_, int TW_>
__global__ __launch_bounds__(256, 2) void conv32_s2_fwd_p_kernel(const f2g_conv32_desc d, int tiles_w,
                                                                int tiles_h, int ntiles) {
  constexpr int IHv = TH_ + KH - 1, IWv = TW_ + (KW - 1) / 2;     // staged rows; columns per parity
  constexpr int SUBv = IHv * IWv * PITCH + 16;
  constexpr int XW = 2 * IWv - 1;                                  // staged input columns
  constexpr int NCHK = (IHv * XW * (C / 4) + 255) / 256;          // 16-byte chunks per thread
  static_assert(TH_ * TW_ == 128, "a block owns 128 output pixels");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* At = sm;                  // [2 parities][IHv][IWv][PITCH]
  float* Bt = sm + 2 * SUBv;       // [2 buffers][TG taps][C][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int c4 = tid & 7;          // (256 threads: the channel chunk of a thread is the same in every pass)
  // staging plan, tile-independent: LDS offset (floats; -1 = no such chunk), source offset relative
  // to the patch origin, (row, column) for the bounds test
  int so[NCHK], go[NCHK], rx[NCHK];
#pragma unroll
  for (int q = 0; q < NCHK; ++q) {
    const int px = (tid >> 3) + 32 * q;
    const int r = px / XW, xr = px - r * XW;
    const bool v = px < IHv * XW;
    so[q] = v ? (xr & 1) * SUBv + (r * IWv + (xr >> 1)) * PITCH + c4 * 4 : -1;
    go[q] = (int)(r * d.x_line) + xr * C + c4 * 4;
    rx[q] = r | (xr << 8);
  }
  auto tile_pos = [&](int tile, int& sq, int& h0, int& w0) {
    const int tw = tile % tiles_w, rest = tile / tiles_w;
    const int th = rest % tiles_h;
    sq = rest / tiles_h;
    h0 = th * TH_;
    w0 = tw * TW_;
  };
  auto load_patch = [&](int tile, f32x4 (&pf)[NCHK]) {
    int sq, h0, w0;
    tile_pos(tile, sq, h0, w0);
    const int x0 = 2 * w0 - (KW - 1) / 2;
    const float* org = d.x + (long long)sq * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)x0 * C;
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int h = h0 - 1 + (rx[q] & 255), x = x0 + (rx[q] >> 8);
      const bool ok = so[q] >= 0 && h >= 0 && h < d.H && x >= 0 && x < d.Win;
      pf[q] = *reinterpret_cast<const f32x4*>(ok ? org + go[q] : c32_zero);
    }
  };
  auto store_patch = [&](const f32x4 (&pf)[NCHK]) {
#pragma unroll
    for (int q = 0; q < NCHK; ++q)
      if (so[q] >= 0) *reinterpret_cast<f32x4*>(At + so[q]) = pf[q];
  };
  // weights: a tap group = 2 taps x 32 co x 8 chunks = two 16-byte pieces per thread
  const int wco = tid >> 3;
  const float* wrow = d.w + (long long)wco * (KH * KW * C) + c4 * 4;
  auto load_w = [&](int g, f32x4 (&wn)[TG]) {        // group g (g == NG: group 0 of the next tile)
    constexpr int NGc = (KH * KW + TG - 1) / TG;
#pragma unroll
    for (int u = 0; u < TG; ++u) {
      int t = g < NGc ? g * TG + u : u;
      t = t < KH * KW ? t : KH * KW - 1;
      wn[u] = *reinterpret_cast<const f32x4*>(wrow + t * C);
    }
  };
  auto store_w = [&](int buf, const f32x4 (&wn)[TG]) {
#pragma unroll
    for (int u = 0; u < TG; ++u)
      *reinterpret_cast<f32x4*>(Bt + (buf * TG + u) * WB + wco * PITCH + c4 * 4) = wn[u];
  };
  const int p = wave * 32 + li;                  // this lane's output pixel inside the tile
  const int ph = p / TW_, pw = p % TW_;
  const int kk0 = hh * 16;                       // lane halves split the 32 input channels
  const float bias = d.bias ? d.bias[li] : 0.f;
  constexpr int NG = (KH * KW + TG - 1) / TG;
  static_assert(NG % 2 == 0, "the weight double buffer must come back to buffer 0 at a tile's end");

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  // weight pipeline: group g's tiles are requested two groups ahead (register stage g & 1), go to
  // LDS buffer g & 1 at the end of group g - 1 and are consumed in group g -- one group of latency
  // cover was not enough with two waves per SIMD (the barrier of every group waited for L2)
  f32x4 ws[2][TG];
  {
    f32x4 pf[NCHK];
    load_patch(tile, pf);
    load_w(0, ws[0]);
    store_patch(pf);
    store_w(0, ws[0]);
    load_w(1, ws[1]);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    // origin of the NEXT tile's patch (the last tile re-requests its own: harmless, never stored)
    int sq2, h02, w02;
    tile_pos(more ? nxt : tile, sq2, h02, w02);
    const int x02 = 2 * w02 - (KW - 1) / 2;
    const float* org2 = d.x + (long long)sq2 * d.x_seq + (long long)(h02 - 1) * d.x_line + (long long)x02 * C;
    f32x4 pf[NCHK];
    f32x16 acc, acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
    const float* Ap = At + (ph * IWv + pw) * PITCH + kk0;
    const float* Bp = Bt + li * PITCH + kk0;
    // one tap group, expanded 14 times with a literal index: every register-array index below is a
    // constant the front end sees (as a loop or a generic lambda the stages went to scratch memory).
    // Requests of a group: the weights of group G + 2 (wrapping into the next tile) and two chunks
    // of the next tile's patch -- spread over the first groups so that no barrier ever waits for a
    // burst of them (loads return in order).
#define F2G_C32_PF(Q)                                                                              \
  if ((Q) < NCHK) {                                                                                \
    const int h_ = h02 - 1 + (rx[(Q) < NCHK ? (Q) : 0] & 255), x_ = x02 + (rx[(Q) < NCHK ? (Q) : 0] >> 8); \
    const bool ok_ = so[(Q) < NCHK ? (Q) : 0] >= 0 && h_ >= 0 && h_ < d.H && x_ >= 0 && x_ < d.Win;  \
    pf[(Q) < NCHK ? (Q) : 0] = *reinterpret_cast<const f32x4*>(ok_ ? org2 + go[(Q) < NCHK ? (Q) : 0] : zero4); \
  }
#define F2G_C32_TAP(G, U)                                                                          \
  if ((G) * TG + (U) < KH * KW) {                                                                  \
    constexpr int t_ = (G) * TG + (U), dh_ = t_ / KW, j_ = t_ - dh_ * KW;                          \
    const float* Ab = Ap + (j_ & 1) * SUBv + (dh_ * IWv + (j_ >> 1)) * PITCH;                      \
    const float* Bb = Bp + (((G) & 1) * TG + (U)) * WB;                                            \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                             \
      const float4 a = *reinterpret_cast<const float4*>(Ab + s4 * 4);                              \
      const float4 b = *reinterpret_cast<const float4*>(Bb + s4 * 4);                              \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);                          \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);                        \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);                          \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);                        \
    }                                                                                              \
  }
#define F2G_C32_GROUP(G)                                                                           \
  {                                                                                                \
    constexpr int gw_ = (G) + 2 < NG ? (G) + 2 : (G) + 2 - NG;                                     \
    constexpr int t0_ = gw_ * TG < KH * KW - 1 ? gw_ * TG : KH * KW - 1;                           \
    constexpr int t1_ = gw_ * TG + 1 < KH * KW - 1 ? gw_ * TG + 1 : KH * KW - 1;                   \
    ws[(G) & 1][0] = *reinterpret_cast<const f32x4*>(wrow + t0_ * C);                             \
    ws[(G) & 1][1] = *reinterpret_cast<const f32x4*>(wrow + t1_ * C);                             \
    F2G_C32_PF(2 * (G))                                                                            \
    F2G_C32_PF(2 * (G) + 1)                                                                        \
    F2G_C32_TAP(G, 0)                                                                              \
    F2G_C32_TAP(G, 1)                                                                              \
    *reinterpret_cast<f32x4*>(wst + ((((G) & 1) ^ 1) * TG + 0) * WB) = ws[((G) & 1) ^ 1][0];      \
    *reinterpret_cast<f32x4*>(wst + ((((G) & 1) ^ 1) * TG + 1) * WB) = ws[((G) & 1) ^ 1][1];      \
    __syncthreads();                                                                               \
  }
    static_assert(TG == 2, "the group macro handles two taps");
    float* wst = Bt + wco * PITCH + c4 * 4;
    const float* zero4 = c32_zero;
    F2G_C32_GROUP(0) F2G_C32_GROUP(1) F2G_C32_GROUP(2) F2G_C32_GROUP(3) F2G_C32_GROUP(4)
    F2G_C32_GROUP(5) F2G_C32_GROUP(6) F2G_C32_GROUP(7) F2G_C32_GROUP(8) F2G_C32_GROUP(9)
    F2G_C32_GROUP(10) F2G_C32_GROUP(11) F2G_C32_GROUP(12) F2G_C32_GROUP(13)
#undef F2G_C32_GROUP
#undef F2G_C32_TAP
#undef F2G_C32_PF
    static_assert(NG == 14, "the call list above is the tap-group loop");
    // ---- epilogue: bias + leaky ReLU; accumulator element e = pixel (e&3) + 8 (e>>2) + 4 hh of the wave
    int sq, h0, w0;
    tile_pos(tile, sq, h0, w0);
    float* ys = d.y + (long long)sq * d.y_seq;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const int oh = h0 + q / TW_, ow = w0 + q % TW_;
      if (oh < d.H && ow < d.Wout) {
        float v = acc[e] + acc2[e] + bias;
        if (d.lrelu_slope != 0.f) v = v > 0.f ? v : d.lrelu_slope * v;
        ys[(long long)oh * d.y_line + (long long)ow * C + li] = v;
      }
    }
    // (the barrier that closed the last tap group: every wave is done with this tile's patch)
    if (more) {
#pragma unroll
      for (int q = 0; q < NCHK; ++q)
        if (so[q] >= 0) *reinterpret_cast<f32x4*>(At + so[q]) = pf[q];
    }
    __syncthreads();
  }
}

// ---- data gradient of the same layer (transposed conv) -------------------------------------
// gx[h, x, ci] = sum_{dh, j, co} g[h + 1 - dh, (x + 4 - j) / 2, co] * w[co, ci, dh, j]   over the taps
// with x + 4 - j even.  Input columns of one parity e = x & 1 use the taps j = e + 2u (5 taps for even
// x, 4 for odd) and read CONSECUTIVE gradient columns m + 2 - u (x = 2m + e), so per parity this is a
// stride-1 direct convolution over g: a block stages the gradient patch of its 8 x 16 outputs once
// (10 rows x 20 columns x 32 channels) and streams the 15 / 12 transposed 32 x 32 weight tiles.
// As residue GEMMs (N = 32, windows re-gathered per tap) the same work ran at 65-70 TFLOP/s.
// d.x = g (S, H, Wout, 32), d.y = gx (S, H, Win, 32), d.w = wT[27 taps][ci][co].
constexpr int GW = TW + 4;               // staged gradient columns
constexpr int GSUB = IH * GW * PITCH;

template <bool P3>
__global__ __launch_bounds__(512, 2) void conv32_s2_dgrad_kernel(const f2g_conv32_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* At = sm;                 // [IH][GW][PITCH]
  float* Bt = sm + GSUB;          // [2 buffers][TG taps][C][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int e = blockIdx.z;                      // parity of the input column
  const int Wp = (d.Win + 1 - e) / 2;            // columns of this parity
  const int tiles_w = ((d.Win + 1) / 2 + TW - 1) / TW;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int s = blockIdx.y;
  const int h0 = th * TH, m0 = tw * TW;
  if (m0 >= Wp) return;
  const float* gs = d.x + (long long)s * d.x_seq;
  for (int i = tid; i < IH * GW * (C / 4); i += 512) {
    const int c4 = i & 7;
    int px = i >> 3;
    // (split-bf16: 8-byte stores, two pixels per 16-lane group: pair pixels p and p+4, whose 144-byte
    // pitch puts them on disjoint banks; IH*GW is a multiple of 8)
    if constexpr (P3) px = (px & ~7) | ((px & 1) << 2) | ((px & 7) >> 1);
    const int r = px / GW, xc = px - r * GW;
    const int h = h0 - 1 + r, c = m0 - 2 + xc;
    const bool ok = h >= 0 && h < d.H && c >= 0 && c < d.Wout;
    const float* p = ok ? gs + (long long)h * d.x_line + (long long)c * C + c4 * 4 : c32_zero;
    if constexpr (P3) store_split4(At + (r * GW + xc) * PITCH + c4 * 2, *reinterpret_cast<const float4*>(p));
    else *reinterpret_cast<float4*>(At + (r * GW + xc) * PITCH + c4 * 4) = *reinterpret_cast<const float4*>(p);
  }
  const int nu = e ? 4 : 5, NT = KH * nu;
  // weights: thread = (tap of the group, ci, 4 output channels of the forward conv)
  const int wu = tid >> 8, wci = (tid & 255) >> 3, wc4 = tid & 7;
  auto wsrc = [&](int ti) {                      // tap index inside this parity -> weight tile
    ti = ti < NT ? ti : NT - 1;
    const int dh = ti / nu, u = ti - dh * nu;
    return d.w + (long long)(dh * KW + e + 2 * u) * (C * C) + wci * C + wc4 * 4;
  };
  if constexpr (P3) store_presplit4(Bt + wu * WB + wci * PITCH + wc4 * 2,
                                    *reinterpret_cast<const float4*>(wsrc(wu)));
  else *reinterpret_cast<float4*>(Bt + wu * WB + wci * PITCH + wc4 * 4) =
      *reinterpret_cast<const float4*>(wsrc(wu));
  __syncthreads();

  f32x16 acc, acc2;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
  const int pg = wave & 3, myu = wave >> 2;
  const int p = pg * 32 + li;
  const int ph = p >> 4, pw = p & 15;
  const int kk0 = hh * 16;
  const int NG = (NT + TG - 1) / TG;
  for (int gidx = 0; gidx < NG; ++gidx) {
    const int cur = gidx & 1;
    const float4 wn = *reinterpret_cast<const float4*>(wsrc((gidx + 1) * TG + wu));
    {
      const int ti = gidx * TG + myu;
      if (ti < NT) {
        const int dh = ti / nu, u = ti - dh * nu;
        const float* Ab = At + ((ph + 2 - dh) * GW + pw + 4 - u) * PITCH + (P3 ? 0 : kk0);
        const float* Bb = Bt + (cur * TG + myu) * WB + li * PITCH + (P3 ? 0 : kk0);
        if constexpr (P3) tap_mfma3(Ab, Bb, hh, acc, acc2);
        else
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const float4 a = *reinterpret_cast<const float4*>(Ab + s4 * 4);
          const float4 b = *reinterpret_cast<const float4*>(Bb + s4 * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
        }
      }
    }
    if constexpr (P3) store_presplit4(Bt + ((cur ^ 1) * TG + wu) * WB + wci * PITCH + wc4 * 2, wn);
    else *reinterpret_cast<float4*>(Bt + ((cur ^ 1) * TG + wu) * WB + wci * PITCH + wc4 * 4) = wn;
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] += acc2[q];
  float* red = At;  // [4 pixel groups][16][64 lanes] = 16 KB <= the patch
  if (myu == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(pg * 16 + q) * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (myu == 1) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] += red[(pg * 16 + q) * 64 + lane];
  // optional: leaky-ReLU backward of the layer below (+ feature-matching term) fused into the
  // store, column sums of the result = that layer's bias gradient
  const bool msk = d.mask_src != nullptr, fm = d.fm_ref != nullptr;
  const float fmw = fm ? d.fm_w * (d.fm_wdev ? d.fm_wdev[0] : 1.f) : 0.f;
  float cs = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int px = pg * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
    const int oh = h0 + (px >> 4), ox = 2 * (m0 + (px & 15)) + e;
    if (oh < d.H && ox < d.Win) {
      const long long off = (long long)s * d.y_seq + (long long)oh * d.y_line + (long long)ox * C + li;
      float v = acc[q];
      if (msk) {
        const float y = d.mask_src[off];
        if (fm) {
          const float dl = y - d.fm_ref[off];
          v += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
        }
        v *= y > 0.f ? 1.f : d.mask_slope;
      }
      cs += v;
      d.y[off] = v;
    }
  }
  if (d.colsum) {
    cs += __shfl_xor(cs, 32);
    if (hh == 0) atomicAdd(d.colsum + li, cs);
  }
}

// ---- persistent data-gradient kernel (round 3; exact fp32): the forward kernel's structure --------
// One column parity E per block (blockIdx.y): 15 (even) / 12 (odd) taps = 8 / 6 tap groups, all run by
// every wave on its 32 input pixels; a block walks tiles blockIdx.x, + gridDim.x, ... of its parity
// with the next gradient patch (7 x 16 bytes per thread) in flight; the column sums of the masked
// result (bias gradient of the layer below) are kept per lane across tiles: one atomic per block.
template <int TH_, int TW_, int E>
__device__ __forceinline__ void conv32_dgrad_p_body(const f2g_conv32_desc& d, float* sm, int tiles_w,
                                                    int tiles_h, int ntiles) {
  constexpr int IHv = TH_ + KH - 1, GWv = TW_ + 4;
  constexpr int NCHK = (IHv * GWv * (C / 4) + 255) / 256;
  constexpr int NU = E ? 4 : 5, NTAP = KH * NU, NG = (NTAP + TG - 1) / TG;
  static_assert(NG % 2 == 0 && TG == 2, "weight double buffer: two taps per group, an even group count");
  static_assert(NCHK <= 2 * NG, "patch chunks are requested one or two per tap group");
  float* At = sm;                       // [IHv][GWv][PITCH]
  float* Bt = sm + IHv * GWv * PITCH;   // [2 buffers][TG taps][C][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int c4 = tid & 7;
  const int Wp = (d.Win + 1 - E) / 2;   // input columns of this parity
  int so[NCHK], go[NCHK], rx[NCHK];
#pragma unroll
  for (int q = 0; q < NCHK; ++q) {
    const int px = (tid >> 3) + 32 * q;
    const int r = px / GWv, xc = px - r * GWv;
    so[q] = px < IHv * GWv ? (r * GWv + xc) * PITCH + c4 * 4 : -1;
    go[q] = (int)(r * d.x_line) + xc * C + c4 * 4;
    rx[q] = r | (xc << 8);
  }
  auto tile_pos = [&](int tile, int& sq, int& h0, int& m0) {
    const int tw = tile % tiles_w, rest = tile / tiles_w;
    const int th = rest % tiles_h;
    sq = rest / tiles_h;
    h0 = th * TH_;
    m0 = tw * TW_;
  };
  const int wci = tid >> 3;
  const float* wrow = d.w + wci * C + c4 * 4;
  float* wst = Bt + wci * PITCH + c4 * 4;
  const float* zero4 = c32_zero;
  const int p = wave * 32 + li;
  const int ph = p / TW_, pw = p % TW_;
  const int kk0 = hh * 16;
  const bool msk = d.mask_src != nullptr, fm = d.fm_ref != nullptr;
  const float fmw = fm ? d.fm_w * (d.fm_wdev ? d.fm_wdev[0] : 1.f) : 0.f;
  float cs = 0.f;

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  f32x4 ws[2][TG];
#define F2G_C32_WT(T) ((((T) < NTAP ? (T) : NTAP - 1) / NU) * KW + E + 2 * (((T) < NTAP ? (T) : NTAP - 1) % NU))
  {
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
    const float* org = d.x + (long long)sq * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)(m0 - 2) * C;
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int h = h0 - 1 + (rx[q] & 255), c = m0 - 2 + (rx[q] >> 8);
      const bool ok = so[q] >= 0 && h >= 0 && h < d.H && c >= 0 && c < d.Wout;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? org + go[q] : zero4);
      if (so[q] >= 0) *reinterpret_cast<f32x4*>(At + so[q]) = v;
    }
    *reinterpret_cast<f32x4*>(wst) = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(0) * (C * C));
    *reinterpret_cast<f32x4*>(wst + WB) = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(1) * (C * C));
    ws[1][0] = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(2) * (C * C));
    ws[1][1] = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(3) * (C * C));
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    int sq2, h02, m02;
    tile_pos(more ? nxt : tile, sq2, h02, m02);
    const float* org2 = d.x + (long long)sq2 * d.x_seq + (long long)(h02 - 1) * d.x_line + (long long)(m02 - 2) * C;
    f32x4 pf[NCHK];
    f32x16 acc, acc2;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
    const float* Ap = At + ((ph + 2) * GWv + pw + 4) * PITCH + kk0;
    const float* Bp = Bt + li * PITCH + kk0;
#define F2G_C32_PF(Q)                                                                              \
  if ((Q) < NCHK) {                                                                                \
    constexpr int q_ = (Q) < NCHK ? (Q) : 0;                                                       \
    const int h_ = h02 - 1 + (rx[q_] & 255), c_ = m02 - 2 + (rx[q_] >> 8);                         \
    const bool ok_ = so[q_] >= 0 && h_ >= 0 && h_ < d.H && c_ >= 0 && c_ < d.Wout;                 \
    pf[q_] = *reinterpret_cast<const f32x4*>(ok_ ? org2 + go[q_] : zero4);                         \
  }
#define F2G_C32_TAP(G, U)                                                                          \
  if ((G) * TG + (U) < NTAP) {                                                                     \
    constexpr int t_ = (G) * TG + (U), dh_ = t_ / NU, u_ = t_ - dh_ * NU;                          \
    const float* Ab = Ap - (dh_ * GWv + u_) * PITCH;                                               \
    const float* Bb = Bp + (((G) & 1) * TG + (U)) * WB;                                            \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                             \
      const float4 a = *reinterpret_cast<const float4*>(Ab + s4 * 4);                              \
      const float4 b = *reinterpret_cast<const float4*>(Bb + s4 * 4);                              \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);                          \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);                        \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);                          \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);                        \
    }                                                                                              \
  }
#define F2G_C32_GROUP(G)                                                                           \
  if ((G) < NG) {                                                                                  \
    constexpr int gw_ = ((G) + 2) % NG;                                                            \
    ws[(G) & 1][0] = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(gw_ * TG) * (C * C));       \
    ws[(G) & 1][1] = *reinterpret_cast<const f32x4*>(wrow + F2G_C32_WT(gw_ * TG + 1) * (C * C));   \
    F2G_C32_PF(G)                                                                                  \
    F2G_C32_PF((G) + NG)                                                                           \
    F2G_C32_TAP(G, 0)                                                                              \
    F2G_C32_TAP(G, 1)                                                                              \
    *reinterpret_cast<f32x4*>(wst + ((((G) & 1) ^ 1) * TG + 0) * WB) = ws[((G) & 1) ^ 1][0];       \
    *reinterpret_cast<f32x4*>(wst + ((((G) & 1) ^ 1) * TG + 1) * WB) = ws[((G) & 1) ^ 1][1];       \
    __syncthreads();                                                                               \
  }
    F2G_C32_GROUP(0) F2G_C32_GROUP(1) F2G_C32_GROUP(2) F2G_C32_GROUP(3)
    F2G_C32_GROUP(4) F2G_C32_GROUP(5) F2G_C32_GROUP(6) F2G_C32_GROUP(7)
    static_assert(NG <= 8, "the call list above is the tap-group loop");
#undef F2G_C32_GROUP
#undef F2G_C32_TAP
#undef F2G_C32_PF
    // ---- epilogue: optional leaky-ReLU backward of the layer below (+ feature-matching term)
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int px = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      const int oh = h0 + px / TW_, om = m0 + px % TW_;
      if (oh < d.H && om < Wp) {
        const long long off = (long long)sq * d.y_seq + (long long)oh * d.y_line + (long long)(2 * om + E) * C + li;
        float v = acc[q] + acc2[q];
        if (msk) {
          const float y = d.mask_src[off];
          if (fm) {
            const float dl = y - d.fm_ref[off];
            v += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
          }
          v *= y > 0.f ? 1.f : d.mask_slope;
        }
        cs += v;
        d.y[off] = v;
      }
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < NCHK; ++q)
        if (so[q] >= 0) *reinterpret_cast<f32x4*>(At + so[q]) = pf[q];
    }
    __syncthreads();
  }
#undef F2G_C32_WT
  if (d.colsum) {
    cs += __shfl_xor(cs, 32);
    if (hh == 0) atomicAdd(d.colsum + li, cs);
  }
}

template <int TH_, int TW_>
__global__ __launch_bounds__(256, 2) void conv32_s2_dgrad_p_kernel(const f2g_conv32_desc d, int tiles_w0,
                                                                  int tiles_w1, int tiles_h) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  if (blockIdx.y == 0) conv32_dgrad_p_body<TH_, TW_, 0>(d, sm, tiles_w0, tiles_h, tiles_w0 * tiles_h * d.S);
  else conv32_dgrad_p_body<TH_, TW_, 1>(d, sm, tiles_w1, tiles_h, tiles_w1 * tiles_h * d.S);
}

// ---- weight gradient of the same layer ---------------------------------------------------------
// gw[co][tap][ci] += sum_px g[px][co] * x[px -> tap][ci].  MFMA rows = co, columns = ci, reduction =
// the pixels of an 8 x 16 output tile; a block walks many tiles with the 27 (32 x 32) accumulator
// tiles spread over its 8 waves (wave w owns taps w, w+8, w+16 over all 128 pixels and pixels
// [16w, 16w+16) of taps 24..26) and leaves with one atomic per (block, element).  Both fragments are
// single-float LDS reads at per-lane base + immediate offsets (the tile's pixels are walked in a
// fully unrolled loop): no address arithmetic next to the MFMAs.  The implicit GEMM (M = 32,
// windows re-gathered per tap) ran this at 57-61 TFLOP/s.
// d.x = layer input (S, H, Win, 32), d.y = gradient of the pre-activation (S, H, Wout, 32) (read),
// d.w is unused, d.gw = (32, 27*32) accumulated atomically.
constexpr int GT = TH * TW * C;          // floats of the staged gradient tile [px][co]

__device__ __forceinline__ int px_off(int px) { return ((px >> 4) * IW + (px & 15)) * PITCH; }

__global__ __launch_bounds__(512, 2) void conv32_s2_wgrad_kernel(const f2g_conv32_desc d, float* gw,
                                                                 int tiles_h, int tiles_w,
                                                                 int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* At = sm;                 // x patch [2 parities][IH][IW][PITCH]
  float* Gt = sm + 2 * SUB;       // g tile [128 px][32 co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int ntiles = d.S * tiles_h * tiles_w;
  f32x16 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // per-lane bases: tap t -> patch offset; k slot hh = pixel parity inside a k step
  int tb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int t = wave + 8 * a;
    const int dh = t / KW, j = t - dh * KW;
    tb[a] = (j & 1) * SUB + (dh * IW + (j >> 1)) * PITCH + li + hh * PITCH;
  }
  int ts[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int t = 24 + a;
    const int dh = t / KW, j = t - dh * KW;
    ts[a] = (j & 1) * SUB + (dh * IW + (j >> 1)) * PITCH + li + hh * PITCH;
  }
  const int gb = hh * C + li;
  const int t0 = blockIdx.x * tiles_per_block;
  for (int ti = t0; ti < t0 + tiles_per_block && ti < ntiles; ++ti) {
    const int s = ti / (tiles_h * tiles_w), rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const float* xs = d.x + (long long)s * d.x_seq;
    const float* gs = d.y + (long long)s * d.y_seq;
    __syncthreads();   // the previous tile's readers are done
    const int x0 = 2 * w0 - (KW - 1) / 2;
    for (int i = tid; i < IH * (2 * IW - 1) * (C / 4); i += 512) {
      const int c4 = i & 7;
      const int px = i >> 3;
      const int r = px / (2 * IW - 1), xr = px - r * (2 * IW - 1);
      const int h = h0 - 1 + r, x = x0 + xr;
      const bool ok = h >= 0 && h < d.H && x >= 0 && x < d.Win;
      const float* p = ok ? xs + (long long)h * d.x_line + (long long)x * C + c4 * 4 : c32_zero;
      *reinterpret_cast<float4*>(At + (xr & 1) * SUB + (r * IW + (xr >> 1)) * PITCH + c4 * 4) =
          *reinterpret_cast<const float4*>(p);
    }
    for (int i = tid; i < TH * TW * (C / 4); i += 512) {
      const int c4 = i & 7, px = i >> 3;
      const int h = h0 + (px >> 4), w = w0 + (px & 15);
      const bool ok = h < d.H && w < d.Wout;
      const float* p = ok ? gs + (long long)h * d.y_line + (long long)w * C + c4 * 4 : c32_zero;
      *reinterpret_cast<float4*>(Gt + px * C + c4 * 4) = *reinterpret_cast<const float4*>(p);
    }
    __syncthreads();
    // taps owned over the whole tile: 64 k steps of two pixels
#pragma unroll
    for (int st = 0; st < TH * TW / 2; ++st) {
      const float a = Gt[gb + st * 2 * C];
      const int po = px_off(2 * st);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, At[tb[q] + po], acc[q], 0, 0, 0);
    }
    // taps 24..26: this wave's 16 pixels (8 k steps); pixel index = 16*wave + 2*st + hh
    {
      const float* Gw = Gt + gb + wave * 16 * C;
      const int pbase = ((wave)*IW) * PITCH;   // pixel row = wave (16 pixels per tile row)
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const float a = Gw[st * 2 * C];
        const int po = pbase + (2 * st) * PITCH;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          acc[3 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, At[ts[q] + po], acc[3 + q], 0, 0, 0);
      }
    }
  }
  // ---- flush: taps owned by one wave go straight out; the three shared taps are summed over the
  // waves through LDS first
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = wave + 8 * q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
      atomicAdd(gw + co * (KH * KW * C) + t * C + li, acc[q][e]);
    }
  }
  float* red = sm;   // [8 waves][16][64]
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[3 + q][e];
    __syncthreads();
    if (wave == q) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 16 + e) * 64 + lane];
        const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
        atomicAdd(gw + co * (KH * KW * C) + (24 + q) * C + li, v);
      }
    }
  }
}

// ---- weight gradient, double-buffered (round 4; exact fp32) ----------------------------------------
// conv32_s2_wgrad_kernel above keeps the matrix pipe 64 % busy (PMC, profiles/r04_pmc_lean_fp32.txt):
// a block stages a tile with nothing to overlap it but the CU's other block -- launched in the same
// phase -- and two barriers bracket every tile.  Same decomposition here (wave w owns taps w, w + 8,
// w + 16 over the whole tile and tile row w of taps 24..26; single-float fragment reads at per-lane bases +
// immediate offsets), but ONE block of 8 waves per CU with TWO staged tiles in its 147 KB of LDS: the
// next tile's rows (9 x 16 bytes per thread) are requested before the current tile's 216 MFMAs per wave
// and stored into the other buffer behind them; one barrier per tile.  (Four waves with seven taps each
// and two blocks per CU -- no shared taps at all -- need 112 accumulator + 68 staging registers: spills.)
template <int TH_, int TW_>
__global__ __launch_bounds__(512, 1) void conv32_s2_wgrad_p_kernel(const f2g_conv32_desc d, float* gw,
                                                                   int tiles_h, int tiles_w,
                                                                   int tiles_per_block) {
  // tiles of 8 x 16 or 16 x 8 output pixels (the launcher takes the shape that wastes fewer columns of
  // the band); a k step = two neighbouring columns of a tile row in both
  constexpr int IHv = TH_ + KH - 1, IWv = TW_ + (KW - 1) / 2, XW = 2 * IWv - 1;
  constexpr int SUBv = IHv * IWv * PITCH + 16;
  constexpr int XCH = (IHv * XW * (C / 4) + 511) / 512;    // 7 patch chunks per thread
  constexpr int GCH = TH_ * TW_ * (C / 4) / 512;           // 2 gradient chunks per thread
  constexpr int BUF = 2 * SUBv + GT;                       // floats of one staged tile
  static_assert(TH_ * TW_ == 128 && (XW == 39 || XW == 23), "tile shapes of the magic divisions below");
  auto pxo = [](int px) { return ((px / TW_) * IWv + (px % TW_)) * PITCH; };   // patch offset of tile pixel px
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int c4 = tid & 7, px0 = tid >> 3;
  const int ntiles = d.S * tiles_h * tiles_w;
  f32x16 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  int tb[3], ts[3];               // per-lane patch offsets of the owned / shared taps (k slot hh = pixel parity)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int t = wave + 8 * a;
    int dh = t / KW, j = t - dh * KW;
    tb[a] = (j & 1) * SUBv + (dh * IWv + (j >> 1)) * PITCH + li + hh * PITCH;
    t = 24 + a;
    dh = t / KW;
    j = t - dh * KW;
    ts[a] = (j & 1) * SUBv + (dh * IWv + (j >> 1)) * PITCH + li + hh * PITCH;
  }
  const int gb = 2 * SUBv + hh * C + li;
  auto tile_pos = [&](int ti, int& s, int& h0, int& w0) {
    s = ti / (tiles_h * tiles_w);
    const int rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w;
    h0 = th * TH_;
    w0 = (rem - th * tiles_w) * TW_;
  };
  // chunk q of this thread: patch pixel px0 + 64 q = (row r, column xr), recomputed per tile
  auto load_tile = [&](int ti, f32x4 (&pxr)[XCH], f32x4 (&pgr)[GCH]) {
    int s, h0, w0;
    tile_pos(ti, s, h0, w0);
    const int x0 = 2 * w0 - (KW - 1) / 2;
    const float* org = d.x + (long long)s * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)x0 * C + c4 * 4;
#pragma unroll
    for (int q = 0; q < XCH; ++q) {
      const int px = px0 + 64 * q;
      const int r = (px * (XW == 39 ? 1681 : 2850)) >> 16, xr = px - r * XW;     // px / XW for px < 512
      const int h = h0 - 1 + r, x = x0 + xr;
      const bool ok = px < IHv * XW && h >= 0 && h < d.H && x >= 0 && x < d.Win;
      pxr[q] = *reinterpret_cast<const f32x4*>(ok ? org + (long long)r * d.x_line + xr * C : c32_zero);
    }
    const float* gs = d.y + (long long)s * d.y_seq + c4 * 4;
#pragma unroll
    for (int q = 0; q < GCH; ++q) {
      const int px = px0 + 64 * q;
      const int h = h0 + px / TW_, w = w0 + px % TW_;
      const bool ok = h < d.H && w < d.Wout;
      pgr[q] = *reinterpret_cast<const f32x4*>(ok ? gs + (long long)h * d.y_line + (long long)w * C : c32_zero);
    }
  };
  auto store_tile = [&](float* buf, const f32x4 (&pxr)[XCH], const f32x4 (&pgr)[GCH]) {
#pragma unroll
    for (int q = 0; q < XCH; ++q) {
      const int px = px0 + 64 * q;
      const int r = (px * (XW == 39 ? 1681 : 2850)) >> 16, xr = px - r * XW;
      if (px < IHv * XW)
        *reinterpret_cast<f32x4*>(buf + (xr & 1) * SUBv + (r * IWv + (xr >> 1)) * PITCH + c4 * 4) = pxr[q];
    }
#pragma unroll
    for (int q = 0; q < GCH; ++q)
      *reinterpret_cast<f32x4*>(buf + 2 * SUBv + (px0 + 64 * q) * C + c4 * 4) = pgr[q];
  };
  const int t0 = blockIdx.x * tiles_per_block;
  int tend = t0 + tiles_per_block;
  if (tend > ntiles) tend = ntiles;
  if (t0 >= tend) return;
  {
    f32x4 pxr[XCH], pgr[GCH];
    load_tile(t0, pxr, pgr);
    store_tile(sm, pxr, pgr);
  }
  __syncthreads();
  int cur = 0;
  for (int ti = t0; ti < tend; ++ti) {
    const bool more = ti + 1 < tend;
    const float* T = sm + cur * BUF;
    f32x4 pxr[XCH], pgr[GCH];
    load_tile(more ? ti + 1 : ti, pxr, pgr);      // (the last tile re-requests its own: never stored)
    // taps owned over the whole tile: 64 k steps of two pixels
#pragma unroll
    for (int st = 0; st < TH_ * TW_ / 2; ++st) {
      const float a = T[gb + st * 2 * C];
      const int po = pxo(2 * st);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, T[tb[q] + po], acc[q], 0, 0, 0);
    }
    // taps 24..26: this wave's 16 pixels (8 k steps); pixel index = 16*wave + 2*st + hh
    {
      const float* Gw = T + gb + wave * 16 * C;
      const int pbase = pxo(16 * wave);        // the wave's 16 pixels: tile row `wave`, or rows 2 wave, 2 wave + 1
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const float a = Gw[st * 2 * C];
        const int po = pbase + pxo(2 * st);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          acc[3 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, T[ts[q] + po], acc[3 + q], 0, 0, 0);
      }
    }
    // the other buffer was last read in the previous iteration, which every wave left through the barrier below
    if (more) store_tile(sm + (cur ^ 1) * BUF, pxr, pgr);
    __syncthreads();
    cur ^= 1;
  }
  // ---- flush: taps owned by one wave go straight out; the three shared taps are summed over the
  // waves through LDS first
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = wave + 8 * q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
      atomicAdd(gw + co * (KH * KW * C) + t * C + li, acc[q][e]);
    }
  }
  float* red = sm;   // [8 waves][16][64]
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[3 + q][e];
    __syncthreads();
    if (wave == q) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 16 + e) * 64 + lane];
        const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
        atomicAdd(gw + co * (KH * KW * C) + (24 + q) * C + li, v);
      }
    }
  }
}

// ---- weight gradient, split-bf16 ---------------------------------------------------------------
// Same decomposition as conv32_s2_wgrad_kernel (MFMA rows = co, columns = ci, reduction = the pixels
// of an 8 x 16 output tile; wave w owns taps w, w+8, w+16 over the whole tile and tile row w of taps
// 24..26), but a bf16 MFMA wants 8 CONSECUTIVE k per lane and k = pixels is the slow axis of both
// staged operands ([pixel][channel]).  The tile is therefore staged as bf16 planes with 64-byte
// pixels (hi and lo of the gradient tile and of the two column parities of the input patch, split
// once while they are staged) and the fragments come from ds_read_b64_tr_b16: a 16-lane group reads
// 4 pixels x 16 channels and every lane receives the 4 pixels of its channel; two reads = the 8 k of
// a lane half, a tile row of 16 pixels = one k step.  The four pixels of a read are consecutive
// columns = 256 contiguous bytes: conflict-free without padding.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_pix8(const unsigned char* p) {   // pixels +0..3 and +4..7
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * 64));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ void store_split_planes(unsigned char* hi, unsigned char* lo, const float4 v) {
  float r0, r1, r2, r3, z0, z1;
  u32x2 h, l;
  h.x = pack_bf16(v.x, v.y, r0, r1);
  h.y = pack_bf16(v.z, v.w, r2, r3);
  l.x = pack_bf16(r0, r1, z0, z1);
  l.y = pack_bf16(r2, r3, z0, z1);
  *reinterpret_cast<u32x2*>(hi) = h;
  *reinterpret_cast<u32x2*>(lo) = l;
}

constexpr int XPAR = IH * IW * 64 + 64;  // bytes of one parity of a plane (+64: see SUB)
constexpr int XPL = 2 * XPAR;           // bytes of one plane of the input patch (both parities)
constexpr int GPL = TH * TW * 64;       // bytes of one plane of the gradient tile

__global__ __launch_bounds__(512, 2) void conv32_s2_wgrad3_kernel(const f2g_conv32_desc d, float* gw,
                                                                  int tiles_h, int tiles_w,
                                                                  int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  unsigned char* sm = reinterpret_cast<unsigned char*>(smf);
  unsigned char* Xh = sm;              // [2 parities][IH][IW] pixels x 32 bf16
  unsigned char* Xl = sm + XPL;
  unsigned char* Gh = sm + 2 * XPL;    // [128 px] x 32 bf16
  unsigned char* Gl = Gh + GPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int ntiles = d.S * tiles_h * tiles_w;
  f32x16 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // transposed-read lane roles: 16-lane group g4 = (channel half, k half); lane i = (pixel i>>2, quad i&3)
  const int g4 = lane >> 4, i16 = lane & 15;
  const int kpix = (g4 >> 1) * 8 + (i16 >> 2);             // pixel (tile column) of the first read
  const int chb = (g4 & 1) * 32 + (i16 & 3) * 8;           // byte offset of this lane's 4 channels
  int xo[3], xs[3];                                        // patch offsets of the owned / shared taps
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int t = wave + 8 * a;
    int dh = t / KW, j = t - dh * KW;
    xo[a] = (j & 1) * XPAR + (dh * IW + (j >> 1) + kpix) * 64 + chb;
    t = 24 + a;
    dh = t / KW;
    j = t - dh * KW;
    xs[a] = (j & 1) * XPAR + (dh * IW + (j >> 1) + kpix) * 64 + chb;
  }
  const int go = kpix * 64 + chb;
  const int t0 = blockIdx.x * tiles_per_block;
  for (int ti = t0; ti < t0 + tiles_per_block && ti < ntiles; ++ti) {
    const int s = ti / (tiles_h * tiles_w), rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const float* xsrc = d.x + (long long)s * d.x_seq;
    const float* gsrc = d.y + (long long)s * d.y_seq;
    __syncthreads();   // the previous tile's readers are done
    const int x0 = 2 * w0 - (KW - 1) / 2;
    for (int i = tid; i < IH * (2 * IW - 1) * (C / 4); i += 512) {
      const int c4 = i & 7;
      const int px = i >> 3;
      const int r = px / (2 * IW - 1), xr = px - r * (2 * IW - 1);
      const int h = h0 - 1 + r, x = x0 + xr;
      const bool ok = h >= 0 && h < d.H && x >= 0 && x < d.Win;
      const float* p = ok ? xsrc + (long long)h * d.x_line + (long long)x * C + c4 * 4 : c32_zero;
      const int off = (xr & 1) * XPAR + (r * IW + (xr >> 1)) * 64 + c4 * 8;
      store_split_planes(Xh + off, Xl + off, *reinterpret_cast<const float4*>(p));
    }
    if (tid < IH * 8) {   // the odd parity has one column less: keep its last column defined
      const int r = tid >> 3, c4 = tid & 7;
      const int off = XPAR + (r * IW + IW - 1) * 64 + c4 * 8;
      *reinterpret_cast<u32x2*>(Xh + off) = u32x2{0u, 0u};
      *reinterpret_cast<u32x2*>(Xl + off) = u32x2{0u, 0u};
    }
    for (int i = tid; i < TH * TW * (C / 4); i += 512) {
      const int c4 = i & 7, px = i >> 3;
      const int h = h0 + (px >> 4), w = w0 + (px & 15);
      const bool ok = h < d.H && w < d.Wout;
      const float* p = ok ? gsrc + (long long)h * d.y_line + (long long)w * C + c4 * 4 : c32_zero;
      store_split_planes(Gh + px * 64 + c4 * 8, Gl + px * 64 + c4 * 8, *reinterpret_cast<const float4*>(p));
    }
    __syncthreads();
    // owned taps: the 8 tile rows = 8 k steps of 16 pixels
#pragma unroll
    for (int row = 0; row < TH; ++row) {
      const bf16x8 ah = tr_pix8(Gh + go + row * 16 * 64), al = tr_pix8(Gl + go + row * 16 * 64);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bf16x8 bh = tr_pix8(Xh + xo[q] + row * IW * 64), bl = tr_pix8(Xl + xo[q] + row * IW * 64);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[q], 0, 0, 0);
      }
    }
    // taps 24..26: this wave's tile row
    {
      const int row = wave;
      const bf16x8 ah = tr_pix8(Gh + go + row * 16 * 64), al = tr_pix8(Gl + go + row * 16 * 64);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bf16x8 bh = tr_pix8(Xh + xs[q] + row * IW * 64), bl = tr_pix8(Xl + xs[q] + row * IW * 64);
        acc[3 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[3 + q], 0, 0, 0);
        acc[3 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[3 + q], 0, 0, 0);
        acc[3 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[3 + q], 0, 0, 0);
      }
    }
  }
  // ---- flush (as the fp32 kernel)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = wave + 8 * q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
      atomicAdd(gw + co * (KH * KW * C) + t * C + li, acc[q][e]);
    }
  }
  float* red = smf;   // [8 waves][16][64]
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[3 + q][e];
    __syncthreads();
    if (wave == q) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 16 + e) * 64 + lane];
        const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
        atomicAdd(gw + co * (KH * KW * C) + (24 + q) * C + li, v);
      }
    }
  }
}

}  // namespace

// precision 3 (fp32-class products on the bf16 pipe): conv32x6.hip
int f2g_conv32_fwd6_launch(const f2g_conv32_desc* d, hipStream_t st);
int f2g_conv32_dgrad6_launch(const f2g_conv32_desc* d, hipStream_t st);
int f2g_conv32_wgrad6_launch(const f2g_conv32_desc* d, float* gw, hipStream_t st);

extern "C" int f2g_conv32_s2_fwd(const f2g_conv32_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y) return F2G_EINVAL;
  if (d->S <= 0 || d->H <= 0 || d->Wout <= 0) return F2G_OK;
  if (d->Wout != (d->Win + 8 - 9) / 2 + 1) return F2G_EINVAL;
  if (d->precision != 0 && d->precision != 1 && d->precision != 3) return F2G_EINVAL;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!al(d->x) || !al(d->w) || (d->x_line & 3) || (d->x_seq & 3)) return F2G_EINVAL;
  if (d->precision == 3) {   // w = f2g_split_bf16x3 image; the output leaves as 16-byte row segments
    if (!al(d->y) || (d->y_line & 3) || (d->y_seq & 3)) return F2G_EINVAL;
    return f2g_conv32_fwd6_launch(d, (hipStream_t)stream);
  }
  const size_t smem = (size_t)(2 * SUB + 2 * TG * WB) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const int tiles = ((d->H + TH - 1) / TH) * ((d->Wout + TW - 1) / TW);
  const bool persistent = f2g_opt(F2G_OPT_CONV32_V2) != 0;
  if (d->precision == 0 && persistent && d->x_line < (1ll << 24)) {
    // tile shape: fewer wasted output pixels wins (8 x 16 on a tie)
    auto waste = [&](int th, int tw) {
      return (long long)((d->H + th - 1) / th * th) * ((d->Wout + tw - 1) / tw * tw);
    };
    const bool tall = waste(16, 8) < waste(8, 16);
    const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
    const int tiles_h = (d->H + th - 1) / th, tiles_w = (d->Wout + tw - 1) / tw;
    const long long nt = (long long)tiles_h * tiles_w * d->S;
    if (nt < (1ll << 30)) {
      const size_t sm8 = (size_t)(2 * ((8 + 2) * (16 + 4) * PITCH + 16) + 2 * TG * WB) * sizeof(float);
      const size_t sm16 = (size_t)(2 * ((16 + 2) * (8 + 4) * PITCH + 16) + 2 * TG * WB) * sizeof(float);
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd_p_kernel<8, 16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd_p_kernel<16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16);
        attr2 = true;
      }
      const int grid = (int)(nt < 512 ? nt : 512);     // 256 CUs x 2 resident blocks
      if (tall)
        hipLaunchKernelGGL((conv32_s2_fwd_p_kernel<16, 8>), dim3(grid), dim3(256), sm16,
                           (hipStream_t)stream, *d, tiles_w, tiles_h, (int)nt);
      else
        hipLaunchKernelGGL((conv32_s2_fwd_p_kernel<8, 16>), dim3(grid), dim3(256), sm8,
                           (hipStream_t)stream, *d, tiles_w, tiles_h, (int)nt);
      return f2g_check_launch();
    }
  }
  if (d->precision == 1)   // w = the f2g_split_bf16 image of the packed weights
    hipLaunchKernelGGL(conv32_s2_fwd_kernel<true>, dim3(tiles, d->S), dim3(512), smem,
                       (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(conv32_s2_fwd_kernel<false>, dim3(tiles, d->S), dim3(512), smem,
                       (hipStream_t)stream, *d);
  return f2g_check_launch();
}

extern "C" int f2g_conv32_s2_dgrad(const f2g_conv32_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y) return F2G_EINVAL;
  if (d->S <= 0 || d->H <= 0 || d->Win <= 0) return F2G_OK;
  if (d->Wout != (d->Win + 8 - 9) / 2 + 1) return F2G_EINVAL;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!al(d->x) || !al(d->w) || (d->x_line & 3) || (d->x_seq & 3)) return F2G_EINVAL;
  if (d->precision != 0 && d->precision != 1 && d->precision != 3) return F2G_EINVAL;
  if (d->precision == 3) {   // w = image of wT (864 x 32); output, mask and reference map as 16-byte row segments
    if (!al(d->y) || (d->y_line & 3) || (d->y_seq & 3) || !al(d->mask_src) || !al(d->fm_ref)) return F2G_EINVAL;
    return f2g_conv32_dgrad6_launch(d, (hipStream_t)stream);
  }
  const size_t smem = (size_t)(GSUB + 2 * TG * WB) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const int tiles = ((d->H + TH - 1) / TH) * (((d->Win + 1) / 2 + TW - 1) / TW);
  const bool persistent = f2g_opt(F2G_OPT_CONV32_V2) != 0;
  if (d->precision == 0 && persistent && d->x_line < (1ll << 24)) {
    const int Wp0 = (d->Win + 1) / 2, Wp1 = d->Win / 2;
    auto waste = [&](int th, int tw) {
      return (long long)((d->H + th - 1) / th * th) * ((Wp0 + tw - 1) / tw * tw + (Wp1 + tw - 1) / tw * tw);
    };
    const bool tall = waste(16, 8) < waste(8, 16);
    const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
    const int tiles_h = (d->H + th - 1) / th, tw0 = (Wp0 + tw - 1) / tw, tw1 = (Wp1 + tw - 1) / tw;
    const long long nt0 = (long long)tiles_h * tw0 * d->S;
    if (nt0 < (1ll << 30)) {
      const size_t sm8 = (size_t)((8 + 2) * (16 + 4) * PITCH + 2 * TG * WB) * sizeof(float);
      const size_t sm16 = (size_t)((16 + 2) * (8 + 4) * PITCH + 2 * TG * WB) * sizeof(float);
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad_p_kernel<8, 16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad_p_kernel<16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16);
        attr2 = true;
      }
      const int grid = (int)(nt0 < 256 ? nt0 : 256);   // per parity: 256 CUs x 2 resident blocks in all
      if (tall)
        hipLaunchKernelGGL((conv32_s2_dgrad_p_kernel<16, 8>), dim3(grid, 2), dim3(256), sm16,
                           (hipStream_t)stream, *d, tw0, tw1, tiles_h);
      else
        hipLaunchKernelGGL((conv32_s2_dgrad_p_kernel<8, 16>), dim3(grid, 2), dim3(256), sm8,
                           (hipStream_t)stream, *d, tw0, tw1, tiles_h);
      return f2g_check_launch();
    }
  }
  if (d->precision == 1)   // w = the f2g_split_bf16 image of the transposed tiles
    hipLaunchKernelGGL(conv32_s2_dgrad_kernel<true>, dim3(tiles, d->S, 2), dim3(512), smem,
                       (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(conv32_s2_dgrad_kernel<false>, dim3(tiles, d->S, 2), dim3(512), smem,
                       (hipStream_t)stream, *d);
  return f2g_check_launch();
}

extern "C" int f2g_conv32_s2_wgrad(const f2g_conv32_desc* d, float* gw, f2g_stream_t stream) {
  if (!d || !d->x || !d->y || !gw) return F2G_EINVAL;
  if (d->S <= 0 || d->H <= 0 || d->Wout <= 0) return F2G_OK;
  if (d->Wout != (d->Win + 8 - 9) / 2 + 1) return F2G_EINVAL;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!al(d->x) || !al(d->y) || (d->x_line & 3) || (d->x_seq & 3) || (d->y_line & 3) || (d->y_seq & 3))
    return F2G_EINVAL;
  if (d->precision == 3) return f2g_conv32_wgrad6_launch(d, gw, (hipStream_t)stream);   // operands split while staged
  const int tiles_h = (d->H + TH - 1) / TH, tiles_w = (d->Wout + TW - 1) / TW;
  const int ntiles = d->S * tiles_h * tiles_w;
  int per = (ntiles + 511) / 512;   // <= 512 blocks (two per CU): bounds the atomics
  if (per < 1) per = 1;
  if (d->precision == 1) {   // split-bf16 (the operands are split while they are staged)
    const size_t smem3 = (size_t)(2 * XPL + 2 * GPL) > (size_t)8 * 16 * 64 * 4 ? (size_t)(2 * XPL + 2 * GPL)
                                                                              : (size_t)8 * 16 * 64 * 4;
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      attr3 = true;
    }
    hipLaunchKernelGGL(conv32_s2_wgrad3_kernel, dim3((ntiles + per - 1) / per), dim3(512), smem3,
                       (hipStream_t)stream, *d, gw, tiles_h, tiles_w, per);
    return f2g_check_launch();
  }
  const size_t smem = (size_t)(2 * SUB + GT) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const bool persistent = f2g_opt(F2G_OPT_CONV32_WGRAD_V2) != 0;
  if (persistent && d->x_line < (1ll << 24)) {
    auto waste = [&](int th, int tw) {
      return (long long)((d->H + th - 1) / th * th) * ((d->Wout + tw - 1) / tw * tw);
    };
    const bool tall = waste(16, 8) < waste(8, 16);
    const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
    const int th_n = (d->H + th - 1) / th, tw_n = (d->Wout + tw - 1) / tw;
    const long long nt = (long long)d->S * th_n * tw_n;
    if (nt < (1ll << 30)) {
      int per1 = (int)((nt + 255) / 256);          // <= 256 blocks: one per CU
      if (per1 < 1) per1 = 1;
      const size_t sm8 = (size_t)2 * (2 * ((8 + 2) * (16 + 4) * PITCH + 16) + GT) * sizeof(float);
      const size_t sm16 = (size_t)2 * (2 * ((16 + 2) * (8 + 4) * PITCH + 16) + GT) * sizeof(float);
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad_p_kernel<8, 16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad_p_kernel<16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16);
        attr2 = true;
      }
      const unsigned grid = (unsigned)((nt + per1 - 1) / per1);
      if (tall)
        hipLaunchKernelGGL((conv32_s2_wgrad_p_kernel<16, 8>), dim3(grid), dim3(512), sm16, (hipStream_t)stream,
                           *d, gw, th_n, tw_n, per1);
      else
        hipLaunchKernelGGL((conv32_s2_wgrad_p_kernel<8, 16>), dim3(grid), dim3(512), sm8, (hipStream_t)stream,
                           *d, gw, th_n, tw_n, per1);
      return f2g_check_launch();
    }
  }
  hipLaunchKernelGGL(conv32_s2_wgrad_kernel, dim3((ntiles + per - 1) / per), dim3(512), smem,
                     (hipStream_t)stream, *d, gw, tiles_h, tiles_w, per);
  return f2g_check_launch();
}
