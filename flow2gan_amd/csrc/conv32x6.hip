// fp32-CLASS instances of the direct MRD band convolutions (conv32.hip; reference
// discriminators.py:171-181: Conv2d(32, 32, (3, 9), stride (1, 2), padding (1, 4))) on the bf16 matrix
// pipe: f2g_conv32_desc.precision = 3.  Every fp32 value is three bf16 pieces x = p0 + p1 + p2 and a
// product the six v_mfma_f32_32x32x16_bf16 with i + j <= 2, fp32 accumulation, smallest terms first
// (error <= ~2^-23 per product: the class of fp32 rounding -- gemm.hip, gemm_x6_kernel).  The matrix
// pipe is 2.7x less busy per product than with v_mfma_f32_32x32x2_f32 (12 MFMAs of 32 cycles per tap and
// wave instead of 16 of 64).
//
// The patch of a tile is split ONCE while it is staged (each element then feeds 27 taps x 32 outputs):
// a staged pixel is [32 p0 | 32 p1 | 32 p2 | pad] bf16 = 208 bytes (52 dwords: conflict-free
// ds_read_b128); the weights arrive as the f2g_split_bf16x3 image of the packed matrix (192 contiguous
// bytes per output channel and tap) and go to LDS unchanged.  208-byte pixels need 83 KB for the
// forward patch, so ONE block of 8 waves per CU: the blocks are persistent (a block walks tiles b,
// b + G, ...) and request the next tile's patch into registers before they compute the current one.
#include <stdlib.h>

#include "common.h"


namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 32, KH = 3, KW = 9, NTAP = KH * KW;
constexpr int PB = 208;            // bytes of a staged pixel / weight row
constexpr int TG = 4;              // taps per barrier: two for each half of the block's waves
constexpr int WBB = C * PB;        // bytes of one staged weight tile
constexpr int WCH = C * 12;        // 16-byte chunks of one weight tile in the image (32 rows x 192 bytes)

__device__ __attribute__((aligned(16))) float c6_zero[4] = {0.f, 0.f, 0.f, 0.f};

// three bf16 pieces of four floats (common.h: f2g_split3_pair, round to nearest even at every step)
__device__ __forceinline__ void split3(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  unsigned a0, a1, a2, b0, b1, b2;
  f2g_split3_pair(v.x, v.y, a0, a1, a2);
  f2g_split3_pair(v.z, v.w, b0, b1, b2);
  p0 = u32x2{a0, b0};
  p1 = u32x2{a1, b1};
  p2 = u32x2{a2, b2};
}

__device__ __forceinline__ void store_px3(unsigned char* p, const f32x4 v) {   // p = piece 0 of the chunk
  u32x2 p0, p1, p2;
  split3(v, p0, p1, p2);
  *reinterpret_cast<u32x2*>(p) = p0;
  *reinterpret_cast<u32x2*>(p + 64) = p1;
  *reinterpret_cast<u32x2*>(p + 128) = p2;
}

// one tap: A = this lane's staged pixel, B = this lane's weight row (both: piece q at + 64 q bytes, the
// lane half hh takes channels [16 ks + 8 hh, + 8) of k step ks).  Twelve MFMAs, smallest terms first.
// (Measured and dropped: the fragments of a tap requested one tap AHEAD of its MFMAs through a second
// register stage -- 96 fragment registers beside accumulators, patch and weight staging spill, 6.6 -> 7.5 ms
// over the 45 forward launches of a pass; weights two groups ahead and the patch split interleaved with the
// tap groups: no change.  Lab builds, profiles/r04_conv32x6.txt: no weight loads -12 %, no patch -11 %, no
// barriers -4 %, no MFMAs -8 %; none of them alone is the bound.)
struct frag6 {
  bf16x8 v[2][3];
};
__device__ __forceinline__ void ld6(frag6& f, const unsigned char* p) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int q = 0; q < 3; ++q) f.v[ks][q] = *reinterpret_cast<const bf16x8*>(p + q * 64 + ks * 32);
}
__device__ __forceinline__ void mm6(const frag6& a, const frag6& b, f32x16& acc0, f32x16& acc1) {
#define F2G_X6_PAIR(I, J)                                                               \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0][I], b.v[0][J], acc0, 0, 0, 0);   \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1][I], b.v[1][J], acc1, 0, 0, 0);
  F2G_X6_PAIR(2, 0) F2G_X6_PAIR(1, 1) F2G_X6_PAIR(0, 2) F2G_X6_PAIR(1, 0) F2G_X6_PAIR(0, 1) F2G_X6_PAIR(0, 0)
#undef F2G_X6_PAIR
}
__device__ __forceinline__ void tap6(const unsigned char* Ab, const unsigned char* Bb, f32x16& acc0, f32x16& acc1) {
  frag6 a, b;
  ld6(a, Ab);
  ld6(b, Bb);
  mm6(a, b, acc0, acc1);
}

// x / D for 0 <= x < 512 and the staged widths that occur (39, 23: forward patch columns; 20, 12: gradient patch)
template <int D>
__device__ __forceinline__ int div_small(int x) {
  static_assert(D == 39 || D == 23 || D == 20 || D == 12, "magic constants below");
  return (x * (D == 39 ? 1681 : D == 23 ? 2850 : D == 20 ? 3277 : 5462)) >> 16;
}

// MFMA row r (0..31) of pixel group pg -> pixel (row, column) of a TH x TW tile.  16-wide tiles: the wave's
// rows 16..31 are the next tile row, whose staged pixels sit (IW - 16) * 52 = 16 (mod 64) dwords further than
// "16 lanes on" -- in ds_read_b128's lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} that puts four
// lanes of the second row on the banks of the first (measured: 31 % of the LDS cycles were conflicts).
// Rotating the second row's columns by 12 makes every group's sixteen 16-byte reads hit distinct banks.
template <int TW_>
__device__ __forceinline__ void px_of_row(int pg, int r, int& ph, int& pw) {
  if (TW_ == 16) {
    ph = 2 * pg + (r >> 4);
    pw = (r + 12 * (r >> 4)) & 15;
  } else {
    const int p = pg * 32 + r;
    ph = p / TW_;
    pw = p % TW_;
  }
}

// weights of a tap group: 4 tiles x 384 chunks of 16 bytes = 3 chunks per thread (512 threads)
struct wplan {
  int src[3];      // byte offset inside the image WITHOUT the tap term (row * row_stride + part * 16)
  int dst[3];      // byte offset inside a group's LDS buffer
  int u[3];        // tap of the group this chunk belongs to
};

// ---- forward ----------------------------------------------------------------------------------------
// 8 waves: pixel group pg = wave & 3 (32 of the tile's 128 pixels), tap half = wave >> 2 (taps 4g, 4g+1 /
// 4g+2, 4g+3 of group g); the halves' partial tiles meet through LDS at the end of a tile.
template <int TH_, int TW_>
__global__ __launch_bounds__(512, 1) void conv32_s2_fwd6_kernel(const f2g_conv32_desc d, int tiles_w,
                                                               int tiles_h, int ntiles) {
  constexpr int IHv = TH_ + KH - 1, IWv = TW_ + (KW - 1) / 2, XW = 2 * IWv - 1;
  constexpr int SUBB = IHv * IWv * PB + 64;             // bytes of one column parity of the patch
  constexpr int NCHK = (IHv * XW * (C / 4) + 511) / 512;
  // 7 groups of four taps + one empty pipeline slot: with an EVEN count the LDS buffer and the register stage
  // of a group are the same in every tile (group g: buffer g & 1)
  constexpr int NG = (NTAP + TG - 1) / TG + 1;
  static_assert(NG % 2 == 0, "the weight double buffer must come back to buffer 0 at a tile's end");
  static_assert(TH_ * TW_ == 128, "a block owns 128 output pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* At = smb;                               // [2 parities][IHv][IWv] pixels
  unsigned char* Bt = smb + 2 * SUBB;                    // [2 buffers][TG taps][32 rows]
  float* red = reinterpret_cast<float*>(Bt + 2 * TG * WBB);   // [4 pixel groups][16][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int pg = wave & 3, half = wave >> 2;
  const int c4 = tid & 7;
  const int px0 = tid >> 3;
  auto tile_pos = [&](int tile, int& sq, int& h0, int& w0) {
    const int tw = tile % tiles_w, rest = tile / tiles_w;
    const int th = rest % tiles_h;
    sq = rest / tiles_h;
    h0 = th * TH_;
    w0 = tw * TW_;
  };
  // chunk q of this thread = 4 channels of patch pixel px0 + 64 q = (row r, column xr); recomputed per
  // tile (a multiplication) instead of kept in 21 registers beside the fragment stages
  auto load_patch = [&](int tile, f32x4 (&pf)[NCHK]) {
    int sq, h0, w0;
    tile_pos(tile, sq, h0, w0);
    const int x0 = 2 * w0 - (KW - 1) / 2;
    const float* org = d.x + (long long)sq * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)x0 * C + c4 * 4;
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      const int r = div_small<XW>(px), xr = px - r * XW;
      const int h = h0 - 1 + r, x = x0 + xr;
      const bool ok = px < IHv * XW && h >= 0 && h < d.H && x >= 0 && x < d.Win;
      pf[q] = *reinterpret_cast<const f32x4*>(ok ? org + (long long)r * d.x_line + xr * C : c6_zero);
    }
  };
  auto store_patch = [&](const f32x4 (&pf)[NCHK]) {
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      const int r = div_small<XW>(px), xr = px - r * XW;
      if (px < IHv * XW) store_px3(At + (xr & 1) * SUBB + (r * IWv + (xr >> 1)) * PB + c4 * 8, pf[q]);
    }
  };
  // weights: the image is [co][27 taps][3 pieces][32] bf16 = 192 bytes per (co, tap)
  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(d.w);
  wplan wp;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = tid + 512 * k;
    const int u = c / WCH, rem = c - u * WCH;
    const int row = rem / 12, part = rem - row * 12;
    wp.u[k] = u;
    wp.src[k] = row * (NTAP * 192) + part * 16;
    wp.dst[k] = u * WBB + row * PB + part * 16;
  }
  auto load_w = [&](int g, u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int t = g * TG + wp.u[k];
      t = t < NTAP ? t : NTAP - 1;
      wn[k] = *reinterpret_cast<const u32x4*>(wimg + wp.src[k] + t * 192);
    }
  };
  auto store_w = [&](int buf, const u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(Bt + buf * (TG * WBB) + wp.dst[k]) = wn[k];
  };
  int ph, pw;                                   // this lane's output pixel inside the tile
  px_of_row<TW_>(pg, li, ph, pw);
  const unsigned char* Ap = At + (ph * IWv + pw) * PB + hh * 16;
  const unsigned char* Bp = Bt + li * PB + hh * 16;
  const float bias = d.bias ? d.bias[li] : 0.f;

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  // weight pipeline: group g + 1 is requested at the start of group g and goes to LDS at its end (two
  // groups ahead through a second register stage measured the same and spilled beside the fragment stages)
  u32x4 wn[3];
  {
    f32x4 pf[NCHK];
    load_patch(tile, pf);
    load_w(0, wn);
    store_patch(pf);
    store_w(0, wn);
  }
  __syncthreads();
  // patch offset of tap t (a literal after unrolling)
#define F2G_C6_AOFF(T) ((((T) % KW) & 1) * SUBB + (((T) / KW) * IWv + (((T) % KW) >> 1)) * PB)
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    f32x4 pf[NCHK];
    load_patch(more ? nxt : tile, pf);          // (the last tile re-requests its own: never stored)
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      load_w(g + 1 < NG ? g + 1 : 0, wn);       // next group (group 0 of the next tile after the last)
      const int buf = g & 1;
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2) {
        // tap of this wave half: 4 g + 2 half + u2 (both candidates are literals; the choice is wave-uniform)
        const int t0 = g * TG + u2, t1 = g * TG + 2 + u2;
        if (t0 < NTAP && (half ? t1 : t0) < NTAP)
          tap6(Ap + (half ? F2G_C6_AOFF(t1) : F2G_C6_AOFF(t0)), Bp + (buf * TG + half * 2 + u2) * WBB, acc0, acc1);
      }
      store_w(buf ^ 1, wn);
      __syncthreads();
    }
#undef F2G_C6_AOFF
    // the halves' partial tiles meet; the next patch goes to LDS (every wave is done with this one)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] += acc1[e];
    if (half == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(pg * 16 + e) * 64 + lane] = acc0[e];
    }
    if (more) store_patch(pf);
    __syncthreads();
    if (half == 0) {
      // (round 5) the finished 32 pixels x 32 channels leave through this pixel group's 4 KB of `red`, turned to
      // [pixel][channel]: four 16-byte stores per lane instead of sixteen 4-byte ones
      int sq, h0, w0;
      tile_pos(tile, sq, h0, w0);
      float* ys = d.y + (long long)sq * d.y_seq;
      float* turn = red + pg * (16 * 64);
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = acc0[e] + red[(pg * 16 + e) * 64 + lane] + bias;
        if (d.lrelu_slope != 0.f) v[e] = v[e] > 0.f ? v[e] : d.lrelu_slope * v[e];
      }
      __builtin_amdgcn_wave_barrier();            // (the partner half's partial sums are in registers)
#pragma unroll
      for (int e = 0; e < 16; ++e) turn[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + li] = v[e];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int qh, qw;
        px_of_row<TW_>(pg, (lane >> 3) + 8 * j, qh, qw);
        const int oh = h0 + qh, ow = w0 + qw;
        const f32x4 u = *reinterpret_cast<const f32x4*>(turn + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7));
        if (oh < d.H && ow < d.Wout)
          *reinterpret_cast<f32x4*>(ys + (long long)oh * d.y_line + (long long)ow * C + 4 * (lane & 7)) = u;
      }
    }
  }
}

// ---- data gradient -----------------------------------------------------------------------------------
// gx[h, x, ci] = sum_{dh, j, co} g[h + 1 - dh, (x + 4 - j) / 2, co] * w[co, ci, dh, j] over the taps with
// x + 4 - j even (conv32.hip): input columns of parity E = x & 1 use the taps j = E + 2u (5 / 4 of them per
// row) and read CONSECUTIVE gradient columns m + 2 - u (x = 2 m + E) -- both parities read the SAME
// gradient patch.  Here a tile is 128 positions m x BOTH parities: wave half 0 computes the even input
// columns (15 taps), half 1 the odd ones (12 taps), from one staged patch, and nothing has to be
// reduced across waves.  d.x = g (S, H, Wout, 32), d.y = gx (S, H, Win, 32),
// d.w = the f2g_split_bf16x3 image of wT [27 taps][ci][co] as an (864, 32) matrix: 192 bytes per (tap, ci).
template <int TH_, int TW_>
__global__ __launch_bounds__(512, 1) void conv32_s2_dgrad6_kernel(const f2g_conv32_desc d, int tiles_w,
                                                                 int tiles_h, int ntiles) {
  constexpr int IHv = TH_ + KH - 1, GWv = TW_ + 4;
  constexpr int NCHK = (IHv * GWv * (C / 4) + 511) / 512;
  constexpr int NG = 8;                                  // groups of 2 taps per parity: 15 -> 8, 12 -> 6
  static_assert(TH_ * TW_ == 128, "a block owns 128 positions of each column parity");
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* At = smb;                               // [IHv][GWv] pixels
  unsigned char* Bt = smb + IHv * GWv * PB;              // [2 buffers][TG taps][32 rows]
  // (round 5) epilogue turn: a wave's 32 positions x 32 channels leave through a private 4 KB patch as 16-byte
  // row segments -- 4 stores (and 4 mask loads) per lane and tile instead of 16 four-byte ones
  float* turn = reinterpret_cast<float*>(Bt + 2 * TG * WBB) + (threadIdx.x >> 6) * (32 * 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int pg = wave & 3, E = wave >> 2;                // E = column parity of this wave half
  const int c4 = tid & 7;
  const int NU = E ? 4 : 5, NT = KH * NU;
  const int Wp = (d.Win + 1 - E) / 2;                    // input columns of this parity
  const int px0 = tid >> 3;
  auto tile_pos = [&](int tile, int& sq, int& h0, int& m0) {
    const int tw = tile % tiles_w, rest = tile / tiles_w;
    const int th = rest % tiles_h;
    sq = rest / tiles_h;
    h0 = th * TH_;
    m0 = tw * TW_;
  };
  auto load_patch = [&](int tile, f32x4 (&pf)[NCHK]) {
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
    const float* org = d.x + (long long)sq * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)(m0 - 2) * C + c4 * 4;
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      const int r = div_small<GWv>(px), xc = px - r * GWv;
      const int h = h0 - 1 + r, c = m0 - 2 + xc;
      const bool ok = px < IHv * GWv && h >= 0 && h < d.H && c >= 0 && c < d.Wout;
      pf[q] = *reinterpret_cast<const f32x4*>(ok ? org + (long long)r * d.x_line + xc * C : c6_zero);
    }
  };
  auto store_patch = [&](const f32x4 (&pf)[NCHK]) {
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      if (px < IHv * GWv) store_px3(At + px * PB + c4 * 8, pf[q]);
    }
  };
  // weights of group g: taps 2g, 2g+1 of parity 0 (slots 0, 1) and of parity 1 (slots 2, 3); tap index
  // ti of parity e -> weight tile (ti / nu) * 9 + e + 2 * (ti % nu); image rows = (tile, ci), 192 bytes each
  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(d.w);
  int wsrc[3], wdst[3], wslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = tid + 512 * k;
    const int u = c / WCH, rem = c - u * WCH;
    const int row = rem / 12, part = rem - row * 12;
    wslot[k] = u;
    wsrc[k] = row * 192 + part * 16;
    wdst[k] = u * WBB + row * PB + part * 16;
  }
  auto load_w = [&](int g, u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = wslot[k] >> 1, nu = e ? 4 : 5, nt = KH * nu;
      int ti = 2 * g + (wslot[k] & 1);
      ti = ti < nt ? ti : nt - 1;
      const int dh = e ? ti >> 2 : (ti * 13) >> 6;        // ti / nu for ti < 15 without a division
      const int tile_w = dh * KW + e + 2 * (ti - dh * nu);
      wn[k] = *reinterpret_cast<const u32x4*>(wimg + (long long)tile_w * (C * 192) + wsrc[k]);
    }
  };
  auto store_w = [&](int buf, const u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(Bt + buf * (TG * WBB) + wdst[k]) = wn[k];
  };
  int ph, pw;
  px_of_row<TW_>(pg, li, ph, pw);
  const unsigned char* Ap = At + ((ph + 2) * GWv + pw + 4) * PB + hh * 16;
  const unsigned char* Bp = Bt + (E * 2) * WBB + li * PB + hh * 16;
  const bool msk = d.mask_src != nullptr, fm = d.fm_ref != nullptr;
  const float fmw = fm ? d.fm_w * (d.fm_wdev ? d.fm_wdev[0] : 1.f) : 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};            // column sums of what this lane stores (channels 4 (lane & 7) ..)

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  u32x4 wn[3];                                  // weight pipeline as in the forward kernel
  {
    f32x4 pf[NCHK];
    load_patch(tile, pf);
    load_w(0, wn);
    store_patch(pf);
    store_w(0, wn);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    f32x4 pf[NCHK];
    load_patch(more ? nxt : tile, pf);
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
    // (round 5) the mask of this tile's outputs is requested HERE, four 16-byte segments per lane (item j =
    // position (lane >> 3) + 8 j of the wave's 32, channels 4 (lane & 7) ..): read in the epilogue it put a
    // memory round trip behind every tile's MFMAs with nothing else resident on the CU -- which is why the
    // fused leaky-ReLU backward used to lose against a separate pass over the map
    long long ioff[4];
    f32x4 my[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int qh, qw;
      px_of_row<TW_>(pg, (lane >> 3) + 8 * j, qh, qw);
      const int oh = h0 + qh, om = m0 + qw;
      ioff[j] = (oh < d.H && om < Wp)
                    ? (long long)sq * d.y_seq + (long long)oh * d.y_line + (long long)(2 * om + E) * C + 4 * (lane & 7)
                    : -1;
      // (a clamped offset, not a pointer select against a zero block: that costs a GOT load + wait per request)
      if (msk && !fm) my[j] = *reinterpret_cast<const f32x4*>(d.mask_src + (ioff[j] >= 0 ? ioff[j] : 0));
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    // patch offset of this parity's tap ti (a literal after unrolling; E is wave-uniform)
#define F2G_C6_GOFF(TI) ((E ? ((TI) / 4) * GWv + (TI) % 4 : ((TI) / 5) * GWv + (TI) % 5) * PB)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      load_w(g + 1 < NG ? g + 1 : 0, wn);
      const int buf = g & 1;                    // (NG is even: group 0 is always in buffer 0)
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
        if (2 * g + u2 < NT) tap6(Ap - F2G_C6_GOFF(2 * g + u2), Bp + (buf * TG + u2) * WBB, acc0, acc1);
      store_w(buf ^ 1, wn);
      __syncthreads();
    }
#undef F2G_C6_GOFF
    // ---- epilogue: the wave's tile through its private patch ([position][channel]); optional leaky-ReLU
    // backward of the layer below (+ feature-matching term), column sums
#pragma unroll
    for (int q = 0; q < 16; ++q) turn[((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + li] = acc0[q] + acc1[q];
    __builtin_amdgcn_wave_barrier();
    // three separate paths, so that the plain and the prefetched-mask path carry no wait of the path that
    // loads its mask / reference values here (a merged loop waited vmcnt(0) -- i.e. for the previous item's
    // STORE -- in front of every item: 10.5 instead of 6.4 ms over the 45 masked launches of a pass)
    auto item = [&](int j) {
      return *reinterpret_cast<const f32x4*>(turn + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7));
    };
    if (!msk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 u = item(j);
        if (ioff[j] < 0) continue;
        cs[0] += u.x, cs[1] += u.y, cs[2] += u.z, cs[3] += u.w;
        *reinterpret_cast<f32x4*>(d.y + ioff[j]) = u;
      }
    } else if (!fm) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 u = item(j);
        if (ioff[j] < 0) continue;
        f32x4 v;
        v.x = u.x * (my[j].x > 0.f ? 1.f : d.mask_slope);
        v.y = u.y * (my[j].y > 0.f ? 1.f : d.mask_slope);
        v.z = u.z * (my[j].z > 0.f ? 1.f : d.mask_slope);
        v.w = u.w * (my[j].w > 0.f ? 1.f : d.mask_slope);
        cs[0] += v.x, cs[1] += v.y, cs[2] += v.z, cs[3] += v.w;
        *reinterpret_cast<f32x4*>(d.y + ioff[j]) = v;
      }
    } else {
      f32x4 yv[4], fv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long o = ioff[j] >= 0 ? ioff[j] : 0;
        yv[j] = *reinterpret_cast<const f32x4*>(d.mask_src + o);
        fv[j] = *reinterpret_cast<const f32x4*>(d.fm_ref + o);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 u = item(j);
        if (ioff[j] < 0) continue;
        float v[4] = {u.x, u.y, u.z, u.w};
        const float y[4] = {yv[j].x, yv[j].y, yv[j].z, yv[j].w}, f[4] = {fv[j].x, fv[j].y, fv[j].z, fv[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = y[e] - f[e];
          v[e] += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
          v[e] *= y[e] > 0.f ? 1.f : d.mask_slope;
          cs[e] += v[e];
        }
        *reinterpret_cast<f32x4*>(d.y + ioff[j]) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (more) store_patch(pf);
    __syncthreads();
  }
  if (d.colsum) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 8) atomicAdd(d.colsum + 4 * lane + e, v);
    }
  }
}

// The masked instance (mask_src set: the D-step's fused leaky-ReLU backward): the narrow epilogue of round 4
// with the tile's sixteen mask values per lane requested before the MFMAs.  The LDS-turned epilogue above is
// 3 % faster without a mask and 30 % SLOWER with one (10.5 : 8.0 ms over the 45 masked launches of a pass,
// whatever the form of its loads and waits -- profiles/r05_conv32_dgrad_mask.txt), so both stay.
template <int TH_, int TW_>
__global__ __launch_bounds__(512, 1) void conv32_s2_dgrad6m_kernel(const f2g_conv32_desc d, int tiles_w,
                                                                 int tiles_h, int ntiles) {
  constexpr int IHv = TH_ + KH - 1, GWv = TW_ + 4;
  constexpr int NCHK = (IHv * GWv * (C / 4) + 511) / 512;
  constexpr int NG = 8;                                  // groups of 2 taps per parity: 15 -> 8, 12 -> 6
  static_assert(TH_ * TW_ == 128, "a block owns 128 positions of each column parity");
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* At = smb;                               // [IHv][GWv] pixels
  unsigned char* Bt = smb + IHv * GWv * PB;              // [2 buffers][TG taps][32 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int pg = wave & 3, E = wave >> 2;                // E = column parity of this wave half
  const int c4 = tid & 7;
  const int NU = E ? 4 : 5, NT = KH * NU;
  const int Wp = (d.Win + 1 - E) / 2;                    // input columns of this parity
  const int px0 = tid >> 3;
  auto tile_pos = [&](int tile, int& sq, int& h0, int& m0) {
    const int tw = tile % tiles_w, rest = tile / tiles_w;
    const int th = rest % tiles_h;
    sq = rest / tiles_h;
    h0 = th * TH_;
    m0 = tw * TW_;
  };
  auto load_patch = [&](int tile, f32x4 (&pf)[NCHK]) {
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
    const float* org = d.x + (long long)sq * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)(m0 - 2) * C + c4 * 4;
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      const int r = div_small<GWv>(px), xc = px - r * GWv;
      const int h = h0 - 1 + r, c = m0 - 2 + xc;
      const bool ok = px < IHv * GWv && h >= 0 && h < d.H && c >= 0 && c < d.Wout;
      pf[q] = *reinterpret_cast<const f32x4*>(ok ? org + (long long)r * d.x_line + xc * C : c6_zero);
    }
  };
  auto store_patch = [&](const f32x4 (&pf)[NCHK]) {
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      const int px = px0 + 64 * q;
      if (px < IHv * GWv) store_px3(At + px * PB + c4 * 8, pf[q]);
    }
  };
  // weights of group g: taps 2g, 2g+1 of parity 0 (slots 0, 1) and of parity 1 (slots 2, 3); tap index
  // ti of parity e -> weight tile (ti / nu) * 9 + e + 2 * (ti % nu); image rows = (tile, ci), 192 bytes each
  const unsigned char* wimg = reinterpret_cast<const unsigned char*>(d.w);
  int wsrc[3], wdst[3], wslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = tid + 512 * k;
    const int u = c / WCH, rem = c - u * WCH;
    const int row = rem / 12, part = rem - row * 12;
    wslot[k] = u;
    wsrc[k] = row * 192 + part * 16;
    wdst[k] = u * WBB + row * PB + part * 16;
  }
  auto load_w = [&](int g, u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = wslot[k] >> 1, nu = e ? 4 : 5, nt = KH * nu;
      int ti = 2 * g + (wslot[k] & 1);
      ti = ti < nt ? ti : nt - 1;
      const int dh = e ? ti >> 2 : (ti * 13) >> 6;        // ti / nu for ti < 15 without a division
      const int tile_w = dh * KW + e + 2 * (ti - dh * nu);
      wn[k] = *reinterpret_cast<const u32x4*>(wimg + (long long)tile_w * (C * 192) + wsrc[k]);
    }
  };
  auto store_w = [&](int buf, const u32x4 (&wn)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(Bt + buf * (TG * WBB) + wdst[k]) = wn[k];
  };
  int ph, pw;
  px_of_row<TW_>(pg, li, ph, pw);
  const unsigned char* Ap = At + ((ph + 2) * GWv + pw + 4) * PB + hh * 16;
  const unsigned char* Bp = Bt + (E * 2) * WBB + li * PB + hh * 16;
  const bool msk = d.mask_src != nullptr, fm = d.fm_ref != nullptr;
  const float fmw = fm ? d.fm_w * (d.fm_wdev ? d.fm_wdev[0] : 1.f) : 0.f;
  float cs = 0.f;

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  u32x4 wn[3];                                  // weight pipeline as in the forward kernel
  {
    f32x4 pf[NCHK];
    load_patch(tile, pf);
    load_w(0, wn);
    store_patch(pf);
    store_w(0, wn);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    f32x4 pf[NCHK];
    load_patch(more ? nxt : tile, pf);
    int sq, h0, m0;
    tile_pos(tile, sq, h0, m0);
    // (round 5) the mask of this tile's outputs is requested HERE, sixteen values per lane: read in the
    // epilogue it put a memory round trip behind every tile's MFMAs with nothing else resident on the CU --
    // which is why the fused leaky-ReLU backward used to lose against a separate pass over the map
    float my[16];
    if (msk && !fm) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        int qh, qw;
        px_of_row<TW_>(pg, (q & 3) + 8 * (q >> 2) + 4 * hh, qh, qw);
        const int oh = h0 + qh, om = m0 + qw;
        const long long off = (long long)sq * d.y_seq + (long long)oh * d.y_line + (long long)(2 * om + E) * C + li;
        my[q] = (oh < d.H && om < Wp) ? d.mask_src[off] : 1.f;
      }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    // patch offset of this parity's tap ti (a literal after unrolling; E is wave-uniform)
#define F2G_C6_GOFF(TI) ((E ? ((TI) / 4) * GWv + (TI) % 4 : ((TI) / 5) * GWv + (TI) % 5) * PB)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      load_w(g + 1 < NG ? g + 1 : 0, wn);
      const int buf = g & 1;                    // (NG is even: group 0 is always in buffer 0)
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
        if (2 * g + u2 < NT) tap6(Ap - F2G_C6_GOFF(2 * g + u2), Bp + (buf * TG + u2) * WBB, acc0, acc1);
      store_w(buf ^ 1, wn);
      __syncthreads();
    }
#undef F2G_C6_GOFF
    // ---- epilogue: optional leaky-ReLU backward of the layer below (+ feature-matching term)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int qh, qw;
      px_of_row<TW_>(pg, (q & 3) + 8 * (q >> 2) + 4 * hh, qh, qw);
      const int oh = h0 + qh, om = m0 + qw;
      if (oh < d.H && om < Wp) {
        const long long off = (long long)sq * d.y_seq + (long long)oh * d.y_line + (long long)(2 * om + E) * C + li;
        float v = acc0[q] + acc1[q];
        if (msk) {
          const float y = fm ? d.mask_src[off] : my[q];
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
    if (more) store_patch(pf);
    __syncthreads();
  }
  if (d.colsum) {
    cs += __shfl_xor(cs, 32);
    if (hh == 0) atomicAdd(d.colsum + li, cs);
  }
}

// ---- weight gradient -----------------------------------------------------------------------------------
// gw[co][tap][ci] += sum_px g[px][co] * x[px -> tap][ci]: MFMA rows = co, columns = ci, reduction = the
// pixels of an 8 x 16 output tile (conv32.hip, conv32_s2_wgrad3_kernel: a bf16 MFMA wants 8 consecutive k
// per lane and k = pixels is the slow axis of both staged operands, so the tile is staged as bf16 planes
// with 64-byte pixels and the fragments come from ds_read_b64_tr_b16).  Here THREE planes per operand
// (the pieces p0, p1, p2, split once while the tile is staged) and six MFMAs per product; wave w owns taps
// w, w + 8, w + 16 over the whole tile and tile row w of taps 24..26.  77 + 25 KB of planes = one block per
// CU, so the block requests the next tile's rows into registers before it computes the current one and
// splits them chunk by chunk under the MFMAs of the tile's rows.
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int GPL = 128 * 64;            // bytes of one plane of the gradient tile
constexpr int GCH = 128 * (C / 4) / 512; // gradient chunks per thread: 2

__device__ __forceinline__ bf16x8 tr_pix8(const unsigned char* p) {   // pixels +0..3 and +4..7
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * 64));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int TH, int TW>
__global__ __launch_bounds__(512, 1) void conv32_s2_wgrad6_kernel(const f2g_conv32_desc d, float* gw,
                                                                  int tiles_h, int tiles_w,
                                                                  int tiles_per_block) {
  // tiles of 8 x 16 or 16 x 8 output pixels; a k step = 16 pixels = one tile row or two
  constexpr int IH = TH + KH - 1, IW = TW + (KW - 1) / 2, XW = 2 * IW - 1;
  constexpr int XPAR = IH * IW * 64 + 64;  // bytes of one column parity of a plane (+64: pixels x and x + 1 of the
                                           // 8-byte staging stores would otherwise share every bank)
  constexpr int XPL = 2 * XPAR;            // bytes of one plane of the input patch
  constexpr int XCH = (IH * XW * (C / 4) + 511) / 512;   // patch chunks per thread: 7
  constexpr int KR = 16 / TW;              // tile rows per k step
  constexpr int NK = TH / KR;              // k steps per tile: 8
  static_assert(TH * TW == 128 && NK == 8 && XCH <= NK && GCH <= NK, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* Xp = smb;                 // [3 pieces][2 parities][IH][IW] pixels x 32 bf16
  unsigned char* Gp = smb + 3 * XPL;       // [3 pieces][128 px] x 32 bf16
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
  // pixel of the first read inside a k step: lane halves take pixels 0-7 / 8-15 of it (the next tile row when
  // rows are 8 pixels wide); kpix = offset in the dense gradient tile, kpat = in the staged patch
  const int kpix = (g4 >> 1) * 8 + (i16 >> 2);
  const int kpat = TW == 16 ? kpix : (g4 >> 1) * IW + (i16 >> 2);
  const int chb = (g4 & 1) * 32 + (i16 & 3) * 8;           // byte offset of this lane's 4 channels
  int xo[3], xs[3];                                        // patch offsets of the owned / shared taps
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int t = wave + 8 * a;
    int dh = t / KW, j = t - dh * KW;
    xo[a] = (j & 1) * XPAR + (dh * IW + (j >> 1) + kpat) * 64 + chb;
    t = 24 + a;
    dh = t / KW;
    j = t - dh * KW;
    xs[a] = (j & 1) * XPAR + (dh * IW + (j >> 1) + kpat) * 64 + chb;
  }
  const int go = kpix * 64 + chb;
  // chunk q of this thread = 4 channels of patch pixel px0 + 64 q = (row r, column xr); plane offset and
  // source offset are recomputed per tile (a multiplication) instead of kept in 21 registers
  const int c4 = tid & 7, px0 = tid >> 3;
  auto xoff = [&](int q) {                 // byte offset inside a plane, or -1
    const int px = px0 + 64 * q;
    const int r = div_small<XW>(px), xr = px - r * XW;
    return px < IH * XW ? (xr & 1) * XPAR + (r * IW + (xr >> 1)) * 64 + c4 * 8 : -1;
  };
  auto tile_pos = [&](int ti, int& s, int& h0, int& w0) {
    s = ti / (tiles_h * tiles_w);
    const int rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w;
    h0 = th * TH;
    w0 = (rem - th * tiles_w) * TW;
  };
  auto load_tile = [&](int ti, f32x4 (&px_)[XCH], f32x4 (&pg_)[GCH]) {
    int s, h0, w0;
    tile_pos(ti, s, h0, w0);
    const int x0 = 2 * w0 - (KW - 1) / 2;
    const float* org = d.x + (long long)s * d.x_seq + (long long)(h0 - 1) * d.x_line + (long long)x0 * C + c4 * 4;
#pragma unroll
    for (int q = 0; q < XCH; ++q) {
      const int px = px0 + 64 * q;
      const int r = div_small<XW>(px), xr = px - r * XW;
      const int h = h0 - 1 + r, x = x0 + xr;
      const bool ok = px < IH * XW && h >= 0 && h < d.H && x >= 0 && x < d.Win;
      px_[q] = *reinterpret_cast<const f32x4*>(ok ? org + (long long)r * d.x_line + xr * C : c6_zero);
    }
    const float* gs = d.y + (long long)s * d.y_seq;
#pragma unroll
    for (int q = 0; q < GCH; ++q) {
      const int px = (tid >> 3) + 64 * q;
      const int h = h0 + px / TW, w = w0 + px % TW;
      const bool ok = h < d.H && w < d.Wout;
      pg_[q] = *reinterpret_cast<const f32x4*>(ok ? gs + (long long)h * d.y_line + (long long)w * C + c4 * 4 : c6_zero);
    }
  };
  auto put3 = [&](unsigned char* base, int plane_bytes, int off, const u32x2 (&pk)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(base + q * plane_bytes + off) = pk[q];
  };
  auto mfma6 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  };
  const int t0 = blockIdx.x * tiles_per_block;
  int tend = t0 + tiles_per_block;
  if (tend > ntiles) tend = ntiles;
  if (t0 >= tend) return;
  if (tid < IH * 8) {   // the odd parity has one column less: keep its last column defined (never rewritten)
    const int r = tid >> 3;
    const int off = XPAR + (r * IW + IW - 1) * 64 + c4 * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(Xp + q * XPL + off) = u32x2{0u, 0u};
  }
  {
    f32x4 px_[XCH], pg_[GCH];
    load_tile(t0, px_, pg_);
#pragma unroll
    for (int q = 0; q < XCH; ++q)
      if (xoff(q) >= 0) {
        u32x2 pk[3];
        split3(px_[q], pk[0], pk[1], pk[2]);
        put3(Xp, XPL, xoff(q), pk);
      }
#pragma unroll
    for (int q = 0; q < GCH; ++q) {
      u32x2 pk[3];
      split3(pg_[q], pk[0], pk[1], pk[2]);
      put3(Gp, GPL, ((tid >> 3) + 64 * q) * 64 + c4 * 8, pk);
    }
  }
  __syncthreads();
  for (int ti = t0; ti < tend; ++ti) {
    const bool more = ti + 1 < tend;
    f32x4 px_[XCH], pg_[GCH];
    u32x2 kx[XCH][3], kg[GCH][3];
    load_tile(more ? ti + 1 : ti, px_, pg_);
    // owned taps: the 8 tile rows = 8 k steps of 16 pixels; one staged chunk of the next tile is split per row
#pragma unroll
    for (int row = 0; row < NK; ++row) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = tr_pix8(Gp + q * GPL + go + row * 16 * 64);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = tr_pix8(Xp + q * XPL + xo[t] + row * KR * IW * 64);
        mfma6(a, b, acc[t]);
      }
      if (row < XCH) split3(px_[row < XCH ? row : 0], kx[row < XCH ? row : 0][0], kx[row < XCH ? row : 0][1], kx[row < XCH ? row : 0][2]);
      if (row >= NK - GCH)
        split3(pg_[row >= NK - GCH ? row - (NK - GCH) : 0], kg[row >= NK - GCH ? row - (NK - GCH) : 0][0],
               kg[row >= NK - GCH ? row - (NK - GCH) : 0][1], kg[row >= NK - GCH ? row - (NK - GCH) : 0][2]);
    }
    // taps 24..26: this wave's k step (tile row `wave`, or rows 2 wave and 2 wave + 1)
    {
      const int row = wave;
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = tr_pix8(Gp + q * GPL + go + row * 16 * 64);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = tr_pix8(Xp + q * XPL + xs[t] + row * KR * IW * 64);
        mfma6(a, b, acc[3 + t]);
      }
    }
    __syncthreads();   // every wave is done with this tile's planes
    if (more) {
#pragma unroll
      for (int q = 0; q < XCH; ++q)
        if (xoff(q) >= 0) put3(Xp, XPL, xoff(q), kx[q]);
#pragma unroll
      for (int q = 0; q < GCH; ++q) put3(Gp, GPL, ((tid >> 3) + 64 * q) * 64 + c4 * 8, kg[q]);
    }
    __syncthreads();
  }
  // ---- flush: taps owned by one wave go straight out; the three shared taps are summed over the
  // waves through LDS first
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = wave + 8 * q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
      atomicAdd(gw + co * (NTAP * C) + t * C + li, acc[q][e]);
    }
  }
  float* red = reinterpret_cast<float*>(smb);   // [8 waves][16][64]
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
        atomicAdd(gw + co * (NTAP * C) + (24 + q) * C + li, v);
      }
    }
  }
}

// ---- the fifth layer of a band stack: Conv2d(32, 32, (3, 3), padding (1, 1)), stride 1 (round 5) -------
// (discriminators.py:171-181, last entry of the band stack.)  As an implicit GEMM this is M = 10^5 pixels x
// N = 32 x K = 288 on the generic kernel's 128 x 32 tiles with bounds-tested windows: 51 TFLOP/s forward, 29
// in the weight gradient (profiles/r05_*).  Direct, fp32 class: the images are narrow (7 ... 33 columns after
// three stride-2 layers), so a tile is R WHOLE rows (R * W <= 256 pixels, >= 90 % of the MFMA rows used at
// every band width) staged with one zero column on either side and one row above / below -- the nine taps are
// then nine constant offsets into the staged patch, with no per-tap masks.  One persistent block of 8 waves per
// CU: all nine weight tiles (55 KB image, 208-byte rows) stay in LDS for the block's life, the next tile's
// patch is requested before the current tile's 108 MFMAs per wave and split into its three pieces behind them.
// The data gradient of a stride-1 conv is the same kernel over the gradient map with the taps flipped and the
// channel matrix transposed (the host re-lays the weights once per version).
constexpr int P33 = 352;                 // staged pixels of a tile: (R + 2) * (W + 2) <= P33
constexpr int T33 = 9;

__global__ __launch_bounds__(512, 1) void conv33_x6_kernel(const f2g_conv32_desc d, int R, int tiles_h, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* At = smb;                       // [P33] staged pixels
  unsigned char* Bt = smb + P33 * PB;            // [9 taps][32 co] weight rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int W = d.Win, PW = W + 2, npx = (R + 2) * PW, tpx = R * W;
  const unsigned mgW = magic_of(W), mgPW = magic_of(PW);
  // weights: image [co][9 taps][192 bytes] -> LDS rows (tap, co) of 208 bytes
  {
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(d.w);
    for (int c = tid; c < T33 * C * 12; c += 512) {
      const int row = c / 12, part = c - row * 12;           // row = co * 9 + tap in the image
      const int co = row / T33, tap = row - co * T33;
      *reinterpret_cast<u32x4*>(Bt + (tap * C + co) * PB + part * 16) =
          *reinterpret_cast<const u32x4*>(wimg + row * 192 + part * 16);
    }
  }
  // this thread's chunks of a patch: chunk id = tid + 512 q -> (staged pixel, 4 channels)
  constexpr int NQ = (P33 * 8 + 511) / 512;
  int prow[NQ], poff[NQ], pdst[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int id = tid + 512 * q, px = id >> 3, c4 = id & 7;
    const int pr = fast_div(px, PW, mgPW), pc = px - pr * PW;
    const bool ok = px < npx && pc >= 1 && pc <= W;
    prow[q] = ok ? pr : -(1 << 20);                          // (columns of the zero border: never valid)
    poff[q] = (pr - 1) * (int)d.x_line + (pc - 1) * C + c4 * 4;
    pdst[q] = px < npx ? px * PB + c4 * 8 : -1;
  }
  auto load_patch = [&](int tile, f32x4 (&pf)[NQ]) {
    const int sq = tile / tiles_h, h0 = (tile - sq * tiles_h) * R;
    const float* org = d.x + (long long)sq * d.x_seq + (long long)h0 * d.x_line;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int h = h0 - 1 + prow[q];
      pf[q] = *reinterpret_cast<const f32x4*>((h >= 0 && h < d.H) ? org + poff[q] : c6_zero);
    }
  };
  auto store_patch = [&](const f32x4 (&pf)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (pdst[q] >= 0) store_px3(At + pdst[q], pf[q]);
  };
  // this lane's output pixel: tile pixel 32 wave + li -> (row, column) -> staged position
  const int mypx = wave * 32 + li;
  const int myr = fast_div(mypx, W, mgW), myc = mypx - myr * W;
  const unsigned char* Ap = At + (mypx < tpx ? ((myr + 1) * PW + myc + 1) * PB : (PW + 1) * PB) + hh * 16;
  const unsigned char* Bp = Bt + li * PB + hh * 16;
  const float bias = d.bias ? d.bias[li] : 0.f;
  float cs = 0.f;                                // column sums of what this lane stores (channel li)
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  {
    f32x4 pf[NQ];
    load_patch(tile, pf);
    store_patch(pf);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    f32x4 pf[NQ];
    load_patch(more ? nxt : tile, pf);            // (the last tile re-requests its own: never stored)
    const int sq = tile / tiles_h, h0 = (tile - sq * tiles_h) * R;
    const long long ybase = (long long)sq * d.y_seq + (long long)h0 * d.y_line;
    // data-gradient role: the leaky-ReLU backward mask of the layer below, requested before the MFMAs
    float my[16];
    if (d.mask_src) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int r = fast_div(p, W, mgW), c = p - r * W;
        my[e] = (p < tpx && h0 + r < d.H) ? d.mask_src[ybase + (long long)r * d.y_line + c * C + li] : 1.f;
      }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int t = 0; t < T33; ++t)
      tap6(Ap + ((t / 3 - 1) * PW + (t % 3 - 1)) * PB, Bp + t * (C * PB), acc0, acc1);
    {
      float* ys = d.y + ybase;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int r = fast_div(p, W, mgW), c = p - r * W;
        if (p < tpx && h0 + r < d.H) {
          float v = acc0[e] + acc1[e] + bias;
          if (d.lrelu_slope != 0.f) v = v > 0.f ? v : d.lrelu_slope * v;
          if (d.mask_src) v *= my[e] > 0.f ? 1.f : d.mask_slope;
          cs += v;
          ys[(long long)r * d.y_line + c * C + li] = v;
        }
      }
    }
    __syncthreads();                               // every wave is done with this patch
    if (more) store_patch(pf);
    __syncthreads();
  }
  if (d.colsum) {
    cs += __shfl_xor(cs, 32);
    if (hh == 0) atomicAdd(d.colsum + li, cs);
  }
}


// ---- weight gradient of the (3, 3) fifth layer (round 6) -------------------------------------------------
// gw[co][tap][ci] += sum_px g[px][co] * x[px + tap][ci]   (discriminators.py:171-181 backward; as an implicit
// GEMM on the generic fp32 kernel -- 32 x 288 outputs over 10^5 pixels through bounds-tested windows -- it ran at
// 31 TFLOP/s).  Same tile as conv33_x6_kernel: R whole rows of a 7-112 column band image, staged with a zero
// column on either side and a row above / below, so that in the FLAT order of the staged patch the nine taps are
// nine constant offsets.  The gradient rows are staged in that same padded flat order (zeros in the border
// columns), which makes the reduction a plain walk over flat positions p: gw[tap] += G[p] * X[p + off(tap)], the
// border positions contributing G = 0.  k = pixels is the slow axis of both operands, so they are staged as
// three bf16 planes with 64-byte pixels (split once while staging) and the fragments come from
// ds_read_b64_tr_b16 (conv32_s2_wgrad6_kernel's scheme).  A k step = 16 consecutive flat positions; the block's
// 8 waves take the k steps round-robin and keep all nine 32 x 32 tap tiles (144 accumulator registers) over ALL
// tiles of their persistent block -- 3 gradient + 27 input fragments per 54 MFMAs -- and meet once at the end:
// tap by tap through LDS, one atomic per output and block.
constexpr int W33_GP = 352;      // gradient positions of a tile: R * (W + 2) rounded up to a k step
constexpr int W33_XP = 384;      // input positions: 1 + (R + 2) * (W + 2) + the last k step's overhang
constexpr int W33_GPL = W33_GP * 64, W33_XPL = W33_XP * 64;

__global__ __launch_bounds__(512, 1) void conv33_wgrad6_kernel(const f2g_conv32_desc d, float* gw, int R,
                                                               int tiles_h, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  unsigned char* Xp = smb;                      // [3 pieces][W33_XP] pixels x 32 bf16; position 0 = slack
  unsigned char* Gp = smb + 3 * W33_XPL;        // [3 pieces][W33_GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int W = d.Win, PW = W + 2, npx = (R + 2) * PW, ngp = R * PW, nks = (ngp + 15) >> 4;
  const unsigned mgPW = magic_of(PW);
  // everything a fragment may touch and a tile does not rewrite (slack position, k-step overhang) is zero
  for (int o = tid * 16; o < 3 * (W33_XPL + W33_GPL); o += 512 * 16) *reinterpret_cast<u32x4*>(smb + o) = u32x4{0u, 0u, 0u, 0u};
  // chunk id = tid + 512 q -> (flat position, 4 channels)
  constexpr int NQ = (W33_GP * 8 + 511) / 512;          // 6: covers the patch (<= 352 positions) and the gradient rows
  const int c4 = tid & 7;
  int xrow[NQ], xoff[NQ], grow[NQ], goff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int px = (tid + 512 * q) >> 3;
    {   // input patch position px = (pr + 1) * PW + (pc + 1)
      const int pr = fast_div(px, PW, mgPW), pc = px - pr * PW;
      const bool ok = px < npx && pc >= 1 && pc <= W;
      xrow[q] = px < npx ? (ok ? pr - 1 : -(1 << 20)) : -(1 << 21);     // (< -2^20: border = zeros; < -2^21: not staged)
      xoff[q] = (pr - 1) * (int)d.x_line + (pc - 1) * C + c4 * 4;
    }
    {   // gradient position px = r * PW + (c + 1)
      const int r = fast_div(px, PW, mgPW), pc = px - r * PW;
      const bool ok = px < ngp && pc >= 1 && pc <= W;
      grow[q] = px < ngp ? (ok ? r : -(1 << 20)) : -(1 << 21);
      goff[q] = r * (int)d.y_line + (pc - 1) * C + c4 * 4;
    }
  }
  // transposed-read lane roles (conv32_s2_wgrad6_kernel): 16-lane group g4 = (channel half, k half)
  const int g4 = lane >> 4, i16 = lane & 15;
  const int kpix = (g4 >> 1) * 8 + (i16 >> 2);
  const int chb = (g4 & 1) * 32 + (i16 & 3) * 8;
  const int go = kpix * 64 + chb;
  int xo[T33];
#pragma unroll
  for (int t = 0; t < T33; ++t) xo[t] = (1 + PW + (t / 3 - 1) * PW + (t % 3 - 1) + kpix) * 64 + chb;
  f32x16 acc[T33];
#pragma unroll
  for (int t = 0; t < T33; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  auto mfma6 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  };
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int sq = tile / tiles_h, h0 = (tile - sq * tiles_h) * R;
    const float* xs = d.x + (long long)sq * d.x_seq + (long long)h0 * d.x_line;
    const float* gs = d.y + (long long)sq * d.y_seq + (long long)h0 * d.y_line;
    f32x4 px_[NQ], pg_[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int hx = h0 + xrow[q], hg = h0 + grow[q];
      px_[q] = *reinterpret_cast<const f32x4*>((xrow[q] > -(1 << 20) && hx >= 0 && hx < d.H) ? xs + xoff[q] : c6_zero);
      pg_[q] = *reinterpret_cast<const f32x4*>((grow[q] > -(1 << 20) && hg < d.H) ? gs + goff[q] : c6_zero);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int px = (tid + 512 * q) >> 3;
      u32x2 pk[3];
      if (xrow[q] > -(1 << 21)) {
        split3(px_[q], pk[0], pk[1], pk[2]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x2*>(Xp + pc * W33_XPL + (px + 1) * 64 + c4 * 8) = pk[pc];
      }
      if (grow[q] > -(1 << 21)) {
        split3(pg_[q], pk[0], pk[1], pk[2]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x2*>(Gp + pc * W33_GPL + px * 64 + c4 * 8) = pk[pc];
      }
    }
    __syncthreads();
    for (int ks = wave; ks < nks; ks += 8) {
      bf16x8 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = tr_pix8(Gp + q * W33_GPL + ks * (16 * 64) + go);
#pragma unroll
      for (int t = 0; t < T33; ++t) {
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = tr_pix8(Xp + q * W33_XPL + ks * (16 * 64) + xo[t]);
        mfma6(a, b, acc[t]);
      }
    }
    __syncthreads();   // every wave is done with this tile's planes
  }
  // ---- flush: the waves' partial tap tiles are summed through LDS, one tap at a time
  float* red = reinterpret_cast<float*>(smb);   // [8 waves][16][64]
#pragma unroll
  for (int t = 0; t < T33; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[t][e];
    __syncthreads();
    if (wave == (t & 7)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 16 + e) * 64 + lane];
        const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
        atomicAdd(gw + co * (T33 * C) + t * C + li, v);
      }
    }
  }
}

template <int TH_, int TW_>
constexpr size_t fwd6_smem() {
  return (size_t)2 * ((TH_ + 2) * (TW_ + 4) * PB + 64) + 2 * TG * WBB + 4 * 16 * 64 * sizeof(float);
}
template <int TH_, int TW_>
constexpr size_t dgrad6_smem() {
  return (size_t)(TH_ + 2) * (TW_ + 4) * PB + 2 * TG * WBB + 8 * 32 * 32 * sizeof(float);
}

}  // namespace

// precision-3 launchers, called from conv32.hip's entry points (arguments already checked there)
int f2g_conv32_fwd6_launch(const f2g_conv32_desc* d, hipStream_t st) {
  auto waste = [&](int th, int tw) {
    return (long long)((d->H + th - 1) / th * th) * ((d->Wout + tw - 1) / tw * tw);
  };
  const bool tall = waste(16, 8) < waste(8, 16);
  const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
  const int tiles_h = (d->H + th - 1) / th, tiles_w = (d->Wout + tw - 1) / tw;
  const long long nt = (long long)tiles_h * tiles_w * d->S;
  if (nt >= (1ll << 30) || d->x_line >= (1ll << 24)) return F2G_EINVAL;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd6_kernel<8, 16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fwd6_smem<8, 16>()));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd6_kernel<16, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fwd6_smem<16, 8>()));
    attr = true;
  }
  const int grid = (int)(nt < 256 ? nt : 256);          // one resident block per CU
  if (tall)
    hipLaunchKernelGGL((conv32_s2_fwd6_kernel<16, 8>), dim3(grid), dim3(512), (fwd6_smem<16, 8>()), st, *d,
                       tiles_w, tiles_h, (int)nt);
  else
    hipLaunchKernelGGL((conv32_s2_fwd6_kernel<8, 16>), dim3(grid), dim3(512), (fwd6_smem<8, 16>()), st, *d,
                       tiles_w, tiles_h, (int)nt);
  return f2g_check_launch();
}

int f2g_conv32_dgrad6_launch(const f2g_conv32_desc* d, hipStream_t st) {
  const int Wp0 = (d->Win + 1) / 2;                     // positions m: the even parity has the most
  auto waste = [&](int th, int tw) {
    return (long long)((d->H + th - 1) / th * th) * ((Wp0 + tw - 1) / tw * tw);
  };
  const bool tall = waste(16, 8) < waste(8, 16);
  const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
  const int tiles_h = (d->H + th - 1) / th, tiles_w = (Wp0 + tw - 1) / tw;
  const long long nt = (long long)tiles_h * tiles_w * d->S;
  if (nt >= (1ll << 30) || d->x_line >= (1ll << 24)) return F2G_EINVAL;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad6_kernel<8, 16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dgrad6_smem<8, 16>()));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad6_kernel<16, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dgrad6_smem<16, 8>()));
    attr = true;
  }
  const int grid = (int)(nt < 256 ? nt : 256);
  if (d->mask_src) {      // masked: the narrow epilogue with the prefetched mask (see conv32_s2_dgrad6m_kernel)
    static bool attrm = false;
    if (!attrm) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad6m_kernel<8, 16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dgrad6_smem<8, 16>()));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad6m_kernel<16, 8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dgrad6_smem<16, 8>()));
      attrm = true;
    }
    if (tall)
      hipLaunchKernelGGL((conv32_s2_dgrad6m_kernel<16, 8>), dim3(grid), dim3(512), (dgrad6_smem<16, 8>()), st, *d,
                         tiles_w, tiles_h, (int)nt);
    else
      hipLaunchKernelGGL((conv32_s2_dgrad6m_kernel<8, 16>), dim3(grid), dim3(512), (dgrad6_smem<8, 16>()), st, *d,
                         tiles_w, tiles_h, (int)nt);
    return f2g_check_launch();
  }
  if (tall)
    hipLaunchKernelGGL((conv32_s2_dgrad6_kernel<16, 8>), dim3(grid), dim3(512), (dgrad6_smem<16, 8>()), st, *d,
                       tiles_w, tiles_h, (int)nt);
  else
    hipLaunchKernelGGL((conv32_s2_dgrad6_kernel<8, 16>), dim3(grid), dim3(512), (dgrad6_smem<8, 16>()), st, *d,
                       tiles_w, tiles_h, (int)nt);
  return f2g_check_launch();
}

int f2g_conv32_wgrad6_launch(const f2g_conv32_desc* d, float* gw, hipStream_t st) {
  auto waste = [&](int th, int tw) {
    return (long long)((d->H + th - 1) / th * th) * ((d->Wout + tw - 1) / tw * tw);
  };
  const bool tall = waste(16, 8) < waste(8, 16);
  const int th = tall ? 16 : 8, tw = tall ? 8 : 16;
  const int tiles_h = (d->H + th - 1) / th, tiles_w = (d->Wout + tw - 1) / tw;
  const long long nt = (long long)d->S * tiles_h * tiles_w;
  if (nt >= (1ll << 30) || d->x_line >= (1ll << 24)) return F2G_EINVAL;
  int per = (int)((nt + 255) / 256);        // <= 256 blocks (one per CU): bounds the atomics
  if (per < 1) per = 1;
  constexpr size_t sm8 = (size_t)3 * 2 * ((8 + 2) * (16 + 4) * 64 + 64) + 3 * GPL;
  constexpr size_t sm16 = (size_t)3 * 2 * ((16 + 2) * (8 + 4) * 64 + 64) + 3 * GPL;
  static_assert(sm8 >= (size_t)8 * 16 * 64 * 4 && sm16 >= (size_t)8 * 16 * 64 * 4,
                "the flush reuses the planes as its reduction buffer");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad6_kernel<8, 16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_wgrad6_kernel<16, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16);
    attr = true;
  }
  const unsigned grid = (unsigned)((nt + per - 1) / per);
  if (tall)
    hipLaunchKernelGGL((conv32_s2_wgrad6_kernel<16, 8>), dim3(grid), dim3(512), sm16, st, *d, gw, tiles_h, tiles_w, per);
  else
    hipLaunchKernelGGL((conv32_s2_wgrad6_kernel<8, 16>), dim3(grid), dim3(512), sm8, st, *d, gw, tiles_h, tiles_w, per);
  return f2g_check_launch();
}

// Conv2d(32, 32, (3, 3), padding (1, 1)) + bias + leaky ReLU, fp32 class (include/flow2gan_hip.h)
extern "C" int f2g_conv33_fwd(const f2g_conv32_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y || d->precision != 3 || d->Win != d->Wout || d->fm_ref) return F2G_EINVAL;
  if ((((uintptr_t)d->x) & 15) || (((uintptr_t)d->w) & 15) || (d->x_line & 3) || (d->x_seq & 3)) return F2G_EINVAL;
  if (d->S <= 0 || d->H <= 0 || d->Win <= 0) return F2G_OK;
  const int W = d->Win;
  if (W > 112 || d->x_line >= (1ll << 24)) return F2G_EINVAL;     // 3 * (W + 2) staged pixels <= P33
  // geometry of the output map and of the two roles (forward: bias / slope; data gradient: mask_src + colsum,
  // whose source shares y's geometry by construction -- it is addressed with y's strides)
  if (d->y_line < (long long)W * C || (d->y != d->x && d->y_seq < (long long)d->H * d->y_line)) return F2G_EINVAL;
  if ((d->y_line & 3) || (d->y_seq & 3) || (((uintptr_t)d->y) & 15)) return F2G_EINVAL;
  if (d->mask_src && (d->lrelu_slope != 0.f || d->bias || (((uintptr_t)d->mask_src) & 15))) {
    f2g_set_error("f2g_conv33_fwd: mask_src (data-gradient role) excludes bias / lrelu_slope (forward role)");
    return F2G_EINVAL;
  }
  int R = 256 / W;                                   // whole rows per tile
  if (R > d->H) R = d->H;
  while (R > 1 && (R + 2) * (W + 2) > P33) --R;
  if (R < 1 || (R + 2) * (W + 2) > P33) return F2G_EINVAL;
  const int tiles_h = (d->H + R - 1) / R;
  const long long nt = (long long)tiles_h * d->S;
  if (nt >= (1ll << 30)) return F2G_EINVAL;
  constexpr size_t smem = (size_t)P33 * PB + (size_t)T33 * C * PB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv33_x6_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const int grid = (int)(nt < 256 ? nt : 256);          // one resident block per CU
  hipLaunchKernelGGL(conv33_x6_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream, *d, R, tiles_h, (int)nt);
  return f2g_check_launch();
}

// Weight gradient of that layer, fp32 class: x = the layer's input (S, H, W, 32), y = the gradient of its
// pre-activation (S, H, W, 32; both optionally strided slices of wider maps), gw (32, 9 * 32) [co][tap][ci] +=.
extern "C" int f2g_conv33_wgrad(const f2g_conv32_desc* d, float* gw, f2g_stream_t stream) {
  if (!d || !d->x || !d->y || !gw || d->precision != 3 || d->Win != d->Wout) return F2G_EINVAL;
  if ((((uintptr_t)d->x) & 15) || (((uintptr_t)d->y) & 15) || (d->x_line & 3) || (d->x_seq & 3) || (d->y_line & 3) ||
      (d->y_seq & 3))
    return F2G_EINVAL;
  if (d->S <= 0 || d->H <= 0 || d->Win <= 0) return F2G_OK;
  const int W = d->Win;
  if (W > 112 || d->x_line >= (1ll << 24) || d->y_line >= (1ll << 24) || d->x_line < (long long)W * C ||
      d->y_line < (long long)W * C)
    return F2G_EINVAL;
  int R = 256 / W;                                   // whole rows per tile (conv33_x6_kernel's rule)
  if (R > d->H) R = d->H;
  while (R > 1 && (R + 2) * (W + 2) > P33) --R;
  if (R < 1 || (R + 2) * (W + 2) > P33) return F2G_EINVAL;
  const int tiles_h = (d->H + R - 1) / R;
  const long long nt = (long long)tiles_h * d->S;
  if (nt >= (1ll << 30)) return F2G_EINVAL;
  constexpr size_t smem = (size_t)3 * (W33_XPL + W33_GPL);
  static_assert(smem >= (size_t)8 * 16 * 64 * 4, "the flush reuses the planes as its reduction buffer");
  // (largest input position a fragment touches: (R + 2) * (W + 2) + 17; gradient positions: R * (W + 2) + 15)
  static_assert(P33 + 18 <= W33_XP && P33 <= W33_GP, "plane sizes");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv33_wgrad6_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const int grid = (int)(nt < 256 ? nt : 256);          // one resident block per CU: 256 x 9216 atomics at most
  hipLaunchKernelGGL(conv33_wgrad6_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream, *d, gw, R, tiles_h,
                     (int)nt);
  return f2g_check_launch();
}
