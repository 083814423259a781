// First layer of every band stack of the multi-resolution STFT discriminator: Conv2d(2, 32, (3, 9),
// stride (1, 1), padding (1, 4)) over a frequency band of the complex spectrogram (reference
// discriminators.py:171,195-203), forward, weight gradient and data gradient.
//
// As implicit GEMMs these are K = 54 / N = 2 problems (measured: 18 / 12 / 4 TFLOP/s, 15 ms of a
// 290 ms step for 0.4 % of its FLOPs): the reduction is shorter than one K slab, the 27 taps of a
// pixel overlap those of its neighbours, and the data gradient has two output columns.  Direct
// kernels instead: a block stages the (rows+2) x (32+8) input patch of its tile once in LDS (a few
// hundred floats per channel plane), the forward / weight gradient feed v_mfma_f32_32x32x2_f32
// with ONE tap per instruction (its two k slots are the two input channels), the data gradient is
// a VALU kernel (two outputs per pixel).  All three are bound by the 32-channel side of the layer
// in HBM (128 B per pixel), not by arithmetic.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KH = 3, KW = 9, NTAP = KH * KW, CO = 32;
constexpr int TW = 32;                 // tile columns (= MFMA rows: one pixel per lane)
constexpr int PW = TW + KW - 1;        // staged columns: 40
constexpr int FTH = 8;                 // forward / wgrad tile rows
constexpr int FPH = FTH + KH - 1;      // staged rows: 10

// stage plane[ci][r][c] = x[h0 - 1 + r][w0 - 4 + c][ci] (0 outside the band image)
__device__ __forceinline__ void stage_patch(float* plane, const f2g_conv2ch_desc& d,
                                            const float* xs, int h0, int w0, int rows, int tid,
                                            int nthreads) {
  for (int i = tid; i < rows * PW; i += nthreads) {
    const int r = i / PW, c = i - r * PW;
    const int h = h0 - 1 + r, w = w0 - 4 + c;
    float2 v = make_float2(0.f, 0.f);
    if (h >= 0 && h < d.H && w >= 0 && w < d.W)
      v = *reinterpret_cast<const float2*>(xs + (long long)h * d.x_line + (long long)w * 2);
    plane[i] = v.x;
    plane[rows * PW + i] = v.y;
  }
}

// ---- forward: y[px, co] = lrelu(b[co] + sum_{tap, ci} w[co][tap][ci] x[px + tap][ci]) ----------
__global__ __launch_bounds__(256) void conv2ch_fwd_kernel(const f2g_conv2ch_desc d) {
  __shared__ float plane[2 * FPH * PW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_w = (d.W + TW - 1) / TW;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int s = blockIdx.y;
  const int h0 = th * FTH, w0 = tw * TW;
  stage_patch(plane, d, d.x + (long long)s * d.x_seq, h0, w0, FPH, tid, 256);
  // B fragments of all 27 taps: lane (co = li, k slot = ci = h) holds w[co][tap][ci]
  float bw[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) bw[t] = d.w[li * (NTAP * 2) + t * 2 + h];
  const float bias = d.bias ? d.bias[li] : 0.f;
  __syncthreads();
  const float* pl = plane + h * (FPH * PW);
#pragma unroll
  for (int rr = 0; rr < FTH / 4; ++rr) {
    const int row = wave + 4 * rr;              // tile row of this wave
    if (h0 + row >= d.H) break;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int dh = t / KW, j = t - dh * KW;
      const float a = pl[(row + dh) * PW + li + j];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[t], acc, 0, 0, 0);
    }
    float* yrow = d.y + ((long long)s * d.H + h0 + row) * (long long)d.W * CO;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int col = w0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (col < d.W) {
        float v = acc[e];
        if (d.lrelu_slope != 0.f) v = v > 0.f ? v : d.lrelu_slope * v;
        yrow[(long long)col * CO + li] = v;
      }
    }
  }
}

// ---- forward, persistent over the rows of a column tile (round 4) --------------------------------
// conv2ch_fwd_kernel spends a block on 8 x 32 pixels: 27 weight loads per lane, a patch and a launch for
// 54 MFMAs per wave (PMC: matrix pipe 37 % busy, 2.4 TB/s on the 32-channel side).  Here a block owns a
// column tile of ONE sequence and walks down all its rows, 8 at a time: weights and bias are loaded once,
// the next 10 x 40 patch is requested (2 x 8 bytes per thread) before the current rows' MFMAs and stored
// into the other LDS buffer behind them -- one barrier per 8 rows.
__global__ __launch_bounds__(256) void conv2ch_fwd_p_kernel(const f2g_conv2ch_desc d) {
  __shared__ float plane[2][2 * FPH * PW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int s = blockIdx.y, w0 = blockIdx.x * TW;
  const float* xs = d.x + (long long)s * d.x_seq;
  float bw[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) bw[t] = d.w[li * (NTAP * 2) + t * 2 + h];
  const float bias = d.bias ? d.bias[li] : 0.f;
  // this thread's two patch pixels (row r, column c) of a 10 x 40 patch: i = tid, tid + 256
  constexpr int NP = (FPH * PW + 255) / 256;
  int pr[NP], pc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int i = tid + 256 * q;
    pr[q] = i / PW;
    pc[q] = i - pr[q] * PW;
  }
  auto load_patch = [&](int h0, float2 (&v)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int hh = h0 - 1 + pr[q], ww = w0 - 4 + pc[q];
      const bool ok = tid + 256 * q < FPH * PW && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W;
      v[q] = ok ? *reinterpret_cast<const float2*>(xs + (long long)hh * d.x_line + (long long)ww * 2)
                : make_float2(0.f, 0.f);
    }
  };
  auto store_patch = [&](float* pl, const float2 (&v)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int i = tid + 256 * q;
      if (i < FPH * PW) {
        pl[i] = v[q].x;
        pl[FPH * PW + i] = v[q].y;
      }
    }
  };
  {
    float2 v[NP];
    load_patch(0, v);
    store_patch(plane[0], v);
  }
  __syncthreads();
  int cur = 0;
  for (int h0 = 0; h0 < d.H; h0 += FTH) {
    const bool more = h0 + FTH < d.H;
    float2 nv[NP];
    load_patch(more ? h0 + FTH : h0, nv);
    const float* pl = plane[cur] + h * (FPH * PW);
#pragma unroll
    for (int rr = 0; rr < FTH / 4; ++rr) {
      const int row = wave + 4 * rr;              // tile row of this wave
      if (h0 + row < d.H) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int dh = t / KW, j = t - dh * KW;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pl[(row + dh) * PW + li + j], bw[t], acc, 0, 0, 0);
        }
        float* yrow = d.y + ((long long)s * d.H + h0 + row) * (long long)d.W * CO;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = w0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (col < d.W) {
            float v = acc[e];
            if (d.lrelu_slope != 0.f) v = v > 0.f ? v : d.lrelu_slope * v;
            yrow[(long long)col * CO + li] = v;
          }
        }
      }
    }
    if (more) store_patch(plane[cur ^ 1], nv);   // (its last readers left through the previous barrier)
    __syncthreads();
    cur ^= 1;
  }
}

// ---- weight gradient: gw[co][tap*2+ci] += sum_px g[px][co] x[px + tap][ci] -----------------------
// MFMA rows = co, columns = the 54 (tap, ci) pairs (two 32-column tiles), reduction = pixels: the A
// fragment (g[px][co]) comes from the gradient tile staged in LDS with 16-byte loads (read as single
// floats straight from global memory it cost a 4-byte load per lane and MFMA pair: 224 us per launch
// for a 197 MB map), the B fragment is gathered from the staged patch.  A block walks
// `tiles_per_block` tiles and leaves with one atomic per output element.
__global__ __launch_bounds__(256) void conv2ch_wgrad_kernel(const f2g_conv2ch_desc d, int tiles_h,
                                                            int tiles_w, int tiles_per_block) {
  __shared__ float plane[2 * FPH * PW];
  __shared__ __attribute__((aligned(16))) float gt[FTH * TW * CO];   // gradient tile [row][col][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int ntiles = d.S * tiles_h * tiles_w;
  // this lane's two gather columns n = li, li + 32 -> (tap, ci) -> offset inside the patch
  int boff[2];
  bool bok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = li + 32 * nt;
    bok[nt] = n < NTAP * 2;
    const int t = bok[nt] ? n >> 1 : 0, ci = n & 1;
    const int dh = t / KW, j = t - dh * KW;
    boff[nt] = ci * (FPH * PW) + dh * PW + j;
  }
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
  const int t0 = blockIdx.x * tiles_per_block;
  for (int ti = t0; ti < t0 + tiles_per_block && ti < ntiles; ++ti) {
    const int s = ti / (tiles_h * tiles_w), rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * FTH, w0 = tw * TW;
    __syncthreads();   // the previous tile's readers are done
    stage_patch(plane, d, d.x + (long long)s * d.x_seq, h0, w0, FPH, tid, 256);
    for (int i = tid; i < FTH * TW * (CO / 4); i += 256) {
      const int c4 = i & 7, px = i >> 3;
      const int row = px / TW, col = px - row * TW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h0 + row < d.H && w0 + col < d.W)
        v = *reinterpret_cast<const float4*>(d.y + (((long long)s * d.H + h0 + row) * (long long)d.W + w0 + col) * CO + c4 * 4);
      *reinterpret_cast<float4*>(gt + px * CO + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < FTH / 4; ++rr) {
      const int row = wave + 4 * rr;
      if (h0 + row >= d.H) break;
      const float* grow = gt + row * TW * CO + li;
#pragma unroll
      for (int st = 0; st < TW / 2; ++st) {
        const int col = 2 * st + h;                  // k slot h of step st = tile column
        const float a = grow[col * CO];
        const float b0 = bok[0] ? plane[boff[0] + row * PW + col] : 0.f;
        const float b1 = bok[1] ? plane[boff[1] + row * PW + col] : 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = li + 32 * nt;
    if (n >= NTAP * 2) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      atomicAdd(d.gw + co * (NTAP * 2) + n, acc[nt][e]);
    }
  }
}

// The same with the NEXT tile's gradient rows (8 x 16 bytes per thread) and patch requested before the
// current tile's 64 MFMAs per wave and written to LDS after them (round 4: conv2ch_wgrad_kernel's blocks
// wait a full memory round trip per tile; PMC: matrix pipe 21 % busy, 1.2 TB/s on the gradient map).
__global__ __launch_bounds__(256) void conv2ch_wgrad_p_kernel(const f2g_conv2ch_desc d, int tiles_h,
                                                              int tiles_w, int tiles_per_block) {
  __shared__ float plane[2 * FPH * PW];
  __shared__ __attribute__((aligned(16))) float gt[FTH * TW * CO];   // gradient tile [row][col][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int ntiles = d.S * tiles_h * tiles_w;
  int boff[2];
  bool bok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = li + 32 * nt;
    bok[nt] = n < NTAP * 2;
    const int t = bok[nt] ? n >> 1 : 0, ci = n & 1;
    const int dh = t / KW, j = t - dh * KW;
    boff[nt] = ci * (FPH * PW) + dh * PW + j;
  }
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
  constexpr int NP = (FPH * PW + 255) / 256, NG = FTH * TW * (CO / 4) / 256;
  const int c4 = tid & 7;
  auto load_tile = [&](int ti, float2 (&pv)[NP], float4 (&gv)[NG]) {
    const int s = ti / (tiles_h * tiles_w), rem = ti - s * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * FTH, w0 = tw * TW;
    const float* xs = d.x + (long long)s * d.x_seq;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int i = tid + 256 * q, r = i / PW, c = i - r * PW;
      const int hh = h0 - 1 + r, ww = w0 - 4 + c;
      const bool ok = i < FPH * PW && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W;
      pv[q] = ok ? *reinterpret_cast<const float2*>(xs + (long long)hh * d.x_line + (long long)ww * 2)
                 : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int px = (tid >> 3) + 32 * q;
      const int row = px / TW, col = px - row * TW;
      const bool ok = h0 + row < d.H && w0 + col < d.W;
      gv[q] = ok ? *reinterpret_cast<const float4*>(d.y + (((long long)s * d.H + h0 + row) * (long long)d.W + w0 + col) * CO + c4 * 4)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](const float2 (&pv)[NP], const float4 (&gv)[NG]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int i = tid + 256 * q;
      if (i < FPH * PW) {
        plane[i] = pv[q].x;
        plane[FPH * PW + i] = pv[q].y;
      }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q)
      *reinterpret_cast<float4*>(gt + ((tid >> 3) + 32 * q) * CO + c4 * 4) = gv[q];
  };
  const int t0 = blockIdx.x * tiles_per_block;
  int tend = t0 + tiles_per_block;
  if (tend > ntiles) tend = ntiles;
  if (t0 >= tend) return;
  {
    float2 pv[NP];
    float4 gv[NG];
    load_tile(t0, pv, gv);
    store_tile(pv, gv);
  }
  __syncthreads();
  for (int ti = t0; ti < tend; ++ti) {
    const bool more = ti + 1 < tend;
    float2 pv[NP];
    float4 gv[NG];
    load_tile(more ? ti + 1 : ti, pv, gv);
    // (rows beyond the image hold zero gradients: no row test needed)
#pragma unroll
    for (int rr = 0; rr < FTH / 4; ++rr) {
      const int row = wave + 4 * rr;
      const float* grow = gt + row * TW * CO + li;
#pragma unroll
      for (int st = 0; st < TW / 2; ++st) {
        const int col = 2 * st + h;                  // k slot h of step st = tile column
        const float a = grow[col * CO];
        const float b0 = bok[0] ? plane[boff[0] + row * PW + col] : 0.f;
        const float b1 = bok[1] ? plane[boff[1] + row * PW + col] : 0.f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) store_tile(pv, gv);
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = li + 32 * nt;
    if (n >= NTAP * 2) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
      atomicAdd(d.gw + co * (NTAP * 2) + n, acc[nt][e]);
    }
  }
}

// ---- data gradient: gx[px][ci] = sum_{tap, co} g[px - tap][co] w[co][tap][ci] ---------------------
// Two outputs per pixel: VALU.  A block stages the gradient patch of its 8 x 32 pixels ((8+2) x
// (32+8) pixels x 32 channels, 36-float pitch) and every thread walks the 27 taps x 32 channels of
// its pixel with float4 LDS reads; the weights are wave-uniform (scalar operands).
constexpr int DTH = 8;
constexpr int DPH = DTH + KH - 1;
constexpr int GP = 36;   // floats per staged pixel

__global__ __launch_bounds__(256) void conv2ch_dgrad_kernel(const f2g_conv2ch_desc d) {
  extern __shared__ __attribute__((aligned(16))) float gp[];     // [DPH][PW][GP]
  const int tid = threadIdx.x;
  const int tiles_w = (d.W + TW - 1) / TW;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int s = blockIdx.y;
  const int h0 = th * DTH, w0 = tw * TW;
  const float* gs = d.y + (long long)s * d.H * d.W * CO;
  // patch pixel (r, c) = g[h0 - 1 + r][w0 - 4 + c]
  for (int i = tid; i < DPH * PW * (CO / 4); i += 256) {
    const int c4 = i & 7, px = i >> 3;
    const int r = px / PW, c = px - r * PW;
    const int hh = h0 - 1 + r, ww = w0 - 4 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hh >= 0 && hh < d.H && ww >= 0 && ww < d.W)
      v = *reinterpret_cast<const float4*>(gs + ((long long)hh * d.W + ww) * CO + c4 * 4);
    *reinterpret_cast<float4*>(gp + px * GP + c4 * 4) = v;
  }
  __syncthreads();
  const int pw = tid & 31, ph = tid >> 5;        // 8 rows x 32 columns
  float a0 = 0.f, a1 = 0.f;
  // gx[h][x] takes g[h + 1 - dh][x + 4 - j]  ->  patch row ph + 2 - dh, column pw + 8 - j
#pragma unroll
  for (int dh = 0; dh < KH; ++dh)
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const float* g0 = gp + ((ph + 2 - dh) * PW + pw + 8 - j) * GP;
      const float* wt = d.wt + (dh * KW + j) * (2 * CO);     // [tap][ci][co], uniform
#pragma unroll
      for (int c4 = 0; c4 < CO / 4; ++c4) {
        const float4 gv = *reinterpret_cast<const float4*>(g0 + c4 * 4);
        a0 += gv.x * wt[c4 * 4] + gv.y * wt[c4 * 4 + 1] + gv.z * wt[c4 * 4 + 2] + gv.w * wt[c4 * 4 + 3];
        a1 += gv.x * wt[CO + c4 * 4] + gv.y * wt[CO + c4 * 4 + 1] + gv.z * wt[CO + c4 * 4 + 2] +
              gv.w * wt[CO + c4 * 4 + 3];
      }
    }
  const int hh = h0 + ph, ww = w0 + pw;
  if (hh < d.H && ww < d.W) {
    float* o = d.gx + (long long)s * d.gx_seq + (long long)hh * d.gx_line + (long long)ww * 2;
    *reinterpret_cast<float2*>(o) = make_float2(a0, a1);
  }
}

// ---- conv_post of a resolution: Conv2d(32, 1, (3, 3), padding (1, 1)) (discriminators.py:184) ----
// One output channel: dot products of 288 floats per pixel.  As "narrow" implicit GEMMs (a wave per
// output row, shuffle reduction) these cost 0.4 ms per call for 0.1 GFLOP; here a thread owns a
// pixel and walks its 3 x 3 x 32 window in an LDS-staged patch (forward), a thread owns a weight
// and walks the tile's pixels (weight gradient), or 8 threads share a pixel's 32 channels (data
// gradient, bound by writing the 128-byte gradient rows).
constexpr int QTH = 8, QTW = 32;
constexpr int QPH = QTH + 2, QPW = QTW + 2;

__device__ __forceinline__ void stage32(float* patch, const float* xs, int H, int W, int h0, int w0,
                                        int tid) {
  for (int i = tid; i < QPH * QPW * (CO / 4); i += 256) {
    const int c4 = i & 7, px = i >> 3;
    const int r = px / QPW, c = px - r * QPW;
    const int hh = h0 - 1 + r, ww = w0 - 1 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hh >= 0 && hh < H && ww >= 0 && ww < W)
      v = *reinterpret_cast<const float4*>(xs + ((long long)hh * W + ww) * CO + c4 * 4);
    *reinterpret_cast<float4*>(patch + px * GP + c4 * 4) = v;
  }
}

__global__ __launch_bounds__(256) void convpost_fwd_kernel(const f2g_conv2ch_desc d) {
  extern __shared__ __attribute__((aligned(16))) float patch[];   // [QPH][QPW][GP]
  const int tid = threadIdx.x;
  const int tiles_w = (d.W + QTW - 1) / QTW;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int s = blockIdx.y;
  const int h0 = th * QTH, w0 = tw * QTW;
  // the 288 weights through LDS (every lane reads the same address: a broadcast) -- as uniform global
  // operands they took 288 scalar registers: 224 of them spilled, one wave per SIMD (round 4)
  __shared__ __attribute__((aligned(16))) float wl[9 * CO];
  for (int i = tid; i < 9 * CO; i += 256) wl[i] = d.w[i];
  stage32(patch, d.x + (long long)s * d.H * d.W * CO, d.H, d.W, h0, w0, tid);
  __syncthreads();
  const int pw = tid & 31, ph = tid >> 5;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {       // (not unrolled: all 72 weight quads live at once spill)
    const float* x0 = patch + ((ph + t / 3) * QPW + pw + t % 3) * GP;
    const float* wt = wl + t * CO;                       // [tap][ci]
#pragma unroll
    for (int c4 = 0; c4 < CO / 4; c4 += 2) {
      const float4 u = *reinterpret_cast<const float4*>(x0 + c4 * 4);
      const float4 v = *reinterpret_cast<const float4*>(x0 + c4 * 4 + 4);
      const float4 wu = *reinterpret_cast<const float4*>(wt + c4 * 4);
      const float4 wv = *reinterpret_cast<const float4*>(wt + c4 * 4 + 4);
      a0 += u.x * wu.x + u.y * wu.y + u.z * wu.z + u.w * wu.w;
      a1 += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
    }
  }
  const int hh = h0 + ph, ww = w0 + pw;
  if (hh < d.H && ww < d.W)
    d.y[((long long)s * d.H + hh) * d.W + ww] = a0 + a1 + (d.bias ? d.bias[0] : 0.f);
}

// gw[tap*32 + ci] += sum_px g[px] * x[px + tap][ci]  =  sum_q x[q][ci] * g[q - tap]: every pixel of the
// 32-channel map is read ONCE, straight from global memory (8 threads per pixel, 16 bytes each: whole 128-byte
// rows per lane group), and multiplied with the nine score gradients around it (a 1-channel map: L2-resident);
// 36 accumulators per thread, reduced over the block's 32 pixel groups in LDS, one atomic per weight and block.
// (Round 4's kernel kept a thread per WEIGHT and walked a staged tile's 256 pixels with one LDS read per FMA:
// 0.45 TB/s on the map, 224 us per launch.)
__global__ __launch_bounds__(256) void convpost_wgrad_kernel(const f2g_conv2ch_desc d, int px_per_block) {
  __shared__ float red[32][9 * CO + 1];
  const int tid = threadIdx.x, c4 = tid & 7, pg = tid >> 3;
  const long long npx = (long long)d.S * d.H * d.W;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long p0 = (long long)blockIdx.x * px_per_block;
  const long long pend = p0 + px_per_block < npx ? p0 + px_per_block : npx;
  constexpr int U = 4;                     // pixels in flight per thread
  for (long long p = p0 + pg; p < pend; p += 32 * U) {
    float4 v[U];
    float gn[U][9];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = p + 32 * u;
      const bool on = q < pend;
      const long long qq = on ? q : p;
      v[u] = *reinterpret_cast<const float4*>(d.x + qq * CO + c4 * 4);
      const int w = (int)(qq % d.W);
      const int h = (int)((qq / d.W) % d.H);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // x[q] is the window element of tap t for the output pixel q - (t/3 - 1, t%3 - 1)
        const int oh = h - (t / 3 - 1), ow = w - (t % 3 - 1);
        const bool ok = on && oh >= 0 && oh < d.H && ow >= 0 && ow < d.W;
        gn[u][t] = ok ? d.y[qq - (long long)(t / 3 - 1) * d.W - (t % 3 - 1)] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[t].x += gn[u][t] * v[u].x; acc[t].y += gn[u][t] * v[u].y;
        acc[t].z += gn[u][t] * v[u].z; acc[t].w += gn[u][t] * v[u].w;
      }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    red[pg][t * CO + c4 * 4 + 0] = acc[t].x; red[pg][t * CO + c4 * 4 + 1] = acc[t].y;
    red[pg][t * CO + c4 * 4 + 2] = acc[t].z; red[pg][t * CO + c4 * 4 + 3] = acc[t].w;
  }
  __syncthreads();
  for (int k = tid; k < 9 * CO; k += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int g2 = 0; g2 < 32; ++g2) s += red[g2][k];
    atomicAdd(d.gw + k, s);
  }
}

// gx[px][ci] = sum_tap g[px - tap] * w[tap][ci]: 8 threads per pixel, 4 channels each
__global__ __launch_bounds__(256) void convpost_dgrad_kernel(const f2g_conv2ch_desc d) {
  const long long npx = (long long)d.S * d.H * d.W;
  const int c4 = threadIdx.x & 7;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(d.w + t * CO + c4 * 4);
  for (long long i = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); i < npx;
       i += (long long)gridDim.x * 32) {
    const int ww = (int)(i % d.W);
    const long long q = i / d.W;
    const int hh = (int)(q % d.H);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sh = hh + 1 - t / 3, sw = ww + 1 - t % 3;
      if (sh >= 0 && sh < d.H && sw >= 0 && sw < d.W) {
        const float gv = d.y[i + (long long)(1 - t / 3) * d.W + (1 - t % 3)];
        a.x += gv * wv[t].x; a.y += gv * wv[t].y; a.z += gv * wv[t].z; a.w += gv * wv[t].w;
      }
    }
    *reinterpret_cast<float4*>(d.gx + i * CO + c4 * 4) = a;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

static bool conv2ch_ok(const f2g_conv2ch_desc* d) {
  return d && d->S > 0 && d->H > 0 && d->W > 0;
}

extern "C" int f2g_conv2ch_fwd(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y || (d->x_line & 1) || (d->x_seq & 1)) return F2G_EINVAL;
  if (!conv2ch_ok(d)) return F2G_OK;
  const bool persistent = f2g_opt(F2G_OPT_CONV2CH_V2) != 0;
  if (persistent && (long long)((d->W + TW - 1) / TW) * d->S >= 256) {   // column tiles x sequences fill the chip
    hipLaunchKernelGGL(conv2ch_fwd_p_kernel, dim3((d->W + TW - 1) / TW, d->S), dim3(256), 0, ST, *d);
    return f2g_check_launch();
  }
  const int tiles = ((d->H + FTH - 1) / FTH) * ((d->W + TW - 1) / TW);
  hipLaunchKernelGGL(conv2ch_fwd_kernel, dim3(tiles, d->S), dim3(256), 0, ST, *d);
  return f2g_check_launch();
}

extern "C" int f2g_conv2ch_wgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->y || !d->gw || (d->x_line & 1) || (d->x_seq & 1)) return F2G_EINVAL;
  if (((uintptr_t)d->y) & 15) return F2G_EINVAL;   // the gradient tile is staged with 16-byte loads
  if (!conv2ch_ok(d)) return F2G_OK;
  const int tiles_h = (d->H + FTH - 1) / FTH, tiles_w = (d->W + TW - 1) / TW;
  const int ntiles = d->S * tiles_h * tiles_w;
  int per = (ntiles + 511) / 512;        // <= 512 blocks: 0.9 M atomics on 1728 addresses per launch
  if (per < 1) per = 1;
  const bool persistent = f2g_opt(F2G_OPT_CONV2CH_V2) != 0;
  if (persistent)
    hipLaunchKernelGGL(conv2ch_wgrad_p_kernel, dim3((ntiles + per - 1) / per), dim3(256), 0, ST, *d,
                       tiles_h, tiles_w, per);
  else
    hipLaunchKernelGGL(conv2ch_wgrad_kernel, dim3((ntiles + per - 1) / per), dim3(256), 0, ST, *d,
                       tiles_h, tiles_w, per);
  return f2g_check_launch();
}

extern "C" int f2g_conv2ch_dgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->y || !d->wt || !d->gx || (d->gx_line & 1) || (d->gx_seq & 1)) return F2G_EINVAL;
  if ((((uintptr_t)d->y) & 15) || (((uintptr_t)d->gx) & 7)) return F2G_EINVAL;
  if (!conv2ch_ok(d)) return F2G_OK;
  const size_t smem = (size_t)DPH * PW * GP * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv2ch_dgrad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const int tiles = ((d->H + DTH - 1) / DTH) * ((d->W + TW - 1) / TW);
  hipLaunchKernelGGL(conv2ch_dgrad_kernel, dim3(tiles, d->S), dim3(256), smem, ST, *d);
  return f2g_check_launch();
}

// conv_post (32 -> 1, 3x3): x = (S, H, W, 32) dense, w = (9, 32) tap-major, y = scores (S*H*W)
extern "C" int f2g_convpost_fwd(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y || (((uintptr_t)d->x) & 15)) return F2G_EINVAL;
  if (!conv2ch_ok(d)) return F2G_OK;
  const size_t smem = (size_t)QPH * QPW * GP * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convpost_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const int tiles = ((d->H + QTH - 1) / QTH) * ((d->W + QTW - 1) / QTW);
  hipLaunchKernelGGL(convpost_fwd_kernel, dim3(tiles, d->S), dim3(256), smem, ST, *d);
  return f2g_check_launch();
}

// gw (9*32) += weight gradient; y = gradient of the scores (S*H*W)
extern "C" int f2g_convpost_wgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->x || !d->y || !d->gw || (((uintptr_t)d->x) & 15)) return F2G_EINVAL;
  if (!conv2ch_ok(d)) return F2G_OK;
  const long long npx = (long long)d->S * d->H * d->W;
  long long per = (npx + 1023) / 1024;     // <= 1024 blocks: 0.3 M atomics on 288 addresses per launch
  if (per < 128) per = 128;
  hipLaunchKernelGGL(convpost_wgrad_kernel, dim3((unsigned)((npx + per - 1) / per)), dim3(256), 0, ST, *d, (int)per);
  return f2g_check_launch();
}

// gx (S*H*W, 32) = data gradient (overwrites); y = gradient of the scores, w = (9, 32)
extern "C" int f2g_convpost_dgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream) {
  if (!d || !d->y || !d->w || !d->gx || (((uintptr_t)d->gx) & 15) || (((uintptr_t)d->w) & 15))
    return F2G_EINVAL;
  if (!conv2ch_ok(d)) return F2G_OK;
  const long long npx = (long long)d->S * d->H * d->W;
  hipLaunchKernelGGL(convpost_dgrad_kernel, dim3(f2g_grid_for(npx, 32, 16384)), dim3(256), 0, ST, *d);
  return f2g_check_launch();
}
