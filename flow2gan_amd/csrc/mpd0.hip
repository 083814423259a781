// First layer of a period discriminator (reference discriminators.py:65-67,92-94: Conv2d(1, 32, (5, 1),
// stride (3, 1), padding (2, 0)) + leaky ReLU on the waveform folded by its period): one input
// channel, five taps -- 320 FLOP per 128-byte output row.  As a GEMM (K = 5, N = 32) it ran on the
// element-wise loaders at 2 TFLOP/s (170 us per call, 320 us for the weight gradient, 175 us per
// stride residue of the data gradient); these three kernels are plain HBM streams over the 32-channel
// map (131 MB at B = 64): a row of the map = 8 lanes x float4, eight rows per wave instruction, the
// 20 weights of a lane's four channels in registers.  Exact fp32 VALU arithmetic in every GEMM mode.
//   x: (S, H) folded waveform;  maps: halo layout (S, Hout + 2*halo, 32), zero halo rows.
#include "common.h"

namespace {

constexpr int C = 32, KT = 5, ST = 3, PD = 2;

__device__ __forceinline__ void load_taps(const float* xs, int H, int h, float (&xv)[KT]) {
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int i = ST * h + j - PD;
    xv[j] = (i >= 0 && i < H) ? xs[i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void mpd0_fwd_kernel(const f2g_mpd0_desc d) {
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float w[4][KT], b[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b[c] = d.bias ? d.bias[4 * c4 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < KT; ++j) w[c][j] = d.w[(4 * c4 + c) * KT + j];
  }
  const long long R = (long long)d.S * d.Hout;
  const int Hp = d.Hout + 2 * d.halo;
  for (long long r = (long long)blockIdx.x * 32 + rl; r < R; r += (long long)gridDim.x * 32) {
    const int s = (int)(r / d.Hout), h = (int)(r - (long long)s * d.Hout);
    float xv[KT];
    load_taps(d.x + (long long)s * d.H, d.H, h, xv);
    float4 o;
    float* op = &o.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = b[c];
#pragma unroll
      for (int j = 0; j < KT; ++j) v = fmaf(w[c][j], xv[j], v);
      op[c] = v > 0.f ? v : d.slope * v;
    }
    *reinterpret_cast<float4*>(d.y + ((long long)s * Hp + d.halo + h) * C + 4 * c4) = o;
  }
}

// gw[co][j] += sum_{s,h} g[s, halo + h, co] * x[s, 3h + j - 2]
__global__ __launch_bounds__(256) void mpd0_wgrad_kernel(const f2g_mpd0_desc d, float* gw) {
  __shared__ float red[32][8][4 * KT + 1];
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float acc[4][KT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[c][j] = 0.f;
  const long long R = (long long)d.S * d.Hout;
  const int Hp = d.Hout + 2 * d.halo;
  for (long long r = (long long)blockIdx.x * 32 + rl; r < R; r += (long long)gridDim.x * 32) {
    const int s = (int)(r / d.Hout), h = (int)(r - (long long)s * d.Hout);
    float xv[KT];
    load_taps(d.x + (long long)s * d.H, d.H, h, xv);
    const float4 g = *reinterpret_cast<const float4*>(d.y + ((long long)s * Hp + d.halo + h) * C + 4 * c4);
    const float* gp = &g.x;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < KT; ++j) acc[c][j] = fmaf(gp[c], xv[j], acc[c][j]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < KT; ++j) red[rl][c4][c * KT + j] = acc[c][j];
  __syncthreads();
  if (threadIdx.x < C * KT) {   // 160 outputs: (co, j)
    const int co = threadIdx.x / KT, j = threadIdx.x - co * KT;
    float v = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) v += red[i][co >> 2][(co & 3) * KT + j];
    atomicAdd(gw + co * KT + j, v);
  }
}

// gx[s, i] = sum_{j = (i+2) mod 3, +3} sum_co g[s, halo + (i + 2 - j) / 3, co] * w[co][j]
__global__ __launch_bounds__(256) void mpd0_dgrad_kernel(const f2g_mpd0_desc d, float* gx) {
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float w[4][KT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < KT; ++j) w[c][j] = d.w[(4 * c4 + c) * KT + j];
  const long long R = (long long)d.S * d.H;
  const int Hp = d.Hout + 2 * d.halo;
  for (long long r0 = (long long)blockIdx.x * 32; r0 < R; r0 += (long long)gridDim.x * 32) {
    const long long r = r0 + rl;
    float v = 0.f;
    if (r < R) {
      const int s = (int)(r / d.H), i = (int)(r - (long long)s * d.H);
      const int j0 = (i + PD) % ST;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = j0 + ST * u;
        const int hn = i + PD - j;          // = 3 h
        if (j < KT && hn >= 0) {
          const int h = hn / ST;
          if (h < d.Hout) {
            const float4 g = *reinterpret_cast<const float4*>(d.y + ((long long)s * Hp + d.halo + h) * C + 4 * c4);
            // (j is one of two values per lane: select the weights without dynamic register indexing)
#pragma unroll
            for (int jj = 0; jj < KT; ++jj)
              if (jj == j) v += g.x * w[0][jj] + g.y * w[1][jj] + g.z * w[2][jj] + g.w * w[3][jj];
          }
        }
      }
    }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (c4 == 0 && r < R) gx[r] = v;
  }
}

int grid_rows(long long rows) {
  long long b = (rows + 31) / 32;
  if (b > 256 * 8) b = 256 * 8;
  return (int)(b < 1 ? 1 : b);
}

bool ok_desc(const f2g_mpd0_desc* d) {
  return d && d->x && d->y && d->S >= 0 && d->H > 0 && d->halo >= 0 &&
         d->Hout == (d->H + 2 * PD - KT) / ST + 1 && ((((uintptr_t)d->y) & 15) == 0);
}

}  // namespace

extern "C" int f2g_mpd0_fwd(const f2g_mpd0_desc* d, f2g_stream_t stream) {
  if (!ok_desc(d) || !d->w) return F2G_EINVAL;
  if (d->S == 0) return F2G_OK;
  hipLaunchKernelGGL(mpd0_fwd_kernel, dim3(grid_rows((long long)d->S * d->Hout)), dim3(256), 0,
                     (hipStream_t)stream, *d);
  return f2g_check_launch();
}

extern "C" int f2g_mpd0_wgrad(const f2g_mpd0_desc* d, float* gw, f2g_stream_t stream) {
  if (!ok_desc(d) || !gw) return F2G_EINVAL;
  if (d->S == 0) return F2G_OK;
  long long b = ((long long)d->S * d->Hout + 31) / 32;
  if (b > 512) b = 512;   // bounds the atomics: 160 per block
  hipLaunchKernelGGL(mpd0_wgrad_kernel, dim3((unsigned)(b < 1 ? 1 : b)), dim3(256), 0, (hipStream_t)stream, *d, gw);
  return f2g_check_launch();
}

extern "C" int f2g_mpd0_dgrad(const f2g_mpd0_desc* d, float* gx, f2g_stream_t stream) {
  if (!d || !d->y || !d->w || !gx || d->H <= 0 || d->Hout != (d->H + 2 * PD - KT) / ST + 1) return F2G_EINVAL;
  if (d->S == 0) return F2G_OK;
  hipLaunchKernelGGL(mpd0_dgrad_kernel, dim3(grid_rows((long long)d->S * d->H)), dim3(256), 0,
                     (hipStream_t)stream, *d, gx);
  return f2g_check_launch();
}
