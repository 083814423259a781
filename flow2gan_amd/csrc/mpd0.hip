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

// ---- last layer of a period discriminator (discriminators.py:76,97-99: Conv2d(1024, 1, (3, 1),
// padding (1, 0))): one output channel -- as GEMMs with N = 1 / K = 3 / M = 1 these ran on the narrow VALU
// kernels at 150-300 us per call.  The 1024-channel map (halo layout, zero halo rows: no bounds tests)
// is streamed once: a row = 256 float4, a lane holds 4 of them.
constexpr int CP = 1024, KP = 3;

// scores[s*H + h] = bias + sum_j <y[s, halo + h + j - 1, :], w[j]>: a wave walks RH outputs and the
// RH + 2 rows they touch, every row read once
constexpr int RH = 8;

__global__ __launch_bounds__(256) void mpdpost_fwd_kernel(const f2g_mpdpost_desc d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = (d.H + RH - 1) / RH;
  const long long unit = (long long)blockIdx.x * 4 + wave;
  if (unit >= (long long)d.S * groups) return;
  const int s = (int)(unit / groups), h0 = (int)(unit - (long long)s * groups) * RH;
  float4 w[KP][4];
#pragma unroll
  for (int j = 0; j < KP; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) w[j][q] = reinterpret_cast<const float4*>(d.w + j * CP)[q * 64 + lane];
  const int Hp = d.H + 2 * d.halo;
  const float4* base = reinterpret_cast<const float4*>(d.y + ((long long)s * Hp + d.halo) * CP);
  float acc[RH];
#pragma unroll
  for (int i = 0; i < RH; ++i) acc[i] = 0.f;
  auto dot = [](const float4 (&v)[4], const float4 (&u)[4]) {
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) r += v[q].x * u[q].x + v[q].y * u[q].y + v[q].z * u[q].z + v[q].w * u[q].w;
    return r;
  };
#pragma unroll
  for (int i = -1; i <= RH; ++i) {       // input row h0 + i (halo rows / rows past H read as stored: zeros)
    const int hp = h0 + i;
    if (hp > d.H) break;                  // (row H is the zero halo row the last output touches)
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = base[(long long)hp * (CP / 4) + q * 64 + lane];
    if (i + 1 < RH) acc[i + 1] += dot(v, w[0]);                            // output h = hp + 1, tap 0
    if (i >= 0 && i < RH) acc[i] += dot(v, w[1]);                          // output h = hp, tap 1
    if (i >= 1) acc[i - 1] += dot(v, w[2]);                                // output h = hp - 1, tap 2
  }
  const float b = d.bias ? d.bias[0] : 0.f;
#pragma unroll
  for (int i = 0; i < RH; ++i) {
    const float r = wave_sum(acc[i]);
    if (lane == 0 && h0 + i < d.H) d.out[(long long)s * d.H + h0 + i] = r + b;
  }
}

// gy[s, halo + h, :] = g[s, h+1] w[0] + g[s, h] w[1] + g[s, h-1] w[2]   (rows 0..H-1; halo rows untouched)
// Round 5: optionally the leaky-ReLU backward of the 1024-channel layer the gradient lands on (mask_src = that
// layer's output, same halo layout; + the feature-matching term against fm_ref), the column sums of the result
// (that layer's bias gradient) and the result's three-piece image for the fp32-class data gradient that reads
// it next -- the separate pass over the map (read g, read y, write g) and the image pass (read g, write 6
// bytes per element) both read what this kernel has in registers.
__device__ __forceinline__ void split3_4(const float (&x)[4], uint2& p0, uint2& p1, uint2& p2) {
  unsigned short q[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 a = (__bf16)x[e];
    const float r1 = x[e] - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    q[0][e] = __builtin_bit_cast(unsigned short, a);
    q[1][e] = __builtin_bit_cast(unsigned short, b);
    q[2][e] = __builtin_bit_cast(unsigned short, c);
  }
  p0 = make_uint2(q[0][0] | ((unsigned)q[0][1] << 16), q[0][2] | ((unsigned)q[0][3] << 16));
  p1 = make_uint2(q[1][0] | ((unsigned)q[1][1] << 16), q[1][2] | ((unsigned)q[1][3] << 16));
  p2 = make_uint2(q[2][0] | ((unsigned)q[2][1] << 16), q[2][2] | ((unsigned)q[2][3] << 16));
}

__global__ __launch_bounds__(256) void mpdpost_dgrad_kernel(const f2g_mpdpost_desc d, int rows_per) {
  const int t = threadIdx.x;
  const float4 w0 = reinterpret_cast<const float4*>(d.w)[t], w1 = reinterpret_cast<const float4*>(d.w + CP)[t],
               w2 = reinterpret_cast<const float4*>(d.w + 2 * CP)[t];
  const long long R = (long long)d.S * d.H;
  const int Hp = d.H + 2 * d.halo;
  const float fmw = d.fm_ref ? d.fm_w * (d.fm_wdev ? d.fm_wdev[0] : 1.f) : 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  long long r = (long long)blockIdx.x * rows_per;
  const long long rend = r + rows_per < R ? r + rows_per : R;
  for (; r < rend; ++r) {
    const int s = (int)(r / d.H), h = (int)(r - (long long)s * d.H);
    const float* g = d.g + (long long)s * d.H;
    const float ga = h + 1 < d.H ? g[h + 1] : 0.f, gb = g[h], gc = h > 0 ? g[h - 1] : 0.f;
    float o[4];
    o[0] = ga * w0.x + gb * w1.x + gc * w2.x;
    o[1] = ga * w0.y + gb * w1.y + gc * w2.y;
    o[2] = ga * w0.z + gb * w1.z + gc * w2.z;
    o[3] = ga * w0.w + gb * w1.w + gc * w2.w;
    const long long off = ((long long)s * Hp + d.halo + h) * CP + 4 * t;
    if (d.mask_src) {
      const float4 yv = *reinterpret_cast<const float4*>(d.mask_src + off);
      const float y[4] = {yv.x, yv.y, yv.z, yv.w};
      if (d.fm_ref) {
        const float4 fv = *reinterpret_cast<const float4*>(d.fm_ref + off);
        const float f[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = y[e] - f[e];
          o[e] += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] *= y[e] > 0.f ? 1.f : d.mask_slope;
    }
    cs.x += o[0]; cs.y += o[1]; cs.z += o[2]; cs.w += o[3];
    *reinterpret_cast<float4*>(d.y + off) = make_float4(o[0], o[1], o[2], o[3]);
    if (d.x3_out) {
      uint2 p0, p1, p2;
      split3_4(o, p0, p1, p2);
      __bf16* q = reinterpret_cast<__bf16*>(d.x3_out) + (off >> 5) * 96 + (off & 31);
      *reinterpret_cast<uint2*>(q) = p0;
      *reinterpret_cast<uint2*>(q + 32) = p1;
      *reinterpret_cast<uint2*>(q + 64) = p2;
    }
  }
  if (d.colsum) {
    float* c = d.colsum + 4 * t;
    atomicAdd(c + 0, cs.x); atomicAdd(c + 1, cs.y); atomicAdd(c + 2, cs.z); atomicAdd(c + 3, cs.w);
  }
}

// gw[j][c] += sum_{s,h} g[s, h] * y[s, halo + h + j - 1, c]: every map row read once, it feeds the
// three taps with the scores' gradients of its neighbours
__global__ __launch_bounds__(256) void mpdpost_wgrad_kernel(const f2g_mpdpost_desc d, float* gw, int rows_per) {
  const int t = threadIdx.x;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  const int Hq = d.H + 2;                       // rows -1 .. H of a sequence
  const long long R = (long long)d.S * Hq;
  const int Hp = d.H + 2 * d.halo;
  long long r = (long long)blockIdx.x * rows_per;
  const long long rend = r + rows_per < R ? r + rows_per : R;
  // (round 5: eight map rows requested before any is consumed -- one 16-byte load in flight per thread left
  // the kernel at 1.5 TB/s: these streams are a memory-level-parallelism problem)
  constexpr int U = 8;
  for (; r < rend; r += U) {
    float4 v[U];
    float g0[U], g1[U], g2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long ru = r + u < rend ? r + u : rend - 1;
      const int s = (int)(ru / Hq), hp = (int)(ru - (long long)s * Hq) - 1;
      const float* g = d.g + (long long)s * d.H;
      const bool on = r + u < rend;
      g0[u] = (on && hp + 1 < d.H) ? g[hp + 1] : 0.f;                  // tap 0 of output hp + 1
      g1[u] = (on && hp >= 0 && hp < d.H) ? g[hp] : 0.f;               // tap 1 of output hp
      g2[u] = (on && hp >= 1) ? g[hp - 1] : 0.f;                       // tap 2 of output hp - 1
      v[u] = reinterpret_cast<const float4*>(d.y + ((long long)s * Hp + d.halo + hp) * CP)[t];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0.x += g0[u] * v[u].x; a0.y += g0[u] * v[u].y; a0.z += g0[u] * v[u].z; a0.w += g0[u] * v[u].w;
      a1.x += g1[u] * v[u].x; a1.y += g1[u] * v[u].y; a1.z += g1[u] * v[u].z; a1.w += g1[u] * v[u].w;
      a2.x += g2[u] * v[u].x; a2.y += g2[u] * v[u].y; a2.z += g2[u] * v[u].z; a2.w += g2[u] * v[u].w;
    }
  }
  float* o = gw + 4 * t;
  atomicAdd(o + 0, a0.x); atomicAdd(o + 1, a0.y); atomicAdd(o + 2, a0.z); atomicAdd(o + 3, a0.w);
  o += CP;
  atomicAdd(o + 0, a1.x); atomicAdd(o + 1, a1.y); atomicAdd(o + 2, a1.z); atomicAdd(o + 3, a1.w);
  o += CP;
  atomicAdd(o + 0, a2.x); atomicAdd(o + 1, a2.y); atomicAdd(o + 2, a2.z); atomicAdd(o + 3, a2.w);
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

extern "C" int f2g_mpdpost_fwd(const f2g_mpdpost_desc* d, f2g_stream_t stream) {
  if (!d || !d->y || !d->w || !d->out || d->H <= 0 || d->halo < 1 || (((uintptr_t)d->y) & 15) ||
      (((uintptr_t)d->w) & 15))
    return F2G_EINVAL;
  if (d->S <= 0) return F2G_OK;
  const long long units = (long long)d->S * ((d->H + RH - 1) / RH);
  hipLaunchKernelGGL(mpdpost_fwd_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *d);
  return f2g_check_launch();
}

extern "C" int f2g_mpdpost_dgrad(const f2g_mpdpost_desc* d, f2g_stream_t stream) {
  if (!d || !d->y || !d->w || !d->g || d->H <= 0 || (((uintptr_t)d->y) & 15) || (((uintptr_t)d->w) & 15))
    return F2G_EINVAL;
  if (d->S <= 0) return F2G_OK;
  if ((d->mask_src && (((uintptr_t)d->mask_src) & 15)) || (d->fm_ref && ((((uintptr_t)d->fm_ref) & 15) || !d->mask_src)) ||
      (d->x3_out && (((uintptr_t)d->x3_out) & 7)))
    return F2G_EINVAL;
  const long long R = (long long)d->S * d->H;
  // <= 2048 blocks over contiguous row ranges (with column sums: 1024 atomics per block)
  long long per = (R + 2047) / 2048;
  if (per < 1) per = 1;
  hipLaunchKernelGGL(mpdpost_dgrad_kernel, dim3((unsigned)((R + per - 1) / per)), dim3(256), 0, (hipStream_t)stream,
                     *d, (int)per);
  return f2g_check_launch();
}

extern "C" int f2g_mpdpost_wgrad(const f2g_mpdpost_desc* d, float* gw, f2g_stream_t stream) {
  if (!d || !d->y || !d->g || !gw || d->H <= 0 || d->halo < 1 || (((uintptr_t)d->y) & 15)) return F2G_EINVAL;
  if (d->S <= 0) return F2G_OK;
  const long long R = (long long)d->S * (d->H + 2);
  // <= 256 blocks (one per CU, eight rows in flight per thread): 0.8 M atomics on 3072 addresses per launch --
  // with 1024 blocks the same-address atomics, not the stream, set the time (128 us against 106)
  int per = (int)((R + 255) / 256);
  if (per < 8) per = 8;
  hipLaunchKernelGGL(mpdpost_wgrad_kernel, dim3((unsigned)((R + per - 1) / per)), dim3(256), 0,
                     (hipStream_t)stream, *d, gw, per);
  return f2g_check_launch();
}
