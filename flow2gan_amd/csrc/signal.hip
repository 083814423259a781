// Signal-domain kernels around the DFT GEMMs: iSTFT overlap-add (+branch mean / Euler
// accumulation), its backward, the STFT reflect-pad gradient fold, the MPD period fold and the
// MRD peak normalisation.  All HBM-bound, one thread per output sample, coalesced along time.
// Reference: modules.py:52-116,719; generator.py:165-168; discriminators.py:82-90,186-190.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ola_kernel(const float* frames, long long ldf, float* out,
                                                  int B, int F, int N, int hop, int T,
                                                  const float* window, const float* wbranch,
                                                  float wscale, int accumulate) {
  const int b = blockIdx.y;
  const int Lout = hop * (F - 1);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    float y = 0.f;
    if (j < Lout) {
      const int p = j + N / 2;
      int mhi = p / hop;
      if (mhi > F - 1) mhi = F - 1;
      int mlo = (p - N + hop) / hop;  // ceil((p-N+1)/hop) for p-N+1 > 0
      if (p - N + 1 <= 0) mlo = 0;
      float val = 0.f, env = 0.f;
      for (int m = mlo; m <= mhi; ++m) {
        const int n = p - m * hop;
        val += frames[((long long)b * F + m) * ldf + n];
        const float w = window[n];
        env += w * w;
      }
      y = val / env;
    }
    const float sc = wscale * (wbranch ? wbranch[b] : 1.f);
    const long long o = (long long)b * T + j;
    out[o] = accumulate ? out[o] + sc * y : sc * y;
  }
}

// the overlap-adds of up to four branches in one pass over the output (same arithmetic and the
// same order of the branch sums as n ola_kernel launches)
__global__ __launch_bounds__(256) void ola_multi_kernel(const f2g_ola_multi_desc d, float* out, int B,
                                                        int T, float wscale, int accumulate) {
  const int b = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    const long long o = (long long)b * T + j;
    float acc = accumulate ? out[o] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < d.n) {
        const int F = d.F[i], N = d.n_fft[i], hop = d.hop[i];
        const int Lout = hop * (F - 1);
        float y = 0.f;
        if (j < Lout) {
          const int p = j + N / 2;
          int mhi = p / hop;
          if (mhi > F - 1) mhi = F - 1;
          int mlo = (p - N + hop) / hop;
          if (p - N + 1 <= 0) mlo = 0;
          float val = 0.f, env = 0.f;
          for (int m = mlo; m <= mhi; ++m) {
            const int n = p - m * hop;
            val += d.frames[i][((long long)b * F + m) * d.ldf[i] + n];
            const float w = d.window[i][n];
            env += w * w;
          }
          y = val / env;
        }
        const float sc = wscale * (d.wbranch[i] ? d.wbranch[i][b] : 1.f);
        acc = (i == 0 && !accumulate) ? sc * y : acc + sc * y;
      }
    }
    out[o] = acc;
  }
}

__global__ __launch_bounds__(256) void ola_bwd_kernel(const float* gout, float* gframes,
                                                      long long ldf, int B, int F, int N, int hop,
                                                      int T, const float* window,
                                                      const float* wbranch, float wscale) {
  // one block per frame row (b, m); threads stride n.
  const int row = blockIdx.x;
  const int b = row / F, m = row - b * F;
  const int Lout = hop * (F - 1);
  const int lim = Lout < T ? Lout : T;
  const float sc = wscale * (wbranch ? wbranch[b] : 1.f);
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int p = m * hop + n;
    const int j = p - N / 2;
    float g = 0.f;
    if (j >= 0 && j < lim) {
      int mhi = p / hop;
      if (mhi > F - 1) mhi = F - 1;
      int mlo = (p - N + hop) / hop;
      if (p - N + 1 <= 0) mlo = 0;
      float env = 0.f;
      for (int mm = mlo; mm <= mhi; ++mm) {
        const float w = window[p - mm * hop];
        env += w * w;
      }
      g = sc * gout[(long long)b * T + j] / env;
    }
    gframes[(long long)row * ldf + n] = g;
  }
}

__device__ __forceinline__ float frames_at(const float* g, long long ldf, int b, int F, int N,
                                           int hop, int q) {
  // sum over frames m of g[b, m, q - m*hop]
  int mhi = q / hop;
  if (mhi > F - 1) mhi = F - 1;
  int mlo = (q - N + hop) / hop;
  if (q - N + 1 <= 0) mlo = 0;
  float s = 0.f;
  for (int m = mlo; m <= mhi; ++m) s += g[((long long)b * F + m) * ldf + (q - m * hop)];
  return s;
}

__global__ __launch_bounds__(256) void frames_fold_kernel(const float* gframes, long long ldf,
                                                          float* gx, int B, int F, int N, int hop,
                                                          int T, int accumulate) {
  const int b = blockIdx.y;
  const int half = N / 2;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
    float s = frames_at(gframes, ldf, b, F, N, hop, t + half);
    if (t >= 1 && t <= half) s += frames_at(gframes, ldf, b, F, N, hop, half - t);
    if (t <= T - 2 && t >= T - 1 - half)
      s += frames_at(gframes, ldf, b, F, N, hop, half + 2 * (T - 1) - t);
    const long long o = (long long)b * T + t;
    gx[o] = accumulate ? gx[o] + s : s;
  }
}

__global__ __launch_bounds__(256) void period_fold_kernel(float* out, const float* x, int B, int T,
                                                          int p, int H) {
  const long long total = (long long)B * p * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const long long q = i / H;
    const int w = (int)(q % p);
    const int b = (int)(q / p);
    int t = h * p + w;
    if (t >= T) t = 2 * (T - 1) - t;
    out[i] = x[(long long)b * T + t];
  }
}

__global__ __launch_bounds__(256) void period_fold_bwd_kernel(float* gx, const float* gout, int B,
                                                              int T, int p, int H,
                                                              int accumulate) {
  const long long total = (long long)B * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int b = (int)(i / T);
    float s = gout[((long long)b * p + (t % p)) * H + t / p];
    const int t2 = 2 * (T - 1) - t;  // padded index mirrored onto t
    if (t2 >= T && t2 < H * p) s += gout[((long long)b * p + (t2 % p)) * H + t2 / p];
    gx[i] = accumulate ? gx[i] + s : s;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// one block per row: mean, peak of |x-mean| (first argmax, like torch.max), normalise.
__global__ __launch_bounds__(256) void peaknorm_fwd_kernel(float* y, float* stats, const float* x,
                                                           int rows, int T) {
  __shared__ float sh[4];
  __shared__ float shv[4];
  __shared__ int shi[4];
  const int r = blockIdx.x;
  const float* xr = x + (long long)r * T;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) s += xr[t];
  const float mean = block_sum(s, sh) / (float)T;
  float best = -1.f;
  int bi = 0x7fffffff;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float a = fabsf(xr[t] - mean);
    if (a > best) { best = a; bi = t; }
  }
  // wave argmax with smallest-index tie break
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
  __syncthreads();
  for (int i = 0; i < 4; ++i)
    if (shv[i] > best || (shv[i] == best && shi[i] < bi)) { best = shv[i]; bi = shi[i]; }
  const float inv = 0.8f / (best + 1e-9f);
  float* yr = y + (long long)r * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) yr[t] = (xr[t] - mean) * inv;
  if (threadIdx.x == 0 && stats) {
    stats[r * 3 + 0] = mean;
    stats[r * 3 + 1] = best;
    stats[r * 3 + 2] = (float)bi;
  }
}

__global__ __launch_bounds__(256) void peaknorm_bwd_kernel(float* gx, const float* gy,
                                                           const float* x, const float* stats,
                                                           int rows, int T) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  const float mean = stats[r * 3 + 0], peak = stats[r * 3 + 1];
  const int k = (int)stats[r * 3 + 2];
  const float* xr = x + (long long)r * T;
  const float* gr = gy + (long long)r * T;
  const float d = peak + 1e-9f;
  float s1 = 0.f, s2 = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float g = gr[t];
    s1 += g;
    s2 += g * (xr[t] - mean);
  }
  const float sum_g = block_sum(s1, sh);
  const float sum_gc = block_sum(s2, sh);
  const float ck = xr[k] - mean;
  const float sgn = ck > 0.f ? 1.f : (ck < 0.f ? -1.f : 0.f);
  const float spike = -sgn * 0.8f * sum_gc / (d * d);  // lands on index k
  const float gc_mean = (0.8f * sum_g / d + spike) / (float)T;
  float* gxr = gx + (long long)r * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float g = 0.8f * gr[t] / d;
    if (t == k) g += spike;
    gxr[t] = g - gc_mean;
  }
}

}  // namespace

extern "C" int f2g_istft_ola(const float* frames, int64_t ldf, float* out, int32_t B, int32_t F,
                             int32_t n_fft, int32_t hop, int32_t T, const float* window,
                             const float* wbranch, float wscale, int32_t accumulate,
                             f2g_stream_t stream) {
  if (!frames || !out || !window || n_fft < 2 || hop < 1 || hop > n_fft) return F2G_EINVAL;
  if (B <= 0 || T <= 0) return F2G_OK;
  dim3 grid(f2g_grid_for(T, 256, 1024), B);
  hipLaunchKernelGGL(ola_kernel, grid, dim3(256), 0, (hipStream_t)stream, frames, (long long)ldf,
                     out, B, F, n_fft, hop, T, window, wbranch, wscale, accumulate);
  return f2g_check_launch();
}

extern "C" int f2g_istft_ola_multi(const f2g_ola_multi_desc* d, float* out, int32_t B, int32_t T,
                                   float wscale, int32_t accumulate, f2g_stream_t stream) {
  if (!d || !out || d->n < 1 || d->n > 4) return F2G_EINVAL;
  for (int i = 0; i < d->n; ++i)
    if (!d->frames[i] || !d->window[i] || d->n_fft[i] < 2 || d->hop[i] < 1 || d->hop[i] > d->n_fft[i] ||
        d->F[i] < 1)
      return F2G_EINVAL;
  if (B <= 0 || T <= 0) return F2G_OK;
  dim3 grid(f2g_grid_for(T, 256, 1024), B);
  hipLaunchKernelGGL(ola_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d, out, B, T, wscale,
                     accumulate);
  return f2g_check_launch();
}

extern "C" int f2g_istft_ola_bwd(const float* gout, float* gframes, int64_t ldf, int32_t B,
                                 int32_t F, int32_t n_fft, int32_t hop, int32_t T,
                                 const float* window, const float* wbranch, float wscale,
                                 f2g_stream_t stream) {
  if (!gout || !gframes || !window || n_fft < 2 || hop < 1 || hop > n_fft) return F2G_EINVAL;
  if (B <= 0 || F <= 0) return F2G_OK;
  hipLaunchKernelGGL(ola_bwd_kernel, dim3(B * F), dim3(256), 0, (hipStream_t)stream, gout,
                     gframes, (long long)ldf, B, F, n_fft, hop, T, window, wbranch, wscale);
  return f2g_check_launch();
}

extern "C" int f2g_frames_fold(const float* gframes, int64_t ldf, float* gx, int32_t B, int32_t F,
                               int32_t n_fft, int32_t hop, int32_t T, int32_t accumulate,
                               f2g_stream_t stream) {
  if (!gframes || !gx || n_fft < 2 || hop < 1 || T <= n_fft / 2) return F2G_EINVAL;
  if (B <= 0) return F2G_OK;
  dim3 grid(f2g_grid_for(T, 256, 1024), B);
  hipLaunchKernelGGL(frames_fold_kernel, grid, dim3(256), 0, (hipStream_t)stream, gframes,
                     (long long)ldf, gx, B, F, n_fft, hop, T, accumulate);
  return f2g_check_launch();
}

extern "C" int f2g_period_fold(float* out, const float* x, int32_t B, int32_t T, int32_t p,
                               int32_t H, f2g_stream_t stream) {
  if (!out || !x || p < 1 || H * p < T || H * p >= T + p || H * p - T >= T) return F2G_EINVAL;
  hipLaunchKernelGGL(period_fold_kernel, dim3(f2g_grid_for((int64_t)B * p * H, 256)), dim3(256), 0,
                     (hipStream_t)stream, out, x, B, T, p, H);
  return f2g_check_launch();
}

extern "C" int f2g_period_fold_bwd(float* gx, const float* gout, int32_t B, int32_t T, int32_t p,
                                   int32_t H, int32_t accumulate, f2g_stream_t stream) {
  if (!gx || !gout || p < 1 || H * p < T || H * p >= T + p) return F2G_EINVAL;
  hipLaunchKernelGGL(period_fold_bwd_kernel, dim3(f2g_grid_for((int64_t)B * T, 256)), dim3(256), 0,
                     (hipStream_t)stream, gx, gout, B, T, p, H, accumulate);
  return f2g_check_launch();
}

extern "C" int f2g_peaknorm_fwd(float* y, float* stats, const float* x, int32_t rows, int32_t T,
                                f2g_stream_t stream) {
  if (!y || !x || T < 1) return F2G_EINVAL;
  if (rows <= 0) return F2G_OK;
  hipLaunchKernelGGL(peaknorm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, y, stats,
                     x, rows, T);
  return f2g_check_launch();
}

extern "C" int f2g_peaknorm_bwd(float* gx, const float* gy, const float* x, const float* stats,
                                int32_t rows, int32_t T, f2g_stream_t stream) {
  if (!gx || !gy || !x || !stats || T < 1) return F2G_EINVAL;
  if (rows <= 0) return F2G_OK;
  hipLaunchKernelGGL(peaknorm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, gx, gy, x,
                     stats, rows, T);
  return f2g_check_launch();
}

// ---- on-device data front end (reference flow2gan/dataset.py:122-175) -------------
// x: (B, C, T) crops (item_stride, ch_stride in floats), lens[b] valid samples.
//   wave_stats: stats[b] = { sqrt(mean over channels and time of x^2)  (the silence test of
//               dataset.py:130-131 runs on the loaded multi-channel array),
//               max_t |mean_c x[b,c,t]|  (peak of the mono mix, for sox `norm`) }
//   wave_gain : out[b,t] = mean_c x[b,c,t] * target_peak[b] / peak[b]  (t < lens[b]; 0 beyond:
//               pad_sequence of dataset.py:43); target_peak[b] <= 0 leaves the level untouched
namespace {

__global__ __launch_bounds__(256) void wave_stats_kernel(const float* x, long long item_stride,
                                                         long long ch_stride, int C,
                                                         const int* lens, float* stats) {
  __shared__ float sh[8];
  const int b = blockIdx.x;
  const int n = lens[b];
  const float* xb = x + (long long)b * item_stride;
  float sq = 0.f, pk = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) {
    float m = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = xb[(long long)c * ch_stride + t];
      sq += v * v;
      m += v;
    }
    pk = fmaxf(pk, fabsf(m / (float)C));
  }
  sq = wave_sum(sq);
  pk = wave_max(pk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave] = sq; sh[4 + wave] = pk; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = sh[0] + sh[1] + sh[2] + sh[3];
    const float p = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
    stats[2 * b] = n > 0 ? sqrtf(s / ((float)n * (float)C)) : 0.f;
    stats[2 * b + 1] = p;
  }
}

__global__ __launch_bounds__(256) void wave_gain_kernel(float* out, long long ldo, const float* x,
                                                        long long item_stride, long long ch_stride,
                                                        int C, int T, const int* lens,
                                                        const float* stats,
                                                        const float* target_peak) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float v = 0.f;
  if (t < lens[b]) {
    const float* xb = x + (long long)b * item_stride;
    float m = 0.f;
    for (int c = 0; c < C; ++c) m += xb[(long long)c * ch_stride + t];
    m /= (float)C;
    const float tp = target_peak ? target_peak[b] : 0.f;
    const float pk = stats[2 * b + 1];
    v = (tp > 0.f && pk > 0.f) ? m * (tp / pk) : m;
  }
  out[(long long)b * ldo + t] = v;
}

}  // namespace

extern "C" int f2g_wave_stats(const float* x, int64_t item_stride, int64_t ch_stride, int32_t B,
                              int32_t C, const int32_t* lens, float* stats, f2g_stream_t stream) {
  if (!x || !lens || !stats || C < 1) return F2G_EINVAL;
  if (B <= 0) return F2G_OK;
  hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)item_stride, (long long)ch_stride, C, lens, stats);
  return f2g_check_launch();
}

extern "C" int f2g_wave_gain(float* out, int64_t ldo, const float* x, int64_t item_stride,
                             int64_t ch_stride, int32_t B, int32_t C, int32_t T,
                             const int32_t* lens, const float* stats, const float* target_peak,
                             f2g_stream_t stream) {
  if (!out || !x || !lens || !stats || C < 1 || T < 1) return F2G_EINVAL;
  if (B <= 0) return F2G_OK;
  hipLaunchKernelGGL(wave_gain_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                     out, (long long)ldo, x, (long long)item_stride, (long long)ch_stride, C, T, lens,
                     stats, target_peak);
  return f2g_check_launch();
}

// ---- reflect padding of STFT inputs (torch.stft center=True, modules.py:69-78) as data -------------
namespace {
__global__ __launch_bounds__(256) void reflect_pad_kernel(float* out, const float* x, int B, int T,
                                                          int pad, int Tp) {
  const long long total = (long long)B * Tp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int b = (int)(i / Tp), j = (int)(i - (long long)b * Tp);
    int t = j - pad;
    float v = 0.f;
    if (j < T + 2 * pad) {
      t = t < 0 ? -t : t;
      t = t >= T ? 2 * (T - 1) - t : t;
      v = x[(long long)b * T + t];
    }
    out[i] = v;
  }
}
}  // namespace

extern "C" int f2g_reflect_pad(float* out, const float* x, int32_t B, int32_t T, int32_t pad,
                               int32_t Tp, f2g_stream_t stream) {
  if (!out || !x || pad < 0 || pad >= T || Tp < T + 2 * pad) return F2G_EINVAL;
  if (B <= 0) return F2G_OK;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(f2g_grid_for((int64_t)B * Tp, 256)), dim3(256), 0,
                     (hipStream_t)stream, out, x, B, T, pad, Tp);
  return f2g_check_launch();
}
