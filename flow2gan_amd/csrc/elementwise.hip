// Elementwise / reduction / loss kernels of the flow-matching step and the GAN loss stack.
// Reference: generator.py:186-199,217,263-269; gan.py:57-99; modules.py:217-232,571;
// utils.py:221-232; discriminators.py:94,205.  All HBM-bound, grid-stride, coalesced.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void axpby_rows_kernel(float* y, const float* x0,
                                                         const float* x1, const float* ca,
                                                         const float* cb, float sa, float sb,
                                                         int rows, int cols) {
  const long long total = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols);
    const float a = ca ? ca[r] : sa;
    const float b = cb ? cb[r] : sb;
    float v = a * x0[i];
    if (x1) v += b * x1[i];
    y[i] = v;
  }
}

__global__ __launch_bounds__(256) void clamp_kernel(float* y, const float* x, float lo, float hi,
                                                    long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}

__global__ __launch_bounds__(256) void log_clip_kernel(float* x, long long n, float clip) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = logf(fmaxf(x[i], clip));
}

__global__ __launch_bounds__(256) void fill_kernel(float* x, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = v;
}

// gradient-exchange arenas (dist.py): [n gradients | nflags "used" flags]
__global__ __launch_bounds__(256) void bucket_arm_kernel(float4* x, long long n4, float* tail, int ntail,
                                                         float* flags, int nflags) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n4; i += (long long)gridDim.x * blockDim.x) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i0 < ntail) tail[i0] = 0.f;
  for (long long i = i0; i < nflags; i += (long long)gridDim.x * blockDim.x) flags[i] = 1.f;
}

__global__ __launch_bounds__(256) void scale_kernel(float* x, float s, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] *= s;
}

__global__ __launch_bounds__(256) void silu_kernel(float* y, const float* x, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v / (1.f + expf(-v));
  }
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(float* gx, const float* gy, const float* x,
                                                       long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float s = 1.f / (1.f + expf(-v));
    gx[i] = gy[i] * s * (1.f + v * (1.f - s));
  }
}

__global__ __launch_bounds__(256) void time_embedding_kernel(float* out, const float* t, int B,
                                                             int dim, float scale) {
  const int half = dim / 2;
  const float k = logf(10000.f) / (float)(half - 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
    const int b = i / half, j = i - b * half;
    const float f = expf((float)j * -k);
    const float arg = scale * t[b] * f;
    out[(long long)b * dim + j] = sinf(arg);
    out[(long long)b * dim + half + j] = cosf(arg);
  }
}

__global__ __launch_bounds__(256) void mask_rows_kernel(float* x, long long ld, int B, int F, int C,
                                                        const int* lens) {
  const long long total = (long long)B * F * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const int f = (int)(row % F);
    const int b = (int)(row / F);
    if (f >= lens[b]) x[row * ld + c] = 0.f;
  }
}

// out[c] += sum_r a[r,c] (* b[r,c]).  A block is CS columns x (256/CS) row lanes (CS = power of
// two >= min(cols, 256)), so narrow matrices (32-channel MRD maps) still use all 256 threads and
// every row segment is read coalesced; row lanes are combined through LDS, one atomic per column.
__global__ __launch_bounds__(256) void colsum_kernel(float* out, const float* a, long long lda,
                                                     const float* b, long long ldb, int rows,
                                                     int cols, int rows_per, int cs_log2) {
  __shared__ float red[256];
  const int CS = 1 << cs_log2;
  const int RS = 256 >> cs_log2;
  const int cl = threadIdx.x & (CS - 1), rl = threadIdx.x >> cs_log2;
  const int c = blockIdx.x * CS + cl;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < cols) {
    if (b) {
      for (int r = r0 + rl; r < r1; r += RS) s += a[(long long)r * lda + c] * b[(long long)r * ldb + c];
    } else {
      for (int r = r0 + rl; r < r1; r += RS) s += a[(long long)r * lda + c];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    for (int i = 1; i < RS; ++i) s += red[(i << cs_log2) + cl];
    atomicAdd(out + c, s);
  }
}

// Column sums of a short, wide partial-gradient matrix, scattered into up to 3 outputs (column
// ranges).  The row axis is cut into enough slices for >= ~512 blocks; 4 independent loads per
// thread and iteration; one atomic per (slice, column).
__global__ __launch_bounds__(256) void colsum_seg_kernel(const float* a, long long lda, int rows,
                                                         int rows_per, int ncols,
                                                         f2g_colsegs segs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > rows) r1 = rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += a[(long long)r * lda + c];
    s1 += a[(long long)(r + 1) * lda + c];
    s2 += a[(long long)(r + 2) * lda + c];
    s3 += a[(long long)(r + 3) * lda + c];
  }
  for (; r < r1; ++r) s0 += a[(long long)r * lda + c];
  const float s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (segs.out[k] && c >= segs.begin[k] && c < segs.begin[k] + segs.count[k])
      atomicAdd(segs.out[k] + (c - segs.begin[k]), s);
  }
}

__global__ __launch_bounds__(256) void rows_fold_up_kernel(float* out, long long ldo,
                                                           const float* g, long long ldg, int B,
                                                           int F, int Fc, int up, int C) {
  const long long total = (long long)B * Fc * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const int fc = (int)(row % Fc);
    const int b = (int)(row / Fc);
    float s = 0.f;
    for (int u = 0; u < up; ++u) {
      const int f = fc * up + u;
      if (f < F) s += g[((long long)b * F + f) * ldg + c];
    }
    out[row * ldo + c] += s;
  }
}

// (B, C, F) -> rows (b*F+f) x C through a 32x32 LDS tile (both sides coalesced).
__global__ __launch_bounds__(256) void bct_to_rows_kernel(float* out, long long ldo,
                                                          const float* in, int B, int C, int F) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, f = f0 + tx;
    tile[i][tx] = (c < C && f < F) ? in[((long long)b * C + c) * F + f] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int f = f0 + i, c = c0 + tx;
    if (c < C && f < F) out[((long long)b * F + f) * ldo + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void rows_to_bct_kernel(float* out, const float* in,
                                                          long long ldi, int B, int C, int F) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int f = f0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && f < F) ? in[((long long)b * F + f) * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, f = f0 + tx;
    if (c < C && f < F) out[((long long)b * C + c) * F + f] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void permute4_kernel(float* out, const float* in, int n0, int n1,
                                                       int n2, int n3, long long s0, long long s1,
                                                       long long s2, long long s3) {
  const long long total = (long long)n0 * n1 * n2 * n3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int i3 = (int)(r % n3); r /= n3;
    const int i2 = (int)(r % n2); r /= n2;
    const int i1 = (int)(r % n1); r /= n1;
    const int i0 = (int)r;
    out[i] = in[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
  }
}

// LimitParamValue backward (modules.py:246-256): flip the gradient sign where the parameter is
// outside [lo, hi] and the gradient would push it further out.
__global__ __launch_bounds__(256) void limit_grad_kernel(float* g, const float* p, float lo,
                                                         float hi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gv = g[i];
    const float pv = p[i];
    if (gv > 0.f && pv < lo) gv = -gv;
    if (gv < 0.f && pv > hi) gv = -gv;
    g[i] = gv;
  }
}

__global__ __launch_bounds__(256) void copy3_kernel(float* out, long long so0, long long so1,
                                                    const float* in, long long si0, long long si1,
                                                    int n0, int n1, int n2, int accumulate) {
  const long long total = (long long)n0 * n1 * n2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % n2);
    const long long q = i / n2;
    const int r = (int)(q % n1);
    const int b = (int)(q / n1);
    const float v = in[b * si0 + r * si1 + c];
    float* o = out + b * so0 + r * so1 + c;
    *o = accumulate ? *o + v : v;
  }
}

__global__ __launch_bounds__(256) void spec_power_kernel(float* out, long long ldo,
                                                         const float* packed, long long ldp,
                                                         int rows, int nb, int power) {
  const long long total = (long long)rows * nb;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % nb);
    const long long r = i / nb;
    const float re = packed[r * ldp + k], im = packed[r * ldp + nb + k];
    const float p2 = re * re + im * im;
    out[r * ldo + k] = power == 2 ? p2 : sqrtf(p2);
  }
}

__global__ __launch_bounds__(256) void spec_power_bwd_kernel(float* gpacked, long long ldp,
                                                             const float* gout, long long ldo,
                                                             const float* packed, int rows, int nb,
                                                             int power) {
  const long long total = (long long)rows * nb;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % nb);
    const long long r = i / nb;
    const float re = packed[r * ldp + k], im = packed[r * ldp + nb + k];
    const float g = gout[r * ldo + k];
    float gre, gim;
    if (power == 2) {
      gre = 2.f * g * re;
      gim = 2.f * g * im;
    } else {
      const float mag = sqrtf(re * re + im * im);
      // torch's abs() backward of a complex zero is 0
      const float inv = mag > 0.f ? g / mag : 0.f;
      gre = inv * re;
      gim = inv * im;
    }
    gpacked[r * ldp + k] = gre;
    gpacked[r * ldp + nb + k] = gim;
  }
}

__global__ __launch_bounds__(256) void fm_spec_loss_kernel(float* loss, float* g_err,
                                                           const float* s_err, const float* s_gt,
                                                           int B, int F, int nf, const int* lens,
                                                           float eps, float power, float lo,
                                                           float hi, float inv_denom) {
  __shared__ float sh[4];
  const long long total = (long long)B * F * nf;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / nf;
    const int f = (int)(row % F);
    const int b = (int)(row / F);
    float w = 0.f;
    if (!lens || f < lens[b]) {
      float sc = powf(s_gt[i] + eps, -power);
      sc = fminf(fmaxf(sc, lo), hi);
      w = sc * inv_denom;
    }
    acc += w * s_err[i];
    if (g_err) g_err[i] = w;
  }
  acc = block_sum256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss, acc);
}

__global__ __launch_bounds__(256) void masked_mse_kernel(float* loss, float* g_err, const float* pred,
                                                         const float* ref, int B, int T,
                                                         const int* lens, float inv_denom) {
  __shared__ float sh[4];
  const long long total = (long long)B * T;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / T);
    const int t = (int)(i - (long long)b * T);
    const bool live = !lens || t < lens[b];
    const float e = live ? pred[i] - ref[i] : 0.f;
    acc += e * e;
    if (g_err) g_err[i] = 2.f * inv_denom * e;
  }
  acc = block_sum256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss, acc * inv_denom);
}

__global__ __launch_bounds__(256) void l1_loss_kernel(float* loss, float* gb, const float* a,
                                                      const float* b, int rows, int cols,
                                                      long long ld, float w, float clip,
                                                      const float* wdev) {
  __shared__ float sh[4];
  float acc = 0.f;
  const float wg = w * (wdev ? wdev[0] : 1.f);
  const long long n = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const long long off = r * ld + (i - r * cols);
    float av = a[off], bv = b[off];
    float d, gscale = 1.f;
    if (clip > 0.f) {
      const bool live = bv > clip;  // d/db log(max(b,clip)) = 1/b above the clip, 0 below
      d = logf(fmaxf(av, clip)) - logf(fmaxf(bv, clip));
      gscale = live ? 1.f / bv : 0.f;
    } else {
      d = av - bv;
    }
    acc += fabsf(d);
    if (gb) {
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      gb[off] = -wg * sg * gscale;
    }
  }
  if (loss) {
    acc = block_sum256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss, w * acc);
  }
}

// the same over a CONTIGUOUS span (rows == 1 or ld == cols; n % 4 == 0, 16-byte aligned pointers): 16-byte
// loads, four in flight per thread, no 64-bit division per element (round 5: the scalar kernel above read the
// feature-matching maps at 2.6 TB/s)
__global__ __launch_bounds__(256) void l1_loss4_kernel(float* loss, float* gb, const float4* a, const float4* b,
                                                       long long n4, float w, float clip, const float* wdev) {
  __shared__ float sh[4];
  float acc = 0.f;
  const float wg = w * (wdev ? wdev[0] : 1.f);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    float4 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      av[u] = i < n4 ? a[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[u] = i < n4 ? b[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      if (i >= n4) break;
      const float x[4] = {av[u].x, av[u].y, av[u].z, av[u].w}, y[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dd, gscale = 1.f;
        if (clip > 0.f) {
          dd = logf(fmaxf(x[e], clip)) - logf(fmaxf(y[e], clip));
          gscale = y[e] > clip ? 1.f / y[e] : 0.f;
        } else {
          dd = x[e] - y[e];
        }
        acc += fabsf(dd);
        o[e] = -wg * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f)) * gscale;
      }
      if (gb) reinterpret_cast<float4*>(gb)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (loss) {
    acc = block_sum256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss, w * acc);
  }
}

__global__ __launch_bounds__(256) void hinge_loss_kernel(float* loss, float* gs, const float* s,
                                                         long long n, float sgn, float w,
                                                         const float* wdev) {
  __shared__ float sh[4];
  float acc = 0.f;
  const float wg = w * (wdev ? wdev[0] : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = 1.f + sgn * s[i];
    const bool live = v > 0.f;
    if (live) acc += v;
    if (gs) gs[i] = live ? wg * sgn : 0.f;
  }
  if (loss) {
    acc = block_sum256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss, w * acc);
  }
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(float* g, const float* y_act,
                                                        const float* f_real, float w,
                                                        const float* wdev, float slope, int rows,
                                                        int cols, long long ld) {
  const float wg = w * (wdev ? wdev[0] : 1.f);
  const long long n = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const long long off = r * ld + (i - r * cols);
    const float y = y_act[off];
    float gv = g[off];
    if (f_real) {
      const float d = y - f_real[off];
      gv += wg * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    g[off] = gv * (y > 0.f ? 1.f : slope);
  }
}

// the same over a contiguous span (rows == 1 or ld == cols; n % 4 == 0, aligned): 16-byte accesses, four
// groups in flight per thread, no 64-bit division per element
__global__ __launch_bounds__(256) void lrelu_bwd4_kernel(float4* g, const float4* y_act, const float4* f_real,
                                                         float w, const float* wdev, float slope, long long n4) {
  const float wg = w * (wdev ? wdev[0] : 1.f);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    float4 yv[4], gv[4], fv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      const bool on = i < n4;
      yv[u] = on ? y_act[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      gv[u] = on ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      fv[u] = (on && f_real) ? f_real[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      if (i >= n4) break;
      const float y[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, f[4] = {fv[u].x, fv[u].y, fv[u].z, fv[u].w};
      float o[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (f_real) {
          const float dd = y[e] - f[e];
          o[e] += wg * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
        }
        o[e] *= y[e] > 0.f ? 1.f : slope;
      }
      g[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// Leaky-ReLU backward of a (rows, C) map in place, g = (g + w*sign(y - f_real)) * lrelu'(y), fused
// with the column sums of the result (= the bias gradient of the conv that produced y): one pass
// over g instead of two.  Block = CS4 float4 column groups x (256/CS4) row lanes.
__global__ __launch_bounds__(256) void lrelu_bwd_cs_kernel(float* g, const float* y_act,
                                                           const float* f_real, float w,
                                                           const float* wdev, float slope,
                                                           int rows, int C4, long long ld4,
                                                           int rows_per, int cs_log2,
                                                           float* colsum) {
  __shared__ float4 red[256];
  const float wg = w * (wdev ? wdev[0] : 1.f);
  const int CS = 1 << cs_log2, RS = 256 >> cs_log2;
  const int cl = threadIdx.x & (CS - 1), rl = threadIdx.x >> cs_log2;
  const int c = blockIdx.x * CS + cl;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C4) {
    float4* g4 = reinterpret_cast<float4*>(g);
    const float4* y4 = reinterpret_cast<const float4*>(y_act);
    const float4* f4 = reinterpret_cast<const float4*>(f_real);
    auto upd = [&](float4 gv, const float4 y, const float4 f) {
      if (f_real) {
        const float dx = y.x - f.x, dy = y.y - f.y, dz = y.z - f.z, dw = y.w - f.w;
        gv.x += wg * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f));
        gv.y += wg * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f));
        gv.z += wg * (dz > 0.f ? 1.f : (dz < 0.f ? -1.f : 0.f));
        gv.w += wg * (dw > 0.f ? 1.f : (dw < 0.f ? -1.f : 0.f));
      }
      gv.x *= y.x > 0.f ? 1.f : slope; gv.y *= y.y > 0.f ? 1.f : slope;
      gv.z *= y.z > 0.f ? 1.f : slope; gv.w *= y.w > 0.f ? 1.f : slope;
      s.x += gv.x; s.y += gv.y; s.z += gv.z; s.w += gv.w;
      return gv;
    };
    // four rows per iteration, every load issued before anything is consumed (these kernels are a
    // memory-level-parallelism problem)
    int r = r0 + rl;
    for (; r + 3 * RS < r1; r += 4 * RS) {
      long long off[4];
      float4 y[4], gv[4], f[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) off[k] = (long long)(r + k * RS) * ld4 + c;
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[k] = y4[off[k]]; gv[k] = g4[off[k]]; }
      if (f_real) {
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = f4[off[k]];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) g4[off[k]] = upd(gv[k], y[k], f[k]);
    }
    for (; r < r1; r += RS) {
      const long long off = (long long)r * ld4 + c;
      const float4 f = f_real ? f4[off] : make_float4(0.f, 0.f, 0.f, 0.f);
      g4[off] = upd(g4[off], y4[off], f);
    }
  }
  if (!colsum) return;
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C4) {
    for (int i = 1; i < RS; ++i) {
      const float4 t = red[(i << cs_log2) + cl];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    atomicAdd(colsum + 4 * c, s.x); atomicAdd(colsum + 4 * c + 1, s.y);
    atomicAdd(colsum + 4 * c + 2, s.z); atomicAdd(colsum + 4 * c + 3, s.w);
  }
}

// rows [0, lo) and [rows_per_seq - hi, rows_per_seq) of every sequence := 0 (float4 lanes)
__global__ __launch_bounds__(256) void zero_halo_kernel(float* buf, int nseq, int rows_per_seq,
                                                        int C4, int lo, int hi) {
  const int per = (lo + hi) * C4;
  const long long total = (long long)nseq * per;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int sq = (int)(i / per);
    int r = (int)(i - (long long)sq * per);
    const int hr = r / C4, c = r - hr * C4;
    const int row = hr < lo ? hr : rows_per_seq - hi + (hr - lo);
    reinterpret_cast<float4*>(buf)[((long long)sq * rows_per_seq + row) * C4 + c] =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int f2g_lrelu_bwd_colsum(float* g, const float* y_act, const float* f_real, float w,
                                    const float* wdev, float slope, int32_t rows, int32_t C,
                                    int64_t ld, float* colsum, f2g_stream_t stream) {
  if (!g || !y_act || (C & 3) || (ld & 3)) return F2G_EINVAL;
  if (rows <= 0 || C <= 0) return F2G_OK;
  const int C4 = C / 4;
  int cs_log2 = 0;
  while ((1 << cs_log2) < C4 && cs_log2 < 8) ++cs_log2;
  const int CS = 1 << cs_log2, RS = 256 >> cs_log2;
  int rows_per = 32 * RS;  // 32 rows per thread
  if (rows_per > rows) rows_per = rows;
  dim3 grid((C4 + CS - 1) / CS, (rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(lrelu_bwd_cs_kernel, grid, dim3(256), 0, ST, g, y_act, f_real, w, wdev, slope,
                     rows, C4, (long long)(ld / 4), rows_per, cs_log2, colsum);
  return f2g_check_launch();
}

extern "C" int f2g_zero_halo(float* buf, int32_t nseq, int32_t rows_per_seq, int32_t C, int32_t lo,
                             int32_t hi, f2g_stream_t stream) {
  if (!buf || (C & 3) || lo < 0 || hi < 0 || lo + hi > rows_per_seq) return F2G_EINVAL;
  if (nseq <= 0 || lo + hi == 0) return F2G_OK;
  hipLaunchKernelGGL(zero_halo_kernel, dim3(f2g_grid_for((int64_t)nseq * (lo + hi) * (C / 4), 256)),
                     dim3(256), 0, ST, buf, nseq, rows_per_seq, C / 4, lo, hi);
  return f2g_check_launch();
}

extern "C" int f2g_axpby_rows(float* y, const float* x0, const float* x1, const float* ca,
                              const float* cb, float sa, float sb, int32_t rows, int32_t cols,
                              f2g_stream_t stream) {
  if (!y || !x0) return F2G_EINVAL;
  if (rows <= 0 || cols <= 0) return F2G_OK;
  hipLaunchKernelGGL(axpby_rows_kernel, dim3(f2g_grid_for((int64_t)rows * cols, 256)), dim3(256),
                     0, ST, y, x0, x1, ca, cb, sa, sb, rows, cols);
  return f2g_check_launch();
}

extern "C" int f2g_clamp(float* y, const float* x, float lo, float hi, int64_t n,
                         f2g_stream_t stream) {
  if (!y || !x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(clamp_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, y, x, lo, hi,
                     (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_log_clip(float* x, int64_t n, float clip, f2g_stream_t stream) {
  if (!x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(log_clip_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, x,
                     (long long)n, clip);
  return f2g_check_launch();
}

extern "C" int f2g_fill(float* x, float v, int64_t n, f2g_stream_t stream) {
  if (!x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, x, v,
                     (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_bucket_arm(float* flat, int64_t n, int32_t nflags, f2g_stream_t stream) {
  if (!flat || n < 0 || nflags < 0 || (((uintptr_t)flat) & 15)) return F2G_EINVAL;
  if (n + nflags == 0) return F2G_OK;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(bucket_arm_kernel, dim3(f2g_grid_for(n4 > nflags ? n4 : nflags, 256)), dim3(256), 0, ST,
                     reinterpret_cast<float4*>(flat), n4, flat + 4 * n4, (int)(n - 4 * n4), flat + n, nflags);
  return f2g_check_launch();
}

extern "C" int f2g_scale(float* x, float s, int64_t n, f2g_stream_t stream) {
  if (!x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, x, s, (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_silu(float* y, const float* x, int64_t n, f2g_stream_t stream) {
  if (!y || !x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(silu_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, y, x,
                     (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_silu_bwd(float* gx, const float* gy, const float* x, int64_t n,
                            f2g_stream_t stream) {
  if (!gx || !gy || !x) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, gx, gy, x,
                     (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_time_embedding(float* out, const float* t, int32_t B, int32_t dim, float scale,
                                  f2g_stream_t stream) {
  if (!out || !t || dim < 4 || (dim & 1)) return F2G_EINVAL;
  if (B <= 0) return F2G_OK;
  hipLaunchKernelGGL(time_embedding_kernel, dim3(f2g_grid_for((int64_t)B * dim / 2, 256)),
                     dim3(256), 0, ST, out, t, B, dim, scale);
  return f2g_check_launch();
}

extern "C" int f2g_mask_rows(float* x, int64_t ld, int32_t B, int32_t F, int32_t C,
                             const int32_t* lens, f2g_stream_t stream) {
  if (!x || !lens) return F2G_EINVAL;
  if (B <= 0 || F <= 0 || C <= 0) return F2G_OK;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(f2g_grid_for((int64_t)B * F * C, 256)), dim3(256), 0,
                     ST, x, (long long)ld, B, F, C, lens);
  return f2g_check_launch();
}

extern "C" int f2g_colsum(float* out, const float* a, int64_t lda, const float* b, int64_t ldb,
                          int32_t rows, int32_t cols, f2g_stream_t stream) {
  if (!out || !a) return F2G_EINVAL;
  if (rows <= 0 || cols <= 0) return F2G_OK;
  int cs_log2 = 0;
  while ((1 << cs_log2) < cols && cs_log2 < 8) ++cs_log2;
  const int CS = 1 << cs_log2, RS = 256 >> cs_log2;
  int rows_per = 64 * RS;  // 64 rows per thread
  if (rows_per > rows) rows_per = rows;
  dim3 grid((cols + CS - 1) / CS, (rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, ST, out, a, (long long)lda, b,
                     (long long)ldb, rows, cols, rows_per, cs_log2);
  return f2g_check_launch();
}

int f2g_colsum_segments(const float* a, long long lda, int rows, int ncols,
                        const f2g_colsegs& segs, hipStream_t st) {
  if (rows <= 0 || ncols <= 0) return F2G_OK;
  const int cb = (ncols + 255) / 256;
  int slices = (512 + cb - 1) / cb;
  if (slices > rows) slices = rows;
  const int rows_per = (rows + slices - 1) / slices;
  dim3 grid(cb, (rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(colsum_seg_kernel, grid, dim3(256), 0, st, a, lda, rows, rows_per, ncols,
                     segs);
  return f2g_check_launch();
}

extern "C" int f2g_rows_fold_up(float* out, int64_t ldo, const float* g, int64_t ldg, int32_t B,
                                int32_t F, int32_t Fc, int32_t up, int32_t C,
                                f2g_stream_t stream) {
  if (!out || !g || up < 1) return F2G_EINVAL;
  if (B <= 0 || Fc <= 0 || C <= 0) return F2G_OK;
  hipLaunchKernelGGL(rows_fold_up_kernel, dim3(f2g_grid_for((int64_t)B * Fc * C, 256)), dim3(256),
                     0, ST, out, (long long)ldo, g, (long long)ldg, B, F, Fc, up, C);
  return f2g_check_launch();
}

extern "C" int f2g_bct_to_rows(float* out, int64_t ldo, const float* in, int32_t B, int32_t C,
                               int32_t F, f2g_stream_t stream) {
  if (!out || !in) return F2G_EINVAL;
  if (B <= 0 || C <= 0 || F <= 0) return F2G_OK;
  dim3 grid((F + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(bct_to_rows_kernel, grid, dim3(256), 0, ST, out, (long long)ldo, in, B, C, F);
  return f2g_check_launch();
}

extern "C" int f2g_rows_to_bct(float* out, const float* in, int64_t ldi, int32_t B, int32_t C,
                               int32_t F, f2g_stream_t stream) {
  if (!out || !in) return F2G_EINVAL;
  if (B <= 0 || C <= 0 || F <= 0) return F2G_OK;
  dim3 grid((F + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(rows_to_bct_kernel, grid, dim3(256), 0, ST, out, in, (long long)ldi, B, C, F);
  return f2g_check_launch();
}

extern "C" int f2g_permute4(float* out, const float* in, int32_t n0, int32_t n1, int32_t n2,
                            int32_t n3, int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                            f2g_stream_t stream) {
  if (!out || !in) return F2G_EINVAL;
  const int64_t total = (int64_t)n0 * n1 * n2 * n3;
  if (total <= 0) return F2G_OK;
  hipLaunchKernelGGL(permute4_kernel, dim3(f2g_grid_for(total, 256)), dim3(256), 0, ST, out, in, n0,
                     n1, n2, n3, (long long)s0, (long long)s1, (long long)s2, (long long)s3);
  return f2g_check_launch();
}

extern "C" int f2g_limit_grad(float* g, const float* p, float lo, float hi, int64_t n,
                              f2g_stream_t stream) {
  if (!g || !p) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(limit_grad_kernel, dim3(f2g_grid_for(n, 256)), dim3(256), 0, ST, g, p, lo, hi,
                     (long long)n);
  return f2g_check_launch();
}

extern "C" int f2g_copy3(float* out, int64_t so0, int64_t so1, const float* in, int64_t si0,
                         int64_t si1, int32_t n0, int32_t n1, int32_t n2, int32_t accumulate,
                         f2g_stream_t stream) {
  if (!out || !in) return F2G_EINVAL;
  const int64_t total = (int64_t)n0 * n1 * n2;
  if (total <= 0) return F2G_OK;
  hipLaunchKernelGGL(copy3_kernel, dim3(f2g_grid_for(total, 256)), dim3(256), 0, ST, out,
                     (long long)so0, (long long)so1, in, (long long)si0, (long long)si1, n0, n1, n2,
                     accumulate);
  return f2g_check_launch();
}

extern "C" int f2g_spec_power(float* out, int64_t ldo, const float* packed, int64_t ldp,
                              int32_t rows, int32_t nb, int32_t power, f2g_stream_t stream) {
  if (!out || !packed || (power != 1 && power != 2)) return F2G_EINVAL;
  if (rows <= 0 || nb <= 0) return F2G_OK;
  hipLaunchKernelGGL(spec_power_kernel, dim3(f2g_grid_for((int64_t)rows * nb, 256)), dim3(256), 0,
                     ST, out, (long long)ldo, packed, (long long)ldp, rows, nb, power);
  return f2g_check_launch();
}

extern "C" int f2g_spec_power_bwd(float* gpacked, int64_t ldp, const float* gout, int64_t ldo,
                                  const float* packed, int32_t rows, int32_t nb, int32_t power,
                                  f2g_stream_t stream) {
  if (!gpacked || !gout || !packed || (power != 1 && power != 2)) return F2G_EINVAL;
  if (rows <= 0 || nb <= 0) return F2G_OK;
  hipLaunchKernelGGL(spec_power_bwd_kernel, dim3(f2g_grid_for((int64_t)rows * nb, 256)), dim3(256),
                     0, ST, gpacked, (long long)ldp, gout, (long long)ldo, packed, rows, nb, power);
  return f2g_check_launch();
}

extern "C" int f2g_fm_spec_loss(float* loss, float* g_err, const float* s_err, const float* s_gt,
                                int32_t B, int32_t F, int32_t n_filt, const int32_t* lens,
                                float eps, float power, float lo, float hi, float inv_denom,
                                f2g_stream_t stream) {
  if (!loss || !s_err || !s_gt) return F2G_EINVAL;
  if (B <= 0 || F <= 0 || n_filt <= 0) return F2G_OK;
  hipLaunchKernelGGL(fm_spec_loss_kernel, dim3(f2g_grid_for((int64_t)B * F * n_filt, 256, 1024)),
                     dim3(256), 0, ST, loss, g_err, s_err, s_gt, B, F, n_filt, lens, eps, power, lo,
                     hi, inv_denom);
  return f2g_check_launch();
}

extern "C" int f2g_masked_mse(float* loss, float* g_err, const float* pred, const float* ref,
                              int32_t B, int32_t T, const int32_t* lens, float inv_denom,
                              f2g_stream_t stream) {
  if (!loss || !pred || !ref) return F2G_EINVAL;
  if (B <= 0 || T <= 0) return F2G_OK;
  hipLaunchKernelGGL(masked_mse_kernel, dim3(f2g_grid_for((int64_t)B * T, 256, 1024)), dim3(256), 0,
                     ST, loss, g_err, pred, ref, B, T, lens, inv_denom);
  return f2g_check_launch();
}

extern "C" int f2g_l1_loss(float* loss, float* gb, const float* a, const float* b, int32_t rows,
                           int32_t cols, int64_t ld, float w, float clip, const float* wdev,
                           f2g_stream_t stream) {
  if (!a || !b || (!loss && !gb)) return F2G_EINVAL;
  if (rows <= 0 || cols <= 0) return F2G_OK;
  const long long n = (long long)rows * cols;
  if ((rows == 1 || ld == cols) && !(n & 3) && !((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)gb)) & 15)) {
    hipLaunchKernelGGL(l1_loss4_kernel, dim3(f2g_grid_for(n / 4, 256 * 4, 2048)), dim3(256), 0, ST, loss, gb,
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), n / 4, w, clip, wdev);
    return f2g_check_launch();
  }
  hipLaunchKernelGGL(l1_loss_kernel, dim3(f2g_grid_for((int64_t)rows * cols, 256, 1024)),
                     dim3(256), 0, ST, loss, gb, a, b, rows, cols, (long long)ld, w, clip, wdev);
  return f2g_check_launch();
}

extern "C" int f2g_hinge_loss(float* loss, float* gs, const float* s, int64_t n, float sgn, float w,
                              const float* wdev, f2g_stream_t stream) {
  if (!s || (!loss && !gs)) return F2G_EINVAL;
  if (n <= 0) return F2G_OK;
  hipLaunchKernelGGL(hinge_loss_kernel, dim3(f2g_grid_for(n, 256, 1024)), dim3(256), 0, ST, loss,
                     gs, s, (long long)n, sgn, w, wdev);
  return f2g_check_launch();
}

extern "C" int f2g_lrelu_bwd(float* g, const float* y_act, const float* f_real, float w,
                             const float* wdev, float slope, int32_t rows, int32_t cols,
                             int64_t ld, f2g_stream_t stream) {
  if (!g || !y_act) return F2G_EINVAL;
  if (rows <= 0 || cols <= 0) return F2G_OK;
  const long long n = (long long)rows * cols;
  if ((rows == 1 || ld == cols) && !(n & 3) && !((((uintptr_t)g) | ((uintptr_t)y_act) | ((uintptr_t)f_real)) & 15)) {
    hipLaunchKernelGGL(lrelu_bwd4_kernel, dim3(f2g_grid_for(n / 4, 256 * 4, 2048)), dim3(256), 0, ST,
                       reinterpret_cast<float4*>(g), reinterpret_cast<const float4*>(y_act),
                       reinterpret_cast<const float4*>(f_real), w, wdev, slope, n / 4);
    return f2g_check_launch();
  }
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(f2g_grid_for((int64_t)rows * cols, 256)), dim3(256), 0,
                     ST, g, y_act, f_real, w, wdev, slope, rows, cols, (long long)ld);
  return f2g_check_launch();
}
