// HBM-bound kernels of the ConvNeXt block (reference modules.py:286-416, 419-495):
// depthwise k<=7 conv + BiasNorm + condition add + time scale, fused, forward and backward.
//
// Layout: channels-last rows (b*F + f) x C.  A block owns TF=16 consecutive frames of one batch
// item; the (TF+6) x C input strip is staged once in LDS (halo rows re-read from L2, not HBM),
// then each wave walks 4 frames with lanes striding the channel axis, so the per-frame channel
// reduction of BiasNorm is a 6-step wave shuffle and every global access is a coalesced row.
#include "common.h"

namespace {

constexpr int TF = 16;    // frames per block
constexpr int HALO = 3;   // (7-1)/2
constexpr int CPL = 12;   // channels per lane: C <= 768

__device__ __forceinline__ void stage_strip(float* xs, const float* x, long long ldx, int b, int F,
                                            int C, int f0, int nrows, int len_b) {
  // xs[(i)*C + c] = x[b, f0 - HALO + i, c] * mask, zero outside [0, min(F, len)).
  const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  if (vec) {
    const int c4 = C >> 2;
    for (int idx = threadIdx.x; idx < nrows * c4; idx += blockDim.x) {
      int i = idx / c4, q = idx - i * c4;
      int f = f0 - HALO + i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f >= 0 && f < F && f < len_b)
        v = *reinterpret_cast<const float4*>(x + ((long long)b * F + f) * ldx + q * 4);
      *reinterpret_cast<float4*>(xs + i * C + q * 4) = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < nrows * C; idx += blockDim.x) {
      int i = idx / C, c = idx - i * C;
      int f = f0 - HALO + i;
      float v = 0.f;
      if (f >= 0 && f < F && f < len_b) v = x[((long long)b * F + f) * ldx + c];
      xs[i * C + c] = v;
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void dwnorm_kernel(const f2g_dwnorm_bwd_desc D) {
  const f2g_dwnorm_fwd_desc& P = D.f;
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int b = blockIdx.y, f0 = blockIdx.x * TF;
  const int F = P.F, C = P.C, K = P.K;
  const int len_b = P.lens ? P.lens[b] : F;
  stage_strip(xs, P.x, P.ldx, b, F, C, f0, TF + 2 * HALO, len_b);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int koff = (7 - K) / 2;
  float w[CPL][7], bdw[CPL], beta[CPL], te1[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
#pragma unroll
    for (int j = 0; j < 7; ++j) w[k][j] = 0.f;
    bdw[k] = 0.f; beta[k] = 0.f; te1[k] = 1.f;
    if (c < C) {
      for (int j = 0; j < K; ++j) w[k][j + koff] = P.w_dw[c * K + j];
      bdw[k] = P.b_dw ? P.b_dw[c] : 0.f;
      beta[k] = P.beta[c];
      if (P.te) te1[k] = 1.f + P.te[(long long)b * P.ldte + c];
    }
  }
  const float escale = expf(P.log_scale[0]);
  const float invC = 1.f / (float)C;

  float gbeta[CPL], gte[CPL], gcp[CPL];
  float glam = 0.f;
  if (BWD) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) { gbeta[k] = 0.f; gte[k] = 0.f; gcp[k] = 0.f; }
  }

  for (int i = 0; i < 4; ++i) {
    const int fl = wave * 4 + i;  // local frame
    const int f = f0 + fl;
    if (f >= F) break;
    const long long row = (long long)b * F + f;
    float u[CPL];
    float ssq = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      u[k] = 0.f;
      if (c < C) {
        float acc = bdw[k];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += w[k][j] * xs[(fl + j) * C + c];
        u[k] = acc;
        const float dlt = acc - beta[k];
        ssq += dlt * dlt;
      }
    }
    ssq = wave_sum(ssq);
    const float r = ssq * invC;
    const float s = escale / sqrtf(r);
    const int fc = P.cproj ? f / P.up : 0;
    const bool has_cp = P.cproj && fc < P.Fc;
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          float v = u[k] * s;
          if (has_cp) v += P.cproj[((long long)b * P.Fc + fc) * P.ldcp + c];
          P.z[row * P.ldz + c] = v * te1[k];
        }
      }
      if (P.rstd && lane == 0) P.rstd[row] = s;
    } else {
      float gv[CPL];
      float dsum = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        gv[k] = 0.f;
        if (c < C) {
          const float g = D.gz[row * D.ldgz + c];
          gv[k] = g * te1[k];
          dsum += gv[k] * u[k];
          if (D.g_te) {
            float v = u[k] * s;
            if (has_cp) v += P.cproj[((long long)b * P.Fc + fc) * P.ldcp + c];
            gte[k] += g * v;
          }
          gcp[k] += gv[k];
        }
      }
      dsum = wave_sum(dsum);
      const float coef = s * dsum / ((float)C * r);
      glam += s * dsum;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          const float t = coef * (u[k] - beta[k]);
          gbeta[k] += t;
          D.du[row * D.lddu + c] = s * gv[k] - t;
        }
      }
      // condition gradient: frames sharing one condition row (f/up) are summed in-wave (the wave's
      // 4 frames start at a multiple of 4 >= up), then added to the exclusively-owned slot.
      const bool group_end = ((f + 1) % P.up == 0) || (f == F - 1) || (i == 3);
      if (D.g_cproj && group_end) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (c < C && has_cp) D.g_cproj[((long long)b * P.Fc + fc) * P.ldcp + c] += gcp[k];
          gcp[k] = 0.f;
        }
      }
    }
  }
  if (BWD) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        if (D.g_beta) atomicAdd(D.g_beta + c, gbeta[k]);
        if (D.g_te) atomicAdd(D.g_te + (long long)b * P.ldte + c, gte[k]);
      }
    }
    if (D.g_log_scale && lane == 0) atomicAdd(D.g_log_scale, glam);
  }
}

// dx / dw / db / dgamma of the depthwise conv: thread = channel, register sliding window over a
// strip of TFB frames (halo 6 -> 1.09x reads), partial parameter gradients reduced by atomics.
constexpr int TFB = 64;

__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const f2g_dwconv_bwd_desc P) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.z;
  const int fs = blockIdx.y * TFB;
  const int F = P.F, C = P.C, K = P.K;
  if (c >= C) return;
  int fe = fs + TFB;
  if (fe > F) fe = F;
  const int len_b = P.lens ? P.lens[b] : F;
  const int koff = (7 - K) / 2;
  float w[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) w[j] = 0.f;
  for (int j = 0; j < K; ++j) w[j + koff] = P.w_dw[c * K + j];
  const float gam = P.gamma ? P.gamma[c] : 1.f;
  const long long rb = (long long)b * F;

  auto ld_du = [&](int f) -> float {
    return (f >= 0 && f < F) ? P.du[(rb + f) * P.lddu + c] : 0.f;
  };
  auto ld_xm = [&](int f) -> float {
    return (f >= 0 && f < F && f < len_b) ? P.x[(rb + f) * P.ldx + c] : 0.f;
  };
  float dw_[7], xw[7];
#pragma unroll
  for (int j = 0; j < 6; ++j) { dw_[j] = ld_du(fs - 3 + j); xw[j] = ld_xm(fs - 3 + j); }
  float gw[7], gb = 0.f, gg = 0.f;
#pragma unroll
  for (int j = 0; j < 7; ++j) gw[j] = 0.f;

  for (int f = fs; f < fe; ++f) {
    dw_[6] = ld_du(f + 3);
    xw[6] = ld_xm(f + 3);
    float dx = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) dx += w[j] * dw_[6 - j];
    if (f >= len_b) dx = 0.f;
    if (P.gres) {
      const float gr = P.gres[(rb + f) * P.ldgres + c];
      dx += gam * gr;
      if (P.g_gamma) gg += gr * P.x[(rb + f) * P.ldx + c];
    }
    P.gx[(rb + f) * P.ldgx + c] = dx;
    const float duc = dw_[3];
    gb += duc;
#pragma unroll
    for (int j = 0; j < 7; ++j) gw[j] += duc * xw[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) { dw_[j] = dw_[j + 1]; xw[j] = xw[j + 1]; }
  }
  if (P.g_w)
    for (int j = 0; j < K; ++j) atomicAdd(P.g_w + c * K + j, gw[j + koff]);
  if (P.g_b) atomicAdd(P.g_b + c, gb);
  if (P.g_gamma && P.gres) atomicAdd(P.g_gamma + c, gg);
}

// BiasNorm alone (decoder.in_norm / cond_encoder.in_norm): one wave per row.
__global__ __launch_bounds__(256) void biasnorm_fwd_kernel(const float* x, long long ldx, float* y,
                                                           long long ldy, int rows, int C,
                                                           const float* beta,
                                                           const float* log_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long long)row * ldx;
  float ssq = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - beta[c];
    ssq += d * d;
  }
  ssq = wave_sum(ssq);
  const float s = expf(log_scale[0]) / sqrtf(ssq / (float)C);
  float* yr = y + (long long)row * ldy;
  for (int c = lane; c < C; c += 64) yr[c] = xr[c] * s;
}

constexpr int BN_ROWS = 8;  // rows per wave in backward (register partials for d(beta))

__global__ __launch_bounds__(256) void biasnorm_bwd_kernel(const float* x, long long ldx,
                                                           const float* gy, long long ldgy,
                                                           float* gx, long long ldgx, int rows,
                                                           int C, const float* beta,
                                                           const float* log_scale, float* g_beta,
                                                           float* g_log_scale) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * BN_ROWS;
  const float es = expf(log_scale[0]);
  float gb[CPL], bt[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    gb[k] = 0.f;
    const int c = lane + 64 * k;
    bt[k] = c < C ? beta[c] : 0.f;
  }
  float glam = 0.f;
  for (int i = 0; i < BN_ROWS; ++i) {
    const int row = row0 + i;
    if (row >= rows) break;
    const float* xr = x + (long long)row * ldx;
    const float* gr = gy + (long long)row * ldgy;
    float xv[CPL], gv[CPL];
    float ssq = 0.f, dsum = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      xv[k] = 0.f; gv[k] = 0.f;
      if (c < C) {
        xv[k] = xr[c];
        gv[k] = gr[c];
        const float d = xv[k] - bt[k];
        ssq += d * d;
        dsum += gv[k] * xv[k];
      }
    }
    ssq = wave_sum(ssq);
    dsum = wave_sum(dsum);
    const float r = ssq / (float)C;
    const float s = es / sqrtf(r);
    const float coef = s * dsum / ((float)C * r);
    glam += s * dsum;
    float* gxr = gx + (long long)row * ldgx;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float t = coef * (xv[k] - bt[k]);
        gb[k] += t;
        gxr[c] = s * gv[k] - t;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    if (c < C && g_beta) atomicAdd(g_beta + c, gb[k]);
  }
  if (g_log_scale && lane == 0) atomicAdd(g_log_scale, glam);
}

int check_dw(const f2g_dwnorm_fwd_desc& f) {
  if (!f.x || !f.w_dw || !f.beta || !f.log_scale) return F2G_EINVAL;
  if (f.C < 1 || f.C > 64 * CPL || f.K < 1 || f.K > 7 || !(f.K & 1)) return F2G_EINVAL;
  if (f.cproj && (f.up < 1 || f.up > 4 || (4 % f.up) != 0)) return F2G_EINVAL;
  return F2G_OK;
}

template <bool BWD>
int launch_dwnorm(const f2g_dwnorm_bwd_desc& d, hipStream_t st) {
  const f2g_dwnorm_fwd_desc& f = d.f;
  if (f.B <= 0 || f.F <= 0) return F2G_OK;
  size_t smem = (size_t)(TF + 2 * HALO) * f.C * sizeof(float);
  auto kern = dwnorm_kernel<BWD>;
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 22 * 64 * CPL * 4);
    done = true;
  }
  dim3 grid((f.F + TF - 1) / TF, f.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, d);
  return f2g_check_launch();
}

}  // namespace

extern "C" int f2g_dwnorm_fwd(const f2g_dwnorm_fwd_desc* d, f2g_stream_t stream) {
  if (!d || !d->z) return F2G_EINVAL;
  int rc = check_dw(*d);
  if (rc) return rc;
  f2g_dwnorm_bwd_desc full = {};
  full.f = *d;
  return launch_dwnorm<false>(full, (hipStream_t)stream);
}

extern "C" int f2g_dwnorm_bwd(const f2g_dwnorm_bwd_desc* d, f2g_stream_t stream) {
  if (!d || !d->gz || !d->du) return F2G_EINVAL;
  int rc = check_dw(d->f);
  if (rc) return rc;
  return launch_dwnorm<true>(*d, (hipStream_t)stream);
}

extern "C" int f2g_dwconv_bwd(const f2g_dwconv_bwd_desc* d, f2g_stream_t stream) {
  if (!d || !d->du || !d->x || !d->gx || !d->w_dw) return F2G_EINVAL;
  if (d->K < 1 || d->K > 7 || !(d->K & 1)) return F2G_EINVAL;
  if (d->B <= 0 || d->F <= 0 || d->C <= 0) return F2G_OK;
  dim3 grid((d->C + 255) / 256, (d->F + TFB - 1) / TFB, d->B);
  hipLaunchKernelGGL(dwconv_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
  return f2g_check_launch();
}

extern "C" int f2g_biasnorm_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows,
                                int32_t C, const float* beta, const float* log_scale,
                                f2g_stream_t stream) {
  if (!x || !y || !beta || !log_scale || C < 1) return F2G_EINVAL;
  if (rows <= 0) return F2G_OK;
  hipLaunchKernelGGL(biasnorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     x, (long long)ldx, y, (long long)ldy, rows, C, beta, log_scale);
  return f2g_check_launch();
}

extern "C" int f2g_biasnorm_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                                float* gx, int64_t ldgx, int32_t rows, int32_t C,
                                const float* beta, const float* log_scale, float* g_beta,
                                float* g_log_scale, f2g_stream_t stream) {
  if (!x || !gy || !gx || !beta || !log_scale || C < 1 || C > 64 * CPL) return F2G_EINVAL;
  if (rows <= 0) return F2G_OK;
  int per_block = 4 * BN_ROWS;
  hipLaunchKernelGGL(biasnorm_bwd_kernel, dim3((rows + per_block - 1) / per_block), dim3(256), 0,
                     (hipStream_t)stream, x, (long long)ldx, gy, (long long)ldgy, gx,
                     (long long)ldgx, rows, C, beta, log_scale, g_beta, g_log_scale);
  return f2g_check_launch();
}
