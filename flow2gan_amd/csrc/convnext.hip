// HBM-bound kernels of the ConvNeXt block (reference modules.py:286-416, 419-495):
// depthwise k<=7 conv + BiasNorm + condition add + time scale, fused, forward and backward.
//
// Layout: channels-last rows (b*F + f) x C.  One WAVE owns 2 (or 4) consecutive frames; lanes
// stride the channel axis, so the per-frame channel reduction of BiasNorm is a 6-step wave
// shuffle and every global access is a coalesced 256-byte row; the 7-tap halo re-reads are L1/L2
// hits (HBM sees each row once).  No LDS, no barriers, thousands of independent waves.
#include "common.h"

namespace {

constexpr int HALO = 3;   // (7-1)/2
constexpr int CPL = 12;   // channels per lane: C <= 768

// One wave owns FW consecutive frames of one batch item; lanes stride the channel axis
// (c = lane + 64k).  No LDS, no barriers: the (FW+6) x C input window is read straight from
// global memory as coalesced 256-byte rows (the 7x tap reuse is served by L1/L2, HBM sees each row
// once), channel chunk by channel chunk, so a lane keeps (FW+6) independent loads in flight per
// chunk and only u[FW][CPL] lives across the BiasNorm reduction (a 6-step wave shuffle).
template <bool BWD, int FW>
__global__ __launch_bounds__(256) void dwnorm_kernel(const f2g_dwnorm_bwd_desc D) {
  const f2g_dwnorm_fwd_desc& P = D.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = P.F, C = P.C, K = P.K;
  const int groups = (F + FW - 1) / FW;           // frame groups per batch item
  // grid: x = blocks of 4 groups inside one batch item, y = batch item
  const int b = blockIdx.y;
  const int grp = blockIdx.x * 4 + wave;
  const bool live = grp < groups;                 // idle waves still take part in the block reduce
  const int f0 = (live ? grp : 0) * FW;
  const int len_b = P.lens ? P.lens[b] : F;
  const int koff = (7 - K) / 2;
  const int nk = (C + 63) >> 6;
  const float escale = expf(P.log_scale[0]);
  const float invC = 1.f / (float)C;
  const long long rb = (long long)b * F;

  float u[FW][CPL];
  float ssq[FW];
#pragma unroll
  for (int i = 0; i < FW; ++i) ssq[i] = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
#pragma unroll
    for (int i = 0; i < FW; ++i) u[i][k] = 0.f;
    const int c = lane + 64 * k;
    if (k < nk && c < C) {
      float w[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) w[j] = 0.f;
      for (int j = 0; j < K; ++j) w[j + koff] = P.w_dw[c * K + j];
      const float bdw = P.b_dw ? P.b_dw[c] : 0.f;
      const float bt = P.beta[c];
      float xr[FW + 6];
#pragma unroll
      for (int r = 0; r < FW + 6; ++r) {
        const int f = f0 - HALO + r;
        xr[r] = (f >= 0 && f < F && f < len_b) ? P.x[(rb + f) * P.ldx + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < FW; ++i) {
        float acc = bdw;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += w[j] * xr[i + j];
        u[i][k] = acc;
        const float dlt = acc - bt;
        ssq[i] += dlt * dlt;
      }
    }
  }
  float gcp[CPL], gbeta[CPL], gte[CPL];
  float glam = 0.f;
  if (BWD) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) { gcp[k] = 0.f; gbeta[k] = 0.f; gte[k] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    const int f = f0 + i;
    if (f >= F || !live) break;
    const long long row = rb + f;
    const float r = wave_sum(ssq[i]) * invC;
    const float s = escale / sqrtf(r);
    const int fc = P.cproj ? f / P.up : 0;
    const bool has_cp = P.cproj && fc < P.Fc;
    const float* cprow = has_cp ? P.cproj + ((long long)b * P.Fc + fc) * P.ldcp : nullptr;
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (k < nk && c < C) {
          float v = u[i][k] * s;
          if (has_cp) v += cprow[c];
          const float te1 = P.te ? 1.f + P.te[(long long)b * P.ldte + c] : 1.f;
          P.z[row * P.ldz + c] = v * te1;
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
        if (k < nk && c < C) {
          const float g = D.gz[row * D.ldgz + c];
          const float te1 = P.te ? 1.f + P.te[(long long)b * P.ldte + c] : 1.f;
          gv[k] = g * te1;
          dsum += gv[k] * u[i][k];
          if (D.g_te) {
            float v = u[i][k] * s;
            if (has_cp) v += cprow[c];
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
        if (k < nk && c < C) {
          const float t = coef * (u[i][k] - P.beta[c]);
          gbeta[k] += t;
          D.du[row * D.lddu + c] = s * gv[k] - t;
        }
      }
      // condition gradient: the frames sharing one condition row (f / up) lie in this wave (FW is a
      // multiple of up and groups start at multiples of FW): sum them here, then add to the slot
      // this wave owns exclusively.
      const bool group_end = ((f + 1) % P.up == 0) || (f == F - 1) || (i == FW - 1);
      if (D.g_cproj && group_end) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (k < nk && c < C && has_cp)
            D.g_cproj[((long long)b * P.Fc + fc) * P.ldcp + c] += gcp[k];
          gcp[k] = 0.f;
        }
      }
    }
  }
  if (BWD) {
    if (D.partials) {
      // block partials (no atomics): [ (b*nxb + xb) ][ beta(C) | te(C) | log_scale(1) ]
      __shared__ float red[4][2 * 64 * CPL + 1];
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (k < nk && c < C) { red[wave][c] = gbeta[k]; red[wave][C + c] = gte[k]; }
      }
      if (lane == 0) red[wave][2 * C] = glam;
      __syncthreads();
      float* prow = D.partials + ((long long)b * gridDim.x + blockIdx.x) * (2 * C + 1);
      for (int i = threadIdx.x; i < 2 * C + 1; i += 256)
        prow[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    } else {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (k < nk && c < C) {
          if (D.g_beta) atomicAdd(D.g_beta + c, gbeta[k]);
          if (D.g_te) atomicAdd(D.g_te + (long long)b * P.ldte + c, gte[k]);
        }
      }
      if (D.g_log_scale && lane == 0) atomicAdd(D.g_log_scale, glam);
    }
  }
}

// second stage of the parameter-gradient reduction of dwnorm_kernel<true>
__global__ __launch_bounds__(256) void dwnorm_reduce_kernel(const float* partials, int B, int nxb,
                                                            int C, float* g_beta, float* g_te,
                                                            long long ldte, float* g_log_scale) {
  // time-embedding gradient only: per batch item, over its nxb block partials (short loop);
  // g_beta / g_log_scale are column sums over ALL rows and go through f2g_colsum.
  const int W = 2 * C + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * C) {
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int r = 0; r < nxb; ++r) s += partials[((long long)b * nxb + r) * W + C + c];
    if (g_te) g_te[(long long)b * ldte + c] += s;
  }
}

// dx / dw / db / dgamma of the depthwise conv: thread = channel, register sliding window over a
// strip of TFB frames (halo 6 -> 1.09x reads), partial parameter gradients reduced by atomics.
constexpr int TFB = 32;

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
  if (P.partials) {
    // row (b*nstrips + strip) = [ g_w (C*K, checkpoint layout) | g_b (C) | g_gamma (C) ]: no
    // atomics here; the rows are summed by three f2g_colsum launches
    float* prow = P.partials + ((long long)b * gridDim.y + blockIdx.y) * (long long)(K + 2) * C;
    for (int j = 0; j < K; ++j) prow[(long long)c * K + j] = gw[j + koff];
    prow[(long long)K * C + c] = gb;
    prow[(long long)(K + 1) * C + c] = gg;
    return;
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
  // frames per wave: a multiple of the condition upsampling factor (4 | up)
  const bool four = BWD && f.cproj && f.up == 4;
  const int FW = four ? 4 : 2;
  const int groups = (f.F + FW - 1) / FW;
  const int nxb = (groups + 3) / 4;
  dim3 grid(nxb, f.B);
  if (four) hipLaunchKernelGGL((dwnorm_kernel<BWD, 4>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((dwnorm_kernel<BWD, 2>), grid, dim3(256), 0, st, d);
  int rc = f2g_check_launch();
  if (rc || !BWD || !d.partials) return rc;
  const int W = 2 * f.C + 1, prow = f.B * nxb;
  if (d.g_beta && (rc = f2g_colsum(d.g_beta, d.partials, W, nullptr, 0, prow, f.C, st))) return rc;
  if (d.g_log_scale &&
      (rc = f2g_colsum(d.g_log_scale, d.partials + 2 * f.C, W, nullptr, 0, prow, 1, st)))
    return rc;
  if (d.g_te) {
    const int n = f.B * f.C;
    hipLaunchKernelGGL(dwnorm_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d.partials,
                       f.B, nxb, f.C, d.g_beta, d.g_te, (long long)f.ldte, d.g_log_scale);
    rc = f2g_check_launch();
  }
  return rc;
}

}  // namespace

extern "C" int64_t f2g_dwnorm_bwd_workspace(int32_t B, int32_t F, int32_t C, int32_t up) {
  const int FW = up == 4 ? 4 : 2;
  const int groups = (F + FW - 1) / FW;
  return (int64_t)B * ((groups + 3) / 4) * (2 * C + 1);
}

extern "C" int64_t f2g_dwconv_bwd_workspace(int32_t B, int32_t F, int32_t C, int32_t K) {
  return (int64_t)B * ((F + TFB - 1) / TFB) * (K + 2) * C;
}

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
  int rc = f2g_check_launch();
  if (rc || !d->partials) return rc;
  const int W = (d->K + 2) * d->C, prow = (int)(grid.y * grid.z), CK = d->C * d->K;
  if (d->g_w && (rc = f2g_colsum(d->g_w, d->partials, W, nullptr, 0, prow, CK, stream))) return rc;
  if (d->g_b && (rc = f2g_colsum(d->g_b, d->partials + CK, W, nullptr, 0, prow, d->C, stream)))
    return rc;
  if (d->g_gamma && d->gres &&
      (rc = f2g_colsum(d->g_gamma, d->partials + CK + d->C, W, nullptr, 0, prow, d->C, stream)))
    return rc;
  return rc;
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
