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

// ---- vectorised variant (C % 4 == 0, 16-byte aligned rows): the one the model shapes use --------
//
// Same ownership as dwnorm_kernel (one wave = FW frames, block = 4 waves), but
//   * a lane owns 4 CONSECUTIVE channels per 256-channel chunk (dwordx4 loads / stores: a wave
//     moves 1 KB per instruction, 4x fewer memory instructions),
//   * ALL (FW+6) x C/256 input rows, the gradient rows, the condition rows and the time-embedding
//     row are requested before anything is consumed (24-40 KB in flight per wave -- the kernel is
//     a latency / memory-level-parallelism problem, not an ALU one),
//   * the depthwise taps are transposed once per block into LDS ([tap][C], conflict-free float4
//     reads) while those loads fly, instead of 7 strided gathers per lane and chunk.
// Row indices are clamped and the loads unconditional; out-of-range rows are zeroed at use.
__device__ __forceinline__ void ld4(float (&d)[4], const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}
__device__ __forceinline__ void st4(float* p, const float (&s)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(s[0], s[1], s[2], s[3]);
}

template <bool BWD, int FW, int NCH>
__global__ __launch_bounds__(256, (BWD || (FW == 4 && NCH == 3)) ? 2 : 3) void dwnorm4_kernel(const f2g_dwnorm_bwd_desc D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const f2g_dwnorm_fwd_desc& P = D.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = P.F, C = P.C, K = P.K;
  float* wl = sm;           // [7][C] taps, zero padded to 7
  float* bl = sm + 7 * C;   // depthwise bias
  float* tl = sm + 8 * C;   // BiasNorm bias
  const int groups = (F + FW - 1) / FW;
  const int b = blockIdx.y;
  const int grp = blockIdx.x * 4 + wave;
  const bool live = grp < groups;
  const int f0 = (live ? grp : 0) * FW;
  const int len_b = P.lens ? P.lens[b] : F;
  const int lim = len_b < F ? len_b : F;          // rows >= lim read as zero
  const long long rb = (long long)b * F;
  const float escale = expf(P.log_scale[0]);
  const float invC = 1.f / (float)C;

  int c4[NCH];
  bool cok[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = 256 * k + 4 * lane;
    cok[k] = c < C;
    c4[k] = cok[k] ? c : 0;
  }
  // 1. every global read of this wave, back to back
  float xr[NCH][FW + 6][4];
#pragma unroll
  for (int r = 0; r < FW + 6; ++r) {
    int f = f0 - HALO + r;
    f = f < 0 ? 0 : (f > F - 1 ? F - 1 : f);
    const float* xrow = P.x + (rb + f) * P.ldx;
#pragma unroll
    for (int k = 0; k < NCH; ++k) ld4(xr[k][r], xrow + c4[k]);
  }
  float gz[BWD ? FW : 1][NCH][4];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      int f = f0 + i;
      f = f > F - 1 ? F - 1 : f;
      const float* grow = D.gz + (rb + f) * D.ldgz;
#pragma unroll
      for (int k = 0; k < NCH; ++k) ld4(gz[i][k], grow + c4[k]);
    }
  }
  const bool need_cp = P.cproj && (!BWD || D.g_te);
  float cp[FW][NCH][4];
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    int fc = P.cproj ? (f0 + i) / P.up : 0;
    const bool has = need_cp && fc < P.Fc && f0 + i < F;
    fc = has ? fc : 0;
    const float* cprow = need_cp ? P.cproj + ((long long)b * P.Fc + fc) * P.ldcp : P.x;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      ld4(cp[i][k], cprow + c4[k]);
      if (!has) { cp[i][k][0] = cp[i][k][1] = cp[i][k][2] = cp[i][k][3] = 0.f; }
    }
  }
  float te1[NCH][4];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    ld4(te1[k], (P.te ? P.te + (long long)b * P.ldte : P.x) + c4[k]);
  }
  // 2. taps -> LDS, transposed, while the rows are on their way
  if (K == 7) {
    // the model's case: the (C, 1, 7) block is read as 7C/4 float4s, all of a thread's loads in
    // flight together (one memory round trip instead of one per loop trip), divisions by the
    // constant 7 (the runtime-K loop below costs ~35 VALU instructions per element)
    constexpr int NQ = (7 * NCH * 256 / 4 + 255) / 256;
    const int nq = 7 * C / 4;       // C % 4 == 0
    float4 q[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int i4 = threadIdx.x + 256 * n;
      q[n] = reinterpret_cast<const float4*>(P.w_dw)[i4 < nq ? i4 : 0];
    }
    float bq[NCH], tq[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int ic = i < C ? i : 0;
      bq[k] = P.b_dw ? P.b_dw[ic] : 0.f;
      tq[k] = P.beta[ic];
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int i4 = threadIdx.x + 256 * n;
      if (i4 < nq) {
        const float v[4] = {q[n].x, q[n].y, q[n].z, q[n].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned i = 4u * i4 + e, c = i / 7u, j = i - 7u * c;
          wl[j * C + c] = v[e];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < C) { bl[i] = bq[k]; tl[i] = tq[k]; }
    }
  } else {
    for (int i = threadIdx.x; i < 7 * C; i += 256) wl[i] = 0.f;
    const int koff = (7 - K) / 2;
    __syncthreads();
    for (int i = threadIdx.x; i < C * K; i += 256) {
      const int c = i / K, j = i - c * K;
      wl[(j + koff) * C + c] = P.w_dw[i];
    }
    for (int i = threadIdx.x; i < C; i += 256) {
      bl[i] = P.b_dw ? P.b_dw[i] : 0.f;
      tl[i] = P.beta[i];
    }
  }
  __syncthreads();

  // 3. depthwise conv + sum of squares
  float u[FW][NCH][4];
  float ssq[FW];
#pragma unroll
  for (int i = 0; i < FW; ++i) ssq[i] = 0.f;
#pragma unroll
  for (int r = 0; r < FW + 6; ++r) {
    const int f = f0 - HALO + r;
    if (f < 0 || f >= lim) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) xr[k][r][0] = xr[k][r][1] = xr[k][r][2] = xr[k][r][3] = 0.f;
    }
  }
  float bt[NCH][4];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    float b4[4];
    ld4(b4, bl + c4[k]);
    ld4(bt[k], tl + c4[k]);
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) u[i][k][e] = b4[e];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      float w4[4];
      ld4(w4, wl + j * C + c4[k]);
#pragma unroll
      for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) u[i][k][e] += w4[e] * xr[k][i + j][e];
    }
    if (cok[k]) {
#pragma unroll
      for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = u[i][k][e] - bt[k][e];
          ssq[i] += dlt * dlt;
        }
    }
    if (P.te) {
#pragma unroll
      for (int e = 0; e < 4; ++e) te1[k][e] += 1.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) te1[k][e] = 1.f;
    }
  }

  float gcp[NCH][4], gbeta[NCH][4], gte[NCH][4];
  float glam = 0.f;
  if (BWD) {
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { gcp[k][e] = 0.f; gbeta[k][e] = 0.f; gte[k][e] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    const int f = f0 + i;
    if (f >= F || !live) break;
    const long long row = rb + f;
    const float r = wave_sum(ssq[i]) * invC;
    const float s = escale / sqrtf(r);
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (cok[k]) {
          float z4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) z4[e] = (u[i][k][e] * s + cp[i][k][e]) * te1[k][e];
          if (P.z_format == 0) {
            st4(P.z + row * P.ldz + c4[k], z4);
          } else {   // the GEMM operand as it is consumed: split-bf16 image (1) or bf16 tensor (2)
            unsigned short hb[4], lb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const __bf16 hh = (__bf16)z4[e];
              const __bf16 ll = (__bf16)(z4[e] - (float)hh);
              hb[e] = __builtin_bit_cast(unsigned short, hh);
              lb[e] = __builtin_bit_cast(unsigned short, ll);
            }
            const unsigned h01 = hb[0] | ((unsigned)hb[1] << 16), h23 = hb[2] | ((unsigned)hb[3] << 16);
            if (P.z_format == 1)
              *reinterpret_cast<uint4*>(P.z + row * P.ldz + c4[k]) =
                  make_uint4(h01, h23, lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
            else
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(P.z) + row * P.ldz + c4[k]) =
                  make_uint2(h01, h23);
          }
        }
      }
      if (P.rstd && lane == 0) P.rstd[row] = s;
    } else {
      float dsum = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (cok[k]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = gz[i][k][e];
            const float gv = g * te1[k][e];
            gz[i][k][e] = gv;
            dsum += gv * u[i][k][e];
            gte[k][e] += g * (u[i][k][e] * s + cp[i][k][e]);
            gcp[k][e] += gv;
          }
        }
      }
      dsum = wave_sum(dsum);
      const float coef = s * dsum / ((float)C * r);
      glam += s * dsum;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (cok[k]) {
          float d4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = coef * (u[i][k][e] - bt[k][e]);
            gbeta[k][e] += t;
            d4[e] = s * gz[i][k][e] - t;
          }
          st4(D.du + row * D.lddu + c4[k], d4);
        }
      }
      // condition gradient: the frames sharing one condition row lie inside this wave (up | FW)
      const bool group_end = ((f + 1) % P.up == 0) || (f == F - 1) || (i == FW - 1);
      if (D.g_cproj && group_end) {
        const int fc = f / P.up;
        const bool has_cp = fc < P.Fc;
        float* grow = D.g_cproj + ((long long)b * P.Fc + (has_cp ? fc : 0)) * P.ldcp;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          if (cok[k] && has_cp) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            if (!D.g_cproj_store) ld4(o, grow + c4[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += gcp[k][e];
            st4(grow + c4[k], o);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) gcp[k][e] = 0.f;
        }
      }
    }
  }
  if (BWD) {
    if (D.partials) {
      // block partials: [ (b*nxb + xb) ][ beta(C) | te(C) | log_scale(1) ]; the tap area is dead
      __syncthreads();
      float* red = sm;  // [4][2C+1] <= 9C floats
      const int W = 2 * C + 1;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (cok[k]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            red[wave * W + c4[k] + e] = gbeta[k][e];
            red[wave * W + C + c4[k] + e] = gte[k][e];
          }
        }
      }
      if (lane == 0) red[wave * W + 2 * C] = glam;
      __syncthreads();
      float* prow = D.partials + ((long long)b * gridDim.x + blockIdx.x) * W;
      for (int i = threadIdx.x; i < W; i += 256)
        prow[i] = red[i] + red[W + i] + red[2 * W + i] + red[3 * W + i];
    } else {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (cok[k]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (D.g_beta) atomicAdd(D.g_beta + c4[k] + e, gbeta[k][e]);
            if (D.g_te) atomicAdd(D.g_te + (long long)b * P.ldte + c4[k] + e, gte[k][e]);
          }
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
  // grid (column blocks of [beta | te | log_scale], batch item): sum the item's nxb block
  // partials; the time-embedding gradient is per item (exclusive owner, plain +=), beta and
  // log_scale are sums over all items (B-way atomics on C + 1 addresses).
  const int W = 2 * C + 1;
  const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (q >= W) return;
  const float* p = partials + (long long)b * nxb * W + q;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < nxb; r += 4) {
    s0 += p[(long long)r * W];
    s1 += p[(long long)(r + 1) * W];
    s2 += p[(long long)(r + 2) * W];
    s3 += p[(long long)(r + 3) * W];
  }
  for (; r < nxb; ++r) s0 += p[(long long)r * W];
  const float s = (s0 + s1) + (s2 + s3);
  if (q < C) {
    if (g_beta) atomicAdd(g_beta + q, s);
  } else if (q < 2 * C) {
    if (g_te) g_te[(long long)b * ldte + (q - C)] += s;
  } else if (g_log_scale) {
    atomicAdd(g_log_scale, s);
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


// Vectorised variant (K == 7, C % 4 == 0, aligned rows).  Work unit of a WAVE: 256 channels (4 per
// lane) x a strip of DWS_ROUNDS x DWS_FR frames; per round all 14 + 14 + 8 rows (du and x with the
// +-3 halo, the residual gradient) are requested back to back as dwordx4 loads, the halo re-reads
// are cache hits.  Tap / bias / scale gradients stay in registers over the strip, the 4 waves of
// a block (4 consecutive strips of the same channels) are combined through LDS, and one partial
// row per block goes to the workspace (reduced by f2g_colsum: no atomics on the hot rows).
constexpr int DWS_FR = 8;        // frames per round: 8 + 6 rows of du and x, 8 of the residual gradient in flight
constexpr int DWS_ROUNDS = 2;    // (two memory round trips per wave; four rounds of 4 frames measured 23-48 us)
constexpr int DWS_STRIP = DWS_FR * DWS_ROUNDS;

__global__ __launch_bounds__(256) void dwconv4_bwd_kernel(const f2g_dwconv_bwd_desc P) {
  __shared__ float red[4][9][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = P.F, C = P.C;
  const int b = blockIdx.z;
  const int cbase = blockIdx.x * 256 + 4 * lane;
  const bool cok = cbase < C;
  const int c4 = cok ? cbase : 0;
  const int strips = (F + DWS_STRIP - 1) / DWS_STRIP;
  const int strip = blockIdx.y * 4 + wave;
  const bool live = strip < strips;
  const int fs0 = (live ? strip : 0) * DWS_STRIP;
  const int len_b = P.lens ? P.lens[b] : F;
  const int lim = len_b < F ? len_b : F;
  const long long rb = (long long)b * F;

  float w[4][7];  // [channel][tap]: 28 consecutive floats of the checkpoint layout
  {
    float t[28];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(P.w_dw + (long long)c4 * 7 + 4 * q);
      t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 7; ++j) w[e][j] = t[e * 7 + j];
  }
  float gam[4] = {1.f, 1.f, 1.f, 1.f};
  if (P.gamma) ld4(gam, P.gamma + c4);
  float gw[4][7], gb[4], gg[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gb[e] = 0.f; gg[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) gw[e][j] = 0.f;
  }

  for (int rd = 0; rd < DWS_ROUNDS; ++rd) {
    const int fs = fs0 + DWS_FR * rd;
    if (fs >= F || !live) break;
    float du[DWS_FR + 6][4], xr[DWS_FR + 6][4], gr[DWS_FR][4];
#pragma unroll
    for (int r = 0; r < DWS_FR + 6; ++r) {
      int f = fs - 3 + r;
      f = f < 0 ? 0 : (f > F - 1 ? F - 1 : f);
      ld4(du[r], P.du + (rb + f) * P.lddu + c4);
      ld4(xr[r], P.x + (rb + f) * P.ldx + c4);
    }
#pragma unroll
    for (int i = 0; i < DWS_FR; ++i) {
      int f = fs + i;
      f = f > F - 1 ? F - 1 : f;
      ld4(gr[i], (P.gres ? P.gres + (rb + f) * P.ldgres : P.x + (rb + f) * P.ldx) + c4);
    }
    // residual-scale gradient uses the UNMASKED input row
#pragma unroll
    for (int i = 0; i < DWS_FR; ++i) {
      const bool in = fs + i < F && P.gres;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gr[i][e] = in ? gr[i][e] : 0.f;
        gg[e] += gr[i][e] * xr[i + 3][e];
      }
    }
#pragma unroll
    for (int r = 0; r < DWS_FR + 6; ++r) {
      const int f = fs - 3 + r;
      if (f < 0 || f >= F) { du[r][0] = du[r][1] = du[r][2] = du[r][3] = 0.f; }
      if (f < 0 || f >= lim) { xr[r][0] = xr[r][1] = xr[r][2] = xr[r][3] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < DWS_FR; ++i) {
      const int f = fs + i;
      float dx[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) a += w[e][j] * du[i + 6 - j][e];
        a = f >= len_b ? 0.f : a;
        dx[e] = a + gam[e] * gr[i][e];
        const float duc = du[i + 3][e];
        gb[e] += duc;
#pragma unroll
        for (int j = 0; j < 7; ++j) gw[e][j] += duc * xr[i + j][e];
      }
      if (f < F && cok) st4(P.gx + (rb + f) * P.ldgx + c4, dx);
    }
  }
  // combine the block's 4 strips, then one partial row (or atomics without a workspace)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int j = 0; j < 7; ++j) red[wave][j][4 * lane + e] = gw[e][j];
    red[wave][7][4 * lane + e] = gb[e];
    red[wave][8][4 * lane + e] = gg[e];
  }
  __syncthreads();
  const int cl = threadIdx.x, c = blockIdx.x * 256 + cl;
  if (c >= C) return;
  float s[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) s[q] = red[0][q][cl] + red[1][q][cl] + red[2][q][cl] + red[3][q][cl];
  if (P.partials) {
    float* prow = P.partials + ((long long)b * gridDim.y + blockIdx.y) * (long long)9 * C;
#pragma unroll
    for (int j = 0; j < 7; ++j) prow[(long long)c * 7 + j] = s[j];
    prow[(long long)7 * C + c] = s[7];
    prow[(long long)8 * C + c] = s[8];
    return;
  }
  if (P.g_w)
    for (int j = 0; j < 7; ++j) atomicAdd(P.g_w + c * 7 + j, s[j]);
  if (P.g_b) atomicAdd(P.g_b + c, s[7]);
  if (P.g_gamma && P.gres) atomicAdd(P.g_gamma + c, s[8]);
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

// dwnorm4_kernel needs dwordx4-addressable rows everywhere
bool vec4_ok(const f2g_dwnorm_bwd_desc& d, bool bwd) {
  const f2g_dwnorm_fwd_desc& f = d.f;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = (f.C % 4) == 0 && (f.ldx % 4) == 0 && al(f.x) && (f.K != 7 || al(f.w_dw));
  if (f.cproj) ok = ok && (f.ldcp % 4) == 0 && al(f.cproj);
  if (f.te) ok = ok && (f.ldte % 4) == 0 && al(f.te);
  if (!bwd) return ok && (f.ldz % 4) == 0 && al(f.z);
  ok = ok && (d.ldgz % 4) == 0 && al(d.gz) && (d.lddu % 4) == 0 && al(d.du);
  if (d.g_cproj) ok = ok && al(d.g_cproj);
  return ok;
}

template <bool BWD>
int launch_dwnorm(const f2g_dwnorm_bwd_desc& d, hipStream_t st) {
  const f2g_dwnorm_fwd_desc& f = d.f;
  if (f.B <= 0 || f.F <= 0) return F2G_OK;
  if (!BWD && f.z_format != 0) {   // operand-format outputs: vector kernel only
    const bool okf = (f.z_format == 1 || f.z_format == 2) && (f.C % 4) == 0 && (f.ldx % 4) == 0 &&
                     (((uintptr_t)f.x) & 15) == 0 && (((uintptr_t)f.z) & 15) == 0 &&
                     (f.K != 7 || (((uintptr_t)f.w_dw) & 15) == 0) &&
                     (f.ldz % (f.z_format == 1 ? 4 : 8)) == 0 &&
                     (!f.cproj || ((f.ldcp % 4) == 0 && (((uintptr_t)f.cproj) & 15) == 0)) &&
                     (!f.te || ((f.ldte % 4) == 0 && (((uintptr_t)f.te) & 15) == 0));
    if (!okf) return F2G_EINVAL;
  }
  // frames per wave: a multiple of the condition upsampling factor (4 | up)
  // backward: 4 frames per wave wherever the registers allow it (<= 512 channels: 2 chunks per
  // lane); 768 channels would spill, so they keep 2 (the workspace query sizes for 2, the larger)
  const bool four = (BWD && ((f.cproj && f.up == 4) || (f.C <= 512 && (f.C % 4) == 0))) ||
                    (!BWD && (f.C % 4) == 0);
  const int FW = four ? 4 : 2;
  const int groups = (f.F + FW - 1) / FW;
  const int nxb = (groups + 3) / 4;
  dim3 grid(nxb, f.B);
  if (vec4_ok(d, BWD)) {
    const size_t smem = (size_t)9 * f.C * sizeof(float);
    const int nch = (f.C + 255) / 256;
#define F2G_DW4(FWV, NCHV) \
  hipLaunchKernelGGL((dwnorm4_kernel<BWD, FWV, NCHV>), grid, dim3(256), smem, st, d)
    if (four) {
      if (nch == 1) F2G_DW4(4, 1); else if (nch == 2) F2G_DW4(4, 2); else F2G_DW4(4, 3);
    } else {
      if (nch == 1) F2G_DW4(2, 1); else if (nch == 2) F2G_DW4(2, 2); else F2G_DW4(2, 3);
    }
#undef F2G_DW4
  } else if (four) {
    hipLaunchKernelGGL((dwnorm_kernel<BWD, 4>), grid, dim3(256), 0, st, d);
  } else {
    hipLaunchKernelGGL((dwnorm_kernel<BWD, 2>), grid, dim3(256), 0, st, d);
  }
  int rc = f2g_check_launch();
  if (rc || !BWD || !d.partials) return rc;
  const int W = 2 * f.C + 1;
  hipLaunchKernelGGL(dwnorm_reduce_kernel, dim3((W + 255) / 256, f.B), dim3(256), 0, st,
                     d.partials, f.B, nxb, f.C, d.g_beta, d.g_te, (long long)f.ldte,
                     d.g_log_scale);
  return f2g_check_launch();
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
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  const bool vec4 = d->K == 7 && (d->C % 4) == 0 && (d->lddu % 4) == 0 && (d->ldx % 4) == 0 &&
                    (d->ldgx % 4) == 0 && al(d->du) && al(d->x) && al(d->gx) && al(d->w_dw) &&
                    (!d->gres || ((d->ldgres % 4) == 0 && al(d->gres))) &&
                    (!d->gamma || al(d->gamma));
  dim3 grid((d->C + 255) / 256, (d->F + TFB - 1) / TFB, d->B);
  if (vec4) {
    const int strips = (d->F + DWS_STRIP - 1) / DWS_STRIP;
    grid.y = (strips + 3) / 4;
    hipLaunchKernelGGL(dwconv4_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
  } else {
    hipLaunchKernelGGL(dwconv_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
  }
  int rc = f2g_check_launch();
  if (rc || !d->partials) return rc;
  const int W = (d->K + 2) * d->C, prow = (int)(grid.y * grid.z), CK = d->C * d->K;
  f2g_colsegs segs = {{d->g_w, d->g_b, d->gres ? d->g_gamma : nullptr},
                      {0, CK, CK + d->C},
                      {CK, d->C, d->C}};
  return f2g_colsum_segments(d->partials, W, prow, W, segs, (hipStream_t)stream);
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
