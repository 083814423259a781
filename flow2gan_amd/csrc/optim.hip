// ScaledAdam (reference flow2gan/optim.py:125-255, 451-619) as three multi-tensor launches per step
// over ALL parameter tensors of an optimizer, instead of the reference's stack-by-shape copies
// (optim.py:104-122: ~3 full-model copies per step) and ~15 small ATen kernels per shape batch:
//
//   f2g_sadam_stats    per tensor: sum g^2, sum p*g, sum p^2          (one pass over p and g)
//   f2g_sadam_prepare  per group: gradient-clipping factor from the running median of the model
//                      norm (optim.py:509-619, on the device: no .item() sync per step), then per
//                      tensor the learned-scale bookkeeping (optim.py:154-239) -> coefficients
//   f2g_sadam_update   elementwise: clip, second moment, scaled step, scale step, momentum, p += m
//
// HBM-bound: stats reads 8 B/element, update reads 16 and writes 12 B/element.
#include "common.h"

namespace {

constexpr int SADAM_CHUNK = 8192;  // elements per block
constexpr int NCOEF = F2G_SADAM_NCOEF;
constexpr int TS = F2G_SADAM_TSTATE;  // per-tensor state: param_rms, scale_exp_avg_sq, scale_grads[8]

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

__global__ __launch_bounds__(256) void sadam_stats_kernel(const f2g_sadam_tensor* tensors,
                                                          const f2g_sadam_chunk* chunks,
                                                          float* stats) {
  __shared__ float sh[4];
  const f2g_sadam_chunk c = chunks[blockIdx.x];
  const f2g_sadam_tensor t = tensors[c.tensor];
  const float* p = t.p + c.offset;
  const float* g = t.g ? t.g + c.offset : nullptr;
  float gg = 0.f, pg = 0.f, pp = 0.f;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g)) & 15) == 0;
  if (vec) {
    const int n4 = c.count >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 a = reinterpret_cast<const float4*>(p)[i];
      const float4 b = g ? reinterpret_cast<const float4*>(g)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      gg += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
      pg += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      pp += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < c.count; i += 256) {
      const float a = p[i], b = g ? g[i] : 0.f;
      gg += b * b; pg += a * b; pp += a * a;
    }
  } else {
    for (int i = threadIdx.x; i < c.count; i += 256) {
      const float a = p[i], b = g ? g[i] : 0.f;
      gg += b * b; pg += a * b; pp += a * a;
    }
  }
  gg = block_sum(gg, sh);
  pg = block_sum(pg, sh);
  pp = block_sum(pp, sh);
  if (threadIdx.x == 0) {
    atomicAdd(stats + 3 * c.tensor + 0, gg);
    atomicAdd(stats + 3 * c.tensor + 1, pg);
    atomicAdd(stats + 3 * c.tensor + 2, pp);
  }
}

// one block per parameter group
__global__ __launch_bounds__(256) void sadam_prepare_kernel(const f2g_sadam_tensor* tensors,
                                                            const f2g_sadam_group G,
                                                            const float* stats, float* tstate,
                                                            float* gstate, float* coef) {
  __shared__ float sh[4];
  __shared__ unsigned srt[1024];
  __shared__ float s_clip;
  const int tid = threadIdx.x;
  const int step = G.step;
  const int P = G.size_update_period;
  const int period = G.clipping_update_period;
  float* norms = gstate;                 // [period]
  float* thr = gstate + 1024;            // threshold, has_threshold, bad-median flag
  // ---- gradient clipping (optim.py:509-619)
  float clip = 1.f;
  if (G.clipping_scale > 0.f && step > 0) {
    float part = 0.f;
    for (int i = tid; i < G.count; i += 256) {
      const int t = G.first + i;
      const float w = tensors[t].is_scalar ? G.scalar_lr_scale * G.scalar_lr_scale
                                           : tstate[t * TS] * tstate[t * TS];
      part += w * stats[3 * t];
    }
    const float tot_norm = sqrtf(block_sum(part, sh));
    if (tid == 0) norms[step % period] = tot_norm;
    __syncthreads();
    const bool irregular = (step == 10 || step == 20 || step == 40) && step < period;
    if (step % period == 0 || irregular) {
      // sorted_norms = model_norms.sort(); irregular steps keep the last `step` of them
      int n2 = 1;
      while (n2 < period) n2 <<= 1;
      // norms are >= 0, +inf or NaN: their bit patterns order like torch.sort does (NaN last);
      // the power-of-two padding sorts behind everything
      for (int i = tid; i < n2; i += 256) {
        unsigned key = 0xffffffffu;
        if (i < period) {
          const float v = norms[i];
          key = (v != v) ? 0x7fc00000u : __float_as_uint(v);
        }
        srt[i] = key;
      }
      __syncthreads();
      for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < n2; i += 256) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const unsigned a = srt[i], b = srt[ixj];
              const bool up = (i & k) == 0;
              if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
            }
          }
          __syncthreads();
        }
      }
      if (tid == 0) {
        const int num = irregular ? step : period;
        const int base = period - num;
        int idx = (num / 4) * 2;
        if (idx > num - 1) idx = num - 1;
        const float median = __uint_as_float(srt[base + idx]);
        if (median - median != 0.f) thr[2] = 1.f;  // "Too many grads were not finite"
        thr[0] = G.clipping_scale * median * (irregular ? 2.f : 1.f);
        thr[1] = 1.f;
      }
      __syncthreads();
    }
    if (tid == 0) {
      float c = 1.f;
      if (thr[1] != 0.f) {
        c = fminf(1.f, thr[0] / (tot_norm + 1.0e-20f));
        if (c != c) c = 0.f;
      }
      s_clip = c;
    }
    __syncthreads();
    clip = s_clip;
  }
  // ---- per-tensor coefficients (optim.py:125-239)
  const float bc2 = 1.f - powf(G.beta2, (float)(step + 1));
  const float inv_bc2 = bc2 < 0.99f ? 1.f / bc2 : 1.f;
  for (int i = tid; i < G.count; i += 256) {
    const int t = G.first + i;
    float* c = coef + (size_t)t * NCOEF;
    float lr = G.lr, rmsc = 1.f, ss = 0.f;
    if (tensors[t].is_scalar) {
      lr *= G.scalar_lr_scale;
    } else {
      float* st = tstate + t * TS;
      const float numel = (float)tensors[t].numel;
      const float pg = stats[3 * t + 1], pp = stats[3 * t + 2];
      if (step == 0) {
        st[0] = sqrtf(pp / numel);
        st[1] = 0.f;
        for (int k = 0; k < P; ++k) st[2 + k] = 0.f;
      }
      // clip == 0 means a non-finite total norm: the reference zeroes p.grad (optim.py:615-617),
      // so its scale_grads entry is 0, not 0 * NaN
      st[2 + step % P] = (clip == 0.f) ? 0.f : clip * pg;
      const bool due = step % P == P - 1;
      if (due) st[0] = sqrtf(pp / numel);
      const float rms = st[0];
      rmsc = fmaxf(rms, G.param_min_rms);
      if (due && step > 0) {
        const float b2c = powf(G.beta2, (float)P);
        float sq = 0.f, sm = 0.f;
        for (int k = 0; k < P; ++k) { sq += st[2 + k] * st[2 + k]; sm += st[2 + k]; }
        st[1] = st[1] * b2c + (sq / (float)P) * (1.f - b2c);
        const int size_step = (step + 1) / P;
        const float bc = 1.f - powf(b2c, (float)size_step);
        const float denom = sqrtf(st[1]) + G.eps;
        ss = -(G.lr * G.scalar_lr_scale) * sqrtf(bc) * sm / denom;
        if (rms < G.param_min_rms) ss = 0.f;
        ss = fminf(fmaxf(ss, -0.1f), 0.1f);
        ss = fminf(ss, (G.param_max_rms - rms) / rms);
      }
    }
    c[0] = clip; c[1] = lr; c[2] = rmsc; c[3] = ss;
    c[4] = G.beta1; c[5] = G.beta2; c[6] = G.eps; c[7] = inv_bc2;
    c[8] = tensors[t].is_scalar ? G.scalar_max : 0.f;
  }
}

__device__ __forceinline__ void sadam_elem(float& p, float g, float& v, float& m, const float* c) {
  const float clip = c[0];
  const float gc = clip == 0.f ? 0.f : g * clip;
  v = v * c[5] + (1.f - c[5]) * gc * gc;
  const float denom = sqrtf(v * c[7]) + c[6];
  float delta = (-c[1] * gc) / denom;
  delta *= c[2];
  delta += p * c[3];
  m = m * c[4] + delta * (1.f - c[4]);
  p += m;
  if (c[8] > 0.f) p = fminf(fmaxf(p, -c[8]), c[8]);
}

__global__ __launch_bounds__(256) void sadam_update_kernel(const f2g_sadam_tensor* tensors,
                                                           const f2g_sadam_chunk* chunks,
                                                           const float* coef) {
  const f2g_sadam_chunk ch = chunks[blockIdx.x];
  const f2g_sadam_tensor t = tensors[ch.tensor];
  __shared__ float c[NCOEF];
  if (threadIdx.x < NCOEF) c[threadIdx.x] = coef[(size_t)ch.tensor * NCOEF + threadIdx.x];
  __syncthreads();
  float* p = t.p + ch.offset;
  const float* g = t.g ? t.g + ch.offset : nullptr;
  float* v = t.v + ch.offset;
  float* m = t.m + ch.offset;
  const bool vec =
      ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)v) | ((uintptr_t)m)) & 15) == 0;
  int done = 0;
  if (vec) {
    const int n4 = ch.count >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 a = reinterpret_cast<float4*>(p)[i];
      const float4 b = g ? reinterpret_cast<const float4*>(g)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 vv = reinterpret_cast<float4*>(v)[i];
      float4 mm = reinterpret_cast<float4*>(m)[i];
      sadam_elem(a.x, b.x, vv.x, mm.x, c);
      sadam_elem(a.y, b.y, vv.y, mm.y, c);
      sadam_elem(a.z, b.z, vv.z, mm.z, c);
      sadam_elem(a.w, b.w, vv.w, mm.w, c);
      reinterpret_cast<float4*>(p)[i] = a;
      reinterpret_cast<float4*>(v)[i] = vv;
      reinterpret_cast<float4*>(m)[i] = mm;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < ch.count; i += 256) {
    float a = p[i], vv = v[i], mm = m[i];
    sadam_elem(a, g ? g[i] : 0.f, vv, mm, c);
    p[i] = a; v[i] = vv; m[i] = mm;
  }
}

}  // namespace

extern "C" int32_t f2g_sadam_chunk_elems(void) { return SADAM_CHUNK; }

extern "C" int f2g_sadam_stats(const f2g_sadam_tensor* tensors, const f2g_sadam_chunk* chunks,
                               int32_t nchunks, float* stats, int32_t ntensors,
                               f2g_stream_t stream) {
  if (!tensors || !chunks || !stats || nchunks < 0 || ntensors <= 0) return F2G_EINVAL;
  int rc = f2g_fill(stats, 0.f, (int64_t)3 * ntensors, stream);
  if (rc || nchunks == 0) return rc;
  hipLaunchKernelGGL(sadam_stats_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors,
                     chunks, stats);
  return f2g_check_launch();
}

extern "C" int f2g_sadam_prepare(const f2g_sadam_tensor* tensors, const f2g_sadam_group* group,
                                 const float* stats, float* tstate, float* gstate, float* coef,
                                 f2g_stream_t stream) {
  if (!tensors || !group || !stats || !tstate || !gstate || !coef) return F2G_EINVAL;
  if (group->size_update_period < 1 || group->size_update_period > 8 ||
      group->clipping_update_period < 1 || group->clipping_update_period > 1024)
    return F2G_EINVAL;
  if (group->count <= 0) return F2G_OK;
  hipLaunchKernelGGL(sadam_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tensors,
                     *group, stats, tstate, gstate, coef);
  return f2g_check_launch();
}

extern "C" int f2g_sadam_update(const f2g_sadam_tensor* tensors, const f2g_sadam_chunk* chunks,
                                int32_t nchunks, const float* coef, f2g_stream_t stream) {
  if (!tensors || !chunks || !coef || nchunks < 0) return F2G_EINVAL;
  if (nchunks == 0) return F2G_OK;
  hipLaunchKernelGGL(sadam_update_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors,
                     chunks, coef);
  return f2g_check_launch();
}
