// Many small re-layout launches as ONE (round 6).  A train step changes every weight of the sub-model it
// stepped, so the weight images the kernels read -- window-major / transposed copies (f2g_permute4), zero-padded
// and stacked copies (f2g_fill + f2g_copy3), three-piece bf16 images (f2g_split_bf16x3) -- are rebuilt once per
// sub-step: ~830 launches of 3-6 us in a mel_24k_base GAN step, each a grid of a few blocks.  f2g_multi runs a
// TABLE of such operations in one grid: every entry owns a run of consecutive blocks (sized like its own
// launch would have been), a block finds its entry by scanning the table's block counts.  The entries of
// one call must not depend on each other (the host batches by dependency level, flow2gan_amd/ops.py).
// Element arithmetic is exactly that of the single-operation kernels (same rounding of the bf16 pieces).
#include "common.h"

namespace {

__device__ __forceinline__ void run_fill(const f2g_multi_entry& e, long long b, long long nb) {
  float* x = reinterpret_cast<float*>(e.out);
  const long long n = ((long long)e.n[1] << 32) | (unsigned)e.n[0];
  const float v = __int_as_float((int)e.s[0]);
  for (long long i = b * 256 + threadIdx.x; i < n; i += nb * 256) x[i] = v;
}

__device__ __forceinline__ void run_permute4(const f2g_multi_entry& e, long long b, long long nb) {
  float* out = reinterpret_cast<float*>(e.out);
  const float* in = reinterpret_cast<const float*>(e.in);
  const int n1 = e.n[1], n2 = e.n[2], n3 = e.n[3];
  const long long total = (long long)e.n[0] * n1 * n2 * n3;
  for (long long i = b * 256 + threadIdx.x; i < total; i += nb * 256) {
    long long r = i;
    const int i3 = (int)(r % n3); r /= n3;
    const int i2 = (int)(r % n2); r /= n2;
    const int i1 = (int)(r % n1); r /= n1;
    out[i] = in[r * e.s[0] + i1 * e.s[1] + i2 * e.s[2] + i3 * e.s[3]];
  }
}

__device__ __forceinline__ void run_copy3(const f2g_multi_entry& e, long long b, long long nb) {
  float* out = reinterpret_cast<float*>(e.out);
  const float* in = reinterpret_cast<const float*>(e.in);
  const int n1 = e.n[1], n2 = e.n[2], acc = e.n[3];
  const long long total = (long long)e.n[0] * n1 * n2;
  for (long long i = b * 256 + threadIdx.x; i < total; i += nb * 256) {
    const int c = (int)(i % n2);
    const long long q = i / n2;
    const int r = (int)(q % n1);
    const long long bb = q / n1;
    const float v = in[bb * e.s[2] + r * e.s[3] + c];
    float* o = out + bb * e.s[0] + r * e.s[1] + c;
    *o = acc ? *o + v : v;
  }
}

// f2g_split_bf16x3: dst [row][K / 32][piece][32] bf16, src (rows, K) fp32 with row stride ld
__device__ __forceinline__ void run_split3(const f2g_multi_entry& e, long long b, long long nb) {
  __bf16* dst = reinterpret_cast<__bf16*>(e.out);
  const float* src = reinterpret_cast<const float*>(e.in);
  const int K = e.n[1];
  const long long rows = e.n[0], ld = e.s[0], total = rows * (K / 4);
  for (long long i = b * 256 + threadIdx.x; i < total; i += nb * 256) {
    const long long r = i / (K / 4);
    const int k4 = (int)(i - r * (K / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + k4);
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const __bf16 a = (__bf16)x[q];
      const float r1 = x[q] - (float)a;
      const __bf16 c = (__bf16)r1;
      const __bf16 d = (__bf16)(r1 - (float)c);
      p[0][q] = __builtin_bit_cast(unsigned short, a);
      p[1][q] = __builtin_bit_cast(unsigned short, c);
      p[2][q] = __builtin_bit_cast(unsigned short, d);
    }
    __bf16* o = dst + (r * (K / 32) + k4 / 32) * 96 + (k4 & 31);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<uint2*>(o + 32 * q) =
          make_uint2(p[q][0] | ((unsigned)p[q][1] << 16), p[q][2] | ((unsigned)p[q][3] << 16));
  }
}

// F2G_MULTI_SPLIT3G: the same pieces in MFMA fragment order (f2g_operand.split = 4): [row / 32][K / 32][piece]
// [k step][k half][row % 32][8 bf16].  One thread = 8 consecutive k of one row (two 16-byte loads; lanes l and
// l + 32 of a wave take the two halves of a row's 64 bytes) -> one 16-byte unit per piece; consecutive lanes =
// consecutive rows of a 32-row group = consecutive units of the image: 512 contiguous bytes per piece and wave
// half.  (With run_split3's thread map -- 8-byte stores 512 bytes apart, a quarter of every 64-byte sector -- the
// images' launches were 30 % slower than the row-major ones and cost gemm_x6g_kernel its gain in the laned step.)
__device__ __forceinline__ void run_split3g(const f2g_multi_entry& e, long long b, long long nb) {
  unsigned char* dst = reinterpret_cast<unsigned char*>(e.out);
  const float* src = reinterpret_cast<const float*>(e.in);
  const int K = e.n[1], K8 = K / 8, T = K / 32;
  const long long rows = e.n[0], ld = e.s[0], total = rows * K8;
  for (long long i = b * 256 + threadIdx.x; i < total; i += nb * 256) {
    const int li = (int)(i & 31);
    const long long q = i >> 5;
    const int c8 = (int)(q % K8);
    const long long j = q / K8;
    const float* p = src + (j * 32 + li) * ld + c8 * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
    unsigned a[3], bq[3], c[3], dq[3];
    f2g_split3_pair(v0.x, v0.y, a[0], a[1], a[2]);
    f2g_split3_pair(v0.z, v0.w, bq[0], bq[1], bq[2]);
    f2g_split3_pair(v1.x, v1.y, c[0], c[1], c[2]);
    f2g_split3_pair(v1.z, v1.w, dq[0], dq[1], dq[2]);
    // unit (((j T + slab) 3 + piece) 4 + k-chunk of the slab) 32 + row % 32
    unsigned char* o = dst + ((((j * T + (c8 >> 2)) * 12 + (c8 & 3)) * 32) + li) * 16;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      *reinterpret_cast<uint4*>(o + pc * (4 * 32 * 16)) = make_uint4(a[pc], bq[pc], c[pc], dq[pc]);
  }
}

__global__ __launch_bounds__(256) void multi_kernel(const f2g_multi_desc d) {
  int i = 0;
  long long first = 0;
  while (i + 1 < d.n && (long long)blockIdx.x >= first + d.e[i].blocks) first += d.e[i++].blocks;
  const f2g_multi_entry& e = d.e[i];
  const long long b = (long long)blockIdx.x - first, nb = e.blocks;
  switch (e.kind) {
    case F2G_MULTI_FILL: run_fill(e, b, nb); break;
    case F2G_MULTI_PERMUTE4: run_permute4(e, b, nb); break;
    case F2G_MULTI_COPY3: run_copy3(e, b, nb); break;
    case F2G_MULTI_SPLIT3: run_split3(e, b, nb); break;
    default: run_split3g(e, b, nb); break;
  }
}

}  // namespace

extern "C" int f2g_multi(const f2g_multi_desc* d, f2g_stream_t stream) {
  if (!d || d->n < 0 || d->n > F2G_MULTI_MAX) return F2G_EINVAL;
  long long grid = 0;
  for (int i = 0; i < d->n; ++i) {
    const f2g_multi_entry& e = d->e[i];
    if (!e.out || (e.kind != F2G_MULTI_FILL && !e.in) || e.blocks < 1 || e.kind < 0 || e.kind > F2G_MULTI_SPLIT3G)
      return F2G_EINVAL;
    if (e.kind == F2G_MULTI_SPLIT3G && (e.n[0] & 31)) return F2G_EINVAL;
    if ((e.kind == F2G_MULTI_SPLIT3 || e.kind == F2G_MULTI_SPLIT3G) &&
        (e.n[1] < 32 || (e.n[1] % 32) || e.s[0] < e.n[1] || (e.s[0] & 3) || (((uintptr_t)e.in) & 15) ||
         (((uintptr_t)e.out) & 15)))
      return F2G_EINVAL;
    grid += e.blocks;
  }
  if (grid == 0) return F2G_OK;
  if (grid >= (1ll << 31)) return F2G_EINVAL;
  hipLaunchKernelGGL(multi_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *d);
  return f2g_check_launch();
}
