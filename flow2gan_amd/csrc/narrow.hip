// Narrow implicit GEMMs (<= 4 output columns, or <= 4 weight-gradient rows) on the vector ALUs.
//
// conv_post layers (Cout = 1: discriminators.py:76,184) and the data gradient of the first MRD conv
// (Cin = 2) would waste 8-32x of a 32-wide MFMA tile; they are dot products of an im2col row with
// a handful of weight vectors, i.e. bandwidth-bound reads of the (L2-resident, overlapping) window.
//   forms 0/1: one wave per output row, lanes stride K in float4 chunks, wave-shuffle reduction;
//   form 2   : out[m, :] += sum_r A[r, m] * B[r, :], a block owns a run of rows, threads own
//              float4 chunks of the window columns, one atomic per (block, column).
// Same operand descriptor and the same epilogue options as gemm_kernel (bias, row map, leaky
// ReLU, accumulate / atomic), minus the residual / PReLU-derivative paths nobody uses here.
#include "common.h"

namespace {

struct RowCtx {
  long long base;
  int l1b, e0;
};

// magic numbers of an operand's divisors (common.h: fast_div), computed once per thread
struct Mg {
  unsigned seg, p0, p1;
  int seglen;
};
__device__ __forceinline__ Mg magics(const f2g_operand& S) {
  Mg m;
  m.seglen = S.seglen < S.cols ? S.seglen : S.cols;
  m.seg = magic_of(m.seglen);
  m.p0 = magic_of(S.P0);
  m.p1 = magic_of(S.P1);
  return m;
}

__device__ __forceinline__ RowCtx decode_row(const f2g_operand& S, int r, const Mg& mg) {
  RowCtx rc;
  int s, p1, p0;
  if (S.P0 == 1 && S.P1 == 1) {
    s = r; p1 = 0; p0 = 0;
  } else {
    int q = fast_div(r, S.P0, mg.p0);
    p0 = r - q * S.P0;
    s = fast_div(q, S.P1, mg.p1);
    p1 = q - s * S.P1;
  }
  rc.base = (long long)s * S.seq_stride;
  rc.l1b = p1 * S.step1 - S.pad1;
  rc.e0 = (p0 * S.step0 - S.pad0) * S.unit;
  return rc;
}

// element (row, c) of an operand, branch-free clamped load; applies PReLU / leaky-ReLU transforms
__device__ __forceinline__ float elem(const f2g_operand& S, const RowCtx& rc, bool rowok, int c,
                                      const Mg& mg) {
  const int seglen = mg.seglen;
  const int sg = fast_div(c, seglen, mg.seg), oo = c - sg * seglen;
  const int l1 = rc.l1b + sg;
  int off = rc.e0 + oo;
  bool ok = rowok && c < S.cols && (unsigned)l1 < (unsigned)S.L1;
  if (S.reflect) {
    off = off < 0 ? -off : off;
    off = off >= S.L0u ? 2 * (S.L0u - 1) - off : off;
  }
  ok = ok && (unsigned)off < (unsigned)S.L0u;
  const long long a = ok ? rc.base + (long long)l1 * S.line_stride + off : 0;
  float v = S.base[a];
  if (S.lrelu_src) v *= S.lrelu_src[a] > 0.f ? 1.f : S.lrelu_slope;
  v = ok ? v : 0.f;
  if (S.alpha) { const float al = S.alpha[c < S.cols ? c : 0]; v = v > 0.f ? v : al * v; }
  return v;
}

__device__ __forceinline__ long long out_offset(const f2g_epilogue& E, int row, int col) {
  if (E.P0o > 0) {
    const int sq = row / E.P0o;
    return (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o +
           E.off_o + col;
  }
  return (long long)row * E.ldc + col;
}

// 4 consecutive window columns of a decoded row: one vector load when the chunk is interior and
// 16-byte aligned, else element-wise
__device__ __forceinline__ float4 chunk(const f2g_operand& S, const RowCtx& rc, int c,
                                        const Mg& mg) {
  const int seglen = mg.seglen;
  const int sg = fast_div(c, seglen, mg.seg), oo = c - sg * seglen;
  const int l1 = rc.l1b + sg, e = rc.e0 + oo;
  if (!S.alpha && !S.lrelu_src && c + 3 < S.cols && oo + 3 < seglen &&
      (unsigned)l1 < (unsigned)S.L1 && e >= 0 && e + 3 < S.L0u) {
    const float* p = S.base + rc.base + (long long)l1 * S.line_stride + e;
    if ((((uintptr_t)p) & 15) == 0) return *reinterpret_cast<const float4*>(p);
  }
  return make_float4(elem(S, rc, true, c, mg), elem(S, rc, true, c + 1, mg),
                     elem(S, rc, true, c + 2, mg), elem(S, rc, true, c + 3, mg));
}

// same with the column decode (seg, o) of c done by the caller: it does not depend on the row, so
// the kernels below compute it once per thread instead of once per row
__device__ __forceinline__ float4 chunk_at(const f2g_operand& S, const RowCtx& rc, int c, int sg,
                                           int oo, const Mg& mg) {
  const int l1 = rc.l1b + sg, e = rc.e0 + oo;
  if (!S.alpha && !S.lrelu_src && c + 3 < S.cols && oo + 3 < mg.seglen &&
      (unsigned)l1 < (unsigned)S.L1 && e >= 0 && e + 3 < S.L0u) {
    const float* p = S.base + rc.base + (long long)l1 * S.line_stride + e;
    if ((((uintptr_t)p) & 15) == 0) return *reinterpret_cast<const float4*>(p);
  }
  return make_float4(elem(S, rc, true, c, mg), elem(S, rc, true, c + 1, mg),
                     elem(S, rc, true, c + 2, mg), elem(S, rc, true, c + 3, mg));
}

// forms 0 / 1 with N <= 4: the (K x N) weight panel is staged once per block in LDS as [n][k];
// each wave then walks ROWS_PER_WAVE rows, lanes striding K in float4 chunks, wave-shuffle reduce.
constexpr int ROWS_PER_WAVE = 16;  // (4 was measured: +10 % for the 1-column forward, -10 % for the
                                   // 2-column data gradient whose strided weight staging it repeats)

template <bool F1>
__global__ __launch_bounds__(256) void narrow_rows_kernel(const f2g_gemm_desc d, int M, int N,
                                                          int K) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [N][K4], K4 = K rounded up to 4
  const int K4 = (K + 3) & ~3;
  const long long ldb = d.B.seq_stride;
  for (int i = threadIdx.x; i < N * K4; i += 256) {
    const int n = i / K4, k = i - n * K4;
    float v = 0.f;
    if (k < K) v = F1 ? d.B.base[(long long)k * ldb + n] : d.B.base[(long long)n * ldb + k];
    wl[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f2g_epilogue& E = d.E;
  const float scale = E.scale != 0.f ? E.scale : 1.f;
  const Mg mg = magics(d.A);
  const long long rbase = ((long long)blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const long long row = rbase + rr;
    if (row >= M) break;
    const RowCtx rc = decode_row(d.A, (int)row, mg);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < K; k += 256) {
      const float4 a = chunk(d.A, rc, k, mg);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < N) {
          const float4 b = *reinterpret_cast<const float4*>(wl + n * K4 + k);
          acc[n] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = wave_sum(acc[n]);
    if (lane == 0) {
      for (int n = 0; n < N; ++n) {
        float v = acc[n] * scale + (E.bias ? E.bias[n] : 0.f);
        if (E.lrelu_slope != 0.f) v = v > 0.f ? v : E.lrelu_slope * v;
        const long long off = out_offset(E, (int)row, n);
        if (E.atomic) atomicAdd(E.C + off, v);
        else if (E.accumulate) E.C[off] += v;
        else E.C[off] = v;
        if (E.colsum) atomicAdd(E.colsum + n, v);
      }
    }
  }
}

// form 2 with M <= 4: block = run of `rows_per` rows, thread = one window column (strided)
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const f2g_gemm_desc d, int M, int N,
                                                           int R, int rows_per) {
  const int r0 = blockIdx.x * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > R) r1 = R;
  const long long lda = d.A.seq_stride;
  const Mg mg = magics(d.B);
  for (int c = threadIdx.x; c < N; c += 256) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
      const RowCtx rb = decode_row(d.B, r, mg);
      const float x = elem(d.B, rb, true, c, mg);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < M) acc[m] += d.A.base[(long long)r * lda + m] * x;
    }
    for (int m = 0; m < M; ++m) atomicAdd(d.E.C + out_offset(d.E, m, c), acc[m]);
  }
}

// same, N % 4 == 0 and N <= 4096: a thread owns up to 4 float4 chunks of the window (decoded once),
// the row decode is block-uniform, every row costs each thread <= 4 independent dwordx4 loads
template <int MM>  // gradient rows held in registers: 1 (conv_post: Cout = 1) or 4
__global__ __launch_bounds__(256) void narrow_wgrad4_kernel(const f2g_gemm_desc d, int M, int N,
                                                            int R, int rows_per) {
  const int r0 = blockIdx.x * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > R) r1 = R;
  const long long lda = d.A.seq_stride;
  const Mg mg = magics(d.B);
  int sgj[4], ooj[4];
  bool cj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (threadIdx.x + 256 * j) * 4;
    cj[j] = c < N;
    sgj[j] = fast_div(cj[j] ? c : 0, mg.seglen, mg.seg);
    ooj[j] = (cj[j] ? c : 0) - sgj[j] * mg.seglen;
  }
  float acc[MM][4][4];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[m][j][e] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const RowCtx rb = decode_row(d.B, r, mg);
    float a[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) a[m] = m < M ? d.A.base[(long long)r * lda + m] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cj[j]) {
        const float4 x = chunk_at(d.B, rb, (threadIdx.x + 256 * j) * 4, sgj[j], ooj[j], mg);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
          acc[m][j][0] += a[m] * x.x; acc[m][j][1] += a[m] * x.y;
          acc[m][j][2] += a[m] * x.z; acc[m][j][3] += a[m] * x.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!cj[j]) continue;
    const int c = (threadIdx.x + 256 * j) * 4;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (m < M) atomicAdd(d.E.C + out_offset(d.E, m, c + e), acc[m][j][e]);
  }
}

bool plainish(const f2g_operand& S) {
  return S.P0 == 1 && S.P1 == 1 && S.seglen >= S.cols && S.L1 == 1 && S.pad0 == 0 &&
         S.pad1 == 0 && S.L0u >= S.cols && !S.reflect && !S.lrelu_src && !S.alpha;
}

}  // namespace

// Returns 1 when the problem was handled here, 0 when the caller should use the MFMA kernels,
// negative on error.
int f2g_gemm_narrow(const f2g_gemm_desc& d, hipStream_t st) {
  const f2g_epilogue& E = d.E;
  // operands / epilogue fields these VALU kernels do not implement go to the generic path, which
  // applies them or rejects the descriptor
  if (E.res || E.aux || E.colsum_alpha || E.prelu_slope || E.mask_src || E.fm_ref || E.c_bf16 ||
      d.A.split || d.B.split)
    return 0;
  if (d.form == 0 || d.form == 1) {
    const bool f1 = d.form == 1;
    const int M = d.A.rows, N = f1 ? d.B.cols : d.B.rows, K = d.A.cols;
    if (N > 4 || !plainish(d.B) || M <= 0 || (size_t)N * ((K + 3) & ~3) * 4 > 60000) return 0;
    const size_t smem = (size_t)N * ((K + 3) & ~3) * sizeof(float);
    dim3 grid((M + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE));
    if (f1) hipLaunchKernelGGL(narrow_rows_kernel<true>, grid, dim3(256), smem, st, d, M, N, K);
    else hipLaunchKernelGGL(narrow_rows_kernel<false>, grid, dim3(256), smem, st, d, M, N, K);
    int rc = f2g_check_launch();
    return rc ? rc : 1;
  }
  if (d.form == 2) {
    const int M = d.A.cols, N = d.B.cols, R = d.A.rows;
    if (M > 4 || !plainish(d.A) || !E.atomic || E.bias || E.colsum || R <= 0) return 0;
    int rows_per = R / 2048;
    if (rows_per < 64) rows_per = 64;
    if (rows_per > 512) rows_per = 512;
    dim3 grid((R + rows_per - 1) / rows_per);
    if ((N & 3) == 0 && N <= 4096 && M == 1)
      hipLaunchKernelGGL(narrow_wgrad4_kernel<1>, grid, dim3(256), 0, st, d, M, N, R, rows_per);
    else if ((N & 3) == 0 && N <= 4096)
      hipLaunchKernelGGL(narrow_wgrad4_kernel<4>, grid, dim3(256), 0, st, d, M, N, R, rows_per);
    else
      hipLaunchKernelGGL(narrow_wgrad_kernel, grid, dim3(256), 0, st, d, M, N, R, rows_per);
    int rc = f2g_check_launch();
    return rc ? rc : 1;
  }
  return 0;
}
