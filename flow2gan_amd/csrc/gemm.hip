// Implicit-GEMM family on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every GEMM-shaped op of the hot path (pointwise 1x1 convs, the
// k=3 cond-encoder conv, MPD (5,1)/(3,1) convs, MRD (3,9)/(3,3) convs, DFT / inverse-DFT
// matrices for STFT/iSTFT, mel / linear filterbanks) in three forms: forward, data gradient,
// weight gradient.  Operands are described by f2g_operand (include/flow2gan_hip.h): a
// channels-last tensor viewed as rows = pixels, cols = contiguous window (never materialised).
//
// Tiling (wave64): block = 4 waves; each wave owns TM x TN tiles of 32x32, accumulated in
// f32x16 registers by mfma_f32_32x32x2f32.  The 2 k-slots of that instruction are fed from the
// two lane halves: lanes 0-31 walk k in [0,16) of the BK=32 slab, lanes 32-63 walk [16,32), so a
// lane's operands for 4 consecutive MFMAs are one ds_read_b128 (row-major LDS tile) or four
// conflict-free ds_read_b32 (k-major tile).  Global->register->LDS double buffering, one
// barrier per K slab.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDR = BK + 4;  // row-major LDS tile leading dim (conflict-free ds_read_b128)

struct RowCtx {
  long long base;
  int l1b, e0;
};

__device__ __forceinline__ RowCtx decode_row(const f2g_operand& S, int r) {
  RowCtx rc;
  int s, p1, p0;
  if (S.P0 == 1 && S.P1 == 1) {
    s = r; p1 = 0; p0 = 0;
  } else {
    int q = r / S.P0;
    p0 = r - q * S.P0;
    s = q / S.P1;
    p1 = q - s * S.P1;
  }
  rc.base = (long long)s * S.seq_stride;
  rc.l1b = p1 * S.step1 - S.pad1;
  rc.e0 = (p0 * S.step0 - S.pad0) * S.unit;
  return rc;
}

__device__ __forceinline__ float fix_elem(const f2g_operand& S, float v, long long off, int c) {
  if (S.lrelu_src) v *= (S.lrelu_src[off] > 0.f ? 1.f : S.lrelu_slope);
  if (S.alpha) { float al = S.alpha[c]; v = v > 0.f ? v : al * v; }
  return v;
}

__device__ __forceinline__ float load_elem(const f2g_operand& S, const RowCtx& rc, int c) {
  if (c >= S.cols) return 0.f;
  int seg = 0, o = c;
  if (S.seglen < S.cols) { seg = c / S.seglen; o = c - seg * S.seglen; }
  int l1 = rc.l1b + seg;
  if ((unsigned)l1 >= (unsigned)S.L1) return 0.f;
  int e = rc.e0 + o;
  if (S.reflect) {
    if (e < 0) e = -e;
    else if (e >= S.L0u) e = 2 * (S.L0u - 1) - e;
    if ((unsigned)e >= (unsigned)S.L0u) return 0.f;
  } else if ((unsigned)e >= (unsigned)S.L0u) {
    return 0.f;
  }
  long long off = rc.base + (long long)l1 * S.line_stride + e;
  return fix_elem(S, S.base[off], off, c);
}

// 4 consecutive window columns c..c+3 of row r (zero outside the operand).
__device__ __forceinline__ float4 load_chunk(const f2g_operand& S, int r, int c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= S.rows || c >= S.cols) return v;
  RowCtx rc = decode_row(S, r);
  int seglen = S.seglen < S.cols ? S.seglen : S.cols;
  int seg = 0, o = c;
  if (S.seglen < S.cols) { seg = c / S.seglen; o = c - seg * S.seglen; }
  int l1 = rc.l1b + seg;
  int e = rc.e0 + o;
  if (o + 3 < seglen && (unsigned)l1 < (unsigned)S.L1 && e >= 0 && e + 3 < S.L0u) {
    long long off = rc.base + (long long)l1 * S.line_stride + e;
    const float* p = S.base + off;
    if ((((uintptr_t)p) & 15) == 0) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
    }
    if (S.lrelu_src || S.alpha) {
      v.x = fix_elem(S, v.x, off, c);
      v.y = fix_elem(S, v.y, off + 1, c + 1);
      v.z = fix_elem(S, v.z, off + 2, c + 2);
      v.w = fix_elem(S, v.w, off + 3, c + 3);
    }
    return v;
  }
  v.x = load_elem(S, rc, c);
  v.y = load_elem(S, rc, c + 1);
  v.z = load_elem(S, rc, c + 2);
  v.w = load_elem(S, rc, c + 3);
  return v;
}

// Tile staging: TROWS x TCOLS floats, 256 threads, float4 chunks along the contiguous axis.
template <int TROWS, int TCOLS>
struct Stage {
  static constexpr int CH = TCOLS / 4;
  static constexpr int NLD = (TROWS * CH) / 256;
  static_assert((TROWS * CH) % 256 == 0, "tile must divide among 256 threads");
  float4 r[NLD];
  __device__ __forceinline__ void load(const f2g_operand& S, int row0, int col0, int tid) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      int idx = tid + 256 * q;
      int row = idx / CH, ch = idx - row * CH;
      r[q] = load_chunk(S, row0 + row, col0 + ch * 4);
    }
  }
  __device__ __forceinline__ void store(float* lds, int ld, int tid) const {
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      int idx = tid + 256 * q;
      int row = idx / CH, ch = idx - row * CH;
      *reinterpret_cast<float4*>(lds + row * ld + ch * 4) = r[q];
    }
  }
};

template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                   int kchunk) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int LDA = AKM ? BM : LDR;
  constexpr int LDB = BKM ? BN : LDR;
  constexpr int ASZ = AKM ? BK * BM : BM * LDR;
  constexpr int BSZ = BKM ? BK * BN : BN * LDR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;            // 2 buffers
  float* Bs = smem + 2 * ASZ;  // 2 buffers

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  using SA = typename std::conditional<AKM, Stage<BK, BM>, Stage<BM, BK>>::type;
  using SB = typename std::conditional<BKM, Stage<BK, BN>, Stage<BN, BK>>::type;
  SA sa;
  SB sb;

  auto gload = [&](int k0) {
    if (AKM) sa.load(d.A, k0, m0, tid); else sa.load(d.A, m0, k0, tid);
    if (BKM) sb.load(d.B, k0, n0, tid); else sb.load(d.B, n0, k0, tid);
  };
  auto lstore = [&](int buf) {
    sa.store(As + buf * ASZ, LDA, tid);
    sb.store(Bs + buf * BSZ, LDB, tid);
  };

  if (nt > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) gload(kbeg + (t + 1) * BK);
    const float* Ab = As + cur * ASZ;
    const float* Bb = Bs + cur * BSZ;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float a[TM][4], b[TN][4];
      const int kk = h * 16 + s4 * 4;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + li;
        if (AKM) {
#pragma unroll
          for (int q = 0; q < 4; ++q) a[mi][q] = Ab[(kk + q) * LDA + row];
        } else {
          float4 tv = *reinterpret_cast<const float4*>(Ab + row * LDA + kk);
          a[mi][0] = tv.x; a[mi][1] = tv.y; a[mi][2] = tv.z; a[mi][3] = tv.w;
        }
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = (wn * TN + ni) * 32 + li;
        if (BKM) {
#pragma unroll
          for (int q = 0; q < 4; ++q) b[ni][q] = Bb[(kk + q) * LDB + col];
        } else {
          float4 tv = *reinterpret_cast<const float4*>(Bb + col * LDB + kk);
          b[ni][0] = tv.x; b[ni][1] = tv.y; b[ni][2] = tv.z; b[ni][3] = tv.w;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni],
                                                                0, 0, 0);
    }
    if (t + 1 < nt) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  const f2g_epilogue& E = d.E;
  const float scale = E.scale != 0.f ? E.scale : 1.f;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + (wn * TN + ni) * 32 + li;
    const bool cok = col < N;
    float bias = 0.f, gam = 0.f, aln = 0.f;
    if (cok) {
      if (E.bias) bias = E.bias[col];
      if (E.res) gam = E.gamma ? E.gamma[col] : 1.f;
      if (E.aux) aln = E.alpha_n[col];
    }
    float cs = 0.f, csa = 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + (wm * TM + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (!cok || row >= M) continue;
        float v = acc[mi][ni][e] * scale + bias;
        if (E.res) v += gam * E.res[(long long)row * E.ldres + col];
        if (E.aux) {
          float av = E.aux[(long long)row * E.ldaux + col];
          csa += v * fminf(av, 0.f);
          v *= (av > 0.f ? 1.f : aln);
        }
        if (E.lrelu_slope != 0.f) v = v > 0.f ? v : E.lrelu_slope * v;
        cs += v;
        long long off;
        if (E.P0o > 0) {
          int sq = row / E.P0o;
          off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o +
                E.off_o + col;
        } else {
          off = (long long)row * E.ldc + col;
        }
        if (E.atomic) atomicAdd(E.C + off, v);
        else if (E.accumulate) E.C[off] += v;
        else E.C[off] = v;
      }
    }
    if (E.colsum || E.colsum_alpha) {
      cs += __shfl_xor(cs, 32);
      csa += __shfl_xor(csa, 32);
      if (cok && h == 0) {
        if (E.colsum) atomicAdd(E.colsum + col, cs);
        if (E.colsum_alpha) atomicAdd(E.colsum_alpha + col, csa);
      }
    }
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM>
int launch(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int ASZ = AKM ? BK * BM : BM * LDR;
  constexpr int BSZ = BKM ? BK * BN : BN * LDR;
  constexpr size_t smem = (size_t)2 * (ASZ + BSZ) * sizeof(float);
  int kchunk = ((K + split - 1) / split + BK - 1) / BK * BK;
  if (kchunk < BK) kchunk = BK;
  int zs = (K + kchunk - 1) / kchunk;
  if (zs < 1) zs = 1;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, zs);
  if (grid.x == 0 || grid.y == 0) return F2G_OK;
  auto kern = gemm_kernel<WAVES_M, WAVES_N, TM, TN, AKM, BKM>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  return f2g_check_launch();
}

template <bool AKM, bool BKM>
int dispatch(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  if (AKM && M <= 32) return launch<1, 4, 1, 2, AKM, BKM>(d, M, N, K, split, st);  // 32 x 256
  if (N <= 32) return launch<4, 1, 2, 1, AKM, BKM>(d, M, N, K, split, st);         // 256 x 32
  if (N <= 64) return launch<4, 1, 1, 2, AKM, BKM>(d, M, N, K, split, st);         // 128 x 64
  return launch<2, 2, 2, 2, AKM, BKM>(d, M, N, K, split, st);                      // 128 x 128
}

}  // namespace

extern "C" int f2g_gemm(const f2g_gemm_desc* dp, f2g_stream_t stream) {
  if (!dp || !dp->A.base || !dp->B.base || !dp->E.C) return F2G_EINVAL;
  const f2g_gemm_desc& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  int split = d.split_k > 0 ? d.split_k : 1;
  if (d.form == 0) {
    if (d.A.cols != d.B.cols) return F2G_EINVAL;
    return dispatch<false, false>(d, d.A.rows, d.B.rows, d.A.cols, 1, st);
  } else if (d.form == 1) {
    if (d.A.cols != d.B.rows) return F2G_EINVAL;
    return dispatch<false, true>(d, d.A.rows, d.B.cols, d.A.cols, 1, st);
  } else if (d.form == 2) {
    if (d.A.rows != d.B.rows) return F2G_EINVAL;
    if (split > 1 && !d.E.atomic) return F2G_EINVAL;
    return dispatch<true, true>(d, d.A.cols, d.B.cols, d.A.rows, split, st);
  }
  return F2G_EINVAL;
}
