// Implicit-GEMM family on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every GEMM-shaped op of the hot path (pointwise 1x1 convs, the
// k=3 cond-encoder conv, MPD (5,1)/(3,1) convs, MRD (3,9)/(3,3) convs, DFT / inverse-DFT
// matrices for STFT/iSTFT, mel / linear filterbanks) in three forms: forward, data gradient,
// weight gradient.  Operands are described by f2g_operand (include/flow2gan_hip.h): a
// channels-last tensor viewed as rows = pixels, cols = contiguous window (never materialised).
//
// Tiling (wave64): block = WAVES_M x WAVES_N waves (2, 4 or 8, chosen per operand kind in
// dispatch_tile); each wave owns TM x TN tiles of 32x32, accumulated in f32x16 registers by
// mfma_f32_32x32x2f32.  The 2 k-slots of that instruction are fed from the
// two lane halves: lanes 0-31 walk k in [0,16) of the BK=32 slab, lanes 32-63 walk [16,32), so a
// lane's operands for 4 consecutive MFMAs are one ds_read_b128 (row-major LDS tile) or four
// conflict-free ds_read_b32 (k-major tile).  Global->register->LDS double buffering, one
// barrier per K slab.
//
// Loaders are specialised at compile time: PLAIN operands (a row-major matrix: every pointwise
// GEMM, i.e. ~95 % of the FLOPs) keep one pointer per staged chunk and add a constant per K slab;
// GENERIC operands (windowed im2col views) decode rows once before the K loop and columns once
// per slab.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "common.h"
#include "x6_epilogue.h"

namespace {

int g_last_path = 0;   // kernel family of the last dispatch (f2g_gemm_last_path)

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
#ifndef F2G_LABVAR
#define F2G_LABVAR 0   // lab builds only (tools/micro/build_variants.sh): ablations of the bf16 lean K loop
#endif
constexpr int LDR = BK + 4;  // row-major LDS tile leading dim (conflict-free ds_read_b128)

// ------------------------------------------------------------------------------------------
// generic (windowed) element access
// ------------------------------------------------------------------------------------------
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

__device__ __forceinline__ float prelu1(float v, float a) { return v > 0.f ? v : a * v; }

// ------------------------------------------------------------------------------------------
// Tile loaders.  A tile is TROWS x TCOLS floats in memory orientation, staged by 256 threads as
// float4 chunks: chunk idx = tid + 256*q -> (row = idx / CH, ch = idx % CH).  CH divides 256,
// so `ch` is the same for all of a thread's chunks and rows advance by 256/CH.
//   KM = false: rows are the tile's m/n index (fixed), cols walk K   -> advance along columns
//   KM = true : rows walk K (the reduction), cols are m/n (fixed)    -> advance along rows
// MODE (chosen on the host from the operand descriptor):
//   PF  plain matrix, 16-byte aligned rows, reduction extent % BK == 0: every chunk is ONE
//       unconditional global_load_dwordx4 through a clamped pointer; masks + PReLU in store().
//   GF  windowed operand whose offsets are all multiples of 4 floats: one clamped vector load
//       per chunk + a validity bit; leaky-ReLU derivative / PReLU applied in store().
//   SL  anything else (misaligned, reflect padding, odd extents): element-wise predicated.
// store() runs after the slab's MFMAs, so the transforms never wait on the load latency.
// ------------------------------------------------------------------------------------------
enum { PF = 0, GF = 1, SL = 2, GR = 3 };  // GR = GF + reflect padding (STFT framing)

// what an out-of-window chunk of a GF operand reads (16 aligned bytes of zeros)
// (not const: the compiler must keep the address select instead of folding a select of values)
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// v = hi + lo + O(2^-17 |v|): hi = round-to-nearest bf16 of v, lo = bf16 of the remainder
__device__ __forceinline__ void split_bf16(float v, unsigned short& hi, unsigned short& lo) {
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

template <int MODE, bool KM, int TROWS, int TCOLS, int NT = 256>
struct Loader {
  static constexpr int CH = TCOLS / 4;
  static constexpr int NCHUNK = TROWS * CH;
  static constexpr bool PARTIAL = NCHUNK < NT;  // fewer chunks than threads: the rest idle
  static constexpr int NLD = PARTIAL ? 1 : NCHUNK / NT;
  static constexpr int RSTEP = NT / CH;
  static_assert((PARTIAL || NCHUNK % NT == 0) && NT % CH == 0, "tile must divide among the threads");
  bool active;

  // Staging registers of ONE slab.  Kept out of the loader object and declared per loop iteration
  // in the kernels, so that nothing is loop-carried and the compiler can leave the loads in flight
  // across the MFMA phase.
  struct Stg {
    float4 r[NLD];
    float4 r2[MODE != PF ? NLD : 1];  // lrelu_src values of the chunk
    float4 a4;                        // PReLU slopes of the chunk's 4 columns
    unsigned vmask;                   // bit q: chunk q holds valid data
  };
  float4 a4fix;                       // KM: PReLU slopes of the thread's fixed columns
  unsigned cmask;                       // PF/KM: valid columns of the thread's fixed chunk
  unsigned rokm;                        // RM: bit q: row in range
  const float* p[MODE == PF ? NLD : 1];  // PF: chunk pointers, advanced per slab
  long long pstep;                      // PF: floats per unit of k (1 or ld)
  int kbase;                            // PF: k of the pointers p[]
  int c0;                               // first window column (RM: + k0 per slab; KM: fixed)
  int row0;                             // KM: first reduction row of this thread
  RowCtx rc[(MODE != PF && !KM) ? NLD : 1];  // generic RM: decoded rows
  long long rb[(MODE != PF && !KM) ? NLD : 1];  // generic RM: offset of the row's window origin
  int seg, o;                           // generic KM: decoded fixed column
  unsigned mg_seg, mg_p0, mg_p1;        // generic: magic numbers of seglen / P0 / P1

  __device__ __forceinline__ void init(const f2g_operand& S, int tr0, int tc0, int kbeg, int tid) {
    active = !PARTIAL || tid < NCHUNK;
    if (MODE != PF) {
      mg_seg = magic_of(S.seglen);
      mg_p0 = magic_of(S.P0);
      mg_p1 = magic_of(S.P1);
    }
    const int ch = tid % CH, rr = active ? tid / CH : 0;
    rokm = 0; cmask = 0;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == PF) {
      const long long ld = S.seq_stride;
      if (!KM) {
        c0 = ch * 4;
        pstep = 1;
        kbase = kbeg;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          const int row = tr0 + rr + RSTEP * q;
          const bool ok = row < S.rows;
          rokm |= (ok ? 1u : 0u) << q;
          p[q] = S.base + (long long)(ok ? row : 0) * ld + ch * 4 + kbeg;
        }
      } else {
        c0 = tc0 + ch * 4;
        row0 = rr;
        pstep = ld;
        kbase = kbeg;
#pragma unroll
        for (int j = 0; j < 4; ++j) cmask |= (c0 + j < S.cols ? 1u : 0u) << j;
#pragma unroll
        for (int q = 0; q < NLD; ++q)
          p[q] = S.base + (long long)(kbeg + rr * NLD + q) * ld + (cmask ? c0 : 0);
        if (S.alpha) {
          if (cmask & 1) a4.x = S.alpha[c0];
          if (cmask & 2) a4.y = S.alpha[c0 + 1];
          if (cmask & 4) a4.z = S.alpha[c0 + 2];
          if (cmask & 8) a4.w = S.alpha[c0 + 3];
        }
      }
    } else {
      if (!KM) {
        c0 = ch * 4;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          const int row = tr0 + rr + RSTEP * q;
          const bool ok = row < S.rows;
          rokm |= (ok ? 1u : 0u) << q;
          rc[q] = decode_row(S, ok ? row : 0);
          rb[q] = rc[q].base + (long long)rc[q].l1b * S.line_stride + rc[q].e0;
        }
      } else {
        c0 = tc0 + ch * 4;
        row0 = rr;
        seg = 0; o = c0;
        if (S.seglen < S.cols) { seg = c0 / S.seglen; o = c0 - seg * S.seglen; }
        if (MODE != PF && S.alpha) {
          if (c0 < S.cols) a4.x = S.alpha[c0];
          if (c0 + 1 < S.cols) a4.y = S.alpha[c0 + 1];
          if (c0 + 2 < S.cols) a4.z = S.alpha[c0 + 2];
          if (c0 + 3 < S.cols) a4.w = S.alpha[c0 + 3];
        }
      }
    }
    a4fix = a4;
  }

  __device__ __forceinline__ void gchunk(const f2g_operand& S, Stg& g, int q, bool rowok,
                                         const RowCtx& rcx, int c, int sg, int oo) {
    if (MODE == GF || MODE == GR) {
      // out-of-window chunks are read from a 16-byte block of zeros: nothing to mask afterwards
      // (off = offset of the chunk relative to S.base, precombined by the caller)
      const int l1 = rcx.l1b + sg, e = rcx.e0 + oo;
      const bool v = rowok && c < S.cols && (unsigned)l1 < (unsigned)S.L1 && e >= 0 &&
                     e + 3 < S.L0u;
      const long long off = rcx.base;
      g.r[q] = *reinterpret_cast<const float4*>(v ? S.base + off : g_zero16);
      if (S.lrelu_src)
        g.r2[q] = *reinterpret_cast<const float4*>(v ? S.lrelu_src + off : g_zero16);
      if (MODE == GR) {
        // STFT framing (center=True, reflect): only the chunks that straddle an end of the
        // sequence -- the first / last two frames -- take this element-wise mirrored path
        const bool inwin = rowok && c < S.cols && (unsigned)l1 < (unsigned)S.L1;
        if (inwin && !v) {
          const long long rowb = off - e;
          float t[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            int ee = e + k;
            ee = ee < 0 ? -ee : ee;
            ee = ee >= S.L0u ? 2 * (S.L0u - 1) - ee : ee;
            const bool ok = (unsigned)ee < (unsigned)S.L0u;
            t[k] = ok ? S.base[rowb + (ok ? ee : 0)] : 0.f;
          }
          g.r[q] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    } else {
      // SL: the same clamped-address scheme element by element (4 unconditional scalar loads, no
      // divergent branches); handles reflect padding, odd segment lengths and misaligned rows.
      const int seglen = S.seglen < S.cols ? S.seglen : S.cols;
      float vv[4], lv[4];
      int sg_e = sg, oo_e = oo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int l1 = rcx.l1b + sg_e;
        int off_e = rcx.e0 + oo_e;
        bool ok = rowok && (c + e) < S.cols && (unsigned)l1 < (unsigned)S.L1;
        if (S.reflect) {
          off_e = off_e < 0 ? -off_e : off_e;
          off_e = off_e >= S.L0u ? 2 * (S.L0u - 1) - off_e : off_e;
        }
        ok = ok && (unsigned)off_e < (unsigned)S.L0u;
        const long long a = ok ? rcx.base + (long long)l1 * S.line_stride + off_e : 0;
        const float x = S.base[a];
        lv[e] = S.lrelu_src ? S.lrelu_src[a] : 1.f;
        vv[e] = ok ? x : 0.f;
        ++oo_e;
        const bool wrap = oo_e >= seglen;
        oo_e = wrap ? 0 : oo_e;
        sg_e += wrap ? 1 : 0;
      }
      g.r[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      g.r2[q] = make_float4(lv[0], lv[1], lv[2], lv[3]);
      g.vmask |= 1u << q;
    }
  }

  __device__ __forceinline__ void load(const f2g_operand& S, int k0, Stg& g) {
    g.a4 = a4fix;
    if (MODE == PF) {
      if (!KM) {
        // PReLU slopes of this slab's columns (full slabs only: c+3 < cols); unconditional load
        // through a valid dummy address keeps it off the control-flow / waitcnt path
        const float* ap = S.alpha ? S.alpha + (c0 + k0) : S.base;
        g.a4 = *reinterpret_cast<const float4*>(ap);
      }
      // explicit slab offset (no running pointers): the kernels issue the load unconditionally
      // with a clamped slab index, which keeps the loaded registers out of any control flow
      const long long adv = (long long)(k0 - kbase) * pstep;
#pragma unroll
      for (int q = 0; q < NLD; ++q) g.r[q] = *reinterpret_cast<const float4*>(p[q] + adv);
      g.vmask = KM ? (cmask ? ~0u : 0u) : rokm;
    } else {
      g.vmask = 0;
      if (!KM) {
        const int c = c0 + k0;
        int sg = 0, oo = c;
        if (S.seglen < S.cols) { sg = fast_div(c, S.seglen, mg_seg); oo = c - sg * S.seglen; }
        if (MODE != PF && S.alpha) {
          g.a4.x = c < S.cols ? S.alpha[c] : 0.f;
          g.a4.y = c + 1 < S.cols ? S.alpha[c + 1] : 0.f;
          g.a4.z = c + 2 < S.cols ? S.alpha[c + 2] : 0.f;
          g.a4.w = c + 3 < S.cols ? S.alpha[c + 3] : 0.f;
        }
        const int so = sg * (int)S.line_stride + oo;  // the slab's offset inside a row's window
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          RowCtx rx = rc[q];
          if (MODE == GF || MODE == GR) rx.base = rb[q] + so;
          gchunk(S, g, q, (rokm >> q) & 1, rx, c, sg, oo);
        }
      } else {
        // the thread's NLD rows are consecutive pixels: decode the first (two divisions), then
        // step (p0, p1, s) with carries
        const int rfirst = k0 + row0 * NLD;
        int s_, p1_, p0_;
        {
          const int r = rfirst < S.rows ? rfirst : 0;
          if (S.P0 == 1 && S.P1 == 1) { s_ = r; p1_ = 0; p0_ = 0; }
          else {
            const int qq = fast_div(r, S.P0, mg_p0);
            p0_ = r - qq * S.P0;
            s_ = fast_div(qq, S.P1, mg_p1);
            p1_ = qq - s_ * S.P1;
          }
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          const bool ok = rfirst + q < S.rows;
          RowCtx rcx;
          rcx.base = (long long)s_ * S.seq_stride;
          rcx.l1b = p1_ * S.step1 - S.pad1;
          rcx.e0 = (p0_ * S.step0 - S.pad0) * S.unit;
          if (MODE == GF || MODE == GR)
            rcx.base += (long long)(rcx.l1b + seg) * S.line_stride + (rcx.e0 + o);
          gchunk(S, g, q, ok, rcx, c0, seg, o);
          ++p0_;
          const bool c0w = p0_ >= S.P0;
          p0_ = c0w ? 0 : p0_;
          p1_ += c0w ? 1 : 0;
          const bool c1w = p1_ >= S.P1;
          p1_ = c1w ? 0 : p1_;
          s_ += c1w ? 1 : 0;
        }
      }
    }
  }

  // masks + on-load transforms of chunk q (runs after the slab's MFMAs)
  __device__ __forceinline__ float4 finalize(const f2g_operand& S, const Stg& g, int q) const {
    float4 v = g.r[q];
    const float4 a4 = g.a4;
    if (MODE != GF && MODE != GR && !((g.vmask >> q) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == PF && KM) {  // column tail of the fixed chunk
      if (!(cmask & 1)) v.x = 0.f;
      if (!(cmask & 2)) v.y = 0.f;
      if (!(cmask & 4)) v.z = 0.f;
      if (!(cmask & 8)) v.w = 0.f;
    }
    if (MODE != PF && S.lrelu_src) {
      const float sl = S.lrelu_slope;
      v.x *= g.r2[q].x > 0.f ? 1.f : sl; v.y *= g.r2[q].y > 0.f ? 1.f : sl;
      v.z *= g.r2[q].z > 0.f ? 1.f : sl; v.w *= g.r2[q].w > 0.f ? 1.f : sl;
    }
    if (S.alpha) {
      v.x = prelu1(v.x, a4.x); v.y = prelu1(v.y, a4.y);
      v.z = prelu1(v.z, a4.z); v.w = prelu1(v.w, a4.w);
    }
    return v;
  }

  // fp32 LDS image in memory orientation: [tile row][tile col]
  __device__ __forceinline__ void store(const f2g_operand& S, const Stg& g, float* lds, int ld,
                                        int tid) const {
    if (PARTIAL && !active) return;
    const int ch = tid % CH, rr = tid / CH;
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int row = KM ? rr * NLD + q : rr + RSTEP * q;
      *reinterpret_cast<float4*>(lds + row * ld + ch * 4) = finalize(S, g, q);
    }
  }

  // split-bf16 LDS image, always [m-or-n index][k] (64-byte rows of 32 bf16, 16-byte chunks
  // XOR-swizzled by (row>>2)&3): hi = bf16(v), lo = bf16(v - hi).  k-major tiles are transposed
  // here: a thread owns NLD consecutive k of 4 columns and packs them per column.
  __device__ __forceinline__ void store_split(const f2g_operand& S, const Stg& g, unsigned char* hi,
                                              unsigned char* lo, int tid, bool with_lo) const {
    if (PARTIAL && !active) return;
    const int ch = tid % CH, rr = tid / CH;
    if (!KM) {
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const float4 v = finalize(S, g, q);
        const int row = rr + RSTEP * q;
        const int k4 = ch * 4;
        const int off = row * 64 + ((((k4 >> 3) ^ ((row >> 2) & 3))) << 4) + ((k4 >> 2) & 1) * 8;
        unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
        split_bf16(v.x, h0, l0); split_bf16(v.y, h1, l1);
        split_bf16(v.z, h2, l2); split_bf16(v.w, h3, l3);
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        if (with_lo)
          *reinterpret_cast<uint2*>(lo + off) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
      }
    } else {
      unsigned short hs[4][NLD], ls[4][NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const float4 v = finalize(S, g, q);
        split_bf16(v.x, hs[0][q], ls[0][q]); split_bf16(v.y, hs[1][q], ls[1][q]);
        split_bf16(v.z, hs[2][q], ls[2][q]); split_bf16(v.w, hs[3][q], ls[3][q]);
      }
      const int k0 = rr * NLD;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int row = ch * 4 + c;
        const int off = row * 64 + ((((k0 >> 3) ^ ((row >> 2) & 3))) << 4) + (k0 & 7) * 2;
        if (NLD == 1) {
          *reinterpret_cast<unsigned short*>(hi + off) = hs[c][0];
          if (with_lo) *reinterpret_cast<unsigned short*>(lo + off) = ls[c][0];
        } else {
#pragma unroll
          for (int q = 0; q < NLD; q += 2) {
            *reinterpret_cast<unsigned*>(hi + off + q * 2) = hs[c][q] | (hs[c][q + 1 < NLD ? q + 1 : q] << 16);
            if (with_lo)
              *reinterpret_cast<unsigned*>(lo + off + q * 2) = ls[c][q] | (ls[c][q + 1 < NLD ? q + 1 : q] << 16);
          }
        }
      }
    }
  }
};

// XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles with the
// n index fastest so that the tiles sharing an A panel hit the same private L2.
__device__ __forceinline__ void tile_of_block(int BM, int BN, int& m0, int& n0) {
  const int tiles_n = gridDim.y, tiles_m = gridDim.x;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.y * tiles_m + blockIdx.x;
  const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  m0 = tm * BM;
  n0 = tn * BN;
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f2g_epilogue& E, f32x16 (&acc)[TM][TN], int M,
                                              int N, int m0, int n0, int wm, int wn, int li,
                                              int h, bool first) {
  const float scale = E.scale != 0.f ? E.scale : 1.f;
  const float fmw = E.fm_ref ? E.fm_w * (E.fm_wdev ? E.fm_wdev[0] : 1.f) : 0.f;
  // first: split-K / stream-K -- bias and residual enter once
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + (wn * TN + ni) * 32 + li;
    const bool cok = col < N;
    float bias = 0.f, gam = 0.f, aln = 0.f, pslope = 0.f;
    if (cok) {
      if (E.prelu_slope) pslope = E.prelu_slope[col];
      if (E.bias && first) bias = E.bias[col];
      if (E.res && first) gam = E.gamma ? E.gamma[col] : 1.f;
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
        if (E.res && first) v += gam * E.res[(long long)row * E.ldres + col];
        if (E.aux) {
          float av = E.aux[(long long)row * E.ldaux + col];
          csa += v * fminf(av, 0.f);
          v *= (av > 0.f ? 1.f : aln);
        }
        if (E.lrelu_slope != 0.f) v = v > 0.f ? v : E.lrelu_slope * v;
        if (E.prelu_slope) {
          const float pv = v > 0.f ? v : pslope * v;
          if (E.prelu_out) E.prelu_out[(long long)row * E.ld_prelu_out + col] = pv;
          else v = pv;
        }
        long long off;
        if (E.P0o > 0) {
          int sq = row / E.P0o;
          off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o +
                E.off_o + col;
        } else {
          off = (long long)row * E.ldc + col;
        }
        if (E.mask_src) {   // leaky-ReLU backward of the layer below (+ feature-matching term)
          const float y = E.mask_src[off];
          if (E.fm_ref) {
            const float dl = y - E.fm_ref[off];
            v += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
          }
          v *= y > 0.f ? 1.f : E.mask_slope;
        }
        cs += v;
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

// ---- exact fp32: v_mfma_f32_32x32x2_f32 --------------------------------------------------
template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM, int AMODE, int BMODE>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2)
void gemm_kernel(const f2g_gemm_desc d, int M, int N, int K, int kchunk) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
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
  int m0, n0;
  tile_of_block(BM, BN, m0, n0);
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

  using SA = typename std::conditional<AKM, Loader<AMODE, true, BK, BM, NT>,
                                       Loader<AMODE, false, BM, BK, NT>>::type;
  using SB = typename std::conditional<BKM, Loader<BMODE, true, BK, BN, NT>,
                                       Loader<BMODE, false, BN, BK, NT>>::type;
  SA sa;
  SB sb;
  if (AKM) sa.init(d.A, 0, m0, kbeg, tid); else sa.init(d.A, m0, 0, kbeg, tid);
  if (BKM) sb.init(d.B, 0, n0, kbeg, tid); else sb.init(d.B, n0, 0, kbeg, tid);

  if (nt > 0) {
    typename SA::Stg ga;
    typename SB::Stg gb;
    sa.load(d.A, kbeg, ga);
    sb.load(d.B, kbeg, gb);
    sa.store(d.A, ga, As, LDA, tid);
    sb.store(d.B, gb, Bs, LDB, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    typename SA::Stg ga;
    typename SB::Stg gb;
    {  // unconditional prefetch of the next slab (the last iteration re-reads slab 0, unused)
      const int kn = (t + 1 < nt) ? kbeg + (t + 1) * BK : kbeg;
      sa.load(d.A, kn, ga);
      sb.load(d.B, kn, gb);
    }
    __builtin_amdgcn_sched_barrier(0);
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
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nt) {
      sa.store(d.A, ga, As + (cur ^ 1) * ASZ, LDA, tid);
      sb.store(d.B, gb, Bs + (cur ^ 1) * BSZ, LDB, tid);
    }
    __syncthreads();
  }
  gemm_epilogue<TM, TN>(d.E, acc, M, N, m0, n0, wm, wn, li, h, blockIdx.z == 0);
}

// ---- split-bf16 ("bf16x3"): each fp32 operand is staged as hi + lo bf16 and every product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: relative error per
// product <= ~2^-16 (vs 2^-24 exact fp32, 2^-9 plain bf16) at 3/16 of the fp32-MFMA cycle cost.
// The MFMA phase of a slab is ~5x shorter than in the fp32 kernel, so latency is hidden with
// thread-level parallelism instead: 8 waves per block (wave tile 64x32 / 32x32, <=128 VGPRs),
// two blocks per CU = 4 waves per SIMD.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM, int AMODE, int BMODE>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N / 2)
void gemm_kernel_b3(const f2g_gemm_desc d, int M, int N, int K, int kchunk) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int ASZ = BM * 64;  // bytes of one bf16 image (hi or lo)
  constexpr int BSZ = BN * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* base = reinterpret_cast<unsigned char*>(smem);
  constexpr int BUF = 2 * ASZ + 2 * BSZ;  // per buffer: [A_hi | A_lo | B_hi | B_lo]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(BM, BN, m0, n0);
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

  using SA = typename std::conditional<AKM, Loader<AMODE, true, BK, BM, NT>,
                                       Loader<AMODE, false, BM, BK, NT>>::type;
  using SB = typename std::conditional<BKM, Loader<BMODE, true, BK, BN, NT>,
                                       Loader<BMODE, false, BN, BK, NT>>::type;
  SA sa;
  SB sb;
  if (AKM) sa.init(d.A, 0, m0, kbeg, tid); else sa.init(d.A, m0, 0, kbeg, tid);
  if (BKM) sb.init(d.B, 0, n0, kbeg, tid); else sb.init(d.B, n0, 0, kbeg, tid);

  // precision 1: hi/lo split, three MFMAs per product (fp32-class accuracy);
  // precision 2: hi only, one MFMA per product = plain bf16 inputs with fp32 accumulation
  const bool hl = d.precision == 1;
  auto compute = [&](const unsigned char* Ah) {
    const unsigned char* Al = Ah + ASZ;
    const unsigned char* Bh = Ah + 2 * ASZ;
    const unsigned char* Bl = Bh + BSZ;
    // per 16-k step: its fragments (one lgkmcnt wait), then the MFMAs back to back with the
    // accumulators interleaved so that consecutive MFMAs never depend on each other
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
      const int c = ks * 2 + h;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + li;
        const int off = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
        ah[mi] = *reinterpret_cast<const bf16x8*>(Ah + off);
        if (hl) al[mi] = *reinterpret_cast<const bf16x8*>(Al + off);
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int row = (wn * TN + ni) * 32 + li;
        const int off = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
        bh[ni] = *reinterpret_cast<const bf16x8*>(Bh + off);
        if (hl) bl[ni] = *reinterpret_cast<const bf16x8*>(Bl + off);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        if (term < 2 && !hl) continue;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            const bf16x8 av = term == 0 ? al[mi] : ah[mi];
            const bf16x8 bv = term == 1 ? bl[ni] : bh[ni];
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mi][ni], 0, 0, 0);
          }
      }
      __builtin_amdgcn_s_setprio(0);
    }
  };
  // clamped slab origin: loads are always issued (never inside control flow); out-of-range slabs
  // re-read slab 0 and are discarded
  auto kof = [&](int t) { return t < nt ? kbeg + t * BK : kbeg; };

  // Software pipeline, prefetch distance 1 (global -> registers during the MFMA phase, converted
  // into the other LDS buffer afterwards); latency is covered by 4 waves per SIMD.
  unsigned char* bufs[2] = {base, base + BUF};
  {
    typename SA::Stg ga;
    typename SB::Stg gb;
    sa.load(d.A, kof(0), ga);
    sb.load(d.B, kof(0), gb);
    if (nt > 0) {
      sa.store_split(d.A, ga, bufs[0], bufs[0] + ASZ, tid, hl);
      sb.store_split(d.B, gb, bufs[0] + 2 * ASZ, bufs[0] + 2 * ASZ + BSZ, tid, hl);
    }
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    typename SA::Stg ga;
    typename SB::Stg gb;
    sa.load(d.A, kof(t + 1), ga);
    sb.load(d.B, kof(t + 1), gb);
    __builtin_amdgcn_sched_barrier(0);
    compute(bufs[cur]);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nt) {
      unsigned char* nb = bufs[cur ^ 1];
      sa.store_split(d.A, ga, nb, nb + ASZ, tid, hl);
      sb.store_split(d.B, gb, nb + 2 * ASZ, nb + 2 * ASZ + BSZ, tid, hl);
    }
    __syncthreads();
  }
  gemm_epilogue<TM, TN>(d.E, acc, M, N, m0, n0, wm, wn, li, h, blockIdx.z == 0);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM, int AMODE, int BMODE,
          bool B3 = false>
int launch(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int ASZ = AKM ? BK * BM : BM * LDR;
  constexpr int BSZ = BKM ? BK * BN : BN * LDR;
  constexpr size_t smem = B3 ? (size_t)2 * (2 * BM * 64 + 2 * BN * 64)
                             : (size_t)2 * (ASZ + BSZ) * sizeof(float);
  int kchunk = ((K + split - 1) / split + BK - 1) / BK * BK;
  if (kchunk < BK) kchunk = BK;
  int zs = (K + kchunk - 1) / kchunk;
  if (zs < 1) zs = 1;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, zs);
  if (grid.x == 0 || grid.y == 0) return F2G_OK;
  static bool attr_done = false;
  if constexpr (B3) {
    auto kern = gemm_kernel_b3<WAVES_M, WAVES_N, TM, TN, AKM, BKM, AMODE, BMODE>;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), smem, st, d, M, N, K, kchunk);
  } else {
    auto kern = gemm_kernel<WAVES_M, WAVES_N, TM, TN, AKM, BKM, AMODE, BMODE>;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), smem, st, d, M, N, K, kchunk);
  }
  return f2g_check_launch();
}

template <bool AKM, bool BKM, int AMODE, int BMODE>
int dispatch_tile(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  // split-bf16 core: fast loader modes only; SL operands (small GEMMs) stay on exact fp32
  // (the reflect-padded STFT framing stays exact: small spectral bins are differences of large
  // terms, and the log-mel / spectral losses take their logarithm)
  if ((d.precision == 1 || d.precision == 2) && !d.A.reflect && !d.B.reflect) {
    if (AKM && M <= 32)
      return launch<1, 8, 1, 1, AKM, BKM, AMODE, BMODE, true>(d, M, N, K, split, st);  // 32 x 256
    if (N <= 32) return launch<8, 1, 1, 1, AKM, BKM, AMODE, BMODE, true>(d, M, N, K, split, st);
    if (N <= 64) return launch<4, 2, 1, 1, AKM, BKM, AMODE, BMODE, true>(d, M, N, K, split, st);
    // deep-K, wide-N problems (the 1024-channel MPD layers): 4 waves with 64x64 wave tiles read
    // less LDS per FLOP; everything else hides latency better with 8 waves of 64x32
    if (K >= 2048 && N >= 512)
      return launch<2, 2, 2, 2, AKM, BKM, AMODE, BMODE, true>(d, M, N, K, split, st);
    return launch<2, 4, 2, 1, AKM, BKM, AMODE, BMODE, true>(d, M, N, K, split, st);  // 128 x 128
  }
  // 32 x 256 for the weight gradients of 32-channel convs: 8 waves of one 32x32 tile each
  // (51 -> 57 TFLOP/s on the MRD band layers vs 4 waves of 32x64)
  if (AKM && M <= 32 && N <= 64)  // first MRD layer (2 -> 32 channels, 54 taps): 32 x 64, 2 waves
    return launch<1, 2, 1, 1, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
  if (AKM && M <= 32) return launch<1, 8, 1, 1, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
  // 128 x 32 (46 KB of LDS -> 3 blocks per CU); a 256 x 32 tile needs 83 KB and leaves ONE block
  // = one wave per SIMD on the CU, which cannot hide anything (measured 50 TFLOP/s on the
  // 32-channel MRD convs)
  if (N <= 32) return launch<4, 1, 1, 1, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
  if (N <= 64) return launch<4, 1, 1, 2, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
  // 128 x 128.  k-major LDS tiles (data / weight gradients) are read with four ds_read_b32 per
  // fragment instead of one ds_read_b128: 8 waves of 32x64 (4 waves per SIMD) hide that latency
  // (measured +5..20 % on every dgrad / wgrad shape); row-major forward tiles are best with 4
  // waves of 64x64 (least LDS traffic per MFMA).
  // (windowed forward operands -- the MPD convs -- spend VALU on im2col addressing: 8 waves too)
  if (AKM || BKM || AMODE == GF || AMODE == GR)
    return launch<4, 2, 1, 2, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
  return launch<2, 2, 2, 2, AKM, BKM, AMODE, BMODE>(d, M, N, K, split, st);
}

// ---- split-K for forward / data-gradient GEMMs ------------------------------------------------
// A 128x128 tiling of e.g. pwconv2 (M=6016, N=768) yields 282 blocks for 256 CUs: 26 CUs carry two
// tiles, the other 230 idle for half the kernel.  With a linear epilogue the reduction can be cut
// into s chunks whose partial tiles are added atomically onto a zeroed output (bias / residual
// enter through chunk 0): s*tiles blocks fill the last wave of resident blocks.
__global__ __launch_bounds__(256) void zero_out_kernel(const f2g_epilogue E, int M, int N) {
  const long long total = (long long)M * N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int row = (int)(i / N), col = (int)(i - (long long)row * N);
    long long off;
    if (E.P0o > 0) {
      const int sq = row / E.P0o;
      off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o +
            E.off_o + col;
    } else {
      off = (long long)row * E.ldc + col;
    }
    E.C[off] = 0.f;
  }
}

inline int auto_split(int M, int N, int K) {
  if (N <= 64) return 1;  // narrow tiles: thousands of blocks already
  // Measured (tools/splitk_sweep.py): splitting pays only for deep reductions (>= 64 slabs, each
  // chunk >= 20 slabs) and only through the fill of the last wave of 512 resident blocks:
  // 282 tiles x K 2304: s=3 +31 %; 1192 x 5120: s=3 +19 %; 188 x 6144: s=8 +22 %; every
  // shallower shape loses to the atomic epilogue.
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int nk = (K + BK - 1) / BK;
  // latency regime (batch-1 streaming synthesis, the per-item time MLPs): fewer tiles than half
  // the CUs -- every block is alone on its CU and the kernel lasts one block's K loop; cutting
  // that loop into <= 8 chunks of >= 8 slabs shortens it proportionally (chunked synthesis at
  // batch 1: 17.2 -> 12.1 ms per 1 s chunk, 9.1 ms replayed from a HIP graph)
  if (tiles * 2 <= 256 && nk >= 16) {
    int s = nk / 8;
    if (s > 8) s = 8;
    if ((long long)s * tiles > 256) s = (int)(256 / tiles);
    return s >= 2 ? s : 1;
  }
  if (nk < 64) return 1;
  auto eff = [&](int s) {
    const double w = (double)(tiles * s) / 512.0;
    return w / (double)((tiles * s + 511) / 512);
  };
  const double e1 = eff(1);
  double best = e1 + 0.15;
  int best_s = 1;
  for (int s = 2; s <= 8; ++s) {
    if (nk / s < 20) break;
    const double e = eff(s);
    if (e > best + 0.02) {
      best = e;
      best_s = s;
    }
  }
  return best_s;
}

inline bool host_plain(const f2g_operand& S) {
  return S.P0 == 1 && S.P1 == 1 && S.seglen >= S.cols && S.L1 == 1 && S.pad0 == 0 &&
         S.pad1 == 0 && S.L0u >= S.cols && !S.reflect && !S.lrelu_src;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Loader mode of an operand; `red_is_cols`: the reduction runs along the operand's columns.
inline int op_mode(const f2g_operand& S, bool red_is_cols) {
  if (host_plain(S)) {
    const int red = red_is_cols ? S.cols : S.rows;
    const bool ok = al16(S.base) && (S.seq_stride & 3) == 0 && S.cols >= 4 && red % BK == 0 &&
                    (!S.alpha || !red_is_cols || al16(S.alpha));
    if (ok) return PF;
  }
  const long long eu0 = (long long)S.step0 * S.unit, ep0 = (long long)S.pad0 * S.unit;
  const bool vec = al16(S.base) && (S.seq_stride & 3) == 0 && (S.line_stride & 3) == 0 &&
                   (S.seglen & 3) == 0 && (eu0 & 3) == 0 && (ep0 & 3) == 0 && (S.L0u & 3) == 0 &&
                   (S.cols & 3) == 0 && (!S.reflect || !S.lrelu_src) &&
                   (!S.lrelu_src || al16(S.lrelu_src)) && S.L0u >= 4;
  if (vec && S.reflect) return red_is_cols ? GR : SL;  // instantiated for forward A operands only
  return vec ? GF : SL;
}


// ---- lean forward kernel --------------------------------------------------------------------
// Measured on this part (tools/micro/gemm_lab.hip + PMC): v_mfma_f32_32x32x2_f32 occupies the
// vector ALU for 64 cycles, and with two waves per SIMD keeping that pipe full every OTHER VALU
// instruction issued on the SIMD costs ~37 cycles of it.  The generic loaders above spend 50-100
// VALU instructions per K slab on addresses, masks and on-load transforms (108 TFLOP/s on the
// 1024-channel MPD layers against 143 for the bare MFMA stream).  This kernel has NO vector
// ALU instruction inside the K loop:
//   * operands are read with buffer_load_dwordx4: resource (base, 2 GiB window) in SGPRs, a
//     per-thread CONSTANT byte offset per staged row (decoded once: sequence / line / position of
//     the im2col row), the K advance in a scalar register (SALU walks the window's segments);
//     rows past the end carry the offset 0x80000000 = out of range = the hardware returns zeros;
//   * LDS addresses are per-thread constants + immediates (K loop unrolled by two);
//   * the next slab is requested before the MFMA phase and written to LDS after it;
//   * the bias enters through the accumulator initialisation.
// It serves form 0 with a row-major B ([n][k] weights) and an A operand whose windows never leave
// their source (plain matrices, and conv windows over buffers that carry their zero padding as
// halo rows): every 1x1 conv, the MPD convs and their data gradients (transposed weights).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lean_row_offset(const f2g_operand& S, int r, int es = 4) {
  if (r >= S.rows) return 0x80000000u;
  long long off;
  if (S.P0 == 1 && S.P1 == 1) {
    off = (long long)r * S.seq_stride;
  } else {
    const int q = r / S.P0, p0 = r - q * S.P0;
    const int sq = q / S.P1, p1 = q - sq * S.P1;
    off = (long long)sq * S.seq_stride + (long long)(p1 * S.step1 - S.pad1) * S.line_stride +
          (long long)(p0 * S.step0 - S.pad0) * S.unit;
  }
  return (unsigned)(off * es);
}

// EP selects the epilogue compiled into an instance (the host picks it from the descriptor): one
// kernel holding all of them needs 256 VGPRs + scratch; each on its own stays near 130-160.
//   0 plain store (+ residual*gamma, leaky ReLU, fused PReLU)   1 PReLU backward (+ column sums)
//   2 row-mapped store (halo layout; + leaky ReLU, or leaky-ReLU backward of the layer below)
//   3 everything else (generic epilogue; the only one stream-K instances use)
// P3 (split-bf16, precision 1): both operands arrive PRE-SPLIT (f2g_split_bf16: every aligned group
// of four floats replaced by its four bf16 high parts and four bf16 remainders, same 16 bytes, same
// addressing), so the K loop stays free of VALU work: a staged 16-byte chunk goes to LDS as two
// 8-byte halves (row = [hi k0..31 | lo k0..31 | pad], the fp32 tile's 144-byte pitch), fragments are
// ds_read_b128 of eight consecutive k, and every product is lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_bf16 (24 MFMAs of 32 cycles per wave and slab instead of 64 of 64).
// PM: 0 exact fp32, 1 split-bf16 (three MFMAs per product), 2 plain bf16 = the high parts of the
// same images only (precision 2: one MFMA per product, the lo halves are neither staged nor read),
// 3 plain bf16 over TRUE bf16 tensors (f2g_to_bf16 images / bf16 producers: 2 bytes per element,
// operand strides in elements): the same 128-byte staged row now holds 64 k, so a slab carries
// twice the reduction for the same load, LDS and barrier work (16 MFMAs per wave and slab).
// WM: wave rows of the block = 2 (128 x 128 tile, 4 waves, two blocks per CU) or 4 (256 x 128, 8
// waves, one block per CU).  The bf16 instances are bound by L2 -> CU operand delivery (PMC: 13 TB/s
// of L2 reads on the 1024-channel MPD layer at 128 x 128 = 32 FLOP per byte): the taller tile
// moves a quarter less per FLOP with the same waves per SIMD.
// TAP (split-bf16, 256 x 128 only): A is a stride-1 (taps, 1) conv window over a halo layout
// (win1d, step 1, pad 0).  Tap-major K order makes the plain kernel fetch every activation row once
// per tap; here the rows a tile needs -- its output rows' padded positions plus taps - 1, including
// the halo rows of the sequence ends inside the tile -- are staged ONCE per 32-channel slab and
// the taps walk over them in LDS (a lane's fragment row = its output row's staged row + tap):
// 260-320 staged rows instead of 5 x 256 per channel slab, about half the L2 -> CU traffic of
// the kernel that is bound by exactly that.
// SK: 0 one tile per block, 1 stream-K with atomic seams (linear epilogues, zeroed output).  (A third mode,
// stream-K with a seam FIX-UP through a per-stream workspace, was measured flat under the launch lanes in
// round 4 and removed in round 6: DESIGN.md section 8, "measured and dropped".)
template <int SK, int EP, int PM, int WM = 2, bool TAP = false>
__global__ __launch_bounds__(WM * 128, 4 / WM)
void gemm_lean_kernel(const f2g_gemm_desc d, int M, int N, int K, int kchunk, int upb) {
  constexpr bool P3 = PM == 1 || PM == 2, HI = PM == 2, BF = PM == 3;
  constexpr int BKE = BF ? 64 : BK;   // elements per slab
  constexpr int ES = BF ? 2 : 4;      // bytes per element
  constexpr int BM = 64 * WM, BN = 128, TSZ = BM * LDR, TSB = BN * LDR;
  constexpr int RS = 16 * WM;         // rows staged per pass of the block (32 or 64)
  constexpr int QB = BN / RS;         // passes over the B tile (4 or 2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  // staged row of this thread (of 32, repeated four times 32 rows apart).  The split-bf16 tile is
  // filled with 8-byte stores, served 16 lanes = two rows at a time over 32 banks: rows r and r + 4
  // (4 x 36 dwords = 16 mod 32) share no bank, rows r and r + 1 would share 12 of 16.
  const int ch = tid & 7;
  const int t8 = tid & 255;
  const int rr = (P3 ? (((t8 >> 4) & 3) + 8 * (t8 >> 6) + 4 * ((t8 >> 3) & 1)) : (t8 >> 3)) + 32 * (tid >> 8);
  __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, 0x80000000u, 0x00020000);
  // B is a plain [n][k] matrix: the resource ends with its last row, so the rows of a partial
  // last tile (n >= N) are out of range = zeros, and ONE per-thread offset serves all four staged
  // rows (their distance, 32 rows, is uniform and rides in the scalar offset)
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.B.base, 0, (unsigned)((long long)N * d.B.seq_stride * ES), 0x00020000);
  const int qstepB = (int)(RS * d.B.seq_stride * ES);
  // scalar K walk of A: segments of `seglen` columns, `line_stride` floats apart
  const int seglen = d.A.seglen < d.A.cols ? d.A.seglen : d.A.cols;
  const int spseg = seglen / BKE;                                 // slabs per segment
  const int segjump = (int)((d.A.line_stride - seglen) * ES);     // bytes skipped at a segment end
  // LDS: [A buffer 0 | A buffer 1 | B buffer 0 | B buffer 1]; a slab offset `bo` (0 / TSZ) selects the
  // A buffer, the B buffer of the same index lies bo / TSZ * TSB further on
  float* wA = smem + rr * LDR + ch * (P3 ? 2 : 4);
  float* wB = smem + 2 * TSZ + rr * LDR + ch * (P3 ? 2 : 4);
  const float* rA = smem + (wm * 64 + li) * LDR + h * ((P3 || BF) ? 4 : 16);
  const float* rB = smem + 2 * TSZ + (wn * 64 + li) * LDR + h * ((P3 || BF) ? 4 : 16);
  auto bofB = [](int bo) { return WM == 2 ? bo : (bo ? TSB : 0); };

  // ---- work of this block.  Classic: one tile (blockIdx.x/y), K chunk blockIdx.z.  Stream-K
  // (upb > 0): the (tile, slab) units of the whole problem are numbered tile-major and every
  // block takes `upb` consecutive ones -- a tile count just above a multiple of the 512 resident
  // blocks no longer costs a nearly empty extra round; tiles cut between blocks are accumulated
  // atomically onto a zeroed output, bias / residual entering with the part that holds slab 0.
  const int nt_all = K / BKE;
  const int tiles_n = (N + BN - 1) / BN;
  int u = 0, u_end = 0;
  if (SK == 1) {
    const int G = gridDim.x;
    const int q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;  // XCD-contiguous
    const int total = ((M + BM - 1) / BM) * tiles_n * nt_all;
    u = b * upb;
    u_end = u + upb < total ? u + upb : total;
    if (u >= u_end) return;
  }
  bool more = true;
  while (more) {
    int m0, n0, s0, nt;
    bool first, partial;
    if (!SK) {
      tile_of_block(BM, BN, m0, n0);
      const int kbeg = blockIdx.z * kchunk;
      int kend = kbeg + kchunk;
      if (kend > K) kend = K;
      s0 = kbeg / BKE;
      nt = (kend - kbeg) / BKE;
      first = blockIdx.z == 0;
      partial = false;
      more = false;
    } else {
      const int tl = u / nt_all;
      s0 = u - tl * nt_all;
      nt = nt_all - s0 < u_end - u ? nt_all - s0 : u_end - u;
      first = s0 == 0;
      partial = nt != nt_all;
      const int tm = tl / tiles_n;
      m0 = tm * BM;
      n0 = (tl - tm * tiles_n) * BN;
      u += nt;
      more = u < u_end;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + (wn * 2 + ni) * 32 + li;
      const float b = (d.E.bias && first && col < N) ? d.E.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = b;
    }
    if constexpr (TAP) {
      constexpr int RAMAX = 320;                 // staged activation rows per channel slab (host-checked)
      constexpr int TSA = RAMAX * LDR;
      float* sA = smem;                          // [2][RAMAX][LDR]
      float* sB = smem + 2 * TSA;                // [2][128][LDR]
      const int Cin = d.A.unit, taps = d.A.cols / Cin, P0 = d.A.P0;
      const int Hp = (int)(d.A.seq_stride / Cin);
      auto qof = [&](int m) { const int sq = m / P0; return sq * Hp + (m - sq * P0); };
      const int mlast = (m0 + BM < M ? m0 + BM : M) - 1;
      const int qb = qof(m0);
      int rowA[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        int r = m0 + wm * 64 + mi * 32 + li;
        r = r > mlast ? mlast : r;
        rowA[mi] = (qof(r) - qb) * LDR + h * 4;
      }
      const long long a_bytes = (long long)(d.A.rows / P0) * d.A.seq_stride * 4;
      __amdgpu_buffer_rsrc_t rat =
          __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, (unsigned)a_bytes, 0x00020000);
      unsigned voA[5];
      int wofA[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int ci = tid + 512 * u, j = ci >> 3, c8 = ci & 7;
        voA[u] = j < RAMAX ? (unsigned)((long long)j * Cin * 4 + c8 * 16) : 0x80000000u;
        wofA[u] = (j < RAMAX ? j : 0) * LDR + c8 * 2;
      }
      const long long sa0 = (long long)qb * Cin * 4;
      const unsigned offBt = (unsigned)((long long)(n0 + rr) * d.B.seq_stride * 4) + ch * 16;
      float* wBt = sB + rr * LDR + ch * 2;
      const float* rBt = sB + (wn * 64 + li) * LDR + h * 4;
      const int ncs = Cin / BK, nit = ncs * taps;
      auto gloadA = [&](int cs, u32x4 (&ax)[5]) {
        const int so = (int)(sa0 + (long long)cs * BK * 4);
#pragma unroll
        for (int u = 0; u < 5; ++u) ax[u] = __builtin_amdgcn_raw_buffer_load_b128(rat, voA[u], so, 0);
      };
      auto lstoreA = [&](int buf, const u32x4 (&ax)[5]) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          if (u < 4 || tid + 512 * 4 < RAMAX * 8) {
            float* p = sA + buf * TSA + wofA[u];
            *reinterpret_cast<u32x2*>(p) = u32x2{ax[u].x, ax[u].y};
            *reinterpret_cast<u32x2*>(p + 16) = u32x2{ax[u].z, ax[u].w};
          }
        }
      };
      auto gloadB = [&](int cs, int tap, u32x4 (&lb)[2]) {
        const int so = (tap * Cin + cs * BK) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offBt, so + q * qstepB, 0);
      };
      auto lstoreB = [&](int buf, const u32x4 (&lb)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float* p = wBt + buf * TSB + q * RS * LDR;
          *reinterpret_cast<u32x2*>(p) = u32x2{lb[q].x, lb[q].y};
          *reinterpret_cast<u32x2*>(p + 16) = u32x2{lb[q].z, lb[q].w};
        }
      };
      bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
      auto fragsT = [&](int cs, int tap, int bbuf, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
        const float* pa = sA + (cs & 1) * TSA + tap * LDR + ks * 8;
        const float* pb = rBt + bbuf * TSB + ks * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const bf16x8*>(pa + rowA[i]);
          fa[2 + i] = *reinterpret_cast<const bf16x8*>(pa + rowA[i] + 16);
          fb[i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * LDR);
          fb[2 + i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * LDR + 16);
        }
      };
      auto mfma12 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const bf16x8 av = term == 0 ? fa[2 + mi] : fa[mi];
              const bf16x8 bv = term == 1 ? fb[2 + ni] : fb[ni];
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mi][ni], 0, 0, 0);
            }
      };
      // (cs, tap) of iteration it, it + 1 and it + 2, walked with scalar counters
      int cs0 = 0, tp0 = 0, cs1 = 0, tp1 = 1, cs2 = 0, tp2 = 2;
      auto wrap = [&](int& c, int& t) { if (t >= taps) { t -= taps; ++c; } };
      wrap(cs1, tp1);
      wrap(cs2, tp2);
      wrap(cs2, tp2);
      u32x4 xb[2], yb[2], ax[5];
      gloadA(0, ax);
      gloadB(0, 0, xb);
      lstoreA(0, ax);
      lstoreB(0, xb);
      gloadB(cs1, tp1, xb);     // (nit >= 2: taps >= 2)
      if (ncs > 1) gloadA(1, ax);
      __syncthreads();
      fragsT(0, 0, 0, 0, fa0, fb0);
      auto stepT = [&](int it, int cur, int nxt, const u32x4 (&wb)[2], u32x4 (&lb)[2]) {
        fragsT(cs0, tp0, cur, 1, fa1, fb1);
        const bool more2 = it + 2 < nit;
        gloadB(more2 ? cs2 : 0, more2 ? tp2 : 0, lb);
        // the next channel slab's rows: requested at tap 0 (the prologue did it for slab 1), stored
        // at tap 2 into the buffer slab cs - 1 has left two barriers ago
        if (tp0 == 0 && cs0 > 0 && cs0 + 1 < ncs) gloadA(cs0 + 1, ax);
        mfma12(fa0, fb0);
        lstoreB(nxt, wb);
        if (tp0 == (taps > 2 ? 2 : taps - 1) && cs0 + 1 < ncs) lstoreA((cs0 + 1) & 1, ax);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        fragsT(cs1, tp1, nxt, 0, fa0, fb0);
        mfma12(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        cs0 = cs1; tp0 = tp1; cs1 = cs2; tp1 = tp2;
        ++tp2;
        wrap(cs2, tp2);
      };
      int it = 0;
      for (; it + 1 < nit; it += 2) {
        stepT(it, 0, 1, xb, yb);
        stepT(it + 1, 1, 0, yb, xb);
      }
      if (it < nit) stepT(it, 0, 1, xb, yb);
    } else {
    unsigned offA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      offA[q] = lean_row_offset(d.A, m0 + rr + RS * q, ES);
      if (offA[q] != 0x80000000u) offA[q] += ch * 16;
    }
    const unsigned offB = (unsigned)((long long)(n0 + rr) * d.B.seq_stride * ES) + ch * 16;
    int left = spseg - (s0 % spseg);
    const int ka0 = (int)(((long long)(s0 / spseg) * d.A.line_stride + (long long)(s0 % spseg) * BKE) * ES);
    const int kb0 = s0 * BKE * ES;
    int ka = ka0, kb = kb0;

    auto gload = [&](int soa, int sob, u32x4 (&la)[4], u32x4 (&lb)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        la[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, offA[q], soa, 0);
        if (q < QB) lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offB, sob + q * qstepB, 0);
      }
    };
    auto lstore = [&](int bufoff, const u32x4 (&la)[4], const u32x4 (&lb)[4]) {
      const int bb = bofB(bufoff);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (P3) {
          *reinterpret_cast<u32x2*>(wA + bufoff + q * RS * LDR) = u32x2{la[q].x, la[q].y};
          if constexpr (!HI)
            *reinterpret_cast<u32x2*>(wA + bufoff + q * RS * LDR + 16) = u32x2{la[q].z, la[q].w};
          if (q < QB) {
            *reinterpret_cast<u32x2*>(wB + bb + q * RS * LDR) = u32x2{lb[q].x, lb[q].y};
            if constexpr (!HI)
              *reinterpret_cast<u32x2*>(wB + bb + q * RS * LDR + 16) = u32x2{lb[q].z, lb[q].w};
          }
        } else {
          *reinterpret_cast<u32x4*>(wA + bufoff + q * RS * LDR) = la[q];
          if (q < QB) *reinterpret_cast<u32x4*>(wB + bb + q * RS * LDR) = lb[q];
        }
      }
    };
    auto mfma_slab = [&](int bufoff) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        float4 a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          a[mi] = *reinterpret_cast<const float4*>(rA + bufoff + mi * 32 * LDR + s4 * 4);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          b[ni] = *reinterpret_cast<const float4*>(rB + bofB(bufoff) + ni * 32 * LDR + s4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const float av = q == 0 ? a[mi].x : q == 1 ? a[mi].y : q == 2 ? a[mi].z : a[mi].w;
              const float bv = q == 0 ? b[ni].x : q == 1 ? b[ni].y : q == 2 ? b[ni].z : b[ni].w;
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
            }
      }
    };
    // (SALU) K offsets of the slab after the current one
    auto advance = [&]() {
      ka += BK * 4;
      kb += BK * 4;
      if (--left == 0) {
        left = spseg;
        ka += segjump;
      }
    };
    if constexpr (P3 || BF) {
      // The bf16 MFMA phase of a slab is 5x shorter than the fp32 one (24 x 32 cycles), too short to
      // hide a load, an LDS fill and a barrier behind it one after the other.  So the phases overlap
      // inside a wave: slab t's MFMAs are interleaved with the LDS stores of slab t+1 (in registers
      // since the previous iteration) while the loads of slab t+2 fly -- two register stages.
      u32x4 xa[4], xb[4], ya[4], yb[4];
      if (nt > 0) {
        gload(ka, kb, xa, xb);
        lstore(0, xa, xb);
      }
      advance();
      gload(nt > 1 ? ka : ka0, nt > 1 ? kb : kb0, xa, xb);
      __syncthreads();
      // fragments: f0 = first k step (8 consecutive k per lane half), f1 = second; [0..1] = hi of
      // the two sub-tiles, [2..3] = lo.  f0 of the NEXT slab is read right after the barrier, under
      // the MFMAs of f1; f1 is read at the top of an iteration, under the MFMAs of f0.
      bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
      auto frags = [&](int off, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
        if constexpr (BF) {   // half `ks` of the slab = k steps 2ks, 2ks+1: [0..1] and [2..3]
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              fa[2 * j + i] = *reinterpret_cast<const bf16x8*>(rA + off + i * 32 * LDR + (2 * ks + j) * 8);
              fb[2 * j + i] = *reinterpret_cast<const bf16x8*>(rB + bofB(off) + i * 32 * LDR + (2 * ks + j) * 8);
            }
          return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const bf16x8*>(rA + off + i * 32 * LDR + ks * 8);
          fb[i] = *reinterpret_cast<const bf16x8*>(rB + bofB(off) + i * 32 * LDR + ks * 8);
          if constexpr (!HI) {
            fa[2 + i] = *reinterpret_cast<const bf16x8*>(rA + off + i * 32 * LDR + ks * 8 + 16);
            fb[2 + i] = *reinterpret_cast<const bf16x8*>(rB + bofB(off) + i * 32 * LDR + ks * 8 + 16);
          }
        }
      };
      f32x16 shadow[2][2];   // (lab, F2G_LABVAR & 16: second accumulator set -> twice the dependent distance)
      if (F2G_LABVAR & 16) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) shadow[mi][ni][e] = 0.f;
      }
      auto mfma12 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
        if constexpr (BF) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * j + mi], fb[2 * j + ni],
                                                                     acc[mi][ni], 0, 0, 0);
          return;
        }
#pragma unroll
        for (int term = HI ? 2 : 0; term < 3; ++term)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const bf16x8 av = term == 0 ? fa[2 + mi] : fa[mi];
              const bf16x8 bv = term == 1 ? fb[2 + ni] : fb[ni];
              if ((F2G_LABVAR & 16) && term == 1)
                shadow[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, shadow[mi][ni], 0, 0, 0);
              else
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mi][ni], 0, 0, 0);
            }
      };
      frags(0, 0, fa0, fb0);
      auto step3 = [&](int t, int curoff, int nxtoff, const u32x4 (&wa)[4], const u32x4 (&wb)[4],
                       u32x4 (&la)[4], u32x4 (&lb)[4]) {
        if (!(F2G_LABVAR & 8)) frags(curoff, 1, fa1, fb1);
        advance();
        const bool again = t + 2 < nt;   // past the end: re-read the first slab (never used)
        if (!(F2G_LABVAR & 1)) gload(again ? ka : ka0, again ? kb : kb0, la, lb);
        mfma12(fa0, fb0);
        if (!(F2G_LABVAR & 2)) lstore(nxtoff, wa, wb);
        // issue order: fragments, the loads of the slab after next, one LDS store behind each of
        // the first MFMAs
        constexpr int NM = HI ? 4 : (BF ? 8 : 12);        // MFMAs per half slab
        constexpr int NW = 4 + QB;                        // LDS store instructions per slab
        constexpr int WPM = (NW + NM - 1) / NM;
        __builtin_amdgcn_sched_group_barrier(0x100, HI ? 4 : 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 4 + QB, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i * WPM < NW) __builtin_amdgcn_sched_group_barrier(0x200, WPM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(F2G_LABVAR & 4)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (!(F2G_LABVAR & 8)) frags(nxtoff, 0, fa0, fb0);
        mfma12(fa1, fb1);
        __builtin_amdgcn_sched_group_barrier(0x100, HI ? 4 : 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, HI ? 4 : (BF ? 8 : 12), 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      int t = 0;
      for (; t + 1 < nt; t += 2) {
        step3(t, 0, TSZ, xa, xb, ya, yb);
        step3(t + 1, TSZ, 0, ya, yb, xa, xb);
      }
      if (t < nt) step3(t, 0, TSZ, xa, xb, ya, yb);
      if (F2G_LABVAR & 16) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] += shadow[mi][ni][e];
      }
    } else {
    if (nt > 0) {
      u32x4 la[4], lb[4];
      gload(ka, kb, la, lb);
      lstore(0, la, lb);
    }
    __syncthreads();
    auto step = [&](int t, int curoff, int nxtoff) {
      u32x4 la[4], lb[4];
      advance();
      const bool again = t + 1 < nt;   // the last iteration re-reads the first slab (never used)
      gload(again ? ka : ka0, again ? kb : kb0, la, lb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_slab(curoff);
      __builtin_amdgcn_sched_barrier(0);
      lstore(nxtoff, la, lb);
      __syncthreads();
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      step(t, 0, TSZ);
      step(t + 1, TSZ, 0);
    }
    if (t < nt) {
      step(t, 0, TSZ);
      // an odd slab count leaves the (unused) restaged slab in buffer 1; the next segment starts in
      // buffer 0, which every wave has finished reading (barrier above)
    }

    }

    }
    const int li_e = li, h_e = h;
    const f2g_epilogue& E = d.E;
    const bool simple = !partial && !E.aux && !E.colsum && !E.colsum_alpha && E.P0o == 0 &&
                        !E.atomic && !E.accumulate && E.scale == 0.f && !E.mask_src;
    (void)simple;
    if constexpr (EP == 0) {
      // plain store (+ leaky ReLU / PReLU): uniform row bases, per-lane constant offset
      const float sl = E.lrelu_slope;
      const bool pre = E.prelu_slope != nullptr, two = pre && E.prelu_out != nullptr;
      const bool cbf = E.c_bf16 != 0;
      const unsigned coff = (unsigned)(((long long)(4 * h_e) * E.ldc + li_e) * 4);
      const unsigned poff = (unsigned)(((long long)(4 * h_e) * E.ld_prelu_out + li_e) * 4);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col0 = n0 + (wn * 2 + ni) * 32;
          const int row0 = m0 + (wm * 2 + mi) * 32;
          const float ps = (pre && col0 + li_e < N) ? E.prelu_slope[col0 + li_e] : 0.f;
          const bool hasres = E.res != nullptr;
          const float gam = (hasres && col0 + li_e < N) ? (E.gamma ? E.gamma[col0 + li_e] : 1.f) : 0.f;
          if (row0 + 32 <= M && col0 + 32 <= N) {
            char* cb = reinterpret_cast<char*>(E.C + (long long)row0 * E.ldc + col0);
            __bf16* cb16 = reinterpret_cast<__bf16*>(E.C) + (long long)row0 * E.ldc + col0;
            char* pb = reinterpret_cast<char*>(E.prelu_out + (long long)row0 * E.ld_prelu_out + col0);
            const char* rb = reinterpret_cast<const char*>(E.res + (long long)row0 * E.ldres + col0);
            const unsigned roff = (unsigned)(((long long)(4 * h_e) * E.ldres + li_e) * 4);
            float rv[16];
            if (hasres) {   // all 16 residual values requested before any is consumed
#pragma unroll
              for (int e = 0; e < 16; ++e)
                rv[e] = *reinterpret_cast<const float*>(rb + (long long)((e & 3) + 8 * (e >> 2)) * E.ldres * 4 + roff);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float v = acc[mi][ni][e];
              if (hasres) v += gam * rv[e];
              if (sl != 0.f) v = fmaxf(v, 0.f) + sl * fminf(v, 0.f);
              const long long ro = (e & 3) + 8 * (e >> 2);
              if (pre) {
                const float pv = fmaxf(v, 0.f) + ps * fminf(v, 0.f);
                if (two) *reinterpret_cast<float*>(pb + ro * E.ld_prelu_out * 4 + poff) = pv;
                else v = pv;
              }
              if (cbf)   // C is a bf16 tensor (ldc in elements): the next GEMM's operand as it is
                cb16[(ro + 4 * h_e) * E.ldc + li_e] = (__bf16)v;
              else
                *reinterpret_cast<float*>(cb + ro * E.ldc * 4 + coff) = v;
            }
            __builtin_amdgcn_sched_barrier(0);   // one sub-tile's loads / stores at a time (registers)
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h_e;
              float v = acc[mi][ni][e];
              if (hasres && row < M && col0 + li_e < N) v += gam * E.res[(long long)row * E.ldres + col0 + li_e];
              if (sl != 0.f) v = fmaxf(v, 0.f) + sl * fminf(v, 0.f);
              if (row < M && col0 + li_e < N) {
                if (pre) {
                  const float pv = fmaxf(v, 0.f) + ps * fminf(v, 0.f);
                  if (two) E.prelu_out[(long long)row * E.ld_prelu_out + col0 + li_e] = pv;
                  else v = pv;
                }
                if (cbf) reinterpret_cast<__bf16*>(E.C)[(long long)row * E.ldc + col0 + li_e] = (__bf16)v;
                else E.C[(long long)row * E.ldc + col0 + li_e] = v;
              }
            }
          }
        }
    } else if constexpr (EP == 1) {
      // PReLU backward fused into the data gradient (modules.py:444,488 backward):
      //   v = acc * (a > 0 ? 1 : alpha[n]);  d alpha[n] += sum_r acc * min(a, 0);  d bias[n] += sum_r v
      // plain store (C may alias aux: each element is read before it is written by the same lane)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + (wn * 2 + ni) * 32 + li_e;
        const bool cok = col < N;
        const float aln = cok ? E.alpha_n[col] : 0.f;
        float cs = 0.f, csa = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row0 = m0 + (wm * 2 + mi) * 32 + 4 * h_e;
          const bool full = cok && row0 - 4 * h_e + 32 <= M;
          const float* ab = E.aux + (long long)row0 * E.ldaux + col;
          float* cb = E.C + (long long)row0 * E.ldc + col;
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {           // four rows (r, r+1, r+2, r+3) at a time
            float av[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = 8 * e4 + k;
              av[k] = (full || (cok && row0 + r < M)) ? ab[(long long)r * E.ldaux] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = 8 * e4 + k;
              const float a0 = acc[mi][ni][e4 * 4 + k];
              csa += a0 * fminf(av[k], 0.f);
              const float v = a0 * (av[k] > 0.f ? 1.f : aln);
              if (full || (cok && row0 + r < M)) {
                cs += v;
                cb[(long long)r * E.ldc] = v;
              }
            }
          }
        }
        if (E.colsum || E.colsum_alpha) {
          cs += __shfl_xor(cs, 32);
          csa += __shfl_xor(csa, 32);
          if (cok && h_e == 0) {
            if (E.colsum) atomicAdd(E.colsum + col, cs);
            if (E.colsum_alpha) atomicAdd(E.colsum_alpha + col, csa);
          }
        }
      }
    } else if constexpr (EP == 2) {
      // row-mapped store: the halo layout of the MPD maps and the stride residues of their data
      // gradients.  One division per 32-row sub-tile instead of one per element (the 32 rows of a
      // sub-tile wrap the sequence length (>= 32) at most once).  Options: leaky ReLU (forward), or
      // the leaky-ReLU backward of the layer below (+ feature-matching term) with the column sums
      // of the result = that layer's bias gradient.
      const float sl = E.lrelu_slope;
      const bool msk = E.mask_src != nullptr, fm = E.fm_ref != nullptr;
      const float fmw = fm ? E.fm_w * (E.fm_wdev ? E.fm_wdev[0] : 1.f) : 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + (wn * 2 + ni) * 32 + li_e;
        const bool cok = col < N;
        float cs = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row0 = m0 + (wm * 2 + mi) * 32;
          const int q0 = row0 / E.P0o;                         // uniform
          const int p0 = row0 - q0 * E.P0o + 4 * h_e;            // position of this lane's first row
          const long long cbase = E.off_o + col;
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            long long off[4];
            bool ok[4];
            float yv[4], rv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int re = 8 * e4 + k;
              int p = p0 + re, q = q0;
              if (p >= E.P0o) { p -= E.P0o; ++q; }
              ok[k] = cok && row0 + re + 4 * h_e < M;
              off[k] = cbase + (long long)q * E.seq_stride_o + (long long)p * E.row_stride_o;
              if (msk) yv[k] = ok[k] ? E.mask_src[off[k]] : 0.f;
              if (fm) rv[k] = ok[k] ? E.fm_ref[off[k]] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float v = acc[mi][ni][e4 * 4 + k];
              if (sl != 0.f) v = fmaxf(v, 0.f) + sl * fminf(v, 0.f);
              if (msk) {
                if (fm) {
                  const float dl = yv[k] - rv[k];
                  v += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
                }
                v *= yv[k] > 0.f ? 1.f : E.mask_slope;
              }
              if (ok[k]) {
                cs += v;
                E.C[off[k]] = v;
              }
            }
            asm volatile("" ::: "memory");   // four rows at a time: keeps the next group's loads
            __builtin_amdgcn_sched_barrier(0);   // from being hoisted (register pressure)
          }
        }
        if (E.colsum) {
          cs += __shfl_xor(cs, 32);
          if (cok && h_e == 0) atomicAdd(E.colsum + col, cs);
        }
      }
    } else {
      f2g_epilogue E2 = E;
      E2.bias = nullptr;   // already in the accumulators
      if (partial) { E2.atomic = 1; E2.accumulate = 0; }
      gemm_epilogue<2, 2>(E2, acc, M, N, m0, n0, wm, wn, li_e, h_e, first);
    }
  }
}

// Can `S` (the A operand of a form-0 GEMM) be read by the lean kernel: aligned, no on-load
// transform, reduction in whole 32-column slabs per segment, and every window inside its source?
inline bool lean_a_ok(const f2g_operand& S) {
  if (S.reflect || S.alpha || S.lrelu_src || S.rows <= 0 || S.cols < BK) return false;
  if (!al16(S.base) || (S.seq_stride & 3) || (S.line_stride & 3)) return false;
  const long long eu0 = (long long)S.step0 * S.unit, ep0 = (long long)S.pad0 * S.unit;
  if ((eu0 & 3) || (ep0 & 3)) return false;
  const int seglen = S.seglen < S.cols ? S.seglen : S.cols;
  if (seglen % BK || S.cols % seglen) return false;
  const int nseg = S.cols / seglen;
  if (S.P0 < 1 || S.P1 < 1) return false;
  if (S.rows % (S.P0 * S.P1)) return false;
  // line range
  if (S.pad1 > 0 || (long long)(S.P1 - 1) * S.step1 - S.pad1 + nseg - 1 >= S.L1) return false;
  // element range inside a line
  if (S.pad0 > 0 || ((long long)(S.P0 - 1) * S.step0 - S.pad0) * S.unit + seglen > S.L0u) return false;
  if (nseg > 1 && S.line_stride < seglen) return false;
  // byte offsets must stay below 2 GiB
  const long long nseq = S.rows / (S.P0 * S.P1);
  const long long last = (nseq - 1) * S.seq_stride + (long long)(S.L1 - 1) * S.line_stride + S.L0u;
  return last * 4 < 0x7ff00000ll;
}

// A as a stride-1 conv window whose rows a 256-row tile can stage once per channel slab (TAP mode)
inline bool lean_tap_ok(const f2g_operand& A) {
  if (A.P1 != 1 || A.step0 != 1 || A.pad0 != 0 || A.unit < BK || A.unit % BK) return false;
  if (A.cols % A.unit || A.cols / A.unit < 2 || A.seglen < A.cols || A.seq_stride % A.unit) return false;
  const int taps = A.cols / A.unit, Hp = (int)(A.seq_stride / A.unit);
  if (A.P0 < 8 || Hp < A.P0 + taps - 1) return false;
  return 256 + taps - 1 + (Hp - A.P0) * (256 / A.P0 + 1) <= 320;
}

// the same operand as a TRUE bf16 tensor (split = 2): 16-byte chunks hold 8 elements, slabs 64
inline bool lean_bf16_ok(const f2g_operand& A, const f2g_operand& B) {
  const long long eu0 = (long long)A.step0 * A.unit, ep0 = (long long)A.pad0 * A.unit;
  if ((A.seq_stride & 7) || (A.line_stride & 7) || (eu0 & 7) || (ep0 & 7) || (B.seq_stride & 7)) return false;
  const int seglen = A.seglen < A.cols ? A.seglen : A.cols;
  return seglen % 64 == 0 && B.cols % 64 == 0;
}

inline bool lean_b_ok(const f2g_operand& S) {
  return host_plain(S) && !S.alpha && al16(S.base) && (S.seq_stride & 3) == 0 && S.cols % BK == 0 &&
         (long long)S.rows * S.seq_stride * 4 < 0x7ff00000ll;
}

int launch_lean(const f2g_gemm_desc& d, int M, int N, int K, int split, int upb, hipStream_t st) {
  // operand images: split = 1 -> split-bf16 pairs (precision 1: all three products, 2: high parts),
  // split = 2 -> true bf16 tensors (precision 2 only)
  const int pm = d.A.split == 2 ? 3 : (d.precision == 1 ? 1 : (d.precision == 2 ? 2 : 0));
  const int bk = pm == 3 ? 64 : BK;
  int kchunk = ((K + split - 1) / split + bk - 1) / bk * bk;
  int zs = (K + kchunk - 1) / kchunk;
  // 256 x 128 tiles (8 waves) for the bf16 instances when the taller grid still fills the chip
  // and the reduction is long enough to amortise the larger prologue / epilogue (measured: +11 % on
  // the 1024-channel MPD layers, -7 % at K = 384 / 512)
  // (option lean_tall: 0 never, 1 when K >= 640 and there are >= 400 tall tiles, 2 whenever possible)
  const int tall_mode = f2g_opt(F2G_OPT_LEAN_TALL);
  const long long tall_tiles = (long long)((M + 255) / 256) * ((N + 127) / 128);
  const bool tall = (pm == 1 || pm == 3) && upb == 0 && zs == 1 && tall_mode > 0 &&
                    (tall_mode > 1 || (tall_tiles >= 400 && K >= 640));
  const int bm = tall ? 256 : 128;
  // tap-reusing variant for stride-1 conv windows (option lean_tap = 0 turns it off)
  const bool tap_on = f2g_opt(F2G_OPT_LEAN_TAP) != 0;
  const bool tap = tall && pm == 1 && tap_on && lean_tap_ok(d.A);
  const size_t smem = tap ? (size_t)(2 * 320 + 2 * 128) * LDR * sizeof(float)
                          : (size_t)(2 * bm + 2 * 128) * LDR * sizeof(float);
  dim3 grid((M + bm - 1) / bm, (N + 127) / 128, zs);
  if (grid.x == 0 || grid.y == 0) return F2G_OK;
  if (upb > 0) {
    const long long total = (long long)grid.x * grid.y * (K / bk);
    grid = dim3((unsigned)((total + upb - 1) / upb), 1, 1);
  }
  // epilogue instance (see gemm_lean_kernel)
  const f2g_epilogue& E = d.E;
  int ep = 3;
  if (upb == 0) {
    const bool plainish = !E.aux && !E.colsum_alpha && !E.atomic && !E.accumulate && E.scale == 0.f;
    if (plainish && !E.colsum && E.P0o == 0 && !E.mask_src) ep = 0;
    else if (E.aux && !E.res && E.P0o == 0 && !E.atomic && !E.accumulate && E.scale == 0.f &&
             !E.prelu_slope && E.lrelu_slope == 0.f && !E.mask_src) ep = 1;
    else if (plainish && !E.res && !E.prelu_slope && E.P0o >= 32) ep = 2;
  }
  if (E.c_bf16 && ep != 0) {
    f2g_set_error("f2g_gemm: a bf16 output needs the plain-store epilogue of the lean kernel");
    return F2G_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    const void* ks[24] = {reinterpret_cast<const void*>(gemm_lean_kernel<2, 0, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<2, 1, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<2, 2, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<2, 3, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<true, 3, 0>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 1>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 1>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 1>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 1>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<true, 3, 1>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 2>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 2>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 2>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 2>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<true, 3, 2>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 3>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 3>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 3>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 3>),
                          reinterpret_cast<const void*>(gemm_lean_kernel<true, 3, 3>)};
    for (const void* k : ks)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * LDR * 4);
    const void* kt[8] = {reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 1, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 1, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 1, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 1, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 0, 3, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 1, 3, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 3, 4>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 3, 4>)};
    for (const void* k : kt)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 256 + 2 * 128) * LDR * 4);
    const void* kp[2] = {reinterpret_cast<const void*>(gemm_lean_kernel<false, 2, 1, 4, true>),
                         reinterpret_cast<const void*>(gemm_lean_kernel<false, 3, 1, 4, true>)};
    for (const void* k : kp)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 320 + 2 * 128) * LDR * 4);
    attr_done = true;
  }
  g_last_path = upb > 0 ? 2 : 1;
#define F2G_LEAN(SKV, EPV)                                                                        \
  do {                                                                                            \
    if (pm == 1)                                                                                  \
      hipLaunchKernelGGL((gemm_lean_kernel<SKV, EPV, 1>), grid, dim3(256), smem, st, d, M, N, K,  \
                         kchunk, upb);                                                         \
    else if (pm == 2)                                                                             \
      hipLaunchKernelGGL((gemm_lean_kernel<SKV, EPV, 2>), grid, dim3(256), smem, st, d, M, N, K,  \
                         kchunk, upb);                                                         \
    else if (pm == 3)                                                                             \
      hipLaunchKernelGGL((gemm_lean_kernel<SKV, EPV, 3>), grid, dim3(256), smem, st, d, M, N, K,  \
                         kchunk, upb);                                                         \
    else                                                                                          \
      hipLaunchKernelGGL((gemm_lean_kernel<SKV, EPV, 0>), grid, dim3(256), smem, st, d, M, N, K,  \
                         kchunk, upb);                                                         \
  } while (0)
#define F2G_LEAN_T(EPV)                                                                           \
  do {                                                                                            \
    if (pm == 1)                                                                                  \
      hipLaunchKernelGGL((gemm_lean_kernel<false, EPV, 1, 4>), grid, dim3(512), smem, st, d, M,   \
                         N, K, kchunk, upb);                                                   \
    else                                                                                          \
      hipLaunchKernelGGL((gemm_lean_kernel<false, EPV, 3, 4>), grid, dim3(512), smem, st, d, M,   \
                         N, K, kchunk, upb);                                                   \
  } while (0)
  if (tap && (ep == 2 || ep == 3)) {
    if (ep == 2)
      hipLaunchKernelGGL((gemm_lean_kernel<false, 2, 1, 4, true>), grid, dim3(512), smem, st, d, M, N, K,
                         kchunk, upb);
    else
      hipLaunchKernelGGL((gemm_lean_kernel<false, 3, 1, 4, true>), grid, dim3(512), smem, st, d, M, N, K,
                         kchunk, upb);
  } else if (tall) {
    if (ep == 0) F2G_LEAN_T(0);
    else if (ep == 1) F2G_LEAN_T(1);
    else if (ep == 2) F2G_LEAN_T(2);
    else F2G_LEAN_T(3);
  } else
  if (upb > 0) F2G_LEAN(true, 3);
  else if (ep == 0) F2G_LEAN(false, 0);
  else if (ep == 1) F2G_LEAN(false, 1);
  else if (ep == 2) F2G_LEAN(false, 2);
  else F2G_LEAN(false, 3);
#undef F2G_LEAN
#undef F2G_LEAN_T
  return f2g_check_launch();
}

// ---- lean weight-gradient kernel, split-bf16 --------------------------------------------------
// C[m,n] (+)= sum_r A[r,m] * B[r,n]: both operands are K-MAJOR (the reduction walks rows, memory is
// contiguous along m / n), while a bf16 MFMA wants 8 consecutive k per lane.  The transposition is
// done by the LDS itself: slabs of 32 rows are staged exactly as they lie in memory (pre-split
// images: a 16-byte chunk = four m as hi | lo, stored as two 8-byte halves into a hi and a lo plane
// of 32 x 128 bf16) and read back with ds_read_b64_tr_b16, which hands lane c of a 16-lane group
// column c of a 4 (k) x 16 (m) block whose 8-byte pieces the group's lanes point at -- two such reads
// = the 8 k of one operand.  Rows are 256 bytes; the 64-byte column blocks are XOR-swizzled with
// (row & 3) so that the four rows of a transposed block fall on different banks (a padded pitch
// would push the tile past two blocks per CU).  Everything else is the lean kernel's recipe:
// buffer loads with the K advance in a scalar register, rows past the end out of range = zeros, two
// register stages, LDS stores behind the MFMAs, fragments prefetched across the barrier.
// A: plain (R x M).  B: plain, or a 1-D window operand (rows = (sequence, position), P1 = 1, one
// segment) flagged `unbounded`: windows may reach past the ends of their sequence because the caller
// guarantees that those rows of A are zero (halo layout of the MPD maps) -- the per-row offsets are
// recomputed every slab (one magic-number division per staged row).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * 256));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <bool BWIN>
__global__ __launch_bounds__(256, 2)
void gemm_leanw3_kernel(const f2g_gemm_desc d, int M, int N, int K, int kchunk) {
  constexpr int PL = 32 * 256;            // bytes of one plane (32 rows x 128 bf16)
  constexpr int BUF = 4 * PL;             // [A hi | A lo | B hi | B lo]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* sm = reinterpret_cast<unsigned char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  const int kbeg = blockIdx.z * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt <= 0) return;

  // staging: thread = (row rid + 8q of the slab, 16-byte chunk c of the 128-wide tile row)
  const int rid = tid >> 5, c = tid & 31;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.A.base, 0, (unsigned)((long long)K * d.A.seq_stride * 4), 0x00020000);
  const long long b_bytes = BWIN ? (long long)(d.B.rows / d.B.P0) * d.B.seq_stride * 4
                                 : (long long)K * d.B.seq_stride * 4;
  __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)b_bytes, 0x00020000);
  unsigned offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    offA[q] = (unsigned)(((long long)(rid + 8 * q) * d.A.seq_stride + m0 + 4 * c) * 4);
    offB[q] = (unsigned)(((long long)(rid + 8 * q) * d.B.seq_stride + n0 + 4 * c) * 4);
  }
  const int stepA = (int)(BK * d.A.seq_stride * 4), stepB = (int)(BK * d.B.seq_stride * 4);
  const unsigned mgP0 = BWIN ? magic_of(d.B.P0) : 0u;
  const int colB = (n0 + 4 * c) * 4;
  // LDS store offsets (row r, chunk c): r*256 + (((c >> 3) ^ (r & 3)) << 6) + (c & 7)*8
  int wofs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rid + 8 * q;
    wofs[q] = r * 256 + ((((c >> 3) ^ (r & 3))) << 6) + (c & 7) * 8;
  }
  // transposed-fragment addresses: 16-lane group g = (m half, k half), lane i = (row i>>2, quad i&3)
  const int g = lane >> 4, i16 = lane & 15;
  const int rrow = (g >> 1) * 8 + (i16 >> 2), sw = i16 >> 2, within = (g & 1) * 32 + (i16 & 3) * 8;
  int rofA[2], rofB[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    rofA[t] = rrow * 256 + ((((wm * 2 + t) ^ sw)) << 6) + within;
    rofB[t] = rrow * 256 + ((((wn * 2 + t) ^ sw)) << 6) + within;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  int ka = (int)((long long)kbeg * d.A.seq_stride * 4), kb = (int)((long long)kbeg * d.B.seq_stride * 4);
  const int ka0 = ka, kb0 = kb;
  int srow = kbeg;   // first row of the slab being loaded (window operands)
  auto gload = [&](bool valid, u32x4 (&la)[4], u32x4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      la[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, offA[q], valid ? ka : ka0, 0);
      if constexpr (BWIN) {
        const int r = (valid ? srow : kbeg) + rid + 8 * q;
        const int sq = fast_div(r, d.B.P0, mgP0), p = r - sq * d.B.P0;
        const long long off = ((long long)sq * d.B.seq_stride + (long long)(p * d.B.step0 - d.B.pad0) * d.B.unit) * 4 + colB;
        const unsigned vo = (r < K && off >= 0 && off < b_bytes) ? (unsigned)off : 0x80000000u;
        lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, vo, 0, 0);
      } else {
        lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offB[q], valid ? kb : kb0, 0);
      }
    }
  };
  auto advance = [&]() {
    ka += stepA;
    kb += stepB;
    srow += BK;
  };
  auto lstore = [&](int bufoff, const u32x4 (&la)[4], const u32x4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned char* pa = sm + bufoff + wofs[q];
      *reinterpret_cast<u32x2*>(pa) = u32x2{la[q].x, la[q].y};
      *reinterpret_cast<u32x2*>(pa + PL) = u32x2{la[q].z, la[q].w};
      *reinterpret_cast<u32x2*>(pa + 2 * PL) = u32x2{lb[q].x, lb[q].y};
      *reinterpret_cast<u32x2*>(pa + 3 * PL) = u32x2{lb[q].z, lb[q].w};
    }
  };
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];   // [0..1] hi of the two sub-tiles, [2..3] lo
  auto frags = [&](int bufoff, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
    const unsigned char* base = sm + bufoff + ks * 16 * 256;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fa[t] = tr_frag(base + rofA[t]);
      fa[2 + t] = tr_frag(base + PL + rofA[t]);
      fb[t] = tr_frag(base + 2 * PL + rofB[t]);
      fb[2 + t] = tr_frag(base + 3 * PL + rofB[t]);
    }
  };
  auto mfma12 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const bf16x8 av = term == 0 ? fa[2 + mi] : fa[mi];
          const bf16x8 bv = term == 1 ? fb[2 + ni] : fb[ni];
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mi][ni], 0, 0, 0);
        }
  };
  u32x4 xa[4], xb[4], ya[4], yb[4];
  gload(true, xa, xb);
  lstore(0, xa, xb);
  advance();
  gload(nt > 1, xa, xb);
  __syncthreads();
  frags(0, 0, fa0, fb0);
  auto step3 = [&](int t, int curoff, int nxtoff, const u32x4 (&wa)[4], const u32x4 (&wb)[4],
                   u32x4 (&la)[4], u32x4 (&lb)[4]) {
    frags(curoff, 1, fa1, fb1);
    advance();
    gload(t + 2 < nt, la, lb);    // past the end: re-read the first slab (never used)
    mfma12(fa0, fb0);
    lstore(nxtoff, wa, wb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    frags(nxtoff, 0, fa0, fb0);
    mfma12(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step3(t, 0, BUF, xa, xb, ya, yb);
    step3(t + 1, BUF, 0, ya, yb, xa, xb);
  }
  if (t < nt) step3(t, 0, BUF, xa, xb, ya, yb);
  gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, blockIdx.z == 0);
}

// ---- weight gradient with fp32-class products on the bf16 pipe (precision 3, form 2) -----------------
// The operands of gemm_leanw3_kernel (A plain R x M, B plain or an unbounded 1-D window), read as the
// fp32 tensors they are: a weight gradient reduces over ROWS, so the row-major three-piece images of
// the forward kernel are of no use here -- instead every thread splits the 4-float chunks it loads
// into three bf16 pieces on their way into LDS (5.5 VALU instructions per element next to 48 MFMAs
// per slab and wave: the other block of the CU runs its MFMAs meanwhile), K-major planes
// [A p0 | A p1 | A p2 | B p0 | B p1 | B p2] of 32 rows x 128 bf16 with gemm_leanw3_kernel's swizzle, the
// transposing ds_read_b64_tr_b16 fragments, and the six products with i + j <= 2 (smallest first).
// One 48 KB LDS buffer, two blocks per CU:  split + store slab t -> request slab t + 1 -> barrier ->
// read its 24 fragments -> barrier -> 48 MFMAs.
// (gemm_x6_kernel raises its waves' priority for the MFMA phase: +2 ... 14 % in the step; the kernels that
// split operands on the VALU lose with it -- the other block's split is what feeds their next slab)
#define X6_MFMA_PRIO 1
// lab builds (tools/micro/x6lab.sh; timing only): F2G_X6LAB bit 1 = the six-product kernels skip their
// epilogue, bit 2 = they skip the read-back that writes the result's three-piece image
#ifndef F2G_X6LAB
#define F2G_X6LAB 0
#endif
#if F2G_X6LAB & 1
#define X6LAB_EPI if (acc[0][0][0] == 1.2345e30f)
#else
#define X6LAB_EPI
#endif
// bits 8 ... 64 (gemm_x6f_kernel, gemm_leanw6_kernel): 8 = only the first slab is loaded from memory, 16 = no split
// arithmetic (the pieces are raw bit fields), 32 = 4 of the 48 MFMAs of a slab, 64 = fragments read once
#define X6LAB_KEEP(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
// bit 512: gemm_x6f_kernel stages without LDS stores.
// bit 256: gemm_x6f_kernel times the three intervals of its slab loop with s_memtime (wave 0 of every block; the
// stamps' results are collected by the loop's own lgkmcnt(0) waits, nothing is added to the critical path) and
// adds them to g_x6prof: [0] MFMA chain with the staging between, [1] wait for the block at the "stores visible"
// barrier, [2] fragment reads + "fragments read" barrier, [3] iterations, [4] kernel entry -> first chain (prologue),
// [5] the last slab's chain (12 x 4 MFMAs issued), [6] epilogue, [7] blocks; f2g_lab_x6prof reads and clears them
#if F2G_X6LAB & 256
__device__ unsigned long long g_x6prof[8];
extern "C" int f2g_lab_x6prof(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x6prof), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_x6prof), z, sizeof(z)) != hipSuccess;
}
#define X6PROF_STAMP(t) asm volatile("s_memtime %0" : "=s"(t) : : "memory")
#define X6PROF_ACC(sum, t1, t0)                                                                              \
  {                                                                                                          \
    unsigned dt_;                                                                                            \
    asm volatile("s_sub_u32 %0, %1, %2" : "=s"(dt_) : "s"((unsigned)(t1)), "s"((unsigned)(t0)) : "memory"); \
    sum += dt_;                                                                                              \
  }
#else
#define X6PROF_STAMP(t)
#define X6PROF_ACC(sum, t1, t0)
#endif
#if F2G_X6LAB & 2
#define X6LAB_X3 acc[0][0][1] == 1.2345e30f &&
#else
#define X6LAB_X3
#endif
// Main-loop barrier of the six-product kernels: the LDS traffic of this wave is done, then the block barrier.
// __syncthreads() also waits for vmcnt(0), i.e. for the NEXT slab's global loads the wave has just issued --
// the prefetch would be drained at every slab (round 5; gemm_x6p.hip has it from the start).
#if F2G_X6LAB & 4      // (lab build: the old barrier, for A/B runs)
#define X6_LDS_BARRIER() __syncthreads()
#else
#define X6_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// three bf16 pieces of four floats (common.h: f2g_split3_pair -- 18 VALU instructions per chunk)
__device__ __forceinline__ void split3x4(const u32x4& v, u32x2& p0, u32x2& p1, u32x2& p2) {
#if F2G_X6LAB & 16
  p0 = u32x2{v.x, v.y}; p1 = u32x2{v.z, v.w}; p2 = u32x2{v.x, v.w};
  return;
#endif
  // (by value first: __builtin_bit_cast applied to a vector-element expression reads element 0)
  const unsigned u0 = v.x, u1 = v.y, u2 = v.z, u3 = v.w;
  unsigned a0, a1, a2, b0, b1, b2;
  f2g_split3_pair(__uint_as_float(u0), __uint_as_float(u1), a0, a1, a2);
  f2g_split3_pair(__uint_as_float(u2), __uint_as_float(u3), b0, b1, b2);
  p0 = u32x2{a0, b0};
  p1 = u32x2{a1, b1};
  p2 = u32x2{a2, b2};
}

template <bool BWIN>
__global__ __launch_bounds__(256, 2)
void gemm_leanw6_kernel(const f2g_gemm_desc d, int M, int N, int K, int kchunk) {
  constexpr int PL = 32 * 256;            // bytes of one plane (32 rows x 128 bf16)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* sm = reinterpret_cast<unsigned char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  const int kbeg = blockIdx.z * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt <= 0) return;
  const int rid = tid >> 5, c = tid & 31;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.A.base, 0, (unsigned)((long long)K * d.A.seq_stride * 4), 0x00020000);
  const long long b_bytes = BWIN ? (long long)(d.B.rows / d.B.P0) * d.B.seq_stride * 4
                                 : (long long)K * d.B.seq_stride * 4;
  __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)b_bytes, 0x00020000);
  unsigned offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    offA[q] = (unsigned)(((long long)(rid + 8 * q) * d.A.seq_stride + m0 + 4 * c) * 4);
    offB[q] = (unsigned)(((long long)(rid + 8 * q) * d.B.seq_stride + n0 + 4 * c) * 4);
  }
  const int stepA = (int)(BK * d.A.seq_stride * 4), stepB = (int)(BK * d.B.seq_stride * 4);
  const unsigned mgP0 = BWIN ? magic_of(d.B.P0) : 0u;
  const int p0one = BWIN && d.B.P0 == 1 ? -1 : 0;
  const int colB = (n0 + 4 * c) * 4;
  int wofs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rid + 8 * q;
    wofs[q] = r * 256 + ((((c >> 3) ^ (r & 3))) << 6) + (c & 7) * 8;
  }
  const int g = lane >> 4, i16 = lane & 15;
  const int rrow = (g >> 1) * 8 + (i16 >> 2), sw = i16 >> 2, within = (g & 1) * 32 + (i16 & 3) * 8;
  int rofA[2], rofB[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    rofA[t] = rrow * 256 + ((((wm * 2 + t) ^ sw)) << 6) + within;
    rofB[t] = rrow * 256 + ((((wn * 2 + t) ^ sw)) << 6) + within;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int ka0 = (int)((long long)kbeg * d.A.seq_stride * 4), kb0 = (int)((long long)kbeg * d.B.seq_stride * 4);
  u32x4 xa[4], xb[4];
  // chunk q (rows rid + 8 q) of slab s of this block's K range; s >= nt: the first slab again (never used)
  auto gload1 = [&](int q, int s) {
    const bool valid = s < nt;
    const int s_ = valid ? s : 0;
    // rows past K pair with nothing: zeros (the resource ends at K rows for A; B is tested)
    const int r = kbeg + s_ * BK + rid + 8 * q;
    // (an offset with bit 31 set lies behind every resource: the load returns zeros.  Written as arithmetic: as a
    // select the compiler turned it into two loads under complementary exec masks -- a branch inside the chain)
    const unsigned past = (unsigned)(r >= K) << 31;
    xa[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, offA[q] | past, ka0 + s_ * stepA, 0);
    if constexpr (BWIN) {
      // (fast_div without its d == 1 branch: control flow would cut the MFMA chain's scheduling region)
      int sq = (int)__umulhi((unsigned)r, mgP0);
      sq -= (sq * d.B.P0 > r) ? 1 : 0;
      sq += (r - sq) & p0one;
      const int pp = r - sq * d.B.P0;
      const long long off = ((long long)sq * d.B.seq_stride + (long long)(pp * d.B.step0 - d.B.pad0) * d.B.unit) * 4 + colB;
      const unsigned vo = (unsigned)off | ((unsigned)!(r < K && off >= 0 && off < b_bytes) << 31);
      xb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, vo, 0, 0);
    } else {
      xb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offB[q] | past, kb0 + s_ * stepB, 0);
    }
  };
  auto stage = [&](int q) {      // split chunk q of both operands into the K-major planes
    u32x2 p0, p1, p2;
    unsigned char* pa = sm + wofs[q];
    split3x4(xa[q], p0, p1, p2);
    *reinterpret_cast<u32x2*>(pa) = p0;
    *reinterpret_cast<u32x2*>(pa + PL) = p1;
    *reinterpret_cast<u32x2*>(pa + 2 * PL) = p2;
    split3x4(xb[q], p0, p1, p2);
    *reinterpret_cast<u32x2*>(pa + 3 * PL) = p0;
    *reinterpret_cast<u32x2*>(pa + 4 * PL) = p1;
    *reinterpret_cast<u32x2*>(pa + 5 * PL) = p2;
  };
  bf16x8 fa[2][3][2], fb[2][3][2];
  auto frags = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          fa[ks][pc][tt] = tr_frag(sm + ks * 16 * 256 + pc * PL + rofA[tt]);
          fb[ks][pc][tt] = tr_frag(sm + ks * 16 * 256 + (3 + pc) * PL + rofB[tt]);
        }
  };
  // the 48 MFMAs of a slab as 12 groups of four (one product term of one k step), smallest terms first
  auto mf4 = [&](int g) {
    const int ks = g / 6, r = g % 6;
    const int i = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : r == 3 ? 0 : r == 4 ? 1 : 0;
    const int j = r < 3 ? 2 - i : r < 5 ? 1 - i : 0;
    if ((F2G_X6LAB & 32) && g != 0) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][mi], fb[ks][j][ni], acc[mi][ni], 0, 0, 0);
  };
  // Schedule (round 6; gemm_x6f_kernel has the measurements): the single operand buffer is dead once every wave
  // holds its fragments, so slab t + 1 is split and stored BETWEEN the MFMAs of slab t, in the same wave's
  // instruction stream -- one chunk of either operand per quarter of the chain, its registers requested again for
  // slab t + 2 at once --, and only barrier, fragment reads, barrier stand between two MFMA chains.
#pragma unroll
  for (int q = 0; q < 4; ++q) gload1(q, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) stage(q);
#pragma unroll
  for (int q = 0; q < 4; ++q) gload1(q, 1);
  X6_LDS_BARRIER();
  frags();
  X6_LDS_BARRIER();
  for (int t = 0; t + 1 < nt; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mf4(3 * q);
      stage(q);
      mf4(3 * q + 1);
      gload1(q, t + 2);
      mf4(3 * q + 2);
#if !(F2G_X6LAB & 128)
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, BWIN ? 5 : 4, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    X6_LDS_BARRIER();
    frags();
    X6_LDS_BARRIER();
  }
#pragma unroll
  for (int g = 0; g < 12; ++g) mf4(g);
  gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, blockIdx.z == 0);
}

// ---- lean weight-gradient kernel, exact fp32 ---------------------------------------------------
// Same operands as gemm_leanw3_kernel (A plain R x M, B plain or an unbounded 1-D window), fp32 MFMA.
// v_mfma_f32_32x32x2_f32 takes ONE k per lane half, so K-major tiles are its natural layout: the
// slab [32 k][128 m] is stored as it arrives (ds_write_b128) and lane (m = li, k = 2s + hh) reads
// single floats, 32 consecutive ones per lane half: conflict-free ds_read_b32 at per-lane base +
// immediate offsets.  As in the forward lean kernel nothing in the K loop touches the vector ALU:
// the K advance of both operands is scalar.  For a window operand the slab's first row (sequence,
// position) is walked by SALU and the rows of a slab add a per-thread constant; only a slab that
// straddles a sequence end (or starts before the buffer) pays a few VALU instructions to redirect
// the rows behind the boundary.
template <bool BWIN>
__global__ __launch_bounds__(256, 2)
void gemm_leanw_kernel(const f2g_gemm_desc d, int M, int N, int K, int kchunk) {
  constexpr int TP = 32 * 128;            // floats of one operand tile
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [buf][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  const int kbeg = blockIdx.z * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt <= 0) return;
  const int rid = tid >> 5, c = tid & 31;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.A.base, 0, (unsigned)((long long)K * d.A.seq_stride * 4), 0x00020000);
  const long long b_bytes = BWIN ? (long long)(d.B.rows / d.B.P0) * d.B.seq_stride * 4
                                 : (long long)K * d.B.seq_stride * 4;
  __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)b_bytes, 0x00020000);
  const long long rowB = BWIN ? (long long)d.B.step0 * d.B.unit * 4 : d.B.seq_stride * 4;   // bytes per row
  unsigned offA[4], offB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    offA[q] = (unsigned)(((long long)(rid + 8 * q) * d.A.seq_stride + m0 + 4 * c) * 4);
    offB[q] = (unsigned)((long long)(rid + 8 * q) * rowB + (n0 + 4 * c) * 4);
  }
  const int stepA = (int)(BK * d.A.seq_stride * 4);
  int ka = (int)((long long)kbeg * d.A.seq_stride * 4);
  const int ka0 = ka;
  // B: scalar byte offset of the slab's first row.  Window operand: (sequence sq, position p0)
  int sq = 0, p0 = 0;
  long long kb = (long long)kbeg * d.B.seq_stride * 4;
  const int wrapjump = BWIN ? (int)((d.B.seq_stride - (long long)d.B.P0 * d.B.step0 * d.B.unit) * 4) : 0;
  if (BWIN) {
    sq = kbeg / d.B.P0;
    p0 = kbeg - sq * d.B.P0;
    kb = ((long long)sq * d.B.seq_stride + (long long)(p0 * d.B.step0 - d.B.pad0) * d.B.unit) * 4;
  }
  const long long kb_first = kb;
  const int p_first = p0;
  float* wA = smem + rid * 128 + c * 4;
  float* wB = smem + TP + rid * 128 + c * 4;
  const float* rA = smem + h * 128 + wm * 64 + li;
  const float* rB = smem + TP + h * 128 + wn * 64 + li;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  auto gload = [&](int soa, long long sob, int pp, u32x4 (&la)[4], u32x4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) la[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, offA[q], soa, 0);
    if (BWIN && (pp + BK > d.B.P0 || sob < 0 || sob + 32 * rowB + 512 > 0x7fffffffll)) {
      // (rare, uniform) the slab straddles a sequence end or touches the buffer's ends: per-row offsets
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rid + 8 * q;
        // (P0 >= 16 is required: a slab's 32 rows cross at most two sequence ends)
        const int wraps = (pp + r >= d.B.P0 ? 1 : 0) + (pp + r >= 2 * d.B.P0 ? 1 : 0);
        long long off = sob + (long long)offB[q] + (long long)wraps * wrapjump;
        if (pp + r >= 3 * d.B.P0) off = -1;   // (defensive)
        const unsigned vo = (off >= 0 && off < b_bytes) ? (unsigned)off : 0x80000000u;
        lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, vo, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) lb[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, offB[q], (int)sob, 0);
    }
  };
  auto advance = [&]() {
    ka += stepA;
    if (BWIN) {
      p0 += BK;
      kb += BK * rowB;
      while (p0 >= d.B.P0) {      // (twice for sequences shorter than a slab)
        p0 -= d.B.P0;
        kb += wrapjump;
      }
    } else {
      kb += BK * rowB;
    }
  };
  auto lstore = [&](int bufoff, const u32x4 (&la)[4], const u32x4 (&lb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(wA + bufoff + q * 8 * 128) = la[q];
      *reinterpret_cast<u32x4*>(wB + bufoff + q * 8 * 128) = lb[q];
    }
  };
  // Fragments: single floats at (k pair s2, sub-tile i) = base + (s2 * 256 + i * 32) floats.  Written
  // as plain loads the compiler pairs them into ds_read2_b32, whose 8-bit offsets cannot span the
  // 1 KB row pitch: it then spends one address VALU per read inside the K loop -- next to fp32 MFMAs
  // that is the expensive kind of instruction.  ds_read_b32 takes a 16-bit immediate: one base VGPR
  // per operand and immediates for everything else (asm), waits by hand, one k pair ahead.
  unsigned aA = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(rA);
  unsigned aB = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(rB);
  auto mfma_slab = [&](auto bufc, unsigned pa, unsigned pb) {
    constexpr int bufoff = decltype(bufc)::value;
    float a[2][2], b[2][2];
    auto rd = [](auto s2c, float (&av)[2], float (&bv)[2], unsigned qa, unsigned qb) {
      constexpr int o = (bufoff + decltype(s2c)::value * 256) * 4;
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(av[0]) : "v"(qa), "n"(o));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(av[1]) : "v"(qa), "n"(o + 128));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bv[0]) : "v"(qb), "n"(o));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bv[1]) : "v"(qb), "n"(o + 128));
    };
    auto mm = [&](const float (&av)[2], const float (&bv)[2]) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    };
    rd(std::integral_constant<int, 0>{}, a[0], b[0], pa, pb);
    auto pair = [&](auto s2c) {
      constexpr int s2 = decltype(s2c)::value;
      constexpr int cu = s2 & 1, nx = cu ^ 1;
      if constexpr (s2 + 1 < 16) {
        rd(std::integral_constant<int, s2 + 1>{}, a[nx], b[nx], pa, pb);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[cu][0]), "+v"(a[cu][1]), "+v"(b[cu][0]), "+v"(b[cu][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cu][0]), "+v"(a[cu][1]), "+v"(b[cu][0]), "+v"(b[cu][1]));
      }
      mm(a[cu], b[cu]);
    };
    pair(std::integral_constant<int, 0>{});
    pair(std::integral_constant<int, 1>{});
    pair(std::integral_constant<int, 2>{});
    pair(std::integral_constant<int, 3>{});
    pair(std::integral_constant<int, 4>{});
    pair(std::integral_constant<int, 5>{});
    pair(std::integral_constant<int, 6>{});
    pair(std::integral_constant<int, 7>{});
    pair(std::integral_constant<int, 8>{});
    pair(std::integral_constant<int, 9>{});
    pair(std::integral_constant<int, 10>{});
    pair(std::integral_constant<int, 11>{});
    pair(std::integral_constant<int, 12>{});
    pair(std::integral_constant<int, 13>{});
    pair(std::integral_constant<int, 14>{});
    pair(std::integral_constant<int, 15>{});
  };
  constexpr int BUFF = 2 * TP;
  {
    u32x4 la[4], lb[4];
    gload(ka, kb, p0, la, lb);
    lstore(0, la, lb);
  }
  __syncthreads();
  auto step = [&](int t, auto curc, int nxtoff) {
    u32x4 la[4], lb[4];
    advance();
    const bool again = t + 1 < nt;   // the last iteration re-reads the first slab (never used)
    gload(again ? ka : ka0, again ? kb : kb_first, again ? p0 : p_first, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    mfma_slab(curc, aA, aB);
    __builtin_amdgcn_sched_barrier(0);
    lstore(nxtoff, la, lb);
    __syncthreads();
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step(t, std::integral_constant<int, 0>{}, BUFF);
    step(t + 1, std::integral_constant<int, BUFF>{}, 0);
  }
  if (t < nt) step(t, std::integral_constant<int, 0>{}, BUFF);
  gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, blockIdx.z == 0);
}

// form 2 on the kernel above: split-bf16 with both operands pre-split, whole 128 x 128 tiles,
// A a plain matrix, B plain or an `unbounded` single-segment 1-D window
inline bool leanw_ok(const f2g_gemm_desc& d) {
  const f2g_operand& A = d.A;
  const f2g_operand& B = d.B;
  if (d.form != 2 || A.rows != B.rows || A.rows <= 0) return false;
  if (!host_plain(A) || A.alpha || B.alpha || B.reflect || B.lrelu_src) return false;
  if (A.cols % 128 || B.cols % 128 || !al16(A.base) || !al16(B.base)) return false;
  if ((A.seq_stride & 3) || (B.seq_stride & 3)) return false;
  if ((long long)A.rows * A.seq_stride * 4 >= 0x7ff00000ll) return false;
  if (host_plain(B)) return (long long)B.rows * B.seq_stride * 4 < 0x7ff00000ll;
  if (!B.unbounded || B.P1 != 1 || B.P0 < 1 || B.rows % B.P0 || B.seglen < B.cols) return false;
  if ((((long long)B.step0 * B.unit) & 3) || (((long long)B.pad0 * B.unit) & 3)) return false;
  return (long long)(B.rows / B.P0) * B.seq_stride * 4 < 0x7ff00000ll;
}

// exact fp32 weight gradient (form 2, both operands fp32): the K-major lean kernel where its shape conditions
// hold and every block walks a long reduction (>= 4096 rows: the MPD weight gradients, 115 -> 125-131
// TFLOP/s, step 254.5 -> 252.6 ms; on the generator's 6016-row weight gradients the generic kernel's 8 waves
// hide the short K loops better: 92 vs 83).  option lean_wgrad: 0 off, 1 auto (default), 2 always.  ONE rule for
// f2g_gemm's dispatch and for the host's query (f2g_gemm_wgrad_lean).
inline bool leanw_fp32_takes(const f2g_gemm_desc& d, int split) {
  const int leanw_mode = f2g_opt(F2G_OPT_LEAN) == 0 ? 0 : f2g_opt(F2G_OPT_LEAN_WGRAD);
  if (d.form != 2 || d.A.split || d.B.split || split < 1) return false;
  // (its scalar row walk assumes that a slab crosses at most two sequence ends)
  return leanw_mode > 0 && d.precision == 0 && d.E.atomic && leanw_ok(d) &&
         (host_plain(d.B) || d.B.P0 >= 16) && (leanw_mode > 1 || d.A.rows / split >= 4096);
}

int launch_leanw(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  constexpr size_t smem = 2 * 2 * 32 * 128 * sizeof(float);
  int kchunk = ((K + split - 1) / split + BK - 1) / BK * BK;
  const int zs = (K + kchunk - 1) / kchunk;
  dim3 grid(M / 128, N / 128, zs);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  g_last_path = 1;
  if (host_plain(d.B))
    hipLaunchKernelGGL(gemm_leanw_kernel<false>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  else
    hipLaunchKernelGGL(gemm_leanw_kernel<true>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  return f2g_check_launch();
}

int launch_leanw3(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  constexpr size_t smem = 2 * 4 * 32 * 256;
  int kchunk = ((K + split - 1) / split + BK - 1) / BK * BK;
  const int zs = (K + kchunk - 1) / kchunk;
  dim3 grid(M / 128, N / 128, zs);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw3_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw3_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  g_last_path = 1;
  if (host_plain(d.B))
    hipLaunchKernelGGL(gemm_leanw3_kernel<false>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  else
    hipLaunchKernelGGL(gemm_leanw3_kernel<true>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  return f2g_check_launch();
}

int launch_leanw6(const f2g_gemm_desc& d, int M, int N, int K, int split, hipStream_t st) {
  if (f2g_leanw6t_ok(d, split)) {       // round 5: all taps of a stride-1 layer from one staged window
    g_last_path = 4;
    return f2g_launch_leanw6t(d, split, st);
  }
  constexpr size_t smem = 6 * 32 * 256;
  int kchunk = ((K + split - 1) / split + BK - 1) / BK * BK;
  const int zs = (K + kchunk - 1) / kchunk;
  dim3 grid(M / 128, N / 128, zs);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw6_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw6_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  g_last_path = 4;
  if (host_plain(d.B))
    hipLaunchKernelGGL(gemm_leanw6_kernel<false>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  else
    hipLaunchKernelGGL(gemm_leanw6_kernel<true>, grid, dim3(256), smem, st, d, M, N, K, kchunk);
  return f2g_check_launch();
}

// Stream-K decision for the lean kernel: units per block, or 0 to keep the classic tile grid.
// Measured on the stage-2 step (B = 64): evening out the rounds lifts the kernels alone on the chip
// (GEMM class 283.8 -> 275.4 ms serialised) but not the step itself, whose launch lanes already
// fill one kernel's idle CUs with another lane's work (266.5 -> 269.0 ms: the zero fill and the
// atomic epilogues remain).  Default: only the latency regime (fewer tiles than half the CUs:
// batch-1 chunked synthesis, the per-item MLPs), where nothing else runs beside the kernel;
// option streamk = 2 applies it to every ragged tile grid.
inline int lean_stream_k(int M, int N, int K, bool all_grids) {
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int nt = K / BK;
  if (nt < 16) return 0;
  const long long total = tiles * nt;
  if (tiles * 2 > 256) {
    if (!all_grids) return 0;
    const double rounds = (double)tiles / 512.0;
    const double eff = rounds / (double)((tiles + 511) / 512);
    if (eff > 0.9) return 0;                     // the tile grid already fills its rounds
  }
  long long upb = (total + 511) / 512;
  const int min_slabs = f2g_opt(F2G_OPT_STREAMK_MIN);      // (default 4; 8 until round 6: 64-row time-path GEMMs 54 -> 44 us)
  if (upb < min_slabs) upb = min_slabs;
  return (int)upb;
}

}  // namespace

// Which kernel family the last f2g_gemm call of this process dispatched to (diagnostics for the
// benchmark's per-kernel roofline; not thread safe): 0 generic MFMA kernels, 1 lean kernel,
// 2 lean kernel in stream-K mode, 3 narrow (VALU) kernels, 4 the precision-3 kernels, 5 their 32-column instance.
// ---- fp32-class GEMM on the bf16 matrix pipe (precision 3; round 3) -------------------------------
// Every fp32 operand is split into THREE bf16 pieces x = p0 + p1 + p2 (24 mantissa bits) and a product
// is the six MFMAs with i + j <= 2 (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0; fp32 accumulation, smallest
// terms first): what is dropped is <= 2^-24 relative -- the error class of fp32 rounding itself
// (measured 8e-8 ... 1e-7 of sum |a w| where the fp32 fmaf chain has 7e-8 ... 2e-7, tools/micro/x6_lab.hip),
// not the 2^-16 of the two-piece mode.  The fp32 MFMA needs 8 x 64 cycles for the block of products
// these six 32-cycle MFMAs cover, so the matrix pipe is 2.7x less busy per FLOP.
// Operand image (f2g_split_bf16x3): row-major, per 32-element slab of a row its three pieces side by
// side, [row][K / 32][piece][32] bf16 = 192 contiguous bytes per row and slab (whole cache lines).
// Kernel: the simplest structure that works -- 128 x 128 x 32 tiles, 4 waves of 64 x 64, single LDS
// buffer (rows 208 bytes apart: 52 dwords, conflict-free for ds_read_b128), the next slab's operands
// requested one pass ahead by buffer loads with per-thread constant offsets, the whole slab's fragments
// in registers, TWO blocks per CU hide each other's store / barrier / read phases:
//   store slab t -> barrier -> read its 24 fragments -> barrier -> 48 MFMAs.
// Plain (rows x K) operands only (the generator's 1x1 convolutions and linears); epilogue = the generic
// kernel's (bias, residual, PReLU with both outputs, PReLU backward with column sums, ...).
__global__ __launch_bounds__(256) void split3_img_kernel(__bf16* __restrict__ dst, const float* __restrict__ src,
                                                         long long ld, long long rows, int K) {
  const long long total = rows * (K / 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (K / 4);
    const int k4 = (int)(i - r * (K / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + k4);
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short p[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 a = (__bf16)x[e];
      const float r1 = x[e] - (float)a;
      const __bf16 b = (__bf16)r1;
      const __bf16 c = (__bf16)(r1 - (float)b);
      p[0][e] = __builtin_bit_cast(unsigned short, a);
      p[1][e] = __builtin_bit_cast(unsigned short, b);
      p[2][e] = __builtin_bit_cast(unsigned short, c);
    }
    __bf16* o = dst + (r * (K / 32) + k4 / 32) * 96 + (k4 & 31);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<uint2*>(o + 32 * q) =
          make_uint2(p[q][0] | ((unsigned)p[q][1] << 16), p[q][2] | ((unsigned)p[q][3] << 16));
  }
}

// The tile's three-piece image for the next GEMM (f2g_epilogue.x3_out): read back what the block has just
// stored (L2; the barrier orders the block's own stores before these loads) and write whole 16-byte pieces.
template <int ROWS = 128, int NTHR = 256>
__device__ __forceinline__ void x3_tile_readback(const f2g_epilogue& E, int M, int N, int m0, int n0, int tid) {
  __syncthreads();
  for (int u = tid; u < ROWS * 16; u += NTHR) {
    const int row = m0 + (u >> 4), col = n0 + (u & 15) * 8;
    if (row >= M || col >= N) continue;
    long long off;
    if (E.P0o > 0) {
      const int sq = row / E.P0o;
      off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o + E.off_o + col;
    } else {
      off = (long long)row * E.ldc + col;
    }
    const float4 v0 = *reinterpret_cast<const float4*>(E.C + off);
    const float4 v1 = *reinterpret_cast<const float4*>(E.C + off + 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned pk[3][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 a = (__bf16)x[e];
      const float r1 = x[e] - (float)a;
      const __bf16 b = (__bf16)r1;
      const __bf16 c = (__bf16)(r1 - (float)b);
      const unsigned sa = __builtin_bit_cast(unsigned short, a), sb = __builtin_bit_cast(unsigned short, b),
                     sc = __builtin_bit_cast(unsigned short, c);
      if (e & 1) pk[0][e >> 1] |= sa << 16, pk[1][e >> 1] |= sb << 16, pk[2][e >> 1] |= sc << 16;
      else pk[0][e >> 1] = sa, pk[1][e >> 1] = sb, pk[2][e >> 1] = sc;
    }
    __bf16* q = reinterpret_cast<__bf16*>(E.x3_out) + (off >> 5) * 96 + (off & 31);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      *reinterpret_cast<uint4*>(q + 32 * pc) = make_uint4(pk[pc][0], pk[pc][1], pk[pc][2], pk[pc][3]);
  }
}

// A rows: plain (row r at r * K * 6 bytes of a dense image) or single-segment windows over the flat image
// of a contiguous buffer (MPD halo maps: row (s, p) at s * seq6 + p * step6 + off6 bytes, K contiguous --
// element e of a contiguous buffer lives at (e / 32) * 192 + piece * 64 + (e % 32) * 2 whatever its row
// length, so a window that starts on a 32-element boundary addresses the image like the tensor)
struct x6_rows {
  int P0;
  unsigned seq6, step6, off6, bytes;
};

__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                         const x6_rows R, const int wide) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  constexpr int PITCH = 208, OPER = 128 * PITCH, NJ = 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned rowbytes = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, R.bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)N * rowbytes, 0x00020000);
  // chunk id = tid + 256 j -> (row of the tile, 16-byte chunk of the row's 192 bytes); rows past the
  // end lie outside the resource: zeros
  unsigned voA[NJ], voW[NJ];
  int lo[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = tid + 256 * j, row = id / 12, c = id - row * 12;
    const int r = m0 + row, sq = r / R.P0;
    voA[j] = r < M ? (unsigned)sq * R.seq6 + (unsigned)(r - sq * R.P0) * R.step6 + R.off6 + c * 16
                   : 0xf0000000u;                 // (outside the resource: zeros)
    voW[j] = (unsigned)(n0 + row) * rowbytes + c * 16;
    lo[j] = row * PITCH + c * 16;
  }
  u32x4 xa[NJ], xw[NJ];
  auto gload = [&](int t) {
    const int so = t * 192;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], so, 0);
      xw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], so, 0);
    }
  };
  const unsigned char* rA = smem6 + (wm * 64 + li) * PITCH + h * 16;
  const unsigned char* rB = smem6 + OPER + (wn * 64 + li) * PITCH + h * 16;
  const int nt = K / 32;
  gload(0);
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      *reinterpret_cast<u32x4*>(smem6 + lo[j]) = xa[j];
      *reinterpret_cast<u32x4*>(smem6 + OPER + lo[j]) = xw[j];
    }
    gload(t + 1 < nt ? t + 1 : 0);       // (past the end: re-read, never used)
    X6_LDS_BARRIER();
    bf16x8 fa[2][3][2], fb[2][3][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[ks][p][i] = *reinterpret_cast<const bf16x8*>(rA + p * 64 + i * 32 * PITCH + ks * 32);
          fb[ks][p][i] = *reinterpret_cast<const bf16x8*>(rB + p * 64 + i * 32 * PITCH + ks * 32);
        }
    X6_LDS_BARRIER();
    __builtin_amdgcn_s_setprio(X6_MFMA_PRIO);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int sdeg = 2; sdeg >= 0; --sdeg)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int j = sdeg - i;
          if (j < 0 || j > 2) continue;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][mi], fb[ks][j][ni], acc[mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  }
  if (wide) {
    // (every fragment read of the main loop lies before its last barrier: a wave that is through its MFMAs
    // may overlay the operand buffers with its private patch)
    X6LAB_EPI x6e::wide_epilogue(d.E, acc, M, N, m0 + wm * 64, n0 + wn * 64, lane, smem6 + wave * x6e::ESZ);
    return;
  }
  X6LAB_EPI gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, true);
  if (X6LAB_X3 d.E.x3_out) x3_tile_readback(d.E, M, N, m0, n0, tid);
}

// (Stride-1 conv windows -- the (5, 1) MPD layers and the two-tap residues of their stride-3 data gradients --
// run on the tap-walking ping-pong kernel of gemm_x6p.hip; its 128-row and single-group 256-row predecessors
// gemm_x6t_kernel / gemm_x6t8_kernel were removed in round 6.  Windows gemm_x6p_kernel does not take -- fewer
// than 64 channels per position, grids that do not fill the chip -- read their rows through the kernel above.)

// The same tile and schedule over the fp32 operands themselves (f2g_operand.split = 0): every thread
// splits the 4-float chunks it loads into the three pieces on their way into LDS, as gemm_leanw6_kernel
// does -- 4 bytes per element from L2 instead of 6, no image pass, no producer, 5.5 VALU instructions per
// element beside the 48 MFMAs per slab and wave.
template <bool WIMG>
__global__ __launch_bounds__(256, 2) void gemm_x6f_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                          const x6_rows R, const int wide) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
#if F2G_X6LAB & 256
  unsigned long long pe0, pe1, pe2, pe3;
  X6PROF_STAMP(pe0);
#endif
  // WIMG (round 5): the WEIGHT operand is its cached f2g_split_bf16x3 image (192 bytes per row and slab, stored
  // to LDS as it comes) -- every one of the M / 128 row tiles used to split the same weight slab again; only
  // the activation rows (read once per column tile) are still split here
  constexpr int PITCH = 208, OPER = 128 * PITCH, NJ = 4, NJW = WIMG ? 6 : 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // (here R.seq6 / step6 / off6 / bytes are in units of 4 bytes per element: x6_rows_of(d, 4))
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, R.bytes, 0x00020000);
  const unsigned rowbytesW = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.B.base, 0, WIMG ? (unsigned)N * rowbytesW : (unsigned)((long long)N * d.B.seq_stride * 4), 0x00020000);
  // chunk id = tid + 256 j -> (row of the tile, 16-byte chunk = 4 of the slab's 32 floats)
  unsigned voA[NJ], voW[NJW];
  int lo[NJ], loW[NJW];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = tid + 256 * j, row = id >> 3, c = id & 7;
    const int r = m0 + row, sq = r / R.P0;
    voA[j] = r < M ? (unsigned)sq * R.seq6 + (unsigned)(r - sq * R.P0) * R.step6 + R.off6 + c * 16 : 0xf0000000u;
    lo[j] = row * PITCH + c * 8;
  }
#pragma unroll
  for (int j = 0; j < NJW; ++j) {
    const int id = tid + 256 * j;
    if (WIMG) {
      const int row = id / 12, c = id - row * 12;
      voW[j] = n0 + row < N ? (unsigned)(n0 + row) * rowbytesW + c * 16 : 0xf0000000u;
      loW[j] = row * PITCH + c * 16;
    } else {
      const int row = id >> 3, c = id & 7;
      voW[j] = n0 + row < N ? (unsigned)((long long)(n0 + row) * d.B.seq_stride * 4) + c * 16 : 0xf0000000u;
      loW[j] = row * PITCH + c * 8;
    }
  }
  u32x4 xa[NJ], xw[NJW];
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], t * 128, 0);
#pragma unroll
    for (int j = 0; j < NJW; ++j) xw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], t * (WIMG ? 192 : 128), 0);
  };
  const unsigned char* rA = smem6 + (wm * 64 + li) * PITCH + h * 16;
  const unsigned char* rB = smem6 + OPER + (wn * 64 + li) * PITCH + h * 16;
  const int nt = K / 32;
  bf16x8 fa[2][3][2], fb[2][3][2];
  // Schedule (round 6).  The slab loop used to be phases -- split + store, barrier, fragments, barrier, 48 MFMAs --
  // and relied on the CU's other block to fill the matrix pipe meanwhile; measured, the phases simply ADD
  // (lab builds, tools/micro/x6lab_run.sh: without the split -32 us, without 44 of the 48 MFMAs -71 us of a 143 us
  // launch), and tools/micro/mfma_valu_overlap.hip shows why: VALU work of ANOTHER wave hides only partly behind a
  // wave's MFMAs, VALU work of the SAME wave's stream, issued between its MFMAs, hides completely.  The operand
  // buffer is dead once every wave holds its fragments, so the staging of slab t + 1 -- the split of the
  // activation chunks, the LDS stores of both operands, the requests for slab t + 2 -- now sits between the MFMAs
  // of slab t, one chunk per quarter of the chain, and only the fragment reads stand between two MFMA chains.
  auto stage_a = [&](int j) {
    u32x2 p0, p1, p2;
    split3x4(xa[j], p0, p1, p2);
#if F2G_X6LAB & 512      // (lab: the split without its LDS stores)
    asm volatile("" : : "v"(p0.x), "v"(p0.y), "v"(p1.x), "v"(p1.y), "v"(p2.x), "v"(p2.y));
    return;
#endif
    *reinterpret_cast<u32x2*>(smem6 + lo[j]) = p0;
    *reinterpret_cast<u32x2*>(smem6 + lo[j] + 64) = p1;
    *reinterpret_cast<u32x2*>(smem6 + lo[j] + 128) = p2;
  };
  auto stage_w = [&](int j) {
#if F2G_X6LAB & 512
    asm volatile("" : : "v"(xw[j].x), "v"(xw[j].y), "v"(xw[j].z), "v"(xw[j].w));
    return;
#endif
    if (WIMG) {
      *reinterpret_cast<u32x4*>(smem6 + OPER + loW[j]) = xw[j];
    } else {
      u32x2 p0, p1, p2;
      split3x4(xw[j], p0, p1, p2);
      *reinterpret_cast<u32x2*>(smem6 + OPER + loW[j]) = p0;
      *reinterpret_cast<u32x2*>(smem6 + OPER + loW[j] + 64) = p1;
      *reinterpret_cast<u32x2*>(smem6 + OPER + loW[j] + 128) = p2;
    }
  };
  auto frags = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[ks][p][i] = *reinterpret_cast<const bf16x8*>(rA + p * 64 + i * 32 * PITCH + ks * 32);
          fb[ks][p][i] = *reinterpret_cast<const bf16x8*>(rB + p * 64 + i * 32 * PITCH + ks * 32);
        }
  };
  // the 48 MFMAs of a slab as 12 groups of four (one product term of one k step), smallest terms first
  auto mf4 = [&](int g) {
    const int ks = g / 6, r = g % 6;
    const int i = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : r == 3 ? 0 : r == 4 ? 1 : 0;
    const int j = r < 3 ? 2 - i : r < 5 ? 1 - i : 0;
    if ((F2G_X6LAB & 32) && g != 0) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][mi], fb[ks][j][ni], acc[mi][ni], 0, 0, 0);
  };
  gload(0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) stage_a(j);
#pragma unroll
  for (int j = 0; j < NJW; ++j) stage_w(j);
  gload(nt > 1 ? 1 : 0);
  X6_LDS_BARRIER();
  frags();
  X6_LDS_BARRIER();
#if F2G_X6LAB & 256
  unsigned long long pt0, pt1, pt2;
  unsigned long long ps0 = 0, ps1 = 0, ps2 = 0;
  X6PROF_STAMP(pt0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pt2 = pt0;
  pe1 = pt0;
#endif
  for (int t = 0; t + 1 < nt; ++t) {
    const int t2 = t + 2 < nt ? t + 2 : 0;       // (past the end: re-read, never used)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // quarter q: MFMA groups 3 q ... 3 q + 2 with the staging of activation chunk q (and its share of the
      // weight chunks) between them; the chunk's registers are requested again for slab t + 2 right away
      mf4(3 * q);
      if (WIMG) {
        if (q < 3) {
          stage_w(2 * q);
          stage_w(2 * q + 1);
          xw[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[2 * q], t2 * 192, 0);
          xw[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[2 * q + 1], t2 * 192, 0);
        }
      } else {
        stage_w(q);
        xw[q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[q], t2 * 128, 0);
      }
      mf4(3 * q + 1);
      stage_a(q);
      xa[q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[q], t2 * 128, 0);
      mf4(3 * q + 2);
#if !(F2G_X6LAB & 128)
      // one MFMA, then its share of the quarter's VALU work; the LDS stores and the requests close the quarter
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, WIMG ? 3 : 5, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    X6PROF_STAMP(pt1);
    X6_LDS_BARRIER();
    X6PROF_ACC(ps0, pt1, pt0);
    X6PROF_ACC(ps2, pt0, pt2);      // (the previous iteration's fragment interval)
    X6PROF_STAMP(pt2);
    frags();
    X6_LDS_BARRIER();
    X6PROF_ACC(ps1, pt2, pt1);
    X6PROF_STAMP(pt0);
  }
#if F2G_X6LAB & 256
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  X6PROF_ACC(ps2, pt0, pt2);
  if (tid == 0 && nt > 1) {
    atomicAdd(&g_x6prof[0], ps0);
    atomicAdd(&g_x6prof[1], ps1);
    atomicAdd(&g_x6prof[2], ps2);
    atomicAdd(&g_x6prof[3], (unsigned long long)(nt - 1));
  }
#endif
  X6PROF_STAMP(pe2);
#pragma unroll
  for (int g = 0; g < 12; ++g) mf4(g);
  X6PROF_STAMP(pe3);
  if (wide) {
    // (every fragment read of the main loop lies before its last barrier: a wave that is through its MFMAs
    // may overlay the operand buffers with its private patch)
    X6LAB_EPI x6e::wide_epilogue(d.E, acc, M, N, m0 + wm * 64, n0 + wn * 64, lane, smem6 + wave * x6e::ESZ);
#if F2G_X6LAB & 256
    unsigned long long pe4;
    X6PROF_STAMP(pe4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (tid == 0) {
      atomicAdd(&g_x6prof[4], pe1 - pe0);
      atomicAdd(&g_x6prof[5], pe3 - pe2);
      atomicAdd(&g_x6prof[6], pe4 - pe3);
      atomicAdd(&g_x6prof[7], 1ull);
    }
#endif
    return;
  }
  X6LAB_EPI gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, true);
  if (X6LAB_X3 d.E.x3_out) x3_tile_readback(d.E, M, N, m0, n0, tid);
}

// gemm_x6f_kernel<true> with the WEIGHT fragments straight from memory (round 6).  Measured on the kernel above
// (tools/micro/x6prof.py: s_memtime stamps inside the slab loop): with the staging between the MFMAs a chain of 48
// MFMAs (1536 clocks) takes 2420 clocks alone on its CU, 1800 without the LDS stores -- a store moves its address
// and data registers to the LDS at 2 clocks per source dword (MI355X_MICROARCH.md, LDS), ~600 clocks per slab
// and block, and the wave's MFMAs wait behind it.  Half of those stores put the weight image into LDS only to read
// it back in MFMA fragment order.  Here the cached weight image is FRAGMENT-MAJOR (B.split = 4: [N / 32][K / 32]
// [piece][k step][lane][16 bytes] -- the pieces of f2g_split_bf16x3 in another order: F2G_MULTI_SPLIT3G), so a
// wave's B fragment is one coalesced 1 KB load into the registers the MFMAs read: no LDS store, no LDS read, no
// staging registers for the weights; the two k-step halves of the fragment set are requested again for slab
// t + 1 as soon as the MFMAs of slab t have consumed them.  LDS carries the activation tile only (half the
// stores, half the fragment reads of gemm_x6f_kernel).
__global__ __launch_bounds__(256, 2) void gemm_x6g_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                          const x6_rows R, const int wide) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  constexpr int PITCH = 208, NJ = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(128, 128, m0, n0);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nt = K / 32;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, R.bytes, 0x00020000);
  // (the image ends with the last whole 32-row group: groups of a ragged last tile read zeros)
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)((long long)N * K * 6), 0x00020000);
  unsigned voA[NJ];
  int lo[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = tid + 256 * j, row = id >> 3, c = id & 7;
    const int r = m0 + row, sq = r / R.P0;
    voA[j] = r < M ? (unsigned)sq * R.seq6 + (unsigned)(r - sq * R.P0) * R.step6 + R.off6 + c * 16 : 0xf0000000u;
    lo[j] = row * PITCH + c * 8;
  }
  // fragment (i, piece p, k step ks) of slab t: 1 KB at (((n0 / 32 + 2 wn + i) nt + t) 12 + 4 p + 2 ks) * 512 bytes
  unsigned voB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) voB[i] = (unsigned)((n0 >> 5) + 2 * wn + i) * (unsigned)nt * 6144u + lane * 16;
  u32x4 xa[NJ];
  u32x2 pa[NJ][3];      // the pieces of slab t + 1, split during the chain of slab t, stored behind it
  bf16x8 fa[2][3][2], fb[2][3][2];
  auto load_b = [&](int ks, int t) {        // the six fragments of k step ks
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fb[ks][p][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, voB[i] + (4 * p + 2 * ks) * 512, t * 6144, 0));
  };
  auto stage_a = [&](int j) {
    u32x2 p0, p1, p2;
    split3x4(xa[j], p0, p1, p2);
    *reinterpret_cast<u32x2*>(smem6 + lo[j]) = p0;
    *reinterpret_cast<u32x2*>(smem6 + lo[j] + 64) = p1;
    *reinterpret_cast<u32x2*>(smem6 + lo[j] + 128) = p2;
  };
  const unsigned char* rA = smem6 + (wm * 64 + li) * PITCH + h * 16;
  auto frags_a = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fa[ks][p][i] = *reinterpret_cast<const bf16x8*>(rA + p * 64 + i * 32 * PITCH + ks * 32);
  };
  // the 48 MFMAs of a slab as 12 groups of four (one product term of one k step), smallest terms first
  auto mf4 = [&](int g) {
    const int ks = g / 6, r = g % 6;
    const int i = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : r == 3 ? 0 : r == 4 ? 1 : 0;
    const int j = r < 3 ? 2 - i : r < 5 ? 1 - i : 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][mi], fb[ks][j][ni], acc[mi][ni], 0, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < NJ; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], 0, 0);
  load_b(0, 0);
  load_b(1, 0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) stage_a(j);
#pragma unroll
  for (int j = 0; j < NJ; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], nt > 1 ? 128 : 0, 0);
  X6_LDS_BARRIER();
  frags_a();
  X6_LDS_BARRIER();
  for (int t = 0; t + 1 < nt; ++t) {
    const int t2 = t + 2 < nt ? t + 2 : 0;       // (past the end: re-read, never used)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // quarter q: MFMA groups 3 q ... 3 q + 2 with the staging of activation chunk q of slab t + 1 between them;
      // the chunk's registers are requested again for slab t + 2 right away.  Groups 0-5 are k step 0, 6-11 k step
      // 1: quarter 2 opens with the requests for k step 0 of slab t + 1, quarter 3 closes with those for k step 1.
      if (q == 2) load_b(0, t + 1);
      mf4(3 * q);
      mf4(3 * q + 1);
#if F2G_X6LAB & 1024      // (lab: the stores inside the chain, quarter by quarter)
      stage_a(q);
#else
      split3x4(xa[q], pa[q][0], pa[q][1], pa[q][2]);
#endif
      xa[q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[q], t2 * 128, 0);
      mf4(3 * q + 2);
      if (q == 3) load_b(1, t + 1);
      if (q == 2) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 7, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if !(F2G_X6LAB & 1024)
    // the twelve LDS stores of the slab BEHIND the chain: a store moves its address and data registers to the LDS
    // over the path the MFMAs read their operands through -- between the MFMAs they cost the chain ~50 clocks each
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x2*>(smem6 + lo[q]) = pa[q][0];
      *reinterpret_cast<u32x2*>(smem6 + lo[q] + 64) = pa[q][1];
      *reinterpret_cast<u32x2*>(smem6 + lo[q] + 128) = pa[q][2];
    }
#endif
    X6_LDS_BARRIER();
    frags_a();
    X6_LDS_BARRIER();
  }
#pragma unroll
  for (int g = 0; g < 12; ++g) mf4(g);
  if (wide) {
    x6e::wide_epilogue(d.E, acc, M, N, m0 + wm * 64, n0 + wn * 64, lane, smem6 + wave * x6e::ESZ);
    return;
  }
  gemm_epilogue<2, 2>(d.E, acc, M, N, m0, n0, wm, wn, li, h, true);
  if (d.E.x3_out) x3_tile_readback(d.E, M, N, m0, n0, tid);
}

// The in-kernel-split kernel for N <= 32 output columns (round 6): the data gradients that land on a 32-channel
// map -- the second MPD layer's stride residues, 341376 x 32 x 256 -- ran on the generic fp32 kernel's 128 x 32
// tiles at 50 TFLOP/s (matrix pipe 0.29 busy behind bounds-tested window loads).  Same schedule as
// gemm_x6f_kernel<true> on a 128 x 32 tile: four waves of 32 x 32, the weight slab (32 rows of the cached image)
// stored as it comes, six fragment reads each way per 12 MFMAs, generic epilogue (row maps, masks, column sums).
__global__ __launch_bounds__(256, 2) void gemm_x6n_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                          const x6_rows R) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  constexpr int PITCH = 208, OPER = 128 * PITCH, NJ = 4, NJW = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128, n0 = 0;
  f32x16 acc[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, R.bytes, 0x00020000);
  const unsigned rowbytesW = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)N * rowbytesW, 0x00020000);
  unsigned voA[NJ], voW[NJW];
  int lo[NJ], loW[NJW];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = tid + 256 * j, row = id >> 3, c = id & 7;
    const int r = m0 + row, sq = r / R.P0;
    voA[j] = r < M ? (unsigned)sq * R.seq6 + (unsigned)(r - sq * R.P0) * R.step6 + R.off6 + c * 16 : 0xf0000000u;
    lo[j] = row * PITCH + c * 8;
  }
#pragma unroll
  for (int j = 0; j < NJW; ++j) {
    const int id = tid + 256 * j, row = id / 12, c = id - row * 12;      // 32 rows x 12 chunks = 384 chunks
    voW[j] = (id < 384 && row < N) ? (unsigned)row * rowbytesW + c * 16 : 0xf0000000u;
    loW[j] = id < 384 ? row * PITCH + c * 16 : -1;
  }
  u32x4 xa[NJ], xw[NJW];
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], t * 128, 0);
#pragma unroll
    for (int j = 0; j < NJW; ++j) xw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], t * 192, 0);
  };
  const unsigned char* rA = smem6 + (wave * 32 + li) * PITCH + h * 16;
  const unsigned char* rB = smem6 + OPER + li * PITCH + h * 16;
  const int nt = K / 32;
  gload(0);
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      u32x2 p0, p1, p2;
      split3x4(xa[j], p0, p1, p2);
      *reinterpret_cast<u32x2*>(smem6 + lo[j]) = p0;
      *reinterpret_cast<u32x2*>(smem6 + lo[j] + 64) = p1;
      *reinterpret_cast<u32x2*>(smem6 + lo[j] + 128) = p2;
    }
#pragma unroll
    for (int j = 0; j < NJW; ++j)
      if (loW[j] >= 0) *reinterpret_cast<u32x4*>(smem6 + OPER + loW[j]) = xw[j];
    gload(t + 1 < nt ? t + 1 : 0);       // (past the end: re-read, never used)
    X6_LDS_BARRIER();
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[ks][p] = *reinterpret_cast<const bf16x8*>(rA + p * 64 + ks * 32);
        fb[ks][p] = *reinterpret_cast<const bf16x8*>(rB + p * 64 + ks * 32);
      }
    X6_LDS_BARRIER();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int sdeg = 2; sdeg >= 0; --sdeg)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int j = sdeg - i;
          if (j < 0 || j > 2) continue;
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[0][0], 0, 0, 0);
        }
  }
  // Epilogue.  The row map of a stride residue costs the generic epilogue one 64-bit division per ELEMENT (16 per
  // lane and tile, ~35 VALU instructions each beside 96 MFMAs); here the 128 output-row offsets of the tile are
  // computed once, one division per ROW, into the (now free) operand buffer, and the elements look them up.
  const f2g_epilogue& E = d.E;
  if (E.aux || E.res || E.prelu_slope || E.atomic || E.accumulate || E.scale != 0.f) {
    gemm_epilogue<1, 1>(E, acc, M, N, m0, n0, wave, 0, li, h, true);
    return;
  }
  __syncthreads();                                   // every wave is through its last fragment reads
  long long* rowoff = reinterpret_cast<long long*>(smem6);
  if (tid < 128) {
    const int row = m0 + tid;
    long long off = -1;
    if (row < M) {
      if (E.P0o > 0) {
        const int sq = row / E.P0o;
        off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o + E.off_o;
      } else {
        off = (long long)row * E.ldc;
      }
    }
    rowoff[tid] = off;
  }
  __syncthreads();
  const int col = li;
  if (col >= N) return;
  const float bias = E.bias ? E.bias[col] : 0.f;
  const float fmw = E.fm_ref ? E.fm_w * (E.fm_wdev ? E.fm_wdev[0] : 1.f) : 0.f;
  float cs = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long long ro = rowoff[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    if (ro < 0) continue;
    const long long off = ro + col;
    float v = acc[0][0][e] + bias;
    if (E.lrelu_slope != 0.f) v = v > 0.f ? v : E.lrelu_slope * v;
    if (E.mask_src) {   // leaky-ReLU backward of the layer below (+ feature-matching term)
      const float y = E.mask_src[off];
      if (E.fm_ref) {
        const float dl = y - E.fm_ref[off];
        v += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
      }
      v *= y > 0.f ? 1.f : E.mask_slope;
    }
    cs += v;
    E.C[off] = v;
  }
  if (E.colsum) {
    cs += __shfl_xor(cs, 32);
    if (h == 0) atomicAdd(E.colsum + col, cs);
  }
}

// 1 if a form-0 descriptor over fp32 tensors could run as precision 3 once both operands are handed
// over as f2g_split_bf16x3 images
// extent in elements of what the A operand's rows may touch, or 0 if precision 3 cannot read it
static long long x6_a_extent(const f2g_operand& A) {
  if (host_plain(A)) return (long long)A.rows * A.cols;
  // single-segment windows that never leave their sequence (halo layouts), everything on slab boundaries
  if (A.P1 != 1 || A.L1 != 1 || A.P0 < 1 || A.seglen < A.cols || A.reflect || A.pad0 > 0 || A.rows % A.P0)
    return 0;
  const long long step = (long long)A.step0 * A.unit, off = -(long long)A.pad0 * A.unit;
  if ((step % 32) || (off % 32) || (A.seq_stride % 32) || step < 0) return 0;
  if ((long long)(A.P0 - 1) * step + off + A.cols > A.L0u) return 0;
  return (long long)(A.rows / A.P0 - 1) * A.seq_stride + A.L0u;
}

static bool x6_shape_ok(const f2g_gemm_desc& d) {
  if (d.form != 0 || !host_plain(d.B) || d.A.cols != d.B.cols) return false;
  const long long M = d.A.rows, N = d.B.rows, K = d.A.cols, ext = x6_a_extent(d.A);
  if (K < 32 || (K % 32) || M < 1 || N < 1 || ext <= 0) return false;
  if (ext * 6 >= 0xe0000000ll || N * K * 6 >= 0xe0000000ll) return false;        // 32-bit buffer offsets
  if (d.A.alpha || d.A.lrelu_src || d.B.alpha || d.B.lrelu_src) return false;    // (no on-load transforms)
  if (d.E.c_bf16 || d.E.atomic || d.split_k > 1) return false;
  if (d.E.x3_out) {     // whole 8-element groups of the output on 8-element boundaries of its buffer
    const f2g_epilogue& E = d.E;
    if (E.prelu_out || (((uintptr_t)E.x3_out) & 15) || (((uintptr_t)E.C) & 15) || (N % 8)) return false;
    if (E.P0o > 0 ? ((E.seq_stride_o | E.row_stride_o | E.off_o) & 7) != 0 : (E.ldc & 7) != 0) return false;
  }
  return true;
}

// stride-1 conv windows of TAPS positions x C channels over a halo map image (gemm_x6t_kernel)
static bool x6_tap_ok(const f2g_gemm_desc& d, int taps) {
  const f2g_operand& A = d.A;
  const bool on = f2g_opt(F2G_OPT_X6_TAP) != 0;
  if (!on || host_plain(A) || A.P1 != 1 || A.step0 != 1 || A.unit < 32 || (A.unit % 32)) return false;
  if (A.cols != taps * A.unit || A.seglen < A.cols || (A.seq_stride % A.unit) || (A.pad0 > 0)) return false;
  const int HpIn = (int)(A.seq_stride / A.unit);
  if (A.P0 < 8 || HpIn < A.P0) return false;
  // staged positions of a tile: its rows, the taps' overhang, the extra positions of every sequence end inside
  return 128 + taps - 1 + (HpIn - A.P0) * (128 / A.P0 + 1) <= 160;
}

// 1: the launch takes the wide epilogue (x6_epilogue.h; option x6_wide = 0: the generic one + image read-back)
static int x6_wide(const f2g_gemm_desc& d) {
  return f2g_opt(F2G_OPT_X6_WIDE) != 0 && x6e::wide_ok(d.E, d.B.rows) ? 1 : 0;
}

static int launch_x6(const f2g_gemm_desc& d, hipStream_t st) {
  // round 5: ping-pong wave groups + wide epilogue (gemm_x6p.hip) where its epilogue subset applies
  for (int taps = 5; taps >= 2; taps -= 3)
    if (x6_tap_ok(d, taps) && f2g_x6p_ok(d, taps)) {
      g_last_path = 4;
      return f2g_launch_x6p(d, taps, x6_a_extent(d.A), st);
    }
  const int M = d.A.rows, N = d.B.rows, K = d.A.cols;
  constexpr size_t smem = 2 * 128 * 208;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  x6_rows R;
  if (host_plain(d.A)) {
    R.P0 = 1, R.seq6 = (unsigned)K * 6u, R.step6 = 0, R.off6 = 0;
  } else {
    R.P0 = d.A.P0, R.seq6 = (unsigned)(d.A.seq_stride * 6);
    R.step6 = (unsigned)((long long)d.A.step0 * d.A.unit * 6), R.off6 = (unsigned)(-(long long)d.A.pad0 * d.A.unit * 6);
  }
  R.bytes = (unsigned)(x6_a_extent(d.A) * 6);
  dim3 grid((M + 127) / 128, (N + 127) / 128);
  hipLaunchKernelGGL(gemm_x6_kernel, grid, dim3(256), smem, st, d, M, N, K, R, x6_wide(d));
  g_last_path = 4;
  return f2g_check_launch();
}

// the same descriptor over the fp32 tensors themselves (split = 0): gemm_x6f_kernel
// (B.split = 3: the weight operand as its cached image -- gemm_x6f_kernel<true>; the activation stays fp32)
static bool x6f_ok(const f2g_gemm_desc& d) {
  if (d.A.split || (d.B.split != 0 && d.B.split != 3 && d.B.split != 4) || !x6_shape_ok(d)) return false;
  // (the fragment-major weight image: whole 32-row groups, plain matrix)
  if (d.B.split == 4 && ((d.B.rows & 31) || d.B.rows <= 32 || d.B.P0 != 1 || d.B.P1 != 1)) return false;
  if (!al16(d.A.base) || !al16(d.B.base) || (d.A.seq_stride & 3) || (d.B.split == 0 && (d.B.seq_stride & 3))) return false;
  const long long ext = host_plain(d.A) ? (long long)d.A.rows * d.A.seq_stride : x6_a_extent(d.A);
  return ext * 4 < 0xe0000000ll && (long long)d.B.rows * (d.B.split ? d.B.cols * 6ll : d.B.seq_stride * 4) < 0xe0000000ll;
}

static int launch_x6f(const f2g_gemm_desc& d, hipStream_t st) {
  const int M = d.A.rows, N = d.B.rows, K = d.A.cols;
  constexpr size_t smem = 2 * 128 * 208;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6f_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6f_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  x6_rows R;      // (byte strides of the fp32 tensor)
  if (host_plain(d.A)) {
    R.P0 = 1, R.seq6 = (unsigned)(d.A.seq_stride * 4), R.step6 = 0, R.off6 = 0;
    R.bytes = (unsigned)((long long)d.A.rows * d.A.seq_stride * 4);
  } else {
    R.P0 = d.A.P0, R.seq6 = (unsigned)(d.A.seq_stride * 4);
    R.step6 = (unsigned)((long long)d.A.step0 * d.A.unit * 4), R.off6 = (unsigned)(-(long long)d.A.pad0 * d.A.unit * 4);
    R.bytes = (unsigned)(x6_a_extent(d.A) * 4);
  }
  if (N <= 32 && d.B.split == 3 && !d.E.x3_out) {     // thin outputs: 128 x 32 tiles (gemm_x6n_kernel)
    constexpr size_t smem_n = (128 + 32) * 208;
    hipLaunchKernelGGL(gemm_x6n_kernel, dim3((M + 127) / 128), dim3(256), smem_n, st, d, M, N, K, R);
    g_last_path = 5;      // (its own family in the benchmark's tables: a different kernel on 128 x 32 tiles)
    return f2g_check_launch();
  }
  dim3 grid((M + 127) / 128, (N + 127) / 128);
  if (d.B.split == 4) {
    constexpr size_t smem_g = 4 * x6e::ESZ > 128 * 208 ? 4 * x6e::ESZ : 128 * 208;
    hipLaunchKernelGGL(gemm_x6g_kernel, grid, dim3(256), smem_g, st, d, M, N, K, R, x6_wide(d));
    g_last_path = 4;
    return f2g_check_launch();
  }
  if (d.B.split == 3) hipLaunchKernelGGL(gemm_x6f_kernel<true>, grid, dim3(256), smem, st, d, M, N, K, R, x6_wide(d));
  else hipLaunchKernelGGL(gemm_x6f_kernel<false>, grid, dim3(256), smem, st, d, M, N, K, R, x6_wide(d));
  g_last_path = 4;
  return f2g_check_launch();
}

// How f2g_gemm would run this form-0 descriptor at precision 3 -- the SAME tests as its dispatch, E.x3_out
// included (set it before asking).  Bits: 1 = over three-piece images of both operands (split = 3; also
// reported for the fp32 tensors the images would be made of), 2 = and then on a tap-walking instance
// (stride-1 conv windows of 5 or 2 positions), 4 = over the fp32 operands as they are handed over
// (gemm_x6f_kernel: alignment and stride conditions of the in-kernel split).  0 = not at precision 3.
extern "C" int f2g_gemm_x6_ok(const f2g_gemm_desc* d) {
  if (!d || !x6_shape_ok(*d)) return 0;
  return 1 | ((x6_tap_ok(*d, 5) || x6_tap_ok(*d, 2)) ? 2 : 0) | (x6f_ok(*d) ? 4 : 0);
}

// Rows of the partial column-sum matrices (E.colsum_part_ld > 0) the launch of `d` writes: one per 64 output
// rows of whole tiles -- or 0 when the kernel f2g_gemm's dispatch picks has no wide epilogue (the SAME tests
// in the same order as launch_x6f / launch_x6).
extern "C" int32_t f2g_gemm_colsum_part_rows(const f2g_gemm_desc* dp) {
  if (!dp || dp->precision != 3 || dp->form != 0) return 0;
  f2g_gemm_desc d = *dp;
  if (d.E.colsum_part_ld <= 0) d.E.colsum_part_ld = 4;          // (alignment of the pointers is the caller's)
  if (!x6_wide(d)) return 0;
  const int M = d.A.rows;
  const int rows128 = 2 * ((M + 127) / 128), rows256 = 4 * ((M + 255) / 256);
  if (x6f_ok(d)) return (d.B.rows <= 32 && d.B.split == 3 && !d.E.x3_out) ? 0 : rows128;   // (gemm_x6n_kernel: generic epilogue)
  if (d.A.split != 3 || d.B.split != 3 || !x6_shape_ok(d)) return 0;
  for (int taps = 5; taps >= 2; taps -= 3)
    if (x6_tap_ok(d, taps) && f2g_x6p_ok(d, taps)) return rows256;
  return rows128;
}

extern "C" int64_t f2g_split_bf16x3_bytes(int32_t rows, int32_t K) { return (int64_t)rows * K * 6; }

extern "C" int f2g_split_bf16x3(void* dst, const float* src, int64_t ld, int32_t rows, int32_t K,
                                f2g_stream_t stream) {
  if (!dst || !src || rows < 0 || K < 32 || (K % 32) || ld < K || (ld & 3) || (((uintptr_t)src) & 15) ||
      (((uintptr_t)dst) & 15))
    return F2G_EINVAL;
  if (rows == 0) return F2G_OK;
  const long long total = (long long)rows * (K / 4);
  hipLaunchKernelGGL(split3_img_kernel, dim3(f2g_grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<__bf16*>(dst), src, (long long)ld, (long long)rows, K);
  return f2g_check_launch();
}

extern "C" int f2g_gemm_last_path(void) { return g_last_path; }

// Would f2g_gemm run this form-0 descriptor on the lean kernel (whatever its precision)?  The host
// asks before it pre-splits the operands of a split-bf16 GEMM.
// 1 if f2g_gemm would run this form-2 descriptor (exact fp32, E.atomic, split_k as set) on the K-major lean
// weight-gradient kernel -- two blocks per CU, so the host deals its blocks in rounds of 512 (ops.split_for)
extern "C" int f2g_gemm_wgrad_lean(const f2g_gemm_desc* dp) {
  if (!dp || !dp->A.base || !dp->B.base) return 0;
  return leanw_fp32_takes(*dp, dp->split_k < 1 ? 1 : dp->split_k) ? 1 : 0;
}

extern "C" int f2g_gemm_lean_ok(const f2g_gemm_desc* dp) {
  if (!dp || !dp->A.base || !dp->B.base) return 0;
  const bool lean_on = f2g_opt(F2G_OPT_LEAN) != 0;
  const f2g_gemm_desc& d = *dp;
  if (d.form == 2) return lean_on && leanw_ok(d) ? 1 : 0;   // split-bf16 weight-gradient kernel
  if (d.form != 0) return 0;
  if (d.A.cols != d.B.cols || !host_plain(d.B)) return 0;
  if (!(lean_on && d.B.rows > 64 && lean_a_ok(d.A) && lean_b_ok(d.B))) return 0;
  return lean_bf16_ok(d.A, d.B) ? 3 : 1;   // bit 1: also as true bf16 tensors (split = 2)
}

// dst = split-bf16 image of src (n4 groups of four floats, both 16-byte aligned): group g becomes
// [hi(x0) hi(x1) hi(x2) hi(x3) | lo(x0) lo(x1) lo(x2) lo(x3)], hi = bf16(x) (round to nearest even),
// lo = bf16(x - hi): the same 16 bytes at the same address, so every operand descriptor (windows,
// halo rows, strides -- all multiples of four floats on the lean path) addresses it unchanged.
__global__ __launch_bounds__(256) void split_bf16_kernel(uint4* __restrict__ dst,
                                                         const float4* __restrict__ src, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = src[i];
    unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
    split_bf16(v.x, h0, l0);
    split_bf16(v.y, h1, l1);
    split_bf16(v.z, h2, l2);
    split_bf16(v.w, h3, l3);
    dst[i] = make_uint4(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16), l0 | ((unsigned)l1 << 16),
                        l2 | ((unsigned)l3 << 16));
  }
}

// dst (bf16, n elements) = round-to-nearest-even of src: the TRUE bf16 image of a tensor (same shape,
// strides in elements) for the plain-bf16 lean instances
__global__ __launch_bounds__(256) void to_bf16_kernel(uint2* __restrict__ dst, const float4* __restrict__ src,
                                                      long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = src[i];
    unsigned short h0, h1, h2, h3, l;
    split_bf16(v.x, h0, l);
    split_bf16(v.y, h1, l);
    split_bf16(v.z, h2, l);
    split_bf16(v.w, h3, l);
    dst[i] = make_uint2(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16));
  }
}

extern "C" int f2g_to_bf16(void* dst, const float* src, int64_t n, f2g_stream_t stream) {
  if (!dst || !src || (n & 3) || (((uintptr_t)dst) & 7) || !al16(src)) return F2G_EINVAL;
  if (n == 0) return F2G_OK;
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint2*>(dst), reinterpret_cast<const float4*>(src), n4);
  return f2g_check_launch();
}

extern "C" int f2g_split_bf16(float* dst, const float* src, int64_t n, f2g_stream_t stream) {
  if (!dst || !src || (n & 3) || !al16(dst) || !al16(src)) return F2G_EINVAL;
  if (n == 0) return F2G_OK;
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint4*>(dst), reinterpret_cast<const float4*>(src), n4);
  return f2g_check_launch();
}

extern "C" int f2g_gemm(const f2g_gemm_desc* dp, f2g_stream_t stream) {
  if (!dp || !dp->A.base || !dp->B.base || !dp->E.C) return F2G_EINVAL;
  const f2g_gemm_desc& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  int split = d.split_k > 0 ? d.split_k : 1;
  if (d.E.colsum_part_ld > 0 && f2g_gemm_colsum_part_rows(dp) == 0) {
    f2g_set_error("f2g_gemm: E.colsum_part_ld needs a precision-3 launch with the wide epilogue "
                  "(f2g_gemm_colsum_part_rows(d) == 0 for this descriptor)");
    return F2G_EINVAL;
  }
  if (d.precision == 3 && d.form == 2) {
    // fp32-class weight gradient: fp32 operands, split into three pieces inside the kernel
    if (d.A.split || d.B.split || d.A.rows != d.B.rows || !leanw_ok(d) || d.E.x3_out ||
        (split > 1 && !d.E.atomic)) {
      f2g_set_error("f2g_gemm precision 3, form 2: fp32 operands the K-major weight-gradient kernel reads");
      return F2G_EINVAL;
    }
    return launch_leanw6(d, d.A.cols, d.B.cols, d.A.rows, split, st);
  }
  if (d.precision == 3) {
    // fp32-class products from three-piece images (both operands f2g_split_bf16x3 images: split = 3)
    if (x6f_ok(d)) return launch_x6f(d, st);       // fp32 operands, split inside the kernel
    if (d.A.split != 3 || d.B.split != 3 || !x6_shape_ok(d)) {
      f2g_set_error("f2g_gemm precision 3: form 0 over plain f2g_split_bf16x3 images (split = 3), K % 32 == 0");
      return F2G_EINVAL;
    }
    return launch_x6(d, st);
  }
  if (d.E.x3_out) {
    f2g_set_error("f2g_gemm: E.x3_out belongs to precision 3");
    return F2G_EINVAL;
  }
  {
    const int nr = f2g_gemm_narrow(d, st);  // <= 4 output columns / gradient rows: VALU kernels
    g_last_path = nr != 0 ? 3 : 0;
    if (nr != 0) return nr < 0 ? nr : F2G_OK;
  }
  if (d.E.prelu_slope && (d.E.atomic || d.E.accumulate || d.E.P0o > 0 || d.form == 2)) return F2G_EINVAL;
  if (d.E.mask_src && (d.E.atomic || d.E.accumulate || d.form == 2)) return F2G_EINVAL;
  if (d.form == 0 || d.form == 1) {
    const bool f1 = d.form == 1;
    if (f1 ? d.A.cols != d.B.rows : d.A.cols != d.B.cols) return F2G_EINVAL;
    if (!host_plain(d.B)) return F2G_EINVAL;
    const int M = d.A.rows, N = f1 ? d.B.cols : d.B.rows, K = d.A.cols;
    const int am = op_mode(d.A, true), bm = op_mode(d.B, !f1);
    // split_k: 1 = off, > 1 = as asked, 0 = decide here (linear epilogues only)
    // (an epilogue input that aliases the output -- in-place residual or PReLU-derivative mask --
    // would be destroyed by the zero fill)
    const bool linear = d.E.lrelu_slope == 0.f && d.E.res != d.E.C && d.E.aux != d.E.C && !d.E.prelu_slope &&
                        !d.E.mask_src;
    int s = d.split_k;
    // (STFT framing GEMMs are never split: atomics would make the spectra -- the input of every
    // discriminator and loss -- differ in the last bit from run to run)
    // option deterministic = 1 (F2G_DETERMINISTIC=1): never split on the library's own initiative (bit-reproducible forward)
    const bool no_auto = f2g_opt(F2G_OPT_DETERMINISTIC) != 0;
    const bool lean_on = f2g_opt(F2G_OPT_LEAN) != 0;
    // pre-split operands (f2g_split_bf16) are understood by the lean kernel's split-bf16 instances only
    const bool presplit = d.A.split != 0 && d.B.split == d.A.split;
    const bool bf16img = d.A.split == 2;
    const bool lean = !f1 && lean_on && N > 64 && lean_a_ok(d.A) && lean_b_ok(d.B) &&
                      (d.precision == 0 || ((d.precision == 1 || d.precision == 2) && presplit)) &&
                      (!bf16img || (d.precision == 2 && lean_bf16_ok(d.A, d.B)));
    if ((d.A.split || d.B.split) && !(lean && d.precision != 0)) return F2G_EINVAL;
    if (d.E.c_bf16 && !lean) return F2G_EINVAL;
    if (lean && d.split_k == 0) {
      // library-chosen work split on the lean kernel: stream-K (same linear-epilogue condition as
      // split-K; option deterministic = 1 (F2G_DETERMINISTIC=1) keeps the plain tile grid)
      const int sk_mode = f2g_opt(F2G_OPT_STREAMK);
      int upb = 0;
      if (sk_mode > 0 && linear && !no_auto && !d.E.atomic && !d.E.c_bf16)
        upb = lean_stream_k(M, N, bf16img ? K / 2 : K, sk_mode > 1);   // (64-element slabs)
      if (upb > 0 && !d.E.accumulate)
        hipLaunchKernelGGL(zero_out_kernel, dim3(f2g_grid_for((int64_t)M * N, 256)), dim3(256), 0,
                           st, d.E, M, N);
      return launch_lean(d, M, N, K, 1, upb, st);
    }
    if (s == 0)
      s = (linear && am != SL && bm != SL && M > 0 && !d.A.reflect && !no_auto) ? auto_split(M, N, K)
                                                                                 : 1;
    if (s > 1 && !linear) return F2G_EINVAL;
    f2g_gemm_desc dd = d;
    if (s > 1 && !d.E.atomic) {
      if (!d.E.accumulate) {
        hipLaunchKernelGGL(zero_out_kernel, dim3(f2g_grid_for((int64_t)M * N, 256)), dim3(256), 0,
                           st, d.E, M, N);
      }
      dd.E.atomic = 1;
      dd.E.accumulate = 0;
    }
    if (!f1) {
      if (lean) return launch_lean(dd, M, N, K, s, 0, st);
      if (am == PF && bm == PF) return dispatch_tile<false, false, PF, PF>(dd, M, N, K, s, st);
      if (am == GF && bm == PF) return dispatch_tile<false, false, GF, PF>(dd, M, N, K, s, st);
      if (am == GR && bm == PF) return dispatch_tile<false, false, GR, PF>(dd, M, N, K, s, st);
      return dispatch_tile<false, false, SL, SL>(dd, M, N, K, s, st);
    }
    if (am == PF && bm == PF) return dispatch_tile<false, true, PF, PF>(dd, M, N, K, s, st);
    if (am == GF && bm == PF) return dispatch_tile<false, true, GF, PF>(dd, M, N, K, s, st);
    return dispatch_tile<false, true, SL, SL>(dd, M, N, K, s, st);
  } else if (d.form == 2) {
    if (d.A.rows != d.B.rows) return F2G_EINVAL;
    if (split > 1 && !d.E.atomic) return F2G_EINVAL;
    const int M = d.A.cols, N = d.B.cols, K = d.A.rows;
    if (d.A.split || d.B.split) {   // pre-split images: the lean weight-gradient kernel only
      const bool lean_on = f2g_opt(F2G_OPT_LEAN) != 0;
      if (!(d.precision == 1 && d.A.split && d.B.split && lean_on && leanw_ok(d))) return F2G_EINVAL;
      return launch_leanw3(d, M, N, K, split, st);
    }
    if (leanw_fp32_takes(d, split)) return launch_leanw(d, M, N, K, split, st);
    int am = op_mode(d.A, false), bm = op_mode(d.B, false);
    // split-K chunks are multiples of BK, so PF only needs the total extent % BK == 0
    if (am == PF && bm == PF) return dispatch_tile<true, true, PF, PF>(d, M, N, K, split, st);
    if (am == PF && bm == GF) return dispatch_tile<true, true, PF, GF>(d, M, N, K, split, st);
    if (am != SL && bm != SL) return dispatch_tile<true, true, GF, GF>(d, M, N, K, split, st);
    return dispatch_tile<true, true, SL, SL>(d, M, N, K, split, st);
  }
  return F2G_EINVAL;
}
