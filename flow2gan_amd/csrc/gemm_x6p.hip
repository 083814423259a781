// fp32-class GEMM over stride-1 conv windows of a halo-map image (the 1024-channel MPD layer, its data
// gradient and the residue data gradients of the stride-3 layers; reference discriminators.py:65-76) --
// round 5 successor of the single-group tap-walking kernels of rounds 3-4 (removed in round 6): a 256 x 128 tile, operand images and K order
// (channel slab outer, tap inner; six v_mfma_f32_32x32x16_bf16 with i + j <= 2 per product, smallest terms
// first), but
//
//  * PING-PONG wave groups.  PMC on that predecessor (profiles/r05_pmc_busy_x6_before.txt): matrix pipe busy 56 % of
//    the kernel's life at K = 5120, 29 % at K = 2048 -- its eight waves run in lockstep (all read their 24
//    fragments, barrier, all issue their 48 MFMAs, barrier), so the pipe idles through every read phase.  Here
//    the block's waves form two groups of four (one wave per SIMD each): group 0 owns rows 0..127 of the tile,
//    group 1 rows 128..255, and group 1 runs HALF A STEP BEHIND: while one group's MFMAs hold the matrix pipe,
//    the other reads its fragments, stores its share of the next weight slab and requests the one after.  A
//    group stages the map positions of its OWN 128 rows (so nobody else reads them: they are replaced at the
//    start of the group's MFMA phase of a channel slab's last tap), the weight slabs stay double-buffered and
//    shared: the slab of step s + 1 is stored by group 0 in slot 2s and by group 1 in slot 2s + 1, after the
//    last reader of step s - 1 (group 1, slot 2s - 1) and before the first of step s + 1 (group 0, slot 2s + 2).
//    Barriers wait for the LDS only (s_waitcnt lgkmcnt(0); s_barrier): global loads fly across them.
//  * a WIDE epilogue without read-back.  The generic epilogue stores 4 bytes per lane and instruction behind
//    one integer division per element, and the result's three-piece image was made by reading the tile back
//    from L2 after a block barrier.  Here a wave turns its 64 x 64 accumulator tile through a private LDS
//    patch (32 rows at a time) into 8-column row segments: bias / leaky ReLU / the leaky-ReLU backward mask of
//    the layer below (+ feature-matching term) with 16-byte loads, 16-byte stores of the fp32 map AND of its
//    image pieces from the same registers, column sums (the bias gradient) reduced over the wave's rows
//    before the atomics; one division per ROW (x6_epilogue.h, shared with gemm.hip's six-product kernels).
#include <stdlib.h>

#include "common.h"
#include "x6_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int PITCH = 208;                 // bytes of a staged position / weight row: 3 x 64 + 16 (52 dwords)
constexpr int LH = 160;                    // staged positions of a group's 128 rows (host check)
constexpr int OPERA = 2 * LH * PITCH;      // both groups' positions
constexpr int OPERB = 128 * PITCH;         // one weight slab
using x6e::ESZ;

struct x6p_tap {
  int P0, HpIn, offpos, C32;
  unsigned bytes;
};

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic of this wave done, then the block barrier; outstanding global loads keep flying
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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

template <int TAPS>
__global__ __launch_bounds__(512, 1) void gemm_x6p_kernel(const f2g_gemm_desc d, int M, int N, int K,
                                                          const x6p_tap R) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  constexpr int NJA = (LH * 12 + 255) / 256, NJB = 128 * 12 / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, gt = tid & 255;
  const int wm = (wave >> 1) & 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
  int m0, n0;
  tile_of_block(256, 128, m0, n0);
  const int mg = m0 + 128 * grp;                  // first row of this group's half of the tile
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto posrow = [&](int r) {
    const int sq = r / R.P0;
    return sq * R.HpIn + (r - sq * R.P0) + R.offpos;
  };
  const int pbase = posrow(mg);
  const int rlast = mg + 127 < M ? mg + 127 : M - 1;
  const int L = mg < M ? posrow(rlast) - pbase + TAPS : 0;     // staged positions (<= LH: host check)
  const unsigned rowbytesA = (unsigned)R.C32 * 192u;           // one position of the map image
  const unsigned rowbytesW = (unsigned)(K / 32) * 192u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A.base, 0, R.bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.B.base, 0, (unsigned)N * rowbytesW, 0x00020000);
  unsigned char* myA = smem6 + grp * (LH * PITCH);
  unsigned voA[NJA], voW[NJB];
  int loA[NJA], loW[NJB];
#pragma unroll
  for (int j = 0; j < NJA; ++j) {
    const int id = gt + 256 * j, q = id / 12, c = id - q * 12;
    voA[j] = q < L ? (unsigned)(pbase + q) * rowbytesA + c * 16 : 0xf0000000u;   // (outside the resource: zeros)
    loA[j] = q < LH ? q * PITCH + c * 16 : -1;
  }
#pragma unroll
  for (int j = 0; j < NJB; ++j) {
    const int id = tid + 512 * j, row = id / 12, c = id - row * 12;
    voW[j] = (unsigned)(n0 + row) * rowbytesW + c * 16;
    loW[j] = OPERA + row * PITCH + c * 16;
  }
  u32x4 xa[NJA], xw[NJB];
  auto gloadA = [&](int cs) {
#pragma unroll
    for (int j = 0; j < NJA; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j], cs * 192, 0);
  };
  auto gloadB = [&](int slab) {
#pragma unroll
    for (int j = 0; j < NJB; ++j) xw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW[j], slab * 192, 0);
  };
  auto storeA = [&]() {
#pragma unroll
    for (int j = 0; j < NJA; ++j)
      if (loA[j] >= 0) *reinterpret_cast<u32x4*>(myA + loA[j]) = xa[j];
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NJB; ++j) *reinterpret_cast<u32x4*>(smem6 + buf * OPERB + loW[j]) = xw[j];
  };
  // fragment rows of this lane: output rows mg + wm * 64 + i * 32 + li -> staged position of the group
  const unsigned char* rA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = mg + wm * 64 + i * 32 + li;
    rA[i] = myA + (r < M ? posrow(r) - pbase : 0) * PITCH + h * 16;
  }
  const unsigned char* rB = smem6 + OPERA + (wn * 64 + li) * PITCH + h * 16;
  // weight slab of step s = (cs, t): K order = channel slab outer, tap inner -> image slab t * C32 + cs
  const int nsteps = R.C32 * TAPS;
  auto slab_of = [&](int s) {
    if (s >= nsteps) s = 0;                      // (past the end: re-read, never used)
    const int c2 = s / TAPS, t2 = s - c2 * TAPS;
    return t2 * R.C32 + c2;
  };
  gloadA(0);
  gloadB(slab_of(0));
  storeA();
  storeB(0);
  gloadA(1 < R.C32 ? 1 : 0);
  gloadB(slab_of(1));
  lds_barrier();
  if (grp == 1) lds_barrier();                   // group 1 runs one slot behind
  int step = 0;
  for (int cs = 0; cs < R.C32; ++cs) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t, ++step) {
      const int buf = step & 1;
      // ---- read slot: this step's fragments; my share of the next weight slab; request the one after
      bf16x8 fa[2][3][2], fb[2][3][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[ks][p][i] = *reinterpret_cast<const bf16x8*>(rA[i] + t * PITCH + p * 64 + ks * 32);
            fb[ks][p][i] = *reinterpret_cast<const bf16x8*>(rB + buf * OPERB + p * 64 + i * 32 * PITCH + ks * 32);
          }
      storeB(buf ^ 1);
      gloadB(slab_of(step + 2));
      lds_barrier();
      // ---- MFMA slot (the other group reads meanwhile)
      if (t == TAPS - 1 && cs + 1 < R.C32) {
        // last tap of this channel slab: the group's staged positions are replaced (their only readers are
        // this group's waves, whose reads were complete before the barrier above)
        storeA();
        gloadA(cs + 2 < R.C32 ? cs + 2 : 0);
      }
      __builtin_amdgcn_s_setprio(1);
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
      // (group 1's last MFMA slot needs no barrier behind it: group 0 is in its epilogue by then, and that
      // slot touches no LDS -- both groups pass 1 + 2 * nsteps barriers)
      if (!(grp == 1 && step == nsteps - 1)) lds_barrier();
    }
  }
  // every fragment read of the main loop is complete (the last ones were group 1's, before the barrier group 0
  // has just passed): the waves turn their tiles through private patches at the bottom of the LDS
  x6e::wide_epilogue(d.E, acc, M, N, mg + wm * 64, n0 + wn * 64, lane, smem6 + wave * ESZ);
}


// (A ping-pong instance over ROW operands -- gemm_x6pr_kernel, 256 x 128 tiles for plain matrices and strided
// windows -- was measured no faster than two free-running 128 x 128 blocks in round 5, 188 : 190 and 147 : 149
// TFLOP/s equivalent, and removed in round 6.)

// three bf16 pieces of four floats (common.h: f2g_split3_pair, round to nearest even at every step)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x4(const u32x4& v, u32x2& p0, u32x2& p1, u32x2& p2) {
  // (by value first: __builtin_bit_cast applied to a vector-element expression reads element 0)
  const unsigned u0 = v.x, u1 = v.y, u2 = v.z, u3 = v.w;
  unsigned a0, a1, a2, b0, b1, b2;
  f2g_split3_pair(__uint_as_float(u0), __uint_as_float(u1), a0, a1, a2);
  f2g_split3_pair(__uint_as_float(u2), __uint_as_float(u3), b0, b1, b2);
  p0 = u32x2{a0, b0};
  p1 = u32x2{a1, b1};
  p2 = u32x2{a2, b2};
}

// ---- tap-walking weight gradient of a stride-1 conv layer over halo maps (round 5) ----------------------
// gw[co][t][ci] += sum_r g[r][co] * x[r + t - pad][ci]  (the 1024-channel MPD layer: t = 0..4; reference
// discriminators.py:65-76 backward).  gemm_leanw6_kernel gives every (tap, 128 ci) column tile its own block:
// the same 32 gradient rows are split into their three bf16 pieces by 40 blocks, the same map rows by 8 x 5
// -- 7.9 VALU instructions per MFMA, matrix pipe 50 % busy.  Here a block owns 128 co x 128 ci x ALL taps: a
// slab stages 32 gradient rows and the 32 + TAPS - 1 map rows they touch ONCE (split once, K-major bf16
// planes with gemm_leanw6_kernel's swizzle, double-buffered: one barrier per slab), and the taps walk over the
// staged map rows by shifting the transposing fragment reads (ds_read_b64_tr_b16) one row down.  8 waves =
// (co half) x (ci quarter), 2 x TAPS accumulator tiles each; 120 MFMAs per wave and slab behind 15 LDS stores
// per thread (the old kernel: 48 behind 24).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * 256));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int TAPS>
__global__ __launch_bounds__(512, 1) void gemm_leanw6t_kernel(const f2g_gemm_desc d, int K, int kchunk,
                                                              long long xrows) {
  constexpr int XR = 32 + TAPS - 1;              // staged map rows: the slab's 32 + the taps' overhang
  constexpr int GPL = 32 * 256, XPL = XR * 256;  // bytes of one piece plane: gradient rows / map rows
  constexpr int BUFB = 3 * GPL + 3 * XPL;        // one stage: [g p0 p1 p2 | x p0 p1 p2]
  constexpr int NXQ = (XR * 32 + 511) / 512;     // map chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wq = wave >> 1, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128, c0 = blockIdx.y * 128;     // co tile, ci tile
  const int kbeg = blockIdx.z * kchunk;
  int kend = kbeg + kchunk;
  if (kend > K) kend = K;
  const int nt = (kend - kbeg + 31) / 32;
  if (nt <= 0) return;
  const int Cin = d.B.unit, pad = d.B.pad0;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.A.base, 0, (unsigned)((long long)K * d.A.seq_stride * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.B.base, 0, (unsigned)(xrows * Cin * 4), 0x00020000);
  // staging: chunk id = tid + 512 q -> (row of the slab, 16-byte chunk c of the 128-wide row)
  const int cc = tid & 31, r0 = tid >> 5;         // rows r0 + 16 q
  unsigned offG[2];
  int wofG[2], wofX[NXQ];
  const unsigned rowX = (unsigned)Cin * 4u, colX = (unsigned)(c0 + 4 * cc) * 4u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = r0 + 16 * q;
    offG[q] = (unsigned)(((long long)r * d.A.seq_stride + m0 + 4 * cc) * 4);
    wofG[q] = r * 256 + ((((cc >> 3) ^ (r & 3))) << 6) + (cc & 7) * 8;
  }
#pragma unroll
  for (int q = 0; q < NXQ; ++q) {
    const int r = r0 + 16 * q;
    wofX[q] = r < XR ? 3 * GPL + r * 256 + ((((cc >> 3) ^ (r & 3))) << 6) + (cc & 7) * 8 : -1;
  }
  const int stepG = (int)(32 * d.A.seq_stride * 4);
  // transposed-read roles (gemm_leanw6_kernel): 16-lane group g4, lane i16 -> row (g4 >> 1) * 8 + (i16 >> 2)
  // (+4 for the second half), 8 bytes at (g4 & 1) * 32 + (i16 & 3) * 8 of the tile's 64-byte column block
  const int g4 = lane >> 4, i16 = lane & 15;
  const int rrow = (g4 >> 1) * 8 + (i16 >> 2), within = (g4 & 1) * 32 + (i16 & 3) * 8;
  int rofA[2], rofB[TAPS];
#pragma unroll
  for (int t = 0; t < 2; ++t) rofA[t] = rrow * 256 + ((((wm * 2 + t) ^ (rrow & 3))) << 6) + within;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) rofB[t] = 3 * GPL + (rrow + t) * 256 + (((wq ^ ((rrow + t) & 3))) << 6) + within;
  f32x16 acc[2][TAPS];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][t][e] = 0.f;
  u32x4 xg[2], xx[NXQ];
  auto gload = [&](int s) {       // slab s of this block's chunk (rows kbeg + 32 s ..); past the end: zeros
    const int kr = kbeg + 32 * s;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = kr + r0 + 16 * q;
      xg[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (s < nt && r < kend) ? offG[q] : 0x80000000u, kr * (stepG / 32), 0);
    }
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
      const long long xr = (long long)kr - pad + r0 + 16 * q;        // flat map row of staged row r0 + 16 q
      const bool ok = s < nt && wofX[q] >= 0 && xr >= 0 && xr < xrows;
      // (the whole offset in the vector register: kr - pad is negative for the first slab, and the scalar
      // offset of a buffer load takes no part in the range check)
      xx[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? (unsigned)(xr * rowX) + colX : 0x80000000u, 0, 0);
    }
  };
  auto store = [&](unsigned char* buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x2 p0, p1, p2;
      split3x4(xg[q], p0, p1, p2);
      *reinterpret_cast<u32x2*>(buf + wofG[q]) = p0;
      *reinterpret_cast<u32x2*>(buf + GPL + wofG[q]) = p1;
      *reinterpret_cast<u32x2*>(buf + 2 * GPL + wofG[q]) = p2;
    }
#pragma unroll
    for (int q = 0; q < NXQ; ++q)
      if (wofX[q] >= 0) {
        u32x2 p0, p1, p2;
        split3x4(xx[q], p0, p1, p2);
        *reinterpret_cast<u32x2*>(buf + wofX[q]) = p0;
        *reinterpret_cast<u32x2*>(buf + XPL + wofX[q]) = p1;
        *reinterpret_cast<u32x2*>(buf + 2 * XPL + wofX[q]) = p2;
      }
  };
  gload(0);
  store(smw);
  gload(1);
  lds_barrier();
  for (int s = 0; s < nt; ++s) {
    unsigned char* cur = smw + (s & 1) * BUFB;
    // the next slab goes into the other stage (its readers left through the barrier that closed slab s - 1)
    store(smw + ((s + 1) & 1) * BUFB);
    gload(s + 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[3][2];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[pc][mi] = tr_frag(cur + ks * 16 * 256 + pc * GPL + rofA[mi]);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        bf16x8 fb[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) fb[pc] = tr_frag(cur + ks * 16 * 256 + pc * XPL + rofB[t]);
#pragma unroll
        for (int sdeg = 2; sdeg >= 0; --sdeg)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int j = sdeg - i;
            if (j < 0 || j > 2) continue;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[j], acc[mi][t], 0, 0, 0);
          }
      }
    }
    lds_barrier();
  }
  // gw[co][t * Cin + ci] += acc: rows m0 + wm * 64 + mi * 32 + (MFMA row), columns t * Cin + c0 + wq * 32 + li
  float* C0 = d.E.C;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        atomicAdd(C0 + (long long)row * d.E.ldc + t * Cin + c0 + wq * 32 + li, acc[mi][t][e]);
      }
}


// ---- the same for STRIDED layers (the stride-3 MPD layers: x row = STRIDE * p + t - pad inside a sequence) --
// Consecutive reduction rows read map rows STRIDE apart and the flat row index is no longer one affine function
// over the whole buffer (a sequence has Hp gradient rows but HpIn != STRIDE * Hp map rows), so the block walks
// its sequences one by one in slabs of 16 gradient rows (one MFMA k step; 16 rather than 32 keeps the waste of
// a sequence's last slab at ~5 % over the five periods) and stages the 15 * STRIDE + TAPS map rows they touch:
// every map row is still loaded and split once for all taps.  Fragment row of reduction index i and tap t =
// staged row STRIDE * i + t (the transposing read takes each lane's own address).
template <int TAPS, int STRIDE>
__global__ __launch_bounds__(512, 1) void gemm_leanw6s_kernel(const f2g_gemm_desc d, int nseq, int spb,
                                                              long long xrows) {
  constexpr int XR = 15 * STRIDE + TAPS;         // staged map rows of a 16-row slab
  constexpr int GPL = 16 * 256, XPL = XR * 256;
  constexpr int BUFB = 3 * GPL + 3 * XPL;
  constexpr int NXQ = (XR * 32 + 511) / 512;
  static_assert((4 * STRIDE) % 4 == 0 && (16 * STRIDE) % 4 == 0, "the swizzle term must not depend on the half / k step");
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wq = wave >> 1, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  const int Cin = d.B.unit, pad = d.B.pad0, Hp = d.B.P0;
  const int HpIn = (int)(d.B.seq_stride / Cin);
  const int sq0 = blockIdx.z * spb;
  int sq1 = sq0 + spb;
  if (sq1 > nseq) sq1 = nseq;
  const int nslab = (Hp + 15) / 16;
  const int nt = (sq1 - sq0) * nslab;
  if (nt <= 0) return;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.A.base, 0, (unsigned)((long long)nseq * Hp * d.A.seq_stride * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.B.base, 0, (unsigned)(xrows * Cin * 4), 0x00020000);
  const int cc = tid & 31, r0 = tid >> 5;         // gradient row r0 of the slab; map rows r0 + 16 q
  const unsigned rowG = (unsigned)(d.A.seq_stride * 4), colG = (unsigned)(m0 + 4 * cc) * 4u;
  const unsigned rowX = (unsigned)Cin * 4u, colX = (unsigned)(c0 + 4 * cc) * 4u;
  const int wofG = r0 * 256 + ((((cc >> 3) ^ (r0 & 3))) << 6) + (cc & 7) * 8;
  int wofX[NXQ];
#pragma unroll
  for (int q = 0; q < NXQ; ++q) {
    const int r = r0 + 16 * q;
    wofX[q] = r < XR ? 3 * GPL + r * 256 + ((((cc >> 3) ^ (r & 3))) << 6) + (cc & 7) * 8 : -1;
  }
  const int g4 = lane >> 4, i16 = lane & 15;
  const int rrow = (g4 >> 1) * 8 + (i16 >> 2), within = (g4 & 1) * 32 + (i16 & 3) * 8;
  int rofA[2], rofB[TAPS];
#pragma unroll
  for (int t = 0; t < 2; ++t) rofA[t] = rrow * 256 + ((((wm * 2 + t) ^ (rrow & 3))) << 6) + within;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int xr = STRIDE * rrow + t;
    rofB[t] = 3 * GPL + xr * 256 + (((wq ^ (xr & 3))) << 6) + within;
  }
  f32x16 acc[2][TAPS];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][t][e] = 0.f;
  u32x4 xg, xx[NXQ];
  auto gload = [&](int j) {       // slab j of this block: sequence sq0 + j / nslab, gradient rows 16 (j % nslab) ..
    const int sj = j / nslab, p0 = (j - sj * nslab) * 16, sq = sq0 + sj;
    const bool on = j < nt;
    const int p = p0 + r0;
    xg = __builtin_amdgcn_raw_buffer_load_b128(
        ra, (on && p < Hp) ? (unsigned)((long long)sq * Hp + p) * rowG + colG : 0x80000000u, 0, 0);
    const long long xb = (long long)sq * HpIn + (long long)STRIDE * p0 - pad;     // map row of staged row 0
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
      const long long xr = xb + r0 + 16 * q;
      const bool ok = on && wofX[q] >= 0 && xr >= 0 && xr < xrows;
      xx[q] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? (unsigned)xr * rowX + colX : 0x80000000u, 0, 0);
    }
  };
  auto store = [&](unsigned char* buf) {
    u32x2 p0, p1, p2;
    split3x4(xg, p0, p1, p2);
    *reinterpret_cast<u32x2*>(buf + wofG) = p0;
    *reinterpret_cast<u32x2*>(buf + GPL + wofG) = p1;
    *reinterpret_cast<u32x2*>(buf + 2 * GPL + wofG) = p2;
#pragma unroll
    for (int q = 0; q < NXQ; ++q)
      if (wofX[q] >= 0) {
        split3x4(xx[q], p0, p1, p2);
        *reinterpret_cast<u32x2*>(buf + wofX[q]) = p0;
        *reinterpret_cast<u32x2*>(buf + XPL + wofX[q]) = p1;
        *reinterpret_cast<u32x2*>(buf + 2 * XPL + wofX[q]) = p2;
      }
  };
  // transposing fragment read whose second half lies HS rows further (4 reduction rows = 4 * STRIDE map rows)
  auto tr_frag_s = [&](const unsigned char* p, int half_rows) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + half_rows * 256));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  gload(0);
  store(smw);
  gload(1);
  lds_barrier();
  for (int j = 0; j < nt; ++j) {
    unsigned char* cur = smw + (j & 1) * BUFB;
    store(smw + ((j + 1) & 1) * BUFB);
    gload(j + 2);
    bf16x8 fa[3][2];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) fa[pc][mi] = tr_frag_s(cur + pc * GPL + rofA[mi], 4);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      bf16x8 fb[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) fb[pc] = tr_frag_s(cur + pc * XPL + rofB[t], 4 * STRIDE);
#pragma unroll
      for (int sdeg = 2; sdeg >= 0; --sdeg)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int jj = sdeg - i;
          if (jj < 0 || jj > 2) continue;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][mi], fb[jj], acc[mi][t], 0, 0, 0);
        }
    }
    lds_barrier();
  }
  float* C0 = d.E.C;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        atomicAdd(C0 + (long long)row * d.E.ldc + t * Cin + c0 + wq * 32 + li, acc[mi][t][e]);
      }
}

}  // namespace

// d: a precision-3 descriptor that passed gemm.hip's x6_tap_ok(d, taps) (stride-1 windows of `taps` positions
// over a halo-map image, <= 160 staged positions per 128 rows).  0 = not taken.
int f2g_x6p_ok(const f2g_gemm_desc& d, int taps) {
  const int mode = f2g_opt(F2G_OPT_X6P);    // 0 off, 1 (default) chip-filling grids, 2 whatever the grid
  if (mode == 0 || (taps != 5 && taps != 2)) return 0;
  if (d.A.unit / 32 < 2) return 0;
  if (!x6e::wide_ok(d.E, d.B.rows)) return 0;
  const long long tiles = (long long)((d.A.rows + 255) / 256) * ((d.B.rows + 127) / 128);
  return (mode >= 2 || tiles >= 256) ? 1 : 0;
}

int f2g_launch_x6p(const f2g_gemm_desc& d, int taps, long long a_extent, hipStream_t st) {
  const int M = d.A.rows, N = d.B.rows, K = d.A.cols;
  constexpr size_t smem = (size_t)OPERA + 2 * OPERB;
  static_assert(8 * ESZ <= (int)smem, "epilogue patches fit under the main loop's buffers");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6p_kernel<5>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6p_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  x6p_tap R;
  R.P0 = d.A.P0, R.HpIn = (int)(d.A.seq_stride / d.A.unit), R.offpos = -d.A.pad0, R.C32 = d.A.unit / 32;
  R.bytes = (unsigned)(a_extent * 6);
  dim3 grid((M + 255) / 256, (N + 127) / 128);
  if (taps == 5) hipLaunchKernelGGL(gemm_x6p_kernel<5>, grid, dim3(512), smem, st, d, M, N, K, R);
  else hipLaunchKernelGGL(gemm_x6p_kernel<2>, grid, dim3(512), smem, st, d, M, N, K, R);
  return f2g_check_launch();
}

// ---- tap-walking weight gradients: host side
int f2g_leanw6t_ok(const f2g_gemm_desc& d, int split) {
  if (f2g_opt(F2G_OPT_W6T) == 0) return 0;
  const f2g_operand& A = d.A;
  const f2g_operand& B = d.B;
  if (d.form != 2 || d.precision != 3 || !d.E.atomic || d.E.P0o > 0 || d.E.bias || d.E.res) return 0;
  // the kernels end in a bare atomic accumulation of the accumulators: any other epilogue term goes back to
  // gemm_leanw6_kernel (which runs the full epilogue)
  const f2g_epilogue& E = d.E;
  if ((E.scale != 0.f && E.scale != 1.f) || E.colsum || E.colsum_alpha || E.lrelu_slope != 0.f || E.mask_src ||
      E.aux || E.prelu_slope || E.c_bf16 || E.x3_out || E.accumulate || A.lrelu_src || A.alpha || B.lrelu_src || B.alpha)
    return 0;
  if (B.P1 != 1 || B.P0 < 1 || (B.step0 != 1 && B.step0 != 3) || B.unit < 128 || (B.unit % 128) || !B.unbounded) return 0;
  if (B.cols != 5 * B.unit || B.seglen < B.cols || (B.seq_stride % B.unit) || (B.rows % B.P0)) return 0;
  if (A.cols % 128 || A.rows != B.rows || B.pad0 < 0 || B.pad0 > 8) return 0;
  if (B.step0 == 1 && B.seq_stride != (long long)B.P0 * B.unit) return 0;    // (one flat row index for the whole buffer)
  const long long xrows = (long long)(B.rows / B.P0) * (B.seq_stride / B.unit);
  if (xrows * B.unit * 4 >= 0x7ff00000ll || (long long)A.rows * A.seq_stride * 4 >= 0x7ff00000ll) return 0;
  return 1;
}

int f2g_launch_leanw6t(const f2g_gemm_desc& d, int split, hipStream_t st) {
  const int M = d.A.cols, K = d.A.rows, Cin = d.B.unit;
  constexpr int TAPS = 5;
  if (d.B.step0 == 3) {     // strided layer: whole sequences per block, slabs of 16 gradient rows
    constexpr size_t smem3 = (size_t)2 * (3 * 16 * 256 + 3 * (15 * 3 + TAPS) * 256);
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw6s_kernel<5, 3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      attr3 = true;
    }
    const int nseq = K / d.B.P0, tiles = (M / 128) * (Cin / 128);
    // blocks = tiles x sequence chunks: the chunk count whose block total fills rounds of 256 best
    int best = 1;
    double beste = 0.0;
    for (int z = 1; z <= nseq && z <= 256; ++z) {
      const int per = (nseq + z - 1) / z;
      if (per * d.B.P0 < 256 && z > 1) break;                    // (>= 256 reduction rows per block)
      const long long blocks = (long long)tiles * ((nseq + per - 1) / per);
      const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
      if (eff > beste + 0.02) beste = eff, best = z;
    }
    const int spb = (nseq + best - 1) / best, zs = (nseq + spb - 1) / spb;
    const long long xrows = (long long)nseq * (d.B.seq_stride / Cin);
    dim3 grid(M / 128, Cin / 128, zs);
    hipLaunchKernelGGL((gemm_leanw6s_kernel<5, 3>), grid, dim3(512), smem3, st, d, nseq, spb, xrows);
    return f2g_check_launch();
  }
  constexpr size_t smem = (size_t)2 * (3 * 32 * 256 + 3 * (32 + TAPS - 1) * 256);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_leanw6t_kernel<5>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  // one block per CU: split the rows so that tiles x chunks fill rounds of 256 blocks
  const int tiles = (M / 128) * (Cin / 128);
  int zs = split;
  {
    int best = 1;
    double beste = 0.0;
    for (int z = 1; z <= 64; ++z) {
      const long long blocks = (long long)tiles * z;
      if ((long long)K / z < 1024) break;
      const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
      if (eff > beste + 0.02) beste = eff, best = z;
    }
    zs = best;
  }
  int kchunk = ((K + zs - 1) / zs + 31) / 32 * 32;
  zs = (K + kchunk - 1) / kchunk;
  const long long xrows = (long long)(d.B.rows / d.B.P0) * d.B.P0;
  dim3 grid(M / 128, Cin / 128, zs);
  hipLaunchKernelGGL(gemm_leanw6t_kernel<5>, grid, dim3(512), smem, st, d, K, kchunk, xrows);
  return f2g_check_launch();
}
