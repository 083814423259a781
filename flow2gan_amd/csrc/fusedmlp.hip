// Fused pointwise MLP of a ConvNeXt block (reference flow2gan/models/modules.py:487-489 and the
// residual of :491-495):
//     out = W2 . PReLU(W1 . z + b1) + b2 + gamma * x
// for bf16 operands with fp32 accumulation (BASELINE config 2, the plain-bf16 inference mode).  The
// 3C-wide hidden activation never exists in HBM: a block owns BM = 32 * RT rows, keeps their z tile
// (BM x C bf16) in LDS for its whole life, walks the hidden dimension in slabs of 128 columns
// (phase A: a = z . W1_slab^T, K = C; PReLU; the slab goes to LDS as bf16; phase B: out += p .
// W2_slab^T, K = 128) and holds the BM x C fp32 output tile in accumulator registers:
//
//   * 4 waves, ONE per SIMD (the 512-register budget): wave w owns hidden columns w*32..+32 of a
//     slab in phase A and output columns w*C/4..+C/4 in phase B -- RT x NT = 12 accumulator tiles
//     (192 registers) for the three shapes of mel_24k_base / mel_44k: (C, BM) = (768, 64),
//     (512, 96), (384, 128).
//   * Weights never pass through LDS.  f2g_mlp_pack writes W1 and W2 as ONE stream per wave in the
//     exact order and lane layout the MFMA B operand wants (1 KiB = one 32 x 16 fragment, lane l =
//     8 consecutive k of row l & 31), so a wave reads its weights with fully coalesced 16-byte
//     loads straight into registers through a 16-deep ring (16 KiB in flight per wave: the L2
//     latency is hidden by prefetch distance instead of by co-resident waves), and every weight
//     byte enters the CU exactly once per block.
//   * LDS serves only the A operands: z rows (pitch 2C + 16 bytes) and the p slab (pitch 272
//     bytes), both conflict-free for ds_read_b128 (pitch = 4 dwords mod 64).
// Algorithmic traffic per block: z tile + residual in, output tile out, the weight stream
// (12 C^2 bytes) from L2; FLOPs 12 C^2 per row.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef F2G_MLPVAR
#define F2G_MLPVAR 0   // lab builds only (tools/micro/fusedmlp_lab.hip): timing ablations, results garbage
#endif
constexpr int HS = 128;            // hidden columns per slab
#ifndef F2G_MLP_RING
#define F2G_MLP_RING 16
#endif
constexpr int RING_WANTED = F2G_MLP_RING;   // weight fragments in flight per wave
constexpr int PP = HS * 2 + 16;    // LDS pitch of a p row (bytes)

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) {
  return __builtin_bit_cast(bf16x8, v);
}

// spb: slabs per block.  gridDim.y = J > 1 cuts the hidden dimension between J blocks of a row tile
// (a tile count well below the 256 CUs -- 94 tiles at C = 768, B = 64 -- would leave most of the chip
// idle, and more rows per block are not to be had: the output tile fills the registers): every
// part adds its partial output tile atomically onto a zeroed `out`, part 0 carries b2 and the residual.
// DW: the block computes its z tile ITSELF from the residual stream (f2g_fused_block): depthwise
// conv (7 taps) + BiasNorm + condition add + time scale of modules.py:473-485 as the prologue, with
// the arithmetic and the lane -> channel mapping of dwnorm4_kernel (convnext.hip: a wave owns 4
// consecutive frames, a lane 4 consecutive channels per 256-channel chunk, all 10 input rows of a
// frame group requested before anything is consumed) -- z never exists in HBM, and the launch,
// the z store and the z re-read of the separate kernel are gone.
// (the body is a device function of (row tile bx, hidden part by of ny) so that one launch can serve
// several problems: fused_block_multi_kernel below)
template <int RT, int NT, bool DW>
__device__ __forceinline__ void fused_mlp_body(const f2g_fused_mlp_desc& d, const f2g_dwnorm_fwd_desc& P,
                                               const int spb, const int bx, const int by, const int ny,
                                               unsigned char* const smem) {
  constexpr int BM = 32 * RT, C = 128 * NT;
  constexpr int ZP = C * 2 + 16;               // LDS pitch of a z row (bytes)
  constexpr int KA = C / 16;                   // k steps (= weight fragments) of phase A
  constexpr int KB = HS / 16;                  // k steps of phase B (NT fragments each)
  constexpr int PER_SLAB = KA + KB * NT;       // fragments per slab and wave (a multiple of RING)
  constexpr int RING = PER_SLAB % RING_WANTED == 0 ? RING_WANTED : 16;
  static_assert(PER_SLAB % RING == 0, "ring indices must be static across slabs");
  unsigned char* zs = smem;
  unsigned char* ps = smem + BM * ZP;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  // (wave-uniform by construction; said explicitly so that everything derived from it -- the scalar
  // offset of the weight stream above all -- lives in SGPRs instead of behind waterfall loops)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = bx * BM;
  const int S = d.H / HS;
  const int s_beg = by * spb;
  const int s_end = s_beg + spb < S ? s_beg + spb : S;
  const bool lead = by == 0, split = ny > 1;

  // ---- weight stream of this wave: fragments [slab][KA of W1 | KB x NT of W2], 1 KiB each
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.wp, 0, (unsigned)((long long)2 * C * d.H * 2), 0x00020000);
  const unsigned wlane = lane * 16;
  int wso = (w * S + s_beg) * PER_SLAB * 1024;  // scalar byte offset of the next fragment to REQUEST
  u32x4 ring[RING];
#pragma unroll
  for (int i = 0; i < RING; ++i) {
    ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wso, 0);
    wso += 1024;
  }

  if constexpr (DW) {
    constexpr int NCQ = (C + 255) / 256, FW = 4, RW = BM / 4;
    const int F = P.F;
    const float escale = expf(P.log_scale[0]);
    constexpr float invC = 1.f / (float)C;
    int c4[NCQ];
    bool cok[NCQ];
#pragma unroll
    for (int k = 0; k < NCQ; ++k) {
      const int c = 256 * k + 4 * lane;
      cok[k] = c < C;
      c4[k] = cok[k] ? c : 0;
    }
    // Everything below works on four-channel vectors (ext_vector_type: the compiler keeps them in
    // registers under literal indices and pairs the lanes' arithmetic into v_pk_fma_f32 / v_pk_mul_f32
    // -- the scalar version of this prologue ran ~40 vector instructions per element with its
    // spill moves, and the prologue is vector-ALU bound).
    auto ldv = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
    // this lane's taps: 28 consecutive floats of the (C, 1, 7) checkpoint layout (channel e, tap j at
    // 7 e + j) -> one vector per tap; depthwise bias and BiasNorm bias
    f32x4 tw4[NCQ][7], bdw4[NCQ], bet4[NCQ];
#pragma unroll
    for (int k = 0; k < NCQ; ++k) {
      float t28[28];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const f32x4 t = ldv(P.w_dw + (long long)c4[k] * 7 + 4 * i);
        t28[4 * i] = t[0]; t28[4 * i + 1] = t[1]; t28[4 * i + 2] = t[2]; t28[4 * i + 3] = t[3];
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) tw4[k][j] = f32x4{t28[j], t28[7 + j], t28[14 + j], t28[21 + j]};
      bdw4[k] = P.b_dw ? ldv(P.b_dw + c4[k]) : f32x4{0.f, 0.f, 0.f, 0.f};
      bet4[k] = ldv(P.beta + c4[k]);
    }
    const long long last = (long long)d.rows - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int upshift = P.up >= 4 ? 2 : (P.up >= 2 ? 1 : 0);
    // A wave's rows are consecutive: consecutive groups share 6 of their FW + 6 input rows.  The
    // window xr slides by FW rows per group, and the FW new rows of group g + 1 (xn) are requested
    // before group g is computed -- with one wave per SIMD nothing else hides their round trip.
    f32x4 xr[NCQ][FW + 6], xn[NCQ][FW];
    {
      const long long r0 = (long long)m0 + w * RW;
#pragma unroll
      for (int r = 0; r < FW + 6; ++r) {
        long long rr = r0 - 3 + r;
        rr = rr < 0 ? 0 : (rr > last ? last : rr);
        const float* xrow = P.x + rr * P.ldx;
#pragma unroll
        for (int k = 0; k < NCQ; ++k) xr[k][r] = ldv(xrow + c4[k]);
      }
    }
#pragma unroll 1
    for (int g = 0; g < RW / FW; ++g) {
      const int lr0 = w * RW + g * FW;              // first frame of the group inside the tile
      const long long r0 = (long long)m0 + lr0;     // ... in the flattened (item, frame) rows
      // item / frame of every output row of the group; condition and time rows
      int fi[FW], li_[FW];
      bool lv[FW], hascp[FW];
      f32x4 cp[FW][NCQ], te1[FW][NCQ];
      // (item, frame) of the group's first row by one division, of the others by counting on (a
      // division again only where the group crosses an item end); rows past the end: the last row
      const int r0c = (int)(r0 <= last ? r0 : last);
      const int b0 = r0c / F, f0 = r0c - b0 * F;
#pragma unroll
      for (int i = 0; i < FW; ++i) {
        const long long r = r0 + i;
        lv[i] = r <= last;
        int b = b0, f = f0 + i;
        if (f >= F) { const int q = f / F; b += q; f -= q * F; }
        if (!lv[i]) { b = P.B - 1; f = F - 1; }
        fi[i] = f;
        const int len_b = P.lens ? P.lens[b] : F;
        li_[i] = len_b < F ? len_b : F;
        const int fc = P.cproj ? f >> upshift : 0;      // up is 1, 2 or 4
        const bool has = P.cproj && fc < P.Fc;
        hascp[i] = has;
        const float* cprow = has ? P.cproj + ((long long)b * P.Fc + fc) * P.ldcp : P.x;
        const float* terow = P.te ? P.te + (long long)b * P.ldte : P.x;
#pragma unroll
        for (int k = 0; k < NCQ; ++k) {
          cp[i][k] = ldv(cprow + c4[k]);       // (consumed at the end of the group: no arithmetic on
          te1[i][k] = ldv(terow + c4[k]);      //  them here, it would wait for them here)
        }
      }
      if (g + 1 < RW / FW) {
#pragma unroll
        for (int i = 0; i < FW; ++i) {
          long long rr = r0 + FW + 3 + i;
          rr = rr < 0 ? 0 : (rr > last ? last : rr);
          const float* xrow = P.x + rr * P.ldx;
#pragma unroll
          for (int k = 0; k < NCQ; ++k) xn[k][i] = ldv(xrow + c4[k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // (the requests stay in front of the arithmetic)
      // (wave-uniform) does any tap of the group leave its item or its valid length?
      const bool interior = lv[FW - 1] && fi[FW - 1] == fi[0] + FW - 1 && fi[0] >= 3 &&
                            fi[FW - 1] + 3 < li_[0];
      f32x4 u[FW][NCQ];
      float ssq[FW];
#pragma unroll
      for (int i = 0; i < FW; ++i) ssq[i] = 0.f;
#pragma unroll
      for (int k = 0; k < NCQ; ++k) {
#pragma unroll
        for (int i = 0; i < FW; ++i) u[i][k] = bdw4[k];
        if (interior) {
#pragma unroll
          for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < FW; ++i) u[i][k] += tw4[k][j] * xr[k][i + j];
        } else {
#pragma unroll
          for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < FW; ++i) {
              const int fj = fi[i] + j - 3;
              const bool ok = fj >= 0 && fj < li_[i];      // (the masked input of modules.py:473)
              u[i][k] += tw4[k][j] * (ok ? xr[k][i + j] : zero4);
            }
        }
        if (cok[k]) {
#pragma unroll
          for (int i = 0; i < FW; ++i) {
            const f32x4 dl = u[i][k] - bet4[k];
            // (the order and the fused multiply-adds of dwnorm4_kernel: the two kernels produce the same z)
#pragma unroll
            for (int e = 0; e < 4; ++e) ssq[i] = fmaf(dl[e], dl[e], ssq[i]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < FW; ++i) {
        const float rms2 = wave_sum(ssq[i]) * invC;
        const float sc = escale / sqrtf(rms2);
#pragma unroll
        for (int k = 0; k < NCQ; ++k) {
          if (cok[k]) {
            const f32x4 cpv = hascp[i] ? cp[i][k] : zero4;
            const f32x4 tev = P.te ? te1[i][k] + 1.f : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 zv = lv[i] ? (u[i][k] * sc + cpv) * tev : zero4;
            unsigned short hb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hb[e] = __builtin_bit_cast(unsigned short, (__bf16)zv[e]);
            *reinterpret_cast<uint2*>(zs + (lr0 + i) * ZP + c4[k] * 2) =
                make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < RW / FW) {        // slide the window
#pragma unroll
        for (int k = 0; k < NCQ; ++k) {
#pragma unroll
          for (int r = 0; r < 6; ++r) xr[k][r] = xr[k][r + FW];
#pragma unroll
          for (int i = 0; i < FW; ++i) xr[k][6 + i] = xn[k][i];
        }
      }
    }
  } else
  // ---- z tile -> LDS (rows past the end read as zeros: out of the buffer's range)
  if (!(F2G_MLPVAR & 4)) {
    __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        (void*)d.z, 0, (unsigned)((long long)d.rows * d.ldz * 2), 0x00020000);
    constexpr int CPR = C / 8;                 // 16-byte chunks per row
    constexpr int NCH = BM * CPR / 256;        // chunks per thread
    static_assert((BM * CPR) % 256 == 0, "tile chunks must divide among the threads");
    constexpr int G = 8;                       // loads in flight per thread
#pragma unroll
    for (int g0 = 0; g0 < NCH; g0 += G) {
      u32x4 v[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g0 + g < NCH) {
          const int id = tid + 256 * (g0 + g), r = id / CPR, cc = id - r * CPR;
          const long long off = (long long)(m0 + r) * d.ldz * 2 + cc * 16;
          v[g] = __builtin_amdgcn_raw_buffer_load_b128(
              rz, (m0 + r) < d.rows ? (unsigned)off : 0xfffffff0u, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g0 + g < NCH) {
          const int id = tid + 256 * (g0 + g), r = id / CPR, cc = id - r * CPR;
          *reinterpret_cast<u32x4*>(zs + r * ZP + cc * 16) = v[g];
        }
      }
    }
  }

  f32x16 out[RT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float b = (d.b2 && lead) ? d.b2[w * (C / 4) + nt * 32 + li] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[rt][nt][e] = b;
  }
  __syncthreads();

  const unsigned char* za = zs + li * ZP + h * 16;     // A fragments of phase A: row li (+ 32 rt)
  const unsigned char* pa = ps + li * PP + h * 16;     // A fragments of phase B
  unsigned char* pw = ps + (4 * h) * PP + (w * 32 + li) * 2;   // this lane's p stores

  for (int s = s_beg; s < s_end; ++s) {
    const int hc = s * HS + w * 32 + li;               // this lane's hidden column in the slab
    const float b1v = d.b1 ? d.b1[hc] : 0.f;
    const float alv = d.alpha[hc];
    // ---- phase A: a (BM x 32 per wave) = z . W1_slab^T
    f32x16 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) a[rt][e] = 0.f;
    // A fragments ZD - 1 k-steps ahead of their MFMAs: a k-step is only RT MFMAs = 64-128 cycles long,
    // about one LDS round trip with four waves reading (one step ahead, the wait in front of every
    // step's first MFMA stalled the pipe)
    constexpr int ZD = 4;
    bf16x8 fz[ZD][RT];
#pragma unroll
    for (int q = 0; q < ZD - 1; ++q)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        fz[q][rt] = *reinterpret_cast<const bf16x8*>(za + rt * 32 * ZP + q * 32);
#pragma unroll
    for (int ks = 0; ks < KA; ++ks) {
      if (ks + ZD - 1 < KA) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          fz[(ks + ZD - 1) % ZD][rt] =
              *reinterpret_cast<const bf16x8*>(za + rt * 32 * ZP + (ks + ZD - 1) * 32);
      }
      const bf16x8 fb = as_frag(ring[ks % RING]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        a[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz[ks % ZD][rt], fb, a[rt], 0, 0, 0);
      if (!(F2G_MLPVAR & 1)) ring[ks % RING] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wso, 0);
      wso += 1024;
      // keep the written order: left alone, the scheduler sinks the refills to just before their
      // use (3 loads in flight instead of 16) and the fragment reads to just before their MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- PReLU, round to bf16, slab -> LDS (C/D layout: column li, rows (e&3) + 8 (e>>2) + 4 h)
    if (F2G_MLPVAR & 8) {
      float t = b1v + alv;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += a[rt][e];
      if (t == 12345.678f) *reinterpret_cast<float*>(pw) = t;
    } else
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = a[rt][e] + b1v;
        const float p = fmaxf(v, 0.f) + alv * fminf(v, 0.f);
        *reinterpret_cast<__bf16*>(pw + (rt * 32 + (e & 3) + 8 * (e >> 2)) * PP) = (__bf16)p;
      }
    if (!(F2G_MLPVAR & 8)) __syncthreads();
    // ---- phase B: out (BM x C/4 per wave) += p . W2_slab^T
    bf16x8 fp[2][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) fp[0][rt] = *reinterpret_cast<const bf16x8*>(pa + rt * 32 * PP);
#pragma unroll
    for (int k2 = 0; k2 < KB; ++k2) {
      if (k2 + 1 < KB) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          fp[(k2 + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(pa + rt * 32 * PP + (k2 + 1) * 32);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        constexpr int base = KA % RING;
        const int slot = (base + k2 * NT + nt) % RING;
        const bf16x8 fb = as_frag(ring[slot]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          out[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[k2 & 1][rt], fb, out[rt][nt], 0, 0, 0);
        if (!(F2G_MLPVAR & 1)) ring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wso, 0);
        wso += 1024;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(F2G_MLPVAR & 8)) __syncthreads();     // every wave has read the slab before the next one overwrites it
  }

  if (F2G_MLPVAR & 2) {
    float t = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += out[rt][nt][e];
    if (t == 12345.678f) d.out[tid] = t;
    return;
  }
  // ---- out = acc (+ b2 already inside) + gamma * x.  In the C/D layout a lane holds ONE column of
  // 16 rows: stored as it lies, that is 16 dword loads + 16 dword stores per tile and lane, and the
  // stores are issue-bound (measured: 38-41 us of a 92-119 us kernel).  Every wave therefore turns
  // its tiles through 4 KiB of the (now dead) z tile: 16 ds_write_b32, 4 ds_read_b128, and the
  // residual / output travel as 16-byte accesses of whole 128-byte row segments.
  const bool hasres = d.res != nullptr && lead;
  if (!split) {
    float* scr = reinterpret_cast<float*>(smem) + w * 1024;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    // tiles in the order t = nt * RT + rt; the residual of tile t + 1 is requested before tile t is
    // turned (two tiles of loads in flight per wave, the schedule pinned tile by tile: left alone
    // the compiler hoists all twelve tiles' loads = 192 registers it does not have).  Buffer
    // addressing: one per-lane offset, the tile / row-group part in a scalar register, rows past
    // the end out of the resources' range (loads return zeros, stores are dropped).
    __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(hasres ? d.res : d.out), 0, (unsigned)((long long)d.rows * (hasres ? d.ldres : d.ldo) * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        (void*)d.out, 0, (unsigned)((long long)d.rows * d.ldo * 4), 0x00020000);
    const int ldr = (int)d.ldres, ldo = (int)d.ldo;
    const unsigned vres = (unsigned)((rr * ldr + c4) * 4), vout = (unsigned)((rr * ldo + c4) * 4);
    auto load_res = [&](int t, u32x4 (&rv)[4]) {
      const int nt = t / RT, rt = t - nt * RT;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rv[j] = __builtin_amdgcn_raw_buffer_load_b128(
            rres, vres, ((m0 + rt * 32 + 8 * j) * ldr + w * (C / 4) + nt * 32) * 4, 0);
    };
    u32x4 rbuf[2][4];
    if (hasres) load_res(0, rbuf[0]);
#pragma unroll
    for (int t = 0; t < RT * NT; ++t) {
      const int nt = t / RT, rt = t - nt * RT;
      if (hasres && t + 1 < RT * NT) load_res(t + 1, rbuf[(t + 1) & 1]);
      float4 gam = {0.f, 0.f, 0.f, 0.f};
      if (hasres)
        gam = d.gamma ? *reinterpret_cast<const float4*>(d.gamma + w * (C / 4) + nt * 32 + c4)
                      : float4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + li] = out[rt][nt][e];
      // (one wave: its LDS operations execute in order; the compiler keeps the order through `scr`)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = *reinterpret_cast<const float4*>(scr + (rr + 8 * j) * 32 + c4);
        if (hasres) {
          const float4 r = __builtin_bit_cast(float4, rbuf[t & 1][j]);
          v.x += gam.x * r.x;
          v.y += gam.y * r.y;
          v.z += gam.z * r.z;
          v.w += gam.w * r.w;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, vout,
                                               ((m0 + rt * 32 + 8 * j) * ldo + w * (C / 4) + nt * 32) * 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // partial tiles of a hidden-dimension split: atomic accumulation, element by element
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = w * (C / 4) + nt * 32 + li;
    const float gam = hasres ? (d.gamma ? d.gamma[col] : 1.f) : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row0 = m0 + rt * 32 + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + (e & 3) + 8 * (e >> 2);
        float v = out[rt][nt][e];
        if (hasres && row < d.rows) v += gam * d.res[(long long)row * d.ldres + col];
        if (row < d.rows) atomicAdd(d.out + (long long)row * d.ldo + col, v);
      }
    }
  }
}

template <int RT, int NT, bool DW>
__global__ __launch_bounds__(256, 1)
void fused_mlp_kernel(const f2g_fused_mlp_desc d, const f2g_dwnorm_fwd_desc P, int spb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fused_mlp_body<RT, NT, DW>(d, P, spb, blockIdx.x, blockIdx.y, gridDim.y, smem);
}

// Several ConvNeXt blocks in ONE launch (f2g_fused_block_multi): the three Fourier branches of a
// decoder run the same layer at the same time on tiles of very different cost (94 tiles of ~90 us at
// 768 channels, 126 of ~63 us at 512, 188 of ~60 us at 384, one block per CU) -- as three launches
// on three streams they need two rounds of the chip for 1.6 rounds of work, and the 768-channel
// branch, the critical path of every Euler step, waits for CUs.  Here the tiles of all entries form
// one grid in order of decreasing cost, so the hardware's in-order workgroup dispatch IS
// longest-processing-time-first list scheduling: the long tiles start first and the short ones
// fill the CUs as they free up.
constexpr int MULTI_MAX = 4;
struct fused_multi_args {
  f2g_fused_mlp_desc d[MULTI_MAX];
  f2g_dwnorm_fwd_desc P[MULTI_MAX];
  int cum[MULTI_MAX + 1];    // first block of entry i (entries sorted by decreasing cost per tile)
  int rt[MULTI_MAX];         // rows per tile / 32 of entry i
  int n;
};

__global__ __launch_bounds__(256, 1)
void fused_block_multi_kernel(const fused_multi_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = blockIdx.x;
  int i = 0, base = 0, rt = a.rt[0];
#pragma unroll
  for (int k = 1; k < MULTI_MAX; ++k)
    if (k < a.n && bid >= a.cum[k]) { i = k; base = a.cum[k]; rt = a.rt[k]; }
  // the entry's descriptors straight from the kernel-argument segment (a uniform index: scalar loads;
  // indexing the by-value copy would put the whole struct into scratch memory)
  const unsigned char* ka = (const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
  f2g_fused_mlp_desc d;
  f2g_dwnorm_fwd_desc P;
  __builtin_memcpy(&d, ka + offsetof(fused_multi_args, d) + (size_t)i * sizeof(f2g_fused_mlp_desc), sizeof d);
  __builtin_memcpy(&P, ka + offsetof(fused_multi_args, P) + (size_t)i * sizeof(f2g_dwnorm_fwd_desc), sizeof P);
  const int bx = bid - base;
  const int S = d.H / HS;
  if (d.C == 768) fused_mlp_body<2, 6, true>(d, P, S, bx, 0, 1, smem);
  else if (d.C == 512) {
    if (rt == 3) fused_mlp_body<3, 4, true>(d, P, S, bx, 0, 1, smem);
    else fused_mlp_body<2, 4, true>(d, P, S, bx, 0, 1, smem);
  } else {
    if (rt == 4) fused_mlp_body<4, 3, true>(d, P, S, bx, 0, 1, smem);
    else fused_mlp_body<2, 3, true>(d, P, S, bx, 0, 1, smem);
  }
}

// W1 (H, C) and W2 (C, H), fp32 row-major -> the per-wave bf16 fragment stream (see the kernel).
__global__ __launch_bounds__(256)
void mlp_pack_kernel(uint4* __restrict__ dst, const float* __restrict__ w1, long long ld1,
                     const float* __restrict__ w2, long long ld2, int C, int H) {
  const int NT = C / 128, KA = C / 16, KB = HS / 16, per = KA + KB * NT, S = H / HS;
  const long long total = (long long)4 * S * per * 64;      // one 16-byte piece per thread
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    long long f = i >> 6;
    const int j = (int)(f % per);
    f /= per;
    const int s = (int)(f % S), w = (int)(f / S);
    const float* src;
    if (j < KA) {
      src = w1 + (long long)(s * HS + w * 32 + (l & 31)) * ld1 + j * 16 + (l >> 5) * 8;
    } else {
      const int jj = j - KA, k2 = jj / NT, nt = jj - k2 * NT;
      src = w2 + (long long)(w * (C / 4) + nt * 32 + (l & 31)) * ld2 + s * HS + k2 * 16 + (l >> 5) * 8;
    }
    union { __bf16 b[8]; uint4 v; } u;
#pragma unroll
    for (int k = 0; k < 8; ++k) u.b[k] = (__bf16)src[k];
    dst[i] = u.v;
  }
}

template <int RT, int NT, bool DW = false>
int launch_fused(const f2g_fused_mlp_desc& d, hipStream_t st, const f2g_dwnorm_fwd_desc* dw = nullptr) {
  constexpr int BM = 32 * RT, C = 128 * NT;
  constexpr size_t smem = (size_t)BM * (C * 2 + 16) + (size_t)BM * PP;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<RT, NT, DW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  const int tiles = (d.rows + BM - 1) / BM, S = d.H / HS;
  // hidden-dimension split: as many parts as keep the grid within one round of the 256 CUs, each
  // part at least 4 slabs long (option mlp_split = 1 turns it off, n forces n parts)
  const int env_parts = f2g_opt(F2G_OPT_MLP_SPLIT);
  const int forced = d.parts > 0 ? d.parts : env_parts;
  // (measured: the atomic accumulation of the partial tiles costs more than the idle CUs -- 6016 x 768:
  // 117 us whole, 127 us in two parts; 24064 x 384: 83 -> 196 us -- so the library never splits on
  // its own; the path stays for callers / experiments that ask for it)
  int J = forced > 0 ? forced : 1;
  if (J > S) J = S;
  if (J < 1) J = 1;
  const int spb = (S + J - 1) / J;
  J = (S + spb - 1) / spb;
  if (J > 1) {   // partial tiles are accumulated atomically
    if (d.ldo == C) {
      if (hipMemsetAsync(d.out, 0, (size_t)d.rows * C * sizeof(float), st) != hipSuccess) return F2G_ELAUNCH;
    } else if (hipMemset2DAsync(d.out, (size_t)d.ldo * sizeof(float), 0, (size_t)C * sizeof(float),
                                (size_t)d.rows, st) != hipSuccess) {
      return F2G_ELAUNCH;
    }
  }
  f2g_dwnorm_fwd_desc P{};
  if (dw) P = *dw;
  hipLaunchKernelGGL((fused_mlp_kernel<RT, NT, DW>), dim3(tiles, J), dim3(256), smem, st, d, P, spb);
  return f2g_check_launch();
}

// Rows per tile.  The default tiles (64 / 96 / 128 rows at 768 / 512 / 384 channels: what the 192
// accumulator registers hold) amortise the weight stream best, but a launch with few rows then has
// far fewer tiles than the chip has CUs (the condition encoder: 6016 rows x 512 channels = 63 tiles):
// smaller tiles trade weight traffic for parallelism.  option mlp_rt forces rows / 32.
int pick_rt(int C, int rows) {
  const int env_rt = f2g_opt(F2G_OPT_MLP_RT);
  const int rt_max = C == 768 ? 2 : (C == 512 ? 3 : 4);
  if (env_rt > 0) {
    int r = env_rt > rt_max ? rt_max : env_rt;
    if (C == 384 && r == 3) r = 2;
    return r;
  }
  // the largest tile that still gives every CU of HALF the chip a tile
  for (int r = rt_max; r > 1; --r) {
    if (C == 384 && r == 3) continue;
    if ((rows + 32 * r - 1) / (32 * r) >= 128) return r;
  }
  return 1;
}

template <bool DW>
int dispatch_fused(const f2g_fused_mlp_desc& d, hipStream_t st, const f2g_dwnorm_fwd_desc* dw) {
  const int rt = pick_rt(d.C, d.rows);
  if (d.C == 768) return rt == 2 ? launch_fused<2, 6, DW>(d, st, dw) : launch_fused<1, 6, DW>(d, st, dw);
  if (d.C == 512)
    return rt == 3 ? launch_fused<3, 4, DW>(d, st, dw)
                   : (rt == 2 ? launch_fused<2, 4, DW>(d, st, dw) : launch_fused<1, 4, DW>(d, st, dw));
  return rt == 4 ? launch_fused<4, 3, DW>(d, st, dw)
                 : (rt == 2 ? launch_fused<2, 3, DW>(d, st, dw) : launch_fused<1, 3, DW>(d, st, dw));
}

}  // namespace

extern "C" int f2g_fused_mlp_ok(int32_t C, int32_t H) {
  return (C == 768 || C == 512 || C == 384) && H % HS == 0 && H >= HS &&
         (long long)2 * C * H * 2 < 0x7ff00000ll;
}

extern "C" int f2g_mlp_pack(void* dst, const float* w1, int64_t ld1, const float* w2, int64_t ld2,
                            int32_t C, int32_t H, f2g_stream_t stream) {
  if (!dst || !w1 || !w2) return F2G_EINVAL;
  if (!f2g_fused_mlp_ok(C, H)) {
    f2g_set_error("f2g_mlp_pack: C must be 384 / 512 / 768 and H a multiple of 128");
    return F2G_EINVAL;
  }
  const long long total = (long long)2 * C * H / 8;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(f2g_grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<uint4*>(dst), w1, (long long)ld1, w2,
                     (long long)ld2, C, H);
  return f2g_check_launch();
}

extern "C" int f2g_fused_mlp(const f2g_fused_mlp_desc* dp, f2g_stream_t stream) {
  if (!dp) return F2G_EINVAL;
  const f2g_fused_mlp_desc& d = *dp;
  if (!d.z || !d.wp || !d.alpha || !d.out || d.rows < 0) return F2G_EINVAL;
  auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!f2g_fused_mlp_ok(d.C, d.H) || (d.ldz & 7) || d.ldz < d.C ||
      (long long)d.rows * d.ldz * 2 >= 0x7ff00000ll || (d.res && (d.ldres < d.C || (d.ldres & 3) || !al16(d.res))) ||
      d.ldo < d.C || (d.ldo & 3) || !al16(d.out) || (long long)d.rows * d.ldo * 4 >= 0x7ff00000ll ||
      (d.res && (long long)d.rows * d.ldres * 4 >= 0x7ff00000ll) || !al16(d.z) || !al16(d.wp) || (d.gamma && !al16(d.gamma))) {
    f2g_set_error("f2g_fused_mlp: unsupported shape (C in {384, 512, 768}, H % 128 == 0, ldz % 8 == 0, "
                  "16-byte aligned tensors with row strides in whole float4s)");
    return F2G_EINVAL;
  }
  if (d.rows == 0) return F2G_OK;
  return dispatch_fused<false>(d, (hipStream_t)stream, nullptr);
}

static int check_block(const f2g_dwnorm_fwd_desc& w, const f2g_fused_mlp_desc& d) {
  auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!w.x || !w.w_dw || !w.beta || !w.log_scale || !d.wp || !d.alpha || !d.out) return F2G_EINVAL;
  if (!f2g_fused_mlp_ok(d.C, d.H) || w.C != d.C || w.K != 7 || w.B <= 0 || w.F <= 0 ||
      (long long)w.B * w.F != d.rows || (w.ldx & 3) || w.ldx < d.C || !al16(w.x) || !al16(w.w_dw) ||
      !al16(w.beta) || (w.b_dw && !al16(w.b_dw)) || (w.cproj && (!al16(w.cproj) || (w.ldcp & 3) || (w.up != 1 && w.up != 2 && w.up != 4))) ||
      (w.te && (!al16(w.te) || (w.ldte & 3))) || (d.res && (d.ldres < d.C || (d.ldres & 3) || !al16(d.res))) ||
      d.ldo < d.C || (d.ldo & 3) || !al16(d.out) || !al16(d.wp) || (d.gamma && !al16(d.gamma)) ||
      (long long)d.rows * d.ldo * 4 >= 0x7ff00000ll || (d.res && (long long)d.rows * d.ldres * 4 >= 0x7ff00000ll)) {
    f2g_set_error("f2g_fused_block: unsupported shape (C in {384, 512, 768}, 7 taps, rows = B * F, "
                  "16-byte aligned tensors with row strides in whole float4s)");
    return F2G_EINVAL;
  }
  return F2G_OK;
}

extern "C" int f2g_fused_block(const f2g_dwnorm_fwd_desc* wp, const f2g_fused_mlp_desc* mp,
                               f2g_stream_t stream) {
  if (!wp || !mp) return F2G_EINVAL;
  f2g_fused_mlp_desc d = *mp;
  const int rc = check_block(*wp, d);
  if (rc) return rc;
  d.z = wp->x;       // (unused by the DW instances; keeps the descriptor valid)
  d.ldz = 8;
  return dispatch_fused<true>(d, (hipStream_t)stream, wp);
}

extern "C" int f2g_fused_block_multi(const f2g_dwnorm_fwd_desc* wp, const f2g_fused_mlp_desc* mp,
                                     int32_t n, f2g_stream_t stream) {
  if (!wp || !mp || n < 1 || n > MULTI_MAX) return F2G_EINVAL;
  for (int i = 0; i < n; ++i) {
    const int rc = check_block(wp[i], mp[i]);
    if (rc) return rc;
  }
  // entries in order of decreasing cost per tile (12 C^2 FLOPs per row x 32 RT rows)
  // rows per tile of an entry: the default tiles, or (options multi_rt512 / multi_rt384 = 2) 64-row
  // tiles for the cheaper branches -- finer granularity at the tail of the list schedule
  const int rt512 = f2g_opt(F2G_OPT_MULTI_RT512), rt384 = f2g_opt(F2G_OPT_MULTI_RT384);
  auto bm_of = [&](int C) { return C == 768 ? 64 : (C == 512 ? (rt512 == 2 ? 64 : 96) : (rt384 == 2 ? 64 : 128)); };
  int order[MULTI_MAX];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0; --j) {
      const long long cj = (long long)mp[order[j]].C * mp[order[j]].C * bm_of(mp[order[j]].C);
      const long long cp = (long long)mp[order[j - 1]].C * mp[order[j - 1]].C * bm_of(mp[order[j - 1]].C);
      if (cj > cp) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    }
  fused_multi_args a{};
  int total = 0;
  size_t smem = 0;
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    a.d[k] = mp[i];
    a.d[k].z = wp[i].x;
    a.d[k].ldz = 8;
    a.P[k] = wp[i];
    a.cum[k] = total;
    const int BM = bm_of(mp[i].C);
    a.rt[k] = BM / 32;
    total += (mp[i].rows + BM - 1) / BM;
    const size_t sm = (size_t)BM * (mp[i].C * 2 + 16) + (size_t)BM * PP;
    smem = sm > smem ? sm : smem;
  }
  for (int k = n; k <= MULTI_MAX; ++k) a.cum[k] = total;
  a.n = n;
  if (total == 0) return F2G_OK;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_block_multi_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (384 * 2 + 16) + 128 * PP);
    attr_done = true;
  }
  hipLaunchKernelGGL(fused_block_multi_kernel, dim3(total), dim3(256), smem, (hipStream_t)stream, a);
  return f2g_check_launch();
}
