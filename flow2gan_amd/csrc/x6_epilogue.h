// Wide epilogue of the fp32-class (six-product) GEMM kernels (gemm.hip: gemm_x6 / x6t / x6f; gemm_x6p.hip).
//
// The generic epilogue (gemm_epilogue) stores 4 bytes per lane and instruction straight from the MFMA
// layout, behind one integer division per ELEMENT for the row map, and the result's three-piece image was
// made by reading the tile back from L2 after a block barrier -- lab builds without them: 24 ms of the
// bf16x6 step's 114 ms of six-product kernel time (profiles/r05_x6p_probe.txt).  Here a wave turns its
// 64 x 64 accumulator tile through a private LDS patch (32 rows at a time, no block barrier) into 8-column
// row segments: every elementwise term of f2g_epilogue with 16-byte loads, 16-byte stores of the fp32 map AND
// of its image pieces from the same registers, column sums reduced over the wave's rows before the atomics,
// one division per ROW.  Everything f2g_epilogue can ask for except atomic accumulation and bf16 output.
#pragma once
#include "common.h"

namespace x6e {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int EPITCH = 72;                      // floats per row of a wave's patch (64 columns + pad)
constexpr int ESZ = 32 * EPITCH * 4 + 64 * 8;   // bytes per wave: 32-row patch + 64 row offsets

// three bf16 pieces of eight floats, 16 bytes per piece (round to nearest even at every step, exactly as
// f2g_split_bf16x3: the produced images are compared bit for bit with it)
__device__ __forceinline__ void split3x8(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
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
  p0 = u32x4{pk[0][0], pk[0][1], pk[0][2], pk[0][3]};
  p1 = u32x4{pk[1][0], pk[1][1], pk[1][2], pk[1][3]};
  p2 = u32x4{pk[2][0], pk[2][1], pk[2][2], pk[2][3]};
}

__device__ __forceinline__ void ld8(const float* p, float (&x)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float (&x)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{x[0], x[1], x[2], x[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{x[4], x[5], x[6], x[7]};
}

// The wave's 64 x 64 accumulator tile (acc[mi][ni] = the 32 x 32 MFMA tiles; rows r0.., columns c0..) ->
// memory.  ep = this wave's private ESZ bytes of LDS; nobody else may touch them, and the block's last reads
// of whatever the patch overlays must be complete.
__device__ __forceinline__ void wide_epilogue(const f2g_epilogue& E, f32x16 (&acc)[2][2], int M, int N, int r0,
                                              int c0, int lane, unsigned char* ep) {
  float* patch = reinterpret_cast<float*>(ep);
  long long* rowoff = reinterpret_cast<long long*>(ep + 32 * EPITCH * 4);
  const int li = lane & 31, h = lane >> 5;
  {
    // element offset of row r0 + lane in the output (row map: one division per row), -1 = past the end
    const int row = r0 + lane;
    long long off = -1;
    if (row < M) {
      if (E.P0o > 0) {
        const int sq = row / E.P0o;
        off = (long long)sq * E.seq_stride_o + (long long)(row - sq * E.P0o) * E.row_stride_o + E.off_o;
      } else {
        off = (long long)row * E.ldc;
      }
    }
    rowoff[lane] = off;
  }
  __builtin_amdgcn_wave_barrier();
  const int c8 = lane & 7, col = c0 + c8 * 8;
  const bool cok = col < N;
  const float scale = E.scale != 0.f ? E.scale : 1.f;
  float bias[8], cs[8], csa[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (E.bias && cok) ? E.bias[col + e] : 0.f, cs[e] = 0.f, csa[e] = 0.f;
  const float fmw = E.fm_ref ? E.fm_w * (E.fm_wdev ? E.fm_wdev[0] : 1.f) : 0.f;
  // The tile's elementwise operand (PReLU pre-activation / residual / leaky-ReLU mask source), all 64 rows of
  // it, requested BEFORE the accumulators go through the patch: loaded inside the row loop below, every one
  // of the eight (mi, j) steps put a full memory round trip behind its LDS read (the fragment registers of
  // the main loop are dead here, so the 64 values per lane cost no occupancy).  One operand is prefetched --
  // aux, else the residual, else the mask source -- which covers every epilogue the path builds but the
  // G-step's mask + feature-matching pair (its second operand is still loaded in the loop).
  const float* pre_src = E.aux ? E.aux : (E.res ? E.res : E.mask_src);
  const int pre_kind = E.aux ? 1 : (E.res ? 2 : (E.mask_src ? 3 : 0));
  float pre[2][4][8];
  if (pre_kind) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = mi * 32 + (lane >> 3) + 8 * j;
        const long long ro = rowoff[r];
        const long long row = r0 + r;
        const long long po = pre_kind == 1 ? row * E.ldaux + col : (pre_kind == 2 ? row * E.ldres + col : ro + col);
        if (ro >= 0 && cok) ld8(pre_src + po, pre[mi][j]);
      }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    // (the patch's previous readers are this wave itself, earlier in program order: the LDS executes a
    // wave's instructions in order, so a wave-private patch needs no barrier)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        patch[((e & 3) + 8 * (e >> 2) + 4 * h) * EPITCH + ni * 32 + li] = acc[mi][ni][e];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (lane >> 3) + 8 * j;                 // row of the patch
      const long long ro = rowoff[mi * 32 + r];
      float v[8];
      ld8(patch + r * EPITCH + c8 * 8, v);
      if (ro < 0 || !cok) continue;
      const long long row = r0 + mi * 32 + r;
      const long long off = ro + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * scale + bias[e];
      if (E.res) {
        float rv[8];
        if (pre_kind == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = pre[mi][j][e];
        } else {
          ld8(E.res + row * E.ldres + col, rv);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (E.gamma ? E.gamma[col + e] : 1.f) * rv[e];
      }
      if (E.aux) {      // PReLU backward against the pre-activation, with the slope's gradient sums
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = pre[mi][j][e];       // (pre_kind == 1 whenever aux is set)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          csa[e] += v[e] * fminf(av[e], 0.f);
          v[e] *= av[e] > 0.f ? 1.f : E.alpha_n[col + e];
        }
      }
      if (E.lrelu_slope != 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : E.lrelu_slope * v[e];
      }
      if (E.prelu_slope) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = v[e] > 0.f ? v[e] : E.prelu_slope[col + e] * v[e];
        if (E.prelu_out) {
          st8(E.prelu_out + row * E.ld_prelu_out + col, pv);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = pv[e];
        }
      }
      if (E.mask_src) {   // leaky-ReLU backward of the layer below (+ feature-matching term)
        float y[8];
        if (pre_kind == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = pre[mi][j][e];
        } else {
          ld8(E.mask_src + off, y);
        }
        if (E.fm_ref) {
          float f[8];
          ld8(E.fm_ref + off, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dl = y[e] - f[e];
            v[e] += fmw * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f));
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= y[e] > 0.f ? 1.f : E.mask_slope;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += v[e];
      if (E.accumulate) {
        float old[8];
        ld8(E.C + off, old);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += old[e];
      }
      st8(E.C + off, v);
      if (E.x3_out) {
        u32x4 p0, p1, p2;
        split3x8(v, p0, p1, p2);
        __bf16* q = reinterpret_cast<__bf16*>(E.x3_out) + (off >> 5) * 96 + (off & 31);
        *reinterpret_cast<u32x4*>(q) = p0;
        *reinterpret_cast<u32x4*>(q + 32) = p1;
        *reinterpret_cast<u32x4*>(q + 64) = p2;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (E.colsum || E.colsum_alpha) {
    float ps[8], pa[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = cs[e], sa = csa[e];
      s += __shfl_xor(s, 8), sa += __shfl_xor(sa, 8);
      s += __shfl_xor(s, 16), sa += __shfl_xor(sa, 16);
      s += __shfl_xor(s, 32), sa += __shfl_xor(sa, 32);
      ps[e] = s, pa[e] = sa;
    }
    if (lane < 8 && cok) {
      if (E.colsum_part_ld > 0) {
        // partial-sum matrices: this wave's row (r0 / 64), plain 16-byte stores, no atomics
        const long long po = (long long)(r0 >> 6) * E.colsum_part_ld + col;
        if (E.colsum) st8(E.colsum + po, ps);
        if (E.colsum_alpha) st8(E.colsum_alpha + po, pa);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (E.colsum) atomicAdd(E.colsum + col + e, ps[e]);
          if (E.colsum_alpha) atomicAdd(E.colsum_alpha + col + e, pa[e]);
        }
      }
    }
  }
}

// what the wide epilogue can do for an epilogue over N output columns (host side)
static inline bool wide_ok(const f2g_epilogue& E, int N) {
  auto a16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (E.atomic || E.c_bf16 || (N & 7)) return false;
  const long long need = E.x3_out ? 7 : 3;      // 16-byte image pieces cover eight elements
  if (!a16(E.C) || !a16(E.x3_out)) return false;
  if (E.P0o > 0 ? ((E.seq_stride_o | E.row_stride_o | E.off_o) & need) : (E.ldc & need)) return false;
  if (E.res && (!a16(E.res) || (E.ldres & 3))) return false;
  if (E.aux && (!a16(E.aux) || (E.ldaux & 3) || !E.alpha_n)) return false;
  if (E.prelu_out && (!a16(E.prelu_out) || (E.ld_prelu_out & 3))) return false;
  if (E.mask_src && !a16(E.mask_src)) return false;
  if (E.fm_ref && (!a16(E.fm_ref) || !E.mask_src)) return false;
  if (E.colsum_part_ld > 0 && ((E.colsum_part_ld & 3) || !a16(E.colsum) || !a16(E.colsum_alpha))) return false;
  return true;
}

}  // namespace x6e
