/*
 * flow2gan_hip.h -- C ABI of libflow2gan_hip.so (gfx950 / MI355X).
 *
 * The reference (k2-fsa/Flow2GAN) has no FFI layer: its hot path is Python calling ATen
 * ops.  This header is the boundary a maintainer would bind instead (ctypes stub in
 * INTEGRATION.md).  Every entry point
 *   - takes raw DEVICE pointers + explicit sizes, no torch types;
 *   - enqueues on the caller's hipStream_t and returns immediately (no allocation, no sync);
 *   - returns 0 on success, a negative F2G_E* code otherwise (never throws);
 *   - computes in fp32 (the reference's precision, run_libritts.sh:152 `--use-fp16 0`).
 *
 * Activation layout is CHANNELS-LAST: a (batch, channels, frames) tensor of the reference is
 * held as rows = batch*frames, cols = channels (row stride `ld` floats).  Audio is (batch, T)
 * and the mel condition is (batch, n_mels, frames) exactly as the reference passes them.
 *
 * Each declaration cites the reference code (file:line under /root/reference) it replaces.
 */
#ifndef FLOW2GAN_HIP_H
#define FLOW2GAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* f2g_stream_t; /* hipStream_t */

enum {
  F2G_OK = 0,
  F2G_EINVAL = -1,  /* bad argument / unsupported shape */
  F2G_ELAUNCH = -2, /* hipGetLastError() after launch was not hipSuccess */
};

/* Library identification: "flow2gan_hip <version> gfx950". */
const char* f2g_version(void);
/* Text of the last hip error seen by this library (thread-unsafe, diagnostics only). */
const char* f2g_last_error(void);

/* Library options: every tunable of the dispatch (which kernel family may take a launch, tile rules, the
 * bit-reproducible mode) is an int in one table, initialised once -- built-in defaults, then
 * F2G_OPTS="name=value,..." and F2G_DETERMINISTIC from the environment -- and changed afterwards only through
 * this setter; nothing reads the environment on a launch path.  Names: lean, lean_tall, lean_tap, lean_wgrad,
 * x6_tap, x6_wide, x6p, w6t, deterministic, streamk, conv2ch_v2, conv32_v2, conv32_wgrad_v2, mlp_rt, mlp_split,
 * multi_rt384, multi_rt512, streamk_min (meanings: csrc/common.h).  Unknown name: F2G_EINVAL.  Not thread safe against
 * concurrent launches. */
int f2g_set_option(const char* name, int32_t value);
int f2g_get_option(const char* name, int32_t* value);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM operand: a matrix whose rows are "pixels" and whose columns are a contiguous
 * window of the channels-last source -- i.e. an im2col view that is never materialised.
 *   row r  -> (s, p1, p0) with r = (s*P1 + p1)*P0 + p0
 *   col c  -> (seg, o)    with seg = c / seglen, o = c % seglen
 *   line   l1 = p1*step1 - pad1 + seg          valid iff 0 <= l1 < L1
 *   offset e  = (p0*step0 - pad0)*unit + o     valid iff 0 <= e < L0u   (or reflected)
 *   addr   = base + s*seq_stride + l1*line_stride + e
 * Invalid elements read as 0.  A plain row-major matrix is {P1=P0=1, seglen=cols, L1=1,
 * L0u=cols, seq_stride=ld}.  With `alpha` set, PReLU(alpha[c]) is applied on load
 * (modules.py:444,488); with `lrelu_src` set the element is multiplied by the leaky-ReLU
 * derivative of lrelu_src at the same address (discriminators.py:94,205 backward).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float* base;
  int32_t rows, cols;
  int32_t P1, P0;
  int32_t seglen;
  int32_t step1, pad1, L1;
  int32_t step0, pad0, unit, L0u;
  int64_t seq_stride, line_stride;
  int32_t reflect;
  /* 3: `base` is a f2g_split_bf16x3 image (precision 3, form 0 only): of the plain (rows, cols) matrix, or
   *    -- for single-segment windows that never leave their sequences (halo maps) -- the flat image of the
   *    contiguous buffer the windows address, `base` moved to the image of the same element (element e of
   *    a contiguous buffer sits at (e / 32) * 192 + piece * 64 + (e % 32) * 2 bytes of its image; every
   *    stride / offset of the descriptor stays in elements and must be a multiple of 32);
   * 4: (operand B of a precision-3, form-0 launch whose A is fp32; rows % 32 == 0, rows > 32, plain matrix) the
   *    f2g_split_bf16x3 image in MFMA FRAGMENT order: unit u = (((g * (cols / 32) + slab) * 3 + piece) * 2 +
   *    k step) * 64 + (k half) * 32 + (row % 32) of 16 bytes holds the 8 bf16 of row 32 g + (row % 32), columns
   *    32 slab + 16 (k step) + 8 (k half) ... + 8 of that piece -- 1 KB per (group, slab, piece, k step), read
   *    straight into the operand registers of v_mfma_f32_32x32x16_bf16 (gemm_x6g_kernel);
   * 1: `base` holds the split-bf16 image written by f2g_split_bf16 (same addressing);
   * 2: `base` is a TRUE bf16 tensor (f2g_to_bf16 or a bf16 producer): strides / offsets stay in
   *    elements, the reduction needs whole 64-element slabs (precision 2, lean kernel only) */
  int32_t split;
  const float* alpha;
  const float* lrelu_src;
  float lrelu_slope;
  /* 1: the windows of this operand may be read past the ends of their sequence (inside the
   * underlying buffer, zeros outside it) because the caller guarantees that whatever they pair
   * with is zero there -- the weight gradient of a conv whose gradient map carries zero halo rows.
   * Only the split-bf16 weight-gradient kernel looks at it. */
  int32_t unbounded;
} f2g_operand;

/* Epilogue of f2g_gemm: v = acc (+bias[n]) (+gamma[n]*res[r,n]); optional PReLU-derivative
 * against `aux` with column sums for d(alpha); optional leaky-ReLU; optional column sums of
 * the stored value (bias gradient); store / accumulate / atomic-add into C with the row map
 *   r -> (P0o ? (r / P0o)*seq_stride_o + (r % P0o)*row_stride_o + off_o : r*ldc). */
typedef struct {
  float* C;
  int64_t ldc;
  int32_t P0o;
  int32_t c_bf16; /* 1: C is a bf16 tensor (ldc in elements): plain stores of the lean kernel only */
  int64_t seq_stride_o, row_stride_o, off_o;
  const float* bias;
  const float* res;
  int64_t ldres;
  const float* gamma;
  const float* aux; /* pre-activation a[r,n]: v *= (a > 0 ? 1 : alpha_n[n]) */
  int64_t ldaux;
  const float* alpha_n;
  float* colsum_alpha; /* += sum_r acc * min(a, 0)   (d PReLU slope) */
  float* colsum;       /* += sum_r v                 (d bias) */
  float lrelu_slope;   /* != 0: v = v > 0 ? v : slope*v */
  float scale;         /* v *= scale before everything else if != 0 (0 means 1) */
  int32_t accumulate;  /* C += v */
  int32_t atomic;      /* atomicAdd(C, v) (split-K / shared gradients) */
  /* PReLU of the result fused into the producer (modules.py:444,488: act(pwconv1(x))):
   * p = v > 0 ? v : prelu_slope[n]*v.  With prelu_out set, C receives v (the pre-activation the
   * backward needs) and prelu_out the activation; with prelu_out NULL, C receives p.  Plain
   * stores only (no row map / split-K / accumulate). */
  const float* prelu_slope;
  float* prelu_out;
  int64_t ld_prelu_out;
  /* Leaky-ReLU backward of the layer BELOW fused into a data gradient (discriminators.py:94,205
   * backward; gan.py:76-87 feature matching): with y = mask_src at the output's own address (same
   * row map; the caller offsets the pointer to the right half of a stacked batch)
   *   v = (v + fm_w * fm_wdev[0] * sign(y - fm_ref)) * (y > 0 ? 1 : mask_slope)
   * (the fm term only when fm_ref is set) -- the gradient of the pre-activation, whose column sums
   * (`colsum`) are that layer's bias gradient.  Plain / row-mapped stores only. */
  const float* mask_src;
  const float* fm_ref;
  const float* fm_wdev;
  float mask_slope;
  float fm_w;
  /* precision 3 only: besides C, the value is written into the f2g_split_bf16x3 image of the CONTIGUOUS
   * buffer C points into -- element e of that buffer (e = the store offset from C) at
   * x3_out + (e / 32) * 192 + piece * 64 + (e % 32) * 2 bytes -- so the next precision-3 GEMM reads its
   * operand without an image pass.  C must sit on a 32-element boundary of that buffer; plain /
   * row-mapped / accumulating stores (no atomics, no prelu_out). */
  void* x3_out;
  /* > 0 (precision 3, form 0, the wide epilogue only -- ask f2g_gemm_colsum_part_rows first): `colsum` /
   * `colsum_alpha` point at PARTIAL-sum matrices with one row per 64 output rows, `colsum_part_ld` floats
   * apart (16-byte aligned, ld % 4 == 0): the wave that owns output rows [64 i, 64 i + 64) STORES its column
   * sums into row i instead of adding them atomically to a shared vector, and the caller sums the rows
   * (f2g_colsum).  The d(bias) / d(PReLU slope) sums of a 24064-row data gradient are 376 same-address
   * atomics per column otherwise -- 13-24 % of those launches.  f2g_gemm refuses the descriptor (F2G_EINVAL)
   * when the kernel it would pick cannot do this. */
  int64_t colsum_part_ld;
} f2g_epilogue;

/* form: 0 = C[r,n] = sum_k A[r,k] * B[n,k]   (forward; B = weights [n][k])
 *       1 = C[r,n] = sum_k A[r,k] * B[k,n]   (data gradient; B = weights [k][n])
 *       2 = C[m,n] = sum_r A[r,m] * B[r,n]   (weight gradient; reduction over rows, split_k>=1)
 * Replaces torch conv1d/conv2d/matmul/linear + their backward on the hot path
 * (modules.py:443-451,487-489,511,563,576-580,593; discriminators.py:65-76,171-184;
 *  modules.py:69-78,106-115 as DFT matrices; modules.py:213, gan.py:47-54 filterbanks). */
typedef struct {
  f2g_operand A, B;
  f2g_epilogue E;
  int32_t form;
  /* reduction split.  form 2: >= 1, needs E.atomic when > 1.  forms 0/1: 1 = off; > 1 = the K
   * range is cut into that many chunks whose partial tiles are added atomically (the library
   * zeroes C first unless E.accumulate; bias / residual enter once; linear epilogues only);
   * 0 = the library picks (splits only deep reductions that would leave the last wave idle). */
  int32_t split_k;
  /* 0 = exact fp32 MFMA (bit-for-bit an fmaf chain); 1 = split-bf16: every fp32 operand is staged as
   * hi+lo bf16 and each product is hi*hi + hi*lo + lo*hi with fp32 accumulation (per-product
   * relative error <= ~2^-16, i.e. ~100x tighter than plain bf16), ~3-5x the throughput;
   * 2 = plain bf16 operands (hi part only, one MFMA per product), fp32 accumulation: the
   * throughput mode of BASELINE config 2 (inference), not a parity mode;
   * 3 = fp32-CLASS products on the bf16 pipe: three bf16 pieces per operand, six MFMAs per product.
   *     form 0: both operands as three-piece images (f2g_split_bf16x3 / E.x3_out of the producing
   *     GEMM; f2g_gemm_x6_ok); form 2: the fp32 operands themselves -- whatever f2g_gemm_lean_ok accepts
   *     for form 2, with E.atomic when split_k > 1 -- split into pieces inside the kernel. */
  int32_t precision;
  int32_t _pad3;
} f2g_gemm_desc;

int f2g_gemm(const f2g_gemm_desc* d, f2g_stream_t stream);
/* 1 if f2g_gemm would run this form-2 (weight-gradient) descriptor -- precision 0, E.atomic, split_k as set --
 * on the K-major lean kernel (two blocks per CU): the same rule as the dispatch itself, so that the host can
 * choose its split factor for that kernel's rounds of 512 blocks without restating the conditions. */
int f2g_gemm_wgrad_lean(const f2g_gemm_desc* d);
/* 1 if f2g_gemm would run this form-0 descriptor on the lean kernel (buffer loads, no VALU in the K
 * loop), whatever its precision: the host asks before pre-splitting the operands of a precision-1
 * GEMM.  Form 2: 1 if the split-bf16 weight-gradient kernel (K-major operands transposed by
 * ds_read_b64_tr_b16) applies.  No launch. */
int f2g_gemm_lean_ok(const f2g_gemm_desc* d);
/* dst = split-bf16 image of src (n floats, n % 4 == 0, both 16-byte aligned): every aligned group
 * of four floats becomes its four bf16 high parts followed by the four bf16 remainders (x = hi + lo
 * to ~2^-17 |x|), the same 16 bytes at the same offset.  An operand with `split = 1` over such an
 * image is read by the lean kernel's split-bf16 instances without any conversion in the K loop
 * (precision 1: three MFMAs per product; precision 2: the high parts only = plain bf16 operands;
 * both operands must be split; anything else is F2G_EINVAL). */
int f2g_split_bf16(float* dst, const float* src, int64_t n, f2g_stream_t stream);
/* dst (bf16, n elements, n % 4 == 0) = round-to-nearest-even of src: a TRUE bf16 tensor for operands
 * with split = 2 (BASELINE config 2: bf16 activations in HBM).  f2g_gemm_lean_ok returns 3 instead
 * of 1 when a form-0 descriptor can also be served from such tensors. */
int f2g_to_bf16(void* dst, const float* src, int64_t n, f2g_stream_t stream);
/* Three-piece image for precision 3 (fp32-class products on the bf16 matrix pipe: every value
 * x = p0 + p1 + p2 with bf16 pieces, a product = the six MFMAs with i + j <= 2, fp32 accumulation;
 * error <= ~2^-23 per product, the class of fp32 rounding -- tools/micro/x6_lab.hip).  dst
 * (f2g_split_bf16x3_bytes(rows, K) = 6 * rows * K bytes) = the (rows, K) row-major fp32 matrix src
 * (row stride ld floats, K % 32 == 0) laid out [row][K / 32][piece][32] bf16: 192 contiguous bytes per
 * row and 32-element slab.  f2g_gemm with precision = 3 takes form 0 descriptors whose operands are BOTH
 * given as such images (f2g_operand.split = 3, rows / cols = the logical extents; B a plain matrix, A a
 * plain matrix or halo-map windows, see f2g_operand.split);
 * f2g_gemm_x6_ok(d) applies the tests of f2g_gemm's own precision-3 dispatch to a form-0 descriptor
 * (E.x3_out included: set it before asking) and returns a bit mask, 0 = not at precision 3: bit 0 = over
 * three-piece images of both operands (also reported for the fp32 tensors the images would be made
 * of), bit 1 = and then on the image kernel's tap-walking instance (stride-1 conv windows of 5 or 2
 * positions over a halo map, whose positions a tile stages once per 32-channel slab), bit 2 = over the
 * fp32 operands exactly as handed over (split = 0: the instance that splits inside the kernel; its
 * alignment / stride conditions hold).  All epilogues of the
 * generic kernel apply (bias, residual, PReLU with both outputs, PReLU backward with column sums, ...). */
int64_t f2g_split_bf16x3_bytes(int32_t rows, int32_t K);
int f2g_split_bf16x3(void* dst, const float* src, int64_t ld, int32_t rows, int32_t K, f2g_stream_t stream);
int f2g_gemm_x6_ok(const f2g_gemm_desc* d);
/* Number of partial-sum rows (= output rows / 64, rounded up to whole tiles) the launch of this descriptor
 * would write when E.colsum_part_ld > 0, or 0 when the kernel f2g_gemm picks for it cannot store partial
 * column sums (then leave colsum_part_ld at 0: atomics).  Ask with the descriptor exactly as it will be
 * launched (precision, operand formats); E.colsum / colsum_alpha / colsum_part_ld may still be unset. */
int32_t f2g_gemm_colsum_part_rows(const f2g_gemm_desc* d);

/* Kernel family the last f2g_gemm call dispatched to (benchmark diagnostics, not thread safe):
 * 0 generic MFMA kernels, 1 lean kernel, 2 lean kernel in stream-K mode, 3 narrow VALU kernels,
 * 4 the precision-3 kernels, 5 the precision-3 kernel for <= 32 output columns (gemm_x6n_kernel). */
int f2g_gemm_last_path(void);

/* ------------------------------------------------------------------------------------------
 * Fused depthwise-conv(k=7) + BiasNorm + cond add + time scale  (modules.py:473-485, A.4):
 *   u = dwconv7(x * mask) + b_dw ; v = u * mean_c((u-beta)^2)^-1/2 * exp(log_scale)
 *   z = (v + cproj[row/up, :]) * (1 + te[b, :])
 * x, z: (B*F rows, C) channels-last.  lens (int32, per batch item, frames) may be NULL (no mask).
 * cproj (may be NULL): rows B*Fc, frame f reads row min(f/up, ...) and 0 beyond Fc*up.
 * te (may be NULL): (B, ld_te).   w_dw: (C,1,7) as in the checkpoint, b_dw: (C).
 * rinv (optional out, B*F): the per-row normaliser s, saved for backward.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float* x;
  int64_t ldx;
  float* z;
  int64_t ldz;
  int32_t B, F, C, K; /* K = taps (odd, <= 7) */
  const int32_t* lens;
  const float* w_dw;
  const float* b_dw;
  const float* beta;
  const float* log_scale;
  const float* cproj;
  int64_t ldcp;
  int32_t Fc, up;
  const float* te;
  int64_t ldte;
  float* rstd; /* (B*F) out: s = r^-1/2 * exp(log_scale) */
  /* forward only -- what z holds (it is read by GEMMs only): 0 fp32; 1 the split-bf16 image of the
   * same values (f2g_split_bf16 layout, same addressing: ldz in floats); 2 a bf16 tensor (ldz in
   * ELEMENTS).  1 / 2 need C % 4 == 0 and 16-byte aligned rows. */
  int32_t z_format;
  int32_t _pad;
} f2g_dwnorm_fwd_desc;
int f2g_dwnorm_fwd(const f2g_dwnorm_fwd_desc* d, f2g_stream_t stream);

/* Backward of the above (A.3/A.4).  Given gz = dL/dz:
 *   g_cproj[row/up,:] += gz*(1+te)       (atomic; may be NULL)
 *   g_te[b,:]        += sum_f gz*(v+cproj)   (atomic; may be NULL)
 *   du (out, B*F x C): gradient w.r.t. the dwconv output u
 *   g_beta[C], g_log_scale[1] += ...      (atomic)
 * then f2g_dwconv_bwd turns du into dx (+= into gx with the residual path) and dw/db. */
typedef struct {
  f2g_dwnorm_fwd_desc f; /* same tensors as forward (z unused) */
  const float* gz;
  int64_t ldgz;
  float* du;
  int64_t lddu;
  float* g_cproj;
  float* g_te;
  float* g_beta;
  float* g_log_scale;
  /* optional workspace of f2g_dwnorm_bwd_workspace(B,F,C,up) floats: block partials of g_beta /
   * g_te / g_log_scale are written there without atomics and summed by a second launch (the
   * sums are ADDED to g_beta/g_te/g_log_scale).  NULL: contended atomics. */
  float* partials;
  /* 1: g_cproj is STORED instead of accumulated (the caller owns these columns exclusively and
   * zero-filled them: every condition row receives its one sum from one wave, so the read of the
   * read-modify-write -- a dependent memory round trip per condition row -- can go). */
  int32_t g_cproj_store;
  int32_t _pad2;
} f2g_dwnorm_bwd_desc;
int f2g_dwnorm_bwd(const f2g_dwnorm_bwd_desc* d, f2g_stream_t stream);
int64_t f2g_dwnorm_bwd_workspace(int32_t B, int32_t F, int32_t C, int32_t up);

/* dx[c,t] (+)= mask * sum_j w[c,j] du[c,t-j+pad] (+ gamma[c]*gres[c,t]);
 * g_w[c,j] += sum du[c,t]*xm[c,t+j-pad]; g_b[c] += sum du; g_gamma[c] += sum gres*x. */
typedef struct {
  const float* du;
  int64_t lddu;
  const float* x;
  int64_t ldx;
  float* gx;
  int64_t ldgx;
  int32_t B, F, C, K;
  const int32_t* lens;
  const float* w_dw;
  const float* gres; /* gradient of the block output (residual path), may be NULL */
  int64_t ldgres;
  const float* gamma; /* ChannelScale (C), may be NULL */
  float* g_w;
  float* g_b;
  float* g_gamma;
  float* partials; /* optional workspace of f2g_dwconv_bwd_workspace(B,F,C,K) floats (as above) */
} f2g_dwconv_bwd_desc;
int f2g_dwconv_bwd(const f2g_dwconv_bwd_desc* d, f2g_stream_t stream);
int64_t f2g_dwconv_bwd_workspace(int32_t B, int32_t F, int32_t C, int32_t K);

/* BiasNorm alone (modules.py:286-416), channels-last, in place allowed (y may equal x). */
int f2g_biasnorm_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t C,
                     const float* beta, const float* log_scale, f2g_stream_t stream);
/* gx = BiasNorm'(x)^T gy; g_beta, g_log_scale accumulated atomically. */
int f2g_biasnorm_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy, float* gx,
                     int64_t ldgx, int32_t rows, int32_t C, const float* beta,
                     const float* log_scale, float* g_beta, float* g_log_scale,
                     f2g_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * iSTFT tail (modules.py:106-115,719 + generator.py:165-168,263-264): overlap-add of windowed
 * inverse-DFT frames (B*F rows of n_fft samples, produced by f2g_gemm with the inverse-DFT
 * matrix), divide by the window envelope, trim n_fft/2, fit to T, then
 *   out[b,t] (=|+=) wbranch[b] * y[b,t]
 * so that the three branches accumulate their mean (and branch dropout weights) in place.
 * ---------------------------------------------------------------------------------------- */
int f2g_istft_ola(const float* frames, int64_t ldf, float* out, int32_t B, int32_t F,
                  int32_t n_fft, int32_t hop, int32_t T, const float* window,
                  const float* wbranch, float wscale, int32_t accumulate, f2g_stream_t stream);
/* The same for up to four branches in one launch:
 *   out[b,t] (=|+=) sum_i wscale * wbranch_i[b] * y_i[b,t]   (summed in the order of the entries:
 * what n f2g_istft_ola calls, the first with `accumulate`, the others accumulating, leave in out). */
typedef struct {
  const float* frames[4];
  int64_t ldf[4];
  int32_t F[4], n_fft[4], hop[4];
  const float* window[4];
  const float* wbranch[4]; /* (B) each, or NULL */
  int32_t n, _pad;
} f2g_ola_multi_desc;
int f2g_istft_ola_multi(const f2g_ola_multi_desc* d, float* out, int32_t B, int32_t T, float wscale,
                        int32_t accumulate, f2g_stream_t stream);
/* Backward: gframes[b,m,n] = gout[b, m*hop+n-n_fft/2] / env * wbranch[b]*wscale (the synthesis
 * window itself is folded into the inverse-DFT matrix). */
int f2g_istft_ola_bwd(const float* gout, float* gframes, int64_t ldf, int32_t B, int32_t F,
                      int32_t n_fft, int32_t hop, int32_t T, const float* window,
                      const float* wbranch, float wscale, f2g_stream_t stream);
/* Fold the gradient of reflect-padded STFT frames back onto the signal (A.1 backward):
 * gx[b,t] (+)= sum over frames/taps of gframes[b,m,n] mapped through the reflect padding. */
int f2g_frames_fold(const float* gframes, int64_t ldf, float* gx, int32_t B, int32_t F,
                    int32_t n_fft, int32_t hop, int32_t T, int32_t accumulate,
                    f2g_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / reductions used by the flow-matching step and the losses.
 * ---------------------------------------------------------------------------------------- */
/* y = a*x0 + b*x1 with per-row (per batch item) coefficients: a = ca[i] or sa, b = cb[i] or sb
 * (generator.py:217 interpolation, :263-264 Euler update). */
int f2g_axpby_rows(float* y, const float* x0, const float* x1, const float* ca, const float* cb,
                   float sa, float sb, int32_t rows, int32_t cols, f2g_stream_t stream);
/* y = clamp(x, lo, hi) (generator.py:268-269) */
int f2g_clamp(float* y, const float* x, float lo, float hi, int64_t n, f2g_stream_t stream);
/* SiLU forward / backward (modules.py:571) */
int f2g_silu(float* y, const float* x, int64_t n, f2g_stream_t stream);
int f2g_silu_bwd(float* gx, const float* gy, const float* x, int64_t n, f2g_stream_t stream);
/* Sinusoidal time embedding (modules.py:217-232): out (B, dim) = [sin | cos](scale*t*f_k). */
int f2g_time_embedding(float* out, const float* t, int32_t B, int32_t dim, float scale,
                       f2g_stream_t stream);
/* Zero frames >= lens[b] of a channels-last tensor (modules.py:714-715). */
int f2g_mask_rows(float* x, int64_t ld, int32_t B, int32_t F, int32_t C, const int32_t* lens,
                  f2g_stream_t stream);
/* out[c] (+)= sum_r a[r,c] (* b[r,c] if b) */
int f2g_colsum(float* out, const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows,
               int32_t cols, f2g_stream_t stream);
/* out[(b*Fc + f/up), c] += g[(b*F+f), c]  (gradient of the repeat-interleave, modules.py:676-678) */
int f2g_rows_fold_up(float* out, int64_t ldo, const float* g, int64_t ldg, int32_t B, int32_t F,
                     int32_t Fc, int32_t up, int32_t C, f2g_stream_t stream);
/* (B, C, F) <-> (B*F, C) transposes at the boundary (mel condition in, features out). */
int f2g_bct_to_rows(float* out, int64_t ldo, const float* in, int32_t B, int32_t C, int32_t F,
                    f2g_stream_t stream);
int f2g_rows_to_bct(float* out, const float* in, int64_t ldi, int32_t B, int32_t C, int32_t F,
                    f2g_stream_t stream);
/* Generic strided 4-d permutation copy: out[i0,i1,i2,i3] (contiguous) = in[sum i_k*stride_k]
 * (weight re-layout (Cout,Cin,kh,kw) <-> [Cout][kh][kw][Cin], einops rearrange at
 * discriminators.py:193). */
int f2g_permute4(float* out, const float* in, int32_t n0, int32_t n1, int32_t n2, int32_t n3,
                 int64_t s0, int64_t s1, int64_t s2, int64_t s3, f2g_stream_t stream);

/* LimitParamValue backward (modules.py:236-256): g = -g where (g>0 & p<lo), then where (g<0 & p>hi). */
int f2g_limit_grad(float* g, const float* p, float lo, float hi, int64_t n, f2g_stream_t stream);
/* Strided 3-d block copy / accumulate: out[b*so0 + r*so1 + c] (+)= in[b*si0 + r*si1 + c]
 * (condition rows fitted to the branch frame count, modules.py:679; stacking per-block weights). */
int f2g_copy3(float* out, int64_t so0, int64_t so1, const float* in, int64_t si0, int64_t si1,
              int32_t n0, int32_t n1, int32_t n2, int32_t accumulate, f2g_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Loss kernels.
 * ---------------------------------------------------------------------------------------- */
/* Spectrogram magnitude from packed DFT rows [Re(0..nb-1) | Im(0..nb-1)] (ld):
 * out[r,k] = (re^2+im^2)^(power/2), power in {1,2}  (torchaudio Spectrogram, modules.py:180-192,
 * gan.py:47-54).  Backward: gpacked = gout * d|.|^p. */
int f2g_spec_power(float* out, int64_t ldo, const float* packed, int64_t ldp, int32_t rows,
                   int32_t nb, int32_t power, f2g_stream_t stream);
int f2g_spec_power_bwd(float* gpacked, int64_t ldp, const float* gout, int64_t ldo,
                       const float* packed, int32_t rows, int32_t nb, int32_t power,
                       f2g_stream_t stream);
/* Stage-1 loss (generator.py:186-199, A.7): given S_err, S_gt (rows=B*F, n_filt cols) and lens
 * (frames): loss += sum mask * S_err * clamp((S_gt+eps)^-p, lo, hi) * inv_denom;
 * g_err = mask * clamp(...) * inv_denom * gscale (optional). */
int f2g_fm_spec_loss(float* loss, float* g_err, const float* s_err, const float* s_gt, int32_t B,
                     int32_t F, int32_t n_filt, const int32_t* lens, float eps, float power,
                     float lo, float hi, float inv_denom, f2g_stream_t stream);
/* Unweighted stage-1 loss (generator.py:181-184, spec_scaling_loss=False): with mask[b,t] = t < lens[b]
 * (lens NULL = all ones), loss += inv_denom * sum mask * (pred - ref)^2 and (optional)
 * g_err = 2 * inv_denom * mask * (pred - ref); the caller passes inv_denom = 1 / sum(mask). */
int f2g_masked_mse(float* loss, float* g_err, const float* pred, const float* ref, int32_t B,
                   int32_t T, const int32_t* lens, float inv_denom, f2g_stream_t stream);
/* L1 terms over a (rows, cols) view with row stride ld (a, b, gb share the layout):
 *   loss += w * sum |f(a) - f(b)|,  f = log(max(., clip)) if clip > 0 (gan.py:89-99 with
 *   utils.py:221-232) else identity (gan.py:77-87);  gb = -(w * wdev[0]) * sign(.) * f'(b).
 * loss or gb may be NULL; wdev (device scalar, may be NULL = 1) carries the upstream gradient so
 * that backward never synchronises with the host. */
int f2g_l1_loss(float* loss, float* gb, const float* a, const float* b, int32_t rows, int32_t cols,
                int64_t ld, float w, float clip, const float* wdev, f2g_stream_t stream);
/* Hinge terms (gan.py:57-75): loss += w * sum relu(1 + sgn*s); gs = (w*wdev[0])*sgn where active. */
int f2g_hinge_loss(float* loss, float* gs, const float* s, int64_t n, float sgn, float w,
                   const float* wdev, f2g_stream_t stream);
/* DiscriminatorR input conditioning (discriminators.py:186-190): y = 0.8*(x-mean)/(max|x-mean|+1e-9)
 * per row; stats (rows,3) = {mean, peak, argmax index} kept for backward. */
int f2g_peaknorm_fwd(float* y, float* stats, const float* x, int32_t rows, int32_t T,
                     f2g_stream_t stream);
int f2g_peaknorm_bwd(float* gx, const float* gy, const float* x, const float* stats, int32_t rows,
                     int32_t T, f2g_stream_t stream);
/* Leaky-ReLU backward in place over a (rows, cols, ld) view (discriminators.py:94,205), with the
 * feature-matching gradient folded in first: g += (w*wdev[0]) * sign(y_act - f_real) (gan.py:77-87);
 * then g *= (y_act > 0 ? 1 : slope).  f_real may be NULL. */
int f2g_lrelu_bwd(float* g, const float* y_act, const float* f_real, float w, const float* wdev,
                  float slope, int32_t rows, int32_t cols, int64_t ld, f2g_stream_t stream);
/* MPD input shaping (discriminators.py:82-90): right reflect-pad to a multiple of p and lay the
 * (B,1,T'/p,p) image out channels-last per column: out[((b*p + w)*H + h)] = x[b, h*p + w].
 * Backward folds it back (gx +=). */
/* The same on a (rows, C) map (C, ld multiples of 4) fused with the column sums of the result:
 * colsum[c] += sum_r g[r,c] after the update -- the bias gradient of the conv that produced y_act
 * (one pass over g instead of lrelu_bwd + colsum).  colsum may be NULL. */
int f2g_lrelu_bwd_colsum(float* g, const float* y_act, const float* f_real, float w,
                         const float* wdev, float slope, int32_t rows, int32_t C, int64_t ld,
                         float* colsum, f2g_stream_t stream);
/* Batched real FFT of STFT frames through LDS butterflies (fft.hip; modules.py:69-78 torch.stft,
 * modules.py:106-115 torch.istft) for n_fft = 64..4096 (power of two).  rows =
 * items * F frames; frame m of an item is the n_fft samples x[item*x_stride + m*hop + n] of the
 * REFLECT-PADDED signal (f2g_reflect_pad).  spec rows are planar [Re(0..N/2) | Im(0..N/2)] or
 * interleaved [Re0, Im0, ...].
 *   mode 0: spec = FFT(window * frame)                          (replaces the windowed-DFT GEMM)
 *   mode 1: frames[row, n] = window[n] * Re sum_{k<=N/2} (spec_r[k] + i spec_i[k]) e^{+i theta}
 *           (the gradient of the frames given the gradient of the stored bins)
 *   mode 2: the iSTFT's windowed inverse transform (what f2g_istft_ola overlap-adds):
 *           frames[row, n] = window[n] (1/N) Re sum_{k<=N/2} c_k (spec_r[k] + i spec_i[k]) e^{+i theta},
 *           c_0 = c_{N/2} = 1, else 2; spec_i[0] and spec_i[N/2] are ignored, as torch.istft does
 *   mode 3: its adjoint (gradient of the bins given the gradient g of those frames, read from
 *           `frames`): spec_r[k] = (c_k/N) sum_n window[n] g[n] cos, spec_i[k] = -(c_k/N) sum_n
 *           window[n] g[n] sin, spec_i[0] = spec_i[N/2] = 0
 * twiddle: n_fft/2 pairs (cos, -sin)(2 pi j / n_fft). */
typedef struct {
  const float* x;
  int64_t x_stride;
  int32_t hop, n_fft, F, rows;
  const float* window;
  const float* twiddle;
  float* spec;
  int64_t ld_spec;
  int32_t interleaved; /* bit 0: interleaved rows (else planar); bit 1: spec is a bf16 tensor (planar
                        * rows only, ld_spec in elements): the operand format of the bf16 GEMMs */
  int32_t spec_cols;   /* modes 0 / 3: columns [n_fft + 2, spec_cols) of every row are written as
                        * zeros (0: leave them alone) */
  float* frames;
  int64_t ld_frames;
  int32_t reflect_T;   /* mode 0: > 0 = x is the UNPADDED signal (items of reflect_T samples, x_stride
                        * apart) and the kernel applies torch.stft's center / reflect padding itself:
                        * frame m, sample n reads x[reflect(m*hop + n - n_fft/2)] (needs
                        * reflect_T > n_fft/2); 0 = x is already padded (f2g_reflect_pad) */
  int32_t _pad;
} f2g_fft_desc;
int f2g_fft_frames(const f2g_fft_desc* d, int32_t mode, f2g_stream_t stream);

/* out (B, Tp) = reflect padding of x (B, T) by `pad` samples on both sides (torch.stft center=True,
 * modules.py:69-78), zeros from T + 2*pad to Tp (row length, a multiple of 4): the STFT framing
 * GEMM then reads plain overlapping rows (frame m = samples [m*hop, m*hop + n_fft) of a row). */
int f2g_reflect_pad(float* out, const float* x, int32_t B, int32_t T, int32_t pad, int32_t Tp,
                    f2g_stream_t stream);
int f2g_period_fold(float* out, const float* x, int32_t B, int32_t T, int32_t p, int32_t H,
                    f2g_stream_t stream);
int f2g_period_fold_bwd(float* gx, const float* gout, int32_t B, int32_t T, int32_t p, int32_t H,
                        int32_t accumulate, f2g_stream_t stream);
/* x = log(max(x, clip)) in place (utils.py:221-232 safe_log) */
int f2g_log_clip(float* x, int64_t n, float clip, f2g_stream_t stream);
/* fill */
int f2g_fill(float* x, float v, int64_t n, f2g_stream_t stream);
/* A table of small, mutually INDEPENDENT re-layout operations as one launch (csrc/multi.hip): what rebuilding the
 * derived weight images of a sub-model after an optimizer step consists of (reference: the optimizer writes
 * every parameter each step, optim.py:451-507, so transposes / window-major copies / operand images cannot be
 * kept across steps).  kind: F2G_MULTI_FILL     out[0..n) = v            n = n[0] | n[1] << 32, v = bits s[0]
 *                          F2G_MULTI_PERMUTE4 f2g_permute4             n = dims, s = input strides
 *                          F2G_MULTI_COPY3    f2g_copy3                n = {n0, n1, n2, accumulate},
 *                                                                      s = {so0, so1, si0, si1}
 *                          F2G_MULTI_SPLIT3   f2g_split_bf16x3         n = {rows, K}, s[0] = ld
 *                          F2G_MULTI_SPLIT3G  the same pieces in MFMA FRAGMENT order (f2g_operand.split = 4;
 *                                             rows % 32 == 0)                  n = {rows, K}, s[0] = ld
 * `blocks` = blocks of 256 threads the entry gets (>= 1; its elements are walked grid-stride). */
enum { F2G_MULTI_FILL = 0, F2G_MULTI_PERMUTE4 = 1, F2G_MULTI_COPY3 = 2, F2G_MULTI_SPLIT3 = 3, F2G_MULTI_SPLIT3G = 4 };
#define F2G_MULTI_MAX 48
typedef struct {
  void* out;
  const void* in;
  int32_t kind;
  int32_t blocks;
  int32_t n[4];
  int64_t s[4];
} f2g_multi_entry;
typedef struct {
  int32_t n;
  int32_t _pad;
  f2g_multi_entry e[F2G_MULTI_MAX];
} f2g_multi_desc;
int f2g_multi(const f2g_multi_desc* d, f2g_stream_t stream);
/* Gradient-exchange arenas of the data-parallel step (reference finetune.py:913-915: DDP with
 * find_unused_parameters; here flow2gan_amd/dist.py): an arena is [n gradients | nflags "used" flags], summed
 * over the ranks by ONE all-reduce.  f2g_bucket_arm zeroes the gradients and sets every flag to 1 (one launch
 * per arena and step: the common case -- every parameter of the bucket got a gradient -- then needs nothing
 * in front of the collective); f2g_scale is the 1 / world behind it (x *= s). */
int f2g_bucket_arm(float* flat, int64_t n, int32_t nflags, f2g_stream_t stream);
int f2g_scale(float* x, float s, int64_t n, f2g_stream_t stream);
/* Zero padding carried by the data: buf is (nseq, rows_per_seq, C) channels-last, rows [0, lo) and
 * [rows_per_seq - hi, rows_per_seq) of every sequence are set to 0 (C % 4 == 0).  Conv inputs
 * laid out this way (the reference's `padding=` of discriminators.py:65-76 as halo rows) are read
 * by f2g_gemm as plain strided windows, without bounds tests in the K loop. */
int f2g_zero_halo(float* buf, int32_t nseq, int32_t rows_per_seq, int32_t C, int32_t lo,
                  int32_t hi, f2g_stream_t stream);

/* ---- first layer of a period discriminator (discriminators.py:65-67,92-94: Conv2d(1, 32, (5, 1),
 * stride (3, 1), padding (2, 0)) + leaky ReLU on the folded waveform) as HBM-stream kernels: as a
 * K = 5 GEMM it ran on element-wise loaders.  x: (S, H) floats; y: the 32-channel map in the halo
 * layout (S, Hout + 2*halo, 32), Hout = (H - 1) / 3 + 1; w: [32][5] (= the checkpoint's (32,1,5,1)).
 *   fwd  : y[s, halo + h, :] = lrelu(bias + sum_j w[:, j] x[s, 3h + j - 2])      (halo rows untouched)
 *   wgrad: gw[32][5] += sum_{s,h} y[s, halo + h, :] (x) x[s, 3h + j - 2]       (y = gradient map, read)
 *   dgrad: gx[s, i] = sum_{co, j} y[s, halo + (i + 2 - j)/3, co] w[co][j]       (y = gradient map, read) */
typedef struct {
  const float* x;
  int32_t S, H, Hout, halo;
  const float* w;
  const float* bias; /* fwd; may be NULL */
  float slope;       /* fwd: leaky-ReLU slope (0 = plain ReLU ... use 1 for none) */
  int32_t _pad;
  float* y;
} f2g_mpd0_desc;
int f2g_mpd0_fwd(const f2g_mpd0_desc* d, f2g_stream_t stream);
int f2g_mpd0_wgrad(const f2g_mpd0_desc* d, float* gw, f2g_stream_t stream);
int f2g_mpd0_dgrad(const f2g_mpd0_desc* d, float* gx, f2g_stream_t stream);

/* ---- last layer of a period discriminator (discriminators.py:76,97-99: Conv2d(1024, 1, (3, 1), padding
 * (1, 0))) as HBM-stream kernels over the 1024-channel map y (halo layout (S, H + 2*halo, 1024), zero
 * halo rows, halo >= 1); w: [3][1024] (tap-major); out / g: (S*H) scores / their gradient.
 *   fwd  : out[s*H + h] = bias + sum_j <y[s, halo + h + j - 1, :], w[j]>
 *   dgrad: y[s, halo + h, :] = g[s,h+1] w[0] + g[s,h] w[1] + g[s,h-1] w[2]      (y = gradient map, written)
 *   wgrad: gw[3][1024] += sum_{s,h} g[s,h] y[s, halo + h + j - 1, :]           (y = forward map, read) */
typedef struct {
  float* y;
  int32_t S, H, halo, _pad;
  const float* w;
  const float* bias;
  float* out;
  const float* g;
  /* dgrad only (all optional): leaky-ReLU backward of the 1024-channel layer the gradient lands on -- with
   * y5 = mask_src at the output's own offsets, v = (v + fm_w * fm_wdev[0] * sign(y5 - fm_ref)) * (y5 > 0 ? 1 :
   * mask_slope) (the fm term only with fm_ref) --, colsum[1024] += column sums of the stored gradient (that
   * layer's bias gradient), x3_out = the f2g_split_bf16x3 flat image of the buffer y points into (f2g_epilogue) */
  const float* mask_src;
  const float* fm_ref;
  const float* fm_wdev;
  float mask_slope;
  float fm_w;
  float* colsum;
  void* x3_out;
} f2g_mpdpost_desc;
int f2g_mpdpost_fwd(const f2g_mpdpost_desc* d, f2g_stream_t stream);
int f2g_mpdpost_dgrad(const f2g_mpdpost_desc* d, f2g_stream_t stream);
int f2g_mpdpost_wgrad(const f2g_mpdpost_desc* d, float* gw, f2g_stream_t stream);

/* ---- direct LDS-tiled conv for the MRD band layers (discriminators.py:171-181): Conv2d(32, 32,
 * (3, 9), stride (1, 2), padding (1, 4)) + bias + leaky ReLU on channels-last images.
 * x: (S, H, Win, 32), y: (S, H, Wout, 32) with Wout = (Win - 1) / 2 + 1; strides in floats
 * (pixel pitch is 32); w: [32][27][32] = (Cout, kh*kw, Cin) as f2g_gemm's forward weights. */
typedef struct {
  const float* x;
  int64_t x_seq, x_line;
  int32_t S, H, Win, Wout;
  const float* w;
  const float* bias;  /* may be NULL */
  float lrelu_slope;  /* 0 = none */
  /* fwd / dgrad: 0 = exact fp32 MFMA; 1 = split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulation):
   * `w` is then the f2g_split_bf16 image of the same weight matrix, the patch is split while it
   * is staged in LDS; wgrad splits both staged operands (bf16 planes read with
   * ds_read_b64_tr_b16: its reduction runs over pixels).
   * 3 = fp32-CLASS products on the bf16 pipe (three bf16 pieces per value, six MFMAs per product, error
   * <= ~2^-23 per product: f2g_gemm_desc.precision 3; conv32x6.hip): `w` is the f2g_split_bf16x3 image
   * of the weight matrix -- fwd: of the packed (32, 27*32) matrix (ld = K = 864); dgrad: of the 27
   * transposed tiles as an (864, 32) matrix (ld = K = 32) --, patch / gradient tiles are split while they
   * are staged; wgrad takes the fp32 operands as they are. */
  int32_t precision;
  float* y;
  int64_t y_seq, y_line;
  /* dgrad only: leaky-ReLU backward of the layer below fused into the store (see f2g_epilogue):
   * mask_src / fm_ref have y's layout; colsum[32] += column sums of the stored gradient */
  const float* mask_src;
  const float* fm_ref;
  const float* fm_wdev;
  float mask_slope;
  float fm_w;
  float* colsum;
} f2g_conv32_desc;
int f2g_conv32_s2_fwd(const f2g_conv32_desc* d, f2g_stream_t stream);
/* Data gradient of that layer as a direct transposed convolution (discriminators.py:171-181
 * backward): x = gradient of the layer's pre-activation (S, H, Wout, 32), y = gradient of its input
 * (S, H, Win, 32, fully overwritten), w = the weights as 27 transposed tiles [tap][ci][co]
 * (tap = kh*9 + kw); bias / lrelu_slope unused. */
int f2g_conv32_s2_dgrad(const f2g_conv32_desc* d, f2g_stream_t stream);
/* Weight gradient of that layer: x = layer input (S, H, Win, 32), y = gradient of the
 * pre-activation (S, H, Wout, 32) (read), gw (32, 27*32) [co][tap][ci] accumulated atomically. */
int f2g_conv32_s2_wgrad(const f2g_conv32_desc* d, float* gw, f2g_stream_t stream);
/* Fifth layer of a band stack, Conv2d(32, 32, (3, 3), padding (1, 1)), stride 1 (discriminators.py:171-181,
 * last entry) as a direct kernel, fp32 class only (precision must be 3; the exact-fp32 step keeps the
 * implicit GEMM): x (S, H, W, 32) and y (S, H, W, 32) with explicit sequence / line strides (Win == Wout
 * == W <= 112; the forward writes its band's slice of the concatenated map), w = the f2g_split_bf16x3 image of the
 * packed (32, 9*32) matrix [co][tap][ci], + bias + leaky ReLU.  The DATA GRADIENT of this layer is the same
 * call over the gradient map with w = the image of [ci][8 - tap][co] (taps flipped, channel matrix
 * transposed) and no bias / slope; there mask_src (+ mask_slope) / colsum apply the leaky-ReLU backward of
 * the layer below and leave the column sums of the stored gradient as in f2g_conv32_s2_dgrad (fm_ref must
 * be NULL). */
int f2g_conv33_fwd(const f2g_conv32_desc* d, f2g_stream_t stream);
/* Weight gradient of that layer (discriminators.py:171-181 backward), fp32 class (precision = 3): x = the layer's
 * input (S, H, W, 32), y = the gradient of its pre-activation (S, H, W, 32) -- both optionally strided slices of
 * wider maps through x_line / x_seq, y_line / y_seq --, gw (32, 9 * 32) [co][tap][ci] += sum over the pixels; W <= 112.
 * The caller zero-initialises gw (blocks accumulate atomically).  w / bias / slope / mask fields unused. */
int f2g_conv33_wgrad(const f2g_conv32_desc* d, float* gw, f2g_stream_t stream);

/* First layer of every MRD band stack, Conv2d(2, 32, (3, 9), stride 1, padding (1, 4))
 * (discriminators.py:171,195-203), as direct kernels (conv2ch.hip).  The input is a frequency band
 * of the interleaved complex spectrogram: image (S, H frames, W bins, 2) with explicit sequence /
 * line strides (floats); the 32-channel side is dense (S*H*W, 32).
 *   fwd  : y = lrelu(conv(x; w) + bias),  w = (32, 27*2) window-major, channel-minor
 *   wgrad: gw (32, 27*2) += sum_px y[px, :]^T (x) patch(px)      (y = gradient of the pre-activation)
 *   dgrad: gx (same layout as x) = transposed conv of y with wt = [27 taps][2][32]   (overwrites) */
typedef struct {
  const float* x;
  int64_t x_seq, x_line;
  int32_t S, H, W, _pad;
  const float* w;
  const float* bias;
  float lrelu_slope;
  int32_t _pad2;
  float* y;          /* fwd: output; wgrad / dgrad: the gradient map (read) */
  float* gw;
  const float* wt;
  float* gx;
  int64_t gx_seq, gx_line;
} f2g_conv2ch_desc;
int f2g_conv2ch_fwd(const f2g_conv2ch_desc* d, f2g_stream_t stream);
int f2g_conv2ch_wgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream);
int f2g_conv2ch_dgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream);
/* conv_post of an MRD resolution, Conv2d(32, 1, (3, 3), padding (1, 1)) (discriminators.py:184),
 * same descriptor: x = (S, H, W, 32) dense feature map, w = (9, 32) tap-major weights.
 *   fwd  : y (S*H*W) = conv + bias[0]
 *   wgrad: gw (9*32) += sum_px y[px] * window(px)        (y = gradient of the scores)
 *   dgrad: gx (S*H*W, 32) = transposed conv of y          (overwrites; x unused) */
int f2g_convpost_fwd(const f2g_conv2ch_desc* d, f2g_stream_t stream);
int f2g_convpost_wgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream);
int f2g_convpost_dgrad(const f2g_conv2ch_desc* d, f2g_stream_t stream);

/* ---- on-device data front end (dataset.py:122-175).  x: (B, C, T) crops with
 * explicit item / channel strides (floats); lens[b] = valid samples of item b.
 *   f2g_wave_stats: stats[2b] = sqrt(mean_{c,t} x^2) (silence test, dataset.py:130-131),
 *                   stats[2b+1] = max_t |mean_c x| (peak of the mono mix);
 *   f2g_wave_gain : out[b,t] = mean_c x[b,c,t] * target_peak[b] / peak[b] for t < lens[b], 0 beyond
 *                   (mono mix :160-162, sox `norm <dB>` :164-168 with target_peak = 10^(dB/20),
 *                   zero padding of pad_sequence :43); target_peak NULL or <= 0: level untouched.
 * Resampling (dataset.py:170-173) is an f2g_gemm over a windowed operand with the polyphase sinc
 * kernel as weights (flow2gan_amd/frontend.py:resample). */
int f2g_wave_stats(const float* x, int64_t item_stride, int64_t ch_stride, int32_t B, int32_t C,
                   const int32_t* lens, float* stats, f2g_stream_t stream);
int f2g_wave_gain(float* out, int64_t ldo, const float* x, int64_t item_stride, int64_t ch_stride,
                  int32_t B, int32_t C, int32_t T, const int32_t* lens, const float* stats,
                  const float* target_peak, f2g_stream_t stream);

/* ---- ScaledAdam (reference optim.py:125-255 basic/scaling/momentum steps,
 * :451-507 step, :509-619 clipping) as multi-tensor launches over device-resident tables.
 * The caller owns every buffer: params p, grads g (NULL = zeros, optim.py:111-113), second moment
 * v and momentum m per tensor; stats (3 floats per tensor), tstate (F2G_SADAM_TSTATE per tensor:
 * param_rms, scale_exp_avg_sq, scale_grads[8]), gstate (F2G_SADAM_GSTATE per group: model_norms
 * ring, threshold, flags), coef (F2G_SADAM_NCOEF per tensor).  One optimizer step =
 * stats -> prepare (once per group) -> update; group->step is the number of steps taken before. */
#define F2G_SADAM_NCOEF 12
#define F2G_SADAM_TSTATE 10
#define F2G_SADAM_GSTATE 1028
typedef struct {
  float* p;
  const float* g;
  float* v;
  float* m;
  int64_t numel;
  int32_t group;
  int32_t is_scalar; /* numel == 1: no learned scale, lr * scalar_lr_scale, clamp to scalar_max */
} f2g_sadam_tensor;
typedef struct {
  int32_t tensor; /* index into the tensor table */
  int32_t count;  /* <= f2g_sadam_chunk_elems() */
  int64_t offset; /* first element of the chunk inside the tensor */
} f2g_sadam_chunk;
typedef struct {
  float lr, beta1, beta2, scalar_lr_scale, eps, param_min_rms, param_max_rms, scalar_max;
  float clipping_scale;           /* <= 0: no clipping (optim.py:529) */
  int32_t size_update_period;     /* <= 8 */
  int32_t clipping_update_period; /* <= 1024 */
  int32_t step;
  int32_t first, count;           /* the group's tensors: table[first .. first+count) */
} f2g_sadam_group;
int32_t f2g_sadam_chunk_elems(void);
/* tensors / chunks: DEVICE pointers to the tables; group: HOST pointer (passed by value) */
int f2g_sadam_stats(const f2g_sadam_tensor* tensors, const f2g_sadam_chunk* chunks,
                    int32_t nchunks, float* stats, int32_t ntensors, f2g_stream_t stream);
int f2g_sadam_prepare(const f2g_sadam_tensor* tensors, const f2g_sadam_group* group,
                      const float* stats, float* tstate, float* gstate, float* coef,
                      f2g_stream_t stream);
int f2g_sadam_update(const f2g_sadam_tensor* tensors, const f2g_sadam_chunk* chunks,
                     int32_t nchunks, const float* coef, f2g_stream_t stream);

/* ---- fused pointwise MLP of a ConvNeXt block, bf16 operands / fp32 accumulate (csrc/fusedmlp.hip) ----
 * Replaces pwconv1 -> PReLU -> pwconv2 and the residual of ConvNeXtBlock.forward
 * (flow2gan/models/modules.py:487-495) for inference in the plain-bf16 mode (BASELINE config 2):
 *     out[r, :] = W2 . PReLU(W1 . z[r, :] + b1; alpha) + b2 + gamma * res[r, :]
 * with the (rows, H) hidden activation kept on chip (a 128-column slab at a time in LDS, the output
 * tile in accumulator registers).  z: bf16 (rows, C), row stride ldz elements (what f2g_dwnorm_fwd
 * writes with z_format = 2); wp: the weights as f2g_mlp_pack lays them out; b1 / alpha: (H) fp32;
 * b2 / gamma: (C) fp32 (b1, b2, gamma may be NULL; res NULL = no residual); out fp32 (rows, C).
 * Supported: C in {384, 512, 768}, H a multiple of 128 (f2g_fused_mlp_ok). */
typedef struct {
  const void* z;
  int64_t ldz;
  const void* wp;
  const float* b1;
  const float* alpha;
  const float* b2;
  const float* res;
  int64_t ldres;
  const float* gamma;
  float* out;
  int64_t ldo;
  int32_t rows, C, H;
  int32_t parts; /* 0: the library decides; n >= 1: cut the hidden dimension between n blocks per row
                  * tile (n > 1: partial output tiles are added atomically onto a zeroed `out`) */
} f2g_fused_mlp_desc;
int f2g_fused_mlp_ok(int32_t C, int32_t H);
/* dst (2*C*H bf16 = 4*C*H bytes) = W1 (H, C; row stride ld1) and W2 (C, H; row stride ld2), fp32,
 * rounded to bf16 and re-ordered into one contiguous stream per wave of the fused kernel, in MFMA
 * B-fragment order (1 KiB = rows n0..n0+31 x 16 consecutive k; lane l holds k = 8*(l>>5)..+7 of
 * row l&31): wave w, slab s: C/16 fragments of W1 rows s*128 + w*32..+32, then for each of the 8
 * k steps of the slab the C/128 fragments of W2 rows w*C/4.. over hidden columns s*128 + 16*k2... */
int f2g_mlp_pack(void* dst, const float* w1, int64_t ld1, const float* w2, int64_t ld2, int32_t C,
                 int32_t H, f2g_stream_t stream);
int f2g_fused_mlp(const f2g_fused_mlp_desc* d, f2g_stream_t stream);
/* The whole ConvNeXt block of modules.py:473-495 in ONE launch (plain-bf16 inference): the z tile of
 * a row block is computed in the kernel's prologue from the residual stream x -- depthwise conv (K = 7),
 * BiasNorm, condition add, time scale, exactly f2g_dwnorm_fwd's arithmetic, rounded to bf16 into LDS --
 * and feeds the fused MLP above; neither z nor the hidden activation exists in HBM.  `dw` names x,
 * the masks and the dwnorm parameters (its z / rstd / z_format fields are ignored), `mlp` the rest
 * (its z / ldz are ignored; mlp->rows must equal dw->B * dw->F, mlp->res is normally dw->x). */
int f2g_fused_block(const f2g_dwnorm_fwd_desc* dw, const f2g_fused_mlp_desc* mlp, f2g_stream_t stream);
/* n <= 4 independent blocks (dw[i], mlp[i]) in ONE launch: the same layer of a decoder's Fourier branches
 * (modules.py:600-612 runs them one after the other).  Their row tiles form one grid in order of
 * decreasing cost per tile, so the in-order workgroup dispatch schedules longest-first and the branches
 * share the chip without waiting for each other's launches.  Results are those of n f2g_fused_block
 * calls (row by row the same arithmetic; tile heights, hence kernel instances, may differ). */
int f2g_fused_block_multi(const f2g_dwnorm_fwd_desc* dw, const f2g_fused_mlp_desc* mlp, int32_t n,
                          f2g_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOW2GAN_HIP_H */
