"""Fused forward/backward schedules of the Flow2GAN generator on the HIP kernels.

Each `torch.autograd.Function` below is one coarse node (condition encoder, per-branch condition
path, one model evaluation = 3 branches, the stage-1 loss) whose forward and backward are explicit
kernel sequences over channels-last buffers; PyTorch supplies memory, streams and the autograd
graph between the nodes only.  Math and reference line numbers: SURVEY.md appendix A.

Layout: a reference tensor (B, C, F) is a rows matrix (B*F, C); audio stays (B, T).
"""
from __future__ import annotations

import os
import random
from typing import List, Optional

import torch

from . import ops
from ._opts import opt
from .models.modules import dft_matrices
from .ops import gemm, mat, win1d

LIMIT_PROB = 0.6  # reference modules.py:260


def _limit_draw(training: bool) -> bool:
    """One Python-RNG draw per BiasNorm / ChannelScale call, as the reference does
    (modules.py:259-270), so that seeded runs make the same choices in the same order."""
    return bool(training and random.random() < LIMIT_PROB)


def frames_lens(audio_lens_cpu: Optional[List[int]], hop: int, device):
    if audio_lens_cpu is None:
        return None
    return torch.tensor([1 + int(l) // hop for l in audio_lens_cpu], dtype=torch.int32,
                        device=device)


# =====================================================================================
# ConvNeXt block (shared by the condition encoder and the three decoders)
# =====================================================================================
BLOCK_KEYS = ("dwconv.weight", "dwconv.bias", "norm.log_scale", "norm.bias", "pwconv1.weight",
              "pwconv1.bias", "act.weight", "pwconv2.weight", "pwconv2.bias",
              "residual_scale.scale")


def block_params(blk) -> list:
    return [blk.dwconv.weight, blk.dwconv.bias, blk.norm.log_scale, blk.norm.bias,
            blk.pwconv1.weight, blk.pwconv1.bias, blk.act.weight, blk.pwconv2.weight,
            blk.pwconv2.bias, blk.residual_scale.scale]


class _Blk:
    """Plain view of one block's parameter tensors + sizes."""

    __slots__ = ("w_dw", "b_dw", "log_scale", "beta", "w1", "b1", "alpha", "w2", "b2", "gamma",
                 "C", "H", "K")

    def __init__(self, p: list):
        (self.w_dw, self.b_dw, self.log_scale, self.beta, self.w1, self.b1, self.alpha, self.w2,
         self.b2, self.gamma) = p
        self.C = self.w_dw.shape[0]
        self.K = self.w_dw.shape[2]
        self.H = self.w1.shape[0]


_APPLY_GRAD = True

# Set by dist.GradReducer.prepare while a gradient exchange is armed (None otherwise): coarse nodes
# whose backward runs several independent launch lanes hand a finished lane's parameter gradients
# over through it (dist._Sink) instead of returning them when the whole node is done.
GRAD_SINK = None


def sink_register(params):
    """Forward side of deliver_grads: returns (sink, key) or None."""
    sink = GRAD_SINK
    if sink is None or not _APPLY_GRAD:
        return None
    key = sink.add_use(params)
    return None if key is None else (sink, key)


def deliver_grads(ticket, params, grads):
    """Backward side: with a ticket from sink_register, accumulate `grads` into the armed arenas on
    the current stream and return Nones for autograd; without one, return `grads` unchanged."""
    if ticket is None:
        return grads
    sink, key = ticket
    if GRAD_SINK is not sink:       # the exchange this forward announced itself to is over
        return grads
    sink.deliver(key, params, grads)
    return [None] * len(grads)


class GradAwareFunction(torch.autograd.Function):
    """autograd.Function whose forward can tell whether a backward can follow: forward itself always
    runs with grad mode off, and ctx.needs_input_grad mirrors the inputs' requires_grad even when
    the caller is under torch.no_grad() (inference over parameters that require grad)."""

    @classmethod
    def apply(cls, *args):
        global _APPLY_GRAD
        prev = _APPLY_GRAD
        _APPLY_GRAD = torch.is_grad_enabled()
        try:
            return super().apply(*args)
        finally:
            _APPLY_GRAD = prev


def _keep(ctx) -> bool:
    """Does this forward have to save activations for a backward pass?"""
    return _APPLY_GRAD and any(ctx.needs_input_grad)


def block_fwd(bp: _Blk, x, B, F, lens, cproj=None, ldcp=0, Fc=0, up=1, cp_off=0, te=None, ldte=0,
              te_off=0, keep: bool = True):
    """x (B*F, C) -> new x.  Returns (x_out, z, a) (z, a kept for backward)."""
    dev = x.device
    rows, Cc, Hh = B * F, bp.C, bp.H
    fmt = ops.operand_formats_ok(Cc, Hh)
    if fmt == 2 and not keep and ops.FUSED_BLOCK and bp.K == 7 and ops.fused_mlp_applies(Cc, Hh) \
            and x.stride(0) % 4 == 0:
        # plain-bf16 inference (BASELINE config 2): the whole block in one launch -- z is computed in
        # the fused kernel's prologue and lives in LDS only, the hidden activation on chip
        out = ops.empty(rows, Cc, device=dev)
        ops.fused_block(x, B, F, Cc, bp.K, lens, bp.w_dw, bp.b_dw, bp.beta, bp.log_scale.reshape(1),
                        ops.mlp_pack(bp.w1, bp.w2), bp.b1, bp.alpha, bp.b2, bp.gamma.reshape(Cc), out,
                        Hh, cproj, ldcp, Fc, up, cp_off, te, ldte, te_off)
        return out, None, (None, None, None, 0)
    if fmt == 2 and not keep:
        # (two launches: z and the hidden activation live in HBM as bf16, written by their producers
        # in the layout the lean GEMM kernel reads)
        z = torch.empty(rows, Cc, device=dev, dtype=torch.bfloat16)
        ops.dwnorm_fwd(x, z, B, F, Cc, bp.K, lens, bp.w_dw, bp.b_dw, bp.beta, bp.log_scale.reshape(1),
                       cproj, ldcp, Fc, up, cp_off, te, ldte, te_off, z_format=2)
        if ops.fused_mlp_applies(Cc, Hh):
            # pwconv1 -> PReLU -> pwconv2 + residual in one launch, hidden activation on chip
            out = ops.empty(rows, Cc, device=dev)
            ops.fused_mlp(z, ops.mlp_pack(bp.w1, bp.w2), bp.b1,
                          bp.alpha, bp.b2, x, bp.gamma.reshape(Cc), out, rows, Cc, Hh)
            return out, z, (None, None, None, 0)
        a = torch.empty(rows, Hh, device=dev, dtype=torch.bfloat16)
        gemm(mat(z, rows, Cc, split=2), mat(bp.w1.reshape(Hh, Cc)), a, bias=bp.b1, prelu=bp.alpha,
             split_k=1)
        out = ops.empty(rows, Cc, device=dev)
        gemm(mat(a, rows, Hh, split=2), mat(bp.w2.reshape(Cc, Hh)), out, bias=bp.b2, res=x,
             gamma=bp.gamma.reshape(Cc))
        return out, z, (a, a, None, 0)
    zsplit = 1 if fmt == 1 else 0   # (split-bf16 mode) dwnorm writes z as the image the GEMMs read
    z = ops.empty(rows, Cc, device=dev)
    ops.dwnorm_fwd(x, z, B, F, Cc, bp.K, lens, bp.w_dw, bp.b_dw, bp.beta, bp.log_scale.reshape(1),
                   cproj, ldcp, Fc, up, cp_off, te, ldte, te_off, z_format=zsplit)
    a = ops.empty(rows, Hh, device=dev)
    if keep:
        p = ops.empty(rows, Hh, device=dev)
        # (split-bf16 mode) the images of z and p made for these GEMMs serve the weight gradients
        # of the backward pass as well
        share = ops.split_sharing(z, p)
        with share:
            gemm(mat(z, rows, Cc, split=zsplit), mat(bp.w1.reshape(Hh, Cc)), a, bias=bp.b1,
                 prelu=bp.alpha, prelu_out=p)
            out = ops.empty(rows, Cc, device=dev)
            gemm(mat(p, rows, Hh), mat(bp.w2.reshape(Cc, Hh)), out, bias=bp.b2, res=x,
                 gamma=bp.gamma.reshape(Cc))
        return out, z, (a, p, share, zsplit)
    p = a
    gemm(mat(z, rows, Cc, split=zsplit), mat(bp.w1.reshape(Hh, Cc)), a, bias=bp.b1, prelu=bp.alpha)
    out = ops.empty(rows, Cc, device=dev)
    gemm(mat(p, rows, Hh), mat(bp.w2.reshape(Cc, Hh)), out, bias=bp.b2, res=x,
         gamma=bp.gamma.reshape(Cc))
    return out, z, (a, p, None, zsplit)


def block_bwd(bp: _Blk, x, z, a, gout, B, F, lens, limit_norm: bool, limit_scale: bool,
              cproj=None, ldcp=0, Fc=0, up=1, cp_off=0, te=None, ldte=0, te_off=0, g_cproj=None,
              g_te=None, g_cproj_store=False):
    """Backward of block_fwd.  Destroys z and a (reused as gradient buffers).
    Returns (gx, grads) with grads ordered like BLOCK_KEYS.  g_cproj_store: this block owns the
    zero-filled columns [cp_off, cp_off + C) of g_cproj (ops.dwnorm_bwd)."""
    a, p_act, share, zsplit = a
    if share is None:
        share = ops.split_sharing()
    dev = x.device
    rows, Cc, Hh = B * F, bp.C, bp.H
    # (g_alpha and g_b1 side by side: the partial column sums of the PReLU-backward GEMM are then reduced into
    # both by ONE f2g_colsum, ops.gemm)
    (g_w2, g_b2, g_alpha, g_b1, g_w1, g_beta, g_ls, g_wdw, g_bdw, g_gamma) = ops.zeros_many(
        [(Cc, Hh), (Cc,), (Hh,), (Hh,), (Hh, Cc), (Cc,), (1,), (Cc, 1, bp.K), (Cc,), (Cc, 1)], dev)
    # pwconv2: out = W2 prelu(a) + b2 + gamma*x
    ops.colsum(g_b2, gout, rows, Cc)
    with share, ops.split_sharing(gout):   # one split-bf16 image of gout for both GEMMs
        ops.wgrad(gout, Cc, gout.stride(0), mat(p_act, rows, Hh), g_w2)
        # da = (gout W2) * prelu'(a)   (in place over a), d alpha, d b1
        gemm(mat(gout, rows, Cc), mat(bp.w2.reshape(Cc, Hh)), a, form=1, aux=a, alpha_n=bp.alpha,
             colsum_alpha=g_alpha, colsum=g_b1)
    with share, ops.split_sharing(a):      # a now holds da
        ops.wgrad(a, Hh, a.stride(0), mat(z, rows, Cc, split=zsplit), g_w1)
        # dz = da W1  (z is dead after the weight gradient above: reuse it)
        gemm(mat(a, rows, Hh), mat(bp.w1.reshape(Hh, Cc)), z, form=1)
    share.drop()
    du = ops.empty(rows, Cc, device=dev)
    ops.dwnorm_bwd(x, z, du, B, F, Cc, bp.K, lens, bp.w_dw, bp.b_dw, bp.beta,
                   bp.log_scale.reshape(1), cproj, ldcp, Fc, up, cp_off, te, ldte, te_off,
                   g_cproj=g_cproj, g_te=g_te, g_beta=g_beta, g_log_scale=g_ls,
                   g_cproj_store=g_cproj_store)
    gx = ops.empty(rows, Cc, device=dev)
    ops.dwconv_bwd(du, x, gx, B, F, Cc, bp.K, lens, bp.w_dw, gres=gout, gamma=bp.gamma.reshape(Cc),
                   g_w=g_wdw, g_b=g_bdw, g_gamma=g_gamma)
    if limit_norm:
        ops.limit_grad(g_ls, bp.log_scale.reshape(1), -1.5, 1.5)
    if limit_scale:
        ops.limit_grad(g_gamma, bp.gamma, 0.5, 1.0)
    grads = [g_wdw, g_bdw, g_ls.reshape(()), g_beta, g_w1.reshape(Hh, Cc, 1), g_b1, g_alpha,
             g_w2.reshape(Cc, Hh, 1), g_b2, g_gamma]
    return gx, grads


# =====================================================================================
# Condition encoder  (reference modules.py:498-542; called without mask, generator.py:312,348)
# =====================================================================================
def cond_encoder_params(enc) -> list:
    p = [enc.in_proj.weight, enc.in_proj.bias, enc.in_norm.log_scale, enc.in_norm.bias]
    for blk in enc.blocks:
        p += block_params(blk)
    return p


class CondEncoderFn(GradAwareFunction):
    """mel (B, n_mels, Fm) -> condition rows (B*Fm, channels)."""

    @staticmethod
    def forward(ctx, mel, training: bool, lens_cpu, *params):
        """lens_cpu: valid frames per item (the `mask` of modules.py:524-542; the generator calls
        the encoder without one, generator.py:312) or None."""
        dev = mel.device
        B, nm, Fm = mel.shape
        lens = None if lens_cpu is None else torch.tensor([int(v) for v in lens_cpu], dtype=torch.int32,
                                                          device=dev)
        w_in, b_in, ls_in, beta_in = params[:4]
        Cc = w_in.shape[0]
        nblk = (len(params) - 4) // len(BLOCK_KEYS)
        blks = [_Blk(list(params[4 + i * 10: 14 + i * 10])) for i in range(nblk)]
        rows = B * Fm
        melr = ops.empty(rows, nm, device=dev)
        ops.bct_to_rows(melr, mel.contiguous(), B, nm, Fm)
        # (Cout, Cin, 3) -> [Cout][tap][Cin] so that the window of a frame is contiguous
        wp = ops.empty(Cc, 3 * nm, device=dev)
        ops.permute4(wp, w_in, (Cc, 3, nm, 1), (nm * 3, 1, 3, 0))
        h0 = ops.empty(rows, Cc, device=dev)
        gemm(win1d(melr, B, Fm, nm, Fm, 1, 1, 3), mat(wp), h0, bias=b_in)
        flags = [_limit_draw(training)]
        x = ops.empty(rows, Cc, device=dev)
        ops.biasnorm_fwd(h0, x, rows, Cc, beta_in, ls_in.reshape(1))
        saved = []
        for bp in blks:
            fn = _limit_draw(training)
            y, z, a = block_fwd(bp, x, B, Fm, lens, keep=_keep(ctx))
            flags.append((fn, _limit_draw(training)))
            saved.append((x, z, a))
            x = y
        if _keep(ctx):
            ctx.saved = (melr, wp, h0, saved)
            ctx.params = params
            ctx.dims = (B, nm, Fm, Cc)
            ctx.flags = flags
            ctx.lens = lens
        return x

    @staticmethod
    def backward(ctx, g):
        melr, wp, h0, saved = ctx.saved
        params = ctx.params
        B, nm, Fm, Cc = ctx.dims
        dev = g.device
        rows = B * Fm
        w_in, b_in, ls_in, beta_in = params[:4]
        nblk = len(saved)
        blks = [_Blk(list(params[4 + i * 10: 14 + i * 10])) for i in range(nblk)]
        g = g.contiguous()
        grads_blocks = [None] * nblk
        for i in reversed(range(nblk)):
            x, z, a = saved[i]
            fn, fs = ctx.flags[1 + i]
            g, gb = block_bwd(blks[i], x, z, a, g, B, Fm, ctx.lens, fn, fs)
            grads_blocks[i] = gb
        g_beta, g_ls, g_bin, g_wp = ops.zeros_many([(Cc,), (1,), (Cc,), (Cc, 3 * nm)], dev)   # (one fill)
        gh0 = ops.empty(rows, Cc, device=dev)
        ops.biasnorm_bwd(h0, g, gh0, rows, Cc, beta_in, ls_in.reshape(1), g_beta, g_ls)
        if ctx.flags[0]:
            ops.limit_grad(g_ls, ls_in.reshape(1), -1.5, 1.5)
        ops.colsum(g_bin, gh0, rows, Cc)
        ops.wgrad(gh0, Cc, gh0.stride(0), win1d(melr, B, Fm, nm, Fm, 1, 1, 3), g_wp)
        g_win = ops.empty(Cc, nm, 3, device=dev)
        ops.permute4(g_win, g_wp, (Cc, nm, 3, 1), (3 * nm, 1, nm, 0))
        g_mel = None
        if ctx.needs_input_grad[0]:
            # d mel (leaf API only: the generator's mel carries no gradient): the k = 3 conv's data gradient
            # = window gradients (rows, 3 * n_mels) folded back over the three taps (row f + tap - 1)
            gw = ops.empty(rows, 3 * nm, device=dev)
            gemm(mat(gh0, rows, Cc), mat(wp), gw, form=1)
            gm = ops.zeros(rows, nm, device=dev)
            for tap in range(3):
                sh = tap - 1
                f0, n = max(0, -sh), Fm - abs(sh)
                if n > 0:
                    ops.copy3(gm, Fm * nm, nm, gw, Fm * 3 * nm, 3 * nm, B, n, nm, accumulate=True,
                              out_offset=(f0 + sh) * nm, in_offset=f0 * 3 * nm + tap * nm)
            g_mel = ops.rows_to_bct(ops.empty(B, nm, Fm, device=dev), gm, B, nm, Fm)
        out = [g_mel, None, None, g_win, g_bin, g_ls.reshape(()), g_beta]
        for gb in grads_blocks:
            out += gb
        ctx.saved = None
        return tuple(out)


# =====================================================================================
# Per-branch condition path: cond_mlp + the 8 stacked cond_proj (modules.py:576-580,448,482,620)
# computed at the condition frame rate (pointwise ops commute with repeat-interleave; rows past
# the condition length are zero rows, exactly what convert_length pads, modules.py:679).
# =====================================================================================
def cond_path_params(dec) -> list:
    p = [dec.cond_mlp[0].weight, dec.cond_mlp[0].bias, dec.cond_mlp[1].weight,
         dec.cond_mlp[2].weight, dec.cond_mlp[2].bias]
    for blk in dec.blocks:
        p += [blk.cond_proj.weight, blk.cond_proj.bias]
    return p


def _stack_rows(ws, cols, dev):
    """Stack L tensors of shape (C, cols[,1]) into (L*C, cols) (cached until a parameter changes)."""
    def build(ts):
        Cc = ts[0].shape[0]
        out = ops.empty(len(ts) * Cc, cols, device=dev)
        for j, w in enumerate(ts):
            ops.copy3(out, 0, cols, w, 0, cols, 1, Cc, cols, out_offset=j * Cc * cols)
        return out
    return ops.derived_multi(list(ws), ("stack_rows", cols), build)


def _stack_vecs(bs, dev):
    def build(ts):
        Cc = ts[0].shape[0]
        out = ops.empty(len(ts) * Cc, device=dev)
        for j, b in enumerate(ts):
            ops.copy3(out, 0, 0, b, 0, 0, 1, 1, Cc, out_offset=j * Cc)
        return out
    return ops.derived_multi(list(bs), "stack_vecs", build)


class CondPathFn(GradAwareFunction):
    """cond rows (B*Fc, Dc) -> cproj_all (B*Fce, nblk*C) with Fce = ceil(F/up) rows per item."""

    @staticmethod
    def forward(ctx, cond, B: int, Fc: int, Fce: int, *params):
        dev = cond.device
        w0, b0, alpha, w2, b2 = params[:5]
        wcs, bcs = list(params[5::2]), list(params[6::2])
        Dc, Hc = w0.shape[1], w0.shape[0]
        Cc, nblk = wcs[0].shape[0], len(wcs)
        if Fce == Fc:
            cext = cond
        else:
            cext = ops.zeros(B * Fce, Dc, device=dev)
            n = min(Fc, Fce)
            ops.copy3(cext, Fce * Dc, Dc, cond, Fc * Dc, Dc, B, n, Dc)
        rows = B * Fce
        keep = _keep(ctx)
        wstack = _stack_rows(wcs, Dc, dev)
        bstack = _stack_vecs(bcs, dev)
        cproj = ops.empty(rows, nblk * Cc, device=dev)
        if (not keep) and ops.GEMM_PRECISION == 2 and ops.BF16_IMAGES and ops.LEAN_SPLIT \
                and Dc % 64 == 0 and Hc % 64 == 0:
            # plain-bf16 inference: the two intermediate results leave their GEMMs as the bf16
            # tensors the next GEMM reads (no conversion launches in between)
            a = torch.empty(rows, Hc, device=dev, dtype=torch.bfloat16)
            gemm(mat(cext, rows, Dc), mat(w0.reshape(Hc, Dc)), a, bias=b0, prelu=alpha, split_k=1)
            cm = torch.empty(rows, Dc, device=dev, dtype=torch.bfloat16)
            gemm(mat(a, rows, Hc, split=2), mat(w2.reshape(Dc, Hc)), cm, bias=b2, split_k=1)
            gemm(mat(cm, rows, Dc, split=2), mat(wstack), cproj, bias=bstack)
            return cproj
        a = ops.empty(rows, Hc, device=dev)
        pact = ops.empty(rows, Hc, device=dev) if keep else a
        gemm(mat(cext, rows, Dc), mat(w0.reshape(Hc, Dc)), a, bias=b0, prelu=alpha,
             prelu_out=pact if keep else None)
        cm = ops.empty(rows, Dc, device=dev)
        gemm(mat(pact, rows, Hc), mat(w2.reshape(Dc, Hc)), cm, bias=b2)
        gemm(mat(cm, rows, Dc), mat(wstack), cproj, bias=bstack)
        if keep:
            ctx.saved = (cext, (a, pact), cm, wstack)
            ctx.params = params
            ctx.dims = (B, Fc, Fce, Dc, Hc, Cc, nblk)
        return cproj

    @staticmethod
    def backward(ctx, g):
        cext, (a, pact), cm, wstack = ctx.saved
        params = ctx.params
        B, Fc, Fce, Dc, Hc, Cc, nblk = ctx.dims
        w0, b0, alpha, w2, b2 = params[:5]
        dev = g.device
        rows = B * Fce
        g = g.contiguous()
        NC = nblk * Cc
        # every gradient accumulator of this node from ONE zeroed allocation (one fill launch)
        g_wstack, g_bstack, g_b2, g_w2, g_alpha, g_b0, g_w0 = ops.zeros_many(
            [(NC, Dc), (NC,), (Dc,), (Dc, Hc), (Hc,), (Hc,), (Hc, Dc)], dev)
        ops.colsum(g_bstack, g, rows, NC)
        ops.wgrad(g, NC, g.stride(0), mat(cm, rows, Dc), g_wstack)
        g_cm = ops.empty(rows, Dc, device=dev)
        gemm(mat(g, rows, NC), mat(wstack), g_cm, form=1)
        ops.colsum(g_b2, g_cm, rows, Dc)
        ops.wgrad(g_cm, Dc, g_cm.stride(0), mat(pact, rows, Hc), g_w2)
        gemm(mat(g_cm, rows, Dc), mat(w2.reshape(Dc, Hc)), a, form=1, aux=a, alpha_n=alpha,
             colsum_alpha=g_alpha, colsum=g_b0)
        ops.wgrad(a, Hc, a.stride(0), mat(cext, rows, Dc), g_w0)
        g_cond = None
        if ctx.needs_input_grad[0]:
            g_cext = ops.empty(rows, Dc, device=dev)
            gemm(mat(a, rows, Hc), mat(w0.reshape(Hc, Dc)), g_cext, form=1)
            if Fce == Fc:
                g_cond = g_cext
            else:
                g_cond = ops.zeros(B * Fc, Dc, device=dev)
                n = min(Fc, Fce)
                ops.copy3(g_cond, Fc * Dc, Dc, g_cext, Fce * Dc, Dc, B, n, Dc)
        out = [g_cond, None, None, None, g_w0.reshape(Hc, Dc, 1), g_b0, g_alpha,
               g_w2.reshape(Dc, Hc, 1), g_b2]
        for j in range(nblk):
            out.append(g_wstack[j * Cc:(j + 1) * Cc].reshape(Cc, Dc, 1))
            out.append(g_bstack[j * Cc:(j + 1) * Cc])
        ctx.saved = None
        return tuple(out)


# =====================================================================================
# One model evaluation = three AudioConvNeXt branches, averaged (generator.py:129-170,
# modules.py:682-721): STFT (windowed-DFT GEMM) -> in_proj -> in_norm -> 8 blocks -> out_proj ->
# mask -> inverse DFT GEMM -> overlap-add, accumulated into one (B, T) prediction.
# =====================================================================================
def decoder_params(d) -> list:
    """Everything of a ConvNeXtDecoder outside its condition path (cond_path_params)."""
    p = [d.in_proj.weight, d.in_proj.bias, d.in_norm.log_scale, d.in_norm.bias,
         d.time_mlp[0].weight, d.time_mlp[0].bias, d.time_mlp[2].weight, d.time_mlp[2].bias,
         d.out_proj.weight, d.out_proj.bias]
    for blk in d.blocks:
        p += block_params(blk) + [blk.time_embed_proj.weight, blk.time_embed_proj.bias]
    return p


def branch_params(est) -> list:
    return decoder_params(est.decoder)


N_BRANCH_HEAD = 10
N_PER_BLOCK = 12


class _BranchView:
    def __init__(self, params: list):
        (self.w_in, self.b_in, self.ls_in, self.beta_in, self.tw0, self.tb0, self.tw2, self.tb2,
         self.w_out, self.b_out) = params[:N_BRANCH_HEAD]
        rest = params[N_BRANCH_HEAD:]
        self.nblk = len(rest) // N_PER_BLOCK
        self.blks = [_Blk(list(rest[i * N_PER_BLOCK: i * N_PER_BLOCK + 10]))
                     for i in range(self.nblk)]
        self.tew = [rest[i * N_PER_BLOCK + 10] for i in range(self.nblk)]
        self.teb = [rest[i * N_PER_BLOCK + 11] for i in range(self.nblk)]
        self.C = self.w_in.shape[0]
        self.Cin = self.w_in.shape[1]
        self.Dt = self.tw0.shape[1]
        self.Ht = self.tw0.shape[0]


# Spectra carry n_fft + 2 = 514 / 258 / 130 channels: no multiple of a K slab, which sends every
# GEMM that reduces over them (in_proj, inverse DFT, STFT backward, out_proj data gradient) to the
# element-wise loaders.  With the lean kernels available the spectrum rows are padded to a multiple
# of 64 columns that are ZERO BY CONSTRUCTION -- the producing GEMM runs against weights / DFT
# tables with zero rows appended, so it writes the zeros itself -- and the consumers reduce over the
# padded width against zero-padded weights: same sums (plus exact zeros), lean kernels.
SPEC_PAD = opt("spec_pad", True) and ops.L.get_option("lean") != 0


def _spec_ld(Cin: int) -> int:
    return (Cin + 63) // 64 * 64 if SPEC_PAD else ops.pad4(Cin)


def _pad_cols(w2d, Kp: int):
    """(n, k) -> cached (n, Kp) copy with zero columns appended."""
    def build(t):
        n, k = t.shape
        out = ops.zeros(n, Kp, device=t.device)
        ops.copy3(out, 0, Kp, t, 0, t.stride(0), 1, n, k)
        return out
    return w2d if w2d.shape[1] == Kp else ops.derived(w2d, ("padc", Kp), build)


def _pad_rows(w2d, Np: int):
    """(n, k) -> cached (Np, k) copy with zero rows appended."""
    def build(t):
        n, k = t.shape
        out = ops.zeros(Np, k, device=t.device)
        ops.copy3(out, 0, k, t, 0, t.stride(0), 1, n, k)
        return out
    return w2d if w2d.shape[0] == Np else ops.derived(w2d, ("padr", Np), build)


def _pad_vec(b, Np: int):
    def build(t):
        out = ops.zeros(Np, device=t.device)
        ops.copy3(out, 0, 0, t, 0, 0, 1, 1, t.shape[0])
        return out
    return b if b.shape[0] == Np else ops.derived(b, ("padv", Np), build)


def _time_path(bv: _BranchView, t):
    """t (n,) -> the time scale rows of all blocks of a decoder, (n, nblk * C) (modules.py:569-573,
    451, 485): sinusoidal embedding -> Linear -> SiLU -> Linear, then every block's time_emb
    Linear as one stacked GEMM."""
    dev = t.device
    n = t.shape[0]
    Dt, Ht = bv.Dt, bv.Ht
    emb = ops.empty(n, Dt, device=dev)
    ops.time_embedding(emb, t, Dt)
    th = ops.empty(n, Ht, device=dev)
    gemm(mat(emb), mat(bv.tw0), th, bias=bv.tb0)
    ts = ops.empty(n, Ht, device=dev)
    ops.silu(ts, th)
    te = ops.empty(n, Dt, device=dev)
    gemm(mat(ts), mat(bv.tw2), te, bias=bv.tb2)
    tew = _stack_rows(bv.tew, Dt, dev)
    teb = _stack_vecs(bv.teb, dev)
    te_all = ops.empty(n, bv.nblk * bv.C, device=dev)
    gemm(mat(te), mat(tew), te_all, bias=teb)
    return emb, th, ts, te, tew, te_all


def time_path_bwd(bv: _BranchView, tp, g_te_all, B: int, acc):
    """Backward of _time_path: g_te_all (B, nblk * C) -> the six parameter-gradient accumulators
    `acc` = (g_tew, g_teb, g_tb2, g_tw2, g_tb0, g_tw0), zero-initialised by the caller (t itself
    needs no gradient)."""
    emb, th, ts, te, tew, _ = tp
    g_tew, g_teb, g_tb2, g_tw2, g_tb0, g_tw0 = acc
    dev = g_te_all.device
    Dt, Ht, NC = bv.Dt, bv.Ht, bv.nblk * bv.C
    ops.colsum(g_teb, g_te_all, B, NC)
    ops.wgrad(g_te_all, NC, NC, mat(te, B, Dt), g_tew)
    g_te = ops.empty(B, Dt, device=dev)
    gemm(mat(g_te_all, B, NC), mat(tew), g_te, form=1)
    ops.colsum(g_tb2, g_te, B, Dt)
    ops.wgrad(g_te, Dt, Dt, mat(ts, B, Ht), g_tw2)
    g_ts = ops.empty(B, Ht, device=dev)
    gemm(mat(g_te, B, Dt), mat(bv.tw2), g_ts, form=1)
    g_th = ops.empty(B, Ht, device=dev)
    ops.silu_bwd(g_th, g_ts, th)
    ops.colsum(g_tb0, g_th, B, Ht)
    ops.wgrad(g_th, Ht, Ht, mat(emb, B, Dt), g_tw0)


def time_paths_ahead(flat, nparams, t_all, n_steps: int):
    """Inference: the time path depends on t only, and the Euler solver knows every t in advance --
    so it is computed ONCE for all steps (t_all = the steps' t vectors stacked, (n_steps * B,)), one
    launch lane per branch, instead of 9 small launches per branch and step in front of the blocks.
    Returns [step][branch] -> (B, nblk * C) row blocks of the stacked results (same arithmetic row by
    row as the per-step path)."""
    dev = t_all.device
    nb = len(nparams)
    B = t_all.shape[0] // n_steps
    views, off = [], 0
    for i in range(nb):
        views.append(_BranchView(list(flat[off: off + nparams[i]])))
        off += nparams[i]
    lanes = ops.Lanes(dev, nb, "timepath")
    full = []
    for i in range(nb):
        with lanes.lane(i):
            full.append(_time_path(views[i], t_all)[-1])
    return lanes, [[full[i][k * B: (k + 1) * B] for i in range(nb)] for k in range(n_steps)]


def _branch_pre(bv: _BranchView, meta, x, t, cproj, training, keep, te_pre=None):
    """STFT -> in_proj -> in_norm and the time path of one Fourier branch (modules.py:590-599,
    569-573): everything in front of the ConvNeXt blocks.  Returns the branch's state."""
    n_fft, hop, up, window = meta
    dev = x.device
    B, T = x.shape
    N = n_fft
    F = 1 + T // hop
    rows = B * F
    Cc, Cin = bv.C, bv.Cin
    Wd, Wi = dft_matrices(N, dev)
    ldp = _spec_ld(Cin)
    Kc = ldp if SPEC_PAD else Cin        # reduction width over a spectrum row
    # plain-bf16 inference: a GEMM result whose only reader is the next GEMM leaves its producer as
    # bf16 (the lean kernel's plain-store epilogue) instead of passing through a conversion launch
    bf16_chain = (not keep) and ops.GEMM_PRECISION == 2 and SPEC_PAD and ops.BF16_IMAGES \
        and ops.LEAN_SPLIT and Kc % 64 == 0 and N % 64 == 0
    if ops.fft_applies(N):
        # LDS-butterfly FFT; the kernel writes the zero padding of the rows itself, and in the
        # plain-bf16 inference chain the spectrum as the bf16 tensor in_proj's GEMM reads
        packed = torch.empty(rows, ldp, device=dev, dtype=torch.bfloat16) if bf16_chain \
            else ops.empty(rows, ldp, device=dev)
        ops.stft_fft(x, N, hop, F, packed, zero_pad=True)
        pk = mat(packed, rows, Kc, split=2 if bf16_chain else 0)
    elif bf16_chain:
        packed = torch.empty(rows, ldp, device=dev, dtype=torch.bfloat16)
        gemm(ops.stft_frames(x, N, hop, F), mat(_pad_rows(Wd, Kc)), packed, split_k=1, true_n=Cin)
        pk = mat(packed, rows, Kc, split=2)
    else:
        packed = ops.empty(rows, ldp, device=dev)
        gemm(ops.stft_frames(x, N, hop, F), mat(_pad_rows(Wd, Kc)), packed, split_k=1, true_n=Cin)   # bit-reproducible
        pk = mat(packed, rows, Kc)
    h0 = ops.empty(rows, Cc, device=dev)
    # (never split, as before the padding: the forward stays bit-reproducible from run to run)
    gemm(pk, mat(_pad_cols(bv.w_in.reshape(Cc, Cin), Kc)), h0, bias=bv.b_in, split_k=1, true_k=Cin)
    flags = [_limit_draw(training)]
    xcur = ops.empty(rows, Cc, device=dev)
    ops.biasnorm_fwd(h0, xcur, rows, Cc, bv.beta_in, bv.ls_in.reshape(1))
    NC = bv.nblk * Cc
    if te_pre is not None:      # computed ahead for all Euler steps (time_paths_ahead)
        emb = th = ts = te = tew = None
        te_all = te_pre
    else:
        emb, th, ts, te, tew, te_all = _time_path(bv, t)
    return dict(packed=packed, h0=h0, xcur=xcur, emb=emb, th=th, ts=ts, te=te, tew=tew, te_all=te_all,
                flags=flags, F=F, rows=rows, NC=NC, Fce=cproj.shape[0] // B, ldp=ldp, Kc=Kc,
                bf16_chain=bf16_chain, blocks=[])


def _branch_post(st, bv: _BranchView, meta, x_shape, wbranch_row, wscale, pred, accumulate, lens_f,
                 lanes=None, ola: bool = True):
    """out_proj -> iSTFT -> overlap-add into the shared prediction (modules.py:613-621,719).
    ola=False: stop in front of the overlap-add and return the frames (the caller adds all branches
    with one launch)."""
    n_fft, hop, up, window = meta
    B, T = x_shape
    N, F, rows, ldp, Kc = n_fft, st["F"], st["rows"], st["ldp"], st["Kc"]
    Cc, Cin = bv.C, bv.Cin
    xcur = st["xcur"]
    dev = xcur.device
    Wd, Wi = dft_matrices(N, dev)
    ifft = ops.fft_applies(N)       # inverse transform through the LDS FFT instead of the DFT GEMM
    ybf = st["bf16_chain"] and lens_f is None       # (the inverse FFT reads bf16 spectra as well)
    yspec = torch.empty(rows, ldp, device=dev, dtype=torch.bfloat16) if ybf else ops.empty(rows, ldp, device=dev)
    gemm(mat(xcur, rows, Cc), mat(_pad_rows(bv.w_out.reshape(Cin, Cc), Kc)), yspec,
         bias=_pad_vec(bv.b_out, Kc), split_k=1, true_n=Cin)
    if lens_f is not None:
        ops.mask_rows(yspec, B, F, Cin, lens_f)
    frames = ops.empty(rows, N, device=dev)
    if ifft:
        ops.istft_fft(yspec, N, F, frames)
    else:
        gemm(mat(yspec, rows, Kc, split=2 if ybf else 0), mat(_pad_cols(Wi, Kc)), frames, split_k=1, true_k=Cin)
    if not ola:
        return frames
    if lanes is not None:
        lanes.chain_enter()  # pred is accumulated branch after branch
    ops.istft_ola(frames, pred, B, F, N, hop, T, window, wbranch_row, wscale, accumulate)
    if lanes is not None:
        lanes.chain_leave()
    return None


def _branch_forward(bv: _BranchView, meta, x, t, cproj, wbranch_row, wscale, pred, accumulate,
                    lens_f, training, keep, lanes=None, te_pre=None):
    up = meta[2]
    B = x.shape[0]
    st = _branch_pre(bv, meta, x, t, cproj, training, keep, te_pre)
    Cc, F, NC = bv.C, st["F"], st["NC"]
    xcur = st["xcur"]
    for j, bp in enumerate(bv.blks):
        fn = _limit_draw(training)
        y, z, a = block_fwd(bp, xcur, B, F, lens_f, cproj, NC, st["Fce"], up, j * Cc, st["te_all"], NC,
                            j * Cc, keep=keep)
        st["flags"].append((fn, _limit_draw(training)))
        if keep:
            st["blocks"].append((xcur, z, a))
        xcur = y
    st["xcur"] = xcur
    _branch_post(st, bv, meta, x.shape, wbranch_row, wscale, pred, accumulate, lens_f, lanes)
    if keep:
        return dict(packed=st["packed"], h0=st["h0"], blocks=st["blocks"], x_last=xcur, emb=st["emb"],
                    th=st["th"], ts=st["ts"], te=st["te"], tew=st["tew"], te_all=st["te_all"],
                    flags=st["flags"], F=F)
    return None


def _multi_applies(views, x, keep) -> bool:
    """All branches' ConvNeXt blocks of a layer as ONE launch (ops.fused_block_multi): plain-bf16
    inference, 2-4 branches of equal depth whose shapes the fused block kernel has instances for."""
    if keep or not (ops.FUSED_MULTI and ops.FUSED_BLOCK):
        return False
    if not (2 <= len(views) <= 4) or len({bv.nblk for bv in views}) != 1:
        return False
    for bv in views:
        for bp in bv.blks:
            if ops.operand_formats_ok(bp.C, bp.H) != 2 or bp.K != 7 or not ops.fused_mlp_applies(bp.C, bp.H):
                return False
    return True


def _branches_forward_multi(views, metas, x, t, cprojs, wbranch, wscale, pred, lens_list, training,
                            te_pre=None):
    """The forward of all Fourier branches with layer-synchronous blocks: the parts in front of and
    behind the blocks run on one launch lane per branch as before, layer j of all branches is one
    launch on the caller's stream (tiles in order of decreasing cost: see fusedmlp.hip)."""
    nb = len(views)
    dev = x.device
    B = x.shape[0]
    sts = [None] * nb
    lanes = ops.Lanes(dev, nb, "branch")
    for i in range(nb):
        with lanes.lane(i):
            sts[i] = _branch_pre(views[i], metas[i], x, t, cprojs[i], training, False,
                                 None if te_pre is None else te_pre[i])
            for _ in range(2 * views[i].nblk):     # (the draws of the per-branch path, in its order)
                _limit_draw(training)
    lanes.join()
    for j in range(views[0].nblk):
        entries = []
        for i in range(nb):
            bv, st = views[i], sts[i]
            bp = bv.blks[j]
            Cc = bv.C
            entries.append(dict(
                x=st["xcur"], B=B, F=st["F"], Cc=Cc, K=bp.K, lens=lens_list[i], w_dw=bp.w_dw, b_dw=bp.b_dw,
                beta=bp.beta, log_scale=bp.log_scale.reshape(1), wp=ops.mlp_pack(bp.w1, bp.w2), b1=bp.b1,
                alpha=bp.alpha, b2=bp.b2, gamma=bp.gamma.reshape(Cc), out=ops.empty(st["rows"], Cc, device=dev),
                Hh=bp.H, cproj=cprojs[i], ldcp=st["NC"], Fc=st["Fce"], up=metas[i][2], cp_off=j * Cc,
                te=st["te_all"], ldte=st["NC"], te_off=j * Cc))
        outs = ops.fused_block_multi(entries)
        for i in range(nb):
            sts[i]["xcur"] = outs[i]
    lanes = ops.Lanes(dev, nb, "branch")
    frames = [None] * nb
    for i in range(nb):
        with lanes.lane(i):
            frames[i] = _branch_post(sts[i], views[i], metas[i], x.shape, None, wscale, pred, i > 0,
                                     lens_list[i], lanes, ola=False)
    lanes.join()
    # the overlap-adds of all branches (their weighted mean) as one pass over the prediction
    ops.istft_ola_multi([(frames[i], sts[i]["F"], metas[i][0], metas[i][1], metas[i][3],
                          None if wbranch is None else wbranch[i]) for i in range(nb)],
                        pred, B, x.shape[1], wscale)


def _branch_backward(bv: _BranchView, meta, sv, x_shape, cproj, g_pred, wbranch_row, wscale,
                     lens_f, g_x, accumulate_gx, g_cproj, need_gx, lanes=None):
    n_fft, hop, up, window = meta
    dev = g_pred.device
    B, T = x_shape
    N = n_fft
    F = sv["F"]
    rows = B * F
    Cc, Cin = bv.C, bv.Cin
    NC = bv.nblk * Cc
    Wd, Wi = dft_matrices(N, dev)
    ldp = _spec_ld(Cin)
    Kc = ldp if SPEC_PAD else Cin
    Fce = cproj.shape[0] // B
    gfr = ops.empty(rows, N, device=dev)
    ops.istft_ola_bwd(g_pred, gfr, B, F, N, hop, T, window, wbranch_row, wscale)
    if ops.fft_applies(N):
        gy = ops.empty(rows, ldp, device=dev)
        ops.istft_fft_adjoint(gfr, N, F, gy, zero_pad=True)             # (pad columns written as zeros)
    else:
        gy = ops.empty(rows, ldp, device=dev)
        gemm(mat(gfr, rows, N), mat(_pad_cols(Wi, Kc)), gy, form=1, true_n=Cin)      # pad columns come out zero
    if lens_f is not None:
        ops.mask_rows(gy, B, F, Cin, lens_f)
    # every gradient accumulator of the branch outside its blocks from ONE zeroed allocation (one fill)
    Dt, Ht = bv.Dt, bv.Ht
    (g_wout, g_bout, g_te_all, g_beta, g_ls, g_bin, g_win, g_tew, g_teb, g_tb2, g_tw2, g_tb0,
     g_tw0) = ops.zeros_many([(Cin, Cc), (Cin,), (B, NC), (Cc,), (1,), (Cc,), (Cc, Cin), (NC, Dt), (NC,), (Dt,),
                              (Dt, Ht), (Ht,), (Ht, Dt)], dev)
    ops.colsum(g_bout, gy, rows, Cin)
    x_last = sv["x_last"]
    ops.wgrad(gy, Cin, ldp, mat(x_last, rows, Cc), g_wout)
    g = ops.empty(rows, Cc, device=dev)
    gemm(mat(gy, rows, Kc), mat(_pad_rows(bv.w_out.reshape(Cin, Cc), Kc)), g, form=1, true_k=Cin)
    block_grads = [None] * bv.nblk
    flags = sv["flags"]
    for j in reversed(range(bv.nblk)):
        xj, z, a = sv["blocks"][j]
        fn, fs = flags[1 + j]
        g, gb = block_bwd(bv.blks[j], xj, z, a, g, B, F, lens_f, fn, fs, cproj, NC, Fce, up,
                          j * Cc, sv["te_all"], NC, j * Cc, g_cproj=g_cproj, g_te=g_te_all,
                          g_cproj_store=True)   # g_cp is zero-filled, block j owns columns j*C..
        block_grads[j] = gb
    # in_norm / in_proj / STFT
    gh0 = ops.empty(rows, Cc, device=dev)
    ops.biasnorm_bwd(sv["h0"], g, gh0, rows, Cc, bv.beta_in, bv.ls_in.reshape(1), g_beta, g_ls)
    if flags[0]:
        ops.limit_grad(g_ls, bv.ls_in.reshape(1), -1.5, 1.5)
    ops.colsum(g_bin, gh0, rows, Cc)
    ops.wgrad(gh0, Cc, gh0.stride(0), mat(sv["packed"], rows, Cin), g_win)
    if need_gx:
        gpacked = ops.empty(rows, ldp, device=dev)
        gemm(mat(gh0, rows, Cc), mat(_pad_cols(bv.w_in.reshape(Cc, Cin), Kc)), gpacked, form=1, true_n=Cin)
        gxf = ops.empty(rows, N, device=dev)
        if ops.fft_applies(N):
            ops.stft_fft_adjoint(gpacked, N, F, gxf)
        else:
            gemm(mat(gpacked, rows, Kc), mat(_pad_rows(Wd, Kc)), gxf, form=1, true_k=Cin)
        if lanes is not None:
            lanes.chain_enter()  # g_x is accumulated branch after branch
        ops.frames_fold(gxf, g_x, B, F, N, hop, T, accumulate_gx)
        if lanes is not None:
            lanes.chain_leave()
    time_path_bwd(bv, (sv["emb"], sv["th"], sv["ts"], sv["te"], sv["tew"], sv["te_all"]), g_te_all, B,
                  (g_tew, g_teb, g_tb2, g_tw2, g_tb0, g_tw0))
    out = [g_win.reshape(Cc, Cin, 1), g_bin, g_ls.reshape(()), g_beta, g_tw0, g_tb0, g_tw2, g_tb2,
           g_wout.reshape(Cin, Cc, 1), g_bout]
    for j in range(bv.nblk):
        out += block_grads[j]
        out.append(g_tew[j * Cc:(j + 1) * Cc])
        out.append(g_teb[j * Cc:(j + 1) * Cc])
    return out


class ModelEvalFn(GradAwareFunction):
    """pred (B, T) = mean_i w[i,b] * branch_i(x, cond, t)."""

    @staticmethod
    def forward(ctx, x, t, wbranch, metas, lens_cpu, training: bool, nparams, *args):
        # metas: one (n_fft, hop, up, window) per branch, optionally followed by ("scale", s): the
        # weight of every branch in the reduction (1/nb = mean, generator.py:165-168; 1 = sum)
        bscale = None
        te_pre = None
        if metas and metas[-1][0] == "te":      # inference: the time path computed ahead (per branch)
            te_pre = metas[-1][1]
            metas = metas[:-1]
        if metas and metas[-1][0] == "scale":
            bscale = float(metas[-1][1])
            metas = metas[:-1]
        nb = len(metas)
        if bscale is None:
            bscale = 1.0 / nb
        cprojs = list(args[:nb])
        flat = args[nb:]
        dev = x.device
        x = x.contiguous()
        B, T = x.shape
        pred = ops.empty(B, T, device=dev)
        keep = _keep(ctx)
        saved, views, lens_list = [], [], []
        off = 0
        for i in range(nb):  # everything the lanes share is created on the caller's stream
            views.append(_BranchView(list(flat[off: off + nparams[i]])))
            off += nparams[i]
            lens_list.append(frames_lens(lens_cpu, metas[i][1], dev))
            dft_matrices(metas[i][0], dev)
        if keep:
            te_pre = None
        if _multi_applies(views, x, keep):
            _branches_forward_multi(views, metas, x, t, cprojs, wbranch, bscale, pred, lens_list, training,
                                    te_pre)
            saved = [None] * nb
        else:
            lanes = ops.Lanes(dev, nb, "branch")  # one launch lane (HIP stream) per Fourier branch
            for i in range(nb):
                wrow = None if wbranch is None else wbranch[i]
                with lanes.lane(i):
                    sv = _branch_forward(views[i], metas[i], x, t, cprojs[i], wrow, bscale, pred,
                                         i > 0, lens_list[i], training, keep, lanes,
                                         None if te_pre is None else te_pre[i])
                saved.append(sv)
            lanes.join()
        if keep:
            # per-branch gradient hand-over to an armed exchange (dist._Sink)
            ctx.tickets = [sink_register(list(flat[sum(nparams[:i]): sum(nparams[:i + 1])]))
                           for i in range(nb)]
            ctx.flat = flat
            ctx.saved = saved
            ctx.views = views
            ctx.lens = lens_list
            ctx.x_shape = (B, T)
            ctx.cprojs = cprojs
            ctx.wbranch = wbranch
            ctx.metas = metas
            ctx.bscale = bscale
            ctx.nparams = nparams
        return pred

    @staticmethod
    def backward(ctx, g_pred):
        nb = len(ctx.metas)
        dev = g_pred.device
        g_pred = g_pred.contiguous()
        B, T = ctx.x_shape
        need_gx = ctx.needs_input_grad[0]
        g_x = ops.empty(B, T, device=dev) if need_gx else None
        g_cprojs, g_flat = [], []
        lanes = ops.Lanes(dev, nb, "branch")  # branch i runs on the lane that holds its activations
        for i in range(nb):
            cproj = ctx.cprojs[i]
            need_gc = ctx.needs_input_grad[7 + i]
            wrow = None if ctx.wbranch is None else ctx.wbranch[i]
            with lanes.lane(i):
                g_cp = ops.zeros(cproj.shape[0], cproj.shape[1], device=dev) if need_gc else None
                gl = _branch_backward(ctx.views[i], ctx.metas[i], ctx.saved[i], (B, T), cproj,
                                      g_pred, wrow, ctx.bscale, ctx.lens[i], g_x, i > 0, g_cp,
                                      need_gx, lanes)
                # branch i is done on its lane: its bucket may leave while the others compute
                off = sum(ctx.nparams[:i])
                g_flat += deliver_grads(ctx.tickets[i], list(ctx.flat[off: off + ctx.nparams[i]]), gl)
            g_cprojs.append(g_cp)
        lanes.join()
        ctx.saved = None
        ctx.flat = None
        return tuple([g_x, None, None, None, None, None, None] + g_cprojs + g_flat)


# =====================================================================================
# Euler / interpolation steps as autograd nodes (generator.py:217,263-264)
# =====================================================================================
class AxpbyFn(torch.autograd.Function):
    """y = a*x0 + b*x1 (scalars)."""

    @staticmethod
    def forward(ctx, x0, x1, a: float, b: float):
        ctx.ab = (a, b)
        y = torch.empty_like(x0)
        return ops.axpby_rows(y, x0.contiguous(), x1.contiguous(), sa=a, sb=b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        g = g.contiguous()
        g0 = g1 = None
        if ctx.needs_input_grad[0]:
            g0 = ops.axpby_rows(torch.empty_like(g), g, None, sa=a)
        if ctx.needs_input_grad[1]:
            g1 = ops.axpby_rows(torch.empty_like(g), g, None, sa=b)
        return g0, g1, None, None


# =====================================================================================
# Spectrogram helpers + stage-1 loss (generator.py:172-200, modules.py:146-214, A.7)
# =====================================================================================
def stft_packed(x, n_fft: int, hop: int):
    """(B, T) -> packed rows (B*F, pad4(n_fft+2)), F = 1 + T//hop (bit-exact frame indexing:
    frame m covers reflect-padded samples [m*hop, m*hop + n_fft))."""
    dev = x.device
    B, T = x.shape
    F = 1 + T // hop
    Wd, _ = dft_matrices(n_fft, dev)
    packed = ops.empty(B * F, ops.pad4(n_fft + 2), device=dev)
    if ops.fft_applies(n_fft):
        ops.stft_fft(x, n_fft, hop, F, packed)   # LDS-butterfly FFT (fft.hip): n_fft >= 1024
    else:
        gemm(ops.stft_frames(x, n_fft, hop, F), mat(Wd), packed, split_k=1)   # never split: bit-reproducible
    return packed, F


def filterbank_spec(x, n_fft: int, hop: int, fb, power: int):
    """|STFT|^power @ fb: returns (S (B*F, n_filt), packed, spec, F)."""
    dev = x.device
    packed, F = stft_packed(x, n_fft, hop)
    rows = packed.shape[0]
    nb = n_fft // 2 + 1
    spec = ops.empty(rows, ops.pad4(nb), device=dev)
    ops.spec_power(spec, packed, rows, nb, power)
    nf = fb.shape[1]
    S = ops.empty(rows, nf, device=dev)
    gemm(mat(spec, rows, nb), mat(fb), S, form=1)
    return S, packed, spec, F


def filterbank_spec_bwd(gS, packed, n_fft: int, hop: int, fb, power: int, B: int, T: int, F: int,
                        g_x, accumulate: bool, lanes=None):
    dev = gS.device
    rows = packed.shape[0]
    nb = n_fft // 2 + 1
    gspec = ops.empty(rows, ops.pad4(nb), device=dev)
    gemm(mat(gS, rows, fb.shape[1]), mat(fb), gspec, form=0)
    gpacked = ops.empty(rows, packed.shape[1], device=dev)
    ops.spec_power_bwd(gpacked, gspec, packed, rows, nb, power)
    Wd, _ = dft_matrices(n_fft, dev)
    gfr = ops.empty(rows, n_fft, device=dev)
    if ops.fft_applies(n_fft):
        ops.stft_fft_adjoint(gpacked, n_fft, F, gfr)
    else:
        gemm(mat(gpacked, rows, n_fft + 2), mat(Wd), gfr, form=1)
    if lanes is not None:
        lanes.chain_enter()  # g_x is accumulated scale after scale
    ops.frames_fold(gfr, g_x, B, F, n_fft, hop, T, accumulate)
    if lanes is not None:
        lanes.chain_leave()


@torch.no_grad()
def log_mel_forward(mel_mod, waveform):
    """LogMelSpectrogram.forward (modules.py:140-143): (B, T) -> (B, n_mels, F)."""
    x = waveform.contiguous()
    squeeze = x.dim() == 1
    if squeeze:
        x = x[None]
    B, T = x.shape
    S, _, _, F = filterbank_spec(x, mel_mod.n_fft, mel_mod.hop_length, mel_mod.mel_scale.fb, 1)
    # log(clip(., 1e-7)) fused with the transpose back to (B, n_mels, F)
    n = mel_mod.n_mels
    ops.log_clip_(S, 1e-7)
    out = ops.empty(B, n, F, device=x.device)
    ops.rows_to_bct(out, S, B, n, F)
    return out[0] if squeeze else out


class FmLossFn(torch.autograd.Function):
    """Stage-1 spectrally scaled endpoint loss (generator.py:172-200)."""

    @staticmethod
    def forward(ctx, pred, x1, lens_cpu, n_fft, hop, fb, eps, power, lo, hi, gt=None):
        """err = pred - x1 (`x1` = the regression target: the audio, or audio - noise for the
        velocity objective); the spectral weights come from `gt` (the audio; default: x1)."""
        dev = pred.device
        B, T = pred.shape
        err = torch.empty_like(pred)
        ops.axpby_rows(err, pred.contiguous(), x1.contiguous(), sa=1.0, sb=-1.0)
        S_gt, _, _, F = filterbank_spec((x1 if gt is None else gt).contiguous(), n_fft, hop, fb, 2)
        S_err, packed_err, _, _ = filterbank_spec(err, n_fft, hop, fb, 2)
        nf = fb.shape[1]
        lens_f = frames_lens(lens_cpu, hop, dev)
        if lens_cpu is not None:
            assert F == max(1 + int(l) // hop for l in lens_cpu)
            nmask = sum(1 + int(l) // hop for l in lens_cpu)
        else:
            nmask = B * F
        loss = ops.zeros(1, device=dev)
        w = ops.empty(B * F, nf, device=dev)
        ops.fm_spec_loss(loss, w, S_err, S_gt, B, F, nf, lens_f, eps, power, lo, hi,
                         1.0 / (nmask * nf))
        ctx.saved = (w, packed_err)
        ctx.meta = (n_fft, hop, fb, B, T, F)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        w, packed_err = ctx.saved
        n_fft, hop, fb, B, T, F = ctx.meta
        dev = w.device
        g_err = ops.empty(B, T, device=dev)
        filterbank_spec_bwd(w, packed_err, n_fft, hop, fb, 2, B, T, F, g_err, False)
        # chain the incoming scalar on the device (no host sync)
        flat = g_err.view(1, -1)
        ops.axpby_rows(flat, flat, None, ca=g.reshape(1).contiguous())
        ctx.saved = None
        return g_err, None, None, None, None, None, None, None, None, None, None


class MseLossFn(torch.autograd.Function):
    """Unweighted stage-1 loss (generator.py:181-184, spec_scaling_loss=False): the masked mean of
    (pred - ref)^2 over the valid samples of every item."""

    @staticmethod
    def forward(ctx, pred, ref, lens_cpu):
        dev = pred.device
        B, T = pred.shape
        if lens_cpu is None:
            lens, n = None, B * T
        else:
            assert T == max(int(l) for l in lens_cpu)
            lens = torch.tensor([int(l) for l in lens_cpu], dtype=torch.int32, device=dev)
            n = sum(int(l) for l in lens_cpu)
        loss = ops.zeros(1, device=dev)
        g_err = ops.empty(B, T, device=dev)
        ops.masked_mse(loss, g_err, pred.contiguous(), ref.contiguous(), B, T, lens, 1.0 / n)
        ctx.saved = g_err
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        g_err = ctx.saved
        flat = g_err.view(1, -1)
        ops.axpby_rows(flat, flat, None, ca=g.reshape(1).contiguous())   # no host sync
        ctx.saved = None
        return g_err, None, None
