"""Leaf entry points of `flow2gan.models.modules` / `gan` in the reference's own calling convention:
(batch, channels, time) tensors in and out, complex spectra for STFT / ISTFT, lists of score / feature
maps for the GAN's loss methods.

The training / inference path never comes through here -- `fused.py` / `fused_disc.py` run whole
branches and loss stacks as coarse nodes over channels-last rows.  These functions give every leaf
module of `flow2gan_amd.models` the `forward` its reference counterpart has (modules.py:52-84, 87-116,
146-232, 273-283, 419-721; gan.py:57-87) ON THE SAME KERNELS, with autograd: a layout turn at the
boundary, then the fused block / decoder / condition-encoder functions.  Nothing here computes in ATen.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import fused, ops
from .fused import GradAwareFunction, _Blk, _BranchView, _keep, _limit_draw
from .ops import gemm, mat


# ------------------------------------------------------------------------------ layout turns
class BctToRowsFn(torch.autograd.Function):
    """(B, C, F) -> rows (B*F, C); backward is the inverse turn."""

    @staticmethod
    def forward(ctx, x):
        B, Cc, F = x.shape
        ctx.dims = (B, Cc, F)
        rows = ops.empty(B * F, Cc, device=x.device)
        return ops.bct_to_rows(rows, x.contiguous().float(), B, Cc, F)

    @staticmethod
    def backward(ctx, g):
        B, Cc, F = ctx.dims
        g = g.contiguous()
        return ops.rows_to_bct(ops.empty(B, Cc, F, device=g.device), g, B, Cc, F)


class RowsToBctFn(torch.autograd.Function):
    """rows (B*F, C) -> (B, C, F)."""

    @staticmethod
    def forward(ctx, rows, B: int, Cc: int, F: int):
        ctx.dims = (B, Cc, F)
        return ops.rows_to_bct(ops.empty(B, Cc, F, device=rows.device), rows, B, Cc, F)

    @staticmethod
    def backward(ctx, g):
        B, Cc, F = ctx.dims
        out = ops.empty(B * F, Cc, device=g.device)
        return ops.bct_to_rows(out, g.contiguous(), B, Cc, F), None, None, None


def mask_to_lens(mask, B: int, F: int) -> Optional[List[int]]:
    """The reference builds every mask as `make_pad_mask(lens).logical_not()` (utils.py:41-66,
    modules.py:706-709): a prefix of ones per item.  The kernels take the prefix LENGTHS; any other
    mask is refused loudly instead of being approximated."""
    if mask is None:
        return None
    m = mask.reshape(B, F).to(torch.bool).cpu()
    lens = m.sum(dim=1)
    if not torch.equal(torch.arange(F)[None, :] < lens[:, None], m):
        raise ValueError("flow2gan_amd: mask must be a padding mask (ones then zeros per item, "
                         "utils.make_pad_mask); arbitrary masks are not supported by the HIP path")
    return [int(v) for v in lens]


def _lens_dev(lens_cpu, device):
    if lens_cpu is None:
        return None
    return torch.tensor([int(v) for v in lens_cpu], dtype=torch.int32, device=device)


# ------------------------------------------------------------------------------ STFT / ISTFT
class StftFn(torch.autograd.Function):
    """audio (B, T) -> (B, K, F, 2) real view of torch.stft's complex output (modules.py:69-78)."""

    @staticmethod
    def forward(ctx, audio, n_fft: int, hop: int):
        x = audio.contiguous().float()
        B, T = x.shape
        packed, F = fused.stft_packed(x, n_fft, hop)      # planar rows [Re(0..N/2) | Im(0..N/2)]
        K = n_fft // 2 + 1
        ld = packed.stride(0)
        out = ops.empty(B, K, F, 2, device=x.device)
        ops.permute4(out, packed, (B, K, F, 2), (F * ld, 1, ld, K))
        ctx.meta = (n_fft, hop, B, T, F, K, ld)
        return out

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, B, T, F, K, ld = ctx.meta
        dev = g.device
        sb, sk, sf, sc = g.stride()
        tight = ops.empty(B * F, 2 * K, device=dev)
        ops.permute4(tight, g, (B, F, 2, K), (sb, sf, sc, sk))
        gfr = ops.empty(B * F, n_fft, device=dev)
        if ops.fft_applies(n_fft):
            ops.stft_fft_adjoint(tight, n_fft, F, gfr)
        else:
            gpacked = ops.zeros(B * F, ld, device=dev)     # 16-byte aligned rows for the GEMM loaders
            ops.copy3(gpacked, 0, ld, tight, 0, 2 * K, 1, B * F, 2 * K)
            Wd, _ = fused.dft_matrices(n_fft, dev)
            gemm(mat(gpacked, B * F, n_fft + 2), mat(Wd), gfr, form=1)
        gx = ops.empty(B, T, device=dev)
        ops.frames_fold(gfr, gx, B, F, n_fft, hop, T, False)
        return gx, None, None


def stft(audio, n_fft: int, hop: int):
    """torch.stft(center=True, reflect, periodic hann, onesided, return_complex=True)."""
    return torch.view_as_complex(StftFn.apply(audio, n_fft, hop))


class IstftFn(torch.autograd.Function):
    """(B, K, F, 2) real view of a complex half spectrum -> audio (B, hop*(F-1)) (modules.py:106-115;
    SURVEY A.2: Im of DC / Nyquist ignored, envelope normalisation, n_fft/2 trimmed on both sides)."""

    @staticmethod
    def forward(ctx, spec_ri, n_fft: int, hop: int, window):
        dev = spec_ri.device
        B, K, F, _ = spec_ri.shape
        assert K == n_fft // 2 + 1, (K, n_fft)
        rows, T = B * F, hop * (F - 1)
        sb, sk, sf, sc = spec_ri.stride()
        ld = ops.pad4(2 * K)
        tight = ops.empty(rows, 2 * K, device=dev)
        ops.permute4(tight, spec_ri, (B, F, 2, K), (sb, sf, sc, sk))
        frames = ops.empty(rows, n_fft, device=dev)
        if ops.fft_applies(n_fft):
            ops.istft_fft(tight, n_fft, F, frames)
        else:
            ys = ops.zeros(rows, ld, device=dev)
            ops.copy3(ys, 0, ld, tight, 0, 2 * K, 1, rows, 2 * K)
            _, Wi = fused.dft_matrices(n_fft, dev)
            gemm(mat(ys, rows, n_fft + 2), mat(Wi), frames, split_k=1)
        out = ops.empty(B, T, device=dev)
        ops.istft_ola(frames, out, B, F, n_fft, hop, T, window, None, 1.0, False)
        ctx.meta = (n_fft, hop, B, K, F, T)
        ctx.window = window
        return out

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, B, K, F, T = ctx.meta
        dev = g.device
        rows = B * F
        gfr = ops.empty(rows, n_fft, device=dev)
        ops.istft_ola_bwd(g.contiguous(), gfr, B, F, n_fft, hop, T, ctx.window, None, 1.0)
        ld = ops.pad4(2 * K)
        gy = ops.empty(rows, ld, device=dev)
        if ops.fft_applies(n_fft):
            ops.istft_fft_adjoint(gfr, n_fft, F, gy, zero_pad=True)
        else:
            _, Wi = fused.dft_matrices(n_fft, dev)
            gemm(mat(gfr, rows, n_fft), mat(Wi), gy, form=1)
        out = ops.empty(B, K, F, 2, device=dev)
        ops.permute4(out, gy, (B, K, F, 2), (F * ld, 1, ld, K))
        return out, None, None, None


def istft(spec, n_fft: int, hop: int, window):
    return IstftFn.apply(torch.view_as_real(spec), n_fft, hop, window)


# ------------------------------------------------------------------------------ filterbank spectrograms
class FilterbankSpecFn(torch.autograd.Function):
    """waveform (B, T) -> (B, n_filter, F): |STFT|^power through a triangular filterbank
    (LinearFilterSpectrogram modules.py:146-214; torchaudio MelSpectrogram with power = 1)."""

    @staticmethod
    def forward(ctx, x, n_fft: int, hop: int, fb, power: int):
        x = x.contiguous().float()
        B, T = x.shape
        S, packed, _, F = fused.filterbank_spec(x, n_fft, hop, fb, power)
        nf = fb.shape[1]
        out = ops.rows_to_bct(ops.empty(B, nf, F, device=x.device), S, B, nf, F)
        ctx.saved = packed
        ctx.meta = (n_fft, hop, power, B, T, F, nf)
        ctx.fb = fb
        return out

    @staticmethod
    def backward(ctx, g):
        n_fft, hop, power, B, T, F, nf = ctx.meta
        dev = g.device
        gS = ops.bct_to_rows(ops.empty(B * F, nf, device=dev), g.contiguous(), B, nf, F)
        gx = ops.empty(B, T, device=dev)
        fused.filterbank_spec_bwd(gS, ctx.saved, n_fft, hop, ctx.fb, power, B, T, F, gx, False)
        ctx.saved = None
        return gx, None, None, None, None


def filterbank_spectrogram(waveform, n_fft: int, hop: int, fb, power: int):
    lead = waveform.shape[:-1]
    x = waveform.reshape(-1, waveform.shape[-1])
    y = FilterbankSpecFn.apply(x, n_fft, hop, fb, power)
    return y.reshape(*lead, y.shape[1], y.shape[2])


# ------------------------------------------------------------------------------ ChannelScale / BiasNorm
class ChannelScaleFn(torch.autograd.Function):
    """x (B, C, T) * scale (C, 1), with LimitParamValue on the scale's gradient when drawn
    (modules.py:236-283)."""

    @staticmethod
    def forward(ctx, x, scale, limit: bool):
        x = x.contiguous().float()
        B, Cc, T = x.shape
        dev = x.device
        tiled = ops.empty(B, Cc, device=dev)                       # scale[c] for every row (b, c)
        ops.copy3(tiled, Cc, 0, scale, 0, 0, B, 1, Cc)
        y = ops.axpby_rows(ops.empty(B * Cc, T, device=dev), x.view(B * Cc, T), None, ca=tiled.view(-1))
        ctx.saved = (x, scale, tiled)
        ctx.limit = limit
        return y.view(B, Cc, T)

    @staticmethod
    def backward(ctx, g):
        x, scale, tiled = ctx.saved
        B, Cc, T = x.shape
        dev = g.device
        g = g.contiguous()
        gx = ops.axpby_rows(ops.empty(B * Cc, T, device=dev), g.view(B * Cc, T), None, ca=tiled.view(-1))
        # d scale[c] = sum_{b,t} g x: column sums of the product in the rows layout
        gr = ops.bct_to_rows(ops.empty(B * T, Cc, device=dev), g, B, Cc, T)
        xr = ops.bct_to_rows(ops.empty(B * T, Cc, device=dev), x, B, Cc, T)
        gs = ops.zeros(Cc, 1, device=dev)
        ops.colsum(gs, gr, B * T, Cc, b=xr)
        if ctx.limit:
            ops.limit_grad(gs, scale, 0.5, 1.0)
        ctx.saved = None
        return gx.view(B, Cc, T), gs, None


class BiasNormFn(torch.autograd.Function):
    """BiasNorm over the channel axis of (B, C, T) (modules.py:286-416)."""

    @staticmethod
    def forward(ctx, x, bias, log_scale, limit: bool):
        B, Cc, T = x.shape
        dev = x.device
        xr = ops.bct_to_rows(ops.empty(B * T, Cc, device=dev), x.contiguous().float(), B, Cc, T)
        yr = ops.biasnorm_fwd(xr, ops.empty(B * T, Cc, device=dev), B * T, Cc, bias, log_scale.reshape(1))
        ctx.saved = (xr, bias, log_scale)
        ctx.limit = limit
        ctx.dims = (B, Cc, T)
        return ops.rows_to_bct(ops.empty(B, Cc, T, device=dev), yr, B, Cc, T)

    @staticmethod
    def backward(ctx, g):
        xr, bias, log_scale = ctx.saved
        B, Cc, T = ctx.dims
        dev = g.device
        gr = ops.bct_to_rows(ops.empty(B * T, Cc, device=dev), g.contiguous(), B, Cc, T)
        g_beta, g_ls = ops.zeros_many([(Cc,), (1,)], dev)
        gxr = ops.biasnorm_bwd(xr, gr, ops.empty(B * T, Cc, device=dev), B * T, Cc, bias,
                               log_scale.reshape(1), g_beta, g_ls)
        if ctx.limit:
            ops.limit_grad(g_ls, log_scale.reshape(1), -1.5, 1.5)
        ctx.saved = None
        return ops.rows_to_bct(ops.empty(B, Cc, T, device=dev), gxr, B, Cc, T), g_beta, g_ls.reshape(()), None


# ------------------------------------------------------------------------------ ConvNeXt block
class BlockFn(GradAwareFunction):
    """ConvNeXtBlock.forward (modules.py:455-495) on (B, C, T): depthwise conv + BiasNorm + condition /
    time terms (`f2g_dwnorm_fwd`), pwconv1 -> PReLU -> pwconv2 + gamma * residual (the GEMM kernels)."""

    @staticmethod
    def forward(ctx, x, cond, te, lens_cpu, training: bool, *params):
        dev = x.device
        B, Cc, F = x.shape
        bp = _Blk(list(params[:10]))
        rest = list(params[10:])
        wc = bc = wt = bt = None
        if cond is not None:
            wc, bc = rest[0], rest[1]
            rest = rest[2:]
        if te is not None:
            wt, bt = rest[0], rest[1]
        rows = B * F
        keep = _keep(ctx)
        xr = ops.bct_to_rows(ops.empty(rows, Cc, device=dev), x.contiguous().float(), B, Cc, F)
        cr = cproj = tep = None
        if cond is not None:
            Dc = cond.shape[1]
            cr = ops.bct_to_rows(ops.empty(rows, Dc, device=dev), cond.contiguous().float(), B, Dc, F)
            cproj = ops.empty(rows, Cc, device=dev)
            gemm(mat(cr, rows, Dc), mat(wc.reshape(Cc, Dc)), cproj, bias=bc)
        if te is not None:
            te = te.contiguous().float()
            tep = ops.empty(B, Cc, device=dev)
            gemm(mat(te), mat(wt), tep, bias=bt)
        lens = _lens_dev(lens_cpu, dev)
        fn = _limit_draw(training)                      # BiasNorm's draw comes first (modules.py:476)
        y, z, a = fused.block_fwd(bp, xr, B, F, lens, cproj, Cc, F, 1, 0, tep, Cc, 0, keep=keep)
        fs = _limit_draw(training)                      # then ChannelScale's (modules.py:491)
        if keep:
            ctx.saved = (xr, z, a, cr, cproj, te, tep, lens)
            ctx.params = params
            ctx.flags = (fn, fs)
            ctx.dims = (B, Cc, F)
        return ops.rows_to_bct(ops.empty(B, Cc, F, device=dev), y, B, Cc, F)

    @staticmethod
    def backward(ctx, g):
        xr, z, a, cr, cproj, te, tep, lens = ctx.saved
        params = ctx.params
        B, Cc, F = ctx.dims
        dev = g.device
        rows = B * F
        bp = _Blk(list(params[:10]))
        rest = list(params[10:])
        gr = ops.bct_to_rows(ops.empty(rows, Cc, device=dev), g.contiguous(), B, Cc, F)
        g_cp = ops.zeros(rows, Cc, device=dev) if cproj is not None else None
        g_tep = ops.zeros(B, Cc, device=dev) if tep is not None else None
        gx, gb = fused.block_bwd(bp, xr, z, a, gr, B, F, lens, ctx.flags[0], ctx.flags[1], cproj, Cc, F, 1, 0,
                                 tep, Cc, 0, g_cproj=g_cp, g_te=g_tep, g_cproj_store=True)
        out = [ops.rows_to_bct(ops.empty(B, Cc, F, device=dev), gx, B, Cc, F), None, None, None, None] + gb
        if cproj is not None:
            wc = rest[0]
            rest = rest[2:]
            Dc = cr.shape[1]
            g_wc, g_bc = ops.zeros_many([(Cc, Dc), (Cc,)], dev)
            ops.colsum(g_bc, g_cp, rows, Cc)
            ops.wgrad(g_cp, Cc, g_cp.stride(0), mat(cr, rows, Dc), g_wc)
            if ctx.needs_input_grad[1]:
                g_cr = ops.empty(rows, Dc, device=dev)
                gemm(mat(g_cp, rows, Cc), mat(wc.reshape(Cc, Dc)), g_cr, form=1)
                out[1] = ops.rows_to_bct(ops.empty(B, Dc, F, device=dev), g_cr, B, Dc, F)
            out += [g_wc.reshape(Cc, Dc, 1), g_bc]
        if tep is not None:
            wt = rest[0]
            Dt = te.shape[1]
            g_wt, g_bt = ops.zeros_many([(Cc, Dt), (Cc,)], dev)
            ops.colsum(g_bt, g_tep, B, Cc)
            ops.wgrad(g_tep, Cc, Cc, mat(te, B, Dt), g_wt)
            if ctx.needs_input_grad[2]:
                out[2] = ops.empty(B, Dt, device=dev)
                gemm(mat(g_tep, B, Cc), mat(wt), out[2], form=1)
            out += [g_wt, g_bt]
        ctx.saved = None
        return tuple(out)


def block_forward(blk, x, cond=None, time_embed=None, mask=None):
    B, _, F = x.shape
    params = fused.block_params(blk)
    if cond is not None:
        params += [blk.cond_proj.weight, blk.cond_proj.bias]
    if time_embed is not None:
        params += [blk.time_embed_proj.weight, blk.time_embed_proj.bias]
    return BlockFn.apply(x, cond, time_embed, mask_to_lens(mask, B, F), blk.training, *params)


# ------------------------------------------------------------------------------ ConvNeXt decoder
class DecoderRowsFn(GradAwareFunction):
    """in_proj -> in_norm -> [time path] -> blocks -> out_proj of a ConvNeXtDecoder (modules.py:590-627)
    over rows; the per-block condition projections arrive as `cproj` (fused.CondPathFn)."""

    @staticmethod
    def forward(ctx, xr, t, cproj, B: int, F: int, lens_cpu, training: bool, *params):
        dev = xr.device
        bv = _BranchView(list(params))
        Cc, Cin, Cout = bv.C, bv.Cin, bv.w_out.shape[0]
        rows, NC = B * F, bv.nblk * bv.C
        keep = _keep(ctx)
        lens = _lens_dev(lens_cpu, dev)
        h0 = ops.empty(rows, Cc, device=dev)
        gemm(mat(xr, rows, Cin), mat(bv.w_in.reshape(Cc, Cin)), h0, bias=bv.b_in, split_k=1)
        flags = [_limit_draw(training)]
        xcur = ops.biasnorm_fwd(h0, ops.empty(rows, Cc, device=dev), rows, Cc, bv.beta_in, bv.ls_in.reshape(1))
        tp = fused._time_path(bv, t) if t is not None else None
        te_all = None if tp is None else tp[-1]
        blocks = []
        for j, bp in enumerate(bv.blks):
            fn = _limit_draw(training)
            y, z, a = fused.block_fwd(bp, xcur, B, F, lens, cproj, NC, F, 1, j * Cc, te_all, NC, j * Cc, keep=keep)
            flags.append((fn, _limit_draw(training)))
            if keep:
                blocks.append((xcur, z, a))
            xcur = y
        out = ops.empty(rows, Cout, device=dev)
        gemm(mat(xcur, rows, Cc), mat(bv.w_out.reshape(Cout, Cc)), out, bias=bv.b_out, split_k=1)
        if keep:
            ctx.saved = (xr, h0, blocks, xcur, tp, cproj, lens)
            ctx.params = params
            ctx.flags = flags
            ctx.dims = (B, F)
        return out

    @staticmethod
    def backward(ctx, gy):
        xr, h0, blocks, x_last, tp, cproj, lens = ctx.saved
        B, F = ctx.dims
        bv = _BranchView(list(ctx.params))
        flags = ctx.flags
        dev = gy.device
        Cc, Cin, Cout = bv.C, bv.Cin, bv.w_out.shape[0]
        rows, NC = B * F, bv.nblk * bv.C
        Dt, Ht = bv.Dt, bv.Ht
        gy = gy.contiguous()
        (g_wout, g_bout, g_te_all, g_beta, g_ls, g_bin, g_win, g_tew, g_teb, g_tb2, g_tw2, g_tb0,
         g_tw0) = ops.zeros_many([(Cout, Cc), (Cout,), (B, NC), (Cc,), (1,), (Cc,), (Cc, Cin), (NC, Dt), (NC,),
                                  (Dt,), (Dt, Ht), (Ht,), (Ht, Dt)], dev)
        ops.colsum(g_bout, gy, rows, Cout)
        ops.wgrad(gy, Cout, gy.stride(0), mat(x_last, rows, Cc), g_wout)
        g = ops.empty(rows, Cc, device=dev)
        gemm(mat(gy, rows, Cout), mat(bv.w_out.reshape(Cout, Cc)), g, form=1)
        need_gc = ctx.needs_input_grad[2]
        g_cp = ops.zeros(rows, NC, device=dev)
        block_grads = [None] * bv.nblk
        te_all = None if tp is None else tp[-1]
        for j in reversed(range(bv.nblk)):
            xj, z, a = blocks[j]
            fn, fs = flags[1 + j]
            g, gb = fused.block_bwd(bv.blks[j], xj, z, a, g, B, F, lens, fn, fs, cproj, NC, F, 1, j * Cc,
                                    te_all, NC, j * Cc, g_cproj=g_cp, g_te=g_te_all if tp is not None else None,
                                    g_cproj_store=True)
            block_grads[j] = gb
        gh0 = ops.empty(rows, Cc, device=dev)
        ops.biasnorm_bwd(h0, g, gh0, rows, Cc, bv.beta_in, bv.ls_in.reshape(1), g_beta, g_ls)
        if flags[0]:
            ops.limit_grad(g_ls, bv.ls_in.reshape(1), -1.5, 1.5)
        ops.colsum(g_bin, gh0, rows, Cc)
        ops.wgrad(gh0, Cc, gh0.stride(0), mat(xr, rows, Cin), g_win)
        g_xr = None
        if ctx.needs_input_grad[0]:
            g_xr = ops.empty(rows, Cin, device=dev)
            gemm(mat(gh0, rows, Cc), mat(bv.w_in.reshape(Cc, Cin)), g_xr, form=1)
        if tp is not None:
            fused.time_path_bwd(bv, tp, g_te_all, B, (g_tew, g_teb, g_tb2, g_tw2, g_tb0, g_tw0))
        out = [g_xr, None, g_cp if need_gc else None, None, None, None, None,
               g_win.reshape(Cc, Cin, 1), g_bin, g_ls.reshape(()), g_beta, g_tw0, g_tb0, g_tw2, g_tb2,
               g_wout.reshape(Cout, Cc, 1), g_bout]
        for j in range(bv.nblk):
            out += block_grads[j]
            out.append(g_tew[j * Cc:(j + 1) * Cc])
            out.append(g_teb[j * Cc:(j + 1) * Cc])
        ctx.saved = None
        return tuple(out)


def decoder_forward(dec, x, cond, t=None, mask=None):
    """ConvNeXtDecoder.forward (modules.py:590-627): x (B, in_channels, F), cond (B, cond_channels, F)
    already at the frame rate, t (B,), mask (B, 1, F)."""
    B, _, F = x.shape
    assert cond.shape[0] == B and cond.shape[2] == F, "cond must be at the frame rate of x"
    xr = BctToRowsFn.apply(x)
    cr = BctToRowsFn.apply(cond)
    cproj = fused.CondPathFn.apply(cr, B, F, F, *fused.cond_path_params(dec))
    tt = None if t is None else t.flatten().contiguous().float()
    yr = DecoderRowsFn.apply(xr, tt, cproj, B, F, mask_to_lens(mask, B, F), dec.training,
                             *fused.decoder_params(dec))
    return RowsToBctFn.apply(yr, B, dec.out_channels, F)


# ------------------------------------------------------------------------------ AudioConvNeXt
def audio_convnext_forward(est, audio, cond, t=None, audio_lens=None):
    """AudioConvNeXt.forward (modules.py:682-721) = one Fourier branch of the generator: the same
    coarse node as a model evaluation (fused.ModelEvalFn) with a single branch of weight 1."""
    assert t is not None, "the HIP branch is time-conditioned (generator.py always passes t)"
    B, T = audio.shape
    cr = BctToRowsFn.apply(cond)
    F = 1 + T // est.hop_length
    up = est.cond_upsample_factor
    Fce = (F + up - 1) // up
    cproj = fused.CondPathFn.apply(cr, B, cond.shape[2], Fce, *fused.cond_path_params(est.decoder))
    params = fused.branch_params(est)
    metas = ((est.n_fft, est.hop_length, up, est.ifft.window), ("scale", 1.0))
    lens_cpu = None if audio_lens is None else [int(v) for v in audio_lens]
    return fused.ModelEvalFn.apply(audio.float(), t.flatten().contiguous().float(), None, metas, lens_cpu,
                                   est.training, (len(params),), cproj, *params)


# ------------------------------------------------------------------------------ condition encoder
def cond_encoder_forward(enc, x, mask=None):
    """CondEncoder.forward (modules.py:524-542): (B, n_mels, F) -> (B, channels, F)."""
    B, _, F = x.shape
    rows = fused.CondEncoderFn.apply(x.float(), enc.training, mask_to_lens(mask, B, F),
                                     *fused.cond_encoder_params(enc))
    return RowsToBctFn.apply(rows, B, enc.channels, F)


# ------------------------------------------------------------------------------ GAN loss terms
class HingeMeanFn(torch.autograd.Function):
    """mean(relu(1 + sgn * s)) over all elements of a score map (gan.py:57-75)."""

    @staticmethod
    def forward(ctx, s, sgn: float):
        s = s.contiguous().float()
        n = s.numel()
        loss = ops.zeros(1, device=s.device)
        ops.hinge_loss(loss, None, s, n, sgn, 1.0 / n)
        ctx.saved = s
        ctx.sgn = sgn
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        s = ctx.saved
        n = s.numel()
        gs = torch.empty_like(s)
        ops.hinge_loss(None, gs, s, n, ctx.sgn, 1.0 / n, wdev=g.reshape(1).contiguous())
        return gs, None


class L1MeanFn(torch.autograd.Function):
    """mean |r - f| with the gradient going to f only (gan.py:77-87: `r.detach()`)."""

    @staticmethod
    def forward(ctx, r, f):
        r, f = r.detach().contiguous().float(), f.contiguous().float()
        assert r.shape == f.shape, (r.shape, f.shape)
        n = f.numel()
        loss = ops.zeros(1, device=f.device)
        ops.l1_loss(loss, None, r.view(1, n), f.view(1, n), 1, n, n, 1.0 / n)
        ctx.saved = (r, f)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        r, f = ctx.saved
        n = f.numel()
        gf = torch.empty_like(f)
        ops.l1_loss(None, gf.view(1, n), r.view(1, n), f.view(1, n), 1, n, n, 1.0 / n,
                    wdev=g.reshape(1).contiguous())
        ctx.saved = None
        return None, gf


class SumFn(torch.autograd.Function):
    """Sum of scalar loss terms on the device (no host round trip, no ATen arithmetic)."""

    @staticmethod
    def forward(ctx, *terms):
        v = torch.cat([t.reshape(1) for t in terms])
        out = ops.zeros(1, device=v.device)
        ops.colsum(out, v.view(-1, 1), v.numel(), 1)
        ctx.n = len(terms)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in range(ctx.n))


def discriminator_loss(score_real: List, score_fake: List):
    terms = []
    for s_real, s_fake in zip(score_real, score_fake):
        terms.append(HingeMeanFn.apply(s_real, -1.0))
        terms.append(HingeMeanFn.apply(s_fake, 1.0))
    return SumFn.apply(*terms)


def generator_loss(score_fake: List):
    return SumFn.apply(*[HingeMeanFn.apply(s, -1.0) for s in score_fake])


def feature_matching_loss(fmap_real: List, fmap_fake: List):
    terms = []
    for f_real, f_fake in zip(fmap_real, fmap_fake):
        assert isinstance(f_real, list) and isinstance(f_fake, list)
        for r, f in zip(f_real, f_fake):
            terms.append(L1MeanFn.apply(r, f))
    return SumFn.apply(*terms)
