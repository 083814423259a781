"""Fused forward/backward schedules of the GAN-stage loss stack on the HIP kernels:
multi-period discriminator (reference discriminators.py:18-107), multi-resolution STFT
discriminator (discriminators.py:110-219), hinge / feature-matching / multi-scale mel losses
(gan.py:57-99).  Math: SURVEY.md appendix A.8-A.10.

Real and generated audio are stacked to one (2B, T) batch so that every conv runs once; all
feature maps are channels-last rows x C and every conv (forward, data gradient by stride residue,
weight gradient) is an implicit GEMM on the fp32 matrix cores.  Each multi-discriminator is ONE
autograd node that returns its loss terms; upstream gradients enter the kernels as device scalars
(no host sync).  D-step: gradients for the discriminator parameters only.  G-step: gradient for
the generated audio only (the reference also computes, then discards, the discriminator weight
gradients -- skipping them does not change any generator gradient).
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from ._opts import opt
from .fused import filterbank_spec, filterbank_spec_bwd
from .models.modules import dft_matrices
from .ops import gemm, mat, win1d, win2d

SLOPE = 0.1  # LeakyReLU slope, discriminators.py:94,205


def stack_pair(real, fake):
    B, T = real.shape
    x2 = ops.empty(2 * B, T, device=real.device)
    ops.copy3(x2, 0, T, real.contiguous(), 0, T, 1, B, T)
    ops.copy3(x2, 0, T, fake.contiguous(), 0, T, 1, B, T, out_offset=B * T)
    return x2


def pack_conv_weight(w):
    """(Cout, Cin, kh, kw) -> (Cout, kh*kw*Cin): window-major, channel-minor."""
    Cout, Cin, kh, kw = w.shape
    out = ops.empty(Cout, kh * kw * Cin, device=w.device)
    ops.permute4(out, w, (Cout, kh * kw, Cin, 1), (Cin * kh * kw, 1, kh * kw, 0))
    return out


def unpack_conv_grad(gp, shape):
    Cout, Cin, kh, kw = shape
    out = ops.empty(Cout, Cin, kh, kw, device=gp.device)
    ops.permute4(out, gp, (Cout, Cin, kh * kw, 1), (Cin * kh * kw, 1, Cin, 0))
    return out


def _residues(k: int, stride: int, pad: int, n_in: int):
    """Transposed-conv decomposition: for input residue rho, the taps j = j0 + stride*i that reach
    it, the first source offset and the number of input positions."""
    out = []
    for rho in range(stride):
        j0 = (rho + pad) % stride
        nt = len(range(j0, k, stride))
        e0 = (rho + pad - j0) // stride
        Lq = (n_in - rho + stride - 1) // stride
        out.append((rho, j0, nt, e0, max(Lq, 0)))
    return out


# =====================================================================================
# Multi-period discriminator
# =====================================================================================
MPD_CH = (1, 32, 128, 512, 1024, 1024)
MPD_STRIDE = (3, 3, 3, 3, 1)


def mpd_params(mpd) -> list:
    p = []
    for d in mpd.discriminators:
        for c in d.convs:
            p += [c.weight, c.bias]
        p += [d.conv_post.weight, d.conv_post.bias]
    return p


HALO = 2  # zero rows kept on both sides of every sequence of an MPD map (conv padding (2, 0))


def _halo_rows(S: int, H: int, Cc: int, dev, x3: bool = False):
    """(S, H + 2*HALO, Cc) channels-last map whose halo rows are zero: the (5,1)/(3,1) convs and
    their data gradients read it as plain strided windows (no bounds tests in the GEMM K loop)."""
    buf = ops.empty(S * (H + 2 * HALO), Cc, device=dev)
    if Cc % 4 == 0:
        ops.zero_halo(buf, S, H + 2 * HALO, Cc, HALO, HALO)
    else:
        ops.fill_(buf, 0.0)
    if x3:     # (bf16x6 mode) room for the map's three-piece image, written by its producers' epilogues
        ops.x3_reserve(buf, (S, H + 2 * HALO, Cc, HALO, HALO))
    return buf


def _halo_map(H: int, Cc: int):
    """Epilogue row map (sequence, position) -> row HALO + position of the padded layout."""
    return (H, (H + 2 * HALO) * Cc, Cc, HALO * Cc)


def unhalo(y, S: int, H: int):
    """(S*(H+2*HALO), C) padded map -> (S, H, C) view of the valid rows."""
    return y.view(S, H + 2 * HALO, y.shape[1])[:, HALO:HALO + H]


def _mpd_forward_one(x2, p: int, prm: list, keep_images: bool = False):
    """x2 (2B, T) -> dict with per-layer activations (channels-last; layers 1..5 in the halo
    layout, see _halo_rows), heights, scores."""
    dev = x2.device
    S2, T = x2.shape
    H = (T + p - 1) // p
    S = S2 * p
    img = ops.empty(S * H, 1, device=dev)
    ops.period_fold(img, x2, S2, T, p, H)
    acts, hs = [img], [H]
    shares = [None] * 6   # (split-bf16 mode) images of the maps, kept for the weight gradients
    x = img
    for l in range(5):
        w, b = prm[2 * l], prm[2 * l + 1]
        Cin, Cout, st = MPD_CH[l], MPD_CH[l + 1], MPD_STRIDE[l]
        Hout = (H + 4 - 5) // st + 1
        wp = ops.derived(w, "pack", pack_conv_weight)
        y = _halo_rows(S, Hout, Cout, dev, x3=(0 < l < 4))   # (the next layer's GEMM operand)
        if l == 0 and ops.MPD0_DIRECT and Cout == 32 and st == 3:
            # 1 -> 32 channels, 5 taps: an HBM stream, not a GEMM (mpd0.hip)
            ops.mpd0_fwd(x, S, H, Hout, HALO, w.reshape(Cout, 5), b, SLOPE, y)
            acts.append(y)
            hs.append(Hout)
            x, H = y, Hout
            continue
        if l == 0:   # the folded waveform has no halo (one channel): bounds-tested windows
            A = win1d(x, S, H, Cin, Hout, st, 2, 5)
        else:        # window of output row h starts at padded row h*st
            A = win1d(x, S, H + 2 * HALO, Cin, Hout, st, 0, 5)
        if keep_images and l > 0:
            shares[l] = ops.split_sharing(x)
            with shares[l]:
                gemm(A, mat(wp), y, bias=b, lrelu=SLOPE, rowmap=_halo_map(Hout, Cout), x3_out=(l < 4))
        else:
            gemm(A, mat(wp), y, bias=b, lrelu=SLOPE, rowmap=_halo_map(Hout, Cout), x3_out=(0 < l < 4))
        acts.append(y)
        hs.append(Hout)
        x, H = y, Hout
    wpost, bpost = prm[10], prm[11]
    wpp = ops.derived(wpost, "pack", pack_conv_weight)
    scores = ops.empty(S * H, 1, device=dev)
    if ops.MPD0_DIRECT and x.shape[1] == 1024:    # 1024 -> 1 channel, 3 taps: an HBM stream (mpd0.hip)
        ops.mpdpost_fwd(x, S, H, HALO, wpp, bpost, scores)
    else:
        gemm(win1d(x, S, H + 2 * HALO, 1024, H, 1, -(HALO - 1), 3), mat(wpp), scores, bias=bpost)
    return dict(acts=acts, hs=hs, scores=scores, S=S, p=p, shares=shares)


def _dgrad_weight(w, stride: int, j0: int, nt: int):
    """Weights of one stride residue of the transposed conv as a forward GEMM operand
    [Cin][nt*Cout] (k index = tap-major, channel-minor, taps in window order)."""
    def build(t):
        Cout, Cin, K = t.shape[0], t.shape[1], t.shape[2] * t.shape[3]
        out = ops.empty(Cin, nt * Cout, device=t.device)
        ops.permute4(out, t, (Cin, nt, Cout, 1), (K, -stride, Cin * K, 0),
                     in_offset=j0 + stride * (nt - 1))
        return out
    return ops.derived(w, ("dgradT", stride, j0, nt), build)


def _conv1d_dgrad(g_pre, S, Hout, Cout, w, stride, pad, Hin, g_off=0, g_halo=True, out_halo=True,
                  mask=None, fm=None, colsum=None):
    """g_x from g_pre (S sequences of Hout rows x Cout, starting g_off floats in; halo layout when
    g_halo).  Returns g_x in the halo layout (S, Hin + 2*HALO, Cin) when out_halo, else (S*Hin, Cin).
    One forward-form GEMM per stride residue against the cached re-laid weights."""
    Cin, K = w.shape[1], w.shape[2] * w.shape[3]
    dev = g_pre.device
    # (bf16x6 mode: the gradient map's image for the data gradient of the layer below -- only where that GEMM
    # reads images: its residues reduce over at most two taps x Cin, and below X6_MIN_K the kernel splits the
    # fp32 map itself; the 512-channel map's image was written for nobody: 192 KB per 256 x 128 tile)
    want_img = 2 * Cin >= ops.X6_MIN_K
    gx = _halo_rows(S, Hin, Cin, dev, x3=want_img) if out_halo else ops.empty(S * Hin, Cin, device=dev)
    for rho, j0, nt, e0, Lq in _residues(K, stride, pad, Hin):
        if Lq == 0:
            continue
        wq = _dgrad_weight(w, stride, j0, nt)
        wpad = (nt - 1) - e0
        if g_halo:
            A = win1d(g_pre, S, Hout + 2 * HALO, Cout, Lq, 1, wpad - HALO, nt, offset=g_off)
        else:
            A = win1d(g_pre, S, Hout, Cout, Lq, 1, wpad, nt, offset=g_off)
        if out_halo:
            rm = (Lq, (Hin + 2 * HALO) * Cin, stride * Cin, (HALO + rho) * Cin)
        else:
            rm = (Lq, Hin * Cin, stride * Cin, rho * Cin)
        # mask / fm / colsum: leaky-ReLU backward of the layer whose output gradient this is (every
        # element of gx is written by exactly one stride residue)
        gemm(A, mat(wq), gx, rowmap=rm, mask=mask, fm=fm, colsum=colsum, x3_out=out_halo and want_img)
    return gx


class MPDLossFn(torch.autograd.Function):
    """(real, fake) -> (loss0, loss1): D-step (hinge_D, 0); G-step (hinge_G, feature matching)."""

    @staticmethod
    def forward(ctx, real, fake, train_disc: bool, periods, *params):
        dev = real.device
        B, T = real.shape
        x2 = stack_pair(real, fake)
        losses = ops.zeros(2, device=dev)
        saved = []
        lanes = ops.Lanes(dev, len(periods), "mpd")  # one launch lane per sub-discriminator
        for i, p in enumerate(periods):
          with lanes.lane(i):
            prm = list(params[12 * i: 12 * i + 12])
            st = _mpd_forward_one(x2, p, prm, keep_images=train_disc)
            S, H5 = st["S"], st["hs"][5]
            nh = (S // 2) * H5  # elements of one half's score map
            sc = st["scores"]
            if train_disc:
                ops.hinge_loss(losses, None, sc, nh, -1.0, 1.0 / nh)
                ops.hinge_loss(losses, None, sc, nh, +1.0, 1.0 / nh, s_off=nh)
            else:
                ops.hinge_loss(losses, None, sc, nh, -1.0, 1.0 / nh, s_off=nh)
                for l in range(2, 6):  # fmaps: conv layers 1..4 (discriminators.py:95-96)
                    y = st["acts"][l]
                    Hl, Cl = st["hs"][l], y.shape[1]
                    nflat = (S // 2) * (Hl + 2 * HALO) * Cl     # halo rows are 0 on both sides
                    nval = (S // 2) * Hl * Cl
                    ops.l1_loss_ab(losses, None, y, 0, y, nflat, 1, nflat, nflat, 1.0 / nval,
                                   loss_offset=1)
                ops.l1_loss_ab(losses, None, sc, 0, sc, nh, 1, nh, nh, 1.0 / nh, loss_offset=1)
            saved.append(st)
        lanes.join()
        ctx.saved = saved
        ctx.params = params
        ctx.meta = (B, T, train_disc, tuple(periods))
        # D-step: each period discriminator's gradients are handed to an armed exchange as soon as
        # its launch lane has finished its backward (dist._Sink)
        from .fused import sink_register
        ctx.tickets = [sink_register(list(params[12 * i: 12 * i + 12])) if train_disc else None
                       for i in range(len(periods))]
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, g1):
        B, T, train_disc, periods = ctx.meta
        params = ctx.params
        dev = g0.device
        g0 = g0.reshape(1).contiguous()
        g1 = g1.reshape(1).contiguous()
        pgrads: List = []
        g_fake = None if train_disc else ops.zeros(B, T, device=dev)
        lanes = ops.Lanes(dev, len(periods), "mpd")
        for i, p in enumerate(periods):
          with lanes.lane(i):
            prm = list(params[12 * i: 12 * i + 12])
            st = ctx.saved[i]
            acts, hs, sc, S = st["acts"], st["hs"], st["scores"], st["S"]
            H5 = hs[5]
            nh = (S // 2) * H5
            wpost = prm[10]
            if train_disc:
                gs = ops.empty(S * H5, 1, device=dev)
                ops.hinge_loss(None, gs, sc, nh, -1.0, 1.0 / nh, wdev=g0)
                ops.hinge_loss(None, gs, sc, nh, +1.0, 1.0 / nh, wdev=g0, s_off=nh)
                Sx, roff = S, 0            # sequences taking part in backward, row offset (seqs)
            else:
                gs = ops.empty(S * H5, 1, device=dev)  # only the fake half is used
                ops.hinge_loss(None, gs, sc, nh, -1.0, 1.0 / nh, wdev=g0, s_off=nh)
                ops.lrelu_bwd(gs, sc, sc, 1.0 / nh, 1.0, 1, nh, nh, wdev=g1, g_off=nh, y_off=nh,
                              r_off=0)
                Sx, roff = S // 2, S // 2
            grads_p = [None] * 12
            # conv_post (1024 -> 1, 3 taps, stride 1)
            y5 = acts[5]
            if train_disc:
                # every weight / bias gradient accumulator of this sub-discriminator: one fill
                zshapes = [(1, 3 * 1024), (1,)]
                for l in range(5):
                    zshapes += [(MPD_CH[l + 1], 5 * MPD_CH[l]), (MPD_CH[l + 1],)]
                zbuf = ops.zeros_many(zshapes, dev)
                gwp = zbuf[0]
                if ops.MPD0_DIRECT and y5.shape[1] == 1024:
                    ops.mpdpost_wgrad(y5, S, H5, HALO, gs, gwp)
                else:
                    ops.wgrad(gs, 1, 1, win1d(y5, S, H5 + 2 * HALO, 1024, H5, 1, -(HALO - 1), 3), gwp)
                unpack = [(10, gwp, wpost.shape)]     # (re-laid at the end of the period: ONE f2g_multi launch)
                gb = zbuf[1]
                ops.colsum(gb, gs, S * H5, 1)
                grads_p[11] = gb
            # Every data gradient applies the leaky-ReLU backward of the layer it lands on in its own
            # epilogue (mask by that layer's activation, in the G-step plus the feature-matching
            # term) and leaves the column sums = the bias gradient of the conv that produced it:
            # the maps are not re-read by a separate pass.
            def below(l_out):
                """(mask, fm, colsum) for a gradient that lands on acts[l_out] (l_out = 1..5)."""
                y = acts[l_out]
                Hp_, C_ = hs[l_out] + 2 * HALO, y.shape[1]
                yo = roff * Hp_ * C_
                fm_ = None
                if (not train_disc) and l_out >= 2:     # fmaps: conv layers 1..4 (acts[2..5])
                    fm_ = (y, 0, 1.0 / (Sx * hs[l_out] * C_), g1)
                cs_ = zbuf[2 + 2 * (l_out - 1) + 1] if train_disc else None
                return (y, yo, SLOPE), fm_, cs_

            def land(gmap, l_out, producer):
                """Run `producer(mask, fm, colsum)` -> gradient map landing on acts[l_out], with its
                leaky-ReLU backward fused (FUSE_LRELU & 1) or as a separate pass."""
                mk, fk, ck = below(l_out)
                # (bf16x6 mode: where the data gradient runs on the six-product kernel -- the 1024-channel
                # layers, K >= 2048 -- that kernel has the generic epilogue anyway, and the image it
                # leaves for the next data gradient must be of the final map)
                if (FUSE_LRELU & 1) or (ops.GEMM_PRECISION == 3 and l_out in (3, 4)):
                    return producer(mk, fk, ck)
                gm = producer(None, None, None)
                gm._f2g_x3_bad = True      # (changed in place below: no producer-written image of it)
                y = acts[l_out]
                Hp_, C_ = hs[l_out] + 2 * HALO, y.shape[1]
                n_ = Sx * Hp_ * C_
                if train_disc:
                    ops.lrelu_bwd_colsum(gm, y, None, 0.0, SLOPE, Sx * Hp_, C_, C_, ck, y_off=mk[1])
                elif fk is not None:
                    ops.lrelu_bwd(gm, y, y, fk[2], SLOPE, 1, n_, n_, wdev=g1, y_off=mk[1], r_off=0)
                else:
                    ops.lrelu_bwd(gm, y, None, 0.0, SLOPE, 1, n_, n_, y_off=mk[1])
                return gm

            def post_dgrad(mk, fk, ck):
                if ops.MPD0_DIRECT and y5.shape[1] == 1024 and (mk is None or MPDPOST_FUSE):
                    # (round 5) the stream kernel applies the mask / feature-matching term / bias sums itself and
                    # leaves the image the fp32-class data gradient of the 1024-channel layer reads next
                    gy5 = _halo_rows(Sx, H5, 1024, dev, x3=mk is not None)
                    return ops.mpdpost_dgrad(gs, Sx, H5, HALO, ops.derived(wpost, "pack", pack_conv_weight),
                                             gy5, g_off=roff * H5, mask=mk, fm=fk, colsum=ck)
                return _conv1d_dgrad(gs, Sx, H5, 1, wpost, 1, 1, H5, g_off=roff * H5, g_halo=False,
                                     mask=mk, fm=fk, colsum=ck)
            if ops.MPD0_DIRECT and y5.shape[1] == 1024 and MPDPOST_FUSE:
                g = post_dgrad(*below(5))
            else:
                g = land(None, 5, post_dgrad)
            for l in reversed(range(5)):
                w = prm[2 * l]
                Cin, Cout, stv = MPD_CH[l], MPD_CH[l + 1], MPD_STRIDE[l]
                Hin, Hout = hs[l], hs[l + 1]
                Hp = Hout + 2 * HALO
                # g: gradient of layer l's PRE-activation (S or Sx sequences, halo layout);
                # (split-bf16 mode) one image of it serves the weight gradient and every stride
                # residue of the data gradient, the forward image of acts[l] the weight gradient
                with ops.split_sharing(g), (st["shares"][l] or ops.split_sharing()):
                    if train_disc:
                        grads_p[2 * l + 1] = zbuf[2 + 2 * l + 1]
                        # reduction over ALL rows of the padded gradient map (its halo rows are 0, so
                        # the windows they pair with -- partly outside the input -- contribute nothing)
                        gwp = zbuf[2 + 2 * l]
                        if l == 0 and ops.MPD0_DIRECT and Cout == 32 and stv == 3:
                            ops.mpd0_wgrad(acts[0], S, Hin, Hout, HALO, g, gwp)
                        else:
                            if l == 0:
                                X = win1d(acts[0], S, Hin, Cin, Hp, stv, 2 + HALO * stv, 5)
                            else:
                                X = win1d(acts[l], S, Hin + 2 * HALO, Cin, Hp, stv, HALO * stv, 5,
                                          unbounded=True)   # g's halo rows are zero
                            ops.wgrad(g, Cout, Cout, X, gwp)
                        unpack.append((2 * l, gwp, w.shape))
                    if l > 0:
                        g = land(None, l, lambda mk, fk, ck, g=g, w=w, Hout=Hout, Cout=Cout, stv=stv, Hin=Hin:
                                 _conv1d_dgrad(g, Sx, Hout, Cout, w, stv, 2, Hin, mask=mk, fm=fk, colsum=ck))
                    elif not train_disc and ops.MPD0_DIRECT and Cout == 32 and stv == 3:
                        gx0 = ops.empty(Sx * Hin, 1, device=dev)
                        g = ops.mpd0_dgrad(g, Sx, Hin, Hout, HALO, w.reshape(Cout, 5), gx0)
                    elif not train_disc:
                        g = _conv1d_dgrad(g, Sx, Hout, Cout, w, stv, 2, Hin, out_halo=False)
            if not train_disc:
                # g: (B*p*H0, 1) gradient of the folded image of the generated half
                lanes.chain_enter()  # g_fake is accumulated period after period
                ops.period_fold_bwd(g_fake, g, B, T, p, hs[0], True)
                lanes.chain_leave()
            if train_disc:
                with ops.weight_batch():
                    for slot_, gwp_, shape_ in unpack:
                        grads_p[slot_] = unpack_conv_grad(gwp_, shape_)
            from .fused import deliver_grads
            pgrads += deliver_grads(ctx.tickets[i], list(params[12 * i: 12 * i + 12]), grads_p)
        lanes.join()
        ctx.saved = None
        return tuple([None, g_fake, None, None] + pgrads)


# =====================================================================================
# Multi-resolution STFT discriminator
# =====================================================================================
MRD_BANDS = ((0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0))
MRD_LAYERS = ((9, 1), (9, 2), (9, 2), (9, 2), (3, 1))  # (kw, stride_w); kh = 3, pad = (1, kw//2)
MRD_CH = 32

_DFT_INT_CACHE = {}


def dft_interleaved(n_fft: int, device):
    """Analysis matrix with rows [Re0, Im0, Re1, Im1, ...]: the STFT GEMM then writes the
    (frame, freq, {re,im}) channels-last image of discriminators.py:191-193 directly."""
    key = (n_fft, str(device))
    if key not in _DFT_INT_CACHE:
        wd, _ = dft_matrices(n_fft, device)
        nb = n_fft // 2 + 1
        wi = torch.stack([wd[:nb], wd[nb:]], dim=1).reshape(2 * nb, n_fft).contiguous()
        wi._f2g_const = True
        _DFT_INT_CACHE[key] = wi
    return _DFT_INT_CACHE[key]


def mrd_params(mrd) -> list:
    p = []
    for d in mrd.discriminators:
        for stack in d.band_convs:
            for c in stack:
                p += [c.weight, c.bias]
        p += [d.conv_post.weight, d.conv_post.bias]
    return p


N_MRD_PARAMS = 5 * 5 * 2 + 2


import os as _os

# F2G_DIRECT_CONV=0 routes the band layers through the implicit GEMM again (A/B switch)
DIRECT_CONV32 = opt("direct_conv", True)
# leaky-ReLU backward fused into the data-gradient epilogue that lands on a map (1) or run as its
# own pass over the map afterwards (0): bit 0 = MPD, bit 1 = MRD.  Measured on the stage-2 step
# (B = 64, 10 steps each): separate passes 257.5 ms, MPD fused 258.6, MRD fused 260.6, both 261.2 --
# the HBM-bound passes overlap with other lanes' MFMA work, while masking in the epilogue (two more
# loads per element, strided by the row map) holds an MFMA wave's registers and LDS idle.  Default 0.
FUSE_LRELU = opt("fuse_lrelu", 0)
# the five frequency bands of a resolution as nested launch lanes (their conv stacks are independent
# until conv_post; the narrow bands' launches fill a fraction of the chip): 0 = one after the other
BAND_LANES = opt("band_lanes", False)
# D-step of the MRD on the direct fp32-class kernels: leaky-ReLU backward mask + bias-gradient sums fused into
# the data gradients (conv32x6.hip requests a tile's mask before its MFMAs); 0 = the separate passes
MRD_FUSE_MASK = opt("mrd_fuse_mask", True)
# conv_post's data gradient (mpd0.hip stream kernel) with the leaky-ReLU backward of the layer below, its bias
# sums and the result's image fused; 0 = the separate pass over the 1024-channel map
MPDPOST_FUSE = opt("mpdpost_fuse", True)


def _band_edges(n_fft: int):
    nb = n_fft // 2 + 1
    return [(int(lo * nb), int(hi * nb)) for lo, hi in MRD_BANDS]


def _mrd_forward_one(x2, win: int, prm: list):
    dev = x2.device
    S, T = x2.shape
    hop = win // 4
    Ft = 1 + T // hop
    nb = win // 2 + 1
    xn = ops.empty(S, T, device=dev)
    stats = ops.empty(S, 3, device=dev)
    ops.peaknorm_fwd(xn, stats, x2, S, T)
    ldp = ops.pad4(2 * nb)
    packed = ops.empty(S * Ft, ldp, device=dev)
    if ops.fft_applies(win):
        ops.stft_fft(xn, win, hop, Ft, packed, interleaved=True)
    else:
        gemm(ops.stft_frames(xn, win, hop, Ft), mat(dft_interleaved(win, dev)),
             packed, split_k=1)
    bands = _band_edges(win)
    # widths per band per layer
    widths = []
    for lo, hi in bands:
        w = [hi - lo]
        for kw, sw in MRD_LAYERS:
            w.append((w[-1] + 2 * (kw // 2) - kw) // sw + 1)
        widths.append(w)
    Wcat = sum(w[5] for w in widths)
    cat = ops.empty(S * Ft * Wcat, MRD_CH, device=dev)
    acts = []
    foff = 0
    foffs = []
    for bi in range(len(bands)):
        foffs.append(foff)
        foff += widths[bi][5]
    blanes = ops.Lanes(dev, len(bands) if BAND_LANES else 1, "mrd_band%d" % win)
    for bi, (lo, hi) in enumerate(bands):
      with blanes.lane(bi if BAND_LANES else 0):
        foff = foffs[bi]
        ws = widths[bi]
        layer_out = []
        x, x_is_spec = packed, True
        for l, (kw, sw) in enumerate(MRD_LAYERS):
            w, b = prm[(bi * 5 + l) * 2], prm[(bi * 5 + l) * 2 + 1]
            Cin = 2 if l == 0 else MRD_CH
            Win, Wout = ws[l], ws[l + 1]
            wp = ops.derived(w, "pack", pack_conv_weight)
            if x_is_spec:
                A = win2d(x, S, Ft, Win, Cin, Wout, 3, kw, sw, 1, kw // 2, line_stride=ldp,
                          seq_stride=Ft * ldp, offset=lo * 2)
            else:
                A = win2d(x, S, Ft, Win, Cin, Wout, 3, kw, sw, 1, kw // 2)
            if l == 0 and DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                # 2 -> 32 channels over the band of the spectrogram: direct kernel (conv2ch.hip)
                y = ops.empty(S * Ft * Wout, MRD_CH, device=dev)
                ops.conv2ch_fwd(packed, Ft * ldp, ldp, lo * 2, S, Ft, Win, wp, b, SLOPE, y)
            elif l in (1, 2, 3) and DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                # 32 -> 32 channels, (3, 9) taps, stride (1, 2): direct LDS-tiled kernel (conv32.hip)
                y = ops.empty(S * Ft * Wout, MRD_CH, device=dev)
                ops.conv32_s2_fwd(x, S, Ft, Win, Wout, wp, b, SLOPE, y)
            elif l < 4:
                y = ops.empty(S * Ft * Wout, MRD_CH, device=dev)
                gemm(A, mat(wp), y, bias=b, lrelu=SLOPE)
            elif DIRECT_CONV32 and ops.GEMM_PRECISION == 3 and ops.CONV33_X6 and Wout <= ops.CONV33_MAX_W:
                # 32 -> 32 channels, (3, 3) taps, stride 1, into the band's slice of the concatenated map
                y = None
                ops.conv33(x, S, Ft, Wout, wp, b, SLOPE, cat, y_off=foff * MRD_CH, y_line=Wcat * MRD_CH,
                           y_seq=Ft * Wcat * MRD_CH)
            else:
                y = None
                gemm(A, mat(wp), cat, bias=b, lrelu=SLOPE,
                     rowmap=(Wout, Wcat * MRD_CH, MRD_CH, foff * MRD_CH))
            layer_out.append(y)
            x, x_is_spec = y, False
        acts.append(layer_out)
    blanes.join()
    wpost, bpost = prm[50], prm[51]
    scores = ops.empty(S * Ft * Wcat, 1, device=dev)
    w9 = ops.derived(wpost, "pack", pack_conv_weight)      # (1, 9*32): [tap][ci]
    if DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
        ops.convpost_fwd(cat, S, Ft, Wcat, w9, bpost, scores)
    else:
        gemm(win2d(cat, S, Ft, Wcat, MRD_CH, Wcat, 3, 3, 1, 1, 1), mat(w9), scores, bias=bpost)
    return dict(xn=xn, stats=stats, packed=packed, ldp=ldp, Ft=Ft, nb=nb, hop=hop, bands=bands,
                widths=widths, Wcat=Wcat, cat=cat, acts=acts, scores=scores)


def _conv2d_dgrad(g_pre, S, H, Wout, Cout, w, sw, Win, gx, *, g_line=None, g_seq=None, g_off=0,
                  x_line=None, x_off=0, mask=None, fm=None, colsum=None):
    """Data gradient of a (3, kw) conv with stride (1, sw), pad (1, kw//2).
    g_pre: image (S, H, Wout, Cout) starting g_off floats in (line / seq strides optional);
    gx: destination image with line stride x_line floats (default Win*Cin) starting x_off."""
    Cin, kh, kw = w.shape[1], w.shape[2], w.shape[3]
    dev = g_pre.device
    pw = kw // 2
    if (DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3) and Cin == MRD_CH and Cout == MRD_CH and kw == 9
            and sw == 2 and x_line is None and x_off == 0):
        # 32 -> 32 channels, (3, 9) taps, stride (1, 2): direct transposed convolution (conv32.hip)
        def build_t(t):
            out = ops.empty(kh * kw, Cin, Cout, device=t.device)
            ops.permute4(out, t, (kh * kw, Cin, Cout, 1), (1, kh * kw, Cin * kh * kw, 0))
            return out
        wT = ops.derived(w, "dgrad_taps", build_t)
        ops.conv32_s2_dgrad(g_pre, S, H, Win, Wout, wT, gx, g_seq=g_seq, g_line=g_line, g_off=g_off,
                            mask=mask, fm=fm, colsum=colsum)
        return gx
    if (DIRECT_CONV32 and ops.GEMM_PRECISION == 3 and ops.CONV33_X6 and Cin == MRD_CH and Cout == MRD_CH
            and kh == 3 and kw == 3 and sw == 1 and x_line is None and x_off == 0 and fm is None
            and Win <= ops.CONV33_MAX_W):
        # stride 1: the data gradient is the forward kernel over the gradient map with the taps flipped and
        # the channel matrix transposed, [ci][8 - tap][co]
        def build_f(t):
            out = ops.empty(Cin, kh * kw * Cout, device=t.device)
            ops.permute4(out, t, (Cin, kh * kw, Cout, 1), (kh * kw, -1, Cin * kh * kw, 0), in_offset=kh * kw - 1)
            return out
        wf = ops.derived(w, "dgrad33", build_f)
        ops.conv33(g_pre, S, H, Win, wf, None, 0.0, gx, x_off=g_off, x_line=g_line, x_seq=g_seq, form=1,
                   mask=mask, colsum=colsum)
        return gx
    if x_line is None:
        x_line = Win * Cin
    for rho, j0, ntw, e0, Lq in _residues(kw, sw, pw, Win):
        if Lq == 0:
            continue
        def build(t, j0=j0, ntw=ntw):
            out = ops.empty(kh * ntw * Cout, Cin, device=t.device)
            ops.permute4(out, t, (kh, ntw, Cout, Cin), (-kw, -sw, Cin * kh * kw, kh * kw),
                         in_offset=(kh - 1) * kw + j0 + sw * (ntw - 1))
            return out
        wq = ops.derived(w, ("dgrad2d", sw, j0, ntw), build)
        A = win2d(g_pre, S, H, Wout, Cout, Lq, kh, ntw, 1, 1, (ntw - 1) - e0, line_stride=g_line,
                  seq_stride=g_seq, offset=g_off)
        gemm(A, mat(wq), gx, form=1, rowmap=(Lq, x_line, sw * Cin, x_off + rho * Cin), mask=mask,
             fm=fm, colsum=colsum)
    return gx


class MRDLossFn(torch.autograd.Function):
    """(real, fake) -> (loss0, loss1) as MPDLossFn, for the STFT-band discriminators."""

    @staticmethod
    def forward(ctx, real, fake, train_disc: bool, fft_sizes, *params):
        dev = real.device
        B, T = real.shape
        x2 = stack_pair(real, fake)
        losses = ops.zeros(2, device=dev)
        saved = []
        for win in fft_sizes:
            dft_interleaved(win, dev)  # cached constants are created on the caller's stream
        lanes = ops.Lanes(dev, len(fft_sizes), "mrd")  # one launch lane per STFT resolution
        for i, win in enumerate(fft_sizes):
          with lanes.lane(i):
            prm = list(params[N_MRD_PARAMS * i: N_MRD_PARAMS * (i + 1)])
            st = _mrd_forward_one(x2, win, prm)
            Ft, Wcat = st["Ft"], st["Wcat"]
            nh = B * Ft * Wcat
            sc = st["scores"]
            if train_disc:
                ops.hinge_loss(losses, None, sc, nh, -1.0, 1.0 / nh)
                ops.hinge_loss(losses, None, sc, nh, +1.0, 1.0 / nh, s_off=nh)
            else:
                ops.hinge_loss(losses, None, sc, nh, -1.0, 1.0 / nh, s_off=nh)
                foff = 0
                for bi in range(5):
                    ws = st["widths"][bi]
                    for l in range(1, 4):  # band layers 1..3 (contiguous maps)
                        y = st["acts"][bi][l]
                        n = y.numel() // 2
                        ops.l1_loss_ab(losses, None, y, 0, y, n, 1, n, n, 1.0 / n, loss_offset=1)
                    # layer 4 lives in the concatenated buffer: (B*Ft rows, W4*32 cols) slice
                    cat = st["cat"]
                    cols = ws[5] * MRD_CH
                    ld = Wcat * MRD_CH
                    half = B * Ft * ld
                    ops.l1_loss_ab(losses, None, cat, foff * MRD_CH, cat, half + foff * MRD_CH,
                                   B * Ft, cols, ld, 1.0 / (B * Ft * cols), loss_offset=1)
                    foff += ws[5]
                ops.l1_loss_ab(losses, None, sc, 0, sc, nh, 1, nh, nh, 1.0 / nh, loss_offset=1)
            saved.append(st)
        lanes.join()
        ctx.saved = saved
        ctx.x2 = x2
        ctx.params = params
        ctx.meta = (B, T, train_disc, tuple(fft_sizes))
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, g1):
        B, T, train_disc, fft_sizes = ctx.meta
        params = ctx.params
        dev = g0.device
        g0 = g0.reshape(1).contiguous()
        g1 = g1.reshape(1).contiguous()
        pgrads: List = []
        g_fake = None if train_disc else ops.zeros(B, T, device=dev)
        C = MRD_CH
        lanes = ops.Lanes(dev, len(fft_sizes), "mrd")
        for i, win in enumerate(fft_sizes):
          with lanes.lane(i):
            prm = list(params[N_MRD_PARAMS * i: N_MRD_PARAMS * (i + 1)])
            st = ctx.saved[i]
            Ft, Wcat, nb, ldp, hop = st["Ft"], st["Wcat"], st["nb"], st["ldp"], st["hop"]
            sc, cat, packed = st["scores"], st["cat"], st["packed"]
            S = 2 * B
            nh = B * Ft * Wcat
            grads_w = [None] * N_MRD_PARAMS
            gs = ops.empty(S * Ft * Wcat, 1, device=dev)
            if train_disc:
                ops.hinge_loss(None, gs, sc, nh, -1.0, 1.0 / nh, wdev=g0)
                ops.hinge_loss(None, gs, sc, nh, +1.0, 1.0 / nh, wdev=g0, s_off=nh)
                Sx, soff = S, 0
            else:
                ops.hinge_loss(None, gs, sc, nh, -1.0, 1.0 / nh, wdev=g0, s_off=nh)
                ops.lrelu_bwd(gs, sc, sc, 1.0 / nh, 1.0, 1, nh, nh, wdev=g1, g_off=nh, y_off=nh,
                              r_off=0)
                Sx, soff = B, B   # sequences in backward, first sequence
            wpost = prm[50]
            if train_disc:
                gwp = ops.zeros(1, 9 * C, device=dev)
                if DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                    ops.convpost_wgrad(cat, S, Ft, Wcat, gs, gwp)
                else:
                    ops.wgrad(gs, 1, 1, win2d(cat, S, Ft, Wcat, C, Wcat, 3, 3, 1, 1, 1), gwp)
                unpack = [(50, gwp, wpost.shape)]     # (re-laid at the end of the window: ONE f2g_multi launch)
                gb = ops.zeros(1, device=dev)
                ops.colsum(gb, gs, S * Ft * Wcat, 1)
                grads_w[51] = gb
            # gradient of the concatenated layer-4 maps (only the sequences in backward)
            gcat = ops.empty(Sx * Ft * Wcat, C, device=dev)
            if DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                ops.convpost_dgrad(gs, Sx, Ft, Wcat, ops.derived(wpost, "pack", pack_conv_weight),
                                   gcat, g_off=soff * Ft * Wcat)
            else:
                _conv2d_dgrad(gs, Sx, Ft, Wcat, 1, wpost, 1, Wcat, gcat, g_off=soff * Ft * Wcat)
            g_packed = None
            if not train_disc:
                g_packed = ops.empty(B * Ft, ldp, device=dev)
            ldc = Wcat * C
            foffs, foff = [], 0
            for bi in range(len(st["bands"])):
                foffs.append(foff)
                foff += st["widths"][bi][5]
            # (the bands' stacks are independent: disjoint slices of gcat / g_packed, their own maps)
            blanes = ops.Lanes(dev, len(st["bands"]) if BAND_LANES else 1, "mrd_band%d" % win)
            for bi, (lo, hi) in enumerate(st["bands"]):
              with blanes.lane(bi if BAND_LANES else 0):
                foff = foffs[bi]
                ws = st["widths"][bi]
                # ---- layer 4 (its output is a strided slice of cat / gcat)
                W4 = ws[5]
                cols = W4 * C
                y_off = soff * Ft * ldc + foff * C
                if train_disc:
                    ops.lrelu_bwd(gcat, cat, None, 0.0, SLOPE, Sx * Ft, cols, ldc, g_off=foff * C,
                                  y_off=y_off)
                else:
                    ops.lrelu_bwd(gcat, cat, cat, 1.0 / (B * Ft * cols), SLOPE, Sx * Ft, cols, ldc,
                                  wdev=g1, g_off=foff * C, y_off=y_off, r_off=foff * C)
                g = None
                # bias-gradient accumulators of the band (layers 0..3 get theirs as column sums from
                # the data-gradient epilogue that lands on their output)
                if train_disc:   # every accumulator of the band's five layers from one zeroed allocation
                    zs = ops.zeros_many([(C,)] * 5 + [(C, 3 * kw_ * (2 if l_ == 0 else C))
                                                      for l_, (kw_, _sw) in enumerate(MRD_LAYERS)], dev)
                    gbs, gb4, gwps = zs[:4], zs[4], zs[5:]
                else:
                    gbs = [None] * 4
                for l in reversed(range(5)):
                    kw, sw = MRD_LAYERS[l]
                    w = prm[(bi * 5 + l) * 2]
                    Cin = 2 if l == 0 else C
                    Win, Wout = ws[l], ws[l + 1]
                    # (for l < 4, g already is the gradient of layer l's PRE-activation: the data
                    # gradient that produced it applied this layer's leaky-ReLU backward)
                    # operands describing this layer's pre-activation gradient image
                    if l == 4:
                        dy_line, dy_seq, dy_off, dy_t = ldc, Ft * ldc, foff * C, gcat
                    else:
                        dy_line, dy_seq, dy_off, dy_t = None, None, 0, g
                    if train_disc:
                        x_in = packed if l == 0 else st["acts"][bi][l - 1]
                        if l == 0:
                            X = win2d(x_in, S, Ft, Win, Cin, Wout, 3, kw, sw, 1, kw // 2,
                                      line_stride=ldp, seq_stride=Ft * ldp, offset=lo * 2)
                        else:
                            X = win2d(x_in, S, Ft, Win, Cin, Wout, 3, kw, sw, 1, kw // 2)
                        gwp = gwps[l]
                        if l == 4:
                            dY = win1d(gcat, S * Ft, Wcat, C, W4, 1, -foff, 1)
                        else:
                            dY = mat(g, S * Ft * Wout, C)
                        tiles = ((3 * kw * Cin + 255) // 256)
                        if l == 0 and DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                            ops.conv2ch_wgrad(packed, Ft * ldp, ldp, lo * 2, S, Ft, Win, g, gwp)
                        elif l in (1, 2, 3) and DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                            ops.conv32_s2_wgrad(x_in, g, S, Ft, Win, Wout, gwp)
                        elif (l == 4 and DIRECT_CONV32 and ops.GEMM_PRECISION == 3 and ops.CONV33_X6
                              and Wout <= ops.CONV33_MAX_W):
                            # (round 6) the (3, 3) layer's weight gradient as a direct fp32-class kernel over the
                            # band's slice of the concatenated gradient map (31 TFLOP/s as an implicit GEMM)
                            ops.conv33_wgrad(x_in, gcat, S, Ft, Wout, gwp, g_off=foff * C, g_line=ldc, g_seq=Ft * ldc)
                        else:
                            gemm(dY, X, gwp, form=2, atomic=True,
                                 split_k=ops.split_for(X.rows, tiles))
                        unpack.append(((bi * 5 + l) * 2, gwp, w.shape))
                        if l == 4:
                            gb = gb4
                            _colsum_strided(gb, gcat, S * Ft, W4, C, ldc, foff * C)
                        else:
                            gb = gbs[l]
                        grads_w[(bi * 5 + l) * 2 + 1] = gb
                    if l > 0:
                        # lands on acts[bi][l-1]: its leaky-ReLU backward (+ feature matching for the
                        # maps the reference lists, band layers 1..4) rides in this epilogue
                        yb = st["acts"][bi][l - 1]
                        nb_ = Sx * Ft * Win * C
                        mk = (yb, soff * Ft * Win * C, SLOPE)
                        fmk = (yb, 0, 1.0 / nb_, g1) if ((not train_disc) and l - 1 >= 1) else None
                        gx = ops.empty(Sx * Ft * Win, Cin, device=dev)
                        # (round 5) D-step on the direct fp32-class kernels: their data gradients request the
                        # mask of a tile BEFORE its MFMAs, so the fused leaky-ReLU backward (+ bias-gradient
                        # column sums) costs no exposed round trip and saves a pass over the map
                        fuse_d = (train_disc and MRD_FUSE_MASK and DIRECT_CONV32 and ops.GEMM_PRECISION == 3
                                  and ops.CONV32_X6 and (l < 4 or (ops.CONV33_X6 and Win <= ops.CONV33_MAX_W)))
                        if (FUSE_LRELU & 2) or fuse_d:
                            _conv2d_dgrad(dy_t, Sx, Ft, Wout, C, w, sw, Win, gx, g_line=dy_line,
                                          g_seq=dy_seq, g_off=dy_off, mask=mk, fm=fmk,
                                          colsum=gbs[l - 1])
                        else:
                            _conv2d_dgrad(dy_t, Sx, Ft, Wout, C, w, sw, Win, gx, g_line=dy_line,
                                          g_seq=dy_seq, g_off=dy_off)
                            if train_disc:
                                ops.lrelu_bwd_colsum(gx, yb, None, 0.0, SLOPE, Sx * Ft * Win, C, C,
                                                     gbs[l - 1], y_off=mk[1])
                            elif fmk is not None:
                                ops.lrelu_bwd(gx, yb, yb, fmk[2], SLOPE, 1, nb_, nb_, wdev=g1,
                                              y_off=mk[1], r_off=0)
                            else:
                                ops.lrelu_bwd(gx, yb, None, 0.0, SLOPE, 1, nb_, nb_, y_off=mk[1])
                        g = gx
                    elif not train_disc and DIRECT_CONV32 and ops.GEMM_PRECISION in (0, 1, 3):
                        def build_c2t(t):
                            out = ops.empty(27, 2, C, device=t.device)   # [tap][ci][co]
                            ops.permute4(out, t, (27, 2, C, 1), (1, 27, 2 * 27, 0))
                            return out
                        ops.conv2ch_dgrad(g, Sx, Ft, Win, ops.derived(w, "c2t", build_c2t), g_packed,
                                          Ft * ldp, ldp, lo * 2)
                    elif not train_disc:
                        _conv2d_dgrad(dy_t, Sx, Ft, Wout, C, w, sw, Win, g_packed, g_line=dy_line,
                                      g_seq=dy_seq, g_off=dy_off, x_line=ldp, x_off=lo * 2)
            blanes.join()
            if not train_disc:
                gfr = ops.empty(B * Ft, win, device=dev)
                if ops.fft_applies(win):
                    ops.stft_fft_adjoint(g_packed, win, Ft, gfr, interleaved=True)
                else:
                    gemm(mat(g_packed, B * Ft, 2 * nb), mat(dft_interleaved(win, dev)), gfr, form=1)
                gxn = ops.empty(B, T, device=dev)
                ops.frames_fold(gfr, gxn, B, Ft, win, hop, T, False)
                gx2 = ops.empty(B, T, device=dev)
                # x2 = [real; fake]: the generated half starts at row B
                ops.call("f2g_peaknorm_bwd", ops.ptr(gx2), ops.ptr(gxn),
                         ops.ptr(ctx.x2) + 4 * B * T, ops.ptr(st["stats"]) + 4 * 3 * B, B, T)
                lanes.chain_enter()  # g_fake is accumulated resolution after resolution
                ops.axpby_rows(g_fake, g_fake, gx2, sa=1.0, sb=1.0)
                lanes.chain_leave()
            if train_disc:
                with ops.weight_batch():
                    for slot_, gwp_, shape_ in unpack:
                        grads_w[slot_] = unpack_conv_grad(gwp_, shape_)
            pgrads += grads_w
        lanes.join()
        ctx.saved = None
        return tuple([None, g_fake, None, None] + pgrads)


def _colsum_strided(out, a, nrows, W, C, ld, off):
    """out[c] += sum over (row, w) of a[row*ld + off + w*C + c]: column sums of the (nrows, W*C)
    slice, then of the resulting (W, C) table."""
    tmp = ops.zeros(W * C, device=a.device)
    ops.call("f2g_colsum", ops.ptr(tmp), ops.ptr(a) + 4 * off, ld, None, 0, nrows, W * C)
    ops.call("f2g_colsum", ops.ptr(out), ops.ptr(tmp), C, None, 0, W, C)


# =====================================================================================
# Multi-scale log-mel reconstruction loss (gan.py:89-99)
# =====================================================================================
class MelReconLossFn(torch.autograd.Function):
    """sum_i mean |log clip(mel_i(real)) - log clip(mel_i(fake))|; gradient to `fake` only."""

    @staticmethod
    def forward(ctx, real, fake, specs):
        """specs: tuple of (n_fft, hop, fb (n_freq, n_mels))."""
        dev = real.device
        B, T = real.shape
        x2 = stack_pair(real, fake)
        loss = ops.zeros(1, device=dev)
        saved = []
        for n_fft, hop, fb in specs:
            dft_matrices(n_fft, dev)
        lanes = ops.Lanes(dev, len(specs), "mel")  # one launch lane per mel scale
        for i, (n_fft, hop, fb) in enumerate(specs):
          with lanes.lane(i):
            S, packed, spec, F = filterbank_spec(x2, n_fft, hop, fb, 1)
            n = B * F * fb.shape[1]
            ops.l1_loss_ab(loss, None, S, 0, S, n, 1, n, n, 1.0 / n, clip=1e-7)
            saved.append((S, packed, F))
        lanes.join()
        ctx.saved = saved
        ctx.specs = specs
        ctx.dims = (B, T)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        B, T = ctx.dims
        dev = g.device
        gw = g.reshape(1).contiguous()
        g_fake = ops.empty(B, T, device=dev)
        lanes = ops.Lanes(dev, len(ctx.specs), "mel")
        for i, (n_fft, hop, fb) in enumerate(ctx.specs):
          with lanes.lane(i):
            S, packed, F = ctx.saved[i]
            nm = fb.shape[1]
            n = B * F * nm
            gS = ops.empty(B * F, nm, device=dev)
            # gb follows b's layout: write straight into a fake-half-sized buffer via offsets
            ops.call("f2g_l1_loss", None, ops.ptr(gS), ops.ptr(S), ops.ptr(S) + 4 * n, 1, n, n,
                     1.0 / n, 1e-7, ops.ptr(gw))
            filterbank_spec_bwd(gS, packed[B * F:], n_fft, hop, fb, 1, B, T, F, g_fake, i > 0,
                                lanes)
        lanes.join()
        ctx.saved = None
        return None, g_fake, None
