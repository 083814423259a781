"""GPU parity of every C-ABI kernel against plain PyTorch fp32/fp64 references on CPU.

Tolerances: the GEMMs use exact-fp32 MFMA (k-ordered fmaf chains), so results differ from a
CPU fp32 reference only by summation order: relative 2e-5 of the result scale is the bar.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from flow2gan_amd import ops as o
    return o


DEV = "cuda"


def g(t):
    return t.to(DEV).contiguous()


def close(got, want, rtol=2e-5, name=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= rtol * scale + 1e-30, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# ------------------------------------------------------------------ GEMM forms
@pytest.mark.parametrize("R,K,N", [(300, 514, 96), (128, 32, 128), (1000, 48, 24), (77, 130, 130),
                                   (5, 512, 1536), (2050, 72, 40)])
def test_gemm_forward_plain(ops, R, K, N):
    A, W, b = rnd(R, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    out = torch.empty(R, N, device=DEV)
    ops.gemm(ops.mat(g(A)), ops.mat(g(W)), out, bias=g(b))
    close(out, A.double() @ W.double().t() + b.double(), name="fwd")


def test_gemm_forward_epilogues(ops):
    R, K, N = 257, 96, 160
    A, W, b = rnd(R, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    res, gam, alpha = rnd(R, N, seed=4), rnd(N, seed=5), rnd(K, seed=6) * 0.3
    out = torch.empty(R, N, device=DEV)
    ops.gemm(ops.mat(g(A), alpha=g(alpha)), ops.mat(g(W)), out, bias=g(b), res=g(res), gamma=g(gam))
    Ap = torch.where(A > 0, A, A * alpha[None])
    close(out, Ap.double() @ W.double().t() + b.double() + gam.double() * res.double(), name="res")
    out2 = torch.empty(R, N, device=DEV)
    ops.gemm(ops.mat(g(A)), ops.mat(g(W)), out2, bias=g(b), lrelu=0.1)
    close(out2, F.leaky_relu(A.double() @ W.double().t() + b.double(), 0.1), name="lrelu")
    # padded leading dimension on input and output
    Ap4 = torch.zeros(R, K + 4)
    Ap4[:, :K] = A
    out3 = torch.full((R, N + 4), 7.0, device=DEV)
    ops.gemm(ops.mat(g(Ap4), R, K), ops.mat(g(W)), out3)
    close(out3[:, :N], A.double() @ W.double().t(), name="ld")
    assert float(out3[:, N:].min()) == 7.0


def test_split_bf16_image_roundtrip(ops):
    """f2g_split_bf16: hi + lo reproduces x to ~2^-17 |x|, group layout [hi x4 | lo x4]."""
    x = rnd(37, 64, seed=9)
    img = ops.split_bf16(g(x)).cpu()
    u = img.view(torch.int16).reshape(-1, 8).to(torch.int32) & 0xFFFF
    hi = (u[:, :4] << 16).to(torch.int32).view(torch.float32).reshape(x.shape)
    lo = (u[:, 4:] << 16).to(torch.int32).view(torch.float32).reshape(x.shape)
    assert torch.equal(hi, x.bfloat16().float())
    assert float(((hi.double() + lo.double()) - x.double()).abs().max()) <= 2.0 ** -16 * float(x.abs().max())


@pytest.mark.parametrize("R,K,N", [(300, 512, 96), (1000, 64, 1536), (129, 1536, 512), (4100, 160, 130)])
def test_gemm_split_bf16_on_the_lean_kernel(ops, R, K, N):
    """precision 1 with pre-split operands (lean kernel, no conversion in the K loop): products
    hi*hi + hi*lo + lo*hi, fp32 accumulation -- error ~2^-16 per product; epilogues as in fp32."""
    A, W, b = rnd(R, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    res, gam, ps = rnd(R, N, seed=4), rnd(N, seed=5), rnd(N, seed=6) * 0.3
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision("bf16x3")
    try:
        lib = ops.L.lib
        out = torch.empty(R, N, device=DEV)
        ops.gemm(ops.mat(g(A)), ops.mat(g(W)), out, bias=g(b))
        assert lib.f2g_gemm_last_path() in (1, 2), "split-bf16 GEMM did not take the lean kernel"
        ref = A.double() @ W.double().t() + b.double()
        close(out, ref, rtol=3e-5, name="lean3")
        out2 = torch.empty(R, N, device=DEV)
        ops.gemm(ops.mat(g(A)), ops.mat(g(W)), out2, bias=g(b), res=g(res), gamma=g(gam), split_k=1)
        assert lib.f2g_gemm_last_path() == 1
        close(out2, ref + gam.double() * res.double(), rtol=3e-5, name="lean3-res")
        pre, act = torch.empty(R, N, device=DEV), torch.empty(R, N, device=DEV)
        ops.gemm(ops.mat(g(A)), ops.mat(g(W)), pre, bias=g(b), prelu=g(ps), prelu_out=act, split_k=1)
        close(pre, ref, rtol=3e-5, name="lean3-pre")
        close(act, torch.where(ref > 0, ref, ref * ps.double()[None]), rtol=3e-5, name="lean3-prelu")
        # data gradient against the cached transposed (and split) weight
        Wk = rnd(K, N, seed=7)
        Wk_d = torch.nn.Parameter(g(Wk))
        out3 = torch.empty(R, N, device=DEV)
        ops.gemm(ops.mat(g(A)), ops.mat(Wk_d), out3, form=1)
        if K % 32 == 0 and N > 64:
            assert lib.f2g_gemm_last_path() in (1, 2)
        close(out3, A.double() @ Wk.double(), rtol=3e-5, name="lean3-dgrad")
        # ... which follows an in-place update of the weight
        with torch.no_grad():
            Wk_d.mul_(2.0)
        ops.gemm(ops.mat(g(A)), ops.mat(Wk_d), out3, form=1)
        close(out3, 2.0 * (A.double() @ Wk.double()), rtol=3e-5, name="lean3-dgrad-updated")
    finally:
        ops.GEMM_PRECISION = was


def _lean_wgrad_expected(mode):
    # split-bf16: always the K-major kernel; exact fp32: only long reductions per block by default
    # (library option lean_wgrad = 2 forces it -- the kernel tests below run under that setting as well)
    from flow2gan_amd import _lib
    if mode == "bf16x6":      # gemm_leanw6_kernel (reported as the six-product family)
        return 4
    return 1 if (mode == "bf16x3" or _lib.get_option("lean_wgrad") == 2) else 0


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("R,M,N", [(3000, 128, 256), (777, 384, 128), (64, 256, 1152), (24064, 128, 128)])
def test_wgrad_k_major_lean_kernels(ops, R, M, N, mode, monkeypatch):
    """Weight gradient on the K-major lean kernels (exact fp32: single-float fragments; split-bf16:
    operands transposed by the LDS transpose read): g[m, n] += sum_r dY[r, m] X[r, n], atomic
    split-K onto an initialised output, partial last slab."""
    dY, X, g0 = rnd(R, M, seed=1), rnd(R, N, seed=2), rnd(M, N, seed=3)
    monkeypatch.setattr(ops, "X6_MIN_K", 32)      # (bf16x6: the profitability threshold off)
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision(mode)
    try:
        out = g(g0)
        ops.wgrad(g(dY), M, M, ops.mat(g(X)), out)
        assert ops.L.lib.f2g_gemm_last_path() == _lean_wgrad_expected(mode), "unexpected wgrad kernel family"
    finally:
        ops.GEMM_PRECISION = was
    want = g0.double() + dY.double().t() @ X.double()
    close(out, want, rtol=5e-5, name="leanw")
    if mode == "bf16x6":     # three pieces, six products: the error class of fp32 rounding
        mag = dY.abs().double().t() @ X.abs().double() + g0.abs().double()
        assert float(((out.cpu().double() - want).abs() / mag).max()) < 1e-6


@pytest.mark.parametrize("windowed", [False, True, "short"])
def test_wgrad_fp32_k_major_kernel_on_long_reductions(ops, windowed):
    """The exact-fp32 K-major lean kernel (single-float fragments through ds_read_b32 immediates) is
    chosen when every block walks >= 4096 rows: plain operands and MPD-style unbounded windows."""
    if ops.GEMM_PRECISION != 0:
        pytest.skip("a test of the exact-fp32 kernel (the suite runs under F2G_GEMM=" + str(ops.GEMM_PRECISION) + ")")
    if windowed:
        S, Hin, Cin, Cout, stv, HALO = 40, 610, 128, 128, 3, 2
        if windowed == "short":      # sequences shorter than a 32-row slab (MPD period 11 at 24 kHz: 31 rows)
            S, Hin = 700, 70
        Hout = (Hin + 4 - 5) // stv + 1
        Hp = Hout + 2 * HALO
        x = torch.zeros(S, Hin + 2 * HALO, Cin)
        x[:, HALO:HALO + Hin] = rnd(S, Hin, Cin, seed=1)
        gy = torch.zeros(S, Hp, Cout)
        gy[:, HALO:HALO + Hout] = rnd(S, Hout, Cout, seed=2)
        out = torch.zeros(Cout, 5 * Cin, device=DEV)
        X = ops.win1d(g(x), S, Hin + 2 * HALO, Cin, Hp, stv, HALO * stv, 5, unbounded=True)
        ops.gemm(ops.mat(g(gy).reshape(S * Hp, Cout)), X, out, form=2, atomic=True, split_k=2)
        assert ops.L.lib.f2g_gemm_last_path() == 1
        w = torch.zeros(Cout, Cin, 5, dtype=torch.float64, requires_grad=True)
        F.conv1d(x[:, HALO:HALO + Hin].permute(0, 2, 1).double(), w, None, stride=stv, padding=2).backward(
            gy[:, HALO:HALO + Hout].permute(0, 2, 1).double())
        close(out, w.grad.permute(0, 2, 1).reshape(Cout, 5 * Cin), rtol=2e-5, name="leanw fp32 window")
    else:
        R, M, N = 16411, 256, 128
        dY, X, g0 = rnd(R, M, seed=1), rnd(R, N, seed=2), rnd(M, N, seed=3)
        out = g(g0)
        ops.gemm(ops.mat(g(dY)), ops.mat(g(X)), out, form=2, atomic=True, split_k=3)
        assert ops.L.lib.f2g_gemm_last_path() == 1
        close(out, g0.double() + dY.double().t() @ X.double(), rtol=2e-5, name="leanw fp32")


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("S,Hin,stv", [(5, 50, 3), (7, 131, 1), (3, 400, 3), (40, 300, 1)])
def test_wgrad_unbounded_windows(ops, mode, S, Hin, stv, monkeypatch):
    """MPD-style weight gradient: X = (5,1)-tap windows over a halo layout, read past the sequence
    ends where the gradient map's halo rows are zero (f2g_operand.unbounded); slabs that straddle
    sequence ends, the first rows before the buffer, the last ones behind it.  bf16x6 with stride 1 runs on
    the tap-walking kernel of round 5 (gemm_leanw6t_kernel: all five taps from one staged window of map rows);
    the last case cuts its 12160 rows into several chunks."""
    Cin, Cout, HALO = 128, 256, 2
    Hout = (Hin + 4 - 5) // stv + 1
    Hp = Hout + 2 * HALO
    x = torch.zeros(S, Hin + 2 * HALO, Cin)
    x[:, HALO:HALO + Hin] = rnd(S, Hin, Cin, seed=1)
    gy = torch.zeros(S, Hp, Cout)
    gy[:, HALO:HALO + Hout] = rnd(S, Hout, Cout, seed=2)
    monkeypatch.setattr(ops, "X6_MIN_K", 32)
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision(mode)
    try:
        out = torch.zeros(Cout, 5 * Cin, device=DEV)
        X = ops.win1d(g(x), S, Hin + 2 * HALO, Cin, Hp, stv, HALO * stv, 5, unbounded=True)
        assert X.unbounded == 1
        ops.wgrad(g(gy).reshape(S * Hp, Cout), Cout, Cout, X, out)
        # (sequences shorter than a slab stay on the generic kernel)
        assert ops.L.lib.f2g_gemm_last_path() == (_lean_wgrad_expected(mode) if (Hp >= 16 or mode != "fp32") else 0)
    finally:
        ops.GEMM_PRECISION = was
    # conv1d(k=5, stride, pad=2) over the un-haloed input: output row o reads input rows o*stv-2 .. +2
    xin = x[:, HALO:HALO + Hin].permute(0, 2, 1).double().requires_grad_(False)
    w = torch.zeros(Cout, Cin, 5, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xin, w, None, stride=stv, padding=2)
    y.backward(gy[:, HALO:HALO + Hout].permute(0, 2, 1).double())
    close(out, w.grad.permute(0, 2, 1).reshape(Cout, 5 * Cin), rtol=5e-5, name="leanw3-window")


@pytest.mark.parametrize("R,K,N", [(300, 512, 96), (1000, 64, 1536), (129, 1536, 512)])
def test_gemm_true_bf16_operands_and_bf16_output(ops, R, K, N):
    """precision 2 over TRUE bf16 tensors (f2g_operand.split = 2: 64-element slabs): products of the
    bf16-rounded operands, fp32 accumulation; optional bf16 output written by the epilogue."""
    A, W, b, ps = rnd(R, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(N, seed=6) * 0.3
    ref = A.bfloat16().double() @ W.bfloat16().double().t() + b.double()
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision("bf16")
    try:
        out = torch.empty(R, N, device=DEV)
        ops.gemm(ops.mat(g(A)), ops.mat(g(W)), out, bias=g(b), split_k=1)   # converted here
        assert ops.L.lib.f2g_gemm_last_path() == 1
        close(out, ref, rtol=1e-5, name="bf16 lean")
        a16 = g(A).bfloat16()                                               # a producer's bf16 tensor
        out16 = torch.empty(R, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(ops.mat(a16, split=2), ops.mat(g(W)), out16, bias=g(b), prelu=g(ps), split_k=1)
        want = torch.where(ref > 0, ref, ref * ps.double()[None])
        close(out16.float(), want, rtol=5e-3, name="bf16 lean, bf16 out")
        assert float((out16.float().cpu().double() - want.float().bfloat16().double()).abs().max()) <= \
            2.0 ** -7 * float(want.abs().max())
    finally:
        ops.GEMM_PRECISION = was


@pytest.mark.parametrize("fmt", [1, 2])
def test_dwnorm_writes_gemm_operand_formats(ops, fmt):
    """dwnorm forward with z_format 1 / 2: the split-bf16 image / the bf16 tensor of exactly the
    values the fp32 kernel writes."""
    B, Fr, Cc, K = 2, 37, 256, 7
    x = rnd(B * Fr, Cc, seed=1)
    wdw, bdw, beta, ls = rnd(Cc, 1, K, seed=2) * 0.3, rnd(Cc, seed=3) * 0.1, rnd(Cc, seed=4) * 0.1, rnd(1, seed=5) * 0.1
    z = torch.empty(B * Fr, Cc, device=DEV)
    ops.dwnorm_fwd(g(x), z, B, Fr, Cc, K, None, g(wdw), g(bdw), g(beta), g(ls))
    if fmt == 1:
        zi = torch.empty(B * Fr, Cc, device=DEV)
        ops.dwnorm_fwd(g(x), zi, B, Fr, Cc, K, None, g(wdw), g(bdw), g(beta), g(ls), z_format=1)
        assert torch.equal(zi.view(torch.int32), ops.split_bf16(z).view(torch.int32))
    else:
        zb = torch.empty(B * Fr, Cc, device=DEV, dtype=torch.bfloat16)
        ops.dwnorm_fwd(g(x), zb, B, Fr, Cc, K, None, g(wdw), g(bdw), g(beta), g(ls), z_format=2)
        assert torch.equal(zb, z.bfloat16())


def test_gemm_split_bf16_windowed_operand(ops):
    """MPD-style (5,1) conv over a halo layout in split-bf16: the window addressing of the lean
    kernel is unchanged by the split image."""
    S, H, Cin, Cout = 6, 96, 32, 128          # stride 1, taps 5: rows = S * (H - 4)
    x, w = rnd(S, H, Cin, seed=1), rnd(Cout, 5 * Cin, seed=2)
    Hout = H - 4
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision("bf16x3")
    try:
        out = torch.empty(S * Hout, Cout, device=DEV)
        ops.gemm(ops.win1d(g(x), S, H, Cin, Hout, 1, 0, 5), ops.mat(g(w)), out, split_k=1)
        assert ops.L.lib.f2g_gemm_last_path() == 1
    finally:
        ops.GEMM_PRECISION = was
    cols = torch.stack([x[:, i:i + Hout] for i in range(5)], 2).reshape(S * Hout, 5 * Cin)
    close(out, cols.double() @ w.double().t(), rtol=3e-5, name="lean3-window")


@pytest.mark.parametrize("R,K,N", [(300, 96, 514), (64, 1536, 512), (1000, 24, 72)])
def test_gemm_dgrad_with_prelu_grad(ops, R, K, N):
    G, W = rnd(R, K, seed=1), rnd(K, N, seed=2)
    aux, alpha = rnd(R, N, seed=3), rnd(N, seed=4) * 0.3
    out = torch.empty(R, N, device=DEV)
    cs_a = torch.zeros(N, device=DEV)
    cs = torch.zeros(N, device=DEV)
    ops.gemm(ops.mat(g(G)), ops.mat(g(W)), out, form=1, aux=g(aux), alpha_n=g(alpha),
             colsum_alpha=cs_a, colsum=cs)
    dp = G.double() @ W.double()
    want = dp * torch.where(aux > 0, torch.ones_like(aux), alpha[None].expand_as(aux)).double()
    close(out, want, name="dgrad")
    close(cs, want.sum(0), rtol=1e-4, name="colsum")
    close(cs_a, (dp * aux.clamp(max=0).double()).sum(0), rtol=1e-4, name="colsum_alpha")
    # in place over aux
    auxd = g(aux)
    ops.gemm(ops.mat(g(G)), ops.mat(g(W)), auxd, form=1, aux=auxd, alpha_n=g(alpha))
    close(auxd, want, name="dgrad-inplace")


@pytest.mark.parametrize("R,M,N", [(3000, 96, 40), (700, 514, 48), (5000, 32, 864), (130, 1536, 512)])
def test_gemm_wgrad_splitk(ops, R, M, N):
    dY, X = rnd(R, M, seed=1), rnd(R, N, seed=2)
    alpha = rnd(N, seed=3) * 0.3
    out = torch.zeros(M, N, device=DEV)
    ops.wgrad(g(dY), M, M, ops.mat(g(X), alpha=g(alpha)), out)
    Xp = torch.where(X > 0, X, X * alpha[None])
    close(out, dY.double().t() @ Xp.double(), rtol=1e-4, name="wgrad")


# ------------------------------------------------------------------ windowed operands
@pytest.mark.parametrize("n_fft,hop,T", [(512, 256, 6000), (128, 64, 4097), (1024, 256, 6000),
                                         (32, 8, 1000), (2048, 512, 6000)])
def test_stft_as_windowed_gemm(ops, n_fft, hop, T):
    from flow2gan_amd.fused import stft_packed
    x = rnd(3, T, seed=5, scale=0.1)
    packed, Fr = stft_packed(g(x), n_fft, hop)
    spec = torch.stft(x.double(), n_fft, hop, n_fft, torch.hann_window(n_fft).double(), center=True,
                      return_complex=True)  # (B, nb, F)
    assert Fr == spec.shape[2] == 1 + T // hop  # frame indexing is exact
    nb = n_fft // 2 + 1
    want = torch.cat([spec.real, spec.imag], 1).permute(0, 2, 1).reshape(3 * Fr, 2 * nb)
    close(packed[:, :2 * nb], want, rtol=3e-5, name="stft")


@pytest.mark.parametrize("n_fft,hop,T,inter", [(1024, 256, 6000, False), (2048, 512, 24000, True),
                                               (4096, 1024, 9000, False), (1024, 256, 5003, True),
                                               (512, 256, 6000, False), (256, 128, 5003, False),
                                               (128, 64, 6001, False), (512, 128, 4000, True)])
def test_stft_lds_fft_forward_and_adjoint(ops, n_fft, hop, T, inter):
    """LDS-butterfly FFT (fft.hip) against torch.stft in float64 (modules.py:69-78): spectrum in the
    planar and the interleaved row layout, and its adjoint against autograd."""
    B = 3
    assert ops.fft_applies(n_fft)
    x = rnd(B, T, seed=11, scale=0.1)
    Fr = 1 + T // hop
    nb = n_fft // 2 + 1
    ld = ops.pad4(2 * nb)
    spec = torch.full((B * Fr, ld), 7.0, device=DEV)
    ops.stft_fft(g(x), n_fft, hop, Fr, spec, interleaved=inter)
    xd = x.double().requires_grad_(True)
    ref = torch.stft(xd, n_fft, hop, n_fft, torch.hann_window(n_fft).double(), center=True,
                     return_complex=True)
    assert ref.shape[2] == Fr
    if inter:
        want = torch.stack([ref.real, ref.imag], -1).permute(0, 2, 1, 3).reshape(B * Fr, 2 * nb)
    else:
        want = torch.cat([ref.real, ref.imag], 1).permute(0, 2, 1).reshape(B * Fr, 2 * nb)
    close(spec[:, :2 * nb], want.detach(), rtol=5e-6, name="fft-stft")
    # (default: the kernel applies the center / reflect padding itself; the padded-copy path -- a
    # f2g_reflect_pad launch in front -- gives the same spectrum bit for bit, and a non-contiguous
    # batch of signals goes through it)
    assert ops.FFT_REFLECT
    spec_p = torch.full((B * Fr, ld), 7.0, device=DEV)
    ops.FFT_REFLECT = False
    try:
        ops.stft_fft(g(x), n_fft, hop, Fr, spec_p, interleaved=inter)
    finally:
        ops.FFT_REFLECT = True
    assert torch.equal(spec_p[:, :2 * nb], spec[:, :2 * nb])
    xw = torch.zeros(B, T + 5, device=DEV)
    xw[:, 2:T + 2] = g(x)
    spec_v = torch.full((B * Fr, ld), 7.0, device=DEV)
    ops.stft_fft(xw[:, 2:T + 2], n_fft, hop, Fr, spec_v, interleaved=inter)      # rows T + 5 apart
    assert torch.equal(spec_v[:, :2 * nb], spec[:, :2 * nb])
    # adjoint: d<gs, stft(x)>/dx through frames_fold
    gs = rnd(B * Fr, ld, seed=12)
    gfr = torch.empty(B * Fr, n_fft, device=DEV)
    ops.stft_fft_adjoint(g(gs), n_fft, Fr, gfr, interleaved=inter)
    gx = torch.empty(B, T, device=DEV)
    ops.frames_fold(gfr, gx, B, Fr, n_fft, hop, T, False)
    (want * gs[:, :2 * nb].double()).sum().backward()
    close(gx, xd.grad, rtol=5e-6, name="fft-stft-adjoint")


@pytest.mark.parametrize("n_fft,hop,T", [(512, 256, 6000), (256, 128, 5003), (128, 64, 6001), (1024, 256, 24000)])
def test_istft_lds_fft_against_torch_istft(ops, n_fft, hop, T):
    """The windowed inverse real transform of the iSTFT through the LDS FFT (f2g_fft_frames mode 2)
    + the overlap-add kernel against torch.istft in float64 (modules.py:106-115), with junk in Im(DC) /
    Im(Nyquist) (torch ignores them) and ragged row counts (frames not a multiple of 4); and its
    adjoint (mode 3) against autograd through torch.istft."""
    B = 3
    assert ops.fft_applies(n_fft)
    Fr = 1 + T // hop
    nb = n_fft // 2 + 1
    ld = (2 * nb + 63) // 64 * 64
    spec = rnd(B * Fr, ld, seed=21, scale=0.3)
    win = torch.hann_window(n_fft)
    frames = torch.full((B * Fr, n_fft), 7.0, device=DEV)
    ops.istft_fft(g(spec), n_fft, Fr, frames)
    out = torch.empty(B, T, device=DEV)
    ops.istft_ola(frames, out, B, Fr, n_fft, hop, T, g(win), None, 1.0, False)
    sr = spec[:, :nb].reshape(B, Fr, nb).permute(0, 2, 1).double().requires_grad_(True)
    si = spec[:, nb:2 * nb].reshape(B, Fr, nb).permute(0, 2, 1).double().requires_grad_(True)
    y = torch.istft(torch.complex(sr, si), n_fft, hop, n_fft, win.double(), center=True)
    Ty = y.shape[1]
    n = min(T, Ty)
    close(out[:, :n], y[:, :n].detach(), rtol=2e-5, name="ifft + ola")
    if T > Ty:
        assert float(out[:, Ty:].abs().max()) == 0.0       # convert_length pads with zeros
    # adjoint of the inverse transform alone: <gf, frames(spec)> differentiated w.r.t. the bins
    gf = rnd(B * Fr, n_fft, seed=22)
    gsp = torch.zeros(B * Fr, ld, device=DEV)
    ops.istft_fft_adjoint(g(gf), n_fft, Fr, gsp)
    k = torch.arange(nb).double()
    nn = torch.arange(n_fft).double()
    ang = 2 * torch.pi * k[:, None] * nn[None] / n_fft
    c = torch.full((nb,), 2.0, dtype=torch.double)
    c[0] = c[-1] = 1.0
    gw = gf.double() * win.double()[None]
    want_r = (gw @ torch.cos(ang).t()) * c / n_fft
    want_i = -(gw @ torch.sin(ang).t()) * c / n_fft
    want_i[:, 0] = 0.0
    want_i[:, -1] = 0.0
    close(gsp[:, :nb], want_r, rtol=2e-5, name="ifft adjoint re")
    close(gsp[:, nb:2 * nb], want_i, rtol=2e-5, name="ifft adjoint im")
    assert float(gsp[:, 2 * nb:].abs().max()) == 0.0


def test_stft_fft_agrees_with_dft_gemm(ops):
    """Both STFT paths of stft_packed give the same spectra (frame indexing identical)."""
    from flow2gan_amd.fused import stft_packed
    x = g(rnd(2, 7000, seed=3, scale=0.1))
    a, Fa = stft_packed(x, 1024, 256)
    old = ops.USE_FFT
    ops.USE_FFT = False
    try:
        b, Fb = stft_packed(x, 1024, 256)
    finally:
        ops.USE_FFT = old
    assert Fa == Fb and a.shape == b.shape
    close(a[:, :1026], b[:, :1026].double(), rtol=3e-5, name="fft-vs-gemm")


def test_conv1d_k3_windowed(ops):
    B, Cin, Fm, Cout = 3, 100, 37, 64
    x, w, b = rnd(B, Cin, Fm, seed=1), rnd(Cout, Cin, 3, seed=2), rnd(Cout, seed=3)
    rows = torch.empty(B * Fm, Cin, device=DEV)
    ops.bct_to_rows(rows, g(x), B, Cin, Fm)
    close(rows, x.permute(0, 2, 1).reshape(B * Fm, Cin), name="bct_to_rows")
    wp = torch.empty(Cout, 3 * Cin, device=DEV)
    ops.permute4(wp, g(w), (Cout, 3, Cin, 1), (Cin * 3, 1, 3, 0))
    out = torch.empty(B * Fm, Cout, device=DEV)
    ops.gemm(ops.win1d(rows, B, Fm, Cin, Fm, 1, 1, 3), ops.mat(wp), out, bias=g(b))
    want = F.conv1d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 1).reshape(B * Fm, Cout)
    close(out, want, name="conv1d")
    back = torch.empty(B, Cout, Fm, device=DEV)
    ops.rows_to_bct(back, out, B, Cout, Fm)
    close(back, F.conv1d(x.double(), w.double(), b.double(), padding=1), name="rows_to_bct")


@pytest.mark.parametrize("Cin,Cout,H,stride", [(1, 32, 100, 3), (32, 128, 67, 3), (16, 8, 40, 1)])
def test_conv_period_stride3(ops, Cin, Cout, H, stride):
    """MPD (5,1) conv, stride (3,1): sequences = (batch, column) pairs."""
    S = 6
    Hout = (H + 4 - 5) // stride + 1
    x, w, b = rnd(S, H, Cin, seed=1), rnd(Cout, Cin, 5, seed=2), rnd(Cout, seed=3)
    wp = torch.empty(Cout, 5 * Cin, device=DEV)
    ops.permute4(wp, g(w), (Cout, 5, Cin, 1), (Cin * 5, 1, 5, 0))
    out = torch.empty(S * Hout, Cout, device=DEV)
    ops.gemm(ops.win1d(g(x), S, H, Cin, Hout, stride, 2, 5), ops.mat(wp), out, bias=g(b), lrelu=0.1)
    want = F.leaky_relu(F.conv1d(x.permute(0, 2, 1).double(), w.double(), b.double(), stride=stride,
                                 padding=2), 0.1).permute(0, 2, 1).reshape(S * Hout, Cout)
    close(out, want, name="mpd-conv")


@pytest.mark.parametrize("S,H", [(6, 100), (3, 4001), (130, 37)])
def test_mpd_first_layer_direct_kernels(ops, S, H):
    """mpd0.hip: Conv2d(1, 32, (5,1), stride (3,1), padding (2,0)) + leaky ReLU into the halo layout,
    its weight gradient and its data gradient, against torch conv1d + autograd in float64."""
    HALO = 2
    Hout = (H + 4 - 5) // 3 + 1
    x = rnd(S, H, seed=1)
    w, b = rnd(32, 5, seed=2) * 0.3, rnd(32, seed=3) * 0.1
    y = torch.full((S * (Hout + 2 * HALO), 32), 7.0, device=DEV)
    ops.mpd0_fwd(g(x), S, H, Hout, HALO, g(w), g(b), 0.1, y)
    xd = x.double()[:, None, :].requires_grad_(True)
    wd = w.double()[:, None, :].requires_grad_(True)
    pre = F.conv1d(xd, wd, b.double(), stride=3, padding=2)
    ref = F.leaky_relu(pre, 0.1)
    yv = y.view(S, Hout + 2 * HALO, 32)
    close(yv[:, HALO:HALO + Hout], ref.permute(0, 2, 1), name="mpd0 fwd")
    assert float(yv[:, :HALO].min()) == 7.0 and float(yv[:, HALO + Hout:].max()) == 7.0
    gy = torch.zeros(S, Hout + 2 * HALO, 32)
    gy[:, HALO:HALO + Hout] = rnd(S, Hout, 32, seed=4)
    pre.backward(gy[:, HALO:HALO + Hout].permute(0, 2, 1).double())
    gw = torch.zeros(32, 5, device=DEV)
    ops.mpd0_wgrad(g(x), S, H, Hout, HALO, g(gy), gw)
    close(gw, wd.grad[:, 0, :], rtol=1e-4, name="mpd0 wgrad")
    gx = torch.full((S * H, 1), 7.0, device=DEV)
    ops.mpd0_dgrad(g(gy), S, H, Hout, HALO, g(w), gx)
    close(gx.view(S, H), xd.grad[:, 0, :], name="mpd0 dgrad")


@pytest.mark.parametrize("S,H", [(5, 27), (3, 100), (40, 9)])
def test_mpd_last_layer_direct_kernels(ops, S, H):
    """mpd0.hip: conv_post of a period discriminator (Conv2d(1024, 1, (3,1), padding (1,0))) over the
    halo layout: scores, weight gradient and data gradient against torch conv1d + autograd (fp64)."""
    HALO, Cc = 2, 1024
    y = torch.zeros(S, H + 2 * HALO, Cc)
    y[:, HALO:HALO + H] = rnd(S, H, Cc, seed=1)
    w, b = rnd(1, Cc, 3, seed=2) * 0.05, rnd(1, seed=3)
    w3 = w[0].t().contiguous()                                    # [tap][c]
    out = torch.full((S * H, 1), 7.0, device=DEV)
    ops.mpdpost_fwd(g(y), S, H, HALO, g(w3), g(b), out)
    yd = y[:, HALO:HALO + H].permute(0, 2, 1).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = F.conv1d(yd, wd, b.double(), padding=1)
    close(out.view(S, H), ref[:, 0], name="mpdpost fwd")
    gs = rnd(S, H, seed=4)
    ref.backward(gs[:, None, :].double())
    gw = torch.zeros(3, Cc, device=DEV)
    ops.mpdpost_wgrad(g(y), S, H, HALO, g(gs), gw)
    close(gw, wd.grad[0].t(), rtol=1e-4, name="mpdpost wgrad")
    gy = torch.zeros(S * (H + 2 * HALO), Cc, device=DEV)
    ops.mpdpost_dgrad(g(gs), S, H, HALO, g(w3), gy)
    gv = gy.view(S, H + 2 * HALO, Cc)
    close(gv[:, HALO:HALO + H], yd.grad.permute(0, 2, 1), name="mpdpost dgrad")
    assert float(gv[:, :HALO].abs().max()) == 0.0 and float(gv[:, HALO + H:].abs().max()) == 0.0
    # round 5: the same with the leaky-ReLU backward of the layer it lands on (+ the feature-matching term
    # against the other half's map), the bias-gradient column sums and the result's three-piece image
    from flow2gan_amd import fused_disc as fd
    ref_map = torch.zeros(S, H + 2 * HALO, Cc)
    ref_map[:, HALO:HALO + H] = rnd(S, H, Cc, seed=6)
    wdev = torch.tensor([0.7], device=DEV)
    was = ops.GEMM_PRECISION
    try:
        ops.set_gemm_precision("bf16x6")
        gm = fd._halo_rows(S, H, Cc, DEV, x3=True)
        cs = torch.zeros(Cc, device=DEV)
        ops.mpdpost_dgrad(g(gs), S, H, HALO, g(w3), gm, mask=(g(y), 0, 0.1), fm=(g(ref_map), 0, 0.3, wdev), colsum=cs)
    finally:
        ops.GEMM_PRECISION = was
    plain = yd.grad.permute(0, 2, 1)
    yv, rv = y[:, HALO:HALO + H].double(), ref_map[:, HALO:HALO + H].double()
    want = (plain + 0.3 * 0.7 * torch.sign(yv - rv)) * torch.where(yv > 0, 1.0, 0.1)
    gmv = gm.view(S, H + 2 * HALO, Cc)
    close(gmv[:, HALO:HALO + H], want, name="mpdpost fused dgrad")
    assert float(gmv[:, :HALO].abs().max()) == 0.0 and float(gmv[:, HALO + H:].abs().max()) == 0.0
    close(cs, want.sum((0, 1)), rtol=1e-4, name="mpdpost fused column sums")
    img = getattr(gm, "_f2g_x3", None)
    assert img is not None and torch.equal(img.view(torch.int16), ops.x3_flat_image(gm).view(torch.int16))


@pytest.mark.parametrize("Cin,Cout,kw,sw", [(2, 32, 9, 1), (32, 32, 9, 2), (32, 32, 3, 1), (32, 1, 3, 1)])
def test_conv2d_band_windowed(ops, Cin, Cout, kw, sw):
    """MRD (3,kw) conv over (time, freq) with stride (1,sw) on a frequency slice of a wider image."""
    B, H, Wtot, lo, hi = 2, 11, 50, 7, 41
    W = hi - lo
    pw = kw // 2
    Wout = (W + 2 * pw - kw) // sw + 1
    img = rnd(B, H, Wtot, Cin, seed=1)
    w, b = rnd(Cout, Cin, 3, kw, seed=2), rnd(Cout, seed=3)
    wp = torch.empty(Cout, 3 * kw * Cin, device=DEV)
    ops.permute4(wp, g(w), (Cout, 3 * kw, Cin, 1), (Cin * 3 * kw, 1, 3 * kw, 0))
    out = torch.empty(B * H * Wout, Cout, device=DEV)
    A = ops.win2d(g(img), B, H, W, Cin, Wout, 3, kw, sw, 1, pw, line_stride=Wtot * Cin,
                  seq_stride=H * Wtot * Cin, offset=lo * Cin)
    ops.gemm(A, ops.mat(wp), out, bias=g(b))
    xin = img[:, :, lo:hi].permute(0, 3, 1, 2).double()
    want = F.conv2d(xin, w.double(), b.double(), stride=(1, sw), padding=(1, pw))
    close(out, want.permute(0, 2, 3, 1).reshape(B * H * Wout, Cout), name="mrd-conv")


# ------------------------------------------------------------------ ConvNeXt block kernels
def _block_ref(x, lens, w_dw, b_dw, beta, ls, cproj, up, te):
    """x (B,C,F) -> z (B,C,F) in fp64: dwconv(x*mask) -> BiasNorm -> +cond -> *(1+te)."""
    B, C, Fr = x.shape
    mask = (torch.arange(Fr)[None] < lens[:, None]).double()[:, None]
    u = F.conv1d(x * mask, w_dw, b_dw, padding=w_dw.shape[-1] // 2, groups=C)
    s = ((u - beta[None, :, None]) ** 2).mean(1, keepdim=True) ** -0.5 * ls.exp()
    v = u * s
    if cproj is not None:
        cu = torch.repeat_interleave(cproj, up, dim=2)[:, :, :Fr]
        v = v + cu
    if te is not None:
        v = v * (1 + te[:, :, None])
    return v


@pytest.mark.parametrize("C,Fr,up,K", [(48, 47, 2, 7), (24, 94, 4, 7), (768, 24, 1, 7), (32, 33, 1, 5)])
def test_dwnorm_fwd_bwd(ops, C, Fr, up, K):
    B = 3
    Fc = (Fr + up - 1) // up
    x = rnd(B, C, Fr, seed=1).double().requires_grad_()
    lens = torch.tensor([Fr, Fr - 5, max(1, Fr // 2)])
    w_dw = (rnd(C, 1, K, seed=2) * 0.3).double().requires_grad_()
    b_dw = (rnd(C, seed=3) * 0.1).double().requires_grad_()
    beta = (rnd(C, seed=4) * 0.1).double().requires_grad_()
    ls = torch.tensor(0.7, dtype=torch.double, requires_grad=True)
    cproj = rnd(B, C, Fc, seed=5).double().requires_grad_()
    te = (rnd(B, C, seed=6) * 0.3).double().requires_grad_()
    z = _block_ref(x, lens, w_dw, b_dw, beta, ls, cproj, up, te)
    gz = rnd(B, C, Fr, seed=7).double()
    z.backward(gz)

    def rows(t):  # (B,C,F) -> (B*F, C) fp32 on device
        return g(t.detach().float().permute(0, 2, 1).reshape(-1, t.shape[1]))

    xr, lens_d = rows(x), g(lens.int())
    NC = 2 * C  # place this block's slice in the middle of a wider stacked buffer
    cp_all = torch.zeros(B * Fc, NC, device=DEV)
    cp_all[:, C // 2: C // 2 + C] = rows(cproj)
    te_all = torch.zeros(B, NC, device=DEV)
    te_all[:, C // 2: C // 2 + C] = g(te.detach().float())
    zr = torch.empty(B * Fr, C, device=DEV)
    args = (B, Fr, C, K, lens_d, g(w_dw.detach().float()), g(b_dw.detach().float()),
            g(beta.detach().float()), g(ls.detach().float().reshape(1)))
    ops.dwnorm_fwd(xr, zr, *args, cp_all, NC, Fc, up, C // 2, te_all, NC, C // 2)
    close(zr, z.detach().permute(0, 2, 1).reshape(-1, C), name="dwnorm_fwd")

    du = torch.empty(B * Fr, C, device=DEV)
    g_cp = torch.zeros(B * Fc, NC, device=DEV)
    g_te = torch.zeros(B, NC, device=DEV)
    g_beta = torch.zeros(C, device=DEV)
    g_ls = torch.zeros(1, device=DEV)
    ops.dwnorm_bwd(xr, rows(gz), du, *args, cp_all, NC, Fc, up, C // 2, te_all, NC, C // 2,
                   g_cproj=g_cp, g_te=g_te, g_beta=g_beta, g_log_scale=g_ls)
    close(g_cp[:, C // 2: C // 2 + C], cproj.grad.permute(0, 2, 1).reshape(-1, C), rtol=1e-4, name="g_cproj")
    assert float(g_cp[:, :C // 2].abs().max()) == 0.0
    # store mode (the block owns its zero-filled columns): same sums without the read-modify-write;
    # a second accumulate-mode call doubles, a second store-mode call does not
    g_cp2 = torch.zeros(B * Fc, NC, device=DEV)
    junk = [torch.zeros(B, NC, device=DEV), torch.zeros(C, device=DEV), torch.zeros(1, device=DEV)]
    for _ in range(2):
        ops.dwnorm_bwd(xr, rows(gz), torch.empty_like(du), *args, cp_all, NC, Fc, up, C // 2, te_all, NC,
                       C // 2, g_cproj=g_cp2, g_te=junk[0], g_beta=junk[1], g_log_scale=junk[2],
                       g_cproj_store=True)
    close(g_cp2, g_cp, rtol=1e-6, name="g_cproj store mode")
    close(g_te[:, C // 2: C // 2 + C], te.grad, rtol=1e-4, name="g_te")
    close(g_beta, beta.grad, rtol=2e-4, name="g_beta")
    close(g_ls, ls.grad.reshape(1), rtol=2e-4, name="g_log_scale")
    gx = torch.empty(B * Fr, C, device=DEV)
    gres = rnd(B * Fr, C, seed=8)
    gam = rnd(C, seed=9)
    g_w = torch.zeros(C, 1, K, device=DEV)
    g_b = torch.zeros(C, device=DEV)
    g_gam = torch.zeros(C, device=DEV)
    ops.dwconv_bwd(du, xr, gx, B, Fr, C, K, lens_d, args[5], gres=g(gres), gamma=g(gam), g_w=g_w,
                   g_b=g_b, g_gamma=g_gam)
    want_gx = x.grad.permute(0, 2, 1).reshape(-1, C) + gam.double()[None] * gres.double()
    close(gx, want_gx, rtol=1e-4, name="dwconv gx")
    close(g_w, w_dw.grad, rtol=2e-4, name="g_w_dw")
    close(g_b, b_dw.grad, rtol=2e-4, name="g_b_dw")
    close(g_gam, (gres.double() * x.detach().permute(0, 2, 1).reshape(-1, C)).sum(0), rtol=2e-4,
          name="g_gamma")


@pytest.mark.parametrize("rows,C", [(100, 48), (33, 768), (257, 24)])
def test_biasnorm(ops, rows, C):
    x = rnd(rows, C, seed=1).double().requires_grad_()
    beta = (rnd(C, seed=2) * 0.1).double().requires_grad_()
    ls = torch.tensor(1.2, dtype=torch.double, requires_grad=True)
    y = x * (((x - beta) ** 2).mean(1, keepdim=True) ** -0.5 * ls.exp())
    gy = rnd(rows, C, seed=3).double()
    y.backward(gy)
    xd, bd, ld = g(x.detach().float()), g(beta.detach().float()), g(ls.detach().float().reshape(1))
    yd = torch.empty(rows, C, device=DEV)
    ops.biasnorm_fwd(xd, yd, rows, C, bd, ld)
    close(yd, y, name="biasnorm fwd")
    gx = torch.empty(rows, C, device=DEV)
    gb, gl = torch.zeros(C, device=DEV), torch.zeros(1, device=DEV)
    ops.biasnorm_bwd(xd, g(gy.float()), gx, rows, C, bd, ld, gb, gl)
    close(gx, x.grad, rtol=1e-4, name="biasnorm gx")
    close(gb, beta.grad, rtol=2e-4, name="biasnorm gbeta")
    close(gl, ls.grad.reshape(1), rtol=2e-4, name="biasnorm gls")


# ------------------------------------------------------------------ iSTFT / folds
@pytest.mark.parametrize("n_fft,hop,T", [(512, 256, 6000), (256, 128, 6000), (128, 64, 6000),
                                         (128, 64, 1024)])
def test_istft_gemm_ola_fwd_bwd(ops, n_fft, hop, T):
    from flow2gan_amd.models.modules import dft_matrices
    B = 2
    Fr = 1 + T // hop
    nb = n_fft // 2 + 1
    packed = rnd(B, 2 * nb, Fr, seed=1).double().requires_grad_()
    win = torch.hann_window(n_fft).double()
    spec = torch.complex(packed[:, :nb], packed[:, nb:])
    y = torch.istft(spec, n_fft, hop, n_fft, win, center=True)
    y = F.pad(y, (0, T - y.shape[-1])) if y.shape[-1] < T else y[..., :T]
    gy = rnd(B, T, seed=2).double()
    y.backward(gy)
    wb = torch.tensor([0.5, 1.5])
    _, Wi = dft_matrices(n_fft, DEV)
    pr = torch.zeros(B * Fr, 2 * nb + 2, device=DEV)
    pr[:, :2 * nb] = g(packed.detach().float().permute(0, 2, 1).reshape(B * Fr, 2 * nb))
    frames = torch.empty(B * Fr, n_fft, device=DEV)
    ops.gemm(ops.mat(pr, B * Fr, 2 * nb), ops.mat(Wi), frames)
    out = torch.full((B, T), 1.0, device=DEV)
    ops.istft_ola(frames, out, B, Fr, n_fft, hop, T, g(win.float()), g(wb), 1.0 / 3, True)
    close(out, 1.0 + y.detach() * wb.double()[:, None] / 3, name="istft")
    # f2g_istft_ola_multi: three "branches" (the same frames with other weights / no weights) in one
    # launch = three launches one after the other, bit for bit
    wb2 = g(torch.tensor([2.0, -1.0]))
    seq = torch.full((B, T), 0.25, device=DEV)
    ops.istft_ola(frames, seq, B, Fr, n_fft, hop, T, g(win.float()), g(wb), 1.0 / 3, True)
    ops.istft_ola(frames, seq, B, Fr, n_fft, hop, T, g(win.float()), None, 1.0 / 3, True)
    ops.istft_ola(frames, seq, B, Fr, n_fft, hop, T, g(win.float()), wb2, 1.0 / 3, True)
    one = torch.full((B, T), 0.25, device=DEV)
    ent = [(frames, Fr, n_fft, hop, g(win.float()), w_) for w_ in (g(wb), None, wb2)]
    ops.istft_ola_multi(ent, one, B, T, 1.0 / 3, accumulate=True)
    assert torch.equal(one, seq)
    seq2 = torch.full((B, T), float("nan"), device=DEV)
    ops.istft_ola(frames, seq2, B, Fr, n_fft, hop, T, g(win.float()), g(wb), 1.0 / 3, False)
    ops.istft_ola(frames, seq2, B, Fr, n_fft, hop, T, g(win.float()), wb2, 1.0 / 3, True)
    one2 = torch.full((B, T), float("nan"), device=DEV)
    ops.istft_ola_multi([ent[0], ent[2]], one2, B, T, 1.0 / 3)
    assert torch.equal(one2, seq2)
    gfr = torch.empty(B * Fr, n_fft, device=DEV)
    ops.istft_ola_bwd(g(gy.float()), gfr, B, Fr, n_fft, hop, T, g(win.float()), g(wb), 1.0 / 3)
    gp = torch.empty(B * Fr, 2 * nb + 2, device=DEV)
    ops.gemm(ops.mat(gfr), ops.mat(Wi), gp, form=1)
    want = (packed.grad * wb.double()[:, None, None] / 3).permute(0, 2, 1).reshape(B * Fr, 2 * nb)
    close(gp[:, :2 * nb], want, rtol=1e-4, name="istft bwd")


@pytest.mark.parametrize("n_fft,hop,T", [(512, 256, 6000), (128, 32, 999), (1024, 256, 6000)])
def test_stft_backward_fold(ops, n_fft, hop, T):
    from flow2gan_amd.models.modules import dft_matrices
    B = 2
    x = rnd(B, T, seed=1).double().requires_grad_()
    win = torch.hann_window(n_fft).double()
    spec = torch.stft(x, n_fft, hop, n_fft, win, center=True, return_complex=True)
    nb, Fr = spec.shape[1], spec.shape[2]
    packed = torch.cat([spec.real, spec.imag], 1)
    gp = rnd(B, 2 * nb, Fr, seed=2).double()
    packed.backward(gp)
    Wd, _ = dft_matrices(n_fft, DEV)
    gpr = g(gp.float().permute(0, 2, 1).reshape(B * Fr, 2 * nb))
    gfr = torch.empty(B * Fr, n_fft, device=DEV)
    ops.gemm(ops.mat(gpr), ops.mat(Wd), gfr, form=1)
    gx = torch.full((B, T), 2.0, device=DEV)
    ops.frames_fold(gfr, gx, B, Fr, n_fft, hop, T, True)
    close(gx, 2.0 + x.grad, rtol=1e-4, name="stft bwd")


# ------------------------------------------------------------------ elementwise / losses
def test_elementwise_small_kernels(ops):
    x0, x1 = rnd(4, 1000, seed=1), rnd(4, 1000, seed=2)
    ca, cb = rnd(4, seed=3), rnd(4, seed=4)
    y = torch.empty(4, 1000, device=DEV)
    ops.axpby_rows(y, g(x0), g(x1), ca=g(ca), cb=g(cb))
    close(y, ca[:, None] * x0 + cb[:, None] * x1, name="axpby")
    ops.axpby_rows(y, g(x0), g(x1), sa=0.25, sb=-2.0)
    close(y, 0.25 * x0 - 2.0 * x1, name="axpby scalar")
    ops.clamp(y, g(x0), -0.5, 0.5)
    close(y, x0.clamp(-0.5, 0.5), name="clamp")
    s = torch.empty(4, 1000, device=DEV)
    ops.silu(s, g(x0))
    close(s, F.silu(x0.double()), name="silu")
    xx = x0.double().requires_grad_()
    F.silu(xx).backward(x1.double())
    ops.silu_bwd(s, g(x1), g(x0))
    close(s, xx.grad, name="silu bwd")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import flow2gan_oracle as O
    t = torch.tensor([0.0, 0.25, 0.5, 0.999])
    emb = torch.empty(4, 512, device=DEV)
    ops.time_embedding(emb, g(t), 512)
    close(emb, O.sinusoid_embedding(t, 512), rtol=2e-4, name="time embedding")
    cs = torch.zeros(1000, device=DEV)
    ops.colsum(cs, g(x0), 4, 1000, b=g(x1))
    close(cs, (x0.double() * x1.double()).sum(0), name="colsum")
    p, gr = torch.tensor([0.3, 0.7, 1.2, 0.4, 1.1]), torch.tensor([1.0, 1.0, -1.0, -1.0, 1.0])
    gd = g(gr)
    ops.limit_grad(gd, g(p), 0.5, 1.0)
    close(gd, torch.tensor([-1.0, 1.0, 1.0, -1.0, 1.0]), name="limit_grad")


def test_loss_kernels(ops):
    a, b = rnd(5000, seed=1).abs() + 1e-9, rnd(5000, seed=2).abs() + 1e-9
    a[:10] = 1e-9
    b[5:15] = 1e-9
    bb = b.double().requires_grad_()
    want = (torch.log(a.double().clamp(min=1e-7)) - torch.log(bb.clamp(min=1e-7))).abs().mean() * 3.0
    want.backward()
    loss = torch.zeros(1, device=DEV)
    gb = torch.empty(5000, device=DEV)
    ops.l1_loss(loss, gb, g(a), g(b), 1, 5000, 5000, 3.0 / 5000, clip=1e-7, wdev=g(torch.tensor([2.0])))
    gb = gb / 2.0
    close(loss, want.detach().reshape(1), name="log l1")
    close(gb, bb.grad, rtol=1e-4, name="log l1 grad")
    s = rnd(3000, seed=3)
    ss = s.double().requires_grad_()
    w = (torch.clamp(1 + ss, min=0).mean())
    w.backward()
    loss = torch.zeros(1, device=DEV)
    gs = torch.empty(3000, device=DEV)
    ops.hinge_loss(loss, gs, g(s), 3000, 1.0, 1.0 / 3000)
    ya, fr, gg = rnd(40, 24, seed=7), rnd(40, 24, seed=8), rnd(40, 24, seed=9)
    gd = g(gg)
    ops.lrelu_bwd(gd, g(ya), g(fr), 0.5, 0.1, 40, 16, 24, wdev=g(torch.tensor([3.0])))
    wantg = gg.clone()
    wantg[:, :16] = (gg[:, :16] + 1.5 * torch.sign(ya[:, :16] - fr[:, :16])) * torch.where(ya[:, :16] > 0, 1.0, 0.1)
    close(gd, wantg, name='lrelu_bwd strided')
    close(loss, w.detach().reshape(1), name="hinge")
    close(gs, ss.grad, name="hinge grad")
    x = rnd(3, 4000, seed=4) * 0.1 + 0.02
    xx = x.double().requires_grad_()
    c = xx - xx.mean(-1, keepdim=True)
    y = 0.8 * c / (c.abs().max(-1, keepdim=True)[0] + 1e-9)
    gy = rnd(3, 4000, seed=5).double()
    y.backward(gy)
    yd, st = torch.empty(3, 4000, device=DEV), torch.empty(3, 3, device=DEV)
    ops.peaknorm_fwd(yd, st, g(x), 3, 4000)
    close(yd, y, name="peaknorm")
    gx = torch.empty(3, 4000, device=DEV)
    ops.peaknorm_bwd(gx, g(gy.float()), g(x), st, 3, 4000)
    close(gx, xx.grad, rtol=2e-4, name="peaknorm bwd")


@pytest.mark.parametrize("p,T", [(2, 6000), (3, 6000), (7, 6001), (11, 6000)])
def test_period_fold(ops, p, T):
    B = 2
    x = rnd(B, T, seed=1).double().requires_grad_()
    xp = x
    if T % p:
        xp = F.pad(x[:, None], (0, p - T % p), "reflect")[:, 0]
    H = xp.shape[-1] // p
    img = xp.view(B, H, p)  # [b, h, w]
    want = img.permute(0, 2, 1).reshape(-1)  # [b][w][h]
    gw = rnd(B * p * H, seed=2).double()
    (want * gw).sum().backward()
    out = torch.empty(B * p * H, device=DEV)
    ops.period_fold(out, g(x.detach().float()), B, T, p, H)
    close(out, want, name="period fold")
    gx = torch.empty(B, T, device=DEV)
    ops.period_fold_bwd(gx, g(gw.float()), B, T, p, H, False)
    close(gx, x.grad, name="period fold bwd")


def test_spec_power_and_fm_loss(ops):
    rows, nb = 50, 33
    packed = rnd(rows, 2 * nb + 2, seed=1)
    pk = packed.double().requires_grad_()
    for power in (1, 2):
        mag = (pk[:, :nb] ** 2 + pk[:, nb:2 * nb] ** 2)
        out_ref = mag if power == 2 else mag.sqrt()
        go = rnd(rows, nb, seed=2).double()
        pk.grad = None
        out_ref.backward(go)
        out = torch.empty(rows, nb + 3, device=DEV)
        ops.spec_power(out, g(packed), rows, nb, power)
        close(out[:, :nb], out_ref, name=f"spec_power{power}")
        gp = torch.zeros(rows, 2 * nb + 2, device=DEV)
        god = torch.zeros(rows, nb + 3, device=DEV)
        god[:, :nb] = g(go.float())
        ops.spec_power_bwd(gp, god, g(packed), rows, nb, power)
        close(gp[:, :2 * nb], pk.grad[:, :2 * nb], rtol=1e-4, name=f"spec_power{power} bwd")
    B, Fr, nf = 2, 25, 16
    s_err, s_gt = rnd(B * Fr, nf, seed=3).abs(), rnd(B * Fr, nf, seed=4).abs() * 1e-3
    lens = torch.tensor([25, 17])
    mask = (torch.arange(Fr)[None] < lens[:, None]).double().reshape(B * Fr, 1)
    sc = ((s_gt.double() + 1e-7) ** -0.5).clamp(1e-2, 1e2)
    inv = 1.0 / (float(lens.sum()) * nf)
    want = (s_err.double() * sc * mask).sum() * inv
    loss = torch.zeros(1, device=DEV)
    w = torch.empty(B * Fr, nf, device=DEV)
    ops.fm_spec_loss(loss, w, g(s_err), g(s_gt), B, Fr, nf, g(lens.int()), 1e-7, 0.5, 1e-2, 1e2, inv)
    close(loss, want.reshape(1), name="fm loss")
    close(w, sc * mask * inv, rtol=1e-4, name="fm loss weights")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (2, 8, 64), (2, 21, 51), (1, 5, 2), (4, 17, 26),
                                     (40, 47, 39), (24, 94, 77), (48, 47, 51)])
def test_direct_conv32_matches_implicit_gemm(ops, S, H, Win):
    """conv32.hip (LDS-tiled direct conv of the MRD band layers) against the implicit-GEMM path and
    torch.conv2d, incl. partial tiles on every edge; the last two cases have more tiles than the
    persistent kernel has blocks (every block walks over several tiles, prefetched patches, the
    weight double buffer wrapping across tiles), one per tile shape (widths 20 and 39)."""
    Wout = (Win - 1) // 2 + 1
    x = rnd(S * H * Win, 32, seed=1)
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05)
    b = rnd(32, seed=3)
    wp = w.permute(0, 2, 3, 1).reshape(32, 27 * 32).contiguous()       # (Cout, kh*kw, Cin)
    y = torch.full((S * H * Wout, 32), 7.0, device=DEV)
    ops.conv32_s2_fwd(g(x), S, H, Win, Wout, g(wp), g(b), 0.1, y)
    ref = torch.nn.functional.conv2d(x.reshape(S, H, Win, 32).permute(0, 3, 1, 2).double(), w.double(),
                                     b.double(), stride=(1, 2), padding=(1, 4))
    ref = torch.nn.functional.leaky_relu(ref, 0.1).permute(0, 2, 3, 1).reshape(S * H * Wout, 32)
    close(y, ref, name="conv32")
    y2 = torch.empty_like(y)
    ops.gemm(ops.win2d(g(x), S, H, Win, 32, Wout, 3, 9, 2, 1, 4), ops.mat(g(wp)), y2, bias=g(b), lrelu=0.1)
    close(y, y2.cpu().double(), name="conv32-vs-gemm")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (2, 8, 64), (2, 21, 51), (1, 5, 2), (4, 17, 26), (1, 9, 33),
                                     (24, 94, 77), (48, 47, 51), (2, 9, 1)])
def test_direct_conv32_dgrad_matches_autograd(ops, S, H, Win):
    """conv32.hip data gradient (direct transposed conv per column parity) against torch's conv2d
    backward in fp64, incl. odd widths and partial tiles."""
    Wout = (Win - 1) // 2 + 1
    gy = rnd(S * H * Wout, 32, seed=4)
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05)
    wT = w.permute(2, 3, 1, 0).reshape(27, 32, 32).contiguous()        # [tap][ci][co]
    gx = torch.full((S * H * Win, 32), 7.0, device=DEV)
    ops.conv32_s2_dgrad(g(gy), S, H, Win, Wout, g(wT), gx)
    x = torch.zeros(S, 32, H, Win, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w.double(), None, stride=(1, 2), padding=(1, 4))
    y.backward(gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    ref = x.grad.permute(0, 2, 3, 1).reshape(S * H * Win, 32)
    close(gx, ref, name="conv32 dgrad")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (24, 94, 77), (2, 8, 64)])
def test_direct_conv32_dgrad_fused_lrelu_backward(ops, S, H, Win):
    """The data gradient's optional epilogue: leaky-ReLU backward of the layer below (mask by that
    layer's activation y), the feature-matching term w * sign(y - y_real) added in front of it, and
    the column sums of the result (= the bias gradient of the layer below), against torch fp64."""
    Wout = (Win - 1) // 2 + 1
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05)
    gy = rnd(S * H * Wout, 32, seed=4)
    yact = rnd(S * H * Win, 32, seed=5)
    yact = torch.where(yact > 0, yact, 0.1 * yact)          # an activation map: leaky_relu(pre)
    yreal = rnd(S * H * Win, 32, seed=6)
    wdev = torch.tensor([0.7])
    wT = w.permute(2, 3, 1, 0).reshape(27, 32, 32).contiguous()
    x = torch.zeros(S, 32, H, Win, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w.double(), None, stride=(1, 2), padding=(1, 4))
    y.backward(gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    base = x.grad.permute(0, 2, 3, 1).reshape(S * H * Win, 32)
    for use_fm in (False, True):
        want = base.clone()
        if use_fm:
            want = want + 0.25 * 0.7 * torch.sign(yact.double() - yreal.double())
        want = want * torch.where(yact > 0, 1.0, 0.1).double()
        gx = torch.full((S * H * Win, 32), 7.0, device=DEV)
        cs = torch.zeros(32, device=DEV)
        ops.conv32_s2_dgrad(g(gy), S, H, Win, Wout, g(wT), gx, mask=(g(yact), 0, 0.1),
                            fm=(g(yreal), 0, 0.25, g(wdev)) if use_fm else None, colsum=cs)
        close(gx, want, name=f"conv32 dgrad + lrelu bwd (fm={use_fm})")
        close(cs, want.sum(0), rtol=1e-4, name="column sums")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (2, 21, 51), (1, 5, 2), (2, 8, 64), (5, 17, 26)])
def test_direct_conv32_split_bf16(ops, S, H, Win):
    """Split-bf16 instances of the direct MRD conv (forward and data gradient): patch split while it
    is staged, pre-split weight tiles, 3 bf16 MFMAs per product -- ~2^-16 per product."""
    Wout = (Win - 1) // 2 + 1
    x = rnd(S * H * Win, 32, seed=1)
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05)
    b = rnd(32, seed=3)
    gy = rnd(S * H * Wout, 32, seed=4)
    wp = w.permute(0, 2, 3, 1).reshape(32, 27 * 32).contiguous()
    wT = w.permute(2, 3, 1, 0).reshape(27, 32, 32).contiguous()
    xd = x.reshape(S, H, Win, 32).permute(0, 3, 1, 2).double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(xd, w.double(), b.double(), stride=(1, 2), padding=(1, 4))
    ref = torch.nn.functional.leaky_relu(pre, 0.1).permute(0, 2, 3, 1).reshape(S * H * Wout, 32)
    (pre - b.double()[None, :, None, None]).backward(gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    was = ops.GEMM_PRECISION
    ops.set_gemm_precision("bf16x3")
    try:
        y = torch.full((S * H * Wout, 32), 7.0, device=DEV)
        ops.conv32_s2_fwd(g(x), S, H, Win, Wout, g(wp), g(b), 0.1, y)
        gx = torch.full((S * H * Win, 32), 7.0, device=DEV)
        ops.conv32_s2_dgrad(g(gy), S, H, Win, Wout, g(wT), gx)
        gw = torch.zeros(32, 27 * 32, device=DEV)
        ops.conv32_s2_wgrad(g(x), g(gy), S, H, Win, Wout, gw)
    finally:
        ops.GEMM_PRECISION = was
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd.detach(), wd, None, stride=(1, 2), padding=(1, 4)).backward(
        gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    close(gw, wd.grad.permute(0, 2, 3, 1).reshape(32, 27 * 32), rtol=1e-4, name="conv32 wgrad split-bf16")
    close(y, ref.detach(), rtol=3e-5, name="conv32 split-bf16")
    close(gx, xd.grad.permute(0, 2, 3, 1).reshape(S * H * Win, 32), rtol=3e-5, name="conv32 dgrad split-bf16")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (2, 21, 51), (1, 5, 2), (2, 8, 64), (5, 17, 26), (2, 9, 1),
                                     (40, 47, 39), (24, 94, 77), (300, 19, 33)])
def test_direct_conv32_fp32_class(ops, S, H, Win):
    """conv32x6.hip: the direct MRD convs with fp32-CLASS products on the bf16 pipe (three bf16 pieces
    per value, six MFMAs per product; f2g_conv32_desc.precision = 3) -- forward, data gradient
    (both column parities from one staged patch) and weight gradient (bf16 planes of the three pieces,
    transposing fragment reads) at the EXACT-fp32 tolerances against float64, and no
    further from it than three times the exact-fp32 kernels' own error.  The large cases have more
    tiles than the persistent kernels have blocks (prefetched patches, the weight double buffer
    flipping from tile to tile), one per tile shape."""
    Wout = (Win - 1) // 2 + 1
    x = rnd(S * H * Win, 32, seed=1)
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05)
    b = rnd(32, seed=3)
    gy = rnd(S * H * Wout, 32, seed=4)
    wp = w.permute(0, 2, 3, 1).reshape(32, 27 * 32).contiguous()
    wT = w.permute(2, 3, 1, 0).reshape(27, 32, 32).contiguous()
    xd = x.reshape(S, H, Win, 32).permute(0, 3, 1, 2).double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(xd, w.double(), b.double(), stride=(1, 2), padding=(1, 4))
    ref = torch.nn.functional.leaky_relu(pre, 0.1).permute(0, 2, 3, 1).reshape(S * H * Wout, 32).detach()
    pre.backward(gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    gref = xd.grad.permute(0, 2, 3, 1).reshape(S * H * Win, 32)
    yact = rnd(S * H * Win, 32, seed=5)
    yact = torch.where(yact > 0, yact, 0.1 * yact)
    was = ops.GEMM_PRECISION
    out = {}
    try:
        for mode in ("fp32", "bf16x6"):
            ops.set_gemm_precision(mode)
            y = torch.full((S * H * Wout, 32), 7.0, device=DEV)
            ops.conv32_s2_fwd(g(x), S, H, Win, Wout, g(wp), g(b), 0.1, y)
            gx = torch.full((S * H * Win, 32), 7.0, device=DEV)
            ops.conv32_s2_dgrad(g(gy), S, H, Win, Wout, g(wT), gx)
            gm = torch.full((S * H * Win, 32), 7.0, device=DEV)
            cs = torch.zeros(32, device=DEV)
            ops.conv32_s2_dgrad(g(gy), S, H, Win, Wout, g(wT), gm, mask=(g(yact), 0, 0.1), colsum=cs)
            gw = torch.zeros(32, 27 * 32, device=DEV)
            ops.conv32_s2_wgrad(g(x), g(gy), S, H, Win, Wout, gw)
            out[mode] = (y.cpu().double(), gx.cpu().double(), gm.cpu().double(), gw.cpu().double(), cs.cpu().double())
    finally:
        ops.GEMM_PRECISION = was
    want_m = gref * torch.where(yact > 0, 1.0, 0.1).double()
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd.detach(), wd, None, stride=(1, 2), padding=(1, 4)).backward(
        gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    want_w = wd.grad.permute(0, 2, 3, 1).reshape(32, 27 * 32)
    for i, (want, name, tol) in enumerate(((ref, "forward", 2e-5), (gref, "data gradient", 2e-5),
                                           (want_m, "masked data gradient", 2e-5),
                                           (want_w, "weight gradient", 1e-4))):
        close(out["bf16x6"][i], want, rtol=tol, name=f"conv32 fp32-class {name}")
        close(out["fp32"][i], want, rtol=tol, name=f"conv32 exact fp32 {name}")     # (not only the yardstick below)
        e6 = float((out["bf16x6"][i] - want).abs().max())
        e0 = float((out["fp32"][i] - want).abs().max())
        assert e6 <= 3.0 * e0 + 1e-7 * float(want.abs().max()), (name, e6, e0)
    close(out["bf16x6"][4], want_m.sum(0), rtol=1e-4, name="column sums")


@pytest.mark.parametrize("S,H,W", [(3, 47, 13), (2, 94, 7), (5, 19, 33), (128, 47, 32), (2, 5, 1), (3, 9, 112),
                                   (300, 23, 17), (1, 200, 20)])
def test_direct_conv33_fp32_class(ops, S, H, W):
    """conv32x6.hip conv33_x6_kernel (round 5): the (3, 3) fifth layer of an MRD band stack as a direct kernel
    with fp32-class products -- forward (bias + leaky ReLU, written into a band's slice of a wider concatenated
    map) and data gradient (the same kernel over a slice of the concatenated gradient map with flipped /
    transposed weights, as fused_disc._conv2d_dgrad builds them) against float64 conv2d + autograd at the
    exact-fp32 tolerances; band widths on both sides of the tile-row rule (R whole rows per 256-pixel tile),
    more tiles than the persistent kernel has blocks, single-column and 112-column (the widest) images."""
    from flow2gan_amd import fused_disc as fd
    x = rnd(S * H * W, 32, seed=1)
    w = rnd(32, 32, 3, 3, seed=2, scale=0.08)
    b = rnd(32, seed=3)
    foff, Wcat = 5, W + 9                                  # the band's slice of the concatenated maps
    gcat = rnd(S * H * Wcat, 32, seed=4)
    xd = x.reshape(S, H, W, 32).permute(0, 3, 1, 2).double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(xd, w.double(), b.double(), padding=(1, 1))
    ref = torch.nn.functional.leaky_relu(pre, 0.1).permute(0, 2, 3, 1).detach()           # (S, H, W, 32)
    gy = gcat.reshape(S, H, Wcat, 32)[:, :, foff:foff + W]
    pre.backward(gy.permute(0, 3, 1, 2).double())
    gref = xd.grad.permute(0, 2, 3, 1).reshape(S * H * W, 32)
    was = ops.GEMM_PRECISION
    try:
        ops.set_gemm_precision("bf16x6")
        wd = torch.nn.Parameter(g(w))
        wp = ops.derived(wd, "pack", fd.pack_conv_weight)
        cat = torch.full((S * H * Wcat, 32), 7.0, device=DEV)
        ops.conv33(g(x), S, H, W, wp, g(b), 0.1, cat, y_off=foff * 32, y_line=Wcat * 32, y_seq=H * Wcat * 32)
        catv = cat.cpu().reshape(S, H, Wcat, 32)
        close(catv[:, :, foff:foff + W].double(), ref, rtol=2e-5, name="conv33 forward")
        assert bool((catv[:, :, :foff] == 7.0).all()) and bool((catv[:, :, foff + W:] == 7.0).all())
        gx = torch.full((S * H * W, 32), 7.0, device=DEV)
        fd._conv2d_dgrad(g(gcat), S, H, W, 32, wd, 1, W, gx, g_line=Wcat * 32, g_seq=H * Wcat * 32, g_off=foff * 32)
        close(gx.cpu().double(), gref, rtol=2e-5, name="conv33 data gradient")
        # the same with the leaky-ReLU backward of the layer below and the bias-gradient column sums fused
        yact = rnd(S * H * W, 32, seed=5)
        yact = torch.where(yact > 0, yact, 0.1 * yact)
        gm = torch.full((S * H * W, 32), 7.0, device=DEV)
        cs = torch.zeros(32, device=DEV)
        fd._conv2d_dgrad(g(gcat), S, H, W, 32, wd, 1, W, gm, g_line=Wcat * 32, g_seq=H * Wcat * 32, g_off=foff * 32,
                         mask=(g(yact), 0, 0.1), colsum=cs)
        want_m = gref * torch.where(yact > 0, 1.0, 0.1).double()
        close(gm.cpu().double(), want_m, rtol=2e-5, name="conv33 masked data gradient")
        close(cs.cpu().double(), want_m.sum(0), rtol=1e-4, name="conv33 column sums")
        # (round 6) the weight gradient as a direct kernel: x = the layer's input, the gradient = the band's slice
        # of the concatenated gradient map; accumulates onto what gw holds
        gwref = torch.autograd.grad(torch.nn.functional.conv2d(xd, (wq := w.double().requires_grad_(True)),
                                                               padding=(1, 1)),
                                    wq, gy.permute(0, 3, 1, 2).double())[0]          # (co, ci, 3, 3)
        gwp = torch.full((32, 9 * 32), 0.5, device=DEV)
        ops.conv33_wgrad(g(x), g(gcat), S, H, W, gwp, g_off=foff * 32, g_line=Wcat * 32, g_seq=H * Wcat * 32)
        got = fd.unpack_conv_grad(gwp, (32, 32, 3, 3)).cpu().double() - 0.5
        close(got, gwref, rtol=3e-5, name="conv33 weight gradient")
    finally:
        ops.GEMM_PRECISION = was


@pytest.mark.parametrize("S,H,W,lo,Wtot", [(2, 11, 34, 7, 50), (3, 8, 32, 0, 32), (1, 5, 3, 2, 9), (2, 21, 77, 10, 100),
                                           (300, 40, 64, 3, 70)])
def test_conv2ch_direct_kernels_match_autograd(ops, S, H, W, lo, Wtot):
    """conv2ch.hip (first MRD layer, 2 -> 32 channels on a band of the interleaved spectrogram):
    forward, weight gradient and data gradient against torch conv2d + autograd in fp64.  The last case has
    600 column tiles x sequences (the persistent forward walking down its rows with prefetched patches) and
    3000 tiles for the weight gradient (more than one tile per block: the prefetching tile walk)."""
    ld = ops.pad4(2 * Wtot)
    spec = torch.zeros(S * H, ld)
    spec[:, :2 * Wtot] = rnd(S * H, 2 * Wtot, seed=1)
    w, b = rnd(32, 2, 3, 9, seed=2, scale=0.2), rnd(32, seed=3)
    wp = w.permute(0, 2, 3, 1).reshape(32, 54).contiguous()
    y = torch.full((S * H * W, 32), 7.0, device=DEV)
    sd = g(spec)
    ops.conv2ch_fwd(sd, H * ld, ld, lo * 2, S, H, W, g(wp), g(b), 0.1, y)
    x = spec[:, :2 * Wtot].reshape(S, H, Wtot, 2)[:, :, lo:lo + W].permute(0, 3, 1, 2).double()
    x.requires_grad_(True)
    wd = w.double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(x, wd, b.double(), padding=(1, 4))
    ref = torch.nn.functional.leaky_relu(pre, 0.1)
    close(y, ref.permute(0, 2, 3, 1).reshape(S * H * W, 32), name="conv2ch fwd")
    gy = rnd(S * H * W, 32, seed=5)
    pre.backward(gy.reshape(S, H, W, 32).permute(0, 3, 1, 2).double())
    gw = torch.zeros(32, 54, device=DEV)
    ops.conv2ch_wgrad(sd, H * ld, ld, lo * 2, S, H, W, g(gy), gw)
    close(gw, wd.grad.permute(0, 2, 3, 1).reshape(32, 54), rtol=1e-4, name="conv2ch wgrad")
    wt = w.permute(2, 3, 1, 0).reshape(27, 2, 32).contiguous()
    gspec = torch.full((S * H, ld), 7.0, device=DEV)
    ops.conv2ch_dgrad(g(gy), S, H, W, g(wt), gspec, H * ld, ld, lo * 2)
    got = gspec[:, :2 * Wtot].reshape(S, H, Wtot, 2)[:, :, lo:lo + W]
    close(got, x.grad.permute(0, 2, 3, 1), name="conv2ch dgrad")
    rest = gspec[:, :2 * Wtot].reshape(S, H, Wtot, 2)
    assert float(rest[:, :, :lo].min()) == 7.0 if lo > 0 else True     # outside the band: untouched


@pytest.mark.parametrize("S,H,W", [(2, 11, 34), (3, 8, 32), (1, 5, 3), (2, 21, 77)])
def test_convpost_direct_kernels_match_autograd(ops, S, H, W):
    """conv_post of an MRD resolution (32 -> 1, 3x3) as direct kernels: forward, weight gradient,
    data gradient against torch conv2d + autograd in fp64."""
    xr = rnd(S * H * W, 32, seed=1)
    w, b = rnd(1, 32, 3, 3, seed=2, scale=0.2), rnd(1, seed=3)
    w9 = w.permute(0, 2, 3, 1).reshape(9, 32).contiguous()
    y = torch.full((S * H * W,), 7.0, device=DEV)
    ops.convpost_fwd(g(xr), S, H, W, g(w9), g(b), y)
    x = xr.reshape(S, H, W, 32).permute(0, 3, 1, 2).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    out = torch.nn.functional.conv2d(x, wd, b.double(), padding=1)
    close(y, out.reshape(-1), name="convpost fwd")
    gy = rnd(S * H * W, seed=5)
    out.backward(gy.reshape(S, 1, H, W).double())
    gw = torch.zeros(9 * 32, device=DEV)
    ops.convpost_wgrad(g(xr), S, H, W, g(gy), gw)
    close(gw, wd.grad.permute(0, 2, 3, 1).reshape(-1), rtol=1e-4, name="convpost wgrad")
    gx = torch.full((S * H * W, 32), 7.0, device=DEV)
    ops.convpost_dgrad(g(gy), S, H, W, g(w9), gx)
    close(gx, x.grad.permute(0, 2, 3, 1).reshape(S * H * W, 32), name="convpost dgrad")


@pytest.mark.parametrize("S,H,Win", [(3, 11, 13), (2, 8, 64), (2, 21, 51), (1, 5, 2), (5, 17, 26), (40, 47, 77),
                                     (64, 94, 19)])
def test_direct_conv32_wgrad_matches_autograd(ops, S, H, Win):
    """conv32.hip weight gradient (direct, taps spread over the waves) vs torch autograd in fp64.  The last
    two cases have 720 / 768 tiles -- more than the 256 blocks of the double-buffered kernel
    (conv32_s2_wgrad_p_kernel: the next tile staged behind the current tile's MFMAs), one per tile shape."""
    Wout = (Win - 1) // 2 + 1
    xr, gy = rnd(S * H * Win, 32, seed=1), rnd(S * H * Wout, 32, seed=4)
    w = rnd(32, 32, 3, 9, seed=2, scale=0.05).double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xr.reshape(S, H, Win, 32).permute(0, 3, 1, 2).double(), w, None,
                                   stride=(1, 2), padding=(1, 4))
    y.backward(gy.reshape(S, H, Wout, 32).permute(0, 3, 1, 2).double())
    gw = torch.zeros(32, 27 * 32, device=DEV)
    ops.conv32_s2_wgrad(g(xr), g(gy), S, H, Win, Wout, gw)
    close(gw, w.grad.permute(0, 2, 3, 1).reshape(32, 27 * 32), rtol=1e-4, name="conv32 wgrad")


@pytest.mark.parametrize("C,rows", [(768, 200), (512, 300), (384, 128 * 3), (512, 31)])
def test_fused_mlp_matches_the_two_gemm_arithmetic(ops, C, rows):
    """csrc/fusedmlp.hip against a torch fp32 restatement of the same arithmetic (bf16 operands,
    fp32 accumulation, the hidden activation rounded to bf16 between the two products):
    out = W2 . bf16(PReLU(W1 . z + b1)) + b2 + gamma * x   (modules.py:487-495).  Row counts that
    end inside a tile, and one smaller than a tile."""
    H = 3 * C
    g = torch.Generator().manual_seed(C + rows)
    z = torch.randn(rows, C, generator=g).to(torch.bfloat16)
    w1 = torch.randn(H, C, generator=g) * 0.05
    w2 = torch.randn(C, H, generator=g) * 0.03
    b1 = torch.randn(H, generator=g) * 0.1
    al = 0.25 + 0.2 * torch.randn(H, generator=g)
    b2 = torch.randn(C, generator=g) * 0.1
    x = torch.randn(rows, C, generator=g)
    gam = 0.5 + torch.rand(C, generator=g)
    w1b, w2b = w1.to(torch.bfloat16).float(), w2.to(torch.bfloat16).float()
    a = z.float().double() @ w1b.double().t() + b1.double()
    p = (a.clamp(min=0) + al.double() * a.clamp(max=0)).float().to(torch.bfloat16)
    want = p.double() @ w2b.double().t() + b2.double() + gam.double() * x.double()
    out = torch.full((rows, C), float("nan"), device=DEV)
    wp = ops.mlp_pack(w1.to(DEV), w2.to(DEV))
    assert wp.dtype == torch.bfloat16 and wp.numel() == 2 * C * H
    # the packed stream is a permutation of the two rounded matrices
    assert abs(float(wp.float().double().sum()) - float(w1b.double().sum() + w2b.double().sum())) < 1e-2
    ops.fused_mlp(z.to(DEV), wp, b1.to(DEV), al.to(DEV), b2.to(DEV), x.to(DEV), gam.to(DEV), out,
                  rows, C, H)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    # p is rounded to bf16 from an fp32 sum whose order differs: a few elements may land on the
    # neighbouring bf16 value (2^-8 relative of one hidden unit's contribution)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err < 2e-3 * scale, (err, scale)
    rms_err = float((got - want).pow(2).mean().sqrt())
    assert rms_err < 1e-4 * scale, (rms_err, scale)
    # every way of cutting the hidden dimension between blocks: one part (plain stores), an uneven
    # cut, one slab per part (atomic accumulation onto the zeroed output)
    for parts in (1, 2, 5, H // 128):
        outp = torch.full((rows, C), float("nan"), device=DEV)
        ops.fused_mlp(z.to(DEV), wp, b1.to(DEV), al.to(DEV), b2.to(DEV), x.to(DEV), gam.to(DEV),
                      outp, rows, C, H, parts=parts)
        errp = float((outp.cpu().double() - want).abs().max())
        assert errp < 2e-3 * scale, (parts, errp, scale)
    # without residual / biases
    out2 = torch.empty(rows, C, device=DEV)
    ops.fused_mlp(z.to(DEV), wp, None, al.to(DEV), None, None, None, out2, rows, C, H)
    a0 = z.float().double() @ w1b.double().t()
    p0 = (a0.clamp(min=0) + al.double() * a0.clamp(max=0)).float().to(torch.bfloat16)
    want2 = p0.double() @ w2b.double().t()
    assert float((out2.cpu().double() - want2).abs().max()) < 2e-3 * float(want2.abs().max())


@pytest.mark.parametrize("kernel", ["images", "in-kernel split"])
@pytest.mark.parametrize("M,N,K", [(6016, 768, 2304), (1500, 200, 96), (4099, 1152, 384)])
def test_gemm_fp32_class_on_the_bf16_pipe(ops, M, N, K, kernel, monkeypatch):
    """precision 3 (`bf16x6`): three bf16 pieces per operand, six MFMAs per product.  Against float64 its
    error must be of the class of the exact-fp32 kernel's on the same operands (max over ALL entries of
    |err| / sum |a w| below 1e-6 and within 3x of the fp32 MFMA's; plain bf16 is at 1e-3), on ragged extents, with bias / residual / PReLU epilogues
    and as a data gradient (form 1 through the cached transpose): gemm_x6_kernel / gemm_x6f_kernel (128 x 128
    tiles, two blocks per CU)."""
    gen = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=gen) * (1.0 + 3.0 * torch.rand(M, K, generator=gen))
    w = torch.randn(N, K, generator=gen) * 0.05
    bias = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen)
    gam = 0.5 + torch.rand(N, generator=gen)
    ref = a.double() @ w.double().t()
    mag = a.abs().double() @ w.abs().double().t()
    ad, wd = g(a), torch.nn.Parameter(g(w))
    # gemm_x6_kernel over f2g_split_bf16x3 images / gemm_x6f_kernel over the fp32 operands themselves
    monkeypatch.setattr(ops, "X6F", 0 if kernel == "images" else 1)
    # (recorded by monkeypatch BEFORE the direct assignments below: its teardown restores what it saw first, and
    # used to leave X6_MIN_K = 32 behind for the rest of the module)
    monkeypatch.setattr(ops, "X6_MIN_K", ops.X6_MIN_K)
    monkeypatch.setattr(ops, "X6F_MIN_K", ops.X6F_MIN_K)
    was = ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS
    try:
        ops.set_gemm_precision("bf16x6")
        ops.X6_MIN_K, ops.X6_MIN_ROWS = 32, 1          # (the model's profitability thresholds off)
        out = torch.full((M, N), float("nan"), device=DEV)
        timer = ops.GemmTimer()
        ops.GEMM_TIMER = timer
        try:
            ops.gemm(ops.mat(ad), ops.mat(wd), out)
        finally:
            ops.GEMM_TIMER = None
        assert timer.paths[-1] == "x6", timer.paths          # (the six-product kernel did run)
        err = float(((out.cpu().double() - ref).abs() / mag).max())
        assert err < 1e-6, err
        out2 = torch.empty(M, N, device=DEV)
        ops.gemm(ops.mat(ad), ops.mat(wd), out2, bias=g(bias), res=g(res), gamma=g(gam))
        want2 = ref + bias.double() + gam.double() * res.double()
        assert float(((out2.cpu().double() - want2).abs() / (mag + 1)).max()) < 1e-6
        # the result's own three-piece image, written by the kernel after its epilogue (E.x3_out)
        monkeypatch.setattr(ops, "X6F", 2 if kernel != "images" else 0)     # (1 = no images anywhere)
        monkeypatch.setattr(ops, "X6F_MIN_K", 32)
        if kernel != "images":
            monkeypatch.setattr(ops, "X6_MIN_K", 1 << 20)                   # every K below it: in-kernel split
        out3 = ops.x3_reserve(torch.full((M, N), float("nan"), device=DEV))
        ops.gemm(ops.mat(ad), ops.mat(wd), out3, bias=g(bias), lrelu=0.1, x3_out=True)
        if kernel == "images" or (N >= ops.X6F_MIN_N and ((M + 127) // 128) * ((N + 127) // 128) >= ops.X6F_MIN_TILES):
            img = getattr(out3, "_f2g_x3", None)
            assert img is not None and not getattr(out3, "_f2g_x3_bad", False)
            assert torch.equal(img.view(torch.int16), ops.x3_flat_image(out3).view(torch.int16))
        v3 = ref + bias.double()
        assert float(((out3.cpu().double() - torch.where(v3 > 0, v3, 0.1 * v3)).abs() / (mag + 1)).max()) < 1e-6
        monkeypatch.setattr(ops, "X6F", 0 if kernel == "images" else 1)
        monkeypatch.setattr(ops, "X6_MIN_K", 32)
        al = 0.25 + 0.2 * torch.rand(N, generator=gen)
        pre, act = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        ops.gemm(ops.mat(ad), ops.mat(wd), pre, bias=g(bias), prelu=g(al), prelu_out=act)
        v = ref + bias.double()
        assert float(((pre.cpu().double() - v).abs() / (mag + 1)).max()) < 1e-6
        assert float(((act.cpu().double() - (v.clamp(min=0) + al.double() * v.clamp(max=0))).abs() / (mag + 1)).max()) < 1e-6
        # data gradient: g (M, N) @ w (N, K) -> (M, K)
        if N % 32 == 0:
            gy = torch.randn(M, N, generator=gen)
            gx = torch.empty(M, K, device=DEV)
            ops.gemm(ops.mat(g(gy)), ops.mat(wd), gx, form=1)
            refg = gy.double() @ w.double()
            magg = gy.abs().double() @ w.abs().double()
            assert float(((gx.cpu().double() - refg).abs() / magg).max()) < 1e-6
        # the exact-fp32 result on the same operands, for scale
        ops.set_gemm_precision("fp32")
        o32 = torch.empty(M, N, device=DEV)
        ops.gemm(ops.mat(ad), ops.mat(wd), o32)
        e32 = float(((o32.cpu().double() - ref).abs() / mag).max())
        assert err < 3 * e32 + 1e-7, (err, e32)
    finally:
        ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS = was


@pytest.mark.parametrize("x6p", [0, 2], ids=["rows", "pingpong"])
@pytest.mark.parametrize("B2,T,p", [(4, 24000, 3), (2, 11025, 11), (6, 6000, 2)])
def test_fp32_class_gemm_over_halo_windows_and_producer_images(ops, B2, T, p, x6p, monkeypatch, lib_option):
    """precision 3 over the MPD stack's operands: conv windows into the halo maps (rows a stride apart, K
    contiguous over the taps) read from the flat three-piece image of the map, and that image written by
    the PRODUCING GEMM (E.x3_out: bias + leaky ReLU forward, leaky-ReLU-backward mask in the data
    gradients, stride residues interleaving rows).  Every produced image must equal f2g_split_bf16x3 of the
    stored map bit for bit (zero halo rows included), and maps / gradients must agree with the exact-fp32
    path to fp32 rounding.  pingpong: the tap-walking gemm_x6p_kernel (two wave groups half a step apart, wide
    epilogue that writes map and image from the same registers) wherever its geometry allows, whatever the
    grid size; rows: switched off -- the windows are read row by row through gemm_x6_kernel (what grids that do
    not fill the chip and layers of fewer than 64 channels get)."""
    from flow2gan_amd import fused_disc as fd
    lib_option("x6p", x6p)
    gen = torch.Generator().manual_seed(T + p)
    x2 = g(0.1 * torch.randn(B2, T, generator=gen))
    ch = fd.MPD_CH
    prm = []
    for l in range(5):
        prm += [g(torch.randn(ch[l + 1], ch[l], 5, 1, generator=gen) * (2.0 / (5 * ch[l]) ** 0.5)),
                g(0.01 * torch.randn(ch[l + 1], generator=gen))]
    prm += [g(0.05 * torch.randn(1, 1024, 3, 1, generator=gen)), g(torch.zeros(1))]
    was = ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS
    try:
        ops.set_gemm_precision("fp32")
        ref = fd._mpd_forward_one(x2, p, prm)
        S, hs = ref["S"], ref["hs"]
        gmaps = {}
        for l in (4, 3, 2):
            gm = fd._halo_rows(S, hs[l + 1], ch[l + 1], DEV)
            gm.view(S, hs[l + 1] + 2 * fd.HALO, ch[l + 1])[:, fd.HALO:fd.HALO + hs[l + 1]] = \
                g(torch.randn(S, hs[l + 1], ch[l + 1], generator=gen))
            gmaps[l] = gm
        ref_gx = {l: fd._conv1d_dgrad(gmaps[l], S, hs[l + 1], ch[l + 1], prm[2 * l], fd.MPD_STRIDE[l], 2, hs[l],
                                      mask=(ref["acts"][l], 0, fd.SLOPE)) for l in (4, 3, 2)}
        ops.set_gemm_precision("bf16x6")
        ops.X6_MIN_K, ops.X6_MIN_ROWS = 32, 1
        timer = ops.GemmTimer()
        ops.GEMM_TIMER = timer
        try:
            st = fd._mpd_forward_one(x2, p, prm)
        finally:
            ops.GEMM_TIMER = None
        assert timer.paths.count("x6") == 4, timer.paths      # layers 1..4 (layer 0 and conv_post are HBM streams)
        # the stride-1 layer 4 runs on the tap-walking instance where a tile's positions fit its LDS
        # window (13-row sequences at p = 11: 172 positions, the plain instance)
        assert getattr(timer, "x6_tap", 0) == (0 if p == 11 else 1), getattr(timer, "x6_tap", 0)
        nimg = 0
        for l, (y, yr) in enumerate(zip(st["acts"], ref["acts"])):
            scale = float(yr.abs().max())
            assert float((y - yr).abs().max()) < 2e-5 * scale, l
            img = getattr(y, "_f2g_x3", None)
            assert (img is not None) == (l in (2, 3, 4)), l      # the maps the next GEMM reads
            if img is not None:
                assert not getattr(y, "_f2g_x3_bad", False)
                assert torch.equal(img.view(torch.int16), ops.x3_flat_image(y).view(torch.int16)), l
                nimg += 1
        for l in (4, 3, 2):
            gx = fd._conv1d_dgrad(gmaps[l], S, hs[l + 1], ch[l + 1], prm[2 * l], fd.MPD_STRIDE[l], 2, hs[l],
                                  mask=(st["acts"][l], 0, fd.SLOPE))
            # (pixels whose activation changed sign between the two forward passes would differ by the slope)
            same = (st["acts"][l] > 0) == (ref["acts"][l] > 0)
            d = ((gx - ref_gx[l]).abs() * same).max()
            assert float(d) < 2e-5 * float(ref_gx[l].abs().max()), l
            img = getattr(gx, "_f2g_x3", None)
            assert img is not None and not getattr(gx, "_f2g_x3_bad", False), l
            assert torch.equal(img.view(torch.int16), ops.x3_flat_image(gx).view(torch.int16)), l
            nimg += 1
        assert nimg == 6
    finally:
        ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS = was


@pytest.mark.parametrize("C,rows,tile", [(768, 64 * 130 + 7, 64), (512, 96 * 128 + 5, 96), (512, 64 * 129, 64),
                                         (384, 128 * 128 + 3, 128), (384, 64 * 130 + 1, 64)])
def test_fused_mlp_taller_tiles(ops, C, rows, tile):
    """The fused kernel picks its tile height by the row count (fusedmlp.hip: pick_rt; the small cases
    above all run on 32-row tiles): row counts that select the 64 / 96 / 128-row instances, against a
    float64 torch restatement of the same arithmetic (computed on the GPU), with and without the z
    prologue (f2g_fused_block against f2g_dwnorm_fwd + f2g_fused_mlp on the same rows)."""
    H = 3 * C
    assert (rows + tile - 1) // tile >= 128            # (what makes pick_rt take this height)
    gen = torch.Generator().manual_seed(C + rows)
    z = torch.randn(rows, C, generator=gen).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(H, C, generator=gen) * 0.05).to(DEV)
    w2 = (torch.randn(C, H, generator=gen) * 0.03).to(DEV)
    b1 = (torch.randn(H, generator=gen) * 0.1).to(DEV)
    al = (0.25 + 0.2 * torch.randn(H, generator=gen)).to(DEV)
    b2 = (torch.randn(C, generator=gen) * 0.1).to(DEV)
    x = torch.randn(rows, C, generator=gen).to(DEV)
    gam = (0.5 + torch.rand(C, generator=gen)).to(DEV)
    w1b, w2b = w1.to(torch.bfloat16).double(), w2.to(torch.bfloat16).double()
    a = z.double() @ w1b.t() + b1.double()
    p = (a.clamp(min=0) + al.double() * a.clamp(max=0)).float().to(torch.bfloat16)
    want = p.double() @ w2b.t() + b2.double() + gam.double() * x.double()
    wp = ops.mlp_pack(w1, w2)
    out = torch.full((rows, C), float("nan"), device=DEV)
    ops.fused_mlp(z, wp, b1, al, b2, x, gam, out, rows, C, H)
    assert torch.isfinite(out).all()
    scale = float(want.abs().max())
    assert float((out.double() - want).abs().max()) < 2e-3 * scale
    assert float((out.double() - want).pow(2).mean().sqrt()) < 1e-4 * scale
    # the block kernel on the same tile height: rows = B * F with ragged lengths
    B = 8
    if rows % B == 0:
        Fr = rows // B
    else:
        Fr, rows = rows // B, (rows // B) * B
    xs = x[:rows]
    lens = g(torch.tensor([Fr, Fr - 3, Fr // 2, Fr, Fr - 1, 7, Fr, Fr - 11]).int())
    args = (B, Fr, C, 7, lens, g(torch.randn(C, 1, 7, generator=gen) * 0.3), g(torch.randn(C, generator=gen) * 0.1),
            g(torch.randn(C, generator=gen) * 0.1), g(torch.tensor([0.7])))
    zz = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16)
    ops.dwnorm_fwd(xs, zz, *args, z_format=2)
    want_b = torch.empty(rows, C, device=DEV)
    ops.fused_mlp(zz, wp, b1, al, b2, xs, gam, want_b, rows, C, H)
    got_b = torch.full((rows, C), float("nan"), device=DEV)
    ops.fused_block(xs, *args, wp, b1, al, b2, gam, got_b, H)
    sb = float(want_b.abs().max())
    assert torch.isfinite(got_b).all()
    assert float((got_b - want_b).abs().max()) < 2e-3 * sb
    assert float((got_b - want_b).pow(2).mean().sqrt()) < 2e-5 * sb


def test_fused_block_multi_equals_separate_launches(ops):
    """f2g_fused_block_multi (the same layer of several branches in one launch, tiles ordered by
    decreasing cost) against one f2g_fused_block per entry: the same body row by row -- entries handed over in an order that is NOT the cost order, ragged
    row counts (partial last tiles), one entry without condition / time inputs."""
    B, K = 3, 7
    gen = torch.Generator().manual_seed(11)
    entries, wants = [], []
    for C, Fr, up, cond in ((384, 94, 4, True), (768, 23, 1, True), (512, 47, 2, False), (768, 5, 1, True)):
        H = 3 * C
        Fc = (Fr + up - 1) // up
        NC = 2 * C
        x = g(torch.randn(B * Fr, C, generator=gen))
        e = dict(x=x, B=B, F=Fr, Cc=C, K=K, lens=g(torch.tensor([Fr, max(1, Fr - 4), max(1, Fr // 2)]).int()),
                 w_dw=g(torch.randn(C, 1, K, generator=gen) * 0.3), b_dw=g(torch.randn(C, generator=gen) * 0.1),
                 beta=g(torch.randn(C, generator=gen) * 0.1), log_scale=g(torch.tensor([0.6])),
                 wp=ops.mlp_pack(g(torch.randn(H, C, generator=gen) * 0.05), g(torch.randn(C, H, generator=gen) * 0.03)),
                 b1=g(torch.randn(H, generator=gen) * 0.1), alpha=g(0.25 + 0.2 * torch.randn(H, generator=gen)),
                 b2=g(torch.randn(C, generator=gen) * 0.1), gamma=g(0.5 + torch.rand(C, generator=gen)),
                 out=torch.full((B * Fr, C), float("nan"), device=DEV), Hh=H)
        if cond:
            e.update(cproj=g(torch.randn(B * Fc, NC, generator=gen)), ldcp=NC, Fc=Fc, up=up, cp_off=C // 2,
                     te=g(torch.randn(B, NC, generator=gen) * 0.3), ldte=NC, te_off=C // 2)
        want = torch.empty(B * Fr, C, device=DEV)
        ops.fused_block(x, B, Fr, C, K, e["lens"], e["w_dw"], e["b_dw"], e["beta"], e["log_scale"], e["wp"],
                        e["b1"], e["alpha"], e["b2"], e["gamma"], want, H, e.get("cproj"), e.get("ldcp", 0),
                        e.get("Fc", 0), e.get("up", 1), e.get("cp_off", 0), e.get("te"), e.get("ldte", 0),
                        e.get("te_off", 0))
        entries.append(e)
        wants.append(want)
    outs = ops.fused_block_multi(entries)
    torch.cuda.synchronize()
    for e, got, want in zip(entries, outs, wants):
        assert torch.isfinite(got).all(), e["Cc"]
        # the same arithmetic row by row; a launch may pick another tile height than the multi launch
        # (another instance of the kernel: the compiler's contraction choices in the z prologue can
        # differ, and a z value that crosses a bf16 rounding boundary moves the outputs it feeds)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) < 2e-3 * scale, (e["Cc"], e["F"])
        assert float((got - want).pow(2).mean().sqrt()) < 2e-5 * scale, (e["Cc"], e["F"])
    # a single entry and the refusal of a fifth
    one = dict(entries[2]); one["out"] = torch.empty_like(wants[2])
    got1 = ops.fused_block_multi([one])[0]
    assert float((got1 - wants[2]).abs().max()) < 2e-3 * float(wants[2].abs().max())
    with pytest.raises(Exception):
        ops.fused_block_multi(entries + [entries[0]])


@pytest.mark.parametrize("C,B,Fr,up", [(768, 3, 47, 1), (512, 5, 94, 2), (384, 3, 94, 4), (512, 2, 9, 1)])
def test_fused_block_matches_dwnorm_plus_fused_mlp(ops, C, B, Fr, up):
    """f2g_fused_block (dwconv7 + BiasNorm + cond + time scale in the fused kernel's prologue, z in
    LDS only) against the two kernels it replaces -- f2g_dwnorm_fwd writing z as bf16, then
    f2g_fused_mlp: the same arithmetic in the same order, so the outputs agree to fp32 rounding --
    with ragged lengths, items shorter than a tile (tiles straddle item boundaries), a condition
    shorter than the frames and a stacked cond / time buffer read at an offset."""
    H, K = 3 * C, 7
    Fc = (Fr + up - 1) // up - (1 if up > 1 else 0)      # the last condition row is missing: zeros
    gen = torch.Generator().manual_seed(C + Fr)
    x = torch.randn(B * Fr, C, generator=gen)
    lens = torch.tensor([Fr, max(1, Fr - 5), max(1, Fr // 2), Fr, 3][:B])
    w_dw = torch.randn(C, 1, K, generator=gen) * 0.3
    b_dw = torch.randn(C, generator=gen) * 0.1
    beta = torch.randn(C, generator=gen) * 0.1
    ls = torch.tensor([0.7])
    NC = 2 * C
    cp_all = torch.randn(B * Fc, NC, generator=gen)
    te_all = torch.randn(B, NC, generator=gen) * 0.3
    w1 = torch.randn(H, C, generator=gen) * 0.05
    w2 = torch.randn(C, H, generator=gen) * 0.03
    b1 = torch.randn(H, generator=gen) * 0.1
    al = 0.25 + 0.2 * torch.randn(H, generator=gen)
    b2 = torch.randn(C, generator=gen) * 0.1
    gam = 0.5 + torch.rand(C, generator=gen)
    xd, lens_d = g(x), g(lens.int())
    wp = ops.mlp_pack(g(w1), g(w2))
    args = (B, Fr, C, K, lens_d, g(w_dw), g(b_dw), g(beta), g(ls))
    for use_cond in (True, False):
        cpa = (g(cp_all), NC, Fc, up, C // 2, g(te_all), NC, C // 2) if use_cond else (None, 0, 0, 1, 0, None, 0, 0)
        z = torch.empty(B * Fr, C, device=DEV, dtype=torch.bfloat16)
        ops.dwnorm_fwd(xd, z, *args, *cpa, z_format=2)
        want = torch.empty(B * Fr, C, device=DEV)
        ops.fused_mlp(z, wp, g(b1), g(al), g(b2), xd, g(gam), want, B * Fr, C, H)
        got = torch.full((B * Fr, C), float("nan"), device=DEV)
        ops.fused_block(xd, *args, wp, g(b1), g(al), g(b2), g(gam), got, H, *cpa)
        assert torch.isfinite(got).all()
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        # identical z (same lane -> channel mapping, same order of operations); what may differ is
        # nothing but the compiler's contraction choices inside the prologue
        assert err < 2e-3 * scale, (use_cond, err, scale)
        assert float((got - want).pow(2).mean().sqrt()) < 2e-5 * scale


@pytest.mark.parametrize("kernel", ["images", "in-kernel split"])
@pytest.mark.parametrize("M,N,K", [(6016, 1152, 384), (4099, 264, 96), (300, 128, 64)])
def test_fp32_class_column_sums_as_partial_rows(ops, M, N, K, kernel, monkeypatch):
    """Round 6: the d(bias) / d(PReLU slope) column sums of a precision-3 data gradient leave the wide
    epilogue as one partial row per 64 output rows (f2g_epilogue.colsum_part_ld, plain stores) and are summed
    by f2g_colsum -- same sums as the atomic path (ragged M, N not a multiple of 128, in place over aux), and
    f2g_gemm refuses the field where no wide epilogue would run."""
    import ctypes as C
    gen = torch.Generator().manual_seed(M + N)
    G, W = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) * 0.05
    aux, alpha = torch.randn(M, N, generator=gen), torch.randn(N, generator=gen) * 0.3
    dp = G.double() @ W.double().t()
    want = dp * torch.where(aux > 0, torch.ones_like(aux), alpha[None].expand_as(aux)).double()
    want_cs, want_csa = want.sum(0), (dp * aux.clamp(max=0).double()).sum(0)
    Wd = torch.nn.Parameter(g(W))
    monkeypatch.setattr(ops, "X6F", 0 if kernel == "images" else 1)
    was = ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS
    res = {}
    try:
        ops.set_gemm_precision("bf16x6")
        ops.X6_MIN_K, ops.X6_MIN_ROWS = 32, 1
        monkeypatch.setattr(ops, "COLSUM_PARTS_MIN_ROWS", 1)
        for parts in (True, False):
            monkeypatch.setattr(ops, "COLSUM_PARTS", parts)
            a = g(aux)
            cs, csa = torch.zeros(N, device=DEV), torch.full((N,), 2.0, device=DEV)     # (csa: accumulates onto 2)
            calls = []
            real_call = ops.call
            monkeypatch.setattr(ops, "call", lambda name, *args: (calls.append(name), real_call(name, *args))[1])
            ops.gemm(ops.mat(g(G)), ops.mat(Wd), a, aux=a, alpha_n=g(alpha), colsum=cs, colsum_alpha=csa)
            monkeypatch.setattr(ops, "call", real_call)
            assert ops.L.lib.f2g_gemm_last_path() == 4
            assert (calls.count("f2g_colsum") == 2) == parts, calls      # (partial rows were used / were not)
            close(a, want, name=f"dgrad parts={parts}")
            close(cs, want_cs, rtol=1e-4, name=f"colsum parts={parts}")
            close(csa - 2.0, want_csa, rtol=1e-4, name=f"colsum_alpha parts={parts}")
            res[parts] = (cs.clone(), csa.clone())
        # only one of the two vectors
        monkeypatch.setattr(ops, "COLSUM_PARTS", True)
        cs1 = torch.zeros(N, device=DEV)
        o1 = torch.empty(M, N, device=DEV)
        ops.gemm(ops.mat(g(G)), ops.mat(Wd), o1, colsum=cs1)
        close(cs1, dp.sum(0), rtol=1e-4, name="colsum alone")
    finally:
        ops.GEMM_PRECISION, ops.X6_MIN_K, ops.X6_MIN_ROWS = was
    # exact fp32 has no wide epilogue: the library refuses partial rows loudly
    d = ops.GemmDesc()
    d.A, d.B = ops.mat(g(G)), ops.mat(g(W))
    o = torch.empty(M, N, device=DEV)
    ws = torch.zeros(2 * ((M + 127) // 128), N, device=DEV)
    e = ops.Epilogue()
    e.C, e.ldc, e.colsum, e.colsum_part_ld = o.data_ptr(), N, ws.data_ptr(), N
    d.E, d.form, d.split_k, d.precision = e, 0, 1, 0
    assert ops.L.lib.f2g_gemm_colsum_part_rows(C.byref(d)) == 0
    assert ops.L.lib.f2g_gemm(C.byref(d), ops.L.stream_ptr()) != 0


def test_batched_rebuild_of_derived_weight_images(ops, monkeypatch):
    """Round 6: after a weight changed, ops.rebuild_derived replays the recipes of the cached copies that were in
    use -- transposes, padded copies, three-piece images of both, window-major conv weights -- with their
    launches collected into f2g_multi tables (one per dependency level).  The rebuilt copies must equal what the
    one-launch-per-copy path builds, bit for bit, and must replace the stale ones in the cache."""
    from flow2gan_amd import fused, fused_disc as fd
    monkeypatch.setattr(ops, "EAGER_REBUILD", True)
    gen = torch.Generator().manual_seed(3)
    w = torch.nn.Parameter(g(torch.randn(96, 64, generator=gen)))          # (N, K)
    wc = torch.nn.Parameter(g(torch.randn(32, 32, 5, 1, generator=gen)))   # conv weight
    wb = torch.nn.Parameter(g(torch.randn(40, generator=gen)))

    def copies():
        t = ops.transposed(w)                                # permute4
        return dict(T=t, Timg=ops.x3_image(t),               # split3 of the transposed copy (second level)
                    img=ops.x3_image(w), padc=fused._pad_cols(w, 128), padr=fused._pad_rows(w, 128),
                    padv=fused._pad_vec(wb, 64), pack=ops.derived(wc, "pack", fd.pack_conv_weight),
                    dg=fd._dgrad_weight(wc, 1, 0, 5))

    first = copies()
    assert all(first[k] is v for k, v in copies().items()), "second call must hit the cache"
    with torch.no_grad():      # an optimizer step written behind autograd's back
        w.data.mul_(1.5)
        wc.data.add_(0.25)
        wb.data.sub_(1.0)
    ops.bump_weight_epoch([w, wc, wb])
    names = []
    real_call = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (names.append(name), real_call(name, *a))[1])
    n = ops.rebuild_derived([w, wc, wb])
    monkeypatch.setattr(ops, "call", real_call)
    assert n >= 8, n
    assert set(names) == {"f2g_multi"} and 2 <= len(names) <= 4, names     # (levels: fill / copies / images)
    batched = copies()
    names.clear()
    assert all(batched[k] is not first[k] for k in first)
    # the same copies built one launch at a time
    monkeypatch.setattr(ops, "EAGER_REBUILD", False)
    ops.bump_weight_epoch([w, wc, wb])
    single = copies()
    torch.cuda.synchronize()
    for k in first:
        assert batched[k] is not single[k]
        a, b = batched[k], single[k]
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a,
                           b.view(torch.int16) if b.dtype == torch.bfloat16 else b), k
    assert torch.equal(batched["T"], w.detach().t().contiguous())
    # a chain that was NOT used since its last rebuild is left to the lazy path
    monkeypatch.setattr(ops, "EAGER_REBUILD", True)
    ops.bump_weight_epoch([w])
    assert ops.rebuild_derived([w]) == 5          # (T, its image, img, padc, padr: all used by `single`)
    ops.transposed(w)                             # the only copy used in this "step"
    ops.bump_weight_epoch([w])
    assert ops.rebuild_derived([w]) == 1


@pytest.mark.parametrize("S,Hp,Cout,nt", [(37, 40, 128, 2), (5, 300, 64, 2), (3, 500, 128, 1)])
def test_fp32_class_gemm_with_32_output_columns(ops, S, Hp, Cout, nt, monkeypatch):
    """Round 6: gemm_x6n_kernel -- the in-kernel-split fp32-class kernel on 128 x 32 tiles, for the data gradients
    that land on a 32-channel map (MPD layer 2's stride residues): windowed A operand over a halo map, cached
    weight image, row-mapped stores (residue interleave) with the leaky-ReLU mask of the layer below and column
    sums, against float64; ragged row counts."""
    monkeypatch.setattr(ops, "X6F_TALL_ROWS", 1000)
    monkeypatch.setattr(ops, "X6_MIN_ROWS", 1)
    gen = torch.Generator().manual_seed(S + Hp)
    Lq = Hp - nt + 1                                        # output positions per sequence
    gmap = torch.randn(S, Hp, Cout, generator=gen)
    w = torch.randn(32, nt * Cout, generator=gen) * 0.05    # [n][k], k = tap-major, channel-minor
    cols = torch.stack([gmap[:, i:i + Lq] for i in range(nt)], 2).reshape(S * Lq, nt * Cout)
    ref = cols.double() @ w.double().t()
    # row map: output row (s, q) -> row 2 + 3 q of a (S, 3 Lq + 4, 32) map (stride-3 residue 2 with a halo of 2... )
    Hin = 3 * Lq + 4
    ymask = torch.randn(S * Hin, 32, generator=gen)
    out = torch.full((S * Hin, 32), 7.0, device=DEV)
    cs = torch.zeros(32, device=DEV)
    wd = torch.nn.Parameter(g(w))
    was = ops.GEMM_PRECISION
    try:
        ops.set_gemm_precision("bf16x6")
        A = ops.win1d(g(gmap.reshape(S * Hp, Cout)), S, Hp, Cout, Lq, 1, 0, nt)
        ops.gemm(A, ops.mat(wd), out, rowmap=(Lq, Hin * 32, 3 * 32, 2 * 32), mask=(g(ymask), 0, 0.1), colsum=cs)
        assert ops.L.lib.f2g_gemm_last_path() == 5, "the 32-column launch did not take gemm_x6n_kernel"
    finally:
        ops.GEMM_PRECISION = was
    o = out.cpu().reshape(S, Hin, 32)
    rows = 2 + 3 * torch.arange(Lq)
    m = ymask.reshape(S, Hin, 32)[:, rows]
    want = ref.reshape(S, Lq, 32) * torch.where(m > 0, 1.0, 0.1).double()
    close(o[:, rows].double(), want, rtol=2e-5, name="x6n rowmapped masked output")
    untouched = torch.ones(Hin, dtype=torch.bool)
    untouched[rows] = False
    assert bool((o[:, untouched] == 7.0).all())
    close(cs.cpu().double(), want.sum((0, 1)), rtol=1e-4, name="x6n column sums")


@pytest.mark.parametrize("M,K,N", [(1000, 64, 128), (777, 384, 160), (2050, 32, 384), (128, 2304, 96)])
def test_fp32_class_gemm_with_fragment_major_weights(ops, M, K, N, monkeypatch):
    """Round 6: gemm_x6g_kernel -- the in-kernel-split fp32-class kernel whose WEIGHT fragments come straight from
    the cached fragment-major image (f2g_operand.split = 4: no LDS staging of the weights): plain and ragged row
    counts, one and many slabs, a ragged last column tile (N = 160, 96), bias + residual epilogue and the plain
    one, against float64 at the fp32-class bound; the same launch over the row-major image (x6g off) must give
    the SAME bits (same products in the same order)."""
    monkeypatch.setattr(ops, "X6_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "X6F", 1)              # every eligible launch on the in-kernel-split kernels
    gen = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) * 0.05
    b = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen)
    ref = a.double() @ w.double().t()
    wd = torch.nn.Parameter(g(w))
    was = ops.GEMM_PRECISION
    outs = {}
    try:
        ops.set_gemm_precision("bf16x6")
        for frag in (True, False):
            monkeypatch.setattr(ops, "X6G", frag)
            o1 = torch.empty(M, N, device=DEV)
            o2 = torch.empty(M, N, device=DEV)
            ops.gemm(ops.mat(g(a)), ops.mat(wd), o1)
            assert ops.L.lib.f2g_gemm_last_path() == 4
            ops.gemm(ops.mat(g(a)), ops.mat(wd), o2, bias=g(b), res=g(res))
            outs[frag] = (o1.cpu(), o2.cpu())
    finally:
        ops.GEMM_PRECISION = was
    scale = float(ref.abs().max())
    assert float((outs[True][0].double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, (K / 256) ** 0.5)
    close(outs[True][1].double(), ref + b.double() + res.double(), rtol=2e-5, name="x6g bias + residual")
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1]), \
        "fragment-major and row-major weight images gave different bits"
