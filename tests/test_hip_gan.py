"""GPU parity of the GAN-stage loss stack: discriminator scores / feature maps against the CPU
oracle, and D-step / G-step losses + gradients against the reference's golden vectors."""
import hashlib
import random

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]
DEV = "cuda"

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)


@pytest.fixture(scope="module")
def f2g():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flow2gan_amd
    return flow2gan_amd


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).abs().max()) / (float(want.abs().max()) + 1e-12)


TINY44 = dict(TINY, sampling_rate=44100, n_mels=128, mel_n_fft=2048, mel_hop_length=512,
              n_ffts=(1024, 512, 256), hop_lengths=(512, 256, 128), loss_n_fft=2048,
              loss_hop_length=512)


def build_gan(f2g, g, cfg=TINY):
    from flow2gan_amd.models.gan import GAN
    gen = f2g.MelAudioGenerator(**cfg)
    gen.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w/")})
    gen.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    gan = GAN(gen)
    sd = {k: v for k, v in gan.discriminator.state_dict().items() if "spec_fn" not in k}
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().astype(np.float32).tobytes())
    assert h.hexdigest() == bytes(g["d_digest"]).decode(), "discriminator init differs from reference"
    return gan.to(DEV)


def test_discriminator_scores_and_fmaps_vs_oracle(f2g):
    import flow2gan_oracle as O
    from flow2gan_amd.models.discriminators import (MultiPeriodDiscriminator,
                                                    MultiResolutionDiscriminator)
    torch.manual_seed(5)
    x = 0.1 * torch.randn(2, 6001)
    x[1] *= 3.0
    for Oc, Hc in ((O.MultiPeriodDiscriminator, MultiPeriodDiscriminator),
                   (O.MultiResolutionDiscriminator, MultiResolutionDiscriminator)):
        torch.manual_seed(9)
        do = Oc()
        dh = Hc()
        dh.load_state_dict(do.state_dict(), strict=False)
        dh = dh.to(DEV)
        with torch.no_grad():
            sr_o, _, fr_o, _ = do(x, x)
        sr_h, _, fr_h, _ = dh(x.to(DEV), x.to(DEV))
        for i, (a, b) in enumerate(zip(sr_h, sr_o)):
            assert relerr(a.reshape(b.shape), b) < 5e-4, (Oc.__name__, "score", i)
        for i, (fa, fb) in enumerate(zip(fr_h, fr_o)):
            assert len(fa) == len(fb)
            for j, (a, b) in enumerate(zip(fa, fb)):
                assert a.shape == b.shape, (Oc.__name__, i, j, a.shape, b.shape)
                assert relerr(a, b) < 5e-4, (Oc.__name__, "fmap", i, j, relerr(a, b))


def build_oracle_gan(g, cfg):
    import flow2gan_oracle as O
    og = O.MelAudioGenerator(**cfg)
    og.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w/")})
    og.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    return O.GAN(og)


def leaky_relu_sign_flips(gan, ogan, real, fake_h, fake_o):
    """Leaky-ReLU is not differentiable at 0.  Where the reference's pre-activation is smaller
    than the fp32 disagreement between the two forward passes (|x| ~ 1e-9 against ~4e-8: measured
    with tools/dbg/flip2.py; the step itself also reorders sums through split-K atomics), the HIP
    path and the CPU reference can land on different sides of the kink, and the gradient through
    that pixel differs by the factor 10 between the two slopes -- a property of the function, not
    an error of either side.  This finds those pixels and returns the parameter prefixes whose
    gradient passes through one (that layer and every layer before it in the same stack),
    separately for the real and the generated input, plus the pixel counts."""
    tainted = {"real": set(), "fake": set(), "flipped": 0, "pixels": 0}
    for di in (0, 1):
        with torch.no_grad():
            _, _, fr_o, ff_o = ogan.discriminator[di](real.cpu(), fake_o.cpu())
            _, _, fr_h, ff_h = gan.discriminator[di](real, fake_h)
        for which, fo, fh in (("real", fr_o, fr_h), ("fake", ff_o, ff_h)):
            for i, (maps_h, maps_o) in enumerate(zip(fh, fo)):
                # the last map is conv_post's output, which has no activation
                for j, (a, b) in enumerate(zip(maps_h[:-1], maps_o[:-1])):
                    a = a.detach().cpu()
                    pre = torch.where(b > 0, b, b / 0.1)          # undo the activation
                    noise = float((a - b).abs().max())
                    near = (pre.abs() <= 4.0 * noise) | ((a > 0) != (b > 0))
                    nflip = int(near.sum())
                    tainted["flipped"] += nflip
                    tainted["pixels"] += b.numel()
                    if nflip == 0:
                        continue
                    if di == 0:     # MPD maps are layers 1..4 (discriminators.py:95-96)
                        tainted[which] |= {f"0.discriminators.{i}.convs.{l}." for l in range(j + 2)}
                    else:           # MRD maps: per band layers 1..4 (discriminators.py:206-207)
                        band, layer = j // 4, j % 4 + 1
                        tainted[which] |= {f"1.discriminators.{i}.band_convs.{band}.{l}."
                                           for l in range(layer + 1)}
    return tainted


class _KinkResolved:
    """The oracle's discriminators with the SIDE of every leaky-ReLU taken from the HIP path's own activations.
    Leaky-ReLU has no derivative at 0; a handful of pre-activations (|x| ~ 1e-9 against a forward disagreement
    of ~4e-8) land on different sides in the two implementations, and the gradient through such a pixel differs
    by the factor 10 between the slopes.  Re-running the oracle with the HIP path's sign pattern (forward
    values change by < 1e-8) removes exactly that ambiguity: against THESE gradients every parameter -- the
    ones downstream of a kink pixel included -- is held to the plain 5e-3 (round-5 verdict, weak 1a).
    Test infrastructure: patches `leaky_relu` inside oracle/flow2gan_oracle.py for the duration of a call."""

    def __init__(self, gan, real, fake_h):
        from flow2gan_amd import fused_disc as FD
        B = real.shape[0]
        x2 = torch.cat([real, fake_h.detach()], 0).contiguous()
        self.masks = {}          # (disc, sub, half) -> list of bool masks in the oracle's call order
        mp, mr = gan.discriminator
        with torch.no_grad():
            prm = FD.mpd_params(mp)
            for i, p in enumerate(mp.periods):
                st = FD._mpd_forward_one(x2, p, prm[12 * i: 12 * i + 12])
                ms = []
                for l in range(1, 6):
                    a = FD.unhalo(st["acts"][l], 2 * B * p, st["hs"][l])
                    ms.append((a.reshape(2 * B, p, st["hs"][l], a.shape[-1]).permute(0, 3, 2, 1) > 0).cpu())
                for h, sl in (("real", slice(0, B)), ("fake", slice(B, 2 * B))):
                    self.masks[(0, i, h)] = [m[sl] for m in ms]
            prm = FD.mrd_params(mr)
            for i, win in enumerate(mr.fft_sizes):
                st = FD._mrd_forward_one(x2, win, prm[FD.N_MRD_PARAMS * i: FD.N_MRD_PARAMS * (i + 1)])
                Ft, Wcat, C = st["Ft"], st["Wcat"], FD.MRD_CH
                cat = st["cat"].view(2 * B, Ft, Wcat, C)
                ms, foff = [], 0
                for bi in range(5):
                    ws = st["widths"][bi]
                    for l in range(4):
                        ms.append((st["acts"][bi][l].view(2 * B, Ft, ws[l + 1], C).permute(0, 3, 1, 2) > 0).cpu())
                    ms.append((cat[:, :, foff:foff + ws[5]].permute(0, 3, 1, 2) > 0).cpu())
                    foff += ws[5]
                for h, sl in (("real", slice(0, B)), ("fake", slice(B, 2 * B))):
                    self.masks[(1, i, h)] = [m[sl] for m in ms]

    def run(self, ogan, fn):
        """fn() runs the oracle GAN; inside it every discriminator call consumes the recorded masks (the
        oracle runs sub-discriminator i on the real batch, then on the generated one: _MultiD.forward)."""
        import collections
        import types
        import flow2gan_oracle as O
        queue = collections.deque()
        for d in (0, 1):
            nsub = len(ogan.discriminator[d].discriminators)
            for i in range(nsub):
                for h in ("real", "fake"):
                    queue.extend(self.masks[(d, i, h)])

        def leaky(x, slope):
            m = queue.popleft()
            assert m.shape == x.shape, (m.shape, x.shape)
            return torch.where(m, x, slope * x)

        proxy = types.SimpleNamespace(**{k: getattr(O.F, k) for k in dir(O.F) if not k.startswith("__")})
        proxy.leaky_relu = leaky
        real_F, O.F = O.F, proxy
        try:
            out = fn()
        finally:
            O.F = real_F
        assert not queue, len(queue)
        return out


@pytest.mark.parametrize("fixture,cfg", [("tiny_stage2", TINY), ("tiny_stage2_44k", TINY44)],
                         ids=["24k", "44k"])
@pytest.mark.parametrize("tag,n", [("n1", 1), ("n2", 2)])
@pytest.mark.parametrize("pingpong", [False, True], ids=["rule", "pingpong"])
def test_gan_steps_against_reference_vectors(f2g, golden, tag, n, fixture, cfg, pingpong, gemm_mode, monkeypatch,
                                             lib_option):
    """D-step / G-step losses and gradients against the REFERENCE's recorded vectors; the 44k
    fixture is BASELINE config 5's geometry (sr 44100 in the seven mel-recon filterbanks and the
    128-band / n_fft 2048 / hop 512 front end, config.py:64-95, gan.py:44-55).
    pingpong (bf16x6 only): the round-5 six-product kernels (gemm_x6p.hip: ping-pong wave groups on 256-row
    tiles) forced onto every launch their geometry allows, whatever the grid size -- the library's rule gives
    them chip-filling grids only, which these small cases never are."""
    if pingpong:
        if gemm_mode not in ("bf16x6", "3"):
            pytest.skip("the ping-pong kernels are six-product kernels")
        lib_option("x6p", 2)
    g = golden(fixture)
    gan = build_gan(f2g, g, cfg)
    assert gan.generator.sampling_rate == cfg["sampling_rate"]
    monkeypatch.setattr(random, "random", lambda: 0.0)
    mel, audio, noise = T(g["mel"]).to(DEV), T(g["audio"]).to(DEV), T(g["noise"]).to(DEV)
    lens = T(g[f"{tag}/lens"])
    # pixels where the two sides sit on different sides of a leaky-ReLU kink (see the helper)
    ogan = build_oracle_gan(g, cfg)
    with torch.no_grad():
        gan.generator.eval(), ogan.generator.eval()
        fake_h = gan.generator.infer(mel, lens, n, noise=noise)
        fake_o = ogan.generator.infer(mel.cpu(), lens, n, noise=noise.cpu())
    flips = leaky_relu_sign_flips(gan, ogan, audio, fake_h, fake_o)
    d_flipped = flips["real"] | flips["fake"]
    # isolated pixels, not a broken layer (their number grows with the forward perturbation: the
    # opt-in split-bf16 GEMM mode carries ~2^-16 per product instead of 2^-24)
    from flow2gan_amd import ops as _ops0
    kink_share = 1e-3 if _ops0.GEMM_PRECISION == 1 else 1e-4
    assert flips["flipped"] <= kink_share * flips["pixels"] + 8, (flips["flipped"], flips["pixels"])
    print("pixels on a leaky-ReLU kink:", flips["flipped"], "of", flips["pixels"])

    def d_tol(k):
        return 5e-2 if any(k.startswith(pfx) for pfx in d_flipped) else 5e-3

    # ---- discriminator step
    d = gan(mel, audio, lens, n, True, noise=noise)
    want = g[f"{tag}/D/losses"]
    assert np.allclose([float(v.detach()) for v in d], want, rtol=2e-5, atol=2e-5), ([float(v.detach()) for v in d], want)
    gan.zero_grad()
    (1.0 * d[0] + 0.1 * d[1]).backward()
    worst = []
    for k, p in gan.discriminator.named_parameters():
        assert p.grad is not None, k
        st = g[f"{tag}/D/gstat/{k}"]
        got_abs = float(p.grad.double().abs().sum())
        worst.append((abs(got_abs - st[1]) / (st[1] + 1e-3) / d_tol(k), k))
        key = f"{tag}/D/g/{k}"
        if key in g:
            ref = T(g[key])
            err = float((p.grad.cpu().double() - ref.double()).abs().max())
            # exactly-zero reference entries (cancelling hinge terms) get an absolute floor
            from flow2gan_amd import ops as _ops
            # split-bf16 GEMM mode (F2G_GEMM=bf16x3): ~2^-16 per product instead of 2^-24
            floor = 2e-4 if _ops.GEMM_PRECISION == 1 else 2e-5
            assert err < d_tol(k) * float(ref.abs().max()) + floor, (k, err)
    worst.sort(reverse=True)
    assert worst[0][0] < 1.0, worst[:5]
    for p in gan.generator.parameters():
        assert p.grad is None
    # the parameters whose gradient passes through a kink pixel: against the oracle with the HIP path's
    # leaky-ReLU sides they are held to the plain 5e-3 as well (exact fp32 and fp32-class arithmetic)
    resolved = None
    # (two more CPU passes of the oracle, ~20 s: the 24 kHz n = 1 case of the two parity arithmetics)
    if (d_flipped or flips["fake"]) and _ops0.GEMM_PRECISION != 1 and not pingpong and n == 1 and fixture == "tiny_stage2":
        resolved = _KinkResolved(gan, audio, fake_h)
        ogan.discriminator.load_state_dict({k: v.cpu() for k, v in gan.discriminator.state_dict().items()},
                                           strict=False)
    if resolved is not None and d_flipped:
        def oracle_d():
            ogan.zero_grad()
            omp, omr = ogan(mel.cpu(), audio.cpu(), lens, n, True, noise=noise.cpu())
            (1.0 * omp + 0.1 * omr).backward()
            return {k: p.grad.clone() for k, p in ogan.discriminator.named_parameters()}
        og = resolved.run(ogan, oracle_d)
        kinked = [k for k, _ in gan.discriminator.named_parameters() if d_tol(k) > 5e-3]
        assert kinked
        for k, p in gan.discriminator.named_parameters():
            if k in kinked:
                err = float((p.grad.cpu().double() - og[k].double()).abs().max())
                assert err < 5e-3 * float(og[k].abs().max()) + 2e-5, ("kink-resolved", k, err)
    # ---- generator step
    gan.zero_grad()
    ls = gan(mel, audio, lens, n, False, noise=noise)
    want = g[f"{tag}/G/losses"]
    assert np.allclose([float(v.detach()) for v in ls], want, rtol=5e-5, atol=2e-5), ([float(v.detach()) for v in ls], want)
    total = sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls))
    total.backward()
    from flow2gan_amd import ops as _ops
    worst, refs = [], {}
    for k, p in gan.generator.named_parameters():
        ref = T(g[f"{tag}/G/g/{k}"])
        assert p.grad is not None, k
        refs[k] = ref
        worst.append((relerr(p.grad, ref), k))
    worst.sort(reverse=True)
    if _ops.GEMM_PRECISION != 1:
        # exact-fp32 and fp32-class (bf16x6) GEMMs: 5e-3 of each gradient's max (5e-2 when a pixel of
        # the generated input's path sits on a leaky-ReLU kink: it reaches every generator gradient)
        g_tol = 5e-2 if flips["fake"] else 5e-3
        assert worst[0][0] < g_tol, (worst[:8], sorted(flips["fake"]))
        if resolved is not None and flips["fake"]:
            # ... and against the kink-resolved oracle every generator gradient to the plain 5e-3
            def oracle_g():
                ogan.zero_grad()
                ols = ogan(mel.cpu(), audio.cpu(), lens, n, False, noise=noise.cpu())
                sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ols)).backward()
                return {k: p.grad.clone() for k, p in ogan.generator.named_parameters()}
            og = resolved.run(ogan, oracle_g)
            wr = sorted(((relerr(p.grad, og[k]), k) for k, p in gan.generator.named_parameters()), reverse=True)
            assert wr[0][0] < 5e-3, ("kink-resolved", wr[:6])
        return
    # The opt-in split-bf16 mode perturbs the generated waveform by ~1e-5, which flips sign() terms of
    # the L1 / hinge / leaky-ReLU gradients (they are discontinuous).  What a flipped pixel adds to a
    # parameter's gradient is small against the gradients of that KIND of parameter, but it can be
    # large against a gradient that is itself the small remainder of cancelling terms (a BiasNorm
    # log_scale scalar: 0.31 of its own value in the 44.1 kHz n = 2 case, 0.02 of the largest
    # log_scale gradient; tools/dbg/g_grad_split.py, deterministic from run to run).  So the bound is
    # kink-aware in what it normalises by -- every tensor to 0.15 of the largest gradient among the
    # tensors of its kind (measured worst: 0.083; 0.005 in the three other cases), all generator
    # gradients together to 0.12 in relative L2 (measured 0.066 / 2e-3 / 6e-4 / 5e-5), and as a
    # backstop each tensor to 0.5 of its own maximum: a sign error or a broken layer on a tensor
    # that matters fails the first two, on any tensor the third.
    def kind(k):
        return k.split(".")[-2] + ".scale" if k.endswith("scale") else k.split(".")[-1]
    kind_max = {}
    for k, ref in refs.items():
        kind_max[kind(k)] = max(kind_max.get(kind(k), 0.0), float(ref.abs().max()))
    err2 = ref2 = 0.0
    by_kind = []
    for k, p in gan.generator.named_parameters():
        dlt = p.grad.detach().cpu().double() - refs[k].double()
        err2 += float(dlt.pow(2).sum())
        ref2 += float(refs[k].double().pow(2).sum())
        by_kind.append((float(dlt.abs().max()) / (kind_max[kind(k)] + 1e-12), k))
    by_kind.sort(reverse=True)
    assert by_kind[0][0] < 0.15, (by_kind[:8], sorted(flips["fake"]))
    assert (err2 / ref2) ** 0.5 < 0.12, ((err2 / ref2) ** 0.5, worst[:8])
    assert worst[0][0] < 0.5, (worst[:8], sorted(flips["fake"]))


def test_concurrent_lanes_match_serial_launch_order(f2g, golden, monkeypatch):
    """The launch lanes (one HIP stream per Fourier branch / sub-discriminator / mel scale) must
    give the serial schedule's losses and gradients: same kernels, only atomics may reorder."""
    from flow2gan_amd import ops
    g = golden("tiny_stage2")
    gan = build_gan(f2g, g)
    monkeypatch.setattr(random, "random", lambda: 0.0)
    mel, audio, noise = T(g["mel"]).to(DEV), T(g["audio"]).to(DEV), T(g["noise"]).to(DEV)
    lens = T(g["n2/lens"])

    def run(concurrent):
        monkeypatch.setattr(ops, "CONCURRENT", concurrent)
        out = {}
        gan.zero_grad()
        d = gan(mel, audio, lens, 2, True, noise=noise)
        (d[0] + 0.1 * d[1]).backward()
        out["D"] = [float(v.detach()) for v in d]
        out["gD"] = {k: p.grad.clone() for k, p in gan.discriminator.named_parameters()}
        gan.zero_grad()
        ls = gan(mel, audio, lens, 2, False, noise=noise)
        sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls)).backward()
        out["G"] = [float(v.detach()) for v in ls]
        out["gG"] = {k: p.grad.clone() for k, p in gan.generator.named_parameters()}
        torch.cuda.synchronize()
        return out

    ref = run(False)
    for _ in range(3):  # a missing dependency between lanes would show up as a flaky mismatch
        got = run(True)
        assert np.allclose(got["D"], ref["D"], rtol=1e-5, atol=1e-6)
        assert np.allclose(got["G"], ref["G"], rtol=1e-5, atol=1e-6)
        for key in ("gD", "gG"):
            for k, v in ref[key].items():
                assert relerr(got[key][k], v) < 2e-4, (key, k, relerr(got[key][k], v))


_ORACLE_RUNS: dict = {}
_ORACLE_WEIGHTS: dict = {}


@pytest.mark.parametrize("model_name,Tn,rep,nts", [("mel_24k_base", 24000, 32, 1),
                                                   ("mel_44k_128band_512x_base", 44100, 16, 1),
                                                   ("mel_24k_base", 24000, 0, 4)],
                         ids=["24k_B64_T24000", "44k_B32_T44100", "24k_B2_T24000_n4"])
def test_full_width_gan_steps_vs_oracle_then_full_batch(f2g, monkeypatch, model_name, Tn, rep, nts):
    """Full-width GAN stage: D-step and G-step losses + selected gradients against the CPU oracle
    at B=2, then at the BASELINE batch (B=64 x 1 s for mel_24k_base = config 4's per-GPU shape;
    B=32 x 1 s of 44.1 kHz audio for mel_44k_128band_512x_base = config 5) as copies of that batch: every loss is a batch mean and
    every sample is independent, so losses and gradients must not move.  The third case unrolls
    n_timesteps = 4 Euler steps inside both steps (gan.py:138-143: the G-step backpropagates through
    all four model evaluations; SURVEY 8d config 4 names n in {1, 4}) at B = 2 against the oracle."""
    import flow2gan_oracle as O
    from flow2gan_amd.models.config import get_generator_config
    from flow2gan_amd.models.gan import GAN
    monkeypatch.setattr(random, "random", lambda: 1.0)   # LimitParamValue off on both sides
    cfg = get_generator_config(model_name)
    d_names = ["0.discriminators.0.convs.4.weight", "0.discriminators.3.conv_post.weight",
               "1.discriminators.1.band_convs.2.1.weight", "1.discriminators.2.conv_post.bias"]
    g_names = ["cond_encoder.in_proj.weight", "estimators.0.decoder.blocks.7.pwconv2.weight",
               "estimators.2.decoder.out_proj.weight"]
    key = (model_name, Tn, nts)
    if key not in _ORACLE_RUNS:
        # the CPU oracle's side does not depend on the GEMM mode of the HIP side: computed once per case
        # (30-50 s of the test at n = 4) and shared by the three modes this module runs in
        torch.manual_seed(31)
        og = O.build_generator(model_name)
        og.branch_dropout = 0.0
        ogan = O.GAN(og)
        rg = torch.Generator().manual_seed(8)
        audio = (0.1 * torch.randn(2, Tn, generator=rg)).clamp(-1, 1)
        audio[1] *= 2.5
        lens = torch.tensor([Tn, Tn])
        noise = 0.1 * torch.randn(2, Tn, generator=rg)
        mel = O.LogMelSpectrogram(cfg["sampling_rate"], cfg["mel_n_fft"], cfg["mel_hop_length"],
                                  cfg["n_mels"])(audio)
        ogan.zero_grad()
        od = ogan(mel, audio, lens, nts, True, noise=noise)
        (od[0] + 0.1 * od[1]).backward()
        od_g = {k: dict(ogan.discriminator.named_parameters())[k].grad.clone() for k in d_names}
        ogan.zero_grad()
        ogl = ogan(mel, audio, lens, nts, False, noise=noise)
        sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ogl)).backward()
        og_g = {k: dict(ogan.generator.named_parameters())[k].grad.clone() for k in g_names}
        # (the cases run mode-major, so all three stay cached: ~0.5 GB of host memory per model)
        if model_name not in _ORACLE_WEIGHTS:
            _ORACLE_WEIGHTS[model_name] = {k: v.detach().clone() for k, v in ogan.state_dict().items()}
        _ORACLE_RUNS[key] = dict(sd=_ORACLE_WEIGHTS[model_name],
                                 audio=audio, lens=lens, noise=noise, mel=mel,
                                 od=[float(v) for v in od], od_g=od_g, ogl=[float(v) for v in ogl], og_g=og_g)
    R = _ORACLE_RUNS[key]
    audio, lens, noise, mel = R["audio"], R["lens"], R["noise"], R["mel"]
    od, od_g, ogl, og_g = R["od"], R["od_g"], R["ogl"], R["og_g"]
    gen = f2g.MelAudioGenerator(**cfg)
    gen.branch_dropout = 0.0
    gan = GAN(gen)
    missing = gan.load_state_dict(R["sd"], strict=False)
    assert not [k for k in missing.missing_keys if "window" not in k and "fb" not in k], missing
    gan = gan.to(DEV)

    def run(rep):
        a, m_, n_, ln = (audio.to(DEV).repeat(rep, 1), mel.to(DEV).repeat(rep, 1, 1),
                         noise.to(DEV).repeat(rep, 1), lens.repeat(rep))
        gan.zero_grad()
        d = gan(m_, a, ln, nts, True, noise=n_)
        (d[0] + 0.1 * d[1]).backward()
        dg = {k: dict(gan.discriminator.named_parameters())[k].grad.detach().cpu().clone() for k in d_names}
        gan.zero_grad()
        ls = gan(m_, a, ln, nts, False, noise=n_)
        sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls)).backward()
        gg = {k: dict(gan.generator.named_parameters())[k].grad.detach().cpu().clone() for k in g_names}
        return [float(v.detach()) for v in d], [float(v.detach()) for v in ls], dg, gg

    d2, l2, dg2, gg2 = run(1)
    assert np.allclose(d2, od, rtol=1e-4, atol=1e-5), (d2, od)
    assert np.allclose(l2, ogl, rtol=2e-4, atol=1e-5), (l2, ogl)
    from flow2gan_amd import ops as _ops
    gtol = 1e-1 if _ops.GEMM_PRECISION == 1 else 1e-2
    def near(got, want, tol):   # exactly-zero references (cancelling hinge terms): absolute floor
        err = float((got.detach().cpu().double() - want.detach().cpu().double()).abs().max())
        return err < tol * float(want.abs().max()) + 1e-7

    for k in d_names:
        assert near(dg2[k], od_g[k], gtol), ("D", k, relerr(dg2[k], od_g[k]))
    for k in g_names:
        assert near(gg2[k], og_g[k], gtol), ("G", k, relerr(gg2[k], og_g[k]))
    if not rep:
        return
    d64, l64, dg64, gg64 = run(rep)   # the BASELINE batch
    assert np.allclose(d64, d2, rtol=2e-5, atol=1e-6), (d64, d2)
    assert np.allclose(l64, l2, rtol=5e-5, atol=1e-6), (l64, l2)
    btol = 2e-2 if _ops.GEMM_PRECISION == 1 else 5e-3   # split-bf16: more pixels on a kink
    for k in d_names:
        assert near(dg64[k], dg2[k], btol), ("D64", k, relerr(dg64[k], dg2[k]))
    for k in g_names:
        assert near(gg64[k], gg2[k], btol), ("G64", k, relerr(gg64[k], gg2[k]))
