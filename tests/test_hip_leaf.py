"""The drop-in boundary at LEAF level (north_star: "flow2gan.models.* keep their signatures"): every
leaf module of `flow2gan_amd.models.modules` is CALLED the way the reference calls its own
(modules.py:52-84, 87-116, 146-232, 273-283, 419-721) and checked against the reference's recorded
per-leaf vectors (`tiny_forward.npz`, written by oracle/make_golden.py from the real reference) at the
leaf tolerances of test_tiny_leafs_against_reference_vectors; gradients of the same module calls against
the oracle's autograd; the GAN's three loss methods (gan.py:57-87) on oracle score / feature maps."""
import random

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode_exact")]

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)
DEV = "cuda"
LEAF_TOL = 2e-5     # rms, as in tests/test_hip_generator.py::test_tiny_leafs_against_reference_vectors
GRAD_TOL = 2e-3     # of the gradient's largest element (stage-1 gradient bound)


@pytest.fixture(scope="module")
def f2g():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flow2gan_amd
    return flow2gan_amd


def T(a):
    return torch.from_numpy(np.asarray(a))


def rms(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).pow(2).mean().sqrt())


def relerr(got, want):
    want = want.detach().double()
    return float((got.detach().cpu().double() - want).abs().max()) / (float(want.abs().max()) + 1e-20)


def pair(f2g, g):
    import flow2gan_oracle as O
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    mh = f2g.MelAudioGenerator(**TINY)
    mh.load_state_dict(sd)
    mo = O.MelAudioGenerator(**TINY)
    mo.load_state_dict(sd)
    return mh.to(DEV), mo


def test_leaf_module_calls_against_reference_vectors(f2g, golden):
    import flow2gan_oracle as O
    g = golden("tiny_forward")
    mh, mo = pair(f2g, g)
    mh.eval(), mo.eval()
    mel, noise, lens = T(g["mel"]), T(g["noise"]), T(g["lens"])
    tt = torch.full((2,), 0.25)
    cond_ref = T(g["cond_enc"])
    with torch.no_grad():
        # CondEncoder(mel)
        y = mh.cond_encoder(mel.to(DEV))
        assert y.shape == cond_ref.shape
        assert rms(y, cond_ref) < LEAF_TOL
        for i, (eh, eo) in enumerate(zip(mh.estimators, mo.estimators)):
            # STFT(audio, audio_lens) -> (complex spectrum, frame counts)
            spec, spec_lens = eh.fft(noise.to(DEV), lens.to(DEV))
            packed_ref = T(g[f"br{i}/stft_packed"])
            assert spec.is_complex() and spec.shape == (2, eh.n_fft // 2 + 1, packed_ref.shape[2])
            assert [int(v) for v in spec_lens] == [1 + int(l) // eh.hop_length for l in lens]   # bit-exact indexing
            assert rms(O.pack_complex(spec.cpu()), packed_ref) < LEAF_TOL
            assert eh.fft(noise.to(DEV))[1] is None
            # ISTFT(spec) against the oracle's torch.istft of the SAME spectrum
            spec_o = O.unpack_complex(packed_ref)
            ya = eh.ifft(spec_o.to(DEV))
            want = eo.ifft(spec_o)
            assert ya.shape == want.shape and rms(ya, want) < LEAF_TOL
            # upsample_cond: data movement, exact
            Fr = packed_ref.shape[2]
            cu = eo.upsample_cond(cond_ref, Fr)
            assert torch.equal(eh.upsample_cond(cond_ref.to(DEV), Fr).cpu(), cu)
            mask = O.pad_mask(1 + lens // eh.hop_length).logical_not().unsqueeze(1)
            # SinusoidalPosEmb(t)
            emb = eh.decoder.time_embed(tt.to(DEV))
            assert rms(emb, O.sinusoid_embedding(tt, TINY["time_embed_channels"])) < LEAF_TOL
            # BiasNorm(x) (in_norm) on the reference's in_proj output
            x_in = eo.decoder.in_proj(packed_ref)
            x0_ref = T(g[f"br{i}/in_norm"])
            assert rms(eh.decoder.in_norm(x_in.to(DEV)), x0_ref) < LEAF_TOL
            # ConvNeXtBlock(x, cond=, time_embed=, mask=)
            te_ref = T(g[f"br{i}/time_embed"])
            cm = eo.decoder.cond_mlp(cu)
            yb = eh.decoder.blocks[0](x0_ref.to(DEV), cond=cm.to(DEV), time_embed=te_ref.to(DEV),
                                      mask=mask.to(DEV))
            assert rms(yb, T(g[f"br{i}/block0"])) < LEAF_TOL
            # ChannelScale(x)
            cs = eh.decoder.blocks[0].residual_scale
            assert rms(cs(x0_ref.to(DEV)), x0_ref * cs.scale.detach().cpu()) < 1e-7
            # ConvNeXtDecoder(x, cond=, t=, mask=)
            yd = eh.decoder(packed_ref.to(DEV), cond=cu.to(DEV), t=tt.to(DEV), mask=mask.to(DEV))
            assert rms(yd, T(g[f"br{i}/decoder_out"])) < LEAF_TOL
            # AudioConvNeXt(audio, cond, t, audio_lens)
            yw = eh(noise.to(DEV), cond_ref.to(DEV), tt.to(DEV), lens.to(DEV))
            assert rms(yw, T(g[f"br{i}/audio"])) < LEAF_TOL
        # LinearFilterSpectrogram(waveform) (the stage-1 loss's spectrogram) and the mel stand-in
        audio = T(g["audio"])
        S = mh.loss_spec(audio.to(DEV))
        So = mo.loss_spec(audio)
        assert S.shape == So.shape and relerr(S, So) < 2e-5
        assert relerr(mh.loss_spec(audio[0].to(DEV)), So[0]) < 2e-5          # (..., time) inputs
    with pytest.raises(ValueError):      # a mask that is no padding mask is refused, not approximated
        bad = torch.ones(2, 1, 24, dtype=torch.bool)
        bad[0, 0, 3] = False
        mh.cond_encoder(mel.to(DEV), mask=bad.to(DEV))


def _grads(out, weight, inputs, module):
    (out * weight).sum().backward()
    gi = [None if (x is None or x.grad is None) else x.grad.detach().cpu().clone() for x in inputs]
    gp = {n: p.grad.detach().cpu().clone() for n, p in module.named_parameters() if p.grad is not None}
    module.zero_grad(set_to_none=True)
    return gi, gp


def _check_module_grads(hmod, omod, args_cpu, kwargs_cpu=None, seed=0, takes=lambda y: y):
    """Same call on the HIP module and the oracle module (train mode, same Python RNG draws for the
    LimitParamValue coin flips), loss = <out, fixed weights>; input and parameter gradients compared."""
    kwargs_cpu = kwargs_cpu or {}
    hmod.train(), omod.train()

    def leafs(xs, dev):
        out = []
        for x in xs:
            if torch.is_tensor(x) and x.is_floating_point() and x.dim() >= 2:   # (t needs no gradient)
                out.append(x.clone().to(dev).requires_grad_(True))
            elif torch.is_tensor(x):
                out.append(x.to(dev))
            else:
                out.append(x)
        return out

    ah, ao = leafs(args_cpu, DEV), leafs(args_cpu, "cpu")
    kh = dict(zip(kwargs_cpu, leafs(kwargs_cpu.values(), DEV)))
    ko = dict(zip(kwargs_cpu, leafs(kwargs_cpu.values(), "cpu")))
    random.seed(seed)
    yo = takes(omod(*ao, **ko))
    w = torch.randn(yo.shape, generator=torch.Generator().manual_seed(3))
    gio, gpo = _grads(yo, w, ao + list(ko.values()), omod)
    random.seed(seed)
    yh = takes(hmod(*ah, **kh))
    assert rms(yh, yo) < 5e-5
    gih, gph = _grads(yh, w.to(DEV), ah + list(kh.values()), hmod)
    for a, b in zip(gih, gio):
        if b is not None and torch.is_tensor(b):
            assert a is not None, "input gradient missing"
            assert relerr(a, b) < GRAD_TOL
    assert set(gpo) <= set(gph), sorted(set(gpo) - set(gph))
    worst = max(((relerr(gph[n], gpo[n]), n) for n in gpo), default=(0.0, ""))
    assert worst[0] < GRAD_TOL, worst
    return worst


def test_leaf_module_gradients_against_oracle(f2g, golden):
    import flow2gan_oracle as O
    g = golden("tiny_forward")
    mh, mo = pair(f2g, g)
    mel, noise, lens = T(g["mel"]), T(g["noise"]), T(g["lens"])
    tt = torch.full((2,), 0.25)
    cond_ref = T(g["cond_enc"])
    for seed in (0, 1):      # different LimitParamValue draws
        _check_module_grads(mh.cond_encoder, mo.cond_encoder, [mel], seed=seed)
        for i, (eh, eo) in enumerate(zip(mh.estimators, mo.estimators)):
            packed = T(g[f"br{i}/stft_packed"])
            Fr = packed.shape[2]
            cu = eo.upsample_cond(cond_ref, Fr)
            mask = O.pad_mask(1 + lens // eh.hop_length).logical_not().unsqueeze(1)
            x0, te = T(g[f"br{i}/in_norm"]), T(g[f"br{i}/time_embed"])
            with torch.no_grad():
                cm = eo.decoder.cond_mlp(cu)
            _check_module_grads(eh.decoder.blocks[1], eo.decoder.blocks[1], [x0],
                                dict(cond=cm, time_embed=te, mask=mask), seed=seed)
            _check_module_grads(eh.decoder.blocks[0].norm, eo.decoder.blocks[0].norm, [x0], seed=seed)
            _check_module_grads(eh.decoder.blocks[0].residual_scale, eo.decoder.blocks[0].residual_scale, [x0],
                                seed=seed)
            _check_module_grads(eh.decoder, eo.decoder, [packed], dict(cond=cu, t=tt, mask=mask), seed=seed)
            _check_module_grads(eh, eo, [noise, cond_ref, tt, lens], seed=seed)
            if seed == 0:
                # STFT: gradient of <Re, w_r> + <Im, w_i> w.r.t. the audio; ISTFT: w.r.t. the spectrum
                _check_module_grads(eh.fft, eo.fft, [noise], takes=lambda y: torch.view_as_real(y[0]))
                spec = O.unpack_complex(packed)

                class _Ri(torch.nn.Module):      # (real view in, so that the harness sees a float leaf)
                    def __init__(self, m):
                        super().__init__()
                        self.m = m

                    def forward(self, ri):
                        return self.m(torch.view_as_complex(ri))

                _check_module_grads(_Ri(eh.ifft), _Ri(eo.ifft), [torch.view_as_real(spec).contiguous()])
    _check_module_grads(mh.loss_spec, mo.loss_spec, [T(g["audio"])])


def test_gan_loss_methods_on_oracle_scores_and_feature_maps(f2g, golden):
    """gan.discriminator_loss / generator_loss / feature_matching_loss (gan.py:57-87) called on lists of
    score / feature maps, as the reference's GAN.forward calls them."""
    import flow2gan_oracle as O
    from flow2gan_amd.models.gan import GAN
    g = golden("tiny_forward")
    mh, mo = pair(f2g, g)
    torch.manual_seed(11)
    ogan = O.GAN(mo)
    gan = GAN(mh)
    gan.discriminator.load_state_dict(ogan.discriminator.state_dict(), strict=False)
    gan = gan.to(DEV)
    gen = torch.Generator().manual_seed(2)
    real = 0.1 * torch.randn(2, 6000, generator=gen)
    fake = 0.1 * torch.randn(2, 6000, generator=gen)
    for d in (0, 1):
        with torch.no_grad():
            s_r, s_f, f_r, f_f = ogan.discriminator[d](real, fake)

        def dev(xs, grad=False):
            return [x.clone().to(DEV).requires_grad_(grad) for x in xs]

        def cpu(xs, grad=False):
            return [x.clone().requires_grad_(grad) for x in xs]

        # discriminator_loss
        so_r, so_f = cpu(s_r, True), cpu(s_f, True)
        lo = O.hinge_d_loss(so_r, so_f)
        lo.backward()
        sh_r, sh_f = dev(s_r, True), dev(s_f, True)
        lh = gan.discriminator_loss(sh_r, sh_f)
        (2.5 * lh).backward()
        assert abs(float(lh.detach()) - float(lo.detach())) < 2e-6 * abs(float(lo.detach()))
        for a, b in zip(sh_r + sh_f, so_r + so_f):
            assert relerr(a.grad, 2.5 * b.grad) < 1e-6
        # generator_loss
        so_f = cpu(s_f, True)
        lo = O.hinge_g_loss(so_f)
        lo.backward()
        sh_f = dev(s_f, True)
        lh = gan.generator_loss(sh_f)
        lh.backward()
        assert abs(float(lh.detach()) - float(lo.detach())) < 2e-6 * abs(float(lo.detach()))
        for a, b in zip(sh_f, so_f):
            assert relerr(a.grad, b.grad) < 1e-6
        # feature_matching_loss: gradient to the fake maps only (real is detached)
        fo_r, fo_f = [cpu(x, True) for x in f_r], [cpu(x, True) for x in f_f]
        lo = O.feature_matching(fo_r, fo_f)
        lo.backward()
        fh_r, fh_f = [dev(x, True) for x in f_r], [dev(x, True) for x in f_f]
        lh = gan.feature_matching_loss(fh_r, fh_f)
        lh.backward()
        assert abs(float(lh.detach()) - float(lo.detach())) < 5e-6 * abs(float(lo.detach()))
        for xs, ys in zip(fh_f, fo_f):
            for a, b in zip(xs, ys):
                assert relerr(a.grad, b.grad) < 1e-6
        for xs in fh_r:
            assert all(a.grad is None for a in xs)
        # and on the HIP discriminators' own forward outputs (same convention as the reference's)
        with torch.no_grad():
            hs_r, hs_f, hf_r, hf_f = gan.discriminator[d](real.to(DEV), fake.to(DEV))
            lh = gan.discriminator_loss(hs_r, hs_f)
            lo = O.hinge_d_loss(s_r, s_f)
            assert abs(float(lh) - float(lo)) < 1e-4 * abs(float(lo))
            lh = gan.feature_matching_loss(hf_r, hf_f)
            lo = O.feature_matching(f_r, f_f)
            assert abs(float(lh) - float(lo)) < 1e-4 * abs(float(lo))
