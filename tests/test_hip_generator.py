"""GPU parity of the fused generator (flow2gan_amd) against the reference's golden vectors and
the CPU oracle: leaf intermediates, Euler inference (<= 1e-4 RMS waveform, north_star), the
stage-1 flow-matching loss and every parameter gradient."""
import random

import numpy as np
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)
DEV = "cuda"
RMS_TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 RMS waveform vs the reference CPU path


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


def tiny_model(f2g, g):
    m = f2g.MelAudioGenerator(**TINY)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w/")})
    return m.to(DEV)


def rows_of(t):  # (B,C,F) cpu -> (B*F, C)
    return t.permute(0, 2, 1).reshape(-1, t.shape[1])


def test_tiny_leafs_against_reference_vectors(f2g, golden):
    from flow2gan_amd import fused, ops
    g = golden("tiny_forward")
    m = tiny_model(f2g, g).eval()
    mel, noise, lens = T(g["mel"]).to(DEV), T(g["noise"]).to(DEV), T(g["lens"])
    with torch.no_grad():
        cond = m.encode_cond(mel)
        err = rms(cond.rows, rows_of(T(g["cond_enc"])))
        assert err < 2e-5, f"cond encoder rms {err}"
        for i, est in enumerate(m.estimators):
            packed, Fr = fused.stft_packed(noise, est.n_fft, est.hop_length)
            want = rows_of(T(g[f"br{i}/stft_packed"]))
            assert Fr == want.shape[0] // 2
            assert rms(packed[:, :est.n_fft + 2], want) < 2e-5
        # single-branch evaluation: zero the other branches through the branch weights
        tt = torch.full((2,), 0.25, device=DEV)
        cprojs = m.cond_paths(cond, noise.shape[1])
        for i in range(3):
            w = torch.zeros(3, 2, device=DEV)
            w[i] = 3.0
            y = m.model_eval(noise, tt, cprojs, [int(v) for v in lens], w)
            err = rms(y, T(g[f"br{i}/audio"]))
            assert err < 2e-5, f"branch {i} rms {err}"


@pytest.mark.parametrize("n", [1, 2, 4])
def test_tiny_infer_against_reference_vectors(f2g, golden, n):
    g = golden("tiny_forward")
    m = tiny_model(f2g, g).eval()
    mel = T(g["mel"]).to(DEV)
    with torch.no_grad():
        y = m.infer(mel, T(g["lens"]), n, clamp_pred=(n == 4), noise=T(g["noise"]).to(DEV))
        assert rms(y, T(g[f"infer_n{n}_ragged"])) < RMS_TOL
        y = m.infer(mel, None, n, clamp_pred=(n == 4), noise=T(g["noise_nolens"]).to(DEV))
        assert rms(y, T(g[f"infer_n{n}_nolens"])) < RMS_TOL


def test_full_width_infer_matches_reference(f2g, golden):
    """mel_24k_base, seeded init (bit-identical to the reference's, digest-checked on CPU),
    reference test mel -> waveform, n = 1 and 4 steps, <= 1e-4 RMS."""
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    from flow2gan_amd.models.config import get_generator_config
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(DEV).eval()
    noise = 0.1 * torch.randn(1, 64 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    with torch.no_grad():
        for n in (1, 4):
            y = m.infer(T(g["mel"]).to(DEV), None, n, True, noise=noise.to(DEV))
            err = rms(y, T(g[f"audio_n{n}"]))
            assert err < RMS_TOL, f"n={n}: rms {err:.3e}"


def test_time_paths_ahead_equal_the_per_step_time_paths(f2g, golden, monkeypatch):
    """Inference computes the time path (sinusoidal embedding -> MLP -> per-block projections,
    modules.py:569-573,451) of ALL Euler steps once, as one batch, ahead of the solver
    (generator._time_ahead / fused.time_paths_ahead) -- row by row the arithmetic of the per-step
    path, so the waveform must not move (the only licence: another split of the small stacked GEMM)."""
    from flow2gan_amd.models import generator as gen_mod
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    from flow2gan_amd.models.config import get_generator_config
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(DEV).eval()
    noise = (0.1 * torch.randn(3, 64 * 256, generator=torch.Generator().manual_seed(5))).to(DEV)
    mel = T(g["mel"]).to(DEV).repeat(3, 1, 1)
    lens = torch.tensor([64 * 256, 50 * 256, 33 * 256 + 7])
    if not gen_mod.TIME_AHEAD:
        pytest.skip("F2G_TIME_AHEAD=0 in the environment")
    with torch.no_grad():
        ahead = m.infer(mel, lens, 4, True, noise=noise).clone()
        monkeypatch.setattr(gen_mod, "TIME_AHEAD", False)
        per_step = m.infer(mel, lens, 4, True, noise=noise).clone()
    assert rms(ahead, per_step) < 2e-6, rms(ahead, per_step)


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_tiny_stage1_loss_and_all_grads(f2g, golden, tag, monkeypatch):
    g = golden("tiny_stage1")
    m = tiny_model(f2g, g).train()
    monkeypatch.setattr(random, "random", lambda: 0.0)  # limiter always on, as in the fixture
    bw = None
    if tag == "drop":
        u, idx = T(g["drop_u"]), T(g["drop_idx"])
        mask = torch.ones(2, 3)
        mask[torch.arange(2), idx] = 0.0
        mask = mask * 1.5
        w = torch.where(u < 0.05, mask, torch.ones_like(mask))
        bw = w.t().contiguous().to(DEV)
    else:
        m.branch_dropout = 0.0
    loss = m(T(g["mel"]).to(DEV), T(g["audio"]).to(DEV), T(g["lens"]), noise=T(g["noise"]).to(DEV),
             t=T(g["t"]).to(DEV), branch_weights=bw)
    loss.backward()
    want = float(g[f"{tag}/loss"])
    assert abs(float(loss) - want) < 2e-5 * abs(want), (float(loss), want)
    worst = []
    for name, p in m.named_parameters():
        ref = T(g[f"{tag}/g/{name}"])
        assert p.grad is not None, name
        err = float((p.grad.cpu().double() - ref.double()).abs().max())
        scale = float(ref.abs().max()) + 1e-12
        worst.append((err / scale, name))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:8]


def test_stage1_against_oracle_other_shape(f2g):
    """Independent of the fixtures: oracle on CPU vs HIP on the same seeded inputs, B=3, odd T."""
    import flow2gan_oracle as O
    torch.manual_seed(3)
    cfg = dict(TINY, channels=(64, 40, 24), num_layers=(2, 1, 1))
    mo = O.MelAudioGenerator(**cfg).train()
    mh = f2g.MelAudioGenerator(**cfg)
    mh.load_state_dict(mo.state_dict())
    mh = mh.to(DEV).train()
    mo.branch_dropout = mh.branch_dropout = 0.0
    gen = torch.Generator().manual_seed(4)
    B, Tn = 3, 5120
    audio = 0.1 * torch.randn(B, Tn, generator=gen)
    lens = torch.tensor([5120, 3000, 4097])
    mel = O.LogMelSpectrogram()(audio)
    noise = 0.1 * torch.randn(B, Tn, generator=gen)
    t = torch.tensor([[0.1], [0.5], [0.9]])
    import random as _r
    st = _r.getstate()
    _r.seed(5)
    lo = mo(mel, audio, lens, noise=noise, t=t)
    lo.backward()
    _r.setstate(st)
    _r.seed(5)
    lh = mh(mel.to(DEV), audio.to(DEV), lens, noise=noise.to(DEV), t=t.to(DEV))
    lh.backward()
    assert abs(float(lh) - float(lo)) < 2e-5 * abs(float(lo))
    po = dict(mo.named_parameters())
    worst = max((float((p.grad.cpu() - po[n].grad).abs().max()) / (float(po[n].grad.abs().max()) + 1e-12), n)
                for n, p in mh.named_parameters())
    assert worst[0] < 2e-3, worst


def test_log_mel_frontend_matches_reference_fixture(f2g, golden):
    g = golden("mel_frontend")
    for tag, kw, tol in (("24k", dict(sampling_rate=24000, n_fft=1024, hop_length=256, n_mels=100), 5e-4),
                         ("44k", dict(sampling_rate=44100, n_fft=2048, hop_length=512, n_mels=128), 5e-4)):
        lm = f2g.LogMelSpectrogram(**kw).to(DEV)
        want = T(g[f"{tag}/logmel"])
        got = lm(T(g[f"{tag}/wave"])[None].to(DEV))[0, :, :want.shape[1]].cpu()
        assert float((got - want).abs().max()) < tol


def test_44k_config_shapes_against_oracle(f2g):
    """mel_44k_128band_512x_base geometry (n_fft 1024/512/256, hop 512/256/128, 128 mels, loss
    n_fft 2048) at reduced width: inference and stage-1 loss/grads vs the CPU oracle."""
    import flow2gan_oracle as O
    from flow2gan_amd.models.config import get_generator_config
    cfg = dict(get_generator_config("mel_44k_128band_512x_base"))
    cfg.update(channels=(40, 32, 24), num_layers=(1, 1, 1), cond_enc_channels=32,
               cond_enc_num_layers=1, time_embed_channels=32, branch_dropout=0.0)
    torch.manual_seed(8)
    mo = O.MelAudioGenerator(**cfg)
    mh = f2g.MelAudioGenerator(**cfg)
    mh.load_state_dict(mo.state_dict())
    mh = mh.to(DEV)
    gen = torch.Generator().manual_seed(9)
    B, Tn = 2, 11025
    audio = 0.1 * torch.randn(B, Tn, generator=gen)
    lens = torch.tensor([11025, 9000])
    lm_o = O.LogMelSpectrogram(44100, 2048, 512, 128)
    lm_h = f2g.LogMelSpectrogram(44100, 2048, 512, 128).to(DEV)
    mel = lm_o(audio)
    assert float((lm_h(audio.to(DEV)).cpu() - mel).abs().max()) < 1e-3
    noise = 0.1 * torch.randn(B, Tn, generator=gen)
    t = torch.tensor([[0.35], [0.6]])
    random.seed(3)
    mo.train()
    lo = mo(mel, audio, lens, noise=noise, t=t)
    lo.backward()
    random.seed(3)
    mh.train()
    lh = mh(mel.to(DEV), audio.to(DEV), lens, noise=noise.to(DEV), t=t.to(DEV))
    lh.backward()
    assert abs(float(lh) - float(lo)) < 5e-5 * abs(float(lo))
    po = dict(mo.named_parameters())
    worst = max((float((p.grad.cpu() - po[n].grad).abs().max()) / (float(po[n].grad.abs().max()) + 1e-12), n)
                for n, p in mh.named_parameters())
    assert worst[0] < 3e-3, worst
    mo.eval(), mh.eval()
    with torch.no_grad():
        nz = 0.1 * torch.randn(B, mel.shape[2] * 512, generator=gen)
        yo = mo.infer(mel, None, 2, True, noise=nz)
        yh = mh.infer(mel.to(DEV), None, 2, True, noise=nz.to(DEV))
    assert rms(yh, yo) < RMS_TOL


def test_harness_compute_loss_matches_oracle(f2g, golden):
    """C1/C2: compute_loss_stage1 / compute_loss_stage2 (cond computed inside the step, reference
    loss weights) against the oracle on the same inputs."""
    import flow2gan_oracle as O
    from flow2gan_amd import harness
    from flow2gan_amd.models.gan import GAN
    g = golden("tiny_stage2")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    mo = O.MelAudioGenerator(**TINY)
    mo.load_state_dict(sd)
    mh = f2g.MelAudioGenerator(**TINY)
    mh.load_state_dict(sd)
    mo.branch_dropout = mh.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    go = O.GAN(mo)
    gh = GAN(mh)
    gh.discriminator.load_state_dict(go.discriminator.state_dict(), strict=False)
    gh = gh.to(DEV)
    audio = T(g["audio"])
    lens = torch.tensor([6000, 6000])
    lm_o, lm_h = O.LogMelSpectrogram(), f2g.LogMelSpectrogram().to(DEV)
    # deterministic noise for both: patch torch.randn used by infer()
    noise = T(g["noise"])
    for train_disc in (True, False):
        want = go(lm_o(audio), audio, lens, 1, train_disc, noise=noise)
        ws = (1.0, 0.1) if train_disc else (1.0, 0.1, 1.0, 0.1, 45.0)
        want_total = float(sum(w * l for w, l in zip(ws, want)))
        orig = torch.randn
        try:
            torch.randn = lambda *a, **k: (noise / 0.1).to(DEV)
            loss, info = harness.compute_loss_stage2(audio.to(DEV), lens, gh, lm_h, 1,
                                                     train_disc=train_disc)
        finally:
            torch.randn = orig
        assert abs(float(loss) - want_total) < 1e-4 * abs(want_total), (train_disc, float(loss), want_total)
        assert info["samples"] == 2


def test_full_size_batch_64_reproduces_reference_waveform(f2g, golden):
    """BASELINE size (B=64, full-width mel_24k_base) through a size-independent property: items of a
    batch are independent, so 64 copies of the reference's test mel must give 64 copies of the
    reference's waveform -- exercised on the B=64 tile / split-K / launch-lane paths."""
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    from flow2gan_amd.models.config import get_generator_config
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(DEV).eval()
    noise = 0.1 * torch.randn(1, 64 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    mel = T(g["mel"]).to(DEV).expand(64, -1, -1).contiguous()
    with torch.no_grad():
        y = m.infer(mel, None, 4, True, noise=noise.to(DEV).expand(64, -1).contiguous())
    want = T(g["audio_n4"])
    assert y.shape == (64, want.shape[1])
    worst = max(rms(y[b:b + 1], want) for b in range(64))
    assert worst < RMS_TOL, worst
    from flow2gan_amd import ops as _ops
    # rows agree with each other (up to the summation order of split tiles; the opt-in split-bf16
    # GEMM mode amplifies those last-bit differences through its 2^-16 products)
    # (bf16x6: the same last-bit differences of the atomically accumulated time-MLP GEMMs, seen through
    # other roundings -- a one-ulp change of an operand re-draws the error of its six-product sum (<= 2^-23
    # of the product, against 2^-24 per fp32 add), so rows that differ in last bits drift apart ~3x faster
    # than on the exact fp32 MFMA: 1.6e-5 with the K >= 2048 GEMMs in this mode (round 4), 3.8e-5 with every
    # GEMM from K = 384 on (round 5).  Every row stays within the 1e-4 RMS of the reference waveform above.)
    assert float((y - y[:1]).abs().max()) < {1: 5e-5, 3: 6e-5}.get(_ops.GEMM_PRECISION, 1e-5)


def test_full_width_stage1_loss_and_grads_vs_oracle_then_batch_64(f2g, monkeypatch):
    """Full-width mel_24k_base stage-1 step: loss and gradients against the CPU oracle at B=2, then
    the B=64 step on 32 copies of that batch must return the same loss and the same (mean) grads."""
    import flow2gan_oracle as O
    from flow2gan_amd.models.config import get_generator_config
    monkeypatch.setattr(random, "random", lambda: 1.0)  # LimitParamValue off on both sides
    torch.manual_seed(21)
    o = O.build_generator("mel_24k_base")
    o.branch_dropout = 0.0
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base"))
    m.load_state_dict(o.state_dict())
    m.branch_dropout = 0.0
    m = m.to(DEV).train()
    gen = torch.Generator().manual_seed(5)
    Tn = 24 * 256
    audio = (0.1 * torch.randn(2, Tn, generator=gen)).clamp(-1, 1)
    lens = torch.tensor([Tn, Tn - 700])
    noise = 0.1 * torch.randn(2, Tn, generator=gen)
    t = torch.rand(2, 1, generator=gen)
    lm = O.LogMelSpectrogram()
    mel = lm(audio)
    o.train()
    lo = o(mel, audio, lens, noise=noise, t=t)
    lo.backward()
    names = ["cond_encoder.in_proj.weight", "estimators.0.decoder.blocks.7.pwconv2.weight",
             "estimators.1.decoder.in_norm.bias", "estimators.2.decoder.blocks.0.dwconv.weight",
             "estimators.2.decoder.blocks.3.residual_scale.scale",
             "estimators.0.decoder.time_mlp.0.weight", "estimators.1.decoder.out_proj.bias"]
    po, pm = dict(o.named_parameters()), dict(m.named_parameters())

    def run(rep):
        m.zero_grad()
        loss = m(mel.to(DEV).repeat(rep, 1, 1), audio.to(DEV).repeat(rep, 1), lens.repeat(rep),
                 noise=noise.to(DEV).repeat(rep, 1), t=t.to(DEV).repeat(rep, 1))
        loss.backward()
        return float(loss), {k: pm[k].grad.detach().cpu().clone() for k in names}

    l2, g2 = run(1)
    assert abs(l2 - float(lo)) < 2e-5 * abs(float(lo)) + 1e-7, (l2, float(lo))
    for k in names:
        ref = po[k].grad
        err = float((g2[k] - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 3e-3, (k, err)
    l64, g64 = run(32)   # B = 64
    assert abs(l64 - l2) < 1e-5 * abs(l2) + 1e-7, (l64, l2)
    for k in names:
        err = float((g64[k] - g2[k]).abs().max()) / (float(g2[k].abs().max()) + 1e-12)
        assert err < 2e-3, (k, err)


def test_plain_bf16_throughput_mode_stays_close_to_fp32(f2g, golden):
    """precision 2 (bf16 operands, fp32 accumulate; BASELINE config 2's inference mode) is not a
    parity mode; this pins its error level so that a broken kernel cannot hide behind it."""
    from flow2gan_amd import ops
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    from flow2gan_amd.models.config import get_generator_config
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(DEV).eval()
    noise = 0.1 * torch.randn(1, 64 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    was = ops.GEMM_PRECISION
    try:
        ops.set_gemm_precision("bf16")
        with torch.no_grad():
            y = m.infer(T(g["mel"]).to(DEV), None, 4, True, noise=noise.to(DEV))
    finally:
        ops.GEMM_PRECISION = was
    want = T(g["audio_n4"])
    err = rms(y, want)
    sig = float(want.double().pow(2).mean().sqrt())
    assert 1e-6 < err < 0.05 * sig, (err, sig)   # bf16-level, far from broken
    A = torch.randn(700, 512, generator=torch.Generator().manual_seed(1))
    W = torch.randn(300, 512, generator=torch.Generator().manual_seed(2)) * 0.05
    out = torch.empty(700, 300, device=DEV)
    try:
        ops.set_gemm_precision("bf16")
        ops.gemm(ops.mat(A.to(DEV)), ops.mat(W.to(DEV)), out)
    finally:
        ops.GEMM_PRECISION = was
    ref = A.double() @ W.double().t()
    rel = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert 1e-5 < rel < 2e-2, rel


def test_plain_bf16_inference_at_the_baseline_batch(f2g):
    """BASELINE config 2's own shape and arithmetic: 4-step bf16 inference at B = 64 x 94 frames.
    The samples of a batch are independent, so a batch of 32 copies of two items must reproduce
    the B = 2 result row for row (the only difference allowed: tile / split choices of the GEMMs),
    and the B = 2 result must sit at bf16 distance from the exact-fp32 path."""
    from flow2gan_amd import ops
    from flow2gan_amd.models.config import get_generator_config
    torch.manual_seed(1234)
    m = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(DEV).eval()
    rg = torch.Generator().manual_seed(12)
    mel = (torch.randn(2, 100, 94, generator=rg) * 1.5 - 4.0).to(DEV)
    noise = (0.1 * torch.randn(2, 94 * 256, generator=rg)).to(DEV)
    was = ops.GEMM_PRECISION
    try:
        ops.set_gemm_precision("fp32")
        with torch.no_grad():
            y32 = m.infer(mel, None, 4, True, noise=noise).clone()
        ops.set_gemm_precision("bf16")
        with torch.no_grad():
            y2 = m.infer(mel, None, 4, True, noise=noise).clone()
            y64 = m.infer(mel.repeat(32, 1, 1), None, 4, True, noise=noise.repeat(32, 1)).clone()
            # (default: every layer of the three branches is ONE launch, f2g_fused_block_multi;
            # the per-branch launches on three lanes must give the same waveform)
            multi_was, ops.FUSED_MULTI = ops.FUSED_MULTI, False
            try:
                y2_lanes = m.infer(mel, None, 4, True, noise=noise).clone()
            finally:
                ops.FUSED_MULTI = multi_was
    finally:
        ops.GEMM_PRECISION = was
    assert ops.FUSED_MULTI or os.environ.get("F2G_FUSED_MULTI") == "0"
    sig = float(y32.double().pow(2).mean().sqrt())
    e2 = rms(y2, y32)
    assert 1e-6 < e2 < 0.05 * sig, (e2, sig)
    assert tuple(y64.shape) == (64, 94 * 256)
    # The B = 64 grid uses other tile shapes / K splits than the B = 2 grid: the fp32 sums come out
    # in another order, single activations cross a bf16 rounding boundary, and 4 steps x 8 blocks
    # carry that on -- so B = 64 differs from B = 2 by about the mode's own error (measured: 0.9 of
    # it), and it is held to the same bound against the exact-fp32 waveform ...
    assert rms(y2_lanes, y2) < 3.0 * e2, (rms(y2_lanes, y2), e2)
    e64 = rms(y64, y32.repeat(32, 1))
    assert 1e-6 < e64 < 0.05 * sig and e64 < 3.0 * e2, (e64, e2, sig)
    # ... and so do the 32 copies of an item among themselves: stream-K / split-K launches (the
    # time MLP, the thin head GEMMs) accumulate atomically, so two rows with equal inputs can differ
    # in the last bit, which the bf16 roundings downstream turn into differences of the mode's own
    # error class (measured: 0.8 of it).  What this pins is that no copy is WRONG (a tile-edge bug at
    # M = 6016 / 12032 / 24064 would show as an O(signal) deviation of the rows it touches).
    for k in (1, 7, 31):
        for item in (0, 1):
            d = rms(y64[2 * k + item], y64[item])
            assert d < 3.0 * e2, (k, item, d, e2)


def test_non_default_constructor_switches_against_reference_vectors(f2g, golden, monkeypatch):
    """use_cond_encoder=False, pred_x1=False (velocity objective), branch_reduction="sum"
    (generator.py:86-97,165-168,218,263; unused by the named configs): stage-1 loss, every
    parameter gradient and a 2-step Euler inference against the REFERENCE's recorded vectors."""
    g = golden("tiny_switches")
    cfg = dict(TINY, use_cond_encoder=False, pred_x1=False, branch_reduction="sum", branch_dropout=0.0)
    m = f2g.MelAudioGenerator(**cfg)
    assert not hasattr(m, "cond_encoder")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    monkeypatch.setattr(random, "random", lambda: 0.0)
    mel, audio, noise = T(g["mel"]).to(DEV), T(g["audio"]).to(DEV), T(g["noise"]).to(DEV)
    lens, t = T(g["lens"]), T(g["t"]).to(DEV)
    loss = m(mel, audio, lens, noise=noise, t=t)
    assert abs(float(loss) - float(g["loss"])) < 2e-5 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
    loss.backward()
    worst = max((float((p.grad.cpu() - T(g[f"g/{n}"])).abs().max()) / (float(T(g[f"g/{n}"]).abs().max()) + 1e-12), n)
                for n, p in m.named_parameters())
    assert worst[0] < 2e-3, worst
    m.eval()
    with torch.no_grad():
        y = m.infer(mel, lens, 2, noise=noise)
    assert rms(y, T(g["infer_n2"])) < RMS_TOL


def test_mel_noise_augmentation_draws_like_the_reference(f2g, golden, monkeypatch):
    """max_add_noise_scale > 0 (generator.py:306-309,342-345): in training mode the condition gets
    randn_like(cond) * rand(B,1,1) * scale added, drawn in the reference's order from torch's
    generator; off in eval mode."""
    g = golden("tiny_forward")
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    m = f2g.MelAudioGenerator(max_add_noise_scale=0.3, **TINY)
    m.load_state_dict(sd)
    m = m.to(DEV)
    plain = f2g.MelAudioGenerator(**TINY)
    plain.load_state_dict(sd)
    plain = plain.to(DEV)
    mel, noise = T(g["mel"]).to(DEV), T(g["noise"]).to(DEV)
    lens = T(g["lens"])
    monkeypatch.setattr(random, "random", lambda: 1.0)
    m.train(), plain.train()
    m.branch_dropout = plain.branch_dropout = 0.0
    with torch.no_grad():
        torch.manual_seed(123)
        got = m.infer(mel, lens, 1, noise=noise)
        torch.manual_seed(123)
        e = torch.randn_like(mel) * torch.rand(mel.shape[0], 1, 1, device=mel.device) * 0.3
        want = plain.infer(mel + e, lens, 1, noise=noise)
        assert float((got - want).abs().max()) < 1e-6
        assert float((got - plain.infer(mel, lens, 1, noise=noise)).abs().max()) > 5e-6
        m.eval(), plain.eval()
        assert torch.equal(m.infer(mel, lens, 1, noise=noise), plain.infer(mel, lens, 1, noise=noise))
