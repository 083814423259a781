"""GPU parity cases added in round 3, all against vectors recorded from the reference
(oracle/make_golden_r3.py):

* `spec_scaling_loss=False` (generator.py:181-184): loss + every parameter gradient;
* `get_model(checkpoint=<local .pt>)` end to end on the FULL reference test mel
  (test_from_mel.py:38-57, flow2gan/__init__.py:29-48);
* a training trajectory: six alternating D / G steps through `harness.GanStepper` with the HIP
  ScaledAdam + Eden2 (finetune.py:569-631 + optim.py:451-507 together).
"""
import hashlib
import random

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode_exact")]
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


def rms(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).pow(2).mean().sqrt())


def test_unweighted_mse_loss_switch(f2g, golden, monkeypatch):
    g = golden("tiny_mse")
    m = f2g.MelAudioGenerator(**dict(TINY, spec_scaling_loss=False, branch_dropout=0.0))
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w/")})
    m = m.to(DEV).train()
    monkeypatch.setattr(random, "random", lambda: 0.0)
    loss = m(T(g["mel"]).to(DEV), T(g["audio"]).to(DEV), T(g["lens"]), noise=T(g["noise"]).to(DEV),
             t=T(g["t"]).to(DEV))
    loss.backward()
    want = float(g["loss"])
    assert abs(float(loss) - want) < 2e-5 * abs(want), (float(loss), want)
    worst = []
    for name, p in m.named_parameters():
        if f"g/{name}" not in g:          # loss_spec's filterbank is no parameter; nothing else is skipped
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        ref = T(g[f"g/{name}"])
        err = float((p.grad.cpu().double() - ref.double()).abs().max()) / (float(ref.abs().max()) + 1e-12)
        worst.append((err, name))
    worst.sort(reverse=True)
    assert len(worst) > 100 and worst[0][0] < 2e-3, worst[:8]


def test_get_model_local_checkpoint_full_test_mel(f2g, golden, tmp_path):
    """The reference's test_from_mel.py flow: get_model(checkpoint=...) -> .to(device) -> eval ->
    infer(cond=mel, n_timesteps=4, clamp_pred=True) on the whole 205-frame test mel."""
    g = golden("full_testmel")
    from flow2gan_amd.models.config import get_generator_config
    torch.manual_seed(int(g["seed"]))
    src = f2g.MelAudioGenerator(**get_generator_config("mel_24k_base"))
    sd = src.state_dict()
    # seeded init == the reference's: bit-identical where the host's erfinv is (the digest test of
    # tests/test_host_side.py, run in the build container); here, on whatever CPU the GPU box has,
    # the probes recorded from the reference must agree to the last few ulps
    probes = golden("full_width")
    for k in probes:
        if k.startswith("probe/"):
            got = sd[k[len("probe/"):]].reshape(-1)[:256]
            assert float((got - T(probes[k])).abs().max()) < 1e-7, k
    ck = tmp_path / "libritts-mel-4-step.pt"
    torch.save({"model": sd, "batch_idx_train": 123}, ck)
    del src
    model, cfg = f2g.get_model(model_name="mel_24k_base", hf_model_name=None, checkpoint=str(ck))
    assert cfg.sampling_rate == 24000 and model.mel_hop_length == 256
    model = model.to(DEV)
    model.eval()
    mel = T(g["mel"])
    assert tuple(mel.shape) == (1, 100, 205)
    noise = 0.1 * torch.randn(1, 205 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    with torch.no_grad():
        y = model.infer(cond=mel.to(DEV), n_timesteps=4, clamp_pred=True, noise=noise.to(DEV))
    want = T(g["audio_n4"])
    assert tuple(y.shape) == tuple(want.shape) == (1, 205 * 256)
    err = rms(y, want)
    assert err < 1e-4, f"rms {err:.3e}"
    assert float(y.abs().max()) <= 1.0


def test_training_trajectory_matches_reference(f2g, golden, monkeypatch):
    from flow2gan_amd import optim
    from flow2gan_amd.harness import GanStepper
    from flow2gan_amd.models.gan import GAN
    g = golden("tiny_traj")
    gen = f2g.MelAudioGenerator(**TINY)
    gen.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w/")})
    gen.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    gan = GAN(gen)
    sd = {k: v for k, v in gan.discriminator.state_dict().items() if "spec_fn" not in k}
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().astype(np.float32).tobytes())
    assert h.hexdigest() == bytes(g["d_digest"]).decode()
    gan = gan.to(DEV)
    lr_g, lr_d, lrb_g, lrb_d = (float(v) for v in g["hyper"])
    opt_g = optim.ScaledAdam(gan.generator.named_parameters(), lr=lr_g, clipping_scale=2.0)
    sch_g = optim.Eden2(opt_g, lr_batches=lrb_g, warmup_start=0.1)
    opt_d = optim.ScaledAdam(gan.discriminator.named_parameters(), lr=lr_d, clipping_scale=2.0)
    sch_d = optim.Eden2(opt_d, lr_batches=lrb_d, warmup_start=0.1)
    monkeypatch.setattr(random, "random", lambda: 0.0)      # limiter always on, as recorded
    mel_mod = f2g.LogMelSpectrogram().to(DEV)
    stepper = GanStepper(gan, mel_mod, n_timesteps=1, gen_start_batch_idx=1,
                         optimizer_d=lambda: (opt_d.step(), sch_d.step_batch()),
                         optimizer_g=lambda: (opt_g.step(), sch_g.step_batch()))
    # the reference draws the Euler start with torch.randn (generator.py:350); feed the recorded one
    real_randn = torch.randn
    k = 0
    while f"s{k}/audio" in g:
        noise = T(g[f"s{k}/noise"]).to(DEV)

        def fake_randn(*a, noise=noise, **kw):
            shape = a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a
            if tuple(shape) == tuple(noise.shape):
                return noise / 0.1
            return real_randn(*a, **kw)

        monkeypatch.setattr(torch, "randn", fake_randn)
        info = stepper.step(T(g[f"s{k}/audio"]).to(DEV), T(g[f"s{k}/lens"]))
        monkeypatch.setattr(torch, "randn", real_randn)
        assert bool(info["train_disc"]) == bool(int(g[f"s{k}/train_disc"])), k
        want = g[f"s{k}/losses"]
        keys = ("disc_loss_mp", "disc_loss_mr") if info["train_disc"] else \
            ("gen_loss_mp", "gen_loss_mr", "feat_map_loss_mp", "feat_map_loss_mr", "mel_recon_loss")
        for name, w in zip(keys, want):
            got = float(info[name])
            assert abs(got - w) < 1e-4 * abs(w) + 1e-6, (k, name, got, float(w))
        k += 1
    assert k == 6
    for tag, mod in (("G", gan.generator), ("D", gan.discriminator)):
        sd = mod.state_dict()
        for key in g:
            if not key.startswith(f"end/{tag}/"):
                continue
            name = key[len(f"end/{tag}/"):]
            want = T(g[key]).double()
            got = sd[name].detach().cpu().double()
            rel = float((got - want).norm() / (want.norm() + 1e-12))
            assert rel < 1e-3, (tag, name, rel)
