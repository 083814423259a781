"""CPU: the oracle (oracle/flow2gan_oracle.py) against the committed reference vectors.

The vectors in tests/golden were produced by oracle/make_golden.py from the real
reference; these tests are what keeps the oracle pinned where the reference is absent.
"""
import hashlib
import random

import numpy as np
import pytest
import torch

import flow2gan_oracle as O

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)


def T(a):
    return torch.from_numpy(np.asarray(a))


def tiny_from(g):
    m = O.MelAudioGenerator(**TINY)
    sd = {k[2:]: T(v) for k, v in g.items() if k.startswith("w/")}
    m.load_state_dict(sd)
    return m


def digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().numpy().astype(np.float32).tobytes())
    return h.hexdigest()


def test_mel_frontend_matches_reference_fixture(golden):
    g = golden("mel_frontend")
    for tag, kw, tol in (("24k", dict(sampling_rate=24000, n_fft=1024, hop_length=256, n_mels=100), 5e-4),
                         ("44k", dict(sampling_rate=44100, n_fft=2048, hop_length=512, n_mels=128), 5e-6)):
        want = T(g[f"{tag}/logmel"])
        got = O.LogMelSpectrogram(**kw)(T(g[f"{tag}/wave"])[None])[0, :, :want.shape[1]]
        assert (got - want).abs().max() < tol


def test_tiny_forward_leafs_and_infer(golden):
    g = golden("tiny_forward")
    m = tiny_from(g).eval()
    mel, noise, lens = T(g["mel"]), T(g["noise"]), T(g["lens"])
    with torch.no_grad():
        cond = m.cond_encoder(mel)
        assert torch.allclose(cond, T(g["cond_enc"]), atol=1e-6)
        tt = torch.full((2,), 0.25)
        for i, est in enumerate(m.estimators):
            spec, _ = est.fft(noise, lens)
            assert torch.allclose(O.pack_complex(spec), T(g[f"br{i}/stft_packed"]), atol=1e-5)
            assert torch.allclose(est(noise, cond, tt, lens), T(g[f"br{i}/audio"]), atol=1e-6)
        for n in (1, 2, 4):
            y = m.infer(mel, lens, n, clamp_pred=(n == 4), noise=noise)
            assert torch.allclose(y, T(g[f"infer_n{n}_ragged"]), atol=1e-6)
            y = m.infer(mel, None, n, clamp_pred=(n == 4), noise=T(g["noise_nolens"]))
            assert torch.allclose(y, T(g[f"infer_n{n}_nolens"]), atol=1e-6)


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_tiny_stage1_loss_and_grads(golden, tag, monkeypatch):
    g = golden("tiny_stage1")
    m = tiny_from(g).train()
    m.branch_dropout = 0.05 if tag == "drop" else 0.0
    q = [T(g["t"]), T(g["drop_u"])]
    monkeypatch.setattr(torch, "rand", lambda *a, **k: q.pop(0))
    monkeypatch.setattr(torch, "randint", lambda *a, **k: T(g["drop_idx"]))
    monkeypatch.setattr(random, "random", lambda: 0.0)
    cond = m.cond_encoder(T(g["mel"]))
    loss = m.fm_loss(T(g["noise"]), T(g["audio"]), cond, T(g["lens"]))
    loss.backward()
    assert abs(float(loss) - float(g[f"{tag}/loss"])) < 1e-5
    for n, p in m.named_parameters():
        want = T(g[f"{tag}/g/{n}"])
        assert (p.grad - want).abs().max() <= 1e-5 * (1 + want.abs().max()), n


def test_tiny_stage2_losses(golden, monkeypatch):
    g = golden("tiny_stage2")
    gen = tiny_from(g)
    gen.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    gan = O.GAN(gen)
    sd = {k: v for k, v in gan.discriminator.state_dict().items() if "spec_fn" not in k}
    assert digest(sd) == bytes(g["d_digest"]).decode()
    monkeypatch.setattr(random, "random", lambda: 0.0)
    mel, audio, noise = T(g["mel"]), T(g["audio"]), T(g["noise"])
    for tag, n in (("n1", 1), ("n2", 2)):
        lens = T(g[f"{tag}/lens"])
        d = gan(mel, audio, lens, n, True, noise=noise)
        assert np.allclose([float(x) for x in d], g[f"{tag}/D/losses"], atol=2e-5)
        gan.zero_grad()
        ls = gan(mel, audio, lens, n, False, noise=noise)
        assert np.allclose([float(x) for x in ls], g[f"{tag}/G/losses"], rtol=1e-5, atol=2e-5)
        total = sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls))
        total.backward()
        for k, p in gan.generator.named_parameters():
            want = T(g[f"{tag}/G/g/{k}"])
            assert (p.grad - want).abs().max() <= 2e-5 * (1 + want.abs().max()), k


def test_full_width_init_and_infer(golden):
    g = golden("full_width")
    torch.manual_seed(int(g["seed"]))
    m = O.build_generator("mel_24k_base").eval()
    assert digest(m.state_dict()) == bytes(g["digest"]).decode()
    noise = 0.1 * torch.randn(1, 64 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    with torch.no_grad():
        y = m.infer(T(g["mel"]), None, 1, True, noise=noise)
    assert float((y - T(g["audio_n1"])).pow(2).mean().sqrt()) < 1e-6


@pytest.mark.parametrize("case", ["p100", "p8", "noclip"])
def test_scaled_adam_oracle_matches_reference_vectors(golden, case):
    """oracle/scaled_adam_oracle.py against the reference optimizer's recorded trajectory
    (tests/golden/scaled_adam.npz, written by oracle/make_golden_optim.py)."""
    from scaled_adam_oracle import ScaledAdamOracle, eden2_lr
    g = golden("scaled_adam")
    n, steps = int(g["n_tensors"]), int(g["n_steps"])
    clip, period, sup = g[f"{case}/kw"]
    params = [T(g[f"init/{i}"]).clone() for i in range(n)]
    opt = ScaledAdamOracle(params, lr=0.045, clipping_scale=(clip if clip > 0 else None),
                           clipping_update_period=int(period), size_update_period=int(sup))
    clipped = 0
    for k in range(steps):
        opt.g["lr"] = 0.045 if k == 0 else eden2_lr(0.045, k, 10, 8, 0.1)
        assert k == 0 or abs(opt.g["lr"] - g[f"{case}/lrs"][k - 1]) < 1e-12
        opt.step([T(g[f"grad/{k}/{i}"]) for i in range(n)])
        clipped += opt.last_clip < 1.0
        if f"{case}/step{k + 1}/0" in g:
            for i in range(n):
                want = T(g[f"{case}/step{k + 1}/{i}"])
                err = float((params[i] - want).abs().max()) / (float(want.abs().max()) + 1e-12)
                assert err < 2e-6, (case, k + 1, i, err)
    assert (clipped > 0) == (clip > 0)   # the trajectory exercises the clipping branch


def test_full_test_mel_4_step_inference(golden):
    """Oracle vs the reference's test_from_mel.py flow on the full 205-frame test mel
    (tests/golden/full_testmel.npz, oracle/make_golden_r3.py)."""
    g = golden("full_testmel")
    torch.manual_seed(int(g["seed"]))
    m = O.build_generator("mel_24k_base").eval()
    assert digest(m.state_dict()) == bytes(g["digest"]).decode()
    noise = 0.1 * torch.randn(1, 205 * 256, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    with torch.no_grad():
        y = m.infer(T(g["mel"]), None, 4, True, noise=noise)
    assert float((y - T(g["audio_n4"])).pow(2).mean().sqrt()) < 1e-6


def test_training_trajectory_oracle(golden, monkeypatch):
    """Oracle GAN + oracle ScaledAdam / Eden2 against the reference's recorded six-step D / G
    trajectory (tests/golden/tiny_traj.npz): finetune.py:569-631 + optim.py:451-507 together."""
    from scaled_adam_oracle import ScaledAdamOracle, eden2_lr
    g = golden("tiny_traj")
    gen = tiny_from(g)
    gen.branch_dropout = 0.0
    torch.manual_seed(int(g["d_seed"]))
    gan = O.GAN(gen)
    monkeypatch.setattr(random, "random", lambda: 0.0)
    lr_g, lr_d, lrb_g, lrb_d = (float(v) for v in g["hyper"])
    pg = [p for _, p in gan.generator.named_parameters()]
    pd = [p for _, p in gan.discriminator.named_parameters()]
    og = ScaledAdamOracle([p.data for p in pg], lr=lr_g, clipping_scale=2.0)
    od = ScaledAdamOracle([p.data for p in pd], lr=lr_d, clipping_scale=2.0)
    nd = ng = 0
    k = 0
    while f"s{k}/audio" in g:
        disc = bool(int(g[f"s{k}/train_disc"]))
        gan.zero_grad()
        ls = gan(T(g[f"s{k}/mel"]), T(g[f"s{k}/audio"]), T(g[f"s{k}/lens"]), 1, disc,
                 noise=T(g[f"s{k}/noise"]))
        assert np.allclose([float(x) for x in ls], g[f"s{k}/losses"], rtol=2e-6, atol=2e-6), k
        total = sum(w * l for w, l in zip((1.0, 0.1) if disc else (1.0, 0.1, 1.0, 0.1, 45.0), ls))
        total.backward()
        if disc:
            od.g["lr"] = lr_d if nd == 0 else eden2_lr(lr_d, nd, lrb_d, 500.0, 0.1)   # (Eden2 sets the lr in step_batch)
            od.step([p.grad for p in pd])
            nd += 1
        else:
            og.g["lr"] = lr_g if ng == 0 else eden2_lr(lr_g, ng, lrb_g, 500.0, 0.1)
            og.step([p.grad for p in pg])
            ng += 1
        k += 1
    assert k == 6
    for tag, sd in (("G", gan.generator.state_dict()), ("D", gan.discriminator.state_dict())):
        for key in g:
            if key.startswith(f"end/{tag}/"):
                want = T(g[key])
                got = sd[key[len(f"end/{tag}/"):]]
                assert float((got - want).norm() / (want.norm() + 1e-12)) < 1e-5, key
