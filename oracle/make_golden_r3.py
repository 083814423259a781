"""Round-3 fixtures, recorded from the REAL reference in the build container (the files
make_golden.py / make_golden_optim.py write stay untouched, so they remain bit-reproducible).

TEST INFRASTRUCTURE.  `python oracle/make_golden_r3.py [mse] [testmel] [traj]` where /root/reference
exists.  Writes inputs / outputs only (never reference source):

  tiny_mse.npz      spec_scaling_loss=False (generator.py:181-184): stage-1 loss + every gradient
  full_testmel.npz  test_from_mel.py:38-57 on the FULL test_data/mel/1089_134686_000002_000000.pt
                    (205 frames): seeded mel_24k_base weights -> 4-step Euler audio, clamp_pred
  tiny_traj.npz     finetune.py:569-631 + optim.py:451-507 together: six alternating D / G steps of
                    the tiny GAN with the reference's ScaledAdam + Eden2 (finetune.py:917-921), each
                    step on its own batch with injected noise; per-step losses, probe parameters
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the shims, imports the reference)
from make_golden import O, Patched, digest, npy, queue_fn, rcfg, rgan, rgen, sd_np  # noqa: E402
from flow2gan import optim as roptim  # noqa: E402
from scaled_adam_oracle import ScaledAdamOracle  # noqa: E402

OUT = MG.OUT
TINY = MG.TINY


def case_mse():
    cfg = dict(TINY, spec_scaling_loss=False, branch_dropout=0.0)
    torch.manual_seed(21)
    ref = rgen.MelAudioGenerator(**cfg)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.ndim >= 2:
                p.mul_(3.0)
            elif n.endswith(".bias") and "norm" not in n:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    B, T = 2, 6000
    audio = 0.1 * torch.randn(B, T, generator=g)
    lens = torch.tensor([6000, 4600])
    mel = O.LogMelSpectrogram()(audio)
    noise = 0.1 * torch.randn(B, T, generator=g)
    t = torch.tensor([[0.25], [0.6]])
    out = dict(sd_np(ref.state_dict()))
    out.update(audio=npy(audio), lens=lens.numpy(), mel=npy(mel), noise=npy(noise), t=npy(t))
    ref.train()
    with Patched(torch__rand=queue_fn([t]), torch__randn_like=queue_fn([noise / 0.1]),
                 random__random=lambda: 0.0):
        loss = ref(cond=mel, audio=audio, audio_lens=lens)
    loss.backward()
    out["loss"] = npy(loss)
    for n, p in ref.named_parameters():
        if p.grad is not None:
            out[f"g/{n}"] = npy(p.grad)
    print(f"[tiny_mse] loss {float(loss):.6e}")
    np.savez_compressed(os.path.join(OUT, "tiny_mse.npz"), **out)


def case_testmel():
    torch.manual_seed(1234)
    ref = rgen.MelAudioGenerator(**rcfg.get_generator_config("mel_24k_base"))
    d_ref = digest(ref.state_dict())
    ref.eval()
    mel = torch.load(os.path.join(MG.ref_shims.REFERENCE_ROOT,
                                  "test_data/mel/1089_134686_000002_000000.pt"))
    assert tuple(mel.shape) == (1, 100, 205)
    T = mel.shape[2] * 256
    noise = 0.1 * torch.randn(1, T, generator=torch.Generator().manual_seed(9876))
    with torch.no_grad(), Patched(torch__randn=queue_fn([noise / 0.1])):
        # exactly test_from_mel.py:51: model.infer(cond=cond, n_timesteps=4, clamp_pred=True)
        y = ref.infer(cond=mel, n_timesteps=4, clamp_pred=True)
    torch.manual_seed(1234)
    orc = O.build_generator("mel_24k_base").eval()
    with torch.no_grad():
        yo = orc.infer(mel, None, 4, True, noise=noise)
    print(f"[full_testmel] oracle-vs-ref rms {float((yo - y).pow(2).mean().sqrt()):.3e}, "
          f"out rms {float(y.pow(2).mean().sqrt()):.3e}, shape {tuple(y.shape)}")
    np.savez_compressed(os.path.join(OUT, "full_testmel.npz"), mel=npy(mel), seed=np.array(1234),
                        noise_seed=np.array(9876), audio_n4=npy(y),
                        digest=np.frombuffer(d_ref.encode(), dtype=np.uint8))


TRAJ_STEPS = 6
PROBES_G = ("estimators.0.decoder.blocks.1.pwconv2.weight", "cond_encoder.in_proj.weight",
            "estimators.2.decoder.in_norm.bias")
PROBES_D = ("0.discriminators.0.convs.1.bias", "1.discriminators.2.band_convs.1.0.weight",
            "0.discriminators.4.conv_post.weight")


def case_traj():
    ref_g, orc_g = MG.tiny_pair()
    ref_g.branch_dropout = orc_g.branch_dropout = 0.0
    torch.manual_seed(77)
    ref = rgan.GAN(ref_g, **rcfg.get_gan_config("gan_multi_scale_mel_recon"))
    torch.manual_seed(77)
    orc = O.GAN(orc_g)
    sd_d = {k: v for k, v in ref.discriminator.state_dict().items() if "spec_fn" not in k}
    # (copies: the arrays sd_np returns alias the live parameters, which the optimizer updates in place)
    out = {k: v.copy() for k, v in sd_np(ref_g.state_dict()).items()}
    out["d_digest"] = np.frombuffer(digest(sd_d).encode(), dtype=np.uint8)
    out["d_seed"] = np.array(77)
    # finetune.py:917-921 with the CLI defaults (:123-153)
    lr_g, lr_d, lrb_g, lrb_d = 0.002, 0.02, 20000, 5000
    opt_g = roptim.ScaledAdam(ref.generator.named_parameters(), lr=lr_g, clipping_scale=2.0)
    sch_g = roptim.Eden2(opt_g, lr_batches=lrb_g, warmup_start=0.1)
    opt_d = roptim.ScaledAdam(ref.discriminator.named_parameters(), lr=lr_d, clipping_scale=2.0)
    sch_d = roptim.Eden2(opt_d, lr_batches=lrb_d, warmup_start=0.1)
    out["hyper"] = np.array([lr_g, lr_d, lrb_g, lrb_d], dtype=np.float64)
    # the oracle runs the same trajectory (its own optimizer restatement) beside the reference
    o_pg = [p for _, p in orc.generator.named_parameters()]
    o_pd = [p for _, p in orc.discriminator.named_parameters()]
    oo_g = ScaledAdamOracle([p.data for p in o_pg], lr=lr_g, clipping_scale=2.0)
    oo_d = ScaledAdamOracle([p.data for p in o_pd], lr=lr_d, clipping_scale=2.0)
    g = torch.Generator().manual_seed(41)
    B, T = 2, 6000
    w_d, w_g = (1.0, 0.1), (1.0, 0.1, 1.0, 0.1, 45.0)
    train_disc = True
    for k in range(TRAJ_STEPS):
        audio = 0.1 * torch.randn(B, T, generator=g)
        audio[1] *= 1.5
        lens = torch.tensor([T, T if k % 3 else 4600 + 128 * k])
        noise = 0.1 * torch.randn(B, T, generator=g)
        mel = O.LogMelSpectrogram()(audio)
        out[f"s{k}/audio"], out[f"s{k}/lens"], out[f"s{k}/noise"] = npy(audio), lens.numpy(), npy(noise)
        out[f"s{k}/mel"] = npy(mel)
        res = []
        for m, is_ref in ((ref, True), (orc, False)):
            m.zero_grad()
            with Patched(torch__randn=queue_fn([noise / 0.1]), random__random=lambda: 0.0):
                if is_ref:
                    ls = m(cond=mel, audio=audio, audio_lens=lens, n_timesteps=1, train_disc=train_disc)
                else:
                    ls = m(mel, audio, lens, 1, train_disc, noise=noise)
            total = sum(w * l for w, l in zip(w_d if train_disc else w_g, ls))
            total.backward()
            res.append([float(l) for l in ls])
        if train_disc:
            opt_d.step()
            oo_d.g["lr"] = opt_d.param_groups[0]["lr"]
            sch_d.step_batch()
            oo_d.step([p.grad for p in o_pd])
        else:
            opt_g.step()
            oo_g.g["lr"] = opt_g.param_groups[0]["lr"]
            sch_g.step_batch()
            oo_g.step([p.grad for p in o_pg])
        out[f"s{k}/train_disc"] = np.array(int(train_disc))
        out[f"s{k}/losses"] = np.array(res[0], dtype=np.float64)
        dl = max(abs(a - b) / (abs(a) + 1e-12) for a, b in zip(*res))
        print(f"[tiny_traj] step {k} {'D' if train_disc else 'G'} ref losses "
              f"{[round(x, 6) for x in res[0]]}  oracle rel diff {dl:.2e}")
        # finetune.py:614-626 with gen_start_batch_idx = 1: strict alternation from the first batch
        train_disc = not train_disc
    sd_g, sd_dd = ref.generator.state_dict(), ref.discriminator.state_dict()
    so_g, so_d = orc.generator.state_dict(), orc.discriminator.state_dict()
    for names, sd, so, tag in ((PROBES_G, sd_g, so_g, "G"), (PROBES_D, sd_dd, so_d, "D")):
        for n in names:
            out[f"end/{tag}/{n}"] = npy(sd[n])
            d = float((sd[n] - so[n]).norm() / (sd[n].norm() + 1e-12))
            print(f"[tiny_traj] end {tag} {n}: oracle rel L2 diff {d:.2e}, "
                  f"max {float((sd[n] - so[n]).abs().max()):.2e} of {float(sd[n].abs().max()):.2e}")
    np.savez_compressed(os.path.join(OUT, "tiny_traj.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["mse", "testmel", "traj"]
    if "mse" in which:
        case_mse()
    if "testmel" in which:
        case_testmel()
    if "traj" in which:
        case_traj()
    for f in ("tiny_mse.npz", "full_testmel.npz", "tiny_traj.npz"):
        pth = os.path.join(OUT, f)
        if os.path.exists(pth):
            print(f, os.path.getsize(pth) // 1024, "KiB")
