"""Generate tests/golden/*.npz by running the REAL reference in the build container.

TEST INFRASTRUCTURE.  Run as `python oracle/make_golden.py` where
`/root/reference` exists.  It (1) checks `flow2gan_oracle.py` against the
reference on every case and prints the deviations, (2) writes the reference's
inputs/outputs (never its source) as small fp32 fixtures.  The fixtures are
what pins the oracle on the GPU box, where the reference cannot travel.
"""
from __future__ import annotations

import hashlib
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import flow2gan_oracle as O  # noqa: E402
from flow2gan.models import config as rcfg  # noqa: E402
from flow2gan.models import discriminators as rdisc  # noqa: E402
from flow2gan.models import gan as rgan  # noqa: E402
from flow2gan.models import generator as rgen  # noqa: E402
from flow2gan.models import modules as rmod  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

TINY = dict(sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
            n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(48, 32, 24),
            time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
            cond_enc_channels=32, cond_enc_num_layers=1)


def npy(t):
    return t.detach().cpu().numpy()


def sd_np(sd, prefix="w/"):
    return {prefix + k: npy(v) for k, v in sd.items()}


def digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(npy(sd[k]).astype(np.float32).tobytes())
    return h.hexdigest()


def maxdiff(a, b):
    return float((a - b).abs().max())


class Patched:
    """Temporarily replace attributes (torch.rand etc.) with queued returns."""

    def __init__(self, **kw):
        self.kw = kw
        self.saved = {}

    def __enter__(self):
        for name, fn in self.kw.items():
            modname, attr = name.rsplit("__", 1)
            m = {"torch": torch, "random": random}[modname]
            self.saved[name] = getattr(m, attr)
            setattr(m, attr, fn)
        return self

    def __exit__(self, *a):
        for name, old in self.saved.items():
            modname, attr = name.rsplit("__", 1)
            setattr({"torch": torch, "random": random}[modname], attr, old)


def queue_fn(values):
    q = list(values)

    def f(*a, **k):
        return q.pop(0).clone()

    return f


def tiny_pair(seed=11, perturb=True):
    torch.manual_seed(seed)
    ref = rgen.MelAudioGenerator(**TINY)
    torch.manual_seed(seed)
    orc = O.MelAudioGenerator(**TINY)
    if perturb:
        # move parameters away from their init so that every term is exercised
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in ref.named_parameters():
                if n.endswith("log_scale"):
                    p.copy_(torch.tensor(1.0) + 0.8 * (torch.rand((), generator=g) - 0.4))
                elif n.endswith("residual_scale.scale"):
                    p.copy_(0.3 + torch.rand(p.shape, generator=g))
                elif n.endswith(".bias") and p.ndim == 1 and "norm" not in n:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g))
                elif "act.weight" in n or "cond_mlp.1.weight" in n:
                    p.copy_(0.25 + 0.2 * torch.randn(p.shape, generator=g))
                elif p.ndim >= 2:
                    p.mul_(3.0)
    orc.load_state_dict(ref.state_dict())
    return ref, orc


# ------------------------------------------------------------------ case 1
def case_mel_frontend():
    """Pin the torchaudio restatement with the reference's own wav<->mel pair."""
    from scipy.io import wavfile
    out = {}
    for tag, wav, mel, kw in [
        ("24k", "test_data/wav/1089_134686_000002_000000.wav",
         "test_data/mel/1089_134686_000002_000000.pt", dict(sampling_rate=24000, n_fft=1024, hop_length=256, n_mels=100)),
        ("44k", "test_data/wav_44k/mixture.wav", "test_data/mel_44k_128band_512x/mixture.pt",
         dict(sampling_rate=44100, n_fft=2048, hop_length=512, n_mels=128)),
    ]:
        sr, pcm = wavfile.read(os.path.join(ref_shims.REFERENCE_ROOT, wav))
        assert sr == kw["sampling_rate"]
        stored = torch.load(os.path.join(ref_shims.REFERENCE_ROOT, mel))
        x = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
        if x.ndim == 2:  # stereo fixture: (T, ch) -> mono mean? check both
            cands = {"mean": x.mean(dim=1), "ch0": x[:, 0], "ch1": x[:, 1]}
        else:
            cands = {"mono": x}
        lm = O.LogMelSpectrogram(**kw)
        best = None
        for name, xx in cands.items():
            got = lm(xx[None])
            if got.shape == stored.shape:
                d = maxdiff(got, stored)
                if best is None or d < best[1]:
                    best = (name, d, xx, got)
            elif stored.shape[0] == 2 and x.ndim == 2 and name == "mean":
                got2 = lm(x.t())
                if got2.shape == stored.shape:
                    best = ("stereo", maxdiff(got2, stored), x.t(), got2)
        print(f"[mel_frontend {tag}] oracle-vs-stored fixture ({best[0]}): max|d| = {best[1]:.3e}, "
              f"rms = {float((best[3]-stored).pow(2).mean().sqrt()):.3e}")
        # commit a 1 s (24k) / 0.5 s (44k) excerpt: fp32 samples + the stored frames that only
        # depend on those samples
        nsamp = 24000 if tag == "24k" else 22050
        nfr = (nsamp - kw["n_fft"] // 2) // kw["hop_length"]
        src = best[2] if best[2].ndim == 1 else best[2][0]
        wave1 = src[:nsamp].clone()
        st = stored if stored.shape[0] == 1 else stored[:1]
        out[f"{tag}/wave"] = wave1.numpy()
        out[f"{tag}/logmel"] = npy(st[0, :, :nfr])
        chk = lm(wave1[None])[0, :, :nfr]
        print(f"   excerpt check: max|d| = {maxdiff(chk, st[0, :, :nfr]):.3e}")
    np.savez_compressed(os.path.join(OUT, "mel_frontend.npz"), **out)


# ------------------------------------------------------------------ case 2
def case_tiny_forward():
    """Tiny-width generator: leaf intermediates + infer outputs, ragged lengths."""
    ref, orc = tiny_pair()
    ref.eval(), orc.eval()
    g = torch.Generator().manual_seed(5)
    B, T = 2, 6000
    audio = 0.1 * torch.randn(B, T, generator=g)
    lens = torch.tensor([6000, 4600])
    mel = O.LogMelSpectrogram()(audio)  # (B,100,24)
    noise = 0.1 * torch.randn(B, T, generator=g)
    out = dict(sd_np(ref.state_dict()))
    out.update(audio=npy(audio), lens=lens.numpy(), mel=npy(mel), noise=npy(noise))
    with torch.no_grad():
        cond_r = ref.cond_encoder(mel)
        out["cond_enc"] = npy(cond_r)
        print("[tiny_forward] cond_enc oracle diff", maxdiff(orc.cond_encoder(mel), cond_r))
        tt = torch.full((B,), 0.25)
        for i, (er, eo) in enumerate(zip(ref.estimators, orc.estimators)):
            spec_r, lens_r = er.fft(noise, lens)
            packed = rmod.fft_to_real(spec_r)
            out[f"br{i}/stft_packed"] = npy(packed)
            cu = er.upsample_cond(cond_r, spec_r.shape[-1])
            mask = (~rmod.make_pad_mask(lens_r)).unsqueeze(1)
            dec = er.decoder(packed, cond=cu, t=tt, mask=mask)
            out[f"br{i}/decoder_out"] = npy(dec)
            x0 = er.decoder.in_norm(er.decoder.in_proj(packed))
            out[f"br{i}/in_norm"] = npy(x0)
            te = er.decoder.time_mlp(er.decoder.time_embed(tt))
            out[f"br{i}/time_embed"] = npy(te)
            cm = er.decoder.cond_mlp(cu)
            out[f"br{i}/block0"] = npy(er.decoder.blocks[0](x0, cond=cm, time_embed=te, mask=mask))
            y = er(noise, cond_r, tt, lens)
            out[f"br{i}/audio"] = npy(y)
            print(f"[tiny_forward] branch {i} oracle diff", maxdiff(eo(noise, cond_r, tt, lens), y))
        for n in (1, 2, 4):
            for tag, ln in (("ragged", lens), ("nolens", None)):
                if ln is None:
                    nz = 0.1 * torch.randn(B, mel.shape[2] * 256, generator=torch.Generator().manual_seed(9))
                    out["noise_nolens"] = npy(nz)
                else:
                    nz = noise
                yr = rgen.BaseAudioGenerator.infer(ref, noise=nz, cond=cond_r, audio_lens=ln,
                                                   n_timesteps=n, clamp_pred=(n == 4))
                yo = orc.infer(mel, ln, n, clamp_pred=(n == 4), noise=nz)
                out[f"infer_n{n}_{tag}"] = npy(yr)
                print(f"[tiny_forward] infer n={n} {tag} oracle diff {maxdiff(yo, yr):.3e} "
                      f"(rms out {float(yr.pow(2).mean().sqrt()):.3e})")
    np.savez_compressed(os.path.join(OUT, "tiny_forward.npz"), **out)


# ------------------------------------------------------------------ case 3
def case_tiny_stage1():
    """Stage-1 FM loss + all parameter grads (limiter always on, injected t / dropout)."""
    out = {}
    for tag, drop in (("nodrop", False), ("drop", True)):
        ref, orc = tiny_pair()
        ref.train(), orc.train()
        ref.branch_dropout = orc.branch_dropout = 0.05 if drop else 0.0
        g = torch.Generator().manual_seed(21)
        B, T = 2, 6000
        audio = 0.1 * torch.randn(B, T, generator=g)
        lens = torch.tensor([6000, 4600])
        mel = O.LogMelSpectrogram()(audio)
        noise = 0.1 * torch.randn(B, T, generator=g)
        t = torch.tensor([[0.3], [0.85]])
        u = torch.tensor([[0.01], [0.5]])  # sample 0 takes the dropout path
        idx = torch.tensor([0, 2])
        losses, grads = [], []
        for m, is_ref in ((ref, True), (orc, False)):
            m.zero_grad()
            with Patched(torch__rand=queue_fn([t, u]), torch__randint=queue_fn([idx]),
                         random__random=lambda: 0.0):
                cond = m.cond_encoder(mel)
                if is_ref:
                    loss = rgen.BaseAudioGenerator.forward(m, x0=noise, x1=audio, cond=cond,
                                                           audio_lens=lens)
                else:
                    loss = m.fm_loss(noise, audio, cond, lens)
            loss.backward()
            losses.append(loss.detach())
            grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
        worst = max(maxdiff(grads[0][n], grads[1][n]) / (grads[0][n].abs().max().item() + 1e-12)
                    for n in grads[0])
        print(f"[tiny_stage1 {tag}] loss ref {losses[0].item():.6f} oracle {losses[1].item():.6f}; "
              f"worst rel grad diff {worst:.3e}")
        if tag == "nodrop":
            out.update(sd_np(ref.state_dict()))
            out.update(audio=npy(audio), lens=lens.numpy(), mel=npy(mel), noise=npy(noise),
                       t=npy(t), drop_u=npy(u), drop_idx=idx.numpy())
        out[f"{tag}/loss"] = npy(losses[0])
        for n, gr in grads[0].items():
            out[f"{tag}/g/{n}"] = npy(gr)
    np.savez_compressed(os.path.join(OUT, "tiny_stage1.npz"), **out)


# ------------------------------------------------------------------ case 4
def grad_stats(named):
    return {n: np.array([float(p.grad.sum()), float(p.grad.abs().sum())], dtype=np.float64)
            for n, p in named if p.grad is not None}


TINY44 = dict(sampling_rate=44100, n_mels=128, mel_n_fft=2048, mel_hop_length=512,
              n_ffts=(1024, 512, 256), hop_lengths=(512, 256, 128), channels=(48, 32, 24),
              time_embed_channels=32, hidden_factor=3, num_layers=(2, 2, 2),
              cond_enc_channels=32, cond_enc_num_layers=1, loss_n_fft=2048, loss_hop_length=512)


def case_tiny_stage2(cfg=None, name="tiny_stage2", T=6000, short=4600, seed=31):
    """GAN D-step and G-step losses/grads: tiny generator + FULL discriminators
    (default torch init under a seed; regenerated from the seed by the oracle).
    With cfg=TINY44: the mel_44k_128band_512x_base geometry (BASELINE config 5) -- sr 44100 drives
    the 7 mel-recon filterbanks (gan.py:44-55), 128 mels / n_fft 2048 / hop 512 the front end."""
    global TINY
    out = {}
    saved = TINY
    if cfg is not None:
        TINY = cfg
    try:
        ref_g, orc_g = tiny_pair()
    finally:
        TINY = saved
    cfg = cfg or TINY
    ref_g.branch_dropout = orc_g.branch_dropout = 0.0
    torch.manual_seed(77)
    ref = rgan.GAN(ref_g, **rcfg.get_gan_config("gan_multi_scale_mel_recon"))
    torch.manual_seed(77)
    orc = O.GAN(orc_g)
    sd_ref_d = ref.discriminator.state_dict()
    sd_orc_d = {k: v for k, v in orc.discriminator.state_dict().items() if "spec_fn" not in k}
    sd_ref_d = {k: v for k, v in sd_ref_d.items() if "spec_fn" not in k}
    assert set(sd_ref_d) == set(sd_orc_d)
    print("[tiny_stage2] D init max diff (seeded default init)",
          max(maxdiff(sd_ref_d[k], sd_orc_d[k]) for k in sd_ref_d))
    out["d_digest"] = np.frombuffer(digest(sd_ref_d).encode(), dtype=np.uint8)
    out["d_seed"] = np.array(77)
    for k in ("0.discriminators.0.conv_post.weight", "1.discriminators.2.band_convs.1.0.weight"):
        out["dprobe/" + k] = npy(sd_ref_d[k])
    g = torch.Generator().manual_seed(seed)
    B = 2
    audio = 0.1 * torch.randn(B, T, generator=g)
    audio[1] *= 2.0
    mel = O.LogMelSpectrogram(cfg["sampling_rate"], cfg["mel_n_fft"], cfg["mel_hop_length"],
                              cfg["n_mels"])(audio)
    noise = 0.1 * torch.randn(B, T, generator=g)
    out.update(sd_np(ref_g.state_dict()))
    out.update(audio=npy(audio), mel=npy(mel), noise=npy(noise))
    w_d = (1.0, 0.1)
    w_g = (1.0, 0.1, 1.0, 0.1, 45.0)
    for n_steps, lens in ((1, torch.tensor([T, T])), (2, torch.tensor([T, short]))):
        tag = f"n{n_steps}"
        out[f"{tag}/lens"] = lens.numpy()
        for train_disc in (True, False):
            res = []
            for m, is_ref in ((ref, True), (orc, False)):
                m.zero_grad()
                with Patched(torch__randn=queue_fn([noise / 0.1]), random__random=lambda: 0.0):
                    if is_ref:
                        ls = m(cond=mel, audio=audio, audio_lens=lens, n_timesteps=n_steps,
                               train_disc=train_disc)
                    else:
                        ls = m(mel, audio, lens, n_steps, train_disc, noise=noise)
                ws = w_d if train_disc else w_g
                total = sum(w * l for w, l in zip(ws, ls))
                total.backward()
                res.append(([l.detach() for l in ls], m))
            lr, lo = res[0][0], res[1][0]
            step = "D" if train_disc else "G"
            print(f"[{name} {tag} {step}] ref losses {[round(float(x), 6) for x in lr]} "
                  f"oracle diff {max(abs(float(a - b)) for a, b in zip(lr, lo)):.3e}")
            out[f"{tag}/{step}/losses"] = np.array([float(x) for x in lr], dtype=np.float32)
            if train_disc:
                st = grad_stats(ref.discriminator.named_parameters())
                st_o = grad_stats(orc.discriminator.named_parameters())
                print("    D grad-stat worst rel diff",
                      max(abs(st[k][1] - st_o[k][1]) / (st[k][1] + 1e-12) for k in st))
                for k, v in st.items():
                    out[f"{tag}/D/gstat/{k}"] = v
                for k, p in ref.discriminator.named_parameters():
                    if p.numel() <= 4096:
                        out[f"{tag}/D/g/{k}"] = npy(p.grad)
            else:
                go = dict(orc.generator.named_parameters())
                worst = 0.0
                for k, p in ref.generator.named_parameters():
                    out[f"{tag}/G/g/{k}"] = npy(p.grad)
                    worst = max(worst, maxdiff(p.grad, go[k].grad) / (p.grad.abs().max().item() + 1e-12))
                print("    G grad worst rel diff", worst)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


# ------------------------------------------------------------------ case 4b
def case_tiny_switches():
    """The constructor switches no named config uses (generator.py:86-97,165-168,218,263):
    use_cond_encoder=False, pred_x1=False (velocity objective), branch_reduction="sum" -- stage-1
    loss + every parameter gradient and a 2-step Euler inference, from the reference alone."""
    cfg = dict(TINY, use_cond_encoder=False, pred_x1=False, branch_reduction="sum",
               branch_dropout=0.0)
    torch.manual_seed(13)
    ref = rgen.MelAudioGenerator(**cfg)
    g = torch.Generator().manual_seed(14)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.ndim >= 2:
                p.mul_(3.0)
            elif n.endswith(".bias") and "norm" not in n:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    assert not hasattr(ref, "cond_encoder")
    B, T = 2, 6000
    audio = 0.1 * torch.randn(B, T, generator=g)
    lens = torch.tensor([6000, 4600])
    mel = O.LogMelSpectrogram()(audio)
    noise = 0.1 * torch.randn(B, T, generator=g)
    t = torch.tensor([[0.3], [0.85]])
    out = dict(sd_np(ref.state_dict()))
    out.update(audio=npy(audio), lens=lens.numpy(), mel=npy(mel), noise=npy(noise), t=npy(t))
    ref.train()
    with Patched(torch__rand=queue_fn([t]), random__random=lambda: 0.0):
        loss = rgen.BaseAudioGenerator.forward(ref, x0=noise, x1=audio, cond=mel, audio_lens=lens)
    loss.backward()
    out["loss"] = npy(loss)
    for n, p in ref.named_parameters():
        out[f"g/{n}"] = npy(p.grad)
    ref.eval()
    with torch.no_grad():
        y = rgen.BaseAudioGenerator.infer(ref, noise=noise, cond=mel, audio_lens=lens, n_timesteps=2)
    out["infer_n2"] = npy(y)
    print(f"[tiny_switches] loss {float(loss):.6f}, infer rms {float(y.pow(2).mean().sqrt()):.4f}")
    np.savez_compressed(os.path.join(OUT, "tiny_switches.npz"), **out)


# ------------------------------------------------------------------ case 5
def case_full_width():
    """Full-width mel_24k_base, weights from seed (init equivalence proven by digest),
    reference test mel (first 64 frames) -> audio for n = 1, 4."""
    torch.manual_seed(1234)
    ref = rgen.MelAudioGenerator(**rcfg.get_generator_config("mel_24k_base"))
    torch.manual_seed(1234)
    orc = O.build_generator("mel_24k_base")
    d_ref, d_orc = digest(ref.state_dict()), digest(orc.state_dict())
    print("[full_width] init digest equal:", d_ref == d_orc)
    ref.eval(), orc.eval()
    mel = torch.load(os.path.join(ref_shims.REFERENCE_ROOT,
                                  "test_data/mel/1089_134686_000002_000000.pt"))[:, :, 40:104].contiguous()
    noise = 0.1 * torch.randn(1, 64 * 256, generator=torch.Generator().manual_seed(4321))
    out = dict(mel=npy(mel), seed=np.array(1234), noise_seed=np.array(4321),
               digest=np.frombuffer(d_ref.encode(), dtype=np.uint8))
    sd = ref.state_dict()
    for k in ("cond_encoder.in_proj.weight", "estimators.2.decoder.blocks.7.pwconv2.weight",
              "estimators.0.decoder.in_norm.bias"):
        out["probe/" + k] = npy(sd[k]).reshape(-1)[:256]
    with torch.no_grad():
        cond = ref.cond_encoder(mel)
        for n in (1, 4):
            yr = rgen.BaseAudioGenerator.infer(ref, noise=noise, cond=cond, n_timesteps=n,
                                               clamp_pred=True)
            yo = orc.infer(mel, None, n, True, noise=noise)
            out[f"audio_n{n}"] = npy(yr)
            print(f"[full_width] n={n} oracle-vs-ref rms diff "
                  f"{float((yo - yr).pow(2).mean().sqrt()):.3e}, out rms {float(yr.pow(2).mean().sqrt()):.3e}")
    np.savez_compressed(os.path.join(OUT, "full_width.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["mel", "fwd", "s1", "s2", "s2_44k", "sw", "full"]
    if "mel" in which:
        case_mel_frontend()
    if "fwd" in which:
        case_tiny_forward()
    if "s1" in which:
        case_tiny_stage1()
    if "s2" in which:
        case_tiny_stage2()
    if "s2_44k" in which:
        case_tiny_stage2(TINY44, "tiny_stage2_44k", T=11025, short=9000, seed=32)
    if "sw" in which:
        case_tiny_switches()
    if "full" in which:
        case_full_width()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
