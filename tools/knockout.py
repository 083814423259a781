"""Marginal cost of the step's components in the LANED schedule: the stage-2 step of bench.py with one
component knocked out at a time (its term replaced by a constant that is connected to the generated audio
with a zero gradient, so everything else still runs).  KO = comma list of {mpd, mrd, mel}; MODE = fp32 |
bf16x6.  What a component costs in the laned step = step(full) - step(without it); the sum of its kernels'
isolated durations (profiles/*_kernel_stats.txt) says what it would cost alone on the chip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flow2gan_amd
from flow2gan_amd import ops
from flow2gan_amd.models.config import get_gan_config, get_generator_config
from flow2gan_amd.models.gan import GAN
import bench

ko = set(filter(None, os.environ.get("KO", "").split(",")))
ops.set_gemm_precision(os.environ.get("MODE", "fp32"))
dev = torch.device("cuda", 0)
gcfg = get_generator_config("mel_24k_base")
torch.manual_seed(1234)
gen = flow2gan_amd.MelAudioGenerator(**gcfg)
gen.branch_dropout = 0.0
gan = GAN(gen, **get_gan_config("gan_multi_scale_mel_recon")).to(dev)
logmel = flow2gan_amd.LogMelSpectrogram(24000, gcfg["mel_n_fft"], gcfg["mel_hop_length"], gcfg["n_mels"]).to(dev)
B, T = 64, 24000
a_d, a_g = bench.synthetic_batch(B, T, 1234, dev), bench.synthetic_batch(B, T, 4321, dev)
lens = torch.full((B,), T, dtype=torch.int64)


def zero_pair(fake):
    z = fake.sum() * 0.0 if fake.requires_grad else torch.zeros((), device=fake.device)
    return z, z


if "mpd" in ko:
    GAN._mp_terms = lambda self, real, fake, td: zero_pair(fake)
if "mrd" in ko:
    GAN._mr_terms = lambda self, real, fake, td: zero_pair(fake)
if "mel" in ko:
    GAN.mel_recon_loss = lambda self, real, fake: zero_pair(fake)[0]


def step():
    for p in gan.parameters():
        p.grad = None
    mp, mr = gan(logmel(a_d), a_d, lens, 1, True)
    tot = 1.0 * mp + 0.1 * mr
    if tot.requires_grad:
        tot.backward()
    for p in gan.parameters():
        p.grad = None
    ls = gan(logmel(a_g), a_g, lens, 1, False)
    sum(w * l for w, l in zip(bench.G_WEIGHTS, ls)).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 6
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"MODE={os.environ.get('MODE', 'fp32')} KO={','.join(sorted(ko)) or '-'}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per step")
