"""Which Python call sites issue the small launches of a stage-2 step?  Counts f2g_* calls by (entry point,
caller line) over one D + G step after warm-up."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import flow2gan_amd
from flow2gan_amd import ops, _lib
from flow2gan_amd.models.config import get_gan_config, get_generator_config
from flow2gan_amd.models.gan import GAN
import bench

dev = torch.device("cuda", 0)
gcfg = get_generator_config("mel_24k_base")
torch.manual_seed(1234)
gen = flow2gan_amd.MelAudioGenerator(**gcfg)
gen.branch_dropout = 0.0
gan = GAN(gen, **get_gan_config("gan_multi_scale_mel_recon")).to(dev)
logmel = flow2gan_amd.LogMelSpectrogram(24000, gcfg["mel_n_fft"], gcfg["mel_hop_length"], gcfg["n_mels"]).to(dev)
B, T = 8, 24000
a = bench.synthetic_batch(B, T, 1234, dev)
lens = torch.full((B,), T, dtype=torch.int64)


def step():
    for p in gan.parameters():
        p.grad = None
    mp, mr = gan(logmel(a), a, lens, 1, True)
    (mp + 0.1 * mr).backward()
    for p in gan.parameters():
        p.grad = None
    ls = gan(logmel(a), a, lens, 1, False)
    sum(w * l for w, l in zip(bench.G_WEIGHTS, ls)).backward()


step(); step()
torch.cuda.synchronize()
counts = collections.Counter()
orig = _lib.call
WATCH = set(sys.argv[1:]) or {"f2g_permute4", "f2g_fill", "f2g_zero_halo", "f2g_copy3", "f2g_colsum"}


def spy(name, *args):
    if name in WATCH:
        st = traceback.extract_stack(limit=6)[:-1]
        site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-4:]))
        counts[(name, site)] += 1
    return orig(name, *args)


_lib.call = spy
ops.call = spy
import flow2gan_amd.fused as F1, flow2gan_amd.fused_disc as F2
for m in (F1, F2):
    if hasattr(m, "call"):
        m.call = spy
step()
torch.cuda.synchronize()
for (name, site), c in counts.most_common(40):
    print(f"{c:5d} {name:16s} {site}")
