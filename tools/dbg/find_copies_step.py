"""Which torch operators of a GAN stage-2 train step (D-step + G-step) make device copies / small ATen
kernels, and from where?  (aten::copy_ between device tensors = the __amd_rocclr_copyBuffer launches.)"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
import flow2gan_amd
from flow2gan_amd import ops
from flow2gan_amd.models.config import get_gan_config, get_generator_config
from flow2gan_amd.models.gan import GAN

dev = "cuda"
torch.manual_seed(0)
B = int(os.environ.get("B", "8"))
gcfg = get_generator_config("mel_24k_base")
gen = flow2gan_amd.MelAudioGenerator(**gcfg)
gen.branch_dropout = 0.0
gan = GAN(gen, **get_gan_config("gan_multi_scale_mel_recon")).to(dev)
logmel = flow2gan_amd.LogMelSpectrogram(24000, 1024, 256, 100).to(dev)
audio = (0.1 * torch.randn(B, 24000)).clamp_(-1, 1).to(dev)
lens = torch.full((B,), 24000, dtype=torch.int64)
ops.set_gemm_precision("bf16x6")


def step():
    gan.zero_grad(set_to_none=True)
    mp, mr = gan(logmel(audio), audio, lens, 1, True)
    (1.0 * mp + 0.1 * mr).backward()
    gan.zero_grad(set_to_none=True)
    ls = gan(logmel(audio), audio, lens, 1, False)
    sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls)).backward()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::view", "aten::reshape", "aten::as_strided",
                                                         "aten::slice", "aten::select", "aten::empty_like", "aten::_unsafe_view",
                                                         "aten::detach", "aten::alias", "aten::empty_strided", "aten::stride",
                                                         "aten::size", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense",
                                                         "aten::t", "aten::transpose", "aten::permute", "aten::unsqueeze", "aten::squeeze",
                                                         "aten::expand", "aten::flatten", "aten::view_as", "aten::narrow",
                                                         "aten::result_type", "aten::lift_fresh", "aten::resolve_conj", "aten::resolve_neg"):
        st = [s for s in ev.stack if "flow2gan_amd" in s or "find_copies_step" in s]
        cnt[(ev.name, st[0] if st else "(autograd engine / no python frame)")] += 1
for (n, where), c in cnt.most_common(60):
    print(c, n, where)
