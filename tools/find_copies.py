"""Which Python lines of the inference path make torch issue device copies (aten::copy_ / clone)?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from torch.profiler import ProfilerActivity, profile
import flow2gan_amd
from flow2gan_amd import ops
from flow2gan_amd.models.config import get_generator_config

dev = "cuda"
torch.manual_seed(0)
gen = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_24k_base")).to(dev).eval()
mel = torch.randn(8, 100, 94, device=dev)
ops.set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
with torch.no_grad():
    gen.infer(mel, None, 4)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        gen.infer(mel, None, 1)
        torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy",
                   "aten::fill_", "aten::zero_", "aten::mul", "aten::add", "aten::empty_strided"):
        st = [s for s in ev.stack if "flow2gan_amd" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (n, where), c in cnt.most_common(30):
    print(c, n, where)
