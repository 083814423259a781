import sys, os, collections, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import flow2gan_amd
from flow2gan_amd import ops, _lib
from flow2gan_amd.models.config import get_generator_config
ops.set_gemm_precision("bf16")
m = flow2gan_amd.MelAudioGenerator(**get_generator_config("mel_24k_base")).cuda().eval()
mel = torch.randn(64, 100, 94, device="cuda")
with torch.no_grad():
    m.infer(mel, None, 1)
    cnt = collections.Counter()
    orig = ops.call
    def call(name, *a):
        if name in ("f2g_to_bf16", "f2g_split_bf16"):
            st = traceback.extract_stack(limit=8)
            key = (name, a[2], " <- ".join(f"{f.name}:{f.lineno}" for f in st[-6:-1]))
            cnt[key] += 1
        return orig(name, *a)
    ops.call = call
    m.infer(mel, None, 1)
    ops.call = orig
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1] * kv[0][1])[:25]:
    print(v, k)
