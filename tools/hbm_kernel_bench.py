"""Achieved HBM bandwidth of the HBM-bound ConvNeXt kernels at the mel_24k_base branch shapes
(B=64): algorithmic bytes / HIP-event time, against the 8 TB/s spec (6.3 TB/s measured copy)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
B = 64
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
print(f"{'kernel':14s} {'C':>4s} {'F':>4s} {'up':>2s} {'us':>8s} {'alg MB':>8s} {'GB/s':>8s} {'% of 8TB/s':>10s}")
ISO = {}      # kernel -> [launches, bytes, seconds]: what bench.py prints as roofline.hbm_class_isolated


def note(name, mb, t):
    a = ISO.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += mb * 1e6
    a[2] += t
for C, F, up in ((768, 94, 1), (512, 188, 2), (384, 376, 4)):
    rows = B * F
    Fc = F // up
    NC = 8 * C
    x = torch.randn(rows, C, device=dev); z = torch.empty(rows, C, device=dev)
    w = torch.randn(C, 1, 7, device=dev) * 0.3; b = torch.zeros(C, device=dev)
    beta = torch.randn(C, device=dev) * 0.01; ls = torch.ones(1, device=dev)
    cp = torch.randn(B * Fc, NC, device=dev); te = torch.randn(B, NC, device=dev) * 0.1
    gz = torch.randn(rows, C, device=dev); du = torch.empty(rows, C, device=dev); gx = torch.empty(rows, C, device=dev)
    gcp = torch.zeros(B * Fc, NC, device=dev); gte = torch.zeros(B, NC, device=dev)
    gb = torch.zeros(C, device=dev); gl = torch.zeros(1, device=dev); gw = torch.zeros(C, 1, 7, device=dev)
    gbb = torch.zeros(C, device=dev); gg = torch.zeros(C, device=dev); gam = torch.ones(C, device=dev)
    args = (B, F, C, 7, None, w, b, beta, ls)
    t = timeit(lambda: ops.dwnorm_fwd(x, z, *args, cp, NC, Fc, up, 0, te, NC, 0))
    mb = rows * C * 4 * (2 + 1.0 / up) / 1e6
    note("dwnorm_fwd", mb, t)
    print(f"{'dwnorm_fwd':14s} {C:4d} {F:4d} {up:2d} {t*1e6:8.1f} {mb:8.1f} {mb/1e3/t:8.0f} {100*mb/1e3/t/8000:10.1f}")
    t = timeit(lambda: ops.dwnorm_bwd(x, gz, du, *args, cp, NC, Fc, up, 0, te, NC, 0, g_cproj=gcp, g_te=gte, g_beta=gb, g_log_scale=gl, g_cproj_store=True))
    mb = rows * C * 4 * (3 + 2.0 / up) / 1e6
    note("dwnorm_bwd", mb, t)
    print(f"{'dwnorm_bwd':14s} {C:4d} {F:4d} {up:2d} {t*1e6:8.1f} {mb:8.1f} {mb/1e3/t:8.0f} {100*mb/1e3/t/8000:10.1f}")
    t = timeit(lambda: ops.dwconv_bwd(du, x, gx, B, F, C, 7, None, w, gres=gz, gamma=gam, g_w=gw, g_b=gbb, g_gamma=gg))
    mb = rows * C * 4 * 4 / 1e6
    note("dwconv_bwd", mb, t)
    print(f"{'dwconv_bwd':14s} {C:4d} {F:4d} {up:2d} {t*1e6:8.1f} {mb:8.1f} {mb/1e3/t:8.0f} {100*mb/1e3/t/8000:10.1f}")

if len(sys.argv) > 1:
    # json for bench.py (roofline.hbm_class_isolated): the same kernels as its in-step hbm_class rows, launched
    # back to back at the branch shapes of mel_24k_base (B = 64), stamped with the library version
    import json
    from flow2gan_amd import _lib
    json.dump({"source": "tools/hbm_kernel_bench.py: 20 back-to-back launches per kernel and branch shape (B = 64), "
                         "algorithmic bytes / HIP-event time against 8 TB/s; in the step the same kernels are bracketed "
                         "one launch at a time together with their parameter-gradient reduction launch",
               "lib_version": _lib.version(),
               "kernels": {k: {"shapes": v[0], "GB_per_pass": round(v[1] / 1e9, 3), "achieved_GBps": round(v[1] / v[2] / 1e9, 1),
                               "frac": round(v[1] / v[2] / 8.0e12, 3)} for k, v in sorted(ISO.items())}},
              open(sys.argv[1], "w"), indent=1)
