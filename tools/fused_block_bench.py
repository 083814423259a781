"""f2g_fused_block (dwnorm prologue + fused MLP) against dwnorm_fwd + fused_mlp, mel_24k_base block
shapes at B = 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
ops.set_gemm_precision("bf16")
B = 64
for C, F, up in ((768, 94, 1), (512, 188, 2), (384, 376, 4)):
    rows, H, Fc, NC = B * F, 3 * C, F // up, 8 * C
    x = torch.randn(rows, C, device=dev); z = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
    w = torch.randn(C, 1, 7, device=dev) * 0.3; b = torch.zeros(C, device=dev)
    beta = torch.randn(C, device=dev) * 0.01; ls = torch.ones(1, device=dev)
    cp = torch.randn(B * Fc, NC, device=dev); te = torch.randn(B, NC, device=dev) * 0.1
    w1 = torch.nn.Parameter(torch.randn(H, C, device=dev) * 0.03); w2 = torch.nn.Parameter(torch.randn(C, H, device=dev) * 0.03)
    b1, al = torch.randn(H, device=dev) * 0.1, torch.full((H,), 0.25, device=dev)
    b2, gam = torch.randn(C, device=dev) * 0.1, torch.ones(C, device=dev)
    out = torch.empty(rows, C, device=dev)
    wp = ops.mlp_pack(w1, w2)
    args = (B, F, C, 7, None, w, b, beta, ls)
    t_dw = timeit(lambda: ops.dwnorm_fwd(x, z, *args, cp, NC, Fc, up, 0, te, NC, 0, z_format=2))
    t_mlp = timeit(lambda: ops.fused_mlp(z, wp, b1, al, b2, x, gam, out, rows, C, H))
    t_blk = timeit(lambda: ops.fused_block(x, *args, wp, b1, al, b2, gam, out, H, cp, NC, Fc, up, 0, te, NC, 0))
    print(f"C={C} rows={rows}: dwnorm {t_dw*1e6:.1f} us + fused MLP {t_mlp*1e6:.1f} us = {(t_dw+t_mlp)*1e6:.1f} us | fused block {t_blk*1e6:.1f} us", flush=True)
