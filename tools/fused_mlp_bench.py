"""Fused pwconv1 -> PReLU -> pwconv2 kernel (csrc/fusedmlp.hip) against the two lean bf16 GEMM launches
it replaces, at the block shapes of mel_24k_base with B = 64: time per block-layer and TFLOP/s."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops

dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


ops.set_gemm_precision("bf16")
for rows, C in [(6016, 768), (12032, 512), (24064, 384), (6016, 512)]:
    H = 3 * C
    z = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    w1 = torch.nn.Parameter(torch.randn(H, C, device=dev) * 0.03)
    w2 = torch.nn.Parameter(torch.randn(C, H, device=dev) * 0.03)
    b1, al = torch.randn(H, device=dev) * 0.1, torch.full((H,), 0.25, device=dev)
    b2, gam = torch.randn(C, device=dev) * 0.1, torch.ones(C, device=dev)
    x = torch.randn(rows, C, device=dev)
    out = torch.empty(rows, C, device=dev)
    a = torch.empty(rows, H, device=dev, dtype=torch.bfloat16)
    wp = ops.mlp_pack(w1, w2)
    fl = 4.0 * rows * C * H

    def two():
        ops.gemm(ops.mat(z, rows, C, split=2), ops.mat(w1), a, bias=b1, prelu=al, split_k=1)
        ops.gemm(ops.mat(a, rows, H, split=2), ops.mat(w2), out, bias=b2, res=x, gamma=gam)

    def fused(parts=0):
        ops.fused_mlp(z, wp, b1, al, b2, x, gam, out, rows, C, H, parts=parts)

    t2, t1 = timeit(two), timeit(fused)
    sweep = " ".join(f"p{n}:{timeit(lambda n=n: fused(n))*1e6:.0f}" for n in (1, 2, 3, 4))
    print(f"rows={rows:6d} C={C:4d} H={H:5d}: two lean GEMMs {t2*1e6:7.1f} us = {fl/t2/1e12:6.1f} TF | "
          f"fused {t1*1e6:7.1f} us = {fl/t1/1e12:6.1f} TF ({fl/t1/2.5e15:.3f} of the bf16 peak) | parts sweep (us) {sweep}", flush=True)
ops.set_gemm_precision("fp32")
