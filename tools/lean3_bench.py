"""Lean kernel, exact fp32 vs split-bf16 (pre-split operands): TFLOP/s (fp32-equivalent) at the hot
forward shapes of the stage-2 step, with the activation split pass (a) included, (b) excluded."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops

dev = "cuda"
shapes = [(6016, 768, 2304), (6016, 2304, 768), (12032, 512, 1536), (12032, 1536, 512),
          (24064, 384, 1152), (24064, 1152, 384), (38016, 5120, 1024), (38016, 2560, 1024),
          (76032, 640, 512)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for R, K, N in shapes:
    A = torch.randn(R, K, device=dev)
    W = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.02)
    out = torch.empty(R, N, device=dev)
    fl = 2.0 * R * K * N
    ops.set_gemm_precision("fp32")
    t0 = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out, split_k=1))
    ops.set_gemm_precision("bf16x3")
    t1 = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out, split_k=1))
    path = ops.L.lib.f2g_gemm_last_path()
    As = ops._split_operand(ops.mat(A))
    Ws = ops._split_operand(ops.mat(W))
    t2 = timeit(lambda: ops.gemm(As, Ws, out, split_k=1))
    t3 = timeit(lambda: ops.split_bf16(A))
    os.environ["X"] = "1"
    ops.LEAN_SPLIT = False
    t4 = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out, split_k=1))
    ops.LEAN_SPLIT = True
    print(f"R={R:6d} K={K:5d} N={N:5d} fp32 {fl/t0/1e12:6.1f} TF ({t0*1e6:6.0f} us) | split+lean3 "
          f"{fl/t1/1e12:6.1f} TF ({t1*1e6:6.0f} us, path {path}) | lean3 alone {fl/t2/1e12:6.1f} TF "
          f"({t2*1e6:6.0f} us) | split pass {t3*1e6:5.0f} us = {8.0*R*K/t3/1e9:6.0f} GB/s | generic b3 "
          f"{fl/t4/1e12:6.1f} TF", flush=True)
ops.set_gemm_precision("fp32")
