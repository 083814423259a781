"""Fixed cost of a tile of the six-product GEMM kernels: full rounds of 128 x 128 tiles at growing K.
F2G_GEMM=bf16x6 F2G_X6_MIN_K=32 F2G_X6F=0|1 python3 tools/micro/x6_tile_overhead.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("X6F", ops.X6F, "MIN_K", ops.X6_MIN_K, "precision", ops.GEMM_PRECISION)
for R, N in [(8192, 1024), (32768, 1024)]:
    for K in [128, 256, 384, 768, 1536, 3072]:
        A = torch.randn(R, K, device="cuda"); W = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.02)
        out = torch.empty(R, N, device="cuda"); bias = torch.randn(N, device="cuda")
        Aop = ops.mat(A)
        if ops.X6F != 1:
            Ai = ops._x3_operand(Aop)      # image made once, outside the timed launches
        us = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias, split_k=1))
        path = ops.L.lib.f2g_gemm_last_path()
        print("tiles %5d  K %5d  %8.1f us  %6.1f TF   path %d" % (R // 128 * (N // 128), K, us, 2.0 * R * K * N / us / 1e6, path), flush=True)
