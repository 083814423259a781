"""Fixed cost of a lean-GEMM tile: one full round (512 tiles of 128 x 128, two per CU) at growing K -> t(K) = a + b K.
Also 1024 and 2048 tiles (two / four rounds) and the epilogue variants, at the generator's K."""
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

os.environ["F2G_SKFIX"] = "0"
for R, N in [(8192, 1024), (16384, 1024), (32768, 1024), (4096, 1024)]:
    for K in [64, 128, 256, 384, 768, 1536, 3072]:
        A = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
        out = torch.empty(R, N, device="cuda"); bias = torch.randn(N, device="cuda")
        us = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias, split_k=1))
        print("tiles %5d  K %5d  %8.1f us  %6.1f TF   %.2f us/slab" % (R // 128 * (N // 128), K, us, 2.0 * R * K * N / us / 1e6, us / (K / 32)), flush=True)
