import sys, os
sys.path.insert(0, "/root/repo")
import torch
from flow2gan_amd import ops, _lib
dev = "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mode in ("fp32", "bf16x6"):
    ops.set_gemm_precision(mode)
    for sk in (1, 0):
        _lib.set_option("streamk", sk)
        for M, N, K in [(64, 1536, 512), (64, 512, 1536), (64, 6144, 512), (64, 3072, 512)]:
            a = torch.randn(M, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.05)
            b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
            t = timeit(lambda: ops.gemm(ops.mat(a), ops.mat(w), out, bias=b))
            print(f"{mode} streamk={sk} {M}x{N}x{K}: {t:6.1f} us  path {ops.L.lib.f2g_gemm_last_path()}")
