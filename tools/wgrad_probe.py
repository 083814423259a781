"""Weight-gradient GEMMs (form 2, C[m,n] += sum_r A[r,m] B[r,n]) of the generator at B = 64: time by
split-K factor (atomic accumulation onto a zeroed output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
SHAPES = [(768, 2304, 6016), (512, 1536, 12032), (384, 1152, 24064), (1536, 512, 6016), (1024, 2560, 43648),
          (1024, 2560, 39168), (512, 640, 119680), (512, 640, 114944), (128, 160, 341376), (768, 576, 6016),
          (512, 320, 12032), (384, 192, 24064), (512, 512, 6016), (4096, 512, 6016), (1024, 5120, 43648), (1024, 5120, 39168)]
SWEEP = [int(v) for v in os.environ.get("SWEEP", "1,2,3,4,6,8,12").split(",")]
for M, N, K in SHAPES:
    dY = torch.randn(K, M, device=dev); X = torch.randn(K, N, device=dev)
    g = torch.zeros(M, N, device=dev)
    fl = 2.0 * M * N * K
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    auto = ops.split_for(K, tiles)
    res = []
    ta = timeit(lambda: ops.gemm(ops.mat(dY, K, M, M), ops.mat(X), g, form=2, atomic=True, split_k=auto))
    res.append(f"auto:{ta*1e6:.0f}us/{fl/ta/1e12:.0f}TF")
    for s in SWEEP:
        if K // s < 256: continue
        A = ops.mat(dY, K, M, M)
        t = timeit(lambda: ops.gemm(A, ops.mat(X), g, form=2, atomic=True, split_k=s))
        res.append(f"s{s}:{t*1e6:.0f}us/{fl/t/1e12:.0f}TF")
    print(f"M={M:5d} N={N:5d} K={K:6d} tiles={tiles:4d} auto split {auto}: " + " ".join(res), flush=True)
