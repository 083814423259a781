"""Plain-operand MPD weight gradient (1024 x 5120 output) by reduction length and split: is the spread between the
periods (110-131 TFLOP/s in the step) the row count (chunk offsets at large powers of two) or the window operand?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops
dev = "cuda"
def timeit(fn, n=6):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
M, N = 1024, 5120
for K in (39168, 39552, 40960, 41024, 42112, 43648, 43680):
    dY = torch.randn(K, M, device=dev); X = torch.randn(K, N, device=dev)
    g = torch.zeros(M, N, device=dev)
    fl = 2.0 * M * N * K
    tiles = 8 * 40
    auto = ops.split_for(K, tiles)
    res = []
    for s in (auto, 3, 5, 7, 8, 9, 11, 13, 16):
        t = timeit(lambda: ops.gemm(ops.mat(dY, K, M, M), ops.mat(X), g, form=2, atomic=True, split_k=s))
        res.append("s%d:%.0f" % (s, fl / t / 1e12))
    print("K=%6d auto %2d path %d: " % (K, auto, ops.L.lib.f2g_gemm_last_path()) + " ".join(res), flush=True)
