"""n streams, each a chain of the same lean GEMM (rows, K, N): wall time against the serial chain -- how far does
co-running go beyond two streams, on big (MPD) and ragged (generator) grids?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops

def mk(R, K, N):
    A = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
    out = torch.empty(R, N, device="cuda")
    return lambda: ops.gemm(ops.mat(A), ops.mat(W), out, split_k=1)

def wall(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3

streams = [torch.cuda.Stream() for _ in range(8)]
for shape, n in [((38016, 5120, 1024), 6), ((38016, 2048, 512), 12), ((24064, 1152, 384), 16), ((6016, 2304, 768), 16)]:
    fs = [mk(*shape) for _ in range(5)]
    for f in fs: f()
    for ns in (1, 2, 3, 4, 5):
        def run():
            for _ in range(n):
                for i in range(ns):
                    with torch.cuda.stream(streams[i]): fs[i]()
        t = min(wall(run) for _ in range(3))
        if ns == 1: t1 = t
        print("%-22s %d stream(s) x %2d launches: %8.2f ms  = %.3f of serial" % (shape, ns, n, t, t / (t1 * ns)), flush=True)
