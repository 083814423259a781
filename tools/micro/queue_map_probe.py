"""Which torch streams share a hardware queue?  Pairs of streams (by creation order) running the same ragged lean
GEMM: a pair on two queues co-runs (~0.58 of serial for 6016 x 768 x 2304), a pair on one queue cannot."""
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

S = [torch.cuda.Stream() for _ in range(12)]
fa, fb = mk(6016, 2304, 768), mk(6016, 2304, 768)
fa(); fb()
n = 16
def serial():
    with torch.cuda.stream(S[0]):
        for _ in range(n): fa(); fb()
t1 = min(wall(serial) for _ in range(3))
print("serial on one stream: %.2f ms" % t1)
for i in range(0, 4):
    row = []
    for j in range(i + 1, 12):
        def run():
            for _ in range(n):
                with torch.cuda.stream(S[i]): fa()
                with torch.cuda.stream(S[j]): fb()
        row.append("%d:%.2f" % (j, min(wall(run) for _ in range(2)) / t1))
    print("stream %d with " % i + "  ".join(row), flush=True)
# default stream with each
row = []
for j in range(12):
    def run():
        for _ in range(n):
            fa()
            with torch.cuda.stream(S[j]): fb()
    row.append("%d:%.2f" % (j, min(wall(run) for _ in range(2)) / t1))
print("default stream with " + "  ".join(row))
