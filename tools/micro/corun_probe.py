"""Do two lean GEMMs on two streams finish sooner than one after the other?  (what the launch lanes can and
cannot recover)  Each case: n launches of A then n of B on one stream, against A on stream 1 beside B on stream 2."""
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

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 20
cases = [((24064, 1152, 384), (12032, 1536, 512)), ((6016, 2304, 768), (6016, 2304, 768)),
         ((38016, 5120, 1024), (38016, 5120, 1024)), ((24064, 384, 1152), (6016, 2304, 768)),
         ((8192, 384, 1024), (8192, 384, 1024)), ((38016, 5120, 1024), (6016, 2304, 768))]
for sa, sb in cases:
    fa, fb = mk(*sa), mk(*sb)
    for f in (fa, fb):
        for _ in range(3): f()
    def serial():
        for _ in range(n): fa()
        for _ in range(n): fb()
    def inter():
        for _ in range(n): fa(); fb()
    def corun():
        for _ in range(n):
            with torch.cuda.stream(s1): fa()
            with torch.cuda.stream(s2): fb()
    ts = min(wall(serial) for _ in range(3)); ti = min(wall(inter) for _ in range(3)); tc = min(wall(corun) for _ in range(3))
    print("%-22s + %-22s  serial %7.2f ms  interleaved %7.2f  two streams %7.2f  (%.3f of serial)" % (sa, sb, ts, ti, tc, tc / ts), flush=True)
