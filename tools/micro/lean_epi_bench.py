"""Time the exact-fp32 lean GEMM on the generator's shapes with its epilogue variants (plain, two-output
PReLU, residual * gamma) -- run once per lab library (tools/micro/lean_epi_lab.sh, F2G_LIB_PATH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = [(24064, 384, 1152, "prelu2"), (24064, 1152, 384, "res"), (24064, 384, 1152, "plain"),
          (24064, 1152, 384, "plain"), (12032, 1536, 512, "res"), (12032, 512, 1536, "prelu2"),
          (6016, 2304, 768, "res"), (6016, 768, 2304, "prelu2"), (38016, 2560, 1024, "plain"),
          (113920, 640, 512, "plain")]
for R, K, N, ep in shapes:
    A = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
    out = torch.empty(R, N, device="cuda"); out2 = torch.empty(R, N, device="cuda")
    res = torch.randn(R, N, device="cuda"); gam = torch.randn(N, device="cuda"); sl = torch.rand(N, device="cuda")
    bias = torch.randn(N, device="cuda")
    if ep == "plain": fn = lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias)
    elif ep == "prelu2": fn = lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias, prelu=sl, prelu_out=out2)
    else: fn = lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias, res=res, gamma=gam)
    us = timeit(fn)
    print("%6d x %4d x %4d %-6s %8.1f us  %6.1f TF" % (R, N, K, ep, us, 2.0 * R * K * N / us / 1e6), flush=True)
