"""Launch one GEMM shape repeatedly (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
R, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6016, 768, 2304)))
form = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
out = torch.empty(R, N, device="cuda"); gA = torch.empty(R, K, device="cuda"); gW = torch.zeros(N, K, device="cuda")
for _ in range(10):
    if form == 0: ops.gemm(ops.mat(A), ops.mat(W), out)
    elif form == 1: ops.gemm(ops.mat(out), ops.mat(W), gA, form=1)
    else: ops.wgrad(out, N, N, ops.mat(A), gW)
torch.cuda.synchronize()
