"""Timing of the split-bf16 lean kernel (pre-split operands, one MPD-shaped GEMM) for the library
given by F2G_LIB_PATH (lab builds of tools/micro/build_variants.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
ops.set_gemm_precision("bf16x3")
R, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (38016, 2560, 1024)))
A = torch.randn(R, K, device="cuda"); W = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.02)
out = torch.empty(R, N, device="cuda")
As, Ws = ops._split_operand(ops.mat(A)), ops._split_operand(ops.mat(W))
for _ in range(3): ops.gemm(As, Ws, out, split_k=1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ops.gemm(As, Ws, out, split_k=1)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"{os.environ.get('F2G_LIB_PATH', 'product library'):50s} {t*1e6:8.1f} us  {2.0*R*K*N/t/1e12:7.1f} TFLOP/s fp32-equivalent")
