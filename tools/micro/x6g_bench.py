"""Forward GEMMs of the generator's blocks with CACHED weights (nn.Parameter: the in-kernel-split kernels with the
weight image -- gemm_x6f_kernel<true> / gemm_x6g_kernel), HIP-event time per launch.  F2G_GEMM=bf16x6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops
out = []
for R, K, N in ((24064, 384, 1152), (24064, 1152, 384), (12032, 512, 1536), (12032, 1536, 512), (6016, 768, 2304), (6016, 2304, 768),
                (6016, 512, 1536), (6016, 1536, 512)):
    A = torch.randn(R, K, device="cuda"); W = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.02)
    o = torch.empty(R, N, device="cuda")
    for _ in range(3): ops.gemm(ops.mat(A), ops.mat(W), o)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.gemm(ops.mat(A), ops.mat(W), o)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 50
    out.append(f"{R}x{N}x{K}: {us:6.1f} us {2.0 * R * K * N / us / 1e6:6.1f} TF")
print(" | ".join(out))
