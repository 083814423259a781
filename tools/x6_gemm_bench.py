"""The generator's plain-matrix GEMMs through ops.gemm: exact fp32 (lean kernel) against bf16x6 (three-piece
images, six products) INCLUDING the image pass over the activation operand; the weight image is cached."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
ops.X6_MIN_K, ops.X6_MIN_ROWS = 32, 1     # (the model's profitability thresholds off: every shape on the kernel)
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
shapes = [(6016, 2304, 768), (6016, 768, 2304), (12032, 1536, 512), (12032, 512, 1536), (24064, 1152, 384),
          (24064, 384, 1152), (6016, 6144, 512), (6016, 512, 1536)]
print("     M     N     K | fp32 us (TF) | bf16x6 us (TF), of which image pass of A us | x6 kernel alone us (TF)")
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.05)
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    ops.set_gemm_precision("fp32")
    t32 = timeit(lambda: ops.gemm(ops.mat(a), ops.mat(w), out, bias=bias))
    ops.set_gemm_precision("bf16x6")
    t6 = timeit(lambda: ops.gemm(ops.mat(a), ops.mat(w), out, bias=bias))
    img = torch.empty(M * K * 3, device=dev, dtype=torch.bfloat16)
    ts = timeit(lambda: ops.call("f2g_split_bf16x3", ops.ptr(img), ops.ptr(a), K, M, K))
    print(f"{M:6d} {N:5d} {K:5d} | {t32*1e6:7.1f} ({fl/t32/1e12:5.1f}) | {t6*1e6:7.1f} ({fl/t6/1e12:5.1f}), {ts*1e6:6.1f} | {(t6-ts)*1e6:7.1f} ({fl/(t6-ts)/1e12:5.1f})")
ops.set_gemm_precision("fp32")
