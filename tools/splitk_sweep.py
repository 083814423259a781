"""Sweep the split-K factor of forward / data-gradient GEMMs at the mid-size generator shapes
(B=64): TFLOP/s per split; split 0 = the library's own choice (auto_split in gemm.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
shapes = [  # (rows, K, N) of the forward GEMM
    (6016, 2304, 768), (6016, 768, 2304), (12032, 1536, 512), (12032, 512, 1536),
    (24064, 1152, 384), (24064, 384, 1152), (6016, 512, 6144), (6016, 1536, 512), (6016, 512, 1536),
    (38016, 2560, 1024), (19008, 5120, 1024), (113920, 640, 512),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
splits = (1, 0, 2, 3, 4, 6, 8)
print("form  rows     K     N  | " + "  ".join(f"s={s:d}" + ("(auto)" if s == 0 else "      ")[:6] for s in splits))
for R, K, N in shapes:
    A = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02
    bias = torch.randn(N, device=dev)
    out = torch.empty(R, N, device=dev); gA = torch.empty(R, K, device=dev)
    ref = A @ W.t() + bias
    fl = 2.0 * R * K * N
    for form in (0, 1):
        cells = []
        for s in splits:
            if form == 0:
                fn = lambda: ops.gemm(ops.mat(A), ops.mat(W), out, bias=bias, split_k=s)
            else:
                fn = lambda: ops.gemm(ops.mat(out), ops.mat(W), gA, form=1, split_k=s)
            t = timeit(fn)
            cells.append(f"{fl / t / 1e12:6.1f}")
        if form == 0:
            err = float((out - ref).abs().max() / ref.abs().max())
            cells.append(f"err {err:.1e}")
        print(f"{form:4d} {R:6d} {K:5d} {N:5d}  | " + "      ".join(cells))
