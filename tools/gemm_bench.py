"""Time f2g_gemm at the hot shapes of mel_24k_base (B=64): TFLOP/s per form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
shapes = [  # (rows, K, N) of the forward GEMM
    (6016, 768, 2304), (6016, 2304, 768), (12032, 512, 1536), (12032, 1536, 512),
    (24064, 384, 1152), (24064, 1152, 384), (6016, 512, 6144), (6016, 514, 768), (24064, 384, 130),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for R, K, N in shapes:
    A = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02
    out = torch.empty(R, N, device=dev); gA = torch.empty(R, K, device=dev); gW = torch.zeros(N, K, device=dev)
    fl = 2.0 * R * K * N
    t0 = timeit(lambda: ops.gemm(ops.mat(A), ops.mat(W), out))
    t1 = timeit(lambda: ops.gemm(ops.mat(out), ops.mat(W), gA, form=1))
    t2 = timeit(lambda: ops.wgrad(out, N, N, ops.mat(A), gW))
    print(f"R={R:6d} K={K:5d} N={N:5d}  fwd {fl/t0/1e12:6.1f} TF  dgrad {fl/t1/1e12:6.1f} TF  wgrad {fl/t2/1e12:6.1f} TF   ({t0*1e6:.0f} / {t1*1e6:.0f} / {t2*1e6:.0f} us)")

# accuracy of the active precision mode vs float64 (one mid-size problem, all three forms)
R, K, N = 1000, 768, 520
A = torch.randn(R, K); W = torch.randn(N, K) * 0.05; G = torch.randn(R, N)
out = torch.empty(R, N, device=dev); gA = torch.empty(R, K, device=dev); gW = torch.zeros(N, K, device=dev)
ops.gemm(ops.mat(A.to(dev)), ops.mat(W.to(dev)), out)
ops.gemm(ops.mat(G.to(dev)), ops.mat(W.to(dev)), gA, form=1)
ops.wgrad(G.to(dev), N, N, ops.mat(A.to(dev)), gW)
def rel(a, b): return float((a.cpu().double() - b).abs().max() / b.abs().max())
print("precision mode", ops.GEMM_PRECISION, "max rel err fwd/dgrad/wgrad:",
      rel(out, A.double() @ W.double().t()), rel(gA, G.double() @ W.double()), rel(gW, G.double().t() @ A.double()))
