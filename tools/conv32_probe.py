"""Direct 32->32 (3,9)/(1,2) conv kernel vs the implicit-GEMM path: correctness and speed."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops, _lib as L
from flow2gan_amd.ops import gemm, mat, win2d
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for S, H, Win in ((128, 188, 64), (128, 188, 51), (128, 94, 26), (6, 11, 13)):
    Wout = (Win - 1) // 2 + 1
    x = torch.randn(S * H * Win, 32, device=dev)
    w = torch.randn(32, 27 * 32, device=dev) * 0.05
    b = torch.randn(32, device=dev)
    y0 = torch.empty(S * H * Wout, 32, device=dev); y1 = torch.zeros_like(y0)
    X = win2d(x, S, H, Win, 32, Wout, 3, 9, 2, 1, 4)
    d = L.Conv32Desc()
    d.x, d.x_seq, d.x_line = x.data_ptr(), H * Win * 32, Win * 32
    d.S, d.H, d.Win, d.Wout = S, H, Win, Wout
    d.w, d.bias, d.lrelu_slope = w.data_ptr(), b.data_ptr(), 0.1
    d.y, d.y_seq, d.y_line = y1.data_ptr(), H * Wout * 32, Wout * 32
    f0 = lambda: gemm(X, mat(w), y0, bias=b, lrelu=0.1)
    f1 = lambda: ops.call("f2g_conv32_s2_fwd", C.byref(d))
    f0(); f1(); torch.cuda.synchronize()
    err = float((y0 - y1).abs().max() / y0.abs().max())
    t0, t1 = timeit(f0), timeit(f1)
    fl = 2.0 * S * H * Wout * 32 * 864
    print(f"S={S} H={H} Win={Win}: err {err:.2e}  gemm {t0*1e6:.0f} us {fl/t0/1e12:.1f} TF   direct {t1*1e6:.0f} us {fl/t1/1e12:.1f} TF")
