"""Launch one MRD band convolution (32 -> 32 channels, (3,9) taps, stride (1,2)) as the implicit
GEMM the hot path uses, repeatedly (for rocprofv3 --pmc runs).  argv: S Ft Win form(0|1|2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
from flow2gan_amd.ops import gemm, mat, win2d

S, Ft, Win = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 188, 64)))
form = int(sys.argv[4]) if len(sys.argv) > 4 else 0
C, kw, sw = 32, 9, 2
Wout = (Win + 2 * (kw // 2) - kw) // sw + 1
dev = "cuda"
x = torch.randn(S * Ft * Win, C, device=dev)
w = torch.randn(C, 3 * kw * C, device=dev) * 0.05
b = torch.zeros(C, device=dev)
y = torch.empty(S * Ft * Wout, C, device=dev)
gw = torch.zeros(C, 3 * kw * C, device=dev)
X = win2d(x, S, Ft, Win, C, Wout, 3, kw, sw, 1, kw // 2)
n = 10


def run():
    if form == 0:
        gemm(X, mat(w), y, bias=b, lrelu=0.1)
    elif form == 2:
        gemm(mat(y, S * Ft * Wout, C), X, gw, form=2, atomic=True,
             split_k=ops.split_for(X.rows, (3 * kw * C + 255) // 256))


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    run()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / n * 1e-3
fl = 2.0 * S * Ft * Wout * C * 3 * kw * C
print(f"form {form} rows {S*Ft*Wout} K {3*kw*C}: {t*1e6:.1f} us  {fl/t/1e12:.1f} TFLOP/s")
