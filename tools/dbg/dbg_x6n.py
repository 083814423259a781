import sys; sys.path.insert(0, "/root/repo")
import torch
from flow2gan_amd import ops, _lib
ops.X6F_TALL_ROWS = 1000; ops.X6_MIN_ROWS = 1
S, Hp, Cout, nt = 37, 40, 128, 2
Lq = Hp - nt + 1
gmap = torch.randn(S * Hp, Cout, device="cuda")
wd = torch.nn.Parameter(torch.randn(32, nt * Cout, device="cuda") * 0.05)
out = torch.empty(S * (3 * Lq + 4), 32, device="cuda")
ops.set_gemm_precision("bf16x6")
A = ops.win1d(gmap, S, Hp, Cout, Lq, 1, 0, nt)
real = ops.call
def spy(name, *a):
    if name == "f2g_gemm":
        d = a[0]._obj
        print("launch: precision", d.precision, "A.split", d.A.split, "B.split", d.B.split, "x3", d.E.x3_out, "N", d.B.rows, "K", d.A.cols, "rows", d.A.rows)
    return real(name, *a)
ops.call = spy
ops.gemm(A, ops.mat(wd), out, rowmap=(Lq, (3 * Lq + 4) * 32, 96, 64))
print("path", _lib.lib.f2g_gemm_last_path())
ymask = torch.randn_like(out); cs = torch.zeros(32, device="cuda")
ops.gemm(A, ops.mat(wd), out, rowmap=(Lq, (3 * Lq + 4) * 32, 96, 64), mask=(ymask, 0, 0.1), colsum=cs)
print("path", _lib.lib.f2g_gemm_last_path())
