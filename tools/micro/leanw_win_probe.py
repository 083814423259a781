"""MPD layer-4 weight gradient (1024 -> 1024 channels, 5 taps, stride 1) through the windowed K-major kernel by
sequence height: the step's table shows 110 (Hp = 31) ... 131 TFLOP/s (Hp = 153) for the same kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops
dev = "cuda"
def timeit(fn, n=6):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
Cout = 1024
HALO = 2
import itertools
for Cin, (Hp, S) in itertools.product((1024, 512), ((31, 1408), (47, 896), (64, 640), (103, 384), (153, 256))):
    Hout = Hp - 2 * HALO
    x = torch.zeros(S, Hout + 2 * HALO, Cin, device=dev); x[:, HALO:HALO + Hout] = torch.randn(S, Hout, Cin, device=dev)
    gy = torch.zeros(S, Hp, Cout, device=dev); gy[:, HALO:HALO + Hout] = torch.randn(S, Hout, Cout, device=dev)
    out = torch.zeros(Cout, 5 * Cin, device=dev)
    X = ops.win1d(x, S, Hout + 2 * HALO, Cin, Hp, 1, HALO, 5, unbounded=True)
    rows = S * Hp
    fl = 2.0 * Cout * 5 * Cin * rows
    tiles = 8 * (5 * Cin // 128)
    auto = ops.split_for(rows, tiles)
    res = []
    for s in (auto, 2, 3, 4, 6, 8, 9, 10, 16):
        t = timeit(lambda: ops.gemm(ops.mat(gy.reshape(rows, Cout)), X, out, form=2, atomic=True, split_k=s))
        res.append("s%d:%.0f" % (s, fl / t / 1e12))
    print("Cin=%4d Hp=%3d S=%4d rows=%6d auto %d path %d: " % (Cin, Hp, S, rows, auto, ops.L.lib.f2g_gemm_last_path()) + " ".join(res), flush=True)
