"""conv32 direct kernels: exact fp32 vs split-bf16 at the stage-2 shapes (S = 128 items)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for S, H, Win in [(128, 47, 410), (128, 94, 205), (128, 188, 103), (128, 47, 205)]:
    Wout = (Win - 1) // 2 + 1
    x = torch.randn(S * H * Win, 32, device=dev); gy = torch.randn(S * H * Wout, 32, device=dev)
    wp = torch.nn.Parameter(torch.randn(32, 27 * 32, device=dev) * 0.05)
    wT = torch.nn.Parameter(torch.randn(27, 32, 32, device=dev) * 0.05)
    b = torch.randn(32, device=dev)
    y = torch.empty(S * H * Wout, 32, device=dev); gx = torch.empty(S * H * Win, 32, device=dev)
    gwb = torch.zeros(32, 27 * 32, device=dev)
    fl = 2.0 * S * H * Wout * 32 * 27 * 32
    r = []
    for mode in ("fp32", "bf16x3"):
        ops.set_gemm_precision(mode)
        r.append(timeit(lambda: ops.conv32_s2_fwd(x, S, H, Win, Wout, wp, b, 0.1, y)))
        r.append(timeit(lambda: ops.conv32_s2_dgrad(gy, S, H, Win, Wout, wT, gx)))
        r.append(timeit(lambda: ops.conv32_s2_wgrad(x, gy, S, H, Win, Wout, gwb)))
    ops.set_gemm_precision("fp32")
    print(f"S={S} H={H} Win={Win}: fwd fp32 {fl/r[0]/1e12:6.1f} TF ({r[0]*1e6:5.0f} us) split {fl/r[3]/1e12:6.1f} TF ({r[3]*1e6:5.0f} us) | "
          f"dgrad fp32 {fl/r[1]/1e12:6.1f} TF ({r[1]*1e6:5.0f} us) split {fl/r[4]/1e12:6.1f} TF ({r[4]*1e6:5.0f} us) | "
          f"wgrad fp32 {fl/r[2]/1e12:6.1f} TF ({r[2]*1e6:5.0f} us) split {fl/r[5]/1e12:6.1f} TF ({r[5]*1e6:5.0f} us)", flush=True)
