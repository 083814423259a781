"""Direct 32->32 (3,9)/(1,2) conv forward at the band shapes of one GAN stage-2 step (B = 64: 128
sequences; frames 47 / 94 / 188; band widths of the three stride-2 layers): TFLOP/s of the kernel
the library picks (F2G_CONV32_V2=0: the round-2 kernel)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops, _lib as L
dev = "cuda"
# MODE=bf16x6: the fp32-class instances on the bf16 pipe (conv32x6.hip); MODE=bf16x3: the split-bf16 ones
ops.set_gemm_precision(os.environ.get("MODE", "fp32"))
ONLY = os.environ.get("ONLY", "")          # "fwd" / "dgrad" / "wgrad": that part alone
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
tot_t = tot_f = 0.0
for H, nb in ((47, 1025), (94, 513), (188, 257)) if ONLY in ("", "fwd") else ():
    edges = [int(f * nb) for f in (0.0, 0.1, 0.25, 0.5, 0.75, 1.0)]
    for b in range(5):
        Win = edges[b + 1] - edges[b]
        for layer in range(3):
            Wout = (Win - 1) // 2 + 1
            S = 128
            x = torch.randn(S * H * Win, 32, device=dev)
            w = torch.randn(32, 27 * 32, device=dev) * 0.05
            bb = torch.randn(32, device=dev)
            y = torch.empty(S * H * Wout, 32, device=dev)
            f = lambda: ops.conv32_s2_fwd(x, S, H, Win, Wout, w, bb, 0.1, y)
            t = timeit(f)
            fl = 2.0 * S * H * Wout * 32 * 864
            tot_t += t; tot_f += fl
            print(f"H={H:3d} Win={Win:3d} Wout={Wout:3d}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF", flush=True)
            Win = Wout
if tot_t:
    print(f"all 45 forward launches of a pass: {tot_t*1e3:.2f} ms, {tot_f/tot_t/1e12:.1f} TFLOP/s")

# ---- data gradient at the same shapes
tot_t = tot_f = 0.0
for H, nb in ((47, 1025), (94, 513), (188, 257)) if ONLY in ("", "dgrad") else ():
    edges = [int(f * nb) for f in (0.0, 0.1, 0.25, 0.5, 0.75, 1.0)]
    for b in range(5):
        Win = edges[b + 1] - edges[b]
        for layer in range(3):
            Wout = (Win - 1) // 2 + 1
            # MASK=1: the D-step's form (both halves, leaky-ReLU mask + bias sums fused); default: the G-step's
            # (generated half only, plain)
            MASK = os.environ.get("MASK", "0") == "1"
            S = 128 if MASK else 64
            gy = torch.randn(S * H * Wout, 32, device=dev)
            wT = torch.randn(27, 32, 32, device=dev) * 0.05
            gx = torch.empty(S * H * Win, 32, device=dev)
            if MASK:
                ym = torch.randn(S * H * Win, 32, device=dev)
                csum = torch.zeros(32, device=dev)
                t = timeit(lambda: ops.conv32_s2_dgrad(gy, S, H, Win, Wout, wT, gx, mask=(ym, 0, 0.1), colsum=csum))
            else:
                t = timeit(lambda: ops.conv32_s2_dgrad(gy, S, H, Win, Wout, wT, gx))
            fl = 2.0 * S * H * Wout * 32 * 864
            tot_t += t; tot_f += fl
            print(f"dgrad H={H:3d} Win={Win:3d}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF", flush=True)
            Win = Wout
if tot_t:
    print(f"all 45 data-gradient launches of a pass (S = {128 if os.environ.get('MASK', '0') == '1' else 64}): {tot_t*1e3:.2f} ms, {tot_f/tot_t/1e12:.1f} TFLOP/s")

# ---- weight gradient at the same shapes (D-step: both halves, S = 128)
tot_t = tot_f = 0.0
for H, nb in ((47, 1025), (94, 513), (188, 257)) if ONLY in ("", "wgrad") else ():
    edges = [int(f * nb) for f in (0.0, 0.1, 0.25, 0.5, 0.75, 1.0)]
    for b in range(5):
        Win = edges[b + 1] - edges[b]
        for layer in range(3):
            Wout = (Win - 1) // 2 + 1
            S = 128
            x = torch.randn(S * H * Win, 32, device=dev)
            gy = torch.randn(S * H * Wout, 32, device=dev)
            gw = torch.zeros(32, 27 * 32, device=dev)
            t = timeit(lambda: ops.conv32_s2_wgrad(x, gy, S, H, Win, Wout, gw))
            fl = 2.0 * S * H * Wout * 32 * 864
            tot_t += t; tot_f += fl
            print(f"wgrad H={H:3d} Win={Win:3d}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF", flush=True)
            Win = Wout
if tot_t:
    print(f"all 45 weight-gradient launches of a pass (S = 128): {tot_t*1e3:.2f} ms, {tot_f/tot_t/1e12:.1f} TFLOP/s")
