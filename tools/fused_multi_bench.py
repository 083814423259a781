"""The same layer of the three Fourier branches (mel_24k_base shapes, B = 64) as (a) three fused block
launches one after the other, (b) three launches on three streams, (c) ONE f2g_fused_block_multi launch.
MODE=multi|lanes|serial restricts the run to one variant (for PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow2gan_amd import ops
dev = "cuda"
ops.set_gemm_precision("bf16")
B = 64
ents = []
for C, F, up in ((768, 94, 1), (512, 188, 2), (384, 376, 4)):
    rows, H, Fc, NC = B * F, 3 * C, F // up, 8 * C
    x = torch.randn(rows, C, device=dev)
    w1 = torch.nn.Parameter(torch.randn(H, C, device=dev) * 0.03); w2 = torch.nn.Parameter(torch.randn(C, H, device=dev) * 0.03)
    ents.append(dict(x=x, B=B, F=F, Cc=C, K=7, lens=None, w_dw=torch.randn(C, 1, 7, device=dev) * 0.3,
                     b_dw=torch.zeros(C, device=dev), beta=torch.randn(C, device=dev) * 0.01,
                     log_scale=torch.ones(1, device=dev), wp=ops.mlp_pack(w1, w2),
                     b1=torch.randn(H, device=dev) * 0.1, alpha=torch.full((H,), 0.25, device=dev),
                     b2=torch.randn(C, device=dev) * 0.1, gamma=torch.ones(C, device=dev),
                     out=torch.empty(rows, C, device=dev), Hh=H, cproj=torch.randn(B * Fc, NC, device=dev),
                     ldcp=NC, Fc=Fc, up=up, cp_off=0, te=torch.randn(B, NC, device=dev) * 0.1, ldte=NC, te_off=0))
def one(e):
    ops.fused_block(e["x"], e["B"], e["F"], e["Cc"], 7, None, e["w_dw"], e["b_dw"], e["beta"], e["log_scale"],
                    e["wp"], e["b1"], e["alpha"], e["b2"], e["gamma"], e["out"], e["Hh"], e["cproj"], e["ldcp"],
                    e["Fc"], e["up"], 0, e["te"], e["ldte"], 0)
def serial():
    for e in ents: one(e)
streams = [torch.cuda.Stream() for _ in ents]
def lanes():
    ev = torch.cuda.Event(); ev.record()
    for s, e in zip(streams, ents):
        s.wait_event(ev)
        with torch.cuda.stream(s): one(e)
    for s in streams:
        d = torch.cuda.Event(); d.record(s); torch.cuda.current_stream().wait_event(d)
def multi():
    ops.fused_block_multi(ents)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
mode = os.environ.get("MODE", "")
flop = sum(4.0 * e["B"] * e["F"] * e["Cc"] * e["Hh"] for e in ents)
for name, fn in (("serial", serial), ("lanes", lanes), ("multi", multi)):
    if mode and mode != name: continue
    t = timeit(fn)
    print(f"{name:7s} {t:8.1f} us per layer of three branches = {flop / t / 1e6:7.1f} TFLOP/s ({flop / t / 1e6 / 2500:.3f} of the bf16 peak)", flush=True)
if not mode:
    for e in ents:
        t = timeit(lambda: one(e))
        print(f"  alone C={e['Cc']} rows={e['B'] * e['F']}: {t:7.1f} us")
    # the condition encoder's blocks: 512 channels at the condition frame rate (6016 rows), no cond / time inputs
    C, F = 512, 94
    rows, H = B * F, 3 * C
    x = torch.randn(rows, C, device=dev); out = torch.empty(rows, C, device=dev)
    w1 = torch.nn.Parameter(torch.randn(H, C, device=dev) * 0.03); w2 = torch.nn.Parameter(torch.randn(C, H, device=dev) * 0.03)
    wp = ops.mlp_pack(w1, w2)
    wd, z1, z3 = torch.randn(C, 1, 7, device=dev) * 0.3, torch.zeros(C, device=dev), torch.zeros(H, device=dev)
    al, ls, g1 = torch.full((H,), 0.25, device=dev), torch.ones(1, device=dev), torch.ones(C, device=dev)
    t = timeit(lambda: ops.fused_block(x, B, F, C, 7, None, wd, z1, z1, ls, wp, z3, al, z1, g1, out, H))
    print(f"  alone C=512 rows={rows} (condition encoder block): {t:7.1f} us")
