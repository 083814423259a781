"""Generate tests/golden/scaled_adam.npz by running the REAL reference optimizer
(/root/reference/flow2gan/optim.py: ScaledAdam + Eden2) in the build container.

TEST INFRASTRUCTURE.  Stores inputs (initial tensors, per-step gradients, hyper-parameters) and
the reference's outputs (parameters after selected steps, learning rates); checks
`scaled_adam_oracle.py` against them on the way.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from flow2gan import optim as roptim  # noqa: E402
from scaled_adam_oracle import ScaledAdamOracle, eden2_lr  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "scaled_adam.npz")
SHAPES = [(), (1,), (16,), (16,), (16,), (8, 16, 1), (8, 16, 1), (8, 1, 7), (24, 8), (3, 5, 2, 2), ()]
STEPS = 26
CHECK = (1, 4, 5, 10, 11, 17, 26)
CASES = {"p100": dict(clipping_scale=2.0, clipping_update_period=100),
         "p8": dict(clipping_scale=2.0, clipping_update_period=8, size_update_period=2),
         "noclip": dict(clipping_scale=None)}


def make_inputs():
    gen = torch.Generator().manual_seed(77)
    init = [torch.randn(s, generator=gen) * (0.5 if len(s) else 1.0) for s in SHAPES]
    init[8] = init[8] * 1e-6          # below param_min_rms: the scale step must not shrink it
    grads = []
    for k in range(STEPS):
        scale = 60.0 if k in (15, 22) else (0.02 if k == 12 else 1.0)   # clipped / tiny steps
        grads.append([torch.randn(s, generator=gen) * scale * (0.3 + 0.1 * i)
                      for i, s in enumerate(SHAPES)])
    return init, grads


def main():
    init, grads = make_inputs()
    out = {"n_tensors": np.int64(len(SHAPES)), "n_steps": np.int64(STEPS)}
    for i, t in enumerate(init):
        out[f"init/{i}"] = t.numpy()
    for k, gs in enumerate(grads):
        for i, g in enumerate(gs):
            out[f"grad/{k}/{i}"] = g.numpy()
    worst = 0.0
    for name, kw in CASES.items():
        params = [torch.nn.Parameter(t.clone()) for t in init]
        named = [(f"t{i}", p) for i, p in enumerate(params)]
        opt = roptim.ScaledAdam(named, lr=0.045, **kw)
        sched = roptim.Eden2(opt, lr_batches=10, warmup_batches=8, warmup_start=0.1)
        mine = [t.clone() for t in init]
        orc = ScaledAdamOracle(mine, lr=0.045, **kw)
        lrs = []
        for k in range(STEPS):
            for p, g in zip(params, grads[k]):
                p.grad = g.clone()
            opt.step()
            orc.g["lr"] = opt.param_groups[0]["lr"]
            orc.step([g.clone() for g in grads[k]])
            sched.step_batch()
            lrs.append(opt.param_groups[0]["lr"])
            assert abs(lrs[-1] - eden2_lr(0.045, k + 1, 10, 8, 0.1)) < 1e-12
            if (k + 1) in CHECK:
                for i, p in enumerate(params):
                    out[f"{name}/step{k + 1}/{i}"] = p.detach().numpy().copy()
                    d = float((p.detach() - mine[i]).abs().max()) / (float(p.detach().abs().max()) + 1e-12)
                    worst = max(worst, d)
        # the reference's own optimizer state dict after the last step (checkpoint-format pin):
        # state index -> tensors; flow2gan_amd.optim must emit the same keys, shapes and values
        sd = opt.state_dict()
        out[f"{name}/sd/indices"] = np.array(sorted(sd["state"].keys()), dtype=np.int64)
        for idx, st in sd["state"].items():
            for k, v in st.items():
                out[f"{name}/sd/{idx}/{k}"] = (v.detach().numpy().copy() if torch.is_tensor(v)
                                                else np.array(v, dtype=np.float64))
        out[f"{name}/sd/group_params"] = np.array(sd["param_groups"][0]["params"], dtype=np.int64)
        out[f"{name}/lrs"] = np.array(lrs)
        out[f"{name}/kw"] = np.array([2.0 if kw.get("clipping_scale") else 0.0,
                                      kw.get("clipping_update_period", 100),
                                      kw.get("size_update_period", 4)], dtype=np.float64)
    print(f"oracle vs reference: worst relative deviation {worst:.3e}")
    assert worst < 2e-6
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
