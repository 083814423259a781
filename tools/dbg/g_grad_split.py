"""How far are the split-bf16 (F2G_GEMM=bf16x3) G-step gradients of the tiny golden cases from the
reference's, tensor by tensor?  Prints, per case and run, the tensors with the largest deviation
relative to (a) their own max, (b) the max over all tensors of the same kind (same last name
component), (c) the reference tensor's L2 norm.  Used to choose the bound of
tests/test_hip_gan.py::test_gan_steps_against_reference_vectors in that mode."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import flow2gan_amd as f2g  # noqa: E402
from flow2gan_amd import ops  # noqa: E402
import test_hip_gan as tg  # noqa: E402

random.random = lambda: 0.0
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_gemm_precision(mode)
for fixture, cfg in (("tiny_stage2", tg.TINY), ("tiny_stage2_44k", tg.TINY44)):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz")))
    for tag, n in (("n1", 1), ("n2", 2)):
        gan = tg.build_gan(f2g, g, cfg)
        mel, audio, noise = (tg.T(g[k]).cuda() for k in ("mel", "audio", "noise"))
        lens = tg.T(g[f"{tag}/lens"])
        for run in range(3):
            gan.zero_grad()
            ls = gan(mel, audio, lens, n, False, noise=noise)
            sum(w * l for w, l in zip((1.0, 0.1, 1.0, 0.1, 45.0), ls)).backward()
            rows, kind_max = [], {}
            for k, p in gan.generator.named_parameters():
                ref = tg.T(g[f"{tag}/G/g/{k}"]).double()
                kind = k.split(".")[-1] if not k.endswith("scale") else k.split(".")[-2] + ".scale"
                kind_max[kind] = max(kind_max.get(kind, 0.0), float(ref.abs().max()))
            tot_err2 = tot_ref2 = 0.0
            for k, p in gan.generator.named_parameters():
                ref = tg.T(g[f"{tag}/G/g/{k}"]).double()
                got = p.grad.detach().cpu().double()
                err = float((got - ref).abs().max())
                kind = k.split(".")[-1] if not k.endswith("scale") else k.split(".")[-2] + ".scale"
                l2 = float((got - ref).norm()) / (float(ref.norm()) + 1e-30)
                tot_err2 += float((got - ref).pow(2).sum())
                tot_ref2 += float(ref.pow(2).sum())
                rows.append((err / (float(ref.abs().max()) + 1e-12), err / (kind_max[kind] + 1e-12), l2, ref.numel(), k))
            rows.sort(reverse=True)
            print(f"## {fixture} {tag} run {run} mode {mode}: global rel L2 {np.sqrt(tot_err2 / tot_ref2):.3e}; "
                  f"worst own-max {rows[0][0]:.3f}; worst kind-max {max(r[1] for r in rows):.3f}; "
                  f"worst L2 (numel>=64) {max(r[2] for r in rows if r[3] >= 64):.3f}")
            for r in rows[:6]:
                print(f"   own {r[0]:.3f} kind {r[1]:.3f} l2 {r[2]:.3f} numel {r[3]:7d} {r[4]}")
