"""CPU restatement of the reference's ScaledAdam and Eden2 (flow2gan/optim.py).

TEST INFRASTRUCTURE ONLY: imported by tests/ and oracle/make_golden_optim.py, never by the
product path.  Parity pinned: `oracle/make_golden_optim.py` runs the REAL reference optimizer in
the build container and stores its inputs/outputs in tests/golden/scaled_adam.npz; this
restatement reproduces them (tests/test_oracle_golden.py).

The reference stacks same-shaped parameters and works on the stack (optim.py:45-122); the stack
dimension only batches independent per-tensor updates, so this restatement keeps one state per
tensor.  Everything is plain torch CPU fp32.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


class ScaledAdamOracle:
    def __init__(self, params: List[torch.Tensor], lr=3e-02, clipping_scale: Optional[float] = None,
                 betas=(0.9, 0.98), scalar_lr_scale=0.1, eps=1.0e-08, param_min_rms=1.0e-05,
                 param_max_rms=3.0, scalar_max=10.0, size_update_period=4,
                 clipping_update_period=100):
        self.params = list(params)
        self.g = dict(lr=lr, clipping_scale=clipping_scale, betas=betas,
                      scalar_lr_scale=scalar_lr_scale, eps=eps, param_min_rms=param_min_rms,
                      param_max_rms=param_max_rms, scalar_max=scalar_max,
                      size_update_period=size_update_period,
                      clipping_update_period=clipping_update_period)
        self.state: List[Dict] = [dict() for _ in self.params]
        self.step_count = 0
        self.model_norms = None
        self.model_norm_threshold = None
        self.last_clip = 1.0

    # optim.py:509-619
    def _clipping_scale(self, grads) -> float:
        g = self.g
        step = self.step_count
        if g["clipping_scale"] is None or step == 0:
            return 1.0
        period = g["clipping_update_period"]
        tot = torch.tensor(0.0)
        for p, grad, st in zip(self.params, grads, self.state):
            if p.numel() == 1:
                tot = tot + (grad ** 2).sum() * (g["scalar_lr_scale"] ** 2)
            else:
                tot = tot + ((grad * st["param_rms"]) ** 2).sum()
        tot_norm = tot.sqrt()
        if self.model_norms is None:
            self.model_norms = torch.zeros(period)
        self.model_norms[step % period] = tot_norm
        irregular = [i for i in (10, 20, 40) if i < period]
        if step % period == 0 or step in irregular:
            sorted_norms = self.model_norms.sort()[0]
            if step in irregular:
                sorted_norms = sorted_norms[-step:]
            n = sorted_norms.numel()
            median = sorted_norms[min(n - 1, (n // 4) * 2)].item()
            threshold = g["clipping_scale"] * median
            if step in irregular:
                threshold = threshold * 2.0
            self.model_norm_threshold = threshold
        if self.model_norm_threshold is None:
            return 1.0
        ans = min(1.0, (self.model_norm_threshold / (tot_norm + 1.0e-20)).item())
        if ans != ans:
            ans = 0.0
        return ans

    @torch.no_grad()
    def step(self, grads: List[torch.Tensor]):
        g = self.g
        clip = self._clipping_scale(grads)
        self.last_clip = clip
        step = self.step_count
        beta1, beta2 = g["betas"]
        P = g["size_update_period"]
        for p, grad, st in zip(self.params, grads, self.state):
            grad = torch.zeros_like(grad) if clip == 0.0 else (grad if clip == 1.0 else grad * clip)
            scalar = p.numel() == 1
            # basic_step (optim.py:125-151)
            lr = g["lr"] * (g["scalar_lr_scale"] if scalar else 1.0)
            if "exp_avg_sq" not in st:
                st["exp_avg_sq"] = torch.zeros_like(p)
            st["exp_avg_sq"].mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            v = st["exp_avg_sq"]
            bc2 = 1 - beta2 ** (step + 1)
            if bc2 < 0.99:
                v = v * (1.0 / bc2)
            delta = -lr * grad / (v.sqrt() + g["eps"])
            # scaling_step (optim.py:154-239)
            if not scalar:
                if "param_rms" not in st:
                    st["param_rms"] = (p ** 2).mean().sqrt()
                    st["scale_exp_avg_sq"] = torch.zeros(())
                    st["scale_grads"] = torch.zeros(P)
                st["scale_grads"][step % P] = (p * grad).sum()
                if step % P == P - 1:
                    st["param_rms"] = (p ** 2).mean().sqrt()
                rms = st["param_rms"]
                delta = delta * rms.clamp(min=g["param_min_rms"])
                if step % P == P - 1 and step > 0:
                    size_lr = g["lr"] * g["scalar_lr_scale"]
                    b2c = beta2 ** P
                    st["scale_exp_avg_sq"] = (st["scale_exp_avg_sq"] * b2c
                                              + (st["scale_grads"] ** 2).mean() * (1 - b2c))
                    size_step = (step + 1) // P
                    bc = 1 - b2c ** size_step
                    denom = st["scale_exp_avg_sq"].sqrt() + g["eps"]
                    ss = -size_lr * (bc ** 0.5) * st["scale_grads"].sum() / denom
                    if rms < g["param_min_rms"]:
                        ss = torch.zeros(())
                    ss = ss.clamp(min=-0.1, max=0.1)
                    ss = torch.minimum(ss, (g["param_max_rms"] - rms) / rms)
                    delta = delta + p * ss
            # momentum_step (optim.py:242-255)
            if "delta" not in st:
                st["delta"] = torch.zeros_like(p)
            st["delta"].mul_(beta1).add_(delta, alpha=1 - beta1)
            p.add_(st["delta"])
            if scalar:
                p.clamp_(min=-g["scalar_max"], max=g["scalar_max"])
        self.step_count += 1


def eden2_lr(base_lr: float, batch: int, lr_batches: float, warmup_batches: float = 500.0,
             warmup_start: float = 0.5) -> float:
    """optim.py:939-951."""
    factor = ((batch ** 2 + lr_batches ** 2) / lr_batches ** 2) ** -0.5
    warm = 1.0 if batch >= warmup_batches else warmup_start + (1.0 - warmup_start) * (batch / warmup_batches)
    return base_lr * factor * warm
