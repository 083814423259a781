"""ScaledAdam + Eden2 with the reference's constructor arguments and semantics
(reference flow2gan/optim.py:258-507 ScaledAdam, :743-840 LRScheduler, :904-951 Eden2), running as
three multi-tensor HIP launches per step (csrc/optim.hip) over device-resident tables.

Differences to the reference that a caller can observe:
  * no stacking of same-shaped parameters (optim.py:104-122): state is kept per tensor in two flat
    arenas (`exp_avg_sq`, `delta`) plus a few floats per tensor; results are identical because the
    stacked dimension is only a batching device there;
  * the gradient-clipping factor never leaves the device (the reference calls `.item()` every
    step, optim.py:599); the quartile log lines are not printed;
  * state is KEPT per tensor; `state_dict()` / `load_state_dict()` convert to and from the reference's
    stacked per-shape layout, so checkpoints are interchangeable.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch.optim import Optimizer

from . import _lib as L
from ._lib import call


class ScaledAdam(Optimizer):
    def __init__(self, params, lr=3e-02, clipping_scale=None, betas=(0.9, 0.98),
                 scalar_lr_scale=0.1, eps=1.0e-08, param_min_rms=1.0e-05, param_max_rms=3.0,
                 scalar_max=10.0, size_update_period=4, clipping_update_period=100):
        defaults = dict(lr=lr, clipping_scale=clipping_scale, betas=betas,
                        scalar_lr_scale=scalar_lr_scale, eps=eps, param_min_rms=param_min_rms,
                        param_max_rms=param_max_rms, scalar_max=scalar_max,
                        size_update_period=size_update_period,
                        clipping_update_period=clipping_update_period)
        param_groups, names = self._get_names_of_parameters(params)
        super().__init__(param_groups, defaults)
        assert len(self.param_groups) == len(names)
        self.parameters_names = names
        self._plan = None
        self._steps: List[int] = [0] * len(self.param_groups)

    # optim.py:341-446: params, groups of params, named params or groups of named params
    def _get_names_of_parameters(self, params_or_named_params) -> Tuple[List[Dict], List[List[str]]]:
        items = list(params_or_named_params)
        if len(items) == 0:
            raise ValueError("optimizer got an empty parameter list")
        groups, group_names = [], []
        if not isinstance(items[0], dict):
            ps, ns = [], []
            for it in items:
                if isinstance(it, tuple):
                    name, p = it
                else:
                    assert isinstance(it, torch.Tensor)
                    name, p = "foo", it
                ps.append(p)
                ns.append(name)
            groups.append({"params": ps})
            group_names.append(ns)
        else:
            for g in items:
                g = dict(g)
                if "named_params" in g:
                    named = list(g.pop("named_params"))
                    g["params"] = [x[1] for x in named]
                    ns = [x[0] for x in named]
                else:
                    assert "params" in g
                    g["params"] = list(g["params"])
                    ns = ["foo" for _ in g["params"]]
                groups.append(g)
                group_names.append(ns)
        return groups, group_names

    # ---------------------------------------------------------------- device tables
    def _build(self):
        tensors = []
        for gi, group in enumerate(self.param_groups):
            first = len(tensors)
            for p in group["params"]:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise L.F2GError("ScaledAdam needs fp32 parameters on an MI355X (no CPU path)")
                if not p.is_contiguous():
                    raise L.F2GError("ScaledAdam needs contiguous parameters")
                tensors.append((p, gi))
            group["_first"], group["_count"] = first, len(tensors) - first
        dev = tensors[0][0].device
        T = len(tensors)
        offs, total = [], 0
        for p, _ in tensors:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # 16-byte aligned slots
        plan = {"tensors": tensors, "dev": dev, "T": T, "offs": offs}
        plan["v"] = torch.zeros(total, device=dev)   # exp_avg_sq (optim.py:134-141)
        plan["m"] = torch.zeros(total, device=dev)   # delta / momentum (optim.py:245-251)
        plan["stats"] = torch.empty(3 * T, device=dev)
        plan["tstate"] = torch.zeros(T * L.SADAM_TSTATE, device=dev)
        plan["gstate"] = torch.zeros(len(self.param_groups) * L.SADAM_GSTATE, device=dev)
        plan["coef"] = torch.zeros(T * L.SADAM_NCOEF, device=dev)
        ce = int(L.lib.f2g_sadam_chunk_elems())
        chunks = []
        for ti, (p, _) in enumerate(tensors):
            n = p.numel()
            for o in range(0, n, ce):
                chunks.append((ti, min(ce, n - o), o))
        arr = np.array(chunks, dtype=L.SADAM_CHUNK_DTYPE)
        plan["chunks"] = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        plan["nchunks"] = len(chunks)
        plan["table_host"] = np.zeros(T, dtype=L.SADAM_TENSOR_DTYPE)
        plan["table"] = torch.zeros(T * plan["table_host"].itemsize, dtype=torch.uint8, device=dev)
        plan["gptrs"] = None
        self._plan = plan

    def _refresh_table(self):
        """(Re-)upload the tensor table when a parameter's storage or gradient moved (gradients
        that are views of the reducer's arenas keep their address, so this is usually a no-op)."""
        plan = self._plan
        gptrs = tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr())
                      for p, _ in plan["tensors"])
        if gptrs == plan["gptrs"]:
            return
        tab = plan["table_host"]
        vbase, mbase = plan["v"].data_ptr(), plan["m"].data_ptr()
        for i, (p, gi) in enumerate(plan["tensors"]):
            if p.grad is not None and (not p.grad.is_contiguous() or p.grad.dtype != torch.float32):
                raise L.F2GError("ScaledAdam needs contiguous fp32 gradients")
            tab[i] = (gptrs[i][0], gptrs[i][1], vbase + 4 * plan["offs"][i],
                      mbase + 4 * plan["offs"][i], p.numel(), gi, 1 if p.numel() == 1 else 0)
        plan["table"].copy_(torch.from_numpy(tab.view(np.uint8)), non_blocking=False)
        plan["gptrs"] = gptrs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._plan is None:
            self._build()
        self._refresh_table()
        plan = self._plan
        tab, chunks = plan["table"].data_ptr(), plan["chunks"].data_ptr()
        call("f2g_sadam_stats", tab, chunks, plan["nchunks"], plan["stats"].data_ptr(), plan["T"])
        for gi, group in enumerate(self.param_groups):
            g = L.SadamGroup()
            g.lr = group["lr"]
            g.beta1, g.beta2 = group["betas"]
            g.scalar_lr_scale = group["scalar_lr_scale"]
            g.eps = group["eps"]
            g.param_min_rms, g.param_max_rms = group["param_min_rms"], group["param_max_rms"]
            g.scalar_max = group["scalar_max"]
            g.clipping_scale = 0.0 if group["clipping_scale"] is None else group["clipping_scale"]
            g.size_update_period = group["size_update_period"]
            g.clipping_update_period = group["clipping_update_period"]
            g.step = self._steps[gi]
            g.first, g.count = group["_first"], group["_count"]
            call("f2g_sadam_prepare", tab, C.byref(g), plan["stats"].data_ptr(),
                 plan["tstate"].data_ptr(),
                 plan["gstate"].data_ptr() + 4 * gi * L.SADAM_GSTATE, plan["coef"].data_ptr())
            self._steps[gi] += 1
        call("f2g_sadam_update", tab, chunks, plan["nchunks"], plan["coef"].data_ptr())
        # the parameters were written through raw pointers (no autograd version bump): tell the
        # derived-weight cache (transposes, window-major conv weights) that THESE tensors changed
        # (the other network's cached copies stay valid)
        from . import ops
        written = [p for p, _ in plan["tensors"]]
        ops.bump_weight_epoch(written)
        ops.rebuild_derived(written)     # (the copies in use since the last step, as a few f2g_multi launches)
        return loss

    # ---------------------------------------------------------------- introspection / checkpoints
    def tensor_state(self, p: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Views of one parameter's optimizer state (names as in the reference's state dict)."""
        plan = self._plan
        i = next(k for k, (q, _) in enumerate(plan["tensors"]) if q is p)
        o, n = plan["offs"][i], p.numel()
        ts = plan["tstate"][i * L.SADAM_TSTATE:(i + 1) * L.SADAM_TSTATE]
        return {"exp_avg_sq": plan["v"][o:o + n].view_as(p), "delta": plan["m"][o:o + n].view_as(p),
                "param_rms": ts[0], "scale_exp_avg_sq": ts[1], "scale_grads": ts[2:]}

    # The checkpoint entry has the REFERENCE's layout (torch.optim.Optimizer.state_dict() of the
    # reference ScaledAdam): same-shaped parameters of a group are stacked along a new first
    # dimension (optim.py:70-122), the stack's state sits under the index of its FIRST parameter,
    # batches are ordered by (dtype, *shape), and the clipping statistics live in the first batch's
    # state (optim.py:509-619).  A reference `epoch-N.pt` optimizer entry loads here and vice versa.
    def _batches(self, gi: int):
        """[(key, [tensor index in the plan / position in the group, ...])] in the reference's order."""
        group = self.param_groups[gi]
        by_key: Dict[tuple, list] = {}
        for pos, p in enumerate(group["params"]):
            by_key.setdefault((str(p.dtype), *p.shape), []).append(pos)
        return [(k, by_key[k]) for k in sorted(by_key)]

    def state_dict(self):
        groups, state, base = [], {}, 0
        for gi, group in enumerate(self.param_groups):
            n = len(group["params"])
            packed = {k: v for k, v in group.items() if k != "params" and not k.startswith("_")}
            packed["params"] = list(range(base, base + n))
            groups.append(packed)
            if self._plan is not None and self._steps[gi] > 0:
                plan = self._plan
                P = group["size_update_period"]
                period = group["clipping_update_period"]
                first = group["_first"]
                gs = plan["gstate"][gi * L.SADAM_GSTATE:(gi + 1) * L.SADAM_GSTATE]
                for bi, (key, members) in enumerate(self._batches(gi)):
                    p0 = group["params"][members[0]]
                    vs, ms, ts = [], [], []
                    for pos in members:
                        ti = first + pos
                        o, cnt = plan["offs"][ti], p0.numel()
                        vs.append(plan["v"][o:o + cnt].view_as(p0))
                        ms.append(plan["m"][o:o + cnt].view_as(p0))
                        ts.append(plan["tstate"][ti * L.SADAM_TSTATE:(ti + 1) * L.SADAM_TSTATE])
                    st = {"step": self._steps[gi], "exp_avg_sq": torch.stack(vs), "delta": torch.stack(ms)}
                    if p0.numel() > 1:
                        tt = torch.stack(ts)                                  # (nb, TSTATE)
                        one = (len(members),) + (1,) * p0.dim()
                        st["param_rms"] = tt[:, 0].reshape(one).clone()
                        st["scale_exp_avg_sq"] = tt[:, 1].reshape(one).clone()
                        st["scale_grads"] = tt[:, 2:2 + P].t().reshape((P,) + one).clone()
                    if bi == 0 and group["clipping_scale"] is not None and self._steps[gi] > 1:
                        st["model_norms"] = gs[:period].clone()
                        if float(gs[1025]) != 0.0:
                            st["model_norm_threshold"] = float(gs[1024])
                            st["num_clipped"] = 0
                    state[base + members[0]] = st
            base += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:            # this class's round-1 layout (flat arenas)
            self._steps = list(sd["steps"])
            for g, saved in zip(self.param_groups, sd["param_groups"]):
                g.update({k: v for k, v in saved.items() if not k.startswith("_")})
            if "v" in sd:
                if self._plan is None:
                    self._build()
                for k in ("v", "m", "tstate", "gstate"):
                    self._plan[k].copy_(sd[k])
            return
        assert len(sd["param_groups"]) == len(self.param_groups)
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params" and not k.startswith("_")})
        if not sd["state"]:
            self._steps = [0] * len(self.param_groups)
            return
        if self._plan is None:
            self._build()
        plan, base = self._plan, 0
        dev = plan["dev"]
        for gi, group in enumerate(self.param_groups):
            n = len(group["params"])
            P = group["size_update_period"]
            period = group["clipping_update_period"]
            first = group["_first"]
            gs = plan["gstate"][gi * L.SADAM_GSTATE:(gi + 1) * L.SADAM_GSTATE]
            gs.zero_()
            for bi, (key, members) in enumerate(self._batches(gi)):
                st = sd["state"].get(base + members[0])
                if st is None:
                    continue
                self._steps[gi] = int(st["step"])
                p0 = group["params"][members[0]]
                for row, pos in enumerate(members):
                    ti = first + pos
                    o, cnt = plan["offs"][ti], p0.numel()
                    plan["v"][o:o + cnt].copy_(st["exp_avg_sq"][row].reshape(-1).to(dev))
                    plan["m"][o:o + cnt].copy_(st["delta"][row].reshape(-1).to(dev))
                    if p0.numel() > 1:
                        ts = plan["tstate"][ti * L.SADAM_TSTATE:(ti + 1) * L.SADAM_TSTATE]
                        ts[0] = float(st["param_rms"][row].reshape(-1)[0])
                        ts[1] = float(st["scale_exp_avg_sq"][row].reshape(-1)[0])
                        ts[2:2 + P].copy_(st["scale_grads"][:, row].reshape(-1).to(dev))
                if bi == 0 and "model_norms" in st:
                    gs[:period].copy_(st["model_norms"].reshape(-1)[:period].to(dev))
                    if "model_norm_threshold" in st:
                        gs[1024] = float(st["model_norm_threshold"])
                        gs[1025] = 1.0
            base += n


class LRScheduler:
    """optim.py:743-840: batch/epoch counters, `base_lrs` from `initial_lr`, lrs set on the groups."""

    def __init__(self, optimizer: Optimizer, verbose: bool = False):
        if not isinstance(optimizer, Optimizer):
            raise TypeError(f"{type(optimizer).__name__} is not an Optimizer")
        self.optimizer = optimizer
        self.verbose = verbose
        for group in optimizer.param_groups:
            group.setdefault("base_lr", group["lr"])
        self.base_lrs = [group["base_lr"] for group in optimizer.param_groups]
        self.epoch = 0
        self.batch = 0

    def state_dict(self):
        return {"epoch": self.epoch, "batch": self.batch}  # base lrs stay the constructor's

    def load_state_dict(self, state_dict):
        base_lrs = self.base_lrs
        self.__dict__.update(state_dict)
        self.base_lrs = base_lrs  # optim.py:786-789: keep the constructor's base lrs

    def get_last_lr(self) -> List[float]:
        return self._last_lr

    def get_lr(self):
        raise NotImplementedError

    def step_batch(self, batch: Optional[int] = None) -> None:
        self.batch = batch if batch is not None else self.batch + 1
        self._set_lrs()

    def step_epoch(self, epoch: Optional[int] = None):
        self.epoch = epoch if epoch is not None else self.epoch + 1
        self._set_lrs()

    def _set_lrs(self):
        values = self.get_lr()
        assert len(values) == len(self.optimizer.param_groups)
        for group, lr in zip(self.optimizer.param_groups, values):
            group["lr"] = lr
        self._last_lr = [group["lr"] for group in self.optimizer.param_groups]


class Eden2(LRScheduler):
    """optim.py:904-951: lr = base_lr * ((batch^2 + lr_batches^2) / lr_batches^2)^-0.5 * warmup,
    warmup rising linearly from `warmup_start` to 1 over `warmup_batches`."""

    def __init__(self, optimizer: Optimizer, lr_batches: Union[int, float],
                 warmup_batches: Union[int, float] = 500.0, warmup_start: float = 0.5,
                 verbose: bool = False):
        super().__init__(optimizer, verbose)
        self.lr_batches = lr_batches
        self.warmup_batches = warmup_batches
        assert 0.0 <= warmup_start <= 1.0, warmup_start
        self.warmup_start = warmup_start

    def get_lr(self):
        factor = ((self.batch ** 2 + self.lr_batches ** 2) / self.lr_batches ** 2) ** -0.5
        warmup = (1.0 if self.batch >= self.warmup_batches
                  else self.warmup_start + (1.0 - self.warmup_start) * (self.batch / self.warmup_batches))
        return [x * factor * warmup for x in self.base_lrs]
