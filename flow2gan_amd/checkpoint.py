"""Checkpoints and model averaging with the reference's file format and key schema (SURVEY §8f-2;
reference flow2gan/checkpoint.py:40-168 save/load, :171-213 average_checkpoints, :378-409
update_averaged_model, :442-501 average_checkpoints_with_averaged_model, :504-531
average_state_dict).  Host-side bookkeeping on state dicts: plain tensor arithmetic, not part of the
kernel path.

File layout (identical to the reference, so either side can read the other's files):
  {"model": state_dict, "optimizer": ..., "scheduler": ..., "grad_scaler": None, "sampler": None,
   ["model_avg": fp32 state_dict], ["model_ema": ...], ["optimizer_disc", "scheduler_disc"],
   **params (epoch, batch_idx_train, ...)}
A "module." prefix left by DDP is stripped on load; loading is non-strict by default because real
torchaudio registers persistent window / filterbank buffers this package does not keep.
The optimizer entry is in the reference's layout too (stacked per-shape state, see
flow2gan_amd.optim.ScaledAdam.state_dict).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import torch
from torch import Tensor, nn


def _unwrap(model: nn.Module) -> nn.Module:
    return model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model


def save_checkpoint(filename: Union[str, Path], model: nn.Module,
                    model_avg: Optional[nn.Module] = None, model_ema: Optional[nn.Module] = None,
                    params: Optional[Dict[str, Any]] = None, optimizer=None, scheduler=None,
                    scaler=None, sampler=None, optimizer_disc=None, scheduler_disc=None,
                    rank: int = 0, **extra) -> None:
    """checkpoint.py:40-108.  Only rank 0 writes."""
    if rank != 0:
        return
    logging.info(f"Saving checkpoint to {filename}")

    def sd(obj):
        return obj.state_dict() if obj is not None else None

    ckpt: Dict[str, Any] = {"model": _unwrap(model).state_dict(), "optimizer": sd(optimizer),
                            "scheduler": sd(scheduler), "grad_scaler": sd(scaler),
                            "sampler": sd(sampler)}
    if model_avg is not None:
        ckpt["model_avg"] = {k: (v.to(torch.float32) if v.is_floating_point() else v)
                             for k, v in model_avg.state_dict().items()}
    if model_ema is not None:
        ckpt["model_ema"] = {k: (v.to(torch.float32) if v.is_floating_point() else v)
                             for k, v in model_ema.state_dict().items()}
    if optimizer_disc is not None:
        ckpt["optimizer_disc"] = optimizer_disc.state_dict()
    if scheduler_disc is not None:
        ckpt["scheduler_disc"] = scheduler_disc.state_dict()
    for k, v in list((params or {}).items()) + list(extra.items()):
        assert k not in ckpt, k
        ckpt[k] = v
    torch.save(ckpt, filename)


def load_checkpoint(filename: Union[str, Path], model: nn.Module,
                    model_avg: Optional[nn.Module] = None, model_ema: Optional[nn.Module] = None,
                    optimizer=None, scheduler=None, scaler=None, sampler=None, optimizer_disc=None,
                    scheduler_disc=None, strict: bool = False) -> Dict[str, Any]:
    """checkpoint.py:111-168.  Returns what is left of the file's dict (epoch, batch_idx_train, ...)."""
    logging.info(f"Loading checkpoint from {filename}")
    checkpoint = torch.load(filename, map_location="cpu", weights_only=False)
    src = checkpoint.pop("model")
    if next(iter(src)).startswith("module."):
        logging.info("Loading checkpoint saved by DDP")
        src = {k[len("module."):]: v for k, v in src.items()}
    model.load_state_dict(src, strict=strict)
    for name, target in (("model_avg", model_avg), ("model_ema", model_ema)):
        if target is not None and name in checkpoint:
            target.load_state_dict(checkpoint.pop(name), strict=strict)
    for name, obj in (("optimizer", optimizer), ("scheduler", scheduler), ("grad_scaler", scaler),
                      ("sampler", sampler), ("optimizer_disc", optimizer_disc),
                      ("scheduler_disc", scheduler_disc)):
        state = checkpoint.get(name)
        if obj is not None and state:
            obj.load_state_dict(state)
            checkpoint.pop(name)
    return checkpoint


def _unique_keys(state_dict: Dict[str, Tensor]) -> List[str]:
    """Tied parameters (same storage under two keys) must be touched once (checkpoint.py:190-199)."""
    seen, keys = set(), []
    for k, v in state_dict.items():
        if v.data_ptr() in seen and v.numel() > 0:
            continue
        seen.add(v.data_ptr())
        keys.append(k)
    return keys


def average_state_dict(state_dict_1: Dict[str, Tensor], state_dict_2: Dict[str, Tensor],
                       weight_1: float, weight_2: float, scaling_factor: float = 1.0) -> None:
    """In place: sd1 = (sd1 * w1 + sd2 * w2) * scaling_factor on floating-point entries
    (checkpoint.py:504-531)."""
    for k in _unique_keys(state_dict_1):
        v = state_dict_1[k]
        if v.is_floating_point():
            v.mul_(weight_1)
            v.add_(state_dict_2[k].to(device=v.device) * weight_2)  # product in sd2's own dtype
            v.mul_(scaling_factor)


def average_checkpoints(filenames: List[Union[str, Path]],
                        device: torch.device = torch.device("cpu")) -> Dict[str, Tensor]:
    """Plain mean of the "model" entries of several checkpoint files (checkpoint.py:171-213)."""
    n = len(filenames)
    avg = torch.load(filenames[0], map_location=device, weights_only=False)["model"]
    keys = _unique_keys(avg)
    for f in filenames[1:]:
        other = torch.load(f, map_location=device, weights_only=False)["model"]
        for k in keys:
            avg[k] += other[k]
    for k in keys:
        if avg[k].is_floating_point():
            avg[k] /= n
        else:
            avg[k] //= n
    return avg


def update_averaged_model(params: Dict[str, Any], model_cur: nn.Module, model_avg: nn.Module) -> None:
    """Running average kept in `model_avg` (the reference holds it in fp64, finetune.py):
    avg = cur * (average_period / batch_idx_train) + avg * (1 - that)   (checkpoint.py:378-409)."""
    w_cur = params["average_period"] / params["batch_idx_train"]
    average_state_dict(model_avg.state_dict(), _unwrap(model_cur).state_dict(), 1.0 - w_cur, w_cur)


def average_checkpoints_with_averaged_model(filename_start: Union[str, Path],
                                            filename_end: Union[str, Path],
                                            device: torch.device = torch.device("cpu")
                                            ) -> Dict[str, Tensor]:
    """Mean of the model over (start, end] from the two files' running averages:
    avg = (avg_end * end - avg_start * start) / (end - start), evaluated as
    (avg_end + avg_start * (w_start / w_end)) * w_end to keep the factors small
    (checkpoint.py:442-501)."""
    start = torch.load(filename_start, map_location=device, weights_only=False)
    end = torch.load(filename_end, map_location=device, weights_only=False)
    interval = end["batch_idx_train"] - start["batch_idx_train"]
    assert interval > 0, interval
    w_end = end["batch_idx_train"] / interval
    w_start = 1.0 - w_end
    avg = end["model_avg"]
    average_state_dict(avg, start["model_avg"], 1.0, w_start / w_end, scaling_factor=w_end)
    return avg
