"""Checkpoint loading with the reference's file format and key schema
(reference flow2gan/checkpoint.py:111-168): {"model": state_dict, ...}, optional "module."
prefix from DDP, non-strict load so torchaudio's persistent buffers are accepted and ignored."""
from __future__ import annotations

import logging
from typing import Any, Dict

import torch
from torch import nn


def load_checkpoint(filename, model: nn.Module, strict: bool = False) -> Dict[str, Any]:
    logging.info(f"Loading checkpoint from {filename}")
    checkpoint = torch.load(filename, map_location="cpu", weights_only=False)
    src = checkpoint["model"]
    if next(iter(src)).startswith("module."):
        logging.info("Loading checkpoint saved by DDP")
        src = {k[len("module."):]: v for k, v in src.items()}
    model.load_state_dict(src, strict=strict)
    checkpoint.pop("model")
    return checkpoint


def save_checkpoint(filename, model: nn.Module, **extra) -> None:
    """reference checkpoint.py:84-108 layout (model state under "model")."""
    ckpt = {"model": model.state_dict()}
    ckpt.update(extra)
    torch.save(ckpt, filename)
