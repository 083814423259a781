"""flow2gan_amd: MI355X-native (gfx950) implementation of the Flow2GAN hot path.

Public surface mirrors the reference package (flow2gan/__init__.py:29-48):
    get_model(model_name, hf_model_name, checkpoint) -> (MelAudioGenerator, AttributeDict)
    flow2gan_amd.models.{config,generator,gan,discriminators,modules}
Importing this package loads libflow2gan_hip.so; there is no CPU or ATen fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

from . import _lib  # noqa: F401  (fails loudly when the HIP library is not built)
from .checkpoint import load_checkpoint
from .models.config import HF_MODEL_NAMES, HF_REPO, AttributeDict, get_generator_config
from .models.generator import MelAudioGenerator
from .models.modules import LogMelSpectrogram

__all__ = ["get_model", "MelAudioGenerator", "LogMelSpectrogram", "load_checkpoint"]


def _resolve_weights(hf_model_name: Optional[str], checkpoint: Optional[str]) -> str:
    """A local file wins; otherwise the named checkpoint of the reference's hub repository
    (network needed; same names and the same failure modes as the reference)."""
    if checkpoint is not None:
        print(f"Using local checkpoint: {checkpoint}")
        return checkpoint
    print("Using checkpoint from HF hub")
    if hf_model_name not in HF_MODEL_NAMES:
        raise AssertionError("Supported names are " + ", ".join(HF_MODEL_NAMES))
    from huggingface_hub import hf_hub_download  # imported late: not needed for local files
    return hf_hub_download(HF_REPO, filename=f"{hf_model_name}.pt")


def get_model(model_name: str = "mel_24k_base",
              hf_model_name: Optional[str] = "libritts-mel-4-step",
              checkpoint: Optional[str] = None) -> Tuple[MelAudioGenerator, AttributeDict]:
    """Same contract as the reference's `flow2gan.get_model`: build the named generator, load a
    local `.pt` or a hub checkpoint into it, return (model, its config)."""
    if checkpoint is None and hf_model_name is None:
        raise AssertionError("Either checkpoint or hf_model_name must be provided.")
    cfg = get_generator_config(model_name)
    generator = MelAudioGenerator(**cfg)
    load_checkpoint(_resolve_weights(hf_model_name, checkpoint), generator)
    return generator, cfg
