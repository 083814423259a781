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


def get_model(
    model_name: str = "mel_24k_base",
    hf_model_name: Optional[str] = "libritts-mel-4-step",
    checkpoint: Optional[str] = None,
) -> Tuple[MelAudioGenerator, AttributeDict]:
    assert (checkpoint is not None) or (hf_model_name is not None), \
        "Either checkpoint or hf_model_name must be provided."
    model_cfg = get_generator_config(model_name)
    model = MelAudioGenerator(**model_cfg)
    if checkpoint is not None:
        print(f"Using local checkpoint: {checkpoint}")
    else:
        print("Using checkpoint from HF hub")
        assert hf_model_name in HF_MODEL_NAMES, \
            "Supported names are " + ", ".join(HF_MODEL_NAMES.keys())
        from huggingface_hub import hf_hub_download
        checkpoint = hf_hub_download(HF_REPO, filename=hf_model_name + ".pt")
    load_checkpoint(checkpoint, model)
    return model, model_cfg
