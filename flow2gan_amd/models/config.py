"""Named hyper-parameter sets (reference flow2gan/models/config.py:22-129): same names, same
values, same error behaviour (ValueError on an unknown name)."""
from __future__ import annotations

import json


class AttributeDict(dict):
    """dict with attribute access (reference flow2gan/utils.py:247-269)."""

    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError(f"No such attribute '{key}'")

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        if key in self:
            del self[key]
            return
        raise AttributeError(f"No such attribute '{key}'")

    def __str__(self, indent: int = 2):
        return json.dumps({k: (v if isinstance(v, (int, float, str, bool, list, tuple, type(None)))
                               else str(v)) for k, v in self.items()}, indent=indent, sort_keys=True)


def _generator(sampling_rate, n_mels, mel_n_fft, mel_hop, n_ffts, hops, loss_n_fft, loss_hop):
    return {
        "sampling_rate": sampling_rate, "n_mels": n_mels, "mel_n_fft": mel_n_fft,
        "mel_hop_length": mel_hop, "n_ffts": n_ffts, "hop_lengths": hops,
        "channels": (768, 512, 384), "time_embed_channels": 512, "hidden_factor": 3,
        "conv_kernel_sizes": (7, 7, 7), "num_layers": (8, 8, 8), "use_cond_encoder": True,
        "cond_enc_channels": 512, "cond_enc_hidden_factor": 3, "cond_enc_conv_kernel_size": 7,
        "cond_enc_num_layers": 4, "residual_scale": 1.0, "init_noise_scale": 0.1, "pred_x1": True,
        "branch_reduction": "mean", "spec_scaling_loss": True, "loss_n_filters": 256,
        "loss_n_fft": loss_n_fft, "loss_hop_length": loss_hop, "loss_power": 0.5,
        "loss_eps": 1e-7, "loss_scale_min": 1e-2, "loss_scale_max": 1e+2, "branch_dropout": 0.05,
        "max_add_noise_scale": 0.0,
    }


mel_24k_base = _generator(24000, 100, 1024, 256, (512, 256, 128), (256, 128, 64), 1024, 256)
mel_44k_128band_512x_base = _generator(44100, 128, 2048, 512, (1024, 512, 256), (512, 256, 128),
                                       2048, 512)


def get_generator_config(model_named: str = "mel_24k_base") -> AttributeDict:
    if model_named == "mel_24k_base":
        return AttributeDict(mel_24k_base)
    elif model_named == "mel_44k_128band_512x_base":
        return AttributeDict(mel_44k_128band_512x_base)
    raise ValueError(f"Unsupported model name: {model_named}")


gan_multi_scale_mel_recon = {
    "mel_recon_n_ffts": (32, 64, 128, 256, 512, 1024, 2048),
    "mel_recon_n_mels": (5, 10, 20, 40, 80, 160, 320),
}
gan_single_scale_mel_recon = {"mel_recon_n_ffts": (1024,), "mel_recon_n_mels": (100,)}


def get_gan_config(model_name: str) -> AttributeDict:
    if model_name == "gan_multi_scale_mel_recon":
        return AttributeDict(gan_multi_scale_mel_recon)
    elif model_name == "gan_single_scale_mel_recon":
        return AttributeDict(gan_single_scale_mel_recon)
    raise ValueError(f"Unsupported model name: {model_name}")


HF_REPO = "k2-fsa/Flow2GAN"
HF_MODEL_NAMES = {
    "libritts-mel-1-step": 1, "libritts-mel-2-step": 2, "libritts-mel-4-step": 4,
    "universal-24k-mel-1-step": 1, "universal-24k-mel-2-step": 2, "universal-24k-mel-4-step": 4,
    "universal-44k-mel-128band-512x-1-step": 1, "universal-44k-mel-128band-512x-2-step": 2,
    "universal-44k-mel-128band-512x-4-step": 4,
}
