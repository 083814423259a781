"""GAN wrapper: generator + [MPD, MRD] + multi-scale mel loss, with the reference's constructor,
attributes, `forward` signature, return tuples and state-dict keys (reference
flow2gan/models/gan.py:30-166).  The loss stack runs as fused HIP schedules
(flow2gan_amd/fused_disc.py)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

import os

from .. import fused_disc as FD
from .. import ops
from .._opts import opt

DISC_LANES = opt("disc_lanes", True)
from .discriminators import MultiPeriodDiscriminator, MultiResolutionDiscriminator
from .modules import MelSpectrogram


class GAN(nn.Module):
    def __init__(
        self,
        generator: nn.Module,
        mel_recon_n_ffts: Tuple[int, ...] = (32, 64, 128, 256, 512, 1024, 2048),
        mel_recon_n_mels: Tuple[int, ...] = (5, 10, 20, 40, 80, 160, 320),
    ):
        super().__init__()
        self.generator = generator
        mp_disc = MultiPeriodDiscriminator()
        mr_disc = MultiResolutionDiscriminator()
        self.discriminator = nn.ModuleList([mp_disc, mr_disc])
        self.mel_recon_modules = nn.ModuleList()
        for n_fft, n_mels in zip(mel_recon_n_ffts, mel_recon_n_mels):
            self.mel_recon_modules.append(
                MelSpectrogram(sample_rate=generator.sampling_rate, n_fft=n_fft,
                               hop_length=n_fft // 4, n_mels=n_mels, power=1))

    # ---- individual terms, same names as the reference (gan.py:57-99) -------------------
    def discriminator_loss(self, score_real: List[Tensor], score_fake: List[Tensor]) -> Tensor:
        """sum_d mean relu(1 - s_real) + mean relu(1 + s_fake)   (gan.py:57-66) on `f2g_hinge_loss`."""
        from ..leaf import discriminator_loss
        return discriminator_loss(score_real, score_fake)

    def generator_loss(self, score_fake: List[Tensor]) -> Tensor:
        """sum_d mean relu(1 - s_fake)   (gan.py:68-75)."""
        from ..leaf import generator_loss
        return generator_loss(score_fake)

    def feature_matching_loss(self, fmap_real: List[List[Tensor]], fmap_fake: List[List[Tensor]]) -> Tensor:
        """sum_d sum_layers mean |f_real.detach() - f_fake|   (gan.py:77-87) on `f2g_l1_loss`."""
        from ..leaf import feature_matching_loss
        return feature_matching_loss(fmap_real, fmap_fake)

    def mel_recon_loss(self, real: Tensor, fake: Tensor) -> Tensor:
        specs = tuple((m.n_fft, m.hop_length, m.mel_scale.fb) for m in self.mel_recon_modules)
        return FD.MelReconLossFn.apply(real, fake, specs)

    def _mp_terms(self, real: Tensor, fake: Tensor, train_disc: bool):
        mp = self.discriminator[0]
        return FD.MPDLossFn.apply(real, fake, train_disc, mp.periods, *FD.mpd_params(mp))

    def _mr_terms(self, real: Tensor, fake: Tensor, train_disc: bool):
        mr = self.discriminator[1]
        return FD.MRDLossFn.apply(real, fake, train_disc, mr.fft_sizes, *FD.mrd_params(mr))

    def forward(
        self,
        cond: Tensor,
        audio: Tensor,
        audio_lens: Optional[Tensor] = None,
        n_timesteps: int = 1,
        train_disc: bool = True,
        *,
        noise: Optional[Tensor] = None,
    ):
        if train_disc:
            # discriminator step (gan.py:109-132)
            self.discriminator.train()
            self.generator.eval()
            with torch.no_grad():
                pred_audio = self.generator.infer(cond=cond, audio_lens=audio_lens,
                                                  n_timesteps=n_timesteps, clamp_pred=False,
                                                  noise=noise)
            # MPD and MRD side by side (F2G_DISC_LANES=0 turns it off): each term forks one launch lane per
            # sub-discriminator inside; autograd runs each node's backward on its forward stream
            if DISC_LANES:
                lanes = ops.Lanes(audio.device, 2, "disc")
                with lanes.lane(0):
                    disc_loss_mp, _ = self._mp_terms(audio, pred_audio, True)
                with lanes.lane(1):
                    disc_loss_mr, _ = self._mr_terms(audio, pred_audio, True)
                lanes.join()
            else:
                disc_loss_mp, _ = self._mp_terms(audio, pred_audio, True)
                disc_loss_mr, _ = self._mr_terms(audio, pred_audio, True)
            return disc_loss_mp, disc_loss_mr
        # generator step (gan.py:133-166)
        self.discriminator.eval()
        self.generator.train()
        pred_audio = self.generator.infer(cond=cond, audio_lens=audio_lens,
                                          n_timesteps=n_timesteps, clamp_pred=False, noise=noise)
        if DISC_LANES:
            lanes = ops.Lanes(audio.device, 3, "disc")
            with lanes.lane(0):
                gen_loss_mp, feat_map_loss_mp = self._mp_terms(audio, pred_audio, False)
            with lanes.lane(1):
                gen_loss_mr, feat_map_loss_mr = self._mr_terms(audio, pred_audio, False)
            with lanes.lane(2):
                mel_recon_loss = self.mel_recon_loss(real=audio, fake=pred_audio)
            lanes.join()
        else:
            gen_loss_mp, feat_map_loss_mp = self._mp_terms(audio, pred_audio, False)
            gen_loss_mr, feat_map_loss_mr = self._mr_terms(audio, pred_audio, False)
            mel_recon_loss = self.mel_recon_loss(real=audio, fake=pred_audio)
        return gen_loss_mp, gen_loss_mr, feat_map_loss_mp, feat_map_loss_mr, mel_recon_loss
