"""Training-step harness with the reference trainers' semantics.

`compute_loss_stage1` mirrors flow2gan/bin/pretrain.py:341-359 and `compute_loss_stage2`
flow2gan/bin/finetune.py:427-492: the mel condition is computed from the audio batch inside the
step, the loss weights and D/G switch are the reference's, `GanStepper` reproduces the batch
schedule of finetune.py:569-631 (discriminator-only until `gen_start_batch_idx`, then strict
D / G alternation, each on a NEW batch), and gradients are averaged across ranks exactly for the
sub-model being stepped.  `optimizer_{d,g}` are callables run after the gradient exchange, e.g.
`lambda: (opt.step(), sched.step_batch())` with flow2gan_amd.optim.ScaledAdam / Eden2.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .dist import GradReducer


@dataclass
class GanLossScales:
    """finetune.py:297-343 defaults."""
    disc_loss_mp_scale: float = 1.0
    disc_loss_mr_scale: float = 0.1
    gen_loss_mp_scale: float = 1.0
    gen_loss_mr_scale: float = 0.1
    feat_map_loss_mp_scale: float = 1.0
    feat_map_loss_mr_scale: float = 0.1
    mel_recon_loss_scale: float = 45.0


def compute_loss_stage1(audio: Tensor, audio_lens: Tensor, cond_module: nn.Module,
                        model: nn.Module, is_training: bool = True) -> Tuple[Tensor, Dict]:
    """pretrain.py:341-359: cond = logmel(audio); loss = model(cond, audio, audio_lens)."""
    with torch.set_grad_enabled(is_training):
        cond = cond_module(audio)
        loss = model(cond=cond, audio=audio, audio_lens=audio_lens)
    assert loss.requires_grad == is_training
    return loss, {"samples": audio.shape[0], "loss": loss.detach()}


def compute_loss_stage2(audio: Tensor, audio_lens: Tensor, model: nn.Module,
                        cond_module: nn.Module, n_timesteps: int = 1,
                        scales: GanLossScales = GanLossScales(), is_training: bool = True,
                        train_disc: bool = True) -> Tuple[Tensor, Dict]:
    """finetune.py:427-492.  Unlike the reference, the per-term `.item()` host syncs are left to
    the caller (`info` holds device scalars)."""
    cond = cond_module(audio)
    info: Dict = {"samples": audio.shape[0]}
    if train_disc:
        mp, mr = model(cond=cond, audio=audio, audio_lens=audio_lens, n_timesteps=n_timesteps,
                       train_disc=True)
        loss = scales.disc_loss_mp_scale * mp + scales.disc_loss_mr_scale * mr
        assert loss.requires_grad == is_training
        info.update(loss_d=loss.detach(), disc_loss_mp=mp.detach(), disc_loss_mr=mr.detach())
        return loss, info
    g_mp, g_mr, fm_mp, fm_mr, mel = model(cond=cond, audio=audio, audio_lens=audio_lens,
                                          n_timesteps=n_timesteps, train_disc=False)
    loss = (scales.gen_loss_mp_scale * g_mp + scales.gen_loss_mr_scale * g_mr
            + scales.feat_map_loss_mp_scale * fm_mp + scales.feat_map_loss_mr_scale * fm_mr
            + scales.mel_recon_loss_scale * mel)
    assert loss.requires_grad == is_training
    info.update(loss_g=loss.detach(), gen_loss_mp=g_mp.detach(), gen_loss_mr=g_mr.detach(),
                feat_map_loss_mp=fm_mp.detach(), feat_map_loss_mr=fm_mr.detach(),
                mel_recon_loss=mel.detach())
    return loss, info


def grad_groups(gan: nn.Module, disc: bool):
    """Parameter groups whose gradients become final together (GradReducer.prepare(groups=...)):
    G-step: one per Fourier branch (the condition encoder and everything else accumulate over all
    branches and close last); D-step: one per period discriminator."""
    from . import fused
    try:
        if disc:
            return [list(d.parameters()) for d in gan.discriminator[0].discriminators]
        gen = gan.generator if hasattr(gan, "generator") else gan
        # exactly the parameters ModelEvalFn hands over per branch (the per-branch condition path
        # belongs to CondPathFn, whose backward runs after every branch)
        return [fused.branch_params(e) for e in gen.estimators]
    except (AttributeError, TypeError, IndexError):
        return None          # not this package's GAN: one group, autograd hooks only


@dataclass
class GanStepper:
    """Batch schedule of finetune.py:569-631 around compute_loss_stage2."""
    gan: nn.Module
    cond_module: nn.Module
    n_timesteps: int = 1
    gen_start_batch_idx: int = 1000          # finetune.py:614
    scales: GanLossScales = field(default_factory=GanLossScales)
    optimizer_d: Optional[Callable[[], None]] = None
    optimizer_g: Optional[Callable[[], None]] = None
    reducer: GradReducer = field(default_factory=GradReducer)
    batch_idx_train: int = 0
    train_disc: bool = True

    def step(self, audio: Tensor, audio_lens: Tensor) -> Dict:
        """Consume ONE batch: a D-step or a G-step, as the reference's flag dictates."""
        self.batch_idx_train += 1
        disc = self.train_disc
        params = list((self.gan.discriminator if disc else self.gan.generator).parameters())
        # zero grads; a Fourier branch's / period discriminator's bucket leaves when its launch lane
        # has finished (fused.deliver_grads), everything else from autograd hooks during backward
        self.reducer.prepare(params, groups=grad_groups(self.gan, disc))
        loss, info = compute_loss_stage2(audio, audio_lens, self.gan, self.cond_module,
                                         self.n_timesteps, self.scales, True, disc)
        loss.backward()
        self.reducer.finish()
        opt = self.optimizer_d if disc else self.optimizer_g
        if opt is not None:
            opt()
        info["train_disc"] = disc
        # finetune.py:614-615,626: generator steps begin after gen_start_batch_idx batches
        if disc:
            if self.batch_idx_train >= self.gen_start_batch_idx:
                self.train_disc = False
        else:
            self.train_disc = True
        return info
