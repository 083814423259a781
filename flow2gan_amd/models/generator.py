"""Flow2GAN generator on MI355X: same constructor arguments, attributes, method signatures and
state-dict keys as the reference (flow2gan/models/generator.py:30-366), executed by the fused HIP
schedules in flow2gan_amd/fused.py.

Extra keyword-only arguments (`noise`, `t`, `branch_weights`) let a caller inject the random
draws the reference takes from torch's global generator, which is how the parity tests drive it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os

import torch
from torch import Tensor, nn

from .. import fused, ops
from .._opts import opt
from .modules import AudioConvNeXt, CondEncoder, LinearFilterSpectrogram

# inference: the time paths of all Euler steps are computed once, ahead of the loop (0: per step)
TIME_AHEAD = opt("time_ahead", True)


class CondRows:
    """Condition-encoder output held channels-last: rows (B*frames, channels)."""

    def __init__(self, rows: Tensor, batch: int, frames: int):
        self.rows, self.batch, self.frames = rows, batch, frames

    def to_bct(self) -> Tensor:
        Cc = self.rows.shape[1]
        out = ops.empty(self.batch, Cc, self.frames, device=self.rows.device)
        return ops.rows_to_bct(out, self.rows.detach(), self.batch, Cc, self.frames)


def _lens_list(audio_lens) -> Optional[List[int]]:
    if audio_lens is None:
        return None
    if isinstance(audio_lens, Tensor):
        return [int(v) for v in audio_lens.tolist()]
    return [int(v) for v in audio_lens]


class BaseAudioGenerator(nn.Module):
    """reference generator.py:30-271."""

    def __init__(
        self,
        sampling_rate: int = 24000,
        n_ffts: Tuple[int, ...] = (512, 256, 128),
        hop_lengths: Tuple[int, ...] = (256, 128, 64),
        channels: Tuple[int, ...] = (768, 512, 384),
        time_embed_channels: int = 512,
        hidden_factor: int = 3,
        conv_kernel_sizes: Tuple[int, ...] = (7, 7, 7),
        num_layers: Tuple[int, ...] = (8, 8, 8),
        use_cond_encoder: bool = True,
        cond_dim: int = 100,
        cond_hop_length: int = 256,
        cond_enc_channels: int = 512,
        cond_enc_hidden_factor: int = 3,
        cond_enc_conv_kernel_size: int = 7,
        cond_enc_num_layers: int = 4,
        residual_scale: Optional[float] = 1.0,
        init_noise_scale: float = 0.1,
        pred_x1: bool = True,
        branch_reduction: str = "mean",
        spec_scaling_loss: bool = True,
        loss_n_filters: int = 256,
        loss_n_fft: int = 1024,
        loss_hop_length: int = 256,
        loss_power: float = 0.5,
        loss_eps: float = 1e-7,
        loss_scale_min: float = 1e-2,
        loss_scale_max: float = 1e+2,
        branch_dropout: float = 0.05,
    ):
        super().__init__()
        self.num_branches = len(n_ffts)
        assert len(hop_lengths) == self.num_branches
        assert len(channels) == self.num_branches
        assert len(conv_kernel_sizes) == self.num_branches
        assert len(num_layers) == self.num_branches
        # Every named config (config.py:31-95) uses the defaults; the other switches of the
        # reference's constructor (use_cond_encoder, pred_x1, branch_reduction, spec_scaling_loss)
        # are implemented as well and pinned by reference vectors (tests/golden/tiny_switches.npz,
        # tiny_mse.npz).
        if branch_reduction not in ("mean", "sum"):
            raise ValueError(f"Unsupported branch_reduction: {branch_reduction}")
        self.sampling_rate = sampling_rate
        self.init_noise_scale = init_noise_scale
        self.pred_x1 = pred_x1
        self.branch_reduction = branch_reduction
        self.spec_scaling_loss = spec_scaling_loss
        self.loss_power = loss_power
        self.loss_eps = loss_eps
        self.loss_scale_min = loss_scale_min
        self.loss_scale_max = loss_scale_max
        self.branch_dropout = branch_dropout
        self.cond_hop_length = cond_hop_length

        if spec_scaling_loss:      # generator.py:85-93: the unweighted loss has no spectrogram module
            self.loss_spec = LinearFilterSpectrogram(
                sample_rate=sampling_rate, n_fft=loss_n_fft, hop_length=loss_hop_length,
                n_filter=loss_n_filters, center=True, power=2)
        if use_cond_encoder:   # generator.py:86-97: without it the estimators see the raw condition
            self.cond_encoder = CondEncoder(
                cond_dim=cond_dim, channels=cond_enc_channels, hidden_factor=cond_enc_hidden_factor,
                conv_kernel_size=cond_enc_conv_kernel_size, num_layers=cond_enc_num_layers,
                residual_scale=residual_scale)
        cond_channels = cond_enc_channels if use_cond_encoder else cond_dim
        self.estimators = nn.ModuleList([
            AudioConvNeXt(
                n_fft=n_ffts[i], hop_length=hop_lengths[i], cond_hop_length=cond_hop_length,
                channels=channels[i], cond_channels=cond_channels,
                time_embed_channels=time_embed_channels, hidden_factor=hidden_factor,
                conv_kernel_size=conv_kernel_sizes[i], num_layers=num_layers[i],
                residual_scale=residual_scale)
            for i in range(self.num_branches)])
        self.apply(self._init_weights)

    @torch.no_grad()
    def _init_weights(self, m):
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.015)
            if hasattr(m, "bias") and isinstance(m.bias, Tensor):
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ fused pieces
    def encode_cond(self, mel: Tensor) -> CondRows:
        """cond_encoder(mel) (generator.py:311-312): (B, n_mels, F) -> CondRows."""
        B, _, Fm = mel.shape
        if not hasattr(self, "cond_encoder"):      # generator.py:311-314: the mel itself
            return self._as_cond_rows(mel)
        rows = fused.CondEncoderFn.apply(mel, self.training, None,
                                         *fused.cond_encoder_params(self.cond_encoder))
        return CondRows(rows, B, Fm)

    def _as_cond_rows(self, cond) -> CondRows:
        if isinstance(cond, CondRows):
            return cond
        B, Cc, Fm = cond.shape
        rows = ops.empty(B * Fm, Cc, device=cond.device)
        ops.bct_to_rows(rows, cond.detach().contiguous(), B, Cc, Fm)
        return CondRows(rows, B, Fm)

    def cond_paths(self, cond: CondRows, T: int) -> List[Tensor]:
        """Per-branch cond_mlp + stacked cond_proj, once per forward/infer: the reference
        recomputes them in every model evaluation (modules.py:620, 482) with identical results."""
        outs = []
        # three independent chains of small / mid-size GEMMs: one launch lane each (serially they
        # are 0.45 ms at the head of every inference, with the chip mostly idle)
        lanes = ops.Lanes(cond.rows.device, len(self.estimators), "condpath")
        for i, est in enumerate(self.estimators):
            F = 1 + T // est.hop_length
            up = est.cond_upsample_factor
            Fce = (F + up - 1) // up
            with lanes.lane(i):
                outs.append(fused.CondPathFn.apply(cond.rows, cond.batch, cond.frames, Fce,
                                                   *fused.cond_path_params(est.decoder)))
        lanes.join()
        return outs

    def _metas(self):
        m = tuple((e.n_fft, e.hop_length, e.cond_upsample_factor, e.ifft.window)
                  for e in self.estimators)
        # weight of a branch in the reduction (generator.py:165-168)
        return m + (("scale", 1.0 / self.num_branches if self.branch_reduction == "mean" else 1.0),)

    def _draw_branch_weights(self, B: int, device) -> Optional[Tensor]:
        """Branch dropout (generator.py:145-162).  Returns (num_branches, B) weights or None."""
        if not (self.training and self.branch_dropout > 0.0 and self.num_branches > 1):
            return None
        nb = self.num_branches
        idx = torch.randint(0, nb, (B,), device=device)
        mask = torch.ones((B, nb), device=device, dtype=torch.float32)
        mask[torch.arange(B, device=device), idx] = 0.0
        mask = mask * (nb / (nb - 1))
        w = torch.where(torch.rand((B, 1), device=device) < self.branch_dropout, mask,
                        torch.ones_like(mask))
        return w.t().contiguous()

    def _branch_flat(self):
        flat, nparams = [], []
        for est in self.estimators:
            p = fused.branch_params(est)
            nparams.append(len(p))
            flat += p
        return flat, nparams

    def model_eval(self, x: Tensor, t: Tensor, cprojs: List[Tensor], lens_cpu,
                   branch_weights: Optional[Tensor] = None, te_pre=None) -> Tensor:
        """te_pre (inference only): this evaluation's time-scale rows per branch, computed ahead by
        fused.time_paths_ahead."""
        flat, nparams = self._branch_flat()
        metas = self._metas()
        if te_pre is not None:
            metas = metas + (("te", te_pre),)
        return fused.ModelEvalFn.apply(x, t, branch_weights, metas, lens_cpu,
                                       self.training, tuple(nparams), *cprojs, *flat)

    def process_model(self, x: Tensor, cond, t: Optional[Tensor] = None,
                      audio_lens: Optional[Tensor] = None, *, cprojs=None,
                      branch_weights: Optional[Tensor] = None) -> Tensor:
        """generator.py:129-170.  `cond` is the condition-encoder output (CondRows, or the
        reference's (B, C, F) tensor)."""
        assert t is not None, "the flow-matching generator is always time-conditioned"
        cond = self._as_cond_rows(cond)
        if cprojs is None:
            cprojs = self.cond_paths(cond, x.shape[-1])
        if branch_weights is None:
            branch_weights = self._draw_branch_weights(x.shape[0], x.device)
        return self.model_eval(x, t.flatten().contiguous().float(), cprojs, _lens_list(audio_lens),
                               branch_weights)

    def compute_loss(self, pred: Tensor, ref: Tensor, audio_lens: Tensor,
                     gt_audio: Optional[Tensor] = None) -> Tensor:
        """generator.py:172-200 (ref is gt_audio for the x1-prediction objective)."""
        if not self.spec_scaling_loss:     # generator.py:181-184: masked mean squared error
            return fused.MseLossFn.apply(pred, ref, _lens_list(audio_lens))
        ls = self.loss_spec
        return fused.FmLossFn.apply(pred, ref, _lens_list(audio_lens), ls.n_fft, ls.hop_length,
                                    ls.fb, self.loss_eps, self.loss_power, self.loss_scale_min,
                                    self.loss_scale_max, None if gt_audio is ref else gt_audio)

    def forward(self, x0: Tensor, x1: Tensor, cond, audio_lens: Optional[Tensor] = None, *,
                t: Optional[Tensor] = None, branch_weights: Optional[Tensor] = None) -> Tensor:
        """Flow-matching loss (generator.py:202-234)."""
        B = x0.shape[0]
        if t is None:
            t = torch.rand((B, 1), device=x0.device, dtype=x0.dtype)
        tf = t.flatten().contiguous().float()
        x = torch.empty_like(x0)
        ops.axpby_rows(x, x0.contiguous(), x1.contiguous(), ca=(1.0 - tf).contiguous(), cb=tf)
        pred = self.process_model(x, cond, t=tf, audio_lens=audio_lens,
                                  branch_weights=branch_weights)
        if self.pred_x1:
            ref = x1
        else:                                    # velocity objective (generator.py:218): x1 - x0
            ref = torch.empty_like(x1)
            ops.axpby_rows(ref, x1.contiguous(), x0.contiguous(), sa=1.0, sb=-1.0)
        return self.compute_loss(pred=pred, ref=ref, audio_lens=audio_lens, gt_audio=x1)

    def _time_ahead(self, B: int, n_timesteps: int, device):
        """Inference: every step's t is known before the solver starts, so the time paths of all
        steps run as one batch on their own launch lanes, next to the condition encoder / condition
        paths (fused.time_paths_ahead).  Returns None when gradients are wanted or F2G_TIME_AHEAD=0;
        the caller joins the lanes (`[0].join()`) before the first model evaluation."""
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if grad or not TIME_AHEAD:
            return None
        t_span = torch.linspace(0, 1, n_timesteps + 1)
        t_all = ops.empty(n_timesteps * B, device=device)
        for k in range(n_timesteps):
            ops.fill_(t_all[k * B: (k + 1) * B], float(t_span[k]))
        flat, nparams = self._branch_flat()
        tlanes, te_ahead = fused.time_paths_ahead(flat, nparams, t_all, n_timesteps)
        return tlanes, te_ahead, t_all

    def infer(self, noise: Tensor, cond, audio_lens: Optional[Tensor] = None,
              n_timesteps: int = 1, clamp_pred: bool = False, *, _time_ahead=None) -> Tensor:
        """Euler solver (generator.py:236-271)."""
        cond = self._as_cond_rows(cond)
        B, T = noise.shape
        t_span = torch.linspace(0, 1, n_timesteps + 1)
        t, dt = float(t_span[0]), float(t_span[1] - t_span[0])
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        te_ahead = tlanes = t_all = None
        if _time_ahead is None:
            _time_ahead = self._time_ahead(B, n_timesteps, noise.device)
        if _time_ahead is not None:
            tlanes, te_ahead, t_all = _time_ahead
        cprojs = self.cond_paths(cond, T)
        if tlanes is not None:
            tlanes.join()
        lens_cpu = _lens_list(audio_lens)
        x = noise.contiguous()
        tdev = ops.empty(B, device=noise.device)
        for step in range(1, n_timesteps + 1):
            if te_ahead is not None:
                tk = t_all[(step - 1) * B: step * B]
            else:
                tk = ops.fill_(ops.empty(B, device=noise.device), t) if grad else ops.fill_(tdev, t)
            bw = self._draw_branch_weights(B, noise.device)
            pred = self.model_eval(x, tk, cprojs, lens_cpu, bw,
                                   None if te_ahead is None else te_ahead[step - 1])
            # generator.py:263-264: x += vt*dt with vt = (pred - x)/(1 - t) (x1 prediction) or pred
            a, b = (1.0 - dt / (1.0 - t), dt / (1.0 - t)) if self.pred_x1 else (1.0, dt)
            if grad:
                x = fused.AxpbyFn.apply(x, pred, a, b)
            else:
                x = ops.axpby_rows(pred, x, pred, sa=a, sb=b)
            t = float(t_span[step])
        if clamp_pred:
            if grad:
                raise NotImplementedError("clamp_pred is an inference-only option")
            x = ops.clamp(x, x, -1.0, 1.0)
        return x


class MelAudioGenerator(BaseAudioGenerator):
    """Mel-conditioned generator (reference generator.py:274-366)."""

    def __init__(self, n_mels: int = 100, mel_n_fft: int = 1024, mel_hop_length: int = 256,
                 max_add_noise_scale: float = 0.0, **kwargs):
        super().__init__(cond_dim=n_mels, cond_hop_length=mel_hop_length, **kwargs)
        self.n_mels = n_mels
        self.mel_n_fft = mel_n_fft
        self.mel_hop_length = mel_hop_length
        self.max_add_noise_scale = max_add_noise_scale

    def _augment(self, cond: Tensor) -> Tensor:
        if self.training and self.max_add_noise_scale > 0.0:
            # generator.py:306-309 (off in every shipped config)
            e = torch.randn_like(cond) * torch.rand(cond.shape[0], 1, 1, device=cond.device) \
                * self.max_add_noise_scale
            cond = cond + e
        return cond

    def forward(self, cond: Tensor, audio: Tensor, audio_lens: Tensor, *,
                noise: Optional[Tensor] = None, t: Optional[Tensor] = None,
                branch_weights: Optional[Tensor] = None) -> Tensor:
        """Flow-matching loss (generator.py:294-325)."""
        cond_rows = self.encode_cond(self._augment(cond))
        if noise is None:
            noise = torch.randn_like(audio) * self.init_noise_scale
        return super().forward(x0=noise, x1=audio, cond=cond_rows, audio_lens=audio_lens, t=t,
                               branch_weights=branch_weights)

    def infer(self, cond: Tensor, audio_lens: Optional[Tensor] = None, n_timesteps: int = 1,
              clamp_pred: bool = False, *, noise: Optional[Tensor] = None) -> Tensor:
        """Euler inference from a mel (generator.py:327-366)."""
        # (the time paths of all steps start first: they overlap the condition encoder, whose
        # 6016-row launches leave most of the chip idle)
        ahead = self._time_ahead(cond.shape[0], n_timesteps, cond.device)
        cond_rows = self.encode_cond(self._augment(cond))
        if noise is None:
            if audio_lens is None:
                length = cond.shape[2] * self.mel_hop_length
            else:
                length = max(_lens_list(audio_lens))
            noise = torch.randn((cond.shape[0], length), device=cond.device,
                                dtype=cond.dtype) * self.init_noise_scale
        return super().infer(noise=noise, cond=cond_rows, audio_lens=audio_lens,
                             n_timesteps=n_timesteps, clamp_pred=clamp_pred, _time_ahead=ahead)
