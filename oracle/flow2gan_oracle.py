"""CPU oracle for the Flow2GAN hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

This file is a from-scratch CPU restatement, in plain PyTorch fp32 ops, of the
reference algorithm that the HIP path in ``flow2gan_amd/`` must reproduce.  It
exists only so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` have something to check against / time on
the GPU box, where ``/root/reference`` does not exist.  Nothing under
``flow2gan_amd/`` may import it.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference
(``/root/reference/flow2gan`` with stand-ins for the absent third-party
modules) in the build container and dumps input/output vectors into
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against
them.  The torchaudio arithmetic (not vendored by the reference, version
unpinned in ``requirements.txt:8``) is restated in ``taudio_*`` below from its
published semantics and pinned by the reference's own wav<->mel fixtures
(``test_data/wav`` <-> ``test_data/mel``).

Every class cites the reference file:line it follows.  State-dict keys equal
the reference's so that checkpoints interchange.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# ----------------------------------------------------------------------------
# torchaudio restatement (third-party dependency, absent from /root/reference)
# ----------------------------------------------------------------------------
def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def _triangles(all_freqs: Tensor, f_pts: Tensor) -> Tensor:
    """torchaudio.functional._create_triangular_filterbank: (n_freqs, n_filt)."""
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def taudio_melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int,
                           sample_rate: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk')."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    return _triangles(all_freqs, f_pts)


def taudio_linear_fbanks(n_freqs: int, f_min: float, f_max: float, n_filter: int,
                         sample_rate: int) -> Tensor:
    """torchaudio.functional.linear_fbanks."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    f_pts = torch.linspace(f_min, f_max, n_filter + 2)
    return _triangles(all_freqs, f_pts)


def taudio_spectrogram(x: Tensor, n_fft: int, hop: int, window: Tensor,
                       power: Optional[float]) -> Tensor:
    """torchaudio.transforms.Spectrogram(center=True, pad_mode='reflect',
    onesided=True, normalized=False): (..., T) -> (..., n_fft/2+1, frames)."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    spec = torch.stft(x2, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    spec = spec.reshape(shape[:-1] + spec.shape[-2:])
    if power is None:
        return spec
    if power == 1.0:
        return spec.abs()
    return spec.abs().pow(power)


class OracleSpectrogram(nn.Module):
    """Stand-in for torchaudio.transforms.Spectrogram (persistent `window` buffer)."""

    def __init__(self, n_fft: int, hop_length: int, power: Optional[float] = 2.0):
        super().__init__()
        self.n_fft, self.hop_length, self.power = n_fft, hop_length, power
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x: Tensor) -> Tensor:
        return taudio_spectrogram(x, self.n_fft, self.hop_length, self.window, self.power)


class OracleMelScale(nn.Module):
    def __init__(self, n_mels: int, sample_rate: int, n_stft: int):
        super().__init__()
        fb = taudio_melscale_fbanks(n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate)
        self.register_buffer("fb", fb)

    def forward(self, spec: Tensor) -> Tensor:
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class OracleMelSpectrogram(nn.Module):
    """Stand-in for torchaudio.transforms.MelSpectrogram(center=True, power=p,
    norm=None, mel_scale='htk', f_min=0, f_max=sr//2)."""

    def __init__(self, sample_rate: int, n_fft: int, hop_length: int, n_mels: int,
                 power: float = 2.0):
        super().__init__()
        self.spectrogram = OracleSpectrogram(n_fft, hop_length, power)
        self.mel_scale = OracleMelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, x: Tensor) -> Tensor:
        return self.mel_scale(self.spectrogram(x))


# ----------------------------------------------------------------------------
# helpers: reference flow2gan/utils.py:41-66, 221-244
# ----------------------------------------------------------------------------
def pad_mask(lengths: Tensor, max_len: int = 0) -> Tensor:
    """True at padded positions (utils.py:41-66)."""
    n = max(max_len, int(lengths.max()))
    return torch.arange(n, device=lengths.device)[None, :] >= lengths[:, None]


def clipped_log(x: Tensor, clip_val: float = 1e-7) -> Tensor:
    """utils.py:221-232."""
    return torch.log(torch.clip(x, min=clip_val))


def fit_length(x: Tensor, length: int) -> Tensor:
    """Truncate or zero-extend the last axis (utils.py:235-244)."""
    if length <= x.shape[-1]:
        return x[..., :length]
    return F.pad(x, (0, length - x.shape[-1]))


# ----------------------------------------------------------------------------
# primitives: reference flow2gan/models/modules.py
# ----------------------------------------------------------------------------
def pack_complex(spec: Tensor) -> Tensor:
    """(B, n, F) complex -> (B, 2n, F) real, channels [Re | Im] (modules.py:31-38)."""
    return torch.cat([spec.real, spec.imag], dim=1)


def unpack_complex(packed: Tensor) -> Tensor:
    """Inverse of pack_complex (modules.py:41-49)."""
    n = packed.shape[1] // 2
    return torch.complex(packed[:, :n].contiguous(), packed[:, n:].contiguous())


class STFT(nn.Module):
    """modules.py:52-84."""

    def __init__(self, n_fft: int, hop_length: int):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, audio: Tensor, audio_lens: Optional[Tensor] = None):
        spec = torch.stft(audio, n_fft=self.n_fft, hop_length=self.hop_length,
                          win_length=self.n_fft, window=self.window, center=True,
                          return_complex=True, onesided=True)
        if audio_lens is None:
            return spec, None
        lens = 1 + torch.div(audio_lens, self.hop_length, rounding_mode="floor")
        assert spec.shape[2] == int(lens.max())
        return spec, lens


class ISTFT(nn.Module):
    """modules.py:87-116."""

    def __init__(self, n_fft: int, hop_length: int):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, spec: Tensor) -> Tensor:
        return torch.istft(spec, n_fft=self.n_fft, hop_length=self.hop_length,
                           win_length=self.n_fft, window=self.window, center=True,
                           onesided=True, return_complex=False)


class LogMelSpectrogram(nn.Module):
    """modules.py:119-143 (A1)."""

    def __init__(self, sampling_rate=24000, n_fft=1024, hop_length=256, n_mels=100,
                 center=True, power=1):
        super().__init__()
        assert center
        self.mel = OracleMelSpectrogram(sampling_rate, n_fft, hop_length, n_mels, power=power)

    def forward(self, waveform: Tensor) -> Tensor:
        return clipped_log(self.mel(waveform))


class LinearFilterSpectrogram(nn.Module):
    """modules.py:146-214 (A15): power-2 spectrogram -> linear triangular bank."""

    def __init__(self, sample_rate: int, n_filter: int, n_fft: int, hop_length: int,
                 power: float = 2.0):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.spectrogram = OracleSpectrogram(n_fft, hop_length, power)
        fb = taudio_linear_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_filter,
                                  sample_rate)
        self.register_buffer("fb", fb)

    def forward(self, waveform: Tensor) -> Tensor:
        s = self.spectrogram(waveform)
        return torch.matmul(s.transpose(-1, -2), self.fb).transpose(-1, -2)


def sinusoid_embedding(t: Tensor, dim: int, scale: float = 1000.0) -> Tensor:
    """modules.py:217-232 (A6); always float32."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=t.device).float() * -k)
    arg = scale * t.unsqueeze(1) * freqs.unsqueeze(0)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


class _GradSignLimiter(torch.autograd.Function):
    """modules.py:236-256 (A.5)."""

    @staticmethod
    def forward(ctx, p, lo, hi):
        ctx.save_for_backward(p)
        ctx.lo, ctx.hi = lo, hi
        return p

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = g * torch.where((g > 0) & (p < ctx.lo), -1.0, 1.0)
        g = g * torch.where((g < 0) & (p > ctx.hi), -1.0, 1.0)
        return g, None, None


def maybe_limit(p: Tensor, lo: float, hi: float, training: bool, prob: float = 0.6) -> Tensor:
    """modules.py:259-270: one Python-RNG draw per call when training."""
    if training and random.random() < prob:
        return _GradSignLimiter.apply(p, lo, hi)
    return p


class ChannelScale(nn.Module):
    """modules.py:273-283 (A8)."""

    def __init__(self, channels: int, scale: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels, 1), scale))

    def forward(self, x: Tensor) -> Tensor:
        return x * maybe_limit(self.scale, 0.5, 1.0, self.training)


class BiasNorm(nn.Module):
    """modules.py:286-416 (A7): y = x * mean_c((x-b)^2)^-0.5 * exp(log_scale).
    The reference's custom Function recomputes the same expression under
    autograd in backward, so plain autograd of the expression is equivalent."""

    def __init__(self, num_channels: int, channel_dim: int = 1):
        super().__init__()
        assert channel_dim == 1
        self.log_scale = nn.Parameter(torch.tensor(1.0))
        self.bias = nn.Parameter(torch.empty(num_channels).normal_(mean=0, std=1e-2))

    def forward(self, x: Tensor) -> Tensor:
        ls = maybe_limit(self.log_scale, -1.5, 1.5, self.training)
        b = self.bias[:, None]
        scales = torch.mean((x - b) ** 2, dim=1, keepdim=True) ** -0.5 * ls.exp()
        return x * scales


class ConvNeXtBlock(nn.Module):
    """modules.py:419-495 (A9)."""

    def __init__(self, channels, hidden_channels, conv_kernel_size=7, cond_channels=None,
                 time_embed_channels=None, residual_scale=1.0):
        super().__init__()
        self.dwconv = nn.Conv1d(channels, channels, conv_kernel_size,
                                padding=conv_kernel_size // 2, groups=channels)
        self.norm = BiasNorm(channels)
        self.pwconv1 = nn.Conv1d(channels, hidden_channels, 1)
        self.act = nn.PReLU(hidden_channels)
        self.pwconv2 = nn.Conv1d(hidden_channels, channels, 1)
        if cond_channels is not None:
            self.cond_proj = nn.Conv1d(cond_channels, channels, 1)
        if time_embed_channels is not None:
            self.time_embed_proj = nn.Linear(time_embed_channels, channels)
        if residual_scale is not None:
            self.residual_scale = ChannelScale(channels)

    def forward(self, x, cond=None, time_embed=None, mask=None):
        res = x
        if mask is not None:
            x = x * mask
        x = self.norm(self.dwconv(x))
        if cond is not None:
            x = x + self.cond_proj(cond)
        if time_embed is not None:
            x = x * (1.0 + self.time_embed_proj(time_embed).unsqueeze(-1))
        x = self.pwconv2(self.act(self.pwconv1(x)))
        if hasattr(self, "residual_scale"):
            res = self.residual_scale(res)
        return x + res


class CondEncoder(nn.Module):
    """modules.py:498-542 (A12)."""

    def __init__(self, cond_dim=100, channels=512, hidden_factor=3, conv_kernel_size=7,
                 num_layers=4, residual_scale=1.0):
        super().__init__()
        self.in_proj = nn.Conv1d(cond_dim, channels, 3, padding=1)
        self.in_norm = BiasNorm(channels)
        self.blocks = nn.ModuleList([
            ConvNeXtBlock(channels, int(channels * hidden_factor), conv_kernel_size,
                          residual_scale=residual_scale) for _ in range(num_layers)])

    def forward(self, x, mask=None):
        x = self.in_norm(self.in_proj(x))
        for blk in self.blocks:
            x = blk(x, mask=mask)
        return x


class ConvNeXtDecoder(nn.Module):
    """modules.py:545-627 (A10)."""

    def __init__(self, in_channels, out_channels, channels=512, cond_channels=512,
                 time_embed_channels=512, hidden_factor=3, conv_kernel_size=7, num_layers=8,
                 residual_scale=1.0):
        super().__init__()
        self.time_embed_channels = time_embed_channels
        self.in_proj = nn.Conv1d(in_channels, channels, 1)
        self.in_norm = BiasNorm(channels)
        th = int(time_embed_channels * hidden_factor)
        self.time_mlp = nn.Sequential(nn.Linear(time_embed_channels, th), nn.SiLU(),
                                      nn.Linear(th, time_embed_channels))
        ch = int(cond_channels * hidden_factor)
        self.cond_mlp = nn.Sequential(nn.Conv1d(cond_channels, ch, 1), nn.PReLU(ch),
                                      nn.Conv1d(ch, cond_channels, 1))
        self.blocks = nn.ModuleList([
            ConvNeXtBlock(channels, int(channels * hidden_factor), conv_kernel_size,
                          cond_channels=cond_channels, time_embed_channels=time_embed_channels,
                          residual_scale=residual_scale) for _ in range(num_layers)])
        self.out_proj = nn.Conv1d(channels, out_channels, 1)

    def forward(self, x, cond, t=None, mask=None):
        x = self.in_norm(self.in_proj(x))
        te = None
        if t is not None:
            te = self.time_mlp(sinusoid_embedding(t, self.time_embed_channels))
        cond = self.cond_mlp(cond)
        for blk in self.blocks:
            x = blk(x, cond=cond, time_embed=te, mask=mask)
        return self.out_proj(x)


class AudioConvNeXt(nn.Module):
    """modules.py:630-721 (A2-A5, A11)."""

    def __init__(self, n_fft=512, hop_length=256, cond_hop_length=256, channels=768,
                 cond_channels=512, time_embed_channels=512, hidden_factor=3,
                 conv_kernel_size=7, num_layers=8, residual_scale=1.0):
        super().__init__()
        self.fft = STFT(n_fft, hop_length)
        self.ifft = ISTFT(n_fft, hop_length)
        assert cond_hop_length % hop_length == 0
        self.cond_upsample_factor = cond_hop_length // hop_length
        self.decoder = ConvNeXtDecoder(n_fft + 2, n_fft + 2, channels, cond_channels,
                                       time_embed_channels, hidden_factor, conv_kernel_size,
                                       num_layers, residual_scale)

    def upsample_cond(self, cond: Tensor, frames: int) -> Tensor:
        if self.cond_upsample_factor != 1:
            cond = torch.repeat_interleave(cond, self.cond_upsample_factor, dim=2)
        return fit_length(cond, frames)

    def forward(self, audio, cond, t=None, audio_lens=None):
        T = audio.shape[-1]
        spec, lens = self.fft(audio, audio_lens)
        x = pack_complex(spec)
        cond = self.upsample_cond(cond, spec.shape[-1])
        mask = None
        if lens is not None:
            mask = pad_mask(lens).logical_not().unsqueeze(1)
        x = self.decoder(x, cond=cond, t=t, mask=mask)
        if mask is not None:
            x = x * mask
        return fit_length(self.ifft(unpack_complex(x)), T)


# ----------------------------------------------------------------------------
# generator: reference flow2gan/models/generator.py
# ----------------------------------------------------------------------------
class BaseAudioGenerator(nn.Module):
    """generator.py:30-271 (A13-A16)."""

    def __init__(self, sampling_rate=24000, n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64),
                 channels=(768, 512, 384), time_embed_channels=512, hidden_factor=3,
                 conv_kernel_sizes=(7, 7, 7), num_layers=(8, 8, 8), use_cond_encoder=True,
                 cond_dim=100, cond_hop_length=256, cond_enc_channels=512,
                 cond_enc_hidden_factor=3, cond_enc_conv_kernel_size=7, cond_enc_num_layers=4,
                 residual_scale=1.0, init_noise_scale=0.1, pred_x1=True,
                 branch_reduction="mean", spec_scaling_loss=True, loss_n_filters=256,
                 loss_n_fft=1024, loss_hop_length=256, loss_power=0.5, loss_eps=1e-7,
                 loss_scale_min=1e-2, loss_scale_max=1e2, branch_dropout=0.05):
        super().__init__()
        assert use_cond_encoder and pred_x1 and spec_scaling_loss and branch_reduction == "mean"
        self.num_branches = len(n_ffts)
        self.sampling_rate = sampling_rate
        self.init_noise_scale = init_noise_scale
        self.loss_power, self.loss_eps = loss_power, loss_eps
        self.loss_scale_min, self.loss_scale_max = loss_scale_min, loss_scale_max
        self.branch_dropout = branch_dropout
        self.loss_spec = LinearFilterSpectrogram(sampling_rate, loss_n_filters, loss_n_fft,
                                                 loss_hop_length, power=2)
        self.cond_encoder = CondEncoder(cond_dim, cond_enc_channels, cond_enc_hidden_factor,
                                        cond_enc_conv_kernel_size, cond_enc_num_layers,
                                        residual_scale)
        self.estimators = nn.ModuleList([
            AudioConvNeXt(n_ffts[i], hop_lengths[i], cond_hop_length, channels[i],
                          cond_enc_channels, time_embed_channels, hidden_factor,
                          conv_kernel_sizes[i], num_layers[i], residual_scale)
            for i in range(self.num_branches)])
        self.apply(self._init_weights)

    @torch.no_grad()
    def _init_weights(self, m):  # generator.py:122-127
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.015)
            if isinstance(getattr(m, "bias", None), Tensor):
                nn.init.constant_(m.bias, 0)

    def process_model(self, x, cond, t=None, audio_lens=None):  # generator.py:129-170
        outs = torch.stack([
            est(audio=x, cond=cond, t=None if t is None else t.flatten(), audio_lens=audio_lens)
            for est in self.estimators], dim=1)
        if self.training and self.branch_dropout > 0.0 and self.num_branches > 1:
            B, nb = outs.shape[0], self.num_branches
            idx = torch.randint(0, nb, (B,), device=outs.device)
            keep = torch.ones((B, nb), device=outs.device, dtype=outs.dtype)
            keep[torch.arange(B, device=outs.device), idx] = 0.0
            keep = keep * (nb / (nb - 1))
            w = torch.where(torch.rand((B, 1), device=outs.device) < self.branch_dropout, keep,
                            torch.ones_like(keep))
            outs = outs * w.unsqueeze(-1)
        return outs.mean(dim=1)

    def compute_loss(self, pred, ref, audio_lens, gt_audio):  # generator.py:172-200
        err = pred - ref
        gt_spec = self.loss_spec(gt_audio)
        err_spec = self.loss_spec(err)
        lens = torch.div(audio_lens, self.loss_spec.hop_length, rounding_mode="floor") + 1
        assert err_spec.shape[2] == int(lens.max())
        mask = pad_mask(lens).logical_not().unsqueeze(1)
        scale = ((gt_spec + self.loss_eps) ** -self.loss_power).clamp(
            min=self.loss_scale_min, max=self.loss_scale_max)
        loss = err_spec * scale
        return (loss * mask).sum() / (mask.sum() * err_spec.shape[1])

    def fm_loss(self, x0, x1, cond, audio_lens, t=None):  # generator.py:202-234
        if t is None:
            t = torch.rand((x0.shape[0], 1), device=x0.device, dtype=x0.dtype)
        x = (1.0 - t) * x0 + t * x1
        pred = self.process_model(x=x, cond=cond, t=t, audio_lens=audio_lens)
        return self.compute_loss(pred, x1, audio_lens, x1)

    def euler(self, noise, cond, audio_lens=None, n_timesteps=1, clamp_pred=False):
        """generator.py:236-271."""
        ts = torch.linspace(0, 1, n_timesteps + 1, device=noise.device)
        t, dt = ts[0], ts[1] - ts[0]
        x = noise
        for step in range(1, len(ts)):
            pred = self.process_model(x=x, cond=cond, t=t[None, None].expand(noise.shape[0], 1),
                                      audio_lens=audio_lens)
            x = x + (pred - x) / (1 - t) * dt
            t = ts[step]
        return x.clamp(min=-1.0, max=1.0) if clamp_pred else x


class MelAudioGenerator(BaseAudioGenerator):
    """generator.py:274-366 (A17).  `noise` / `t` may be injected for parity tests."""

    def __init__(self, n_mels=100, mel_n_fft=1024, mel_hop_length=256, max_add_noise_scale=0.0,
                 **kw):
        super().__init__(cond_dim=n_mels, cond_hop_length=mel_hop_length, **kw)
        assert max_add_noise_scale == 0.0
        self.n_mels, self.mel_n_fft, self.mel_hop_length = n_mels, mel_n_fft, mel_hop_length

    def forward(self, cond, audio, audio_lens, noise=None, t=None):
        cond = self.cond_encoder(cond)
        if noise is None:
            noise = torch.randn_like(audio) * self.init_noise_scale
        return self.fm_loss(noise, audio, cond, audio_lens, t=t)

    def infer(self, cond, audio_lens=None, n_timesteps=1, clamp_pred=False, noise=None):
        cond = self.cond_encoder(cond)
        if noise is None:
            length = cond.shape[2] * self.mel_hop_length if audio_lens is None \
                else int(audio_lens.max())
            noise = torch.randn((cond.shape[0], length), device=cond.device,
                                dtype=cond.dtype) * self.init_noise_scale
        return self.euler(noise, cond, audio_lens, n_timesteps, clamp_pred)


# ----------------------------------------------------------------------------
# discriminators: reference flow2gan/models/discriminators.py
# ----------------------------------------------------------------------------
class DiscriminatorP(nn.Module):
    """discriminators.py:52-107 (A19)."""

    def __init__(self, period: int):
        super().__init__()
        self.period = period
        chans = [1, 32, 128, 512, 1024]
        convs = [nn.Conv2d(chans[i], chans[i + 1], (5, 1), (3, 1), padding=(2, 0))
                 for i in range(4)]
        convs.append(nn.Conv2d(1024, 1024, (5, 1), (1, 1), padding=(2, 0)))
        self.convs = nn.ModuleList(convs)
        self.conv_post = nn.Conv2d(1024, 1, (3, 1), 1, padding=(1, 0))

    def forward(self, x: Tensor):
        x = x.unsqueeze(1)
        b, c, t = x.shape
        if t % self.period != 0:
            x = F.pad(x, (0, self.period - t % self.period), "reflect")
            t = x.shape[-1]
        x = x.view(b, c, t // self.period, self.period)
        fmap = []
        for i, conv in enumerate(self.convs):
            x = F.leaky_relu(conv(x), 0.1)
            if i > 0:
                fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class DiscriminatorR(nn.Module):
    """discriminators.py:143-219 (A20)."""

    BANDS = ((0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0))

    def __init__(self, window_length: int, channels: int = 32):
        super().__init__()
        self.window_length = window_length
        self.spec_fn = OracleSpectrogram(window_length, int(window_length * 0.25), power=None)
        n = window_length // 2 + 1
        self.bands = [(int(lo * n), int(hi * n)) for lo, hi in self.BANDS]

        def stack():
            return nn.ModuleList([
                nn.Conv2d(2, channels, (3, 9), (1, 1), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 3), (1, 1), padding=(1, 1))])

        self.band_convs = nn.ModuleList([stack() for _ in self.bands])
        self.conv_post = nn.Conv2d(channels, 1, (3, 3), (1, 1), padding=(1, 1))

    def spectrogram(self, x: Tensor) -> List[Tensor]:
        x = x - x.mean(dim=-1, keepdim=True)
        x = 0.8 * x / (x.abs().max(dim=-1, keepdim=True)[0] + 1e-9)
        s = torch.view_as_real(self.spec_fn(x))  # (b, f, t, 2)
        s = s.permute(0, 3, 2, 1)  # (b, 2, t, f)
        return [s[..., lo:hi] for lo, hi in self.bands]

    def forward(self, x: Tensor):
        fmap, outs = [], []
        for band, stack in zip(self.spectrogram(x), self.band_convs):
            for i, conv in enumerate(stack):
                band = F.leaky_relu(conv(band), 0.1)
                if i > 0:
                    fmap.append(band)
            outs.append(band)
        x = self.conv_post(torch.cat(outs, dim=-1))
        fmap.append(x)
        return x, fmap


class _MultiD(nn.Module):
    def forward(self, y: Tensor, y_hat: Tensor):
        sr, sg, fr, fg = [], [], [], []
        for d in self.discriminators:
            a, b = d(y)
            c, e = d(y_hat)
            sr.append(a), fr.append(b), sg.append(c), fg.append(e)
        return sr, sg, fr, fg


class MultiPeriodDiscriminator(_MultiD):
    """discriminators.py:18-49."""

    def __init__(self, periods: Sequence[int] = (2, 3, 5, 7, 11)):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p) for p in periods])


class MultiResolutionDiscriminator(_MultiD):
    """discriminators.py:110-141."""

    def __init__(self, fft_sizes: Sequence[int] = (2048, 1024, 512)):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorR(w) for w in fft_sizes])


# ----------------------------------------------------------------------------
# GAN wrapper + losses: reference flow2gan/models/gan.py
# ----------------------------------------------------------------------------
def hinge_d_loss(score_real, score_fake):  # gan.py:57-66
    loss = 0
    for r, f in zip(score_real, score_fake):
        loss = loss + torch.mean(torch.clamp(1 - r, min=0)) + torch.mean(torch.clamp(1 + f, min=0))
    return loss


def hinge_g_loss(score_fake):  # gan.py:68-75
    loss = 0
    for f in score_fake:
        loss = loss + torch.mean(torch.clamp(1 - f, min=0))
    return loss


def feature_matching(fmap_real, fmap_fake):  # gan.py:77-87
    loss = 0
    for fr, ff in zip(fmap_real, fmap_fake):
        for r, f in zip(fr, ff):
            loss = loss + F.l1_loss(r.detach(), f)
    return loss


class GAN(nn.Module):
    """gan.py:30-166 (A18, A21, A22).  `noise` may be injected for parity tests."""

    def __init__(self, generator: MelAudioGenerator,
                 mel_recon_n_ffts=(32, 64, 128, 256, 512, 1024, 2048),
                 mel_recon_n_mels=(5, 10, 20, 40, 80, 160, 320)):
        super().__init__()
        self.generator = generator
        self.discriminator = nn.ModuleList([MultiPeriodDiscriminator(),
                                            MultiResolutionDiscriminator()])
        self.mel_recon_modules = nn.ModuleList([
            OracleMelSpectrogram(generator.sampling_rate, n, n // 4, m, power=1)
            for n, m in zip(mel_recon_n_ffts, mel_recon_n_mels)])

    def mel_recon_loss(self, real, fake):  # gan.py:89-99
        loss = 0
        for m in self.mel_recon_modules:
            loss = loss + F.l1_loss(clipped_log(m(real)), clipped_log(m(fake)))
        return loss

    def forward(self, cond, audio, audio_lens=None, n_timesteps=1, train_disc=True, noise=None):
        mp, mr = self.discriminator
        if train_disc:
            self.discriminator.train()
            self.generator.eval()
            with torch.no_grad():
                fake = self.generator.infer(cond, audio_lens, n_timesteps, False, noise=noise)
            sr, sg, _, _ = mp(audio, fake)
            sr2, sg2, _, _ = mr(audio, fake)
            return hinge_d_loss(sr, sg), hinge_d_loss(sr2, sg2)
        self.discriminator.eval()
        self.generator.train()
        fake = self.generator.infer(cond, audio_lens, n_timesteps, False, noise=noise)
        _, sg, fr, fg = mp(audio, fake)
        _, sg2, fr2, fg2 = mr(audio, fake)
        return (hinge_g_loss(sg), hinge_g_loss(sg2), feature_matching(fr, fg),
                feature_matching(fr2, fg2), self.mel_recon_loss(audio, fake))


# ----------------------------------------------------------------------------
# configs (values only; reference flow2gan/models/config.py:31-115)
# ----------------------------------------------------------------------------
GENERATOR_CONFIGS = {
    "mel_24k_base": dict(
        sampling_rate=24000, n_mels=100, mel_n_fft=1024, mel_hop_length=256,
        n_ffts=(512, 256, 128), hop_lengths=(256, 128, 64), channels=(768, 512, 384),
        loss_n_fft=1024, loss_hop_length=256),
    "mel_44k_128band_512x_base": dict(
        sampling_rate=44100, n_mels=128, mel_n_fft=2048, mel_hop_length=512,
        n_ffts=(1024, 512, 256), hop_lengths=(512, 256, 128), channels=(768, 512, 384),
        loss_n_fft=2048, loss_hop_length=512),
}


def build_generator(name: str = "mel_24k_base", **overrides) -> MelAudioGenerator:
    cfg = dict(GENERATOR_CONFIGS[name])
    cfg.update(overrides)
    return MelAudioGenerator(**cfg)
