"""Primitive modules of the Flow2GAN generator -- parameter containers with the reference's
names, shapes, init and state-dict keys (reference flow2gan/models/modules.py).

The fused training / inference path (flow2gan_amd/fused.py) reads these parameters directly
and runs hand-written HIP kernels in a channels-last layout; every leaf `forward` below is the
drop-in (batch, channels, time) entry point of its reference counterpart -- same arguments, same
return values, autograd included -- and dispatches onto the same kernels (flow2gan_amd/leaf.py).
Nothing here falls back to ATen compute.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops


# ---------------------------------------------------------------------------------- DFT tables
_DFT_CACHE = {}


def dft_matrices(n_fft: int, device) -> tuple:
    """(Wd, Wi): analysis matrix (n_fft+2, n_fft) with the hann window folded in, rows
    [Re(0..N/2) | Im(0..N/2)], and synthesis matrix (n_fft, n_fft+2) with window,
    1/N and the one-sided weights folded in; Im(DC)/Im(Nyquist) columns are exactly 0 (A.2)."""
    key = (n_fft, str(device))
    if key not in _DFT_CACHE:
        N = n_fft
        w = torch.hann_window(N, dtype=torch.float32).double()
        n = torch.arange(N, dtype=torch.int64)
        k = torch.arange(N // 2 + 1, dtype=torch.int64)
        ang = 2.0 * math.pi * ((k[:, None] * n[None, :]) % N).double() / N
        cos, sin = torch.cos(ang), torch.sin(ang)
        wd = torch.cat([cos * w[None, :], -sin * w[None, :]], dim=0)  # (N+2, N)
        ck = torch.full((N // 2 + 1,), 2.0, dtype=torch.float64)
        ck[0] = ck[-1] = 1.0
        wr = (cos * ck[:, None] / N) * w[None, :]
        wi_ = (-sin * 2.0 / N) * w[None, :]
        wi_[0] = 0.0
        wi_[-1] = 0.0
        wi = torch.cat([wr, wi_], dim=0).t().contiguous()  # (N, N+2)
        _DFT_CACHE[key] = (wd.float().contiguous().to(device), wi.float().to(device))
        for t in _DFT_CACHE[key]:
            t._f2g_const = True   # never written again: operand images of it may be cached (ops.derived)
    return _DFT_CACHE[key]


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated (gan.py:47-54,
    modules.py:131-138 depend on it).  Host-side table construction only."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    return _triangular(all_freqs, f_pts)


def linear_fbanks(n_freqs: int, f_min: float, f_max: float, n_filter: int, sample_rate: int):
    """torchaudio.functional.linear_fbanks restated (modules.py:194-200)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    f_pts = torch.linspace(f_min, f_max, n_filter + 2)
    return _triangular(all_freqs, f_pts)


def _triangular(all_freqs, f_pts):
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


# ---------------------------------------------------------------------------------- containers
class STFT(nn.Module):
    """reference modules.py:52-84 (buffer `window` kept for checkpoint compatibility)."""

    def __init__(self, n_fft: int, hop_length: int):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.win_length, self.onesided = n_fft, True
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, audio: Tensor, audio_lens: Optional[Tensor] = None):
        """(B, T) -> (complex (B, n_fft/2 + 1, 1 + T // hop), frame counts or None) -- torch.stft with
        center / reflect padding and the periodic hann window, on the LDS-butterfly FFT kernel."""
        from ..leaf import stft
        spec = stft(audio, self.n_fft, self.hop_length)
        if audio_lens is None:
            return spec, None
        spec_lens = 1 + torch.div(audio_lens, self.hop_length, rounding_mode="floor")
        assert spec.shape[2] == int(spec_lens.max())
        return spec, spec_lens


class ISTFT(nn.Module):
    """reference modules.py:87-116."""

    def __init__(self, n_fft: int, hop_length: int):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.win_length, self.onesided, self.return_complex = n_fft, True, False
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, spec: Tensor) -> Tensor:
        """complex (B, n_fft/2 + 1, F) -> (B, hop * (F - 1)): torch.istft(center=True)."""
        from ..leaf import istft
        return istft(spec, self.n_fft, self.hop_length, self.window)


class SinusoidalPosEmb(nn.Module):
    """reference modules.py:217-232: t (B,) -> (B, dim) = [sin | cos](scale * t * f_k)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        assert self.dim % 2 == 0, "SinusoidalPosEmb requires dim to be even"

    @torch.no_grad()
    def forward(self, x: Tensor, scale=1000) -> Tensor:
        if x.ndim < 1:
            x = x.unsqueeze(0)
        t = x.contiguous().float()
        return ops.time_embedding(ops.empty(t.shape[0], self.dim, device=t.device), t, self.dim, float(scale))


class ChannelScale(nn.Module):
    """reference modules.py:273-283."""

    def __init__(self, channels: int, scale: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels, 1), scale))

    def forward(self, x: Tensor) -> Tensor:
        """x (B, C, T) * scale, the scale's gradient through LimitParamValue [0.5, 1.0] with
        probability 0.6 while training (modules.py:259-283; one Python-RNG draw per call)."""
        from ..leaf import ChannelScaleFn
        from ..fused import _limit_draw
        return ChannelScaleFn.apply(x, self.scale, _limit_draw(self.training))


class BiasNorm(nn.Module):
    """reference modules.py:342-416; forward (B, C, T) -> (B, C, T) on the HIP kernel."""

    def __init__(self, num_channels: int, channel_dim: int = 1, log_scale: float = 1.0,
                 log_scale_min: float = -1.5, log_scale_max: float = 1.5):
        super().__init__()
        assert channel_dim == 1
        self.num_channels = num_channels
        self.log_scale = nn.Parameter(torch.tensor(log_scale))
        self.bias = nn.Parameter(torch.empty(num_channels).normal_(mean=0, std=1e-2))
        self.log_scale_min, self.log_scale_max = log_scale_min, log_scale_max

    def forward(self, x: Tensor) -> Tensor:
        from ..leaf import BiasNormFn
        from ..fused import _limit_draw
        assert x.shape[1] == self.num_channels
        return BiasNormFn.apply(x, self.bias, self.log_scale, _limit_draw(self.training))


class ConvNeXtBlock(nn.Module):
    """reference modules.py:419-495."""

    def __init__(self, channels: int = 512, hidden_channels: int = 1536,
                 conv_kernel_size: int = 7, cond_channels: Optional[int] = None,
                 time_embed_channels: Optional[int] = None,
                 residual_scale: Optional[float] = 1.0):
        super().__init__()
        assert conv_kernel_size % 2 == 1 and conv_kernel_size <= 7, conv_kernel_size
        assert residual_scale is not None
        self.channels, self.hidden_channels = channels, hidden_channels
        self.kernel_size = conv_kernel_size
        self.dwconv = nn.Conv1d(channels, channels, kernel_size=conv_kernel_size,
                                padding=conv_kernel_size // 2, groups=channels)
        self.norm = BiasNorm(channels, channel_dim=1)
        self.pwconv1 = nn.Conv1d(channels, hidden_channels, kernel_size=1)
        self.act = nn.PReLU(hidden_channels)
        self.pwconv2 = nn.Conv1d(hidden_channels, channels, kernel_size=1)
        if cond_channels is not None:
            self.cond_proj = nn.Conv1d(cond_channels, channels, kernel_size=1)
        if time_embed_channels is not None:
            self.time_embed_proj = nn.Linear(time_embed_channels, channels)
        self.residual_scale = ChannelScale(channels)

    def forward(self, x: Tensor, cond: Optional[Tensor] = None, time_embed: Optional[Tensor] = None,
                mask: Optional[Tensor] = None) -> Tensor:
        """x (B, C, T), cond (B, cond_channels, T), time_embed (B, time_embed_channels), mask
        (B, 1, T) -> (B, C, T)   (modules.py:455-495)."""
        from ..leaf import block_forward
        return block_forward(self, x, cond, time_embed, mask)


class CondEncoder(nn.Module):
    """reference modules.py:498-542."""

    def __init__(self, cond_dim: int = 100, channels: int = 512, hidden_factor: int = 3,
                 conv_kernel_size: int = 7, num_layers: int = 4,
                 residual_scale: Optional[float] = 1.0):
        super().__init__()
        self.cond_dim, self.channels = cond_dim, channels
        self.in_proj = nn.Conv1d(cond_dim, channels, kernel_size=3, padding=1)
        self.in_norm = BiasNorm(channels, channel_dim=1)
        self.blocks = nn.ModuleList([
            ConvNeXtBlock(channels=channels, hidden_channels=int(channels * hidden_factor),
                          conv_kernel_size=conv_kernel_size, residual_scale=residual_scale)
            for _ in range(num_layers)])

    def forward(self, x: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        """x (B, n_mels, F), mask (B, 1, F) -> (B, channels, F)   (modules.py:524-542)."""
        from ..leaf import cond_encoder_forward
        return cond_encoder_forward(self, x, mask)


class ConvNeXtDecoder(nn.Module):
    """reference modules.py:545-627."""

    def __init__(self, in_channels: int, out_channels: int, channels: int = 512,
                 cond_channels: int = 512, time_embed_channels: int = 512, hidden_factor: int = 3,
                 conv_kernel_size: int = 7, num_layers: int = 8,
                 residual_scale: Optional[float] = 1.0):
        super().__init__()
        self.in_channels, self.out_channels, self.channels = in_channels, out_channels, channels
        self.cond_channels, self.time_embed_channels = cond_channels, time_embed_channels
        self.in_proj = nn.Conv1d(in_channels, channels, kernel_size=1)
        self.in_norm = BiasNorm(channels, channel_dim=1)
        self.time_embed = SinusoidalPosEmb(time_embed_channels)      # (parameter-free: no state-dict key)
        th = int(time_embed_channels * hidden_factor)
        self.time_mlp = nn.Sequential(nn.Linear(time_embed_channels, th), nn.SiLU(),
                                      nn.Linear(th, time_embed_channels))
        ch = int(cond_channels * hidden_factor)
        self.cond_mlp = nn.Sequential(nn.Conv1d(cond_channels, ch, kernel_size=1), nn.PReLU(ch),
                                      nn.Conv1d(ch, cond_channels, kernel_size=1))
        self.blocks = nn.ModuleList([
            ConvNeXtBlock(channels=channels, hidden_channels=int(channels * hidden_factor),
                          conv_kernel_size=conv_kernel_size, cond_channels=cond_channels,
                          time_embed_channels=time_embed_channels, residual_scale=residual_scale)
            for _ in range(num_layers)])
        self.out_proj = nn.Conv1d(channels, out_channels, kernel_size=1)

    def forward(self, x: Tensor, cond: Tensor, t: Optional[Tensor] = None,
                mask: Optional[Tensor] = None) -> Tensor:
        """x (B, in_channels, F), cond (B, cond_channels, F), t (B,), mask (B, 1, F) ->
        (B, out_channels, F)   (modules.py:590-627)."""
        from ..leaf import decoder_forward
        return decoder_forward(self, x, cond, t, mask)


class AudioConvNeXt(nn.Module):
    """reference modules.py:630-721."""

    def __init__(self, n_fft: int = 512, hop_length: int = 256, cond_hop_length: int = 256,
                 channels: int = 768, cond_channels: int = 512, time_embed_channels: int = 512,
                 hidden_factor: int = 3, conv_kernel_size: int = 7, num_layers: int = 8,
                 residual_scale: Optional[float] = 1.0):
        super().__init__()
        self.n_fft, self.hop_length, self.channels = n_fft, hop_length, channels
        self.fft = STFT(n_fft=n_fft, hop_length=hop_length)
        self.ifft = ISTFT(n_fft=n_fft, hop_length=hop_length)
        assert cond_hop_length % hop_length == 0, \
            "cond_hop_length should be integer multiple of hop_length."
        self.cond_upsample_factor = cond_hop_length // hop_length
        assert self.cond_upsample_factor in (1, 2, 4), "supported upsample factors: 1, 2, 4"
        self.decoder = ConvNeXtDecoder(
            in_channels=n_fft + 2, out_channels=n_fft + 2, channels=channels,
            cond_channels=cond_channels, time_embed_channels=time_embed_channels,
            hidden_factor=hidden_factor, conv_kernel_size=conv_kernel_size,
            num_layers=num_layers, residual_scale=residual_scale)

    def upsample_cond(self, cond: Tensor, fft_frames: int) -> Tensor:
        """modules.py:668-680: repeat every condition frame `cond_upsample_factor` times, then cut or
        zero-pad to `fft_frames` (the fused path never materialises this: `f2g_dwnorm_fwd` reads row
        f // up).  Data movement only."""
        B, Cc, Fc = cond.shape
        up = self.cond_upsample_factor
        src = cond.contiguous().float()
        rep = ops.empty(B * Cc, Fc * up, device=cond.device)       # rep[bc, f * up + r] = cond[bc, f]
        ops.permute4(rep, src, (B * Cc, Fc, up, 1), (Fc, 1, 0, 0))
        if Fc * up == fft_frames:
            return rep.view(B, Cc, fft_frames)
        out = ops.zeros(B, Cc, fft_frames, device=cond.device)    # convert_length (utils.py:235-244)
        ops.copy3(out, fft_frames, 0, rep, Fc * up, 0, B * Cc, 1, min(Fc * up, fft_frames))
        return out

    def forward(self, audio: Tensor, cond: Tensor, t: Optional[Tensor] = None,
                audio_lens: Optional[Tensor] = None) -> Tensor:
        """audio (B, T), cond (B, cond_channels, cond_frames), t (B,), audio_lens (B,) -> (B, T)
        (modules.py:682-721): one Fourier branch, the same coarse node a model evaluation runs."""
        from ..leaf import audio_convnext_forward
        return audio_convnext_forward(self, audio, cond, t, audio_lens)


class LinearFilterSpectrogram(nn.Module):
    """reference modules.py:146-214: power-2 spectrogram -> linear triangular filterbank."""

    def __init__(self, sample_rate: int, n_filter: int, n_fft: int, hop_length: int,
                 center: bool = True, power: float = 2.0):
        super().__init__()
        assert center and power == 2
        self.sample_rate, self.n_filter = sample_rate, n_filter
        self.n_fft, self.hop_length, self.power = n_fft, hop_length, 2
        self.spectrogram = _Window(n_fft)
        self.register_buffer("fb", linear_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2),
                                                 n_filter, sample_rate))

    def forward(self, waveform: Tensor) -> Tensor:
        """(..., T) -> (..., n_filter, 1 + T // hop)   (modules.py:202-214)."""
        from ..leaf import filterbank_spectrogram
        return filterbank_spectrogram(waveform, self.n_fft, self.hop_length, self.fb, 2)


class _Window(nn.Module):
    """Holds the `window` buffer torchaudio's Spectrogram registers (checkpoint key parity)."""

    def __init__(self, n_fft: int):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))


class _MelScale(nn.Module):
    def __init__(self, fb: Tensor):
        super().__init__()
        self.register_buffer("fb", fb)


class MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram stand-in (power=1, htk, norm=None) used by the
    mel front-end (modules.py:131-138) and the multi-scale mel loss (gan.py:47-54)."""

    def __init__(self, sample_rate: int, n_fft: int, hop_length: int, n_mels: int,
                 power: float = 1):
        super().__init__()
        assert power == 1
        self.n_fft, self.hop_length, self.n_mels = n_fft, hop_length, n_mels
        self.spectrogram = _Window(n_fft)
        self.mel_scale = _MelScale(melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2),
                                                   n_mels, sample_rate))

    def forward(self, waveform: Tensor) -> Tensor:
        """(..., T) -> (..., n_mels, 1 + T // hop): |STFT| through the mel filterbank."""
        from ..leaf import filterbank_spectrogram
        return filterbank_spectrogram(waveform, self.n_fft, self.hop_length, self.mel_scale.fb, 1)


class LogMelSpectrogram(nn.Module):
    """reference modules.py:119-143 (A1): (B, T) audio -> (B, n_mels, 1 + T//hop) log-mel.
    STFT as a windowed-DFT GEMM on the matrix cores, |.|, mel filterbank GEMM, log(clip)."""

    def __init__(self, sampling_rate: int = 24000, n_fft: int = 1024, hop_length: int = 256,
                 n_mels: int = 100, center: bool = True, power: float = 1):
        super().__init__()
        assert center and power == 1
        self.mel = MelSpectrogram(sampling_rate, n_fft, hop_length, n_mels, power=1)

    @torch.no_grad()
    def forward(self, waveform: Tensor) -> Tensor:
        from ..fused import log_mel_forward
        return log_mel_forward(self.mel, waveform)
