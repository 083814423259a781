"""Discriminators of the GAN stage with the reference's names, init order and state-dict keys
(reference flow2gan/models/discriminators.py).  Weight-norm is disabled in the reference
(discriminators.py:13-15), so plain Conv2d parameters are what a checkpoint holds.

Training runs through the fused loss nodes in flow2gan_amd/fused_disc.py; `forward(y, y_hat)`
keeps the reference's return convention (scores / feature maps in (B, C, H, W)) for inspection
and parity tests, computed on the same HIP kernels without autograd.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn
from torch.nn import Conv2d

from .. import fused_disc as FD
from .. import ops
from .modules import _Window


class DiscriminatorP(nn.Module):
    """reference discriminators.py:52-107."""

    def __init__(self, period: int, in_channels: int = 1, kernel_size: int = 5, stride: int = 3,
                 lrelu_slope: float = 0.1, num_embeddings: Optional[int] = None):
        super().__init__()
        assert (in_channels, kernel_size, stride, lrelu_slope) == (1, 5, 3, 0.1)
        _no_embeddings(num_embeddings)
        self.period = period
        pad = (kernel_size // 2, 0)
        self.convs = nn.ModuleList([
            Conv2d(in_channels, 32, (kernel_size, 1), (stride, 1), padding=pad),
            Conv2d(32, 128, (kernel_size, 1), (stride, 1), padding=pad),
            Conv2d(128, 512, (kernel_size, 1), (stride, 1), padding=pad),
            Conv2d(512, 1024, (kernel_size, 1), (stride, 1), padding=pad),
            Conv2d(1024, 1024, (kernel_size, 1), (1, 1), padding=pad),
        ])
        self.conv_post = Conv2d(1024, 1, (3, 1), 1, padding=(1, 0))
        self.lrelu_slope = lrelu_slope

    def _params(self):
        p = []
        for c in self.convs:
            p += [c.weight, c.bias]
        return p + [self.conv_post.weight, self.conv_post.bias]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, cond_embedding_id: Optional[torch.Tensor] = None):
        """x (B, T) -> (score (B, H*p), fmap list of (B, C, H, p))."""
        _no_embeddings(cond_embedding_id)
        B = x.shape[0]
        st = FD._mpd_forward_one(x.contiguous(), self.period, self._params())
        p = self.period
        fmap = []
        for l in range(2, 6):
            y = st["acts"][l]
            H, Cc = st["hs"][l], y.shape[1]
            fmap.append(FD.unhalo(y, B * p, H).reshape(B, p, H, Cc).permute(0, 3, 2, 1))
        H5 = st["hs"][5]
        sc = st["scores"].view(B, p, H5, 1).permute(0, 3, 2, 1)
        fmap.append(sc)
        return torch.flatten(sc, 1, -1), fmap


def _no_embeddings(v) -> None:
    """The conditional variant (an embedding table indexed by `bandwidth_id`,
    discriminators.py:73-75,97-99,180-181,210-212) is never built by GAN (gan.py:40-42); the
    arguments are accepted with the reference's default and refused otherwise."""
    if v is not None:
        raise NotImplementedError("flow2gan_amd builds the non-conditional discriminators only "
                                  "(num_embeddings=None, bandwidth_id=None), as GAN does")


class _MultiD(nn.Module):
    def forward(self, y: torch.Tensor, y_hat: torch.Tensor,
                bandwidth_id: Optional[torch.Tensor] = None):
        _no_embeddings(bandwidth_id)
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for d in self.discriminators:
            y_d_r, fmap_r = d(y)
            y_d_g, fmap_g = d(y_hat)
            y_d_rs.append(y_d_r)
            fmap_rs.append(fmap_r)
            y_d_gs.append(y_d_g)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


class MultiPeriodDiscriminator(_MultiD):
    """reference discriminators.py:18-49."""

    def __init__(self, periods: Tuple[int, ...] = (2, 3, 5, 7, 11),
                 num_embeddings: Optional[int] = None):
        super().__init__()
        _no_embeddings(num_embeddings)
        self.periods = tuple(periods)
        self.discriminators = nn.ModuleList([DiscriminatorP(period=p) for p in periods])


class DiscriminatorR(nn.Module):
    """reference discriminators.py:143-219."""

    def __init__(self, window_length: int, num_embeddings: Optional[int] = None,
                 channels: int = 32, hop_factor: float = 0.25, bands=FD.MRD_BANDS):
        super().__init__()
        _no_embeddings(num_embeddings)
        assert channels == FD.MRD_CH and hop_factor == 0.25 and tuple(bands) == FD.MRD_BANDS
        self.window_length = window_length
        self.hop_factor = hop_factor
        self.spec_fn = _Window(window_length)  # torchaudio Spectrogram's persistent buffer
        n_fft = window_length // 2 + 1
        self.bands = [(int(b[0] * n_fft), int(b[1] * n_fft)) for b in bands]

        def convs():
            return nn.ModuleList([
                nn.Conv2d(2, channels, (3, 9), (1, 1), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 9), (1, 2), padding=(1, 4)),
                nn.Conv2d(channels, channels, (3, 3), (1, 1), padding=(1, 1)),
            ])

        self.band_convs = nn.ModuleList([convs() for _ in range(len(self.bands))])
        self.conv_post = nn.Conv2d(channels, 1, (3, 3), (1, 1), padding=(1, 1))

    def _params(self):
        p = []
        for stack in self.band_convs:
            for c in stack:
                p += [c.weight, c.bias]
        return p + [self.conv_post.weight, self.conv_post.bias]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, cond_embedding_id: Optional[torch.Tensor] = None):
        """x (B, T) -> (score (B, 1, frames, freq), fmap list of (B, C, frames, freq))."""
        _no_embeddings(cond_embedding_id)
        B = x.shape[0]
        st = FD._mrd_forward_one(x.contiguous(), self.window_length, self._params())
        Ft, Wcat, C = st["Ft"], st["Wcat"], FD.MRD_CH
        cat = st["cat"].view(B, Ft, Wcat, C)
        fmap = []
        foff = 0
        for bi in range(5):
            ws = st["widths"][bi]
            for l in range(1, 4):
                y = st["acts"][bi][l]
                fmap.append(y.view(B, Ft, ws[l + 1], C).permute(0, 3, 1, 2))
            fmap.append(cat[:, :, foff:foff + ws[5]].permute(0, 3, 1, 2))
            foff += ws[5]
        sc = st["scores"].view(B, Ft, Wcat, 1).permute(0, 3, 1, 2)
        fmap.append(sc)
        return sc, fmap


class MultiResolutionDiscriminator(_MultiD):
    """reference discriminators.py:110-141."""

    def __init__(self, fft_sizes: Tuple[int, ...] = (2048, 1024, 512),
                 num_embeddings: Optional[int] = None):
        super().__init__()
        _no_embeddings(num_embeddings)
        self.fft_sizes = tuple(fft_sizes)
        self.discriminators = nn.ModuleList([DiscriminatorR(window_length=w) for w in fft_sizes])
