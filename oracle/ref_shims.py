"""Import the real reference (`/root/reference/flow2gan`) in the BUILD CONTAINER only.

TEST INFRASTRUCTURE.  The reference needs three third-party packages that are
not installed here and cannot be (no network): torchaudio, lhotse, tensorboard.
`install()` registers `sys.modules` stand-ins for them so that
`flow2gan.models.*` import and run on CPU.  The torchaudio stand-in is backed by
the restatement in `flow2gan_oracle.py` (pinned by the reference's wav<->mel
fixtures); lhotse/tensorboard stand-ins are inert (they are not on the hot
path).  Nothing here is shipped to, or usable on, the GPU box.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "flow2gan"))


def install():
    """Register stand-in modules and put the reference on sys.path."""
    if "flow2gan" in sys.modules and getattr(sys.modules["flow2gan"], "__file__", "").startswith(REFERENCE_ROOT):
        return
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import torch
    import flow2gan_oracle as O

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    # ---- torchaudio -------------------------------------------------------
    ta = mod("torchaudio")
    ta.__path__ = []
    taf = mod("torchaudio.functional")
    tat = mod("torchaudio.transforms")
    tas = mod("torchaudio.sox_effects")
    ta.functional, ta.transforms, ta.sox_effects = taf, tat, tas

    def linear_fbanks(n_freqs, f_min, f_max, n_filter, sample_rate):
        return O.taudio_linear_fbanks(n_freqs, f_min, f_max, n_filter, sample_rate)

    def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
        assert norm is None and mel_scale == "htk"
        return O.taudio_melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate)

    taf.linear_fbanks = linear_fbanks
    taf.melscale_fbanks = melscale_fbanks

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0,
                     window_fn=torch.hann_window, power=2.0, normalized=False, wkwargs=None,
                     center=True, pad_mode="reflect", onesided=True):
            super().__init__()
            win_length = win_length if win_length is not None else n_fft
            assert win_length == n_fft and pad == 0 and not normalized and center
            assert pad_mode == "reflect" and onesided and window_fn is torch.hann_window
            self.n_fft = n_fft
            self.hop_length = hop_length if hop_length is not None else win_length // 2
            self.power = power
            self.register_buffer("window", torch.hann_window(n_fft))

        def forward(self, x):
            return O.taudio_spectrogram(x, self.n_fft, self.hop_length, self.window, self.power)

    class MelScale(torch.nn.Module):
        def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
            super().__init__()
            self.register_buffer("fb", melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate))

        def forward(self, spec):
            return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None,
                     f_min=0.0, f_max=None, pad=0, n_mels=128, window_fn=torch.hann_window,
                     power=2.0, normalized=False, wkwargs=None, center=True,
                     pad_mode="reflect", onesided=None, norm=None, mel_scale="htk"):
            super().__init__()
            f_max = float(sample_rate // 2) if f_max is None else f_max
            self.spectrogram = Spectrogram(n_fft=n_fft, win_length=win_length,
                                           hop_length=hop_length, power=power, center=center)
            self.mel_scale = MelScale(n_mels, sample_rate, f_min, f_max, n_fft // 2 + 1)

        def forward(self, x):
            return self.mel_scale(self.spectrogram(x))

    tat.Spectrogram, tat.MelSpectrogram, tat.MelScale = Spectrogram, MelSpectrogram, MelScale

    # ---- lhotse / tensorboard (inert) --------------------------------------
    lh = mod("lhotse")
    lh.__path__ = []
    lh.RecordingSet = object
    lhu = mod("lhotse.utils")

    def fix_random_seed(seed):
        import random
        random.seed(seed)
        torch.manual_seed(seed)

    lhu.fix_random_seed = fix_random_seed
    lh.utils = lhu
    lhd = mod("lhotse.dataset")
    lhd.__path__ = []
    lhds = mod("lhotse.dataset.sampling")
    lhds.__path__ = []
    lhdb = mod("lhotse.dataset.sampling.base")
    lhdb.CutSampler = object
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = mod("torch.utils.tensorboard")
        tb.SummaryWriter = object
        t0 = mod("tensorboard")
        t0.__path__ = []
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
