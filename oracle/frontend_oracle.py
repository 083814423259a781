"""CPU restatement of the reference's per-item data preparation and collation
(flow2gan/dataset.py:31-45 pad_seq_collate_fn, :122-175 AudioDataset.__getitem__) on already
decoded arrays, plus torchaudio.functional.resample's published algorithm (sinc_interp_hann).

TEST INFRASTRUCTURE ONLY.  *Parity unpinned* for the two third-party pieces: torchaudio
(resample) and sox (`norm`) are not installed in this image and the reference keeps no fixture for
them; `norm <dB>` is restated as "scale so that the peak equals 10^(dB/20)", resample is checked
against the analytic band-limited interpolation in tests/test_hip_frontend.py.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    return (kernels * window * scale).float(), width, orig, new


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """(B, L) -> (B, ceil(new * L / orig))."""
    if orig_freq == new_freq:
        return wave
    k, width, orig, new = resample_kernel(orig_freq, new_freq)
    B, L = wave.shape
    x = torch.nn.functional.pad(wave[:, None], (width, width + orig))
    y = torch.nn.functional.conv1d(x, k, stride=orig)          # (B, new, frames)
    y = y.transpose(1, 2).reshape(B, -1)
    return y[:, :math.ceil(new * L / orig)]


def prepare_item(y: np.ndarray, sr: int, sampling_rate: int, duration: Optional[float], train: bool,
                 apply_effects: bool, max_load_times: int, min_rms: float, rng):
    """dataset.py:122-175 for one decoded recording; returns (waveform 1-D tensor, silence)."""
    y = np.asarray(y, dtype=np.float32)
    if y.ndim == 1:
        y = y[None]
    n = y.shape[1]

    def is_silence(x):
        return bool(np.sqrt(np.mean(x.astype(np.float64) ** 2)) < min_rms)

    if duration is None:
        seg = y
        silence = is_silence(seg)
    else:
        dur = min(duration, n / sr)
        count = int(round(dur * sr))
        if not train:
            seg = y[:, :count]
            silence = is_silence(seg)
        else:
            times = 0
            while times < max_load_times:
                times += 1
                offset = rng.uniform(0, n / sr - dur)
                start = int(round(offset * sr))
                seg = y[:, start:start + count]
                silence = is_silence(seg)
                if not silence:
                    break
    w = torch.from_numpy(seg.mean(axis=0, keepdims=True).astype(np.float32))
    if apply_effects:
        gain = rng.uniform(-1, -6) if train else -3.0
        gain = float(f"{gain:.2f}")
        peak = float(w.abs().max())
        if peak > 0:
            w = w * (10.0 ** (gain / 20.0) / peak)
    if sr != sampling_rate:
        w = resample(w, sr, sampling_rate)
    return w[0], silence


def collate(items: List[Tuple[torch.Tensor, bool]], filter_silence: bool = True):
    """dataset.py:31-45; returns (audios, audio_lens, kept indices)."""
    keep = list(range(len(items)))
    if filter_silence:
        keep = [i for i, (_, s) in enumerate(items) if not s]
        if not keep:
            keep = [0]
    audios = torch.nn.utils.rnn.pad_sequence([items[i][0] for i in keep], batch_first=True)
    lens = torch.tensor([len(items[i][0]) for i in keep], dtype=torch.int32)
    return audios, lens, keep
