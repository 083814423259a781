"""On-device data front end (reference flow2gan/dataset.py:31-45,122-175).

The reference prepares every training item on DataLoader worker processes: random crop, silence test
on the crop's RMS, mono mix, sox `norm <gain dB>` (gain ~ U(-1, -6) dB in training, -3 dB
otherwise), torchaudio sinc resampling to the model rate, then `pad_seq_collate_fn` drops silent
items and zero-pads.  Here the host only chooses crop offsets and slices the decoded arrays; the
crops of a whole batch are uploaded once and everything else runs as kernels on the batch:

  f2g_wave_stats   RMS over channels x time (silence flag) + peak of the mono mix
  f2g_wave_gain    mono mix * 10^(gain/20) / peak, zero padded to the batch length
  f2g_gemm         polyphase windowed-sinc resampling as one implicit GEMM: rows = output frames
                   of `orig` input samples, window = 2*width + orig taps, weights = the `new`
                   phase filters, output row = `new` consecutive output samples

The sinc kernel restates torchaudio.functional.resample's published algorithm (sinc_interp_hann,
lowpass_filter_width 6, rolloff 0.99); torchaudio is not installed in this image, so that
restatement is *unpinned* (checked against the direct interpolation formula in the tests).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .ops import gemm, mat, win1d

_KERNELS = {}


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99):
    """(kernel (new, kw) float32, width, orig, new) with orig/new reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base_freq, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    return (k * window * scale).astype(np.float32), width, orig, new


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """(B, L) on the GPU -> (B, ceil(new * L / orig)): conv1d(stride=orig) with `new` phase filters."""
    if orig_freq == new_freq:
        return wave
    key = (orig_freq, new_freq, str(wave.device))
    if key not in _KERNELS:
        k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
        _KERNELS[key] = (torch.from_numpy(k).to(wave.device), width, orig, new)
    k, width, orig, new = _KERNELS[key]
    wave = wave.contiguous()
    B, L = wave.shape
    kw = k.shape[1]                                   # 2 * width + orig taps
    frames = (L + 2 * width + orig - kw) // orig + 1  # conv1d output length on the padded signal
    out = ops.empty(B * frames, ops.pad4(new), device=wave.device)
    gemm(win1d(wave, B, L, 1, frames, orig, width, kw), mat(k), out)
    target = -(-new * L // orig)
    return out[:, :new].reshape(B, frames * new)[:, :target]


class BatchFrontEnd:
    """Batch-level restatement of AudioDataset.__getitem__ + pad_seq_collate_fn."""

    def __init__(self, sampling_rate: int = 24000, duration: Optional[float] = None,
                 train: bool = False, apply_effects: bool = True, max_load_times: int = 1,
                 min_rms: float = 0.005, filter_silence: bool = True, device="cuda"):
        self.sampling_rate = sampling_rate
        self.duration = duration
        self.train = train
        self.apply_effects = apply_effects
        self.max_load_times = max_load_times
        self.min_rms = min_rms
        self.filter_silence = filter_silence
        self.device = torch.device(device)

    def _crop(self, y: np.ndarray, sr: int, rng) -> np.ndarray:
        """dataset.py:134-153: whole file, first segment (validation) or a random segment."""
        if y.ndim == 1:
            y = y[None]
        n = y.shape[1]
        if self.duration is None:
            return y
        dur = min(self.duration, n / sr)
        count = int(round(dur * sr))
        if not self.train:
            return y[:, :count]
        offset = rng.uniform(0, n / sr - dur)
        start = int(round(offset * sr))
        return y[:, start:start + count]

    def _upload(self, crops: List[np.ndarray]):
        B = len(crops)
        C = max(c.shape[0] for c in crops)
        T = max(c.shape[1] for c in crops)
        host = np.zeros((B, C, T), dtype=np.float32)
        for i, c in enumerate(crops):
            host[i, :c.shape[0], :c.shape[1]] = c
            if c.shape[0] < C:                     # fewer channels: replicate so that the mix and
                host[i, c.shape[0]:, :c.shape[1]] = c.mean(0, keepdims=True)  # the RMS stay right
        lens = torch.tensor([c.shape[1] for c in crops], dtype=torch.int32)
        return torch.from_numpy(host).to(self.device), lens.to(self.device), C, T

    def __call__(self, recordings: Sequence[Tuple[np.ndarray, int]], rng=np.random):
        """recordings: [(decoded waveform (channels, samples) or (samples,), sampling rate)].
        Returns (audios (B', T') on the device, audio_lens int32 (B'), kept indices)."""
        srs = {sr for _, sr in recordings}
        assert len(srs) == 1, "one source sampling rate per batch (resampling is a single GEMM)"
        sr = srs.pop()
        n = len(recordings)
        crops = [None] * n
        silent = np.ones(n, dtype=bool)
        tries = self.max_load_times if (self.train and self.duration is not None) else 1
        x = lens = stats = None
        for _ in range(tries):                     # re-draw only the crops that were silent
            todo = [i for i in range(n) if silent[i]]
            if not todo:
                break
            for i in todo:
                crops[i] = self._crop(np.asarray(recordings[i][0], dtype=np.float32), sr, rng)
            x, lens, C, T = self._upload(crops)
            stats = ops.empty(n, 2, device=self.device)
            ops.call("f2g_wave_stats", ops.ptr(x), C * T, T, n, C, ops.ptr(lens), ops.ptr(stats))
            silent = (stats[:, 0] < self.min_rms).cpu().numpy()    # one host sync per attempt
        target = None
        if self.apply_effects:                     # dataset.py:164-168
            gains = [rng.uniform(-1, -6) if self.train else -3.0 for _ in range(n)]
            target = torch.tensor([10.0 ** (float(f"{g:.2f}") / 20.0) for g in gains],
                                  dtype=torch.float32, device=self.device)
        mono = ops.empty(n, T, device=self.device)
        ops.call("f2g_wave_gain", ops.ptr(mono), T, ops.ptr(x), C * T, T, n, C, T, ops.ptr(lens),
                 ops.ptr(stats), ops.ptr(target))
        out_lens = lens
        if sr != self.sampling_rate:               # dataset.py:170-173
            g = math.gcd(sr, self.sampling_rate)
            mono = resample(mono, sr, self.sampling_rate)
            out_lens = (-(-(self.sampling_rate // g) * lens.long() // (sr // g))).to(torch.int32)
            mono = mono.contiguous()                # padding stays zero after the filter tails
            ops.call("f2g_mask_rows", ops.ptr(mono), 1, n, mono.shape[1], 1, ops.ptr(out_lens))
        keep = np.arange(n)
        if self.filter_silence:                    # dataset.py:33-40
            keep = np.nonzero(~silent)[0]
            if len(keep) == 0:
                keep = np.arange(1)
        kt = torch.from_numpy(keep).to(self.device)
        out_lens = out_lens[kt]
        return mono[kt][:, :int(out_lens.max())].contiguous(), out_lens, keep.tolist()
