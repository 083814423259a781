"""On-device data front end (flow2gan_amd/frontend.py) against the CPU restatement of the
reference's dataset code, and the restated resampler against analytic band-limited interpolation."""
import math

import numpy as np
import pytest
import torch

DEV = "cuda"


def _recordings(sr, n=6, seed=0, two_channel=(1, 4), silent=(2,)):
    rng = np.random.RandomState(seed)
    recs = []
    for i in range(n):
        L = int(sr * (0.35 + 0.1 * i))
        t = np.arange(L) / sr
        y = (0.2 + 0.1 * i) * np.sin(2 * np.pi * (180 + 70 * i) * t) + 0.02 * rng.randn(L)
        if i in silent:
            y = 1e-4 * rng.randn(L)
        if i in two_channel:
            y = np.stack([y, 0.5 * y + 0.01 * rng.randn(L)])
        recs.append((y.astype(np.float32), sr))
    return recs


def test_restated_resampler_interpolates_band_limited_signals():
    """The published torchaudio algorithm, restated (no torchaudio in this image): a tone far below
    both Nyquist rates must come out as the same tone sampled at the new rate."""
    import frontend_oracle as FO
    for orig, new in ((44100, 24000), (16000, 24000), (48000, 24000)):
        L = orig // 4
        f0 = 1000.0
        x = torch.sin(2 * math.pi * f0 * torch.arange(L, dtype=torch.float64) / orig).float()[None]
        y = FO.resample(x, orig, new)
        assert y.shape[1] == math.ceil(new * L / orig)
        want = torch.sin(2 * math.pi * f0 * torch.arange(y.shape[1], dtype=torch.float64) / new).float()
        mid = slice(200, y.shape[1] - 200)                      # away from the zero-padded edges
        assert float((y[0, mid] - want[mid]).abs().max()) < 2e-3, (orig, new)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,train", [(24000, False), (24000, True), (44100, True), (16000, False)])
def test_batch_front_end_matches_dataset_restatement(sr, train):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import frontend_oracle as FO
    from flow2gan_amd.frontend import BatchFrontEnd
    recs = _recordings(sr)
    kw = dict(sampling_rate=24000, duration=0.3, train=train, apply_effects=True, max_load_times=2,
              min_rms=0.005)
    fe = BatchFrontEnd(device=DEV, **kw)
    got_a, got_l, got_keep = fe(recs, rng=np.random.RandomState(5))
    # the reference draws per item: crop offsets (incl. retries) then the gain; the batch version
    # draws all crop offsets of an attempt first, then all gains -> replay that order for the oracle
    rng = np.random.RandomState(5)

    class Replay:
        def __init__(self, vals):
            self.vals = list(vals)

        def uniform(self, a, b):
            return self.vals.pop(0)

    n = len(recs)
    offs = [[] for _ in range(n)]
    silent = [True] * n
    if train:
        for _ in range(kw["max_load_times"]):
            for i in range(n):
                if silent[i]:
                    L = recs[i][0].shape[-1]
                    dur = min(kw["duration"], L / sr)
                    o = rng.uniform(0, L / sr - dur)
                    offs[i].append(o)
                    st = int(round(o * sr))
                    seg = np.atleast_2d(recs[i][0])[:, st:st + int(round(dur * sr))]
                    silent[i] = bool(np.sqrt(np.mean(seg.astype(np.float64) ** 2)) < kw["min_rms"])
    gains = [rng.uniform(-1, -6) if train else -3.0 for _ in range(n)]
    items = []
    for i, (y, _) in enumerate(recs):
        items.append(FO.prepare_item(y, sr, 24000, kw["duration"], train, True, kw["max_load_times"],
                                     kw["min_rms"], Replay(offs[i] + [gains[i]])))
    want_a, want_l, want_keep = FO.collate(items)
    assert got_keep == want_keep and 2 not in got_keep          # the silent item is dropped
    assert got_l.cpu().tolist() == want_l.tolist()
    assert got_a.shape == want_a.shape
    assert float((got_a.cpu() - want_a).abs().max()) < 2e-5


@pytest.mark.gpu
def test_wave_stats_and_gain_kernels():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from flow2gan_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 2, 1000, generator=g) * 0.1
    lens = torch.tensor([1000, 640, 1], dtype=torch.int32)
    xd = x.to(DEV)
    stats = ops.empty(3, 2, device=DEV)
    ops.call("f2g_wave_stats", ops.ptr(xd), 2000, 1000, 3, 2, ops.ptr(lens.to(DEV)), ops.ptr(stats))
    out = ops.empty(3, 1000, device=DEV)
    tp = torch.tensor([0.5, 0.0, 0.25], device=DEV)
    ops.call("f2g_wave_gain", ops.ptr(out), 1000, ops.ptr(xd), 2000, 1000, 3, 2, 1000,
             ops.ptr(lens.to(DEV)), ops.ptr(stats), ops.ptr(tp))
    for b in range(3):
        n = int(lens[b])
        seg = x[b, :, :n]
        mono = seg.mean(0)
        assert abs(float(stats[b, 0]) - float(seg.pow(2).mean().sqrt())) < 1e-6
        assert abs(float(stats[b, 1]) - float(mono.abs().max())) < 1e-7
        want = torch.zeros(1000)
        want[:n] = mono * (float(tp[b]) / float(mono.abs().max())) if float(tp[b]) > 0 else mono
        assert float((out[b].cpu() - want).abs().max()) < 1e-6
