"""Chunked / streaming synthesis (reference flow2gan/bin/infer_dir.py:126-168).

The mel condition is cut into chunks of `chunk_size` frames, each chunk is synthesised with
`side_context` extra frames on both sides (3 frames of receptive field per depthwise-7 layer x 8
layers = 24, infer_dir.py:145) and the context samples are cropped before concatenation -- same
arithmetic as the reference, so a chunked waveform equals the reference's chunked waveform.

`ChunkRunner` additionally replays a chunk shape from a captured HIP graph: at batch 1 a chunk is
~300 kernel launches of a few microseconds each, so the host launch path, not the GPU, sets the
latency; the graph removes it (one hipGraphLaunch per chunk)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

SIDE_CONTEXT = 3 * 8  # frames; conv_kernel_size 7, 8 layers (infer_dir.py:145)


def chunk_plan(num_frames: int, chunk_size: int, hop: int,
               side_context: int = SIDE_CONTEXT) -> List[Tuple[int, int, int, int]]:
    """[(frame_start, frame_end, left_pad_samples, right_pad_samples)] per chunk
    (infer_dir.py:146-154; the last chunk's right pad may be negative = keep everything)."""
    plan = []
    for i in range((num_frames + chunk_size - 1) // chunk_size):
        fs = max(0, i * chunk_size - side_context)
        fe = min(num_frames, (i + 1) * chunk_size + side_context)
        plan.append((fs, fe, (i * chunk_size - fs) * hop, (fe - (i + 1) * chunk_size) * hop))
    return plan


@torch.no_grad()
def streaming_infer(model, cond: Tensor, n_timesteps: int = 1, chunk_size: int = 100,
                    side_context: int = SIDE_CONTEXT, clamp_pred: bool = True,
                    noise_fn: Optional[Callable[[int, int, int], Tensor]] = None,
                    runner: Optional["ChunkRunner"] = None) -> Tensor:
    """cond (B, n_mels, frames) on the GPU -> audio (B, ~frames*hop).  `noise_fn(chunk_index, B, T)`
    may supply each chunk's initial noise (tests); default is the model's own randn draw."""
    hop = model.mel_hop_length
    outs = []
    for i, (fs, fe, lpad, rpad) in enumerate(chunk_plan(cond.size(2), chunk_size, hop, side_context)):
        c = cond[:, :, fs:fe].contiguous()
        noise = None if noise_fn is None else noise_fn(i, c.size(0), (fe - fs) * hop)
        if runner is not None:
            pred = runner(c, noise)
        else:
            pred = model.infer(cond=c, n_timesteps=n_timesteps, clamp_pred=clamp_pred, noise=noise)
        piece = pred[:, lpad: pred.size(1) - rpad]
        # a runner returns its graph's static output buffer, which the next replay of the same
        # chunk shape (every interior chunk) overwrites: take the samples out now
        outs.append(piece.clone() if runner is not None else piece)
    return torch.cat(outs, dim=-1)


class ChunkRunner:
    """`model.infer` for a fixed (batch, frames) chunk shape, replayed from a HIP graph.

    The first call with a new shape runs eagerly once (warm-up: cached DFT tables, allocator
    pools), then captures the launch sequence on a side stream into `torch.cuda.CUDAGraph`
    (hipGraph on ROCm); later calls copy the condition (and noise) into the static input buffers
    and replay.  The returned tensor is the graph's static output: consume or copy it before the
    next call with the same shape.

    Weights: the forward reads cached re-laid copies of the parameters (ops.derived: padded /
    stacked / transposed / bf16 images), and a captured graph holds raw pointers into them.  Every
    graph entry therefore (a) keeps the copies it captured alive and (b) carries the weights'
    signature (autograd version counters + the raw-pointer writers' epochs, ops.weights_signature);
    when the signature has moved -- an optimizer step, load_state_dict, periodic evaluation during
    training -- the shape is captured again instead of replaying against stale copies."""

    def __init__(self, model, n_timesteps: int = 1, clamp_pred: bool = True):
        self.model = model
        self.n_timesteps = n_timesteps
        self.clamp_pred = clamp_pred
        self.graphs: Dict[Tuple[int, int, int], tuple] = {}
        self.recaptures = 0

    @torch.no_grad()
    def __call__(self, cond: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        key = (cond.size(0), cond.size(1), cond.size(2))
        T = cond.size(2) * self.model.mel_hop_length
        if noise is None:  # the model's own draw (generator.py:315), made outside the graph
            noise = torch.randn(cond.size(0), T, device=cond.device) * self.model.init_noise_scale
        from . import ops
        sig = ops.weights_signature(self.model.parameters())
        entry = self.graphs.get(key)
        if entry is not None and entry[4] != sig:
            entry = None                      # the weights moved since this shape was captured
            self.recaptures += 1
        if entry is None:
            static_c, static_n = cond.clone(), noise.clone()
            kw = dict(n_timesteps=self.n_timesteps, clamp_pred=self.clamp_pred)
            keep: list = []
            was, ops.DERIVED_KEEP = ops.DERIVED_KEEP, keep
            try:
                self.model.infer(cond=static_c, noise=static_n, **kw)  # warm-up, eager
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                # the launch lanes fork / join with events, which the capture records as parallel
                # branches of the graph
                with torch.cuda.graph(graph):
                    out = self.model.infer(cond=static_c, noise=static_n, **kw)
            finally:
                ops.DERIVED_KEEP = was
            entry = (graph, static_c, static_n, out, sig, keep)
            self.graphs[key] = entry
        graph, static_c, static_n, out = entry[:4]
        static_c.copy_(cond)
        static_n.copy_(noise)
        graph.replay()
        return out
