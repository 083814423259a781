"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference wraps the whole GAN in torch DDP (finetune.py:915) and therefore ships the
gradients of G *and* D (485.8 MB fp32) on every step.  Only the sub-model being stepped has live
gradients, so `GradReducer` reduces exactly those (D-step 170 MB, G-step 315.8 MB) -- identical
result, 35-65 % of the bytes.  Buckets are flattened and reduced on a dedicated communication
stream so that bucket k's all-reduce overlaps bucket k+1's packing; gradients are averaged over
ranks (DDP semantics).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.

Reference: flow2gan/dist.py:23-48 (process-group setup), pretrain.py:792, finetune.py:915.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def setup_dist(rank: Optional[int] = None, world_size: Optional[int] = None,
               master_port: Optional[int] = None, backend: Optional[str] = None) -> None:
    """Mirror of reference dist.py:23-44; reads torchrun's environment when arguments are None."""
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = "12354" if master_port is None else str(master_port)
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group(backend, rank=rank, world_size=world_size)


def cleanup_dist() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer:
    """Average the .grad of the given parameters across ranks in ~bucket_mb flat buckets."""

    def __init__(self, bucket_mb: float = 128.0, group=None, force: bool = False):
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self.group = group
        self._stream = None
        self.force = force  # run the exchange even with one rank (single-GPU test of the RCCL path)

    def _comm_stream(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    @staticmethod
    def _buckets(params: List[torch.nn.Parameter], limit: int) -> List[List[torch.nn.Parameter]]:
        out, cur, size = [], [], 0
        for g in params:
            nbytes = g.numel() * g.element_size()
            if cur and size + nbytes > limit:
                out.append(cur)
                cur, size = [], 0
            cur.append(g)
            size += nbytes
        if cur:
            out.append(cur)
        return out

    @torch.no_grad()
    def reduce(self, params: Iterable[torch.nn.Parameter]) -> int:
        """All-reduce (mean) the gradients of `params`.  Afterwards every p.grad is a view into its
        bucket's flat buffer (no copy back).  Returns the bytes exchanged."""
        world = get_world_size()
        live = [p for p in params if p.grad is not None]
        if not live or (world == 1 and not (self.force and dist.is_initialized())):
            return 0
        device = live[0].grad.device
        comm = self._comm_stream(device)
        total = 0
        if comm is not None:
            comm.wait_stream(torch.cuda.current_stream(device))
        ctx = torch.cuda.stream(comm) if comm is not None else _Null()
        with ctx:
            for bucket in self._buckets(live, self.bucket_bytes):
                flat = torch.cat([p.grad.reshape(-1) for p in bucket])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                if world > 1:
                    flat.div_(world)
                off = 0
                for p in bucket:
                    n = p.numel()
                    if comm is not None:
                        p.grad.record_stream(comm)  # still being read by the cat on `comm`
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
                if comm is not None:
                    flat.record_stream(torch.cuda.current_stream(device))
                total += flat.numel() * flat.element_size()
        if comm is not None:
            torch.cuda.current_stream(device).wait_stream(comm)
        return total


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
