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

import datetime
import os
from typing import Dict, Iterable, List, Optional

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
    if backend == "gloo" and os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
        # single-node rendezvous over loopback: host names need not resolve.  A multi-node job
        # (any other MASTER_ADDR) keeps gloo's own interface choice.
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # a missing peer must surface as an error within minutes, not after torch's 30-minute default
    timeout = datetime.timedelta(seconds=float(os.environ.get("F2G_DIST_TIMEOUT_S", "600")))
    dist.init_process_group(backend, rank=rank, world_size=world_size, timeout=timeout)


def cleanup_dist() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def reduce_metrics(totals: Dict[str, float], device=None, group=None) -> Dict[str, float]:
    """Sum a small dict of validation totals over ranks -- the reference's
    `MetricsTracker.reduce` (utils.py:318-327): keys in sorted order, one fp32 vector, all-reduce
    SUM, written back.  A no-op with one rank."""
    if get_world_size() == 1:
        return dict(totals)
    keys = sorted(totals.keys())
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
    vec = torch.tensor([float(totals[k]) for k in keys], dtype=torch.float32, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return {k: float(v) for k, v in zip(keys, vec.cpu().tolist())}


def end_barrier(group=None) -> None:
    """The barrier the reference's trainers execute before tearing the group down
    (pretrain.py:876, finetune.py:1010); nothing to wait for with one rank."""
    if get_world_size() > 1:
        dist.barrier(group=group)


class GradReducer:
    """Average the .grad of the given parameters across ranks in ~bucket_mb flat buckets."""

    def __init__(self, bucket_mb: float = 128.0, group=None, force: bool = False):
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self.group = group
        self._stream = None
        self.force = force  # run the exchange even with one rank (single-GPU test of the RCCL path)
        self._plans: dict = {}
        self._hooked: set = set()
        self._active: Optional["_Plan"] = None

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


    # ---- overlapped mode: buckets leave as soon as their gradients are final -----------------
    #
    #   reducer.prepare(params)   # before forward: grads become zeroed views into flat arenas
    #   loss.backward()           # a bucket's all-reduce starts on the communication stream the
    #                             # moment autograd has accumulated its last gradient
    #   reducer.finish()          # flush the stragglers, make the compute stream wait
    #
    # Contract: exactly ONE backward() between prepare() and finish() (a second one raises), and
    # every rank must produce gradients for the same parameters (the reference's DDP has the same
    # constraint, find_unused_parameters aside): buckets whose parameters got no gradient at all
    # are skipped on every rank alike; a bucket that fires on one rank only would hang the group.
    #
    # Buckets are cut in REVERSE parameter order (backward reaches the last layers first): in a
    # D-step the MRD bucket travels while the MPD data-gradient convs still run, in a G-step the
    # branch buckets travel under the remaining branches' backward.  The arenas are persistent
    # (288 GB of HBM: 486 MB of gradient arenas is noise), so there is no pack / unpack copy.

    def prepare(self, params: Iterable[torch.nn.Parameter], groups=None) -> None:
        """`groups`: optional lists of parameters whose gradients become final TOGETHER (one
        Fourier branch, one period discriminator): buckets never span two groups, so a group's
        bucket can leave the moment its kernel lane has finished (fused.deliver_grads)."""
        from . import fused
        params = [p for p in params if p.requires_grad]
        if get_world_size() == 1 and not (self.force and dist.is_initialized()):
            # nothing to exchange: plain zero_grad(set_to_none=True), autograd keeps its own buffers
            for p in params:
                p.grad = None
            self._active = None
            fused.GRAD_SINK = None
            return
        gid = {}
        for gi, grp in enumerate(groups or []):
            for p in grp:
                gid[id(p)] = gi
        key = tuple((id(p), gid.get(id(p), -1)) for p in params)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(params, self.bucket_bytes, gid)
            self._plans[key] = plan
            for p in params:
                if id(p) not in self._hooked:
                    self._hooked.add(id(p))
                    p.register_post_accumulate_grad_hook(self._on_grad)
        world = get_world_size()
        plan.exchange = world > 1 or (self.force and dist.is_initialized())
        plan.arm()
        self._active = plan
        fused.GRAD_SINK = _Sink(self, plan)

    def _on_grad(self, p: torch.nn.Parameter, from_sink: bool = False) -> None:
        plan = self._active
        if plan is None:
            return
        b = plan.bucket_of.get(id(p))
        if b is None:
            return
        if not from_sink and id(p) in plan.echo:
            # autograd runs a parameter's AccumulateGrad node -- and this hook -- even when the
            # node that owns it returned None for it (its gradient went through the sink): the
            # one echo per backward is not a gradient
            plan.echo.discard(id(p))
            return
        if b.sent:
            # a second backward() between prepare() and finish() would accumulate into an arena
            # that has already been averaged and would never be exchanged: ranks would diverge
            # silently.  One backward per prepare(); gradient accumulation needs reduce().
            raise RuntimeError("GradReducer: gradient arrived for a bucket that has already been "
                               "exchanged -- exactly one backward() per prepare()/finish() pair "
                               f"(parameter of shape {tuple(p.shape)}, bucket {b.index})")
        b.fired.add(id(p))
        if b.flat.is_cuda:
            # the stream this gradient was accumulated on (a launch lane, or autograd's stream)
            cur = torch.cuda.current_stream(b.flat.device)
            if cur.cuda_stream not in b.streams:
                b.streams[cur.cuda_stream] = cur
        if len(b.fired) == len(b.params) and not b.sent:
            self._send(plan, b)

    @torch.no_grad()
    def _send(self, plan: "_Plan", b: "_Bucket") -> None:
        b.sent = True
        plan.sent_order.append(b.index)
        if not plan.exchange:
            return
        world = get_world_size()
        comm = self._comm_stream(b.flat.device)
        if comm is None:
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
            if world > 1:
                b.flat.div_(world)
        else:
            comm.wait_stream(torch.cuda.current_stream(b.flat.device))
            for st in b.streams.values():     # every lane that accumulated into this bucket
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
                if world > 1:
                    b.flat.div_(world)
        plan.bytes += b.flat.numel() * b.flat.element_size()

    def finish(self) -> int:
        """Send what has not left yet, wait for the exchange; parameters that received no
        gradient in this backward get `.grad = None` back (as without the reducer).  Returns the
        bytes exchanged."""
        from . import fused
        fused.GRAD_SINK = None
        plan, self._active = self._active, None
        if plan is None:
            return 0
        for b in plan.buckets:
            if b.fired and not b.sent:
                self._send(plan, b)
            for p in b.params:
                if id(p) not in b.fired:
                    p.grad = None
        dev = plan.buckets[0].flat.device if plan.buckets else None
        if dev is not None and dev.type == "cuda" and self._stream is not None:
            torch.cuda.current_stream(dev).wait_stream(self._stream)
        return plan.bytes


class _Sink:
    """What fused.deliver_grads talks to while an exchange is armed: the coarse autograd nodes
    (one model evaluation = three branches on three launch lanes; the five period discriminators)
    hand over a finished lane's parameter gradients HERE, inside their backward, instead of
    returning them to autograd when the whole node is done -- so that lane's bucket starts its
    all-reduce on the communication stream while the other lanes still compute (the reference gets
    this from DDP's bucket hooks, finetune.py:915).  A parameter used by several nodes of one
    backward (n_timesteps > 1) is sent after its last use."""

    def __init__(self, reducer: "GradReducer", plan: "_Plan"):
        self.reducer, self.plan = reducer, plan
        self.uses: dict = {}

    def add_use(self, params) -> Optional[int]:
        """Forward side: announce one more backward contribution for this parameter group.
        Returns the key to deliver with, or None when the group is not (wholly) in the plan."""
        ps = [p for p in params if p.requires_grad]
        if not ps or any(id(p) not in self.plan.view_of for p in ps):
            return None
        if any(p.grad is None or p.grad.data_ptr() != self.plan.view_of[id(p)].data_ptr() for p in ps):
            return None
        key = id(ps[0])
        self.uses[key] = self.uses.get(key, 0) + 1
        return key

    @torch.no_grad()
    def deliver(self, key: int, params, grads) -> None:
        """Backward side, on the stream the gradients were computed on: accumulate them into the
        arena views; after the group's last use, count them in (which sends full buckets)."""
        from . import ops
        for p, g in zip(params, grads):
            if g is None or not p.requires_grad:
                continue
            v = self.plan.view_of[id(p)]
            if v.is_cuda:
                n = v.numel()
                ops.axpby_rows(v.view(1, n), v.view(1, n), g.contiguous().view(1, n), sa=1.0, sb=1.0)
            else:
                v.add_(g)
            self.plan.touched.add(id(p))
        self.uses[key] -= 1
        if self.uses[key] == 0:
            for p in params:
                if id(p) in self.plan.touched:
                    self.plan.echo.add(id(p))
                    self.reducer._on_grad(p, from_sink=True)


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], index: int = 0):
        self.params = params
        self.index = index
        self.streams: dict = {}
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.fired: set = set()
        self.sent = False


class _Plan:
    def __init__(self, params: List[torch.nn.Parameter], limit: int, gid: Optional[dict] = None):
        self.buckets: List[_Bucket] = []
        groups: dict = {}
        gid = gid or {}
        # backward order; one arena per (dtype, device) and per delivery group
        for p in reversed(params):
            groups.setdefault((p.dtype, p.device, gid.get(id(p), -1)), []).append(p)
        for plist in groups.values():
            for chunk in GradReducer._buckets(plist, limit):
                self.buckets.append(_Bucket(chunk, len(self.buckets)))
        self.bucket_of = {id(p): b for b in self.buckets for p in b.params}
        self.view_of = {id(p): v for b in self.buckets for p, v in zip(b.params, b.views)}
        self.exchange = False
        self.bytes = 0
        self.sent_order: List[int] = []
        self.touched: set = set()
        self.echo: set = set()

    @torch.no_grad()
    def arm(self) -> None:
        self.bytes = 0
        self.sent_order = []
        self.touched = set()
        self.echo = set()
        for b in self.buckets:
            b.flat.zero_()
            b.fired.clear()
            b.streams = {}
            b.sent = False
            for p, v in zip(b.params, b.views):
                p.grad = v


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
