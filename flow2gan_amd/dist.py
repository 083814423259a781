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
import weakref
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
        self._handles: list = []       # RemovableHandles of the parameter hooks (close())
        self._active: Optional["_Plan"] = None
        self.measure = False          # bench.py: time the compute stream's wait in finish()
        self.wait_events: list = []

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
    # Contract: exactly ONE backward() between prepare() and finish() (a second one raises).
    #
    # What keeps the ranks' collectives matched (RCCL pairs them by issue order, nothing else):
    # EVERY bucket of the plan is exchanged on EVERY rank in every step, in ONE launch order that
    # all ranks share.  The first step of a plan sends nothing early: it records the order in which
    # the buckets became complete, sends them all at finish() in index order, and the ranks agree on
    # the recorded order with one small all-reduce (identical everywhere -> adopted; otherwise
    # index order).  From then on bucket k leaves as soon as it is complete AND its predecessors in
    # that order have left.  A rank on which some parameter gets no gradient in a step (the case
    # the reference covers with DDP's find_unused_parameters=True, finetune.py:915) therefore
    # cannot hang the group: its incomplete bucket -- and whatever follows it in the order --
    # leaves at finish() with zeros for what never arrived, and the other ranks' early sends pair
    # with those.  Each arena carries one "used" flag per parameter behind the gradients, summed by
    # the same all-reduce: a parameter that got no gradient locally keeps the ranks' mean when some
    # rank used it (DDP's semantics) and gets `.grad = None` when nobody did.
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
                    # (weak reference: the hooks must not keep the reducer and its arenas alive as long as
                    # the model lives; close() removes them)
                    me = weakref.ref(self)

                    def after_accumulate(q, me=me):
                        r = me()
                        if r is not None:
                            r._on_grad(q)

                    def before_accumulate(g, p=p, me=me):
                        r = me()
                        if r is not None:
                            r._on_real_grad(p, g)

                    self._handles.append(p.register_post_accumulate_grad_hook(after_accumulate))
                    self._handles.append(p.register_hook(before_accumulate))
        world = get_world_size()
        plan.exchange = world > 1 or (self.force and dist.is_initialized())
        plan.arm()
        self._active = plan
        fused.GRAD_SINK = _Sink(self, plan)

    def _on_real_grad(self, p: torch.nn.Parameter, g) -> None:
        """Tensor hook of a parameter: autograd is about to accumulate `g` into it -- None when the
        node that owns the parameter returned None for it (its gradient went through the sink)."""
        plan = self._active
        if g is not None and plan is not None and id(p) in plan.bucket_of:
            plan.real.add(id(p))

    def _on_grad(self, p: torch.nn.Parameter, from_sink: bool = False) -> None:
        plan = self._active
        if plan is None:
            return
        b = plan.bucket_of.get(id(p))
        if b is None:
            return
        if not from_sink and id(p) in plan.echo and id(p) not in plan.real:
            # autograd runs a parameter's AccumulateGrad node -- and this hook -- even when the
            # node that owns it returned None for it (its gradient went through the sink): the
            # one echo per backward is not a gradient.  (A parameter that was delivered through
            # the sink AND reached by autograd in the same backward -- one use with a ticket, one
            # without -- is in plan.real: its hook is a gradient and falls through to the checks.)
            plan.echo.discard(id(p))
            return
        if not from_sink:
            plan.echo.discard(id(p))        # (delivered through the sink AND reached by autograd: no echo left behind)
        if b.sent:
            # a second backward() between prepare() and finish() would accumulate into an arena
            # that has already been averaged and would never be exchanged: ranks would diverge
            # silently.  One backward per prepare(); gradient accumulation needs reduce().
            raise RuntimeError("GradReducer: gradient arrived for a bucket that has already been "
                               "exchanged -- exactly one backward() per prepare()/finish() pair, and "
                               "every use of a parameter that is handed over per group must carry a "
                               f"ticket (parameter of shape {tuple(p.shape)}, bucket {b.index})")
        b.fired.add(id(p))
        if b.flat.is_cuda:
            # the stream this gradient was accumulated on (a launch lane, or autograd's stream)
            cur = torch.cuda.current_stream(b.flat.device)
            if cur.cuda_stream not in b.streams:
                b.streams[cur.cuda_stream] = cur
        if len(b.fired) == len(b.params) and not b.ready:
            b.ready = True
            plan.ready_order.append(b.index)
            self._pump(plan)

    def _pump(self, plan: "_Plan") -> None:
        """Send every complete bucket whose predecessors in the agreed order have left."""
        if plan.order is None:          # first step of this plan: the order is being recorded
            return
        while plan.next_pos < len(plan.order) and plan.buckets[plan.order[plan.next_pos]].ready:
            self._send(plan, plan.buckets[plan.order[plan.next_pos]])
            plan.next_pos += 1

    @torch.no_grad()
    def _send(self, plan: "_Plan", b: "_Bucket") -> None:
        b.sent = True
        plan.sent_order.append(b.index)
        if not plan.exchange:
            return
        world = get_world_size()
        comm = self._comm_stream(b.flat.device)

        def exchange():
            # the "used" flags behind the gradients: arm() left them all at 1 (the common case: every
            # parameter of the bucket got a gradient -- nothing to launch in front of the collective);
            # only a partially fired bucket writes its pattern (rare: a branch without gradient)
            if len(b.fired) != len(b.params):
                # (one pinned staging buffer per bucket, allocated on its first partial send: a pinned
                # allocation per send would synchronise the device under the overlapped backward)
                pat = b.flag_staging()
                for i, p in enumerate(b.params):
                    pat[i] = 1.0 if id(p) in b.fired else 0.0
                b.flat[b.n:].copy_(pat, non_blocking=True)
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
            if world > 1:
                if b.hip_fp32:
                    from . import ops
                    ops.call("f2g_scale", ops.ptr(b.flat), 1.0 / world, b.n)      # one launch behind the collective
                else:
                    b.flat[:b.n].div_(world)

        if comm is None:
            exchange()
        else:
            comm.wait_stream(torch.cuda.current_stream(b.flat.device))
            for st in b.streams.values():     # every lane that accumulated into this bucket
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                exchange()
        plan.bytes += b.flat.numel() * b.flat.element_size()

    def _agree_on_order(self, plan: "_Plan") -> None:
        """End of a plan's first step: adopt the recorded completion order if every rank recorded
        the same one (buckets that never became complete here follow in index order)."""
        nb = len(plan.buckets)
        seen = set(plan.ready_order)
        mine = list(plan.ready_order) + [i for i in range(nb) if i not in seen]
        if plan.exchange and get_world_size() > 1:
            dev = plan.buckets[0].flat.device
            comm = self._comm_stream(dev)
            v = torch.tensor(mine + [-i for i in mine], dtype=torch.int64)
            if comm is None:
                dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
            else:
                with torch.cuda.stream(comm):
                    v = v.to(dev)
                    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
                comm.synchronize()
            v = v.cpu().tolist()
            same = all(v[i] == -v[nb + i] for i in range(nb))      # max == min on every position
            plan.order = [int(x) for x in v[:nb]] if same else list(range(nb))
        else:
            plan.order = mine

    def finish(self) -> int:
        """Send what has not left yet (every bucket travels in every step), wait for the exchange.
        A parameter that received no gradient here keeps the ranks' mean if another rank used it and
        gets `.grad = None` back if nobody did (as without the reducer).  Returns the bytes exchanged."""
        from . import fused
        fused.GRAD_SINK = None
        plan, self._active = self._active, None
        if plan is None:
            return 0
        learning = plan.order is None
        for i in (range(len(plan.buckets)) if learning else plan.order[plan.next_pos:]):
            if not plan.buckets[i].sent:
                self._send(plan, plan.buckets[i])
        plan.next_pos = len(plan.buckets)
        if learning:
            self._agree_on_order(plan)
        dev = plan.buckets[0].flat.device if plan.buckets else None
        cuda = dev is not None and dev.type == "cuda" and self._stream is not None
        synced = False
        for b in plan.buckets:
            if len(b.fired) == len(b.params):
                continue
            # (rare) some parameter got no gradient on this rank: did any rank use it?
            if cuda and plan.exchange and not synced:
                self._stream.synchronize()         # (once, whatever the number of partial buckets)
                synced = True
            used = b.flat[b.n:].tolist() if plan.exchange else [0.0] * len(b.params)
            for p, u in zip(b.params, used):
                if id(p) not in b.fired and u <= 0.0:
                    p.grad = None
        if cuda:
            if self.measure:
                # exposed communication: how long the compute stream has to wait for the exchange
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cur = torch.cuda.current_stream(dev)
                e0.record(cur)
                cur.wait_stream(self._stream)
                e1.record(cur)
                self.wait_events.append((e0, e1))
            else:
                torch.cuda.current_stream(dev).wait_stream(self._stream)
        return plan.bytes

    def close(self) -> None:
        """Remove the parameter hooks and drop the arenas (a second reducer over the same parameters would
        otherwise stack its hooks on top of these)."""
        for h in self._handles:
            h.remove()
        self._handles, self._hooked, self._plans, self._active = [], set(), {}, None

    def exposed_comm_ms(self) -> float:
        """Sum of the measured waits of finish() since the last call (self.measure = True); syncs."""
        evs, self.wait_events = self.wait_events, []
        if not evs:
            return 0.0
        evs[-1][1].synchronize()
        return float(sum(a.elapsed_time(b) for a, b in evs))


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
        self.n = n
        # the gradients, then one "used" flag per parameter (summed over ranks by the same all-reduce)
        self.flat = torch.zeros(n + len(params), dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.fired: set = set()
        self.ready = False
        self.sent = False
        # f2g_bucket_arm / f2g_scale are float kernels: only fp32 arenas on the GPU take them (any other
        # parameter dtype keeps the dtype-agnostic zero_ / fill_ / div_)
        self.hip_fp32 = self.flat.is_cuda and self.flat.dtype == torch.float32
        self._staging = None

    def flag_staging(self) -> torch.Tensor:
        """Host buffer of the "used" flags of a partially fired bucket (pinned for CUDA arenas), reused by
        every send: a step's copy has run before the next step's arm() -- finish() waits for the
        communication stream -- so one buffer per bucket is enough."""
        if self._staging is None:
            self._staging = torch.empty(len(self.params), dtype=self.flat.dtype)
            if self.flat.is_cuda:
                self._staging = self._staging.pin_memory()
        return self._staging


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
        self.real: set = set()
        self.order: Optional[List[int]] = None     # the ranks' common launch order (None: first step)
        self.ready_order: List[int] = []
        self.next_pos = 0

    @torch.no_grad()
    def arm(self) -> None:
        self.bytes = 0
        self.sent_order = []
        self.ready_order = []
        self.next_pos = 0
        self.touched = set()
        self.echo = set()
        self.real = set()
        for b in self.buckets:
            if b.hip_fp32:           # gradients = 0, "used" flags = 1: one launch per arena
                from . import ops
                ops.call("f2g_bucket_arm", ops.ptr(b.flat), b.n, len(b.params))
            else:
                b.flat.zero_()
                b.flat[b.n:].fill_(1.0)
            b.fired.clear()
            b.streams = {}
            b.ready = False
            b.sent = False
            for p, v in zip(b.params, b.views):
                p.grad = v


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
