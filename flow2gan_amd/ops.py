"""Thin Python wrappers over the C ABI: build descriptors, launch on the current stream.

Tensors are channels-last "rows" matrices: a 2-D contiguous fp32 tensor (rows, ld) whose first
`cols` columns are meaningful.  Nothing here computes on the host or falls back to ATen.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from ._opts import opt
from ._lib import DwconvBwd, DwnormBwd, DwnormFwd, Epilogue, GemmDesc, Operand, call, ptr


def empty(*shape, device, dtype=torch.float32):
    return torch.empty(*shape, device=device, dtype=dtype)


def zeros(*shape, device):
    t = torch.empty(*shape, device=device, dtype=torch.float32)
    fill_(t, 0.0)
    return t


def fill_(t, v: float):
    if BATCH is not None and t.numel() > 0:
        import struct
        n = t.numel()
        BATCH.add(0, ptr(t), 4 * n, None, 0, (n & 0xffffffff, n >> 32, 0, 0),
                  (struct.unpack("<i", struct.pack("<f", float(v)))[0], 0, 0, 0), n, (t,))
        return t
    call("f2g_fill", ptr(t), float(v), t.numel())
    return t


def zero_halo(buf, nseq: int, rows_per_seq: int, Cc: int, lo: int, hi: int):
    call("f2g_zero_halo", ptr(buf), nseq, rows_per_seq, Cc, lo, hi)
    return buf


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


# ------------------------------------------------------------------ derived-weight cache
# Re-laid copies of parameters (transposes for data gradients, window-major conv weights) are
# rebuilt only when the parameter changed: keyed on the parameter object (weakly), its version
# counter (every in-place torch write bumps it) and an epoch that writers going through RAW
# POINTERS must bump: `bump_weight_epoch(params)` for exactly the tensors they wrote (the HIP
# optimizer does this for its own parameters, so a D-step leaves the generator's copies alone and
# vice versa), `bump_weight_epoch()` for "anything may have changed" (e.g. after `p.data.copy_()`,
# which does not move the version counter of `p`).  Cached copies of copies (`_f2g_const` owners:
# the split / bf16 image of a transposed weight, DFT and filterbank tables) are immutable by
# construction -- when their source changes a NEW tensor replaces them -- and carry no epoch.
import weakref

WEIGHT_EPOCH = 0
_OWNER_EPOCH: dict = {}
_DERIVED: dict = {}


def bump_weight_epoch(params=None) -> None:
    """Tell the derived-weight cache that parameters were written behind autograd's back."""
    global WEIGHT_EPOCH
    if params is None:
        WEIGHT_EPOCH += 1
        return
    for t in params:
        owner = t._base if t._base is not None else t
        _OWNER_EPOCH[id(owner)] = _OWNER_EPOCH.get(id(owner), 0) + 1


def _stamp(owner) -> tuple:
    if getattr(owner, "_f2g_const", False) and not isinstance(owner, torch.nn.Parameter):
        return (owner._version,)
    return (owner._version, WEIGHT_EPOCH, _OWNER_EPOCH.get(id(owner), 0))


# While a list is installed here (streaming.ChunkRunner during its warm-up and capture), every
# cached copy handed out is appended to it: a captured HIP graph holds raw pointers into those
# copies, so its owner must keep them alive for as long as it may replay.
DERIVED_KEEP = None


def weights_signature(params) -> tuple:
    """Changes whenever one of `params` may have changed, through autograd-visible writes (version
    counters) or raw-pointer writers (the epochs above)."""
    ps = list(params)
    return (WEIGHT_EPOCH, tuple(p._version for p in ps),
            tuple(_OWNER_EPOCH.get(id(p), 0) for p in ps))


def derived(t, tag, build):
    """build(t) cached per (tensor object or its view base, tag) until the tensor changes."""
    out = _derived(t, tag, build)
    if DERIVED_KEEP is not None:
        DERIVED_KEEP.append(out)
    return out


def _derived(t, tag, build):
    import os
    if os.environ.get("F2G_WEIGHT_CACHE", "1") == "0":
        return build(t)
    owner = t._base if t._base is not None else t
    oid = id(owner)          # (tensors compare element-wise: never use them as dictionary keys)
    ent = _DERIVED.get(oid)
    if ent is None or ent[0]() is not owner:
        def _gone(_r, oid=oid, d=_DERIVED, e=_OWNER_EPOCH, rc=_RECIPES):   # (bound now: globals are gone at exit)
            d.pop(oid, None)
            e.pop(oid, None)
            rc.pop(oid, None)
        ent = (weakref.ref(owner, _gone), {})
        _DERIVED[oid] = ent
        # (the epoch of a dead tensor whose id was recycled is dropped by ITS weakref callback above;
        # nothing is reset here: the optimizer may have bumped this owner before its first copy was
        # asked for, and weights_signature() needs the counter to never run backwards)
    slot = ent[1]
    key = (tag, t.data_ptr(), tuple(t.shape), tuple(t.stride()))
    stamp = _stamp(owner)
    hit = slot.get(key)
    if hit is not None and hit[0] == stamp:
        if hit[2] is not None and not _REPLAYING:
            hit[2][2] = True          # (this chain is in use: the next rebuild_derived replays it)
        return hit[1]
    out = build(t)
    rec = None
    if isinstance(out, torch.Tensor):
        out._f2g_const = True     # a cached re-layout of a parameter (see _split_operand)
        # how to make it again: the chain of (tag, builder, view) from the root parameter
        spec = (tuple(t.shape), tuple(t.stride()), t.storage_offset() - owner.storage_offset())
        parent = getattr(owner, "_f2g_recipe", None)
        if isinstance(owner, torch.nn.Parameter) or parent is not None:
            root_ref, chain = (weakref.ref(owner), ()) if parent is None else parent
            chain = chain + ((tag, build, spec),)
            root = root_ref()
            if root is not None and len(chain) <= 4:
                ckey = tuple((c[0], c[2]) for c in chain)
                rec = _RECIPES.setdefault(id(root), {}).get(ckey)
                if rec is None or rec[0]() is not root:
                    rec = [root_ref, chain, False]
                    _RECIPES[id(root)][ckey] = rec
                else:
                    rec[1] = chain
                if not _REPLAYING:
                    rec[2] = True
                out._f2g_recipe = (root_ref, chain)
    slot[key] = (stamp, out, rec)
    return out


_DERIVED_MULTI: dict = {}


def derived_multi(ts, tag, build):
    """build(ts) cached for a LIST of parameters (stacked copies) until one of them changes."""
    out = _derived_multi(ts, tag, build)
    if DERIVED_KEEP is not None:
        DERIVED_KEEP.append(out)
    return out


def _derived_multi(ts, tag, build):
    import os
    if os.environ.get("F2G_WEIGHT_CACHE", "1") == "0":
        return build(ts)
    key = (tag,) + tuple(id(t) for t in ts)
    stamp = tuple(_stamp(t._base if t._base is not None else t) for t in ts)
    ent = _DERIVED_MULTI.get(key)
    if ent is not None and ent[0] == stamp and all(r() is t for r, t in zip(ent[2], ts)):
        return ent[1]
    out = build(ts)
    out._f2g_const = True
    if len(_DERIVED_MULTI) > 4096:     # ids of dead parameters: start over
        _DERIVED_MULTI.clear()
    _DERIVED_MULTI[key] = (stamp, out, [weakref.ref(t) for t in ts])
    return out



# ------------------------------------------------------------------ batched rebuild of the weight images
# A train step writes every parameter of the sub-model it stepped, so all copies derived from them are stale at
# the next use: ~830 launches of 3-6 us per mel_24k_base GAN step when each is rebuilt by its own permute4 /
# copy3 / fill / split launch at first use.  Instead (round 6) every cache entry remembers HOW it was made -- the
# chain (tag, builder, view) from its root parameter -- and `rebuild_derived(params)`, called by the optimizer
# (or by bench.py's stand-in for it) right after the weights changed, replays the chains that were USED since
# the last rebuild with the builders' launches recorded into a WeightBatch: the recorded operations are levelled
# by their data dependencies (an image of a transposed copy comes after the transpose) and every level leaves
# as f2g_multi launches of up to 48 operations each.
BATCH = None
EAGER_REBUILD = opt("eager_rebuild", True)
_RECIPES: dict = {}          # id(root parameter) -> {chain key: [weakref(root), chain, used]}
_REPLAYING = False


MULTI_CAP = min(L.MULTI_MAX, max(1, opt("multi_cap", L.MULTI_MAX)))     # operations per f2g_multi table (1: lab, one launch each)


class WeightBatch:
    """Records fill / permute4 / copy3 / split3 operations (ops.fill_, permute4, copy3, split3 while
    ops.BATCH is this object) and launches them as f2g_multi tables, dependency level by level."""

    MAX_BLOCKS = 512

    def __init__(self):
        self.ops = []          # (level, kind, out_ptr, in_ptr, n, s, items)
        self.keep = []         # tensors of the recorded operations (alive until the launches are issued)
        self.starts = []       # sorted start addresses of written ranges
        self.ranges = {}       # start -> (end, level)
        self.rstarts = []      # ... of read ranges (write-after-read ordering when memory is reused)
        self.rranges = {}
        self.longest = 0       # bytes of the longest recorded range

    def _level_over(self, starts, ranges, lo, hi):
        """Highest level among the recorded ranges that overlap [lo, hi), -1 if none.  Ranges are looked up by
        start address; the scan to the left stops as soon as a start lies further below `lo` than the longest
        range recorded so far (a few steps: the operations' buffers are separate allocations)."""
        import bisect
        lvl = -1
        i = bisect.bisect_left(starts, hi)
        while i > 0:
            i -= 1
            st = starts[i]
            if lo - st >= self.longest:
                break
            end, l = ranges[st]
            if end > lo:
                lvl = max(lvl, l)
        return lvl

    def add(self, kind, out_ptr, out_bytes, rd_ptr, rd_bytes, n, s, items, tensors, in_ptr=None, reads_out=False):
        import bisect
        self.longest = max(self.longest, out_bytes, rd_bytes or 0)
        lvl = self._level_over(self.starts, self.ranges, out_ptr, out_ptr + out_bytes)           # write after write
        lvl = max(lvl, self._level_over(self.rstarts, self.rranges, out_ptr, out_ptr + out_bytes))   # ... after read
        if rd_ptr is not None:
            lvl = max(lvl, self._level_over(self.starts, self.ranges, rd_ptr, rd_ptr + rd_bytes))    # read after write
        lvl += 1
        old = self.ranges.get(out_ptr)
        if old is None:
            bisect.insort(self.starts, out_ptr)
        self.ranges[out_ptr] = (max(out_ptr + out_bytes, old[0] if old else 0), max(lvl, old[1] if old else 0))
        if rd_ptr is not None:
            oldr = self.rranges.get(rd_ptr)
            if oldr is None:
                bisect.insort(self.rstarts, rd_ptr)
            self.rranges[rd_ptr] = (max(rd_ptr + rd_bytes, oldr[0] if oldr else 0), max(lvl, oldr[1] if oldr else 0))
        self.ops.append((lvl, kind, out_ptr, in_ptr if in_ptr is not None else rd_ptr, tuple(n), tuple(s), items))
        self.keep.extend(tensors)

    def flush(self):
        if not self.ops:
            return 0
        launches = 0
        ops_, self.ops = sorted(self.ops, key=lambda o: o[0]), []
        i = 0
        while i < len(ops_):
            lvl = ops_[i][0]
            d = L.MultiDesc()
            k = 0
            while i < len(ops_) and ops_[i][0] == lvl and k < MULTI_CAP:
                _, kind, out_ptr, in_ptr, n, s, items = ops_[i]
                e = d.e[k]
                e.out, e.inp, e.kind = out_ptr, in_ptr, kind
                e.blocks = max(1, min(self.MAX_BLOCKS, (items + 255) // 256))
                for j in range(4):
                    e.n[j], e.s[j] = int(n[j]), int(s[j])
                k += 1
                i += 1
            d.n = k
            call("f2g_multi", C.byref(d))
            launches += 1
        self.keep = []
        self.starts, self.ranges, self.rstarts, self.rranges, self.longest = [], {}, [], {}, 0
        return launches


class weight_batch:
    """Context: collect the re-layout launches issued inside into f2g_multi launches (flushed on exit, and
    before any other kernel call made inside)."""

    def __enter__(self):
        global BATCH
        self.prev = BATCH
        if self.prev is None:
            BATCH = WeightBatch()
            L.PRE_CALL = BATCH.flush
        return BATCH

    def __exit__(self, *exc):
        global BATCH
        if self.prev is None:
            b, BATCH = BATCH, None
            L.PRE_CALL = None
            b.flush()
        return False


def rebuild_derived(params) -> int:
    """Rebuild -- batched -- the cached copies derived from `params` (tensors just written by an optimizer
    step) that were used since their last rebuild.  Returns the number of chains replayed."""
    global _REPLAYING
    if not EAGER_REBUILD or GEMM_PRECISION in (1, 2):
        # (the split-bf16 / plain-bf16 modes build their images with f2g_split_bf16 / f2g_to_bf16 / f2g_mlp_pack,
        # which are not f2g_multi operations -- each would flush the batch in front of itself: measured 29.9
        # vs 27.0 ms per stage-1 step against the lazy path -- so those throughput modes keep rebuilding at
        # first use)
        return 0
    n = 0
    roots = {}
    for p_ in params:
        o = p_._base if p_._base is not None else p_
        roots[id(o)] = o
    import time
    t_host = time.perf_counter()
    _REPLAYING = True
    try:
        with torch.no_grad(), weight_batch():
            for rid, root in roots.items():
                rec = _RECIPES.get(rid)
                if not rec:
                    continue
                for key in sorted(rec, key=len):          # parents before children
                    ent = rec[key]
                    if ent[0]() is not root:
                        del rec[key]
                        continue
                    if not ent[2]:
                        continue
                    ent[2] = False
                    cur = root
                    for tag, build, spec in ent[1]:
                        shape, stride, off = spec
                        cur = _derived(cur.as_strided(shape, stride, cur.storage_offset() + off), tag, build)
                    n += 1
    finally:
        _REPLAYING = False
        REBUILD_STATS[0] += time.perf_counter() - t_host
        REBUILD_STATS[1] += n
    return n


REBUILD_STATS = [0.0, 0]       # host seconds spent in rebuild_derived, chains replayed (bench.py reports them)


def transposed(w2d):
    """(n, k) row-major -> cached (k, n) row-major copy."""
    def build(t):
        n, k = t.shape
        out = torch.empty(k, n, device=t.device, dtype=torch.float32)
        permute4(out, t, (k, n, 1, 1), (1, t.stride(0), 0, 0))
        return out
    return derived(w2d, "T", build)


# ------------------------------------------------------------------ operands
def mat(t, rows: Optional[int] = None, cols: Optional[int] = None, ld: Optional[int] = None,
        alpha=None, lrelu_src=None, slope: float = 0.0, offset: int = 0, split: int = 0) -> Operand:
    """Plain row-major matrix view of `t` starting `offset` floats in.  split: what `t` holds --
    0 fp32 values, 1 their split-bf16 image (a producer wrote it: f2g_operand.split), 2 bf16."""
    if rows is None:
        rows = t.shape[0]
    if ld is None:
        ld = t.stride(0) if t.dim() == 2 else (cols if cols is not None else t.shape[-1])
    if cols is None:
        cols = t.shape[-1]
    o = Operand()
    o.base = ptr(t) + (2 if split == 2 else 4) * offset
    o.split = split
    o.rows, o.cols = rows, cols
    o.P1 = o.P0 = 1
    o.seglen = cols
    o.step1, o.pad1, o.L1 = 0, 0, 1
    o.step0, o.pad0, o.unit, o.L0u = 0, 0, 1, cols
    o.seq_stride, o.line_stride = ld, 0
    o.reflect = 0
    o.alpha = ptr(alpha)
    o.lrelu_src = None if lrelu_src is None else ptr(lrelu_src) + 4 * offset
    o.lrelu_slope = slope
    o._keep = (t, alpha, lrelu_src)
    o._src = t if (offset == 0 and t.dim() == 2 and rows == t.shape[0] and cols == t.shape[1]
                   and ld == t.stride(0) and alpha is None and lrelu_src is None) else None
    return o


def win1d(t, nseq: int, L_in: int, unit: int, L_out: int, step: int, pad: int, taps: int,
          reflect: bool = False, seq_stride: Optional[int] = None, lrelu_src=None,
          slope: float = 0.0, offset: int = 0, unbounded: bool = False) -> Operand:
    """rows = (seq, out position); cols = taps*unit contiguous floats starting at
    (pos*step - pad)*unit inside a sequence of L_in*unit floats (zero / reflect outside)."""
    o = Operand()
    o.base = ptr(t) + 4 * offset
    o.rows, o.cols = nseq * L_out, taps * unit
    o.P1, o.P0 = 1, L_out
    o.seglen = taps * unit
    o.step1, o.pad1, o.L1 = 0, 0, 1
    o.step0, o.pad0, o.unit, o.L0u = step, pad, unit, L_in * unit
    o.seq_stride = seq_stride if seq_stride is not None else L_in * unit
    o.line_stride = 0
    o.reflect = 1 if reflect else 0
    o.alpha = None
    o.lrelu_src = None if lrelu_src is None else ptr(lrelu_src) + 4 * offset
    o.lrelu_slope = slope
    # unbounded: the caller guarantees that rows whose window leaves its sequence pair with zeros
    # (weight gradients against a gradient map with zero halo rows), so a kernel may read past the
    # sequence ends inside `t` instead of testing bounds (f2g_operand.unbounded); t = whole buffer
    o.unbounded = 1 if (unbounded and offset == 0 and t.is_contiguous()
                        and t.numel() == nseq * o.seq_stride) else 0
    o._keep = (t, lrelu_src)
    return o


def win2d(t, nseq: int, H: int, W: int, Cin: int, W_out: int, kh: int, kw: int, stride_w: int,
          pad_h: int, pad_w: int, line_stride: Optional[int] = None,
          seq_stride: Optional[int] = None, offset: int = 0, lrelu_src=None,
          slope: float = 0.0) -> Operand:
    """rows = (seq, h, w_out); cols = kh segments of kw*Cin contiguous floats (conv (kh,kw),
    stride (1, stride_w)) over a channels-last image (seq, H, W, Cin)."""
    o = Operand()
    o.base = ptr(t) + 4 * offset
    o.rows, o.cols = nseq * H * W_out, kh * kw * Cin
    o.P1, o.P0 = H, W_out
    o.seglen = kw * Cin
    o.step1, o.pad1, o.L1 = 1, pad_h, H
    o.step0, o.pad0, o.unit, o.L0u = stride_w, pad_w, Cin, W * Cin
    o.line_stride = line_stride if line_stride is not None else W * Cin
    o.seq_stride = seq_stride if seq_stride is not None else H * o.line_stride
    o.reflect = 0
    o.alpha = None
    o.lrelu_src = None if lrelu_src is None else ptr(lrelu_src) + 4 * offset
    o.lrelu_slope = slope
    o._keep = (t, lrelu_src)
    return o


def gemm(A: Operand, Bm: Operand, out, form: int = 0, ldc: Optional[int] = None, *, bias=None,
         res=None, ldres: Optional[int] = None, gamma=None, aux=None, ldaux: Optional[int] = None,
         alpha_n=None, colsum_alpha=None, colsum=None, lrelu: float = 0.0, scale: float = 0.0,
         accumulate: bool = False, atomic: bool = False, split_k: int = 0, out_offset: int = 0,
         rowmap=None, prelu=None, prelu_out=None, mask=None, fm=None, x3_out: bool = False,
         true_k: Optional[int] = None, true_n: Optional[int] = None):
    """Launch f2g_gemm.  rowmap = (P0o, seq_stride_o, row_stride_o, off_o) or None.
    true_k / true_n: the reduction length / output width WITHOUT zero padding (the spectrum rows are
    padded to whole K slabs): what bench.py's FLOP count uses; the launch itself ignores them.
    split_k: 0 = let the library decide (forms 0/1: split-K onto a zeroed output when the tile
    grid would leave most of the last wave of CUs idle), 1 = off, > 1 = as given."""
    if form == 1 and LEAN_DGRAD:
        # data gradient C[r,n] = sum_k A[r,k] W[k,n] as a forward GEMM against the cached transpose
        # W^T [n][k]: same products in the same order, and the lean forward kernel applies
        src = getattr(Bm, "_src", None)
        if src is not None and src.shape[0] % 32 == 0 and src.shape[1] > 64:
            Bm, form = mat(transposed(src)), 0
    d = GemmDesc()
    d.A, d.B = A, Bm
    e = Epilogue()
    e.C = ptr(out) + 4 * out_offset
    e.ldc = ldc if ldc is not None else out.stride(0)
    e.c_bf16 = 1 if out.dtype == torch.bfloat16 else 0   # (lean kernel, plain stores only)
    if rowmap is not None:
        e.P0o, e.seq_stride_o, e.row_stride_o, e.off_o = rowmap
    else:
        e.P0o = 0
    e.bias = ptr(bias)
    e.res = ptr(res)
    e.ldres = ldres if ldres is not None else (res.stride(0) if res is not None else 0)
    e.gamma = ptr(gamma)
    e.aux = ptr(aux)
    e.ldaux = ldaux if ldaux is not None else (aux.stride(0) if aux is not None else 0)
    e.alpha_n = ptr(alpha_n)
    e.colsum_alpha = ptr(colsum_alpha)
    e.colsum = ptr(colsum)
    e.lrelu_slope = lrelu
    e.scale = scale
    e.accumulate = 1 if accumulate else 0
    e.atomic = 1 if atomic else 0
    e.prelu_slope = ptr(prelu)
    e.prelu_out = ptr(prelu_out)
    e.ld_prelu_out = prelu_out.stride(0) if prelu_out is not None else 0
    if mask is not None:      # (activation tensor, float offset, slope): leaky-ReLU backward below
        e.mask_src = ptr(mask[0]) + 4 * mask[1]
        e.mask_slope = float(mask[2])
        if fm is not None:    # (reference activation tensor, float offset, weight, device scalar)
            e.fm_ref = ptr(fm[0]) + 4 * fm[1]
            e.fm_w = float(fm[2])
            e.fm_wdev = ptr(fm[3])
    d.E = e
    d.form = form
    if split_k == -1:
        # weight gradient, factor chosen here: where the exact-fp32 launch takes the K-major lean kernel anyway
        # (>= 4096 rows per block with the general rule's factor), its blocks are dealt in rounds of 512
        tiles = ((A.cols + 127) // 128) * ((Bm.cols + 127) // 128)
        split_k = split_for(A.rows, tiles)
        if WGRAD_SPLIT512 and form == 2 and GEMM_PRECISION == 0 and atomic:
            # (the library's own dispatch rule, asked with the descriptor as it would be launched)
            d.split_k, d.precision = split_k, 0
            if L.lib.f2g_gemm_wgrad_lean(C.byref(d)):
                split_k = split_for(A.rows, tiles, True)
    d.split_k = split_k
    d.precision = GEMM_PRECISION
    if GEMM_PRECISION == 3:
        d.precision = 0       # (what does not qualify below runs on the exact fp32 MFMA)
        in_kernel = X6F == 1
        if X6F == 2 and X6F_MIN_K <= A.cols < X6_MIN_K and (
                (Bm.rows >= X6F_MIN_N and ((A.rows + 127) // 128) * ((Bm.rows + 127) // 128) >= X6F_MIN_TILES)
                or (A.rows >= X6F_TALL_ROWS and Bm.rows >= 128)):
            in_kernel = True      # (mid-length reductions on well-filled grids: see X6F)
        if X6F == 2 and X6N and 128 <= A.cols < X6_MIN_K and A.rows >= X6F_TALL_ROWS and Bm.rows == 32:
            in_kernel = True      # (tall GEMMs with 32 output columns: gemm_x6n_kernel)
        if X6F == 2 and X6_MIN_K <= A.cols < X6_NOPASS_K and A.P0 == 1 and A.P1 == 1 and not _is_const(A._keep[0]):
            in_kernel = True      # (a plain activation matrix has no producer-written image: no image pass)
        if form == 0 and A.split == 0 and Bm.split == 0 and A.rows >= X6_MIN_ROWS \
                and (A.cols >= X6_MIN_K or in_kernel) and not atomic \
                and split_k <= 1 and out.dtype == torch.float32 and _x3_window_ok(A):
            # the three-piece image of `out` for the next GEMM, written by this one's epilogue (every
            # writer of the buffer must do so, or the image is dropped: x3_reserve / _x3_operand).  The
            # library is asked with the descriptor as it will be launched (f2g_gemm_x6_ok applies the
            # tests of f2g_gemm's own dispatch, E.x3_out included)
            buf = getattr(out, "_f2g_x3_buf", None) if x3_out else None
            if buf is not None and X6F != 1 and not getattr(out, "_f2g_x3_bad", False) \
                    and out_offset % 32 == 0 and prelu_out is None:
                d.E.x3_out = ptr(buf) + (out_offset // 32) * 192
            how = L.lib.f2g_gemm_x6_ok(C.byref(d))
            if not how and d.E.x3_out:      # (the image's alignment conditions: the GEMM alone may still go)
                d.E.x3_out = None
                how = L.lib.f2g_gemm_x6_ok(C.byref(d))
            if how:
                if in_kernel and not (how & 4):
                    # the in-kernel split cannot read these views (alignment / strides): the image
                    # kernel where the reduction is long enough for it, else the exact fp32 MFMA
                    in_kernel = False
                    how = how if A.cols >= X6_MIN_K else 0
                if how:
                    if not in_kernel:   # (else gemm_x6f_kernel reads the fp32 operands and splits them itself)
                        d.A, d.B = _x3_operand(A), _x3_operand(Bm)
                    elif X6F_WIMG and Bm.P0 == 1 and Bm.P1 == 1 and _is_const(Bm._keep[0]):
                        # the WEIGHT operand as its cached image: every row tile used to split the same weight
                        # slab again (M / 128 times); only the activation is split in the kernel.  Whole 32-row
                        # groups: the FRAGMENT-MAJOR image, read straight into MFMA registers (gemm_x6g_kernel)
                        d.B = _x3_operand(Bm, frag_major=X6G and Bm.rows % 32 == 0 and Bm.rows > 32)
                    d.precision = 3
            if d.precision != 3:
                d.E.x3_out = None
        if form == 2 and X6_WGRAD and atomic and A.split == 0 and Bm.split == 0 and A.lrelu_src is None \
                and A.rows >= X6_MIN_K and L.lib.f2g_gemm_lean_ok(C.byref(d)):
            d.precision = 3       # weight gradient: gemm_leanw6_kernel splits the fp32 operands itself
        if x3_out:
            if d.E.x3_out:
                out._f2g_x3 = out._f2g_x3_buf
            else:
                out._f2g_x3_bad = True
    if ((GEMM_PRECISION == 1 and form in (0, 2)) or (GEMM_PRECISION == 2 and form == 0)) and LEAN_SPLIT:
        ok = L.lib.f2g_gemm_lean_ok(C.byref(d))
        if GEMM_PRECISION == 2 and (ok & 2) and (A.split == 2 or BF16_IMAGES):
            # plain bf16 over TRUE bf16 tensors (64-element slabs): activations written as bf16 by
            # their producers (or converted here), weights from the derived-weight cache
            d.A, d.B = _bf16_operand(A), _bf16_operand(Bm)
        elif ok:
            # split-bf16 on the lean kernel: both operands as pre-split images (no conversion in
            # the K loop); weights come from the derived-weight cache, activations are split here
            d.A, d.B = _split_operand(A), _split_operand(Bm)
    parts = None
    if COLSUM_PARTS and d.precision == 3 and form == 0 and (colsum is not None or colsum_alpha is not None) \
            and A.rows >= COLSUM_PARTS_MIN_ROWS:
        # d(bias) / d(PReLU slope) column sums of a precision-3 forward-form launch as PARTIAL rows (one per 64
        # output rows, plain stores) summed by one f2g_colsum per vector, instead of rows / 64 same-address
        # atomics per column from the epilogues (13-24 % of the generator's PReLU-backward data gradients)
        nrows = L.lib.f2g_gemm_colsum_part_rows(C.byref(d))
        if nrows > 0:
            ncol = Bm.rows
            both = colsum is not None and colsum_alpha is not None
            parts = torch.empty(nrows, (2 if both else 1) * ncol, device=out.device, dtype=torch.float32)
            d.E.colsum_part_ld = parts.stride(0)
            if colsum_alpha is not None:
                d.E.colsum_alpha = ptr(parts)
            if colsum is not None:
                d.E.colsum = ptr(parts) + (4 * ncol if both else 0)
    if GEMM_TIMER is not None:
        GEMM_TIMER.launch(d, A, Bm, form, true_k, true_n)
    else:
        call("f2g_gemm", C.byref(d))
    if parts is not None:
        nrows, ncol = parts.shape[0], Bm.rows
        if colsum is not None and colsum_alpha is not None and ptr(colsum) == ptr(colsum_alpha) + 4 * ncol:
            # (the two vectors lie side by side, as the partial matrices do: one reduction for both)
            call("f2g_colsum", ptr(colsum_alpha), ptr(parts), parts.stride(0), None, 0, nrows, 2 * ncol)
            return out
        if colsum_alpha is not None:
            call("f2g_colsum", ptr(colsum_alpha), ptr(parts), parts.stride(0), None, 0, nrows, ncol)
        if colsum is not None:
            call("f2g_colsum", ptr(colsum), ptr(parts) + (4 * ncol if colsum_alpha is not None else 0),
                 parts.stride(0), None, 0, nrows, ncol)
    return out


import os as _os

# column sums of precision-3 epilogues through partial rows instead of atomics (gemm() above)
COLSUM_PARTS = opt("colsum_parts", True)
COLSUM_PARTS_MIN_ROWS = 2048

LEAN_DGRAD = opt("lean_dgrad", True)
LEAN_SPLIT = opt("lean_split", True)
CONV32_SPLIT = opt("conv32_split", True)   # split-bf16 direct MRD convs


def split3(img, src, src_off_bytes: int, ld: int, rows: int, K: int):
    """f2g_split_bf16x3: img (bf16, rows * K * 3) = three-piece image of the (rows, K) fp32 matrix at
    src + src_off_bytes with row stride ld (recorded into an open WeightBatch instead of launched)."""
    if BATCH is not None and rows > 0:
        BATCH.add(3, ptr(img), rows * K * 6, ptr(src) + src_off_bytes, 4 * ((rows - 1) * ld + K), (rows, K, 0, 0),
                  (ld, 0, 0, 0), rows * (K // 4), (img, src))
        return img
    call("f2g_split_bf16x3", ptr(img), ptr(src) + src_off_bytes, ld, rows, K)
    return img


def split3g(img, src, src_off_bytes: int, ld: int, rows: int, K: int):
    """The pieces of f2g_split_bf16x3 in MFMA FRAGMENT order (f2g_operand.split = 4; rows % 32 == 0): an
    F2G_MULTI_SPLIT3G entry -- recorded into an open WeightBatch, else launched as a one-entry f2g_multi table."""
    assert rows % 32 == 0 and K % 32 == 0
    if BATCH is not None and rows > 0:
        BATCH.add(4, ptr(img), rows * K * 6, ptr(src) + src_off_bytes, 4 * ((rows - 1) * ld + K), (rows, K, 0, 0),
                  (ld, 0, 0, 0), rows * (K // 8), (img, src))
        return img
    d = L.MultiDesc()
    e = d.e[0]
    e.out, e.inp, e.kind = ptr(img), ptr(src) + src_off_bytes, 4
    e.blocks = max(1, min(WeightBatch.MAX_BLOCKS, (rows * (K // 8) + 255) // 256))
    e.n[0], e.n[1], e.s[0] = rows, K, ld
    d.n = 1
    call("f2g_multi", C.byref(d))
    return img


def split_bf16(t):
    """Split-bf16 image of a contiguous fp32 tensor (f2g_split_bf16): same shape, same bytes per
    element, every aligned group of four floats = four bf16 high parts + four bf16 remainders."""
    out = torch.empty_like(t)
    call("f2g_split_bf16", ptr(out), ptr(t), t.numel())
    out._f2g_const = getattr(t, "_f2g_const", False)
    return out


_SHARES: dict = {}


class split_sharing:
    """`with split_sharing(t, ...)`: inside the block every split-bf16 operand over one of these
    tensors reuses ONE image per (tensor, operand base) instead of splitting again -- the caller
    asserts that the tensors do not change while the block is active.  The object keeps its images,
    so the same scope can be re-entered later (an activation split for the forward GEMM serves the
    weight gradient of the backward pass).  A no-op outside the split-bf16 mode."""

    def __init__(self, *tensors):
        self.ts = [t for t in tensors if t is not None]
        self.imgs = {}

    def __enter__(self):
        for t in self.ts:
            _SHARES.setdefault(id(t), []).append(self)
        return self

    def __exit__(self, *exc):
        for t in self.ts:
            st = _SHARES.get(id(t))
            if st:
                st.pop()
                if not st:
                    del _SHARES[id(t)]
        return False

    def drop(self):
        self.imgs = {}

    def image(self, t, base: int):
        key = (id(t), base)
        img = self.imgs.get(key)
        if img is None:
            n = t.numel() - (base - ptr(t)) // 4
            img = torch.empty(n, device=t.device, dtype=torch.float32)
            call("f2g_split_bf16", ptr(img), base, n)
            self.imgs[key] = img
        return img


BF16_IMAGES = opt("bf16_images", True)   # precision 2: true bf16 operands


def to_bf16(t):
    """bf16 copy (round to nearest even) of a contiguous fp32 tensor through f2g_to_bf16."""
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    call("f2g_to_bf16", ptr(out), ptr(t), t.numel())
    out._f2g_const = getattr(t, "_f2g_const", False)
    return out


def _bf16_operand(o: Operand) -> Operand:
    """Copy of a lean-eligible operand over a true bf16 tensor (f2g_operand.split = 2)."""
    if o.split == 2:
        return o
    if o.split:
        raise L.F2GError("a split-bf16 image cannot serve as a bf16 operand")
    t = o._keep[0]
    n = Operand()
    C.memmove(C.byref(n), C.byref(o), C.sizeof(Operand))
    if _is_const(t) and t.is_contiguous() and t.numel() % 4 == 0 and (o.base - ptr(t)) % 16 == 0:
        img = derived(t, "bf16", to_bf16)
        n.base = ptr(img) + (o.base - ptr(t)) // 2
    else:
        nseq = o.rows // (o.P0 * o.P1)
        nseg = o.cols // min(o.seglen, o.cols)
        if o.P0 == 1 and o.P1 == 1:
            extent = (o.rows - 1) * o.seq_stride + o.cols
        else:
            extent = (nseq - 1) * o.seq_stride + \
                ((o.P1 - 1) * o.step1 - o.pad1 + nseg - 1) * o.line_stride + \
                ((o.P0 - 1) * o.step0 - o.pad0) * o.unit + min(o.seglen, o.cols)
        extent = (extent + 3) // 4 * 4
        img = torch.empty(extent, device=t.device, dtype=torch.bfloat16)
        call("f2g_to_bf16", ptr(img), o.base, extent)
        n.base = ptr(img)
    n.split = 2
    n._keep = (img,) + tuple(o._keep)
    return n


WGRAD_SPLIT512 = opt("wgrad_split512", True)
X6_MIN_ROWS = opt("x6_min_rows", 1024)
# measured in the step (profiles/r03_x6_step.txt): the six-product kernel beats the fp32 lean kernel from
# reductions of ~2000 on (184 against 131 TFLOP/s at K = 5120, 137 : 121 at 2048) and loses below ~1200
X6_MIN_K = opt("x6_min_k", 2048)


def _x3_operand(o: Operand, frag_major: bool = False) -> Operand:
    """Copy of a plain fp32 matrix operand over its three-piece image (f2g_split_bf16x3): cached for
    weights (and cached re-layouts of weights), written here for activations.  frag_major (cached weights only):
    the image in MFMA fragment order (split = 4, gemm_x6g_kernel)."""
    t = o._keep[0]
    rows, K, ld = o.rows, o.cols, o.seq_stride
    off = o.base - ptr(t)
    if o.P0 == 1 and o.P1 == 1:
        def build(tt):
            img = torch.empty(rows * K * 3, device=tt.device, dtype=torch.bfloat16)
            split3(img, tt, off, ld, rows, K)
            return img
        if frag_major:
            # [rows / 32][K / 32][piece][k step][k half][row % 32][8 bf16]: for one 32-row group, slab, piece and
            # 16-element k step the 64 lanes' 16-byte operands of v_mfma_f32_32x32x16_bf16 are 1 KB contiguous
            assert _is_const(t) and rows % 32 == 0

            def build_g(tt):
                imgf = torch.empty(rows * K * 3, device=tt.device, dtype=torch.bfloat16)
                split3g(imgf, tt, off, ld, rows, K)
                return imgf
            imgf = derived(t, ("x3g", off, rows, K, ld), build_g)
            n = Operand()
            C.memmove(C.byref(n), C.byref(o), C.sizeof(Operand))
            n.base = ptr(imgf)
            n.split = 4
            n._keep = (imgf,) + tuple(o._keep)
            return n
        img = derived(t, ("x3", off, rows, K, ld), build) if _is_const(t) else build(t)
        shift = 0
    else:
        # windows over a contiguous map (halo layouts): the flat image of the whole buffer -- element e
        # at (e / 32) * 192 bytes whatever the row length -- addressed by the same window geometry
        img = getattr(t, "_f2g_x3", None)      # left by the producers' epilogues (gemm(x3_out=True))
        if img is None or getattr(t, "_f2g_x3_bad", False):
            img = x3_flat_image(t)
        elif X3_CHECK:
            ref = x3_flat_image(t)
            torch.cuda.synchronize()
            neq = img.view(torch.int16) != ref.view(torch.int16)
            if bool(neq.any()):
                idx = neq.nonzero()[:, 0]
                e = (idx // 96) * 32 + idx % 32
                Cc = t.shape[1]
                print("X3 MISMATCH", tuple(t.shape), int(neq.sum()), "rows", sorted(set((e // Cc).tolist()))[:12],
                      "P0", o.P0, "seq", o.seq_stride, "cols", sorted(set((e % Cc).tolist()))[:6], flush=True)
        shift = (off // 4 // 32) * 192
    n = Operand()
    C.memmove(C.byref(n), C.byref(o), C.sizeof(Operand))
    n.base = ptr(img) + shift
    n.split = 3
    n._keep = (img,) + tuple(o._keep)
    return n


def x3_flat_image(t):
    """Three-piece bf16 image of a whole contiguous fp32 buffer (numel % 32 == 0)."""
    n = t.numel()
    assert t.is_contiguous() and n % 32 == 0 and t.dtype == torch.float32
    img = torch.empty(n * 3, device=t.device, dtype=torch.bfloat16)
    split3(img, t, 0, 32, n // 32, 32)
    return img


def x3_reserve(t, halo=None):
    """Storage for the three-piece image of the contiguous fp32 buffer `t` that the GEMMs writing `t`
    fill in their epilogues (gemm(x3_out=True)); halo = (S, Hp, C, top, bottom): those rows of every
    sequence are zero in `t` and are zeroed in the image too.  No-op outside the bf16x6 mode."""
    if GEMM_PRECISION != 3 or X6F == 1 or not X3_PRODUCERS or t.numel() % 32 or not t.is_contiguous():
        return t
    img = torch.empty(t.numel() * 3, device=t.device, dtype=torch.bfloat16)
    if halo is not None:
        S, Hp, Cc, top, bot = halo
        if Cc % 32:
            return t
        zero_halo(img.view(torch.float32).view(S * Hp, Cc * 3 // 2), S, Hp, Cc * 3 // 2, top, bot)
    t._f2g_x3_buf = img
    return t


X3_PRODUCERS = opt("x3_producers", True)
X3_CHECK = opt("x3_check", False)       # (debug: compare every producer-written image with a fresh split)
X6_WGRAD = opt("x6_wgrad", True)
# gemm_x6f_kernel (the forward kernel over the fp32 operands, pieces made inside the kernel): 0 never,
# 1 instead of the image kernel everywhere, 2 (default) where it was measured faster than both the image
# kernel and the exact-fp32 lean kernel -- reductions of 640 <= K < X6_MIN_K with >= 512 output columns
# and a tile grid that fills the chip (113920 x 512 x 640: 131 against 114 / 115 TFLOP/s;
# 12032 x 512 x 1536: 109 : 111 : 95; 6016 x 2304 x 768: 103 : 109 : 96 -- profiles/r03_x6_step.txt),
# or very tall GEMMs from 128 columns on (113920 x 128 x 1024: 129 against 98; 24064 x 384 x 1152 loses: 88 : 95)
X6F = opt("x6f", 2)
# (round 5: with the wide epilogue -- x6_epilogue.h -- the in-kernel-split kernel also wins on the generator's
# short reductions: K >= 384, >= 384 columns, >= 180 tiles; same-box step 183.4 -> 178.2 ms, profiles/r05_x6_rules.txt)
# (round 6: from K = 160 -- the second MPD layer, 341376 x 128 x 160, through its "tall GEMM" clause: 62.5 -> 83.5
# TFLOP/s per launch, same-box step 164.24 -> 163.51 ms over three interleaved pairs)
X6F_MIN_K = opt("x6f_min_k", 160)
X6F_MIN_N = opt("x6f_min_n", 384)
# (round 6) tall GEMMs with 32 output columns -- the data gradients that land on the 32-channel MPD map -- on the
# 128 x 32 instance of the in-kernel-split kernel (gemm_x6n_kernel) instead of the generic fp32 kernel
X6N = opt("x6n", True)
X6F_TALL_ROWS = opt("x6f_tall_rows", 50000)     # (round 5: the G-step halves of the MPD layer-3 data gradients too)
X6F_MIN_TILES = opt("x6f_min_tiles", 180)
# long reductions over a PLAIN activation matrix (the generator's K = 2304 GEMMs): below this K the in-kernel
# split instead of an image pass (f2g_split_bf16x3: 10 bytes per element) in front of the image kernel
X6_NOPASS_K = opt("x6_nopass_k", 4096)
X6F_WIMG = opt("x6f_wimg", True)     # in-kernel-split kernel: weights from their cached image
X6G = opt("x6g", True)               # ... in MFMA fragment order, straight into registers (gemm_x6g_kernel)


def _x3_window_ok(o: Operand) -> bool:
    if o.P0 == 1 and o.P1 == 1:
        return True
    t = o._keep[0]
    return (t is not None and t.is_contiguous() and t.numel() % 32 == 0 and t.dtype == torch.float32
            and (o.base - ptr(t)) % 128 == 0 and ptr(t) % 16 == 0)


def operand_formats_ok(Cc: int, Hh: int) -> int:
    """Can the producers of a ConvNeXt block write its GEMM operands directly in the format the
    lean kernels consume?  2: bf16 tensors (precision 2: plain-bf16 inference), 1: split-bf16
    images (precision 1), 0: no (fp32 tensors, converted per GEMM where a lean kernel applies)."""
    if not LEAN_SPLIT or not L.get_option("lean") or not OPERAND_PRODUCERS:
        return 0
    if GEMM_PRECISION == 2 and BF16_IMAGES and Cc % 64 == 0 and Hh % 64 == 0 and Cc > 64:
        return 2
    if GEMM_PRECISION == 1 and Cc % 128 == 0 and Hh % 128 == 0:
        return 1
    return 0


OPERAND_PRODUCERS = opt("operand_producers", True)


def _is_const(t) -> bool:
    """Parameters, views of parameters and cached re-layouts of them: safe to cache images of."""
    owner = t._base if t._base is not None else t
    return isinstance(owner, torch.nn.Parameter) or getattr(owner, "_f2g_const", False)


def _split_operand(o: Operand) -> Operand:
    """Copy of a lean-eligible operand over the split-bf16 image of what it reads: the whole tensor
    (cached until it changes) for weights, the touched range (split here, once) for activations."""
    if o.split:
        return o
    t = o._keep[0]
    n = Operand()
    C.memmove(C.byref(n), C.byref(o), C.sizeof(Operand))
    if _is_const(t) and t.is_contiguous() and t.numel() % 4 == 0 and (o.base - ptr(t)) % 16 == 0:
        img = derived(t, "split", split_bf16)
        n.base = ptr(img) + (o.base - ptr(t))
    elif id(t) in _SHARES and t.is_contiguous() and (o.base - ptr(t)) % 16 == 0 and \
            (t.numel() - (o.base - ptr(t)) // 4) % 4 == 0 and (not o.unbounded or o.base == ptr(t)):
        img = _SHARES[id(t)][-1].image(t, o.base)
        n.base = ptr(img)
    elif o.unbounded:       # read anywhere inside the buffer: the whole tensor
        img = split_bf16(t)
        n.base = ptr(img) + (o.base - ptr(t))
    else:
        nseq = o.rows // (o.P0 * o.P1)
        nseg = o.cols // min(o.seglen, o.cols)
        if o.P0 == 1 and o.P1 == 1:
            extent = (o.rows - 1) * o.seq_stride + o.cols
        else:
            extent = (nseq - 1) * o.seq_stride + \
                ((o.P1 - 1) * o.step1 - o.pad1 + nseg - 1) * o.line_stride + \
                ((o.P0 - 1) * o.step0 - o.pad0) * o.unit + min(o.seglen, o.cols)
        extent = (extent + 3) // 4 * 4
        img = torch.empty(extent, device=t.device, dtype=torch.float32)
        call("f2g_split_bf16", ptr(img), o.base, extent)
        n.base = ptr(img)
    n.split = 1
    n._keep = (img,) + tuple(o._keep)
    return n


CONV32_X6 = opt("conv32_x6", True)    # bf16x6 mode: fp32-class direct MRD convs


def x3_image(t2d):
    """f2g_split_bf16x3 image of a contiguous (rows, K) fp32 matrix, K % 32 == 0 (cached for weights)."""
    def build(t):
        rows, K = t.shape
        img = torch.empty(rows * K * 3, device=t.device, dtype=torch.bfloat16)
        split3(img, t, 0, K, rows, K)
        return img
    return derived(t2d, "x3img", build) if _is_const(t2d) else build(t2d)


def conv32_s2_fwd(x, S: int, H: int, Win: int, Wout: int, w_packed, bias, slope: float, y):
    """Conv2d(32, 32, (3, 9), stride (1, 2), padding (1, 4)) + bias + leaky ReLU on channels-last
    images x (S*H*Win, 32) -> y (S*H*Wout, 32); w_packed = (32, 27*32) as pack_conv_weight gives."""
    d = L.Conv32Desc()
    d.x, d.x_seq, d.x_line = ptr(x), H * Win * 32, Win * 32
    d.S, d.H, d.Win, d.Wout = S, H, Win, Wout
    if GEMM_PRECISION == 1 and CONV32_SPLIT:
        w_packed = derived(w_packed, "split", split_bf16)
        d.precision = 1
        d._keep = w_packed
    elif GEMM_PRECISION == 3 and CONV32_X6:
        # fp32-class products on the bf16 pipe (conv32x6.hip): three-piece image of the packed weights
        w_packed = x3_image(w_packed)
        d.precision = 3
        d._keep = w_packed
    d.w, d.bias, d.lrelu_slope = ptr(w_packed), ptr(bias), slope
    d.y, d.y_seq, d.y_line = ptr(y), H * Wout * 32, Wout * 32
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_conv32_s2_fwd", C.byref(d)),
                        2.0 * S * H * Wout * 32 * 27 * 32, (0, S * H * Wout, 32, 27 * 32))
    else:
        call("f2g_conv32_s2_fwd", C.byref(d))
    return y


CONV33_X6 = opt("conv33_x6", True)    # bf16x6 mode: direct (3, 3) band layer
CONV33_MAX_W = 112     # one image row + its border must fit the kernel's 352 staged pixels: 3 * (W + 2) <= 352


def conv33(x, S: int, H: int, W: int, w_packed, bias, slope: float, y, x_off: int = 0, x_line=None, x_seq=None,
           y_off: int = 0, y_line=None, y_seq=None, form: int = 0, mask=None, colsum=None):
    """Conv2d(32, 32, (3, 3), padding (1, 1)) (+ bias + leaky ReLU) on channels-last images, fp32 class
    (conv32x6.hip: conv33_x6_kernel): x (S, H, W, 32) -> y (S, H, W, 32), both optionally strided slices of
    wider maps (floats); w_packed = (32, 9*32) [co][tap][ci] -- for the data gradient the caller passes the
    gradient map and the flipped / transposed matrix [ci][8 - tap][co] (form = 1: bench.py's FLOP table)."""
    d = L.Conv32Desc()
    d.x = ptr(x) + 4 * x_off
    d.x_line = x_line if x_line is not None else W * 32
    d.x_seq = x_seq if x_seq is not None else H * d.x_line
    d.S, d.H, d.Win, d.Wout = S, H, W, W
    img = x3_image(w_packed)
    d.precision = 3
    d._keep = img
    d.w, d.bias, d.lrelu_slope = ptr(img), ptr(bias), slope
    d.y = ptr(y) + 4 * y_off
    d.y_line = y_line if y_line is not None else W * 32
    d.y_seq = y_seq if y_seq is not None else H * d.y_line
    if mask is not None:      # (data-gradient role) leaky-ReLU backward of the layer below fused into the store
        d.mask_src = ptr(mask[0]) + 4 * mask[1]
        d.mask_slope = float(mask[2])
    d.colsum = ptr(colsum)
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_conv33_fwd", C.byref(d)),
                        2.0 * S * H * W * 32 * 9 * 32, (form, S * H * W, 32, 9 * 32))
    else:
        call("f2g_conv33_fwd", C.byref(d))
    return y


def conv33_wgrad(x, g, S: int, H: int, W: int, gw, g_off: int = 0, g_line=None, g_seq=None):
    """Weight gradient of the (3, 3) band layer, fp32 class (conv32x6.hip: conv33_wgrad6_kernel): x (S, H, W, 32)
    the layer's input, g the gradient of its pre-activation -- a band's slice of the concatenated gradient map
    (g_off floats in, line / sequence strides in floats) --, gw (32, 9 * 32) [co][tap][ci] +=."""
    d = L.Conv32Desc()
    d.x, d.x_line, d.x_seq = ptr(x), W * 32, H * W * 32
    d.S, d.H, d.Win, d.Wout = S, H, W, W
    d.precision = 3
    d.y = ptr(g) + 4 * g_off
    d.y_line = g_line if g_line is not None else W * 32
    d.y_seq = g_seq if g_seq is not None else H * d.y_line
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_conv33_wgrad", C.byref(d), ptr(gw)),
                        2.0 * S * H * W * 32 * 9 * 32, (2, 32, 9 * 32, S * H * W))
    else:
        call("f2g_conv33_wgrad", C.byref(d), ptr(gw))
    return gw


def conv32_s2_dgrad(g, S: int, H: int, Win: int, Wout: int, wT, gx, g_seq=None, g_line=None,
                    g_off: int = 0, mask=None, fm=None, colsum=None):
    """Data gradient of Conv2d(32, 32, (3, 9), stride (1, 2), padding (1, 4)): g (S, H, Wout, 32)
    (optionally strided / offset) -> gx (S*H*Win, 32); wT = (27, 32, 32) tiles [tap][ci][co]."""
    d = L.Conv32Desc()
    d.x = ptr(g) + 4 * g_off
    d.x_line = g_line if g_line is not None else Wout * 32
    d.x_seq = g_seq if g_seq is not None else H * d.x_line
    d.S, d.H, d.Win, d.Wout = S, H, Win, Wout
    if GEMM_PRECISION == 1 and CONV32_SPLIT:
        wT = derived(wT, "split", split_bf16)
        d.precision = 1
        d._keep = wT
    elif GEMM_PRECISION == 3 and CONV32_X6:
        wT = x3_image(wT.view(27 * 32, 32))      # rows (tap, ci), reduction over co
        d.precision = 3
        d._keep = wT
    d.w, d.bias, d.lrelu_slope = ptr(wT), None, 0.0
    d.y, d.y_seq, d.y_line = ptr(gx), H * Win * 32, Win * 32
    if mask is not None:      # leaky-ReLU backward of the layer below fused into the store
        d.mask_src = ptr(mask[0]) + 4 * mask[1]
        d.mask_slope = float(mask[2])
        if fm is not None:
            d.fm_ref, d.fm_w, d.fm_wdev = ptr(fm[0]) + 4 * fm[1], float(fm[2]), ptr(fm[3])
    d.colsum = ptr(colsum)
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_conv32_s2_dgrad", C.byref(d)),
                        2.0 * S * H * Wout * 32 * 27 * 32, (1, S * H * Win, 32, 27 * 32 // 2))
    else:
        call("f2g_conv32_s2_dgrad", C.byref(d))
    return gx


def conv32_s2_wgrad(x, g, S: int, H: int, Win: int, Wout: int, gw):
    """gw (32, 27*32) += weight gradient of Conv2d(32, 32, (3, 9), stride (1, 2), padding (1, 4));
    x (S*H*Win, 32) layer input, g (S*H*Wout, 32) gradient of the pre-activation."""
    d = L.Conv32Desc()
    d.x, d.x_seq, d.x_line = ptr(x), H * Win * 32, Win * 32
    d.S, d.H, d.Win, d.Wout = S, H, Win, Wout
    d.w, d.bias, d.lrelu_slope = None, None, 0.0
    d.y, d.y_seq, d.y_line = ptr(g), H * Wout * 32, Wout * 32
    if GEMM_PRECISION == 1 and CONV32_SPLIT:
        d.precision = 1
    elif GEMM_PRECISION == 3 and CONV32_X6:
        d.precision = 3
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_conv32_s2_wgrad", C.byref(d), ptr(gw)),
                        2.0 * S * H * Wout * 32 * 27 * 32, (2, 32, 27 * 32, S * H * Wout))
    else:
        call("f2g_conv32_s2_wgrad", C.byref(d), ptr(gw))
    return gw


def _conv2ch_desc(x, x_seq, x_line, x_off, S, H, W):
    d = L.Conv2chDesc()
    d.x = None if x is None else ptr(x) + 4 * x_off
    d.x_seq, d.x_line = x_seq, x_line
    d.S, d.H, d.W = S, H, W
    return d


def _timed_hbm(name, nbytes, *args):
    """The thin first / last layers of the discriminators (2 -> 32, 32 -> 1, 1 -> 32, 1024 -> 1 channels):
    HBM-bound by design (one side of the layer is a 32- / 1024-channel map, the
    reduction is 5 ... 288 long), so bench.py prices them against the HBM roofline by their algorithmic
    bytes -- the wide map read or written once, the thin side once -- not against the MFMA peak."""
    if GEMM_TIMER is not None:
        GEMM_TIMER.time_hbm(lambda: call(name, *args), float(nbytes), name[4:])
    else:
        call(name, *args)


def conv2ch_fwd(x, x_seq, x_line, x_off, S, H, W, w_packed, bias, slope, y):
    """Conv2d(2, 32, (3, 9), padding (1, 4)) + bias + leaky ReLU on a band (S, H, W, 2) of the
    interleaved spectrogram -> y (S*H*W, 32); w_packed (32, 54) as pack_conv_weight gives."""
    d = _conv2ch_desc(x, x_seq, x_line, x_off, S, H, W)
    d.w, d.bias, d.lrelu_slope, d.y = ptr(w_packed), ptr(bias), slope, ptr(y)
    _timed_hbm("f2g_conv2ch_fwd", 4.0 * S * H * W * (32 + 2), C.byref(d))      # write 32 channels, read 2
    return y


def conv2ch_wgrad(x, x_seq, x_line, x_off, S, H, W, g, gw):
    """gw (32, 54) += weight gradient of that layer; g = (S*H*W, 32) pre-activation gradient."""
    d = _conv2ch_desc(x, x_seq, x_line, x_off, S, H, W)
    d.y, d.gw = ptr(g), ptr(gw)
    _timed_hbm("f2g_conv2ch_wgrad", 4.0 * S * H * W * (32 + 2), C.byref(d))
    return gw


def conv2ch_dgrad(g, S, H, W, wt, gx, gx_seq, gx_line, gx_off):
    """gx band (S, H, W, 2) (strided, overwritten) = data gradient; wt = (27, 2, 32)."""
    d = _conv2ch_desc(None, 0, 0, 0, S, H, W)
    d.y, d.wt = ptr(g), ptr(wt)
    d.gx = ptr(gx) + 4 * gx_off
    d.gx_seq, d.gx_line = gx_seq, gx_line
    _timed_hbm("f2g_conv2ch_dgrad", 4.0 * S * H * W * (32 + 2), C.byref(d))
    return gx


def convpost_fwd(x, S, H, W, w9, bias, y):
    """Conv2d(32, 1, (3, 3), padding 1): x (S*H*W, 32) -> y (S*H*W); w9 = (9, 32) tap-major."""
    d = _conv2ch_desc(x, 0, 0, 0, S, H, W)
    d.w, d.bias, d.y = ptr(w9), ptr(bias), ptr(y)
    _timed_hbm("f2g_convpost_fwd", 4.0 * S * H * W * (32 + 1), C.byref(d))
    return y


def convpost_wgrad(x, S, H, W, g, gw):
    d = _conv2ch_desc(x, 0, 0, 0, S, H, W)
    d.y, d.gw = ptr(g), ptr(gw)
    _timed_hbm("f2g_convpost_wgrad", 4.0 * S * H * W * (32 + 1), C.byref(d))
    return gw


def convpost_dgrad(g, S, H, W, w9, gx, g_off=0):
    d = _conv2ch_desc(None, 0, 0, 0, S, H, W)
    d.y, d.w, d.gx = ptr(g) + 4 * g_off, ptr(w9), ptr(gx)
    _timed_hbm("f2g_convpost_dgrad", 4.0 * S * H * W * (32 + 1), C.byref(d))
    return gx


class GemmTimer:
    """bench.py instrumentation: HIP events around every f2g_gemm launch on the launch stream and
    the launch's algorithmic FLOPs (2 * M * N * K of the implicit GEMM it represents)."""

    def __init__(self):
        self.records = []
        self.shapes = []
        self.paths = []       # kernel family per record
        self.hbm = []         # (start, end, algorithmic bytes, kernel name): HBM-bound kernels
        self.epis = []        # epilogue kind per record (report only)
        self.report_by_epilogue = False

    def time_hbm(self, fn, nbytes: float, name: str):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.hbm.append((s, e, nbytes, name))

    def by_path(self):
        """{family: (launches, flops, seconds)} -- after a device synchronise."""
        out = {}
        for (s, e, f), pth in zip(self.records, self.paths):
            a = out.setdefault(pth, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += f
            a[2] += s.elapsed_time(e) * 1e-3
        return out

    def algorithmic_bytes(self, path: str) -> float:
        """Sum over a family's launches of the bytes the implicit GEMM must move once: both operands
        read, the result written (4 bytes per element; im2col windows counted as the rows they gather)."""
        return float(sum(4.0 * (m * k + n * k + m * n)
                         for (f, m, n, k), pth in zip(self.shapes, self.paths) if pth == path))

    def hbm_summary(self):
        """{kernel: (launches, algorithmic bytes, seconds)} -- after a device synchronise."""
        out = {}
        for s, e, nb, name in self.hbm:
            a = out.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += nb
            a[2] += s.elapsed_time(e) * 1e-3
        return out

    def launch(self, d, A, Bm, form, true_k=None, true_n=None):
        # the implicit GEMM (M, N, K) this launch stands for, zero padding of the operands not counted
        nn = Bm.rows if form == 0 else Bm.cols
        mm, kk = (A.cols, A.rows) if form == 2 else (A.rows, A.cols)
        kk = min(kk, true_k) if true_k else kk
        nn = min(nn, true_n) if true_n else nn
        flops = 2.0 * mm * nn * kk
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        call("f2g_gemm", C.byref(d))
        e.record()
        self.records.append((s, e, flops))
        self.paths.append(("generic", "lean", "lean-streamk", "narrow", "x6", "x6-thin")[L.lib.f2g_gemm_last_path()])
        if self.paths[-1] == "x6" and form == 0 and d.A.split == 3 and (L.lib.f2g_gemm_x6_ok(C.byref(d)) & 2):
            self.x6_tap = getattr(self, "x6_tap", 0) + 1      # (launches on the tap-walking instance)
        self.shapes.append((form, mm, nn, kk))
        e_ = d.E
        self.epis.append("+".join(k for k, on in (
            ("prelu2", bool(e_.prelu_out)), ("prelu", bool(e_.prelu_slope) and not e_.prelu_out),
            ("dprelu", bool(e_.aux)), ("res", bool(e_.res)), ("x3", bool(e_.x3_out)), ("map", e_.P0o > 0),
            ("mask", bool(e_.mask_src)), ("lrelu", e_.lrelu_slope != 0.0), ("bf16", bool(e_.c_bf16)),
            ("atomic", bool(e_.atomic))) if on) or "plain")

    def time(self, fn, flops: float, shape, path: str = "direct-conv"):
        """Any other launch that stands in for an f2g_gemm (the direct band-conv kernel)."""
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((s, e, flops))
        self.paths.append(path)
        self.shapes.append(shape)
        self.epis.append("")

    def report(self, top: int = 25) -> str:
        """Per-shape table (form, M, N, K, kernel family): calls, total ms, TFLOP/s -- after a synchronise."""
        agg = {}
        by_epi = self.report_by_epilogue                        # (split the rows by epilogue kind as well)
        for (s, e, f), shp, pth, ep in zip(self.records, self.shapes, self.paths, self.epis):
            a = agg.setdefault(tuple(shp) + ((pth + " " + ep) if by_epi else pth,), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        out = ["form      M      N      K  calls  total_ms  TFLOP/s  family"]
        for key, (c, ms, f) in rows:
            form, m, n, k = key[:4]
            out.append(f"{form:4d} {m:6d} {n:6d} {k:6d} {c:6d} {ms:9.3f} {f / (ms * 1e-3) / 1e12:8.1f}  {key[-1]}")
        return "\n".join(out)

    def summary(self):
        """(launches, total flops, total seconds) -- call after a device synchronise."""
        t = sum(s.elapsed_time(e) for s, e, _ in self.records) * 1e-3
        return len(self.records), sum(f for _, _, f in self.records), t


GEMM_TIMER = None

# GEMM arithmetic: 0 = exact fp32 MFMA, 1 = split-bf16 (hi/lo, 3 bf16 MFMAs per product, fp32
# accumulate).  Selected with F2G_GEMM=fp32|bf16x3 (default fp32) or set_gemm_precision().
import os as _os

GEMM_PRECISION = {"bf16x3": 1, "split": 1, "1": 1, "bf16": 2, "2": 2, "bf16x6": 3, "3": 3}.get(
    _os.environ.get("F2G_GEMM", "fp32").lower(), 0)


def set_gemm_precision(name: str) -> None:
    global GEMM_PRECISION
    if name not in ("fp32", "bf16x3", "bf16", "bf16x6"):
        raise ValueError("precision must be 'fp32', 'bf16x3', 'bf16x6' or 'bf16'")
    # "bf16": plain bf16 operands, fp32 accumulate -- inference throughput mode (BASELINE config 2),
    # not a parity mode (waveform error ~1e-3 RMS instead of <= 1e-4)
    # "bf16x6": fp32-class -- three bf16 pieces per operand, six MFMAs per product (error ~2^-23 per
    # product, like fp32 rounding itself) for the plain-matrix forward / data-gradient GEMMs (the
    # generator's 1x1 convolutions and linears); everything else stays on the exact fp32 MFMA
    GEMM_PRECISION = {"fp32": 0, "bf16x3": 1, "bf16": 2, "bf16x6": 3}[name]


# ------------------------------------------------------------------ concurrent launch lanes
# The three Fourier branches (and the 5 + 3 sub-discriminators) are independent kernel sequences
# whose mid-size GEMMs each leave part of the last wave of CUs idle (282 tiles for 512 resident
# slots ...).  Launching them on separate HIP streams lets the hardware fill those holes with
# another lane's blocks.  F2G_STREAMS=0 (or an active GemmTimer: per-kernel roofline timing wants
# kernels in isolation) runs the lanes one after the other on the caller's stream.
CONCURRENT = _os.environ.get("F2G_STREAMS", "1") != "0"
_SIDE_STREAMS: dict = {}
# F2G_OPTS="lane_cap_mpd=3,lane_cap_mrd=2": at most that many streams behind the lanes of a pool (lane i ->
# stream i % cap; a measurement aid: how much concurrency the step wants)
_LANE_CAP = {pool: opt("lane_cap_" + pool, 0) for pool in ("branch", "disc", "mpd", "mrd", "mel", "condpath",
                                                           "timepath", "band")}
_LANE_CAP = {k: v for k, v in _LANE_CAP.items() if v > 0}


def _side_streams(device, n: int, pool_name: str):
    # (a HIGH-priority stream for the critical 768-channel branch was measured: no effect under graph
    # replay, eager bf16 inference 8.6 -> 13.0 ms, stage-2 step 236.5 -> 242 ms -- not used)
    idx = torch.device(device).index
    key = (idx if idx is not None else torch.cuda.current_device(), pool_name)
    pool = _SIDE_STREAMS.setdefault(key, [])
    cap = max(1, min(n, _LANE_CAP.get(pool_name, n)))
    while len(pool) < cap:
        pool.append(torch.cuda.Stream(device=device))
    return [pool[i % cap] for i in range(n)]


class Lanes:
    """Fork n launch lanes off the current stream, join them back.

        lanes = Lanes(dev, 3)
        for i in range(3):
            with lanes.lane(i):
                ...launches...
        lanes.join()

    Lane i of a pool always maps to the same persistent side stream, so tensors a lane allocates
    (and keeps for its backward) stay in that stream's allocator pool and are only ever touched by
    that lane or, after join(), by the caller's stream.  Lanes nest (a lane may fork its own
    lanes) as long as every nesting level names its own `pool`."""

    def __init__(self, device, n: int, pool: str = "lanes"):
        self.main = torch.cuda.current_stream(device)
        self.on = CONCURRENT and n > 1 and GEMM_TIMER is None
        self.streams = _side_streams(device, n, pool) if self.on else [self.main] * n
        if self.on:
            fork = torch.cuda.Event()
            fork.record(self.main)
            for s in self.streams:
                s.wait_event(fork)
        self._chain = None

    def lane(self, i: int):
        return torch.cuda.stream(self.streams[i])

    def chain_enter(self):
        """Serialise a read-modify-write of a buffer shared by the lanes (call inside a lane)."""
        if self.on and self._chain is not None:
            torch.cuda.current_stream().wait_event(self._chain)

    def chain_leave(self):
        if self.on:
            self._chain = torch.cuda.Event()
            self._chain.record(torch.cuda.current_stream())

    def join(self):
        if not self.on:
            return
        for s in self.streams:
            ev = torch.cuda.Event()
            ev.record(s)
            self.main.wait_event(ev)


def split_for(reduction_rows: int, out_tiles: int, two_per_cu: bool = False) -> int:
    """Split-K factor for weight-gradient GEMMs (form 2; 128 x 128 output tiles, atomic accumulation).

    two_per_cu: the launch is known to take the exact-fp32 K-major lean kernel (two blocks per CU, >= 4096
    rows per block): rounds of 512 instead of 256.  Measured on the MPD's 1024-channel layers
    (tools/micro/leanw_win_probe.py, 5 sequence heights x 9 factors): 320 tiles x 4 = 1280 blocks = 2.5 rounds
    runs at 105-132 TFLOP/s depending on the height, x 8 = 2560 = 5 rounds at 128-134 for all of them; 160
    tiles x 8 at 106-128, x 3 = 480 co-resident blocks at 122-127.

    Measured (tools/wgrad_probe.py, 14 shapes x 15 factors): what matters is how the blocks =
    tiles * s fill ROUNDS OF 256 (one block per CU) -- a count just above a multiple of 256 is the
    worst case (27 tiles: s = 9 -> 243 blocks 214 us, s = 10 -> 270 blocks 319 us), just below the
    best; among well-filled counts, "all blocks co-resident as two per CU" (<= 512) wins when it
    fills its rounds within 5 % of the best candidate, otherwise the fewest splits among the best
    fills (fewer atomics); a chunk keeps >= 16 K slabs."""
    slabs = max(1, reduction_rows // 32)
    smax = max(1, min(512, slabs // 16))
    R = 256
    if two_per_cu:
        R = 512
        smax = max(1, min(smax, reduction_rows // 4096))
    cands = set(range(1, min(smax, 64) + 1))
    for m in range(1, 9):
        s = (R * m) // out_tiles
        if 1 <= s <= smax:
            cands.add(s)

    def fill(s):
        b = out_tiles * s
        return b / (((b + R - 1) // R) * R)

    best = max(fill(s) for s in cands)
    pair = [s for s in cands if out_tiles * s <= 512 and fill(s) >= best - 0.05]
    if pair:
        return max(pair)
    return min(s for s in cands if fill(s) >= best - 0.02)


def wgrad(dY, M: int, ldy: int, X: Operand, g_out, ldg: Optional[int] = None, out_offset: int = 0,
          dy_lrelu_src=None, slope: float = 0.0):
    """g_out[m, c] += sum_r dY[r, m] * X[r, c]   (atomic split-K; g_out must be initialised)."""
    rows = X.rows
    A = mat(dY, rows, M, ldy, lrelu_src=dy_lrelu_src, slope=slope)
    gemm(A, X, g_out, form=2, ldc=ldg, atomic=True, split_k=-1, out_offset=out_offset)   # -1: split_for, in gemm()


# ------------------------------------------------------------------ fused pointwise MLP (bf16)
FUSED_MLP = opt("fused_mlp", True)


def fused_mlp_applies(Cc: int, Hh: int) -> bool:
    """pwconv1 -> PReLU -> pwconv2 (+ residual) as ONE kernel with the hidden activation on chip
    (csrc/fusedmlp.hip): plain-bf16 inference, C in {384, 512, 768}."""
    return FUSED_MLP and GEMM_PRECISION == 2 and bool(L.lib.f2g_fused_mlp_ok(Cc, Hh))


def mlp_pack(w1, w2):
    """The block's two weight matrices (H, C[, 1]) and (C, H[, 1]) as the fused kernel's bf16
    fragment stream, cached until one of them changes.  Pass the PARAMETERS themselves (the cache
    is keyed on the tensor objects: a fresh reshape() view per call would never hit)."""
    def build(ts):
        a, b = ts
        Hh, Cc = a.shape[0], a.shape[1]
        out = torch.empty(2 * Cc * Hh, device=a.device, dtype=torch.bfloat16)
        call("f2g_mlp_pack", ptr(out), ptr(a), Cc, ptr(b), Hh, Cc, Hh)
        return out
    return derived_multi([w1, w2], "mlp_pack", build)


def fused_mlp(z, wp, b1, alpha, b2, res, gamma, out, rows: int, Cc: int, Hh: int, parts: int = 0):
    """out (rows, C) fp32 = W2 . PReLU(W1 . z + b1) + b2 + gamma * res; z bf16 (rows, C).
    parts: 0 = library's choice, n = cut the hidden dimension between n blocks per row tile."""
    d = L.FusedMlpDesc()
    d.parts = parts
    d.z, d.ldz, d.wp = ptr(z), z.stride(0), ptr(wp)
    d.b1, d.alpha, d.b2 = ptr(b1), ptr(alpha), ptr(b2)
    d.res, d.ldres, d.gamma = ptr(res), (res.stride(0) if res is not None else 0), ptr(gamma)
    d.out, d.ldo = ptr(out), out.stride(0)
    d.rows, d.C, d.H = rows, Cc, Hh
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_fused_mlp", C.byref(d)), 4.0 * rows * Cc * Hh,
                        (0, rows, Cc, 2 * Hh), path="fused-mlp")
    else:
        call("f2g_fused_mlp", C.byref(d))
    return out


def fused_block(x, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, wp, b1, alpha, b2, gamma, out,
                Hh: int, cproj=None, ldcp=0, Fc=0, up=1, cp_off=0, te=None, ldte=0, te_off=0,
                parts: int = 0):
    """out = x-block of ConvNeXtBlock.forward (modules.py:473-495) in one launch (plain-bf16
    inference): dwconv7 + BiasNorm + cond + time scale -> z (bf16, LDS only) -> fused MLP ->
    + gamma * x.  Arguments as dwnorm_fwd / fused_mlp."""
    f = _dw_desc(x, x.stride(0), None, 0, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, cproj, ldcp,
                 Fc, up, cp_off, te, ldte, te_off, None)
    d = L.FusedMlpDesc()
    d.parts = parts
    d.wp, d.b1, d.alpha, d.b2 = ptr(wp), ptr(b1), ptr(alpha), ptr(b2)
    d.res, d.ldres, d.gamma = ptr(x), x.stride(0), ptr(gamma)
    d.out, d.ldo = ptr(out), out.stride(0)
    d.rows, d.C, d.H = B * F, Cc, Hh
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_fused_block", C.byref(f), C.byref(d)), 4.0 * B * F * Cc * Hh,
                        (0, B * F, Cc, 2 * Hh), path="fused-mlp")
    else:
        call("f2g_fused_block", C.byref(f), C.byref(d))
    return out


def fused_block_multi(entries):
    """One launch for several independent blocks (the same layer of a decoder's Fourier branches).
    entries: a list of dicts with the keyword arguments of fused_block (x, B, F, Cc, K, lens, w_dw,
    b_dw, beta, log_scale, wp, b1, alpha, b2, gamma, out, Hh, cproj, ldcp, Fc, up, cp_off, te, ldte,
    te_off).  Same results as len(entries) fused_block calls."""
    n = len(entries)
    fs = (DwnormFwd * n)()
    ds = (L.FusedMlpDesc * n)()
    flops = 0.0
    for i, e in enumerate(entries):
        x, out = e["x"], e["out"]
        fs[i] = _dw_desc(x, x.stride(0), None, 0, e["B"], e["F"], e["Cc"], e["K"], e.get("lens"), e["w_dw"],
                         e.get("b_dw"), e["beta"], e["log_scale"], e.get("cproj"), e.get("ldcp", 0),
                         e.get("Fc", 0), e.get("up", 1), e.get("cp_off", 0), e.get("te"), e.get("ldte", 0),
                         e.get("te_off", 0), None)
        d = ds[i]
        d.parts = 0
        d.wp, d.b1, d.alpha, d.b2 = ptr(e["wp"]), ptr(e.get("b1")), ptr(e["alpha"]), ptr(e.get("b2"))
        d.res, d.ldres, d.gamma = ptr(x), x.stride(0), ptr(e.get("gamma"))
        d.out, d.ldo = ptr(out), out.stride(0)
        d.rows, d.C, d.H = e["B"] * e["F"], e["Cc"], e["Hh"]
        flops += 4.0 * d.rows * d.C * d.H
    if GEMM_TIMER is not None:
        GEMM_TIMER.time(lambda: call("f2g_fused_block_multi", fs, ds, n), flops,
                        (0, sum(e["B"] * e["F"] for e in entries), 0, 0), path="fused-mlp")
    else:
        call("f2g_fused_block_multi", fs, ds, n)
    return [e["out"] for e in entries]


FUSED_BLOCK = opt("fused_block", True)
# all Fourier branches' blocks of a layer in one launch (bf16 inference; 0: one launch per branch and lane)
FUSED_MULTI = opt("fused_multi", True)


# ------------------------------------------------------------------ fused block kernels
def _dw_desc(x, ldx, z, ldz, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, cproj, ldcp, Fc, up,
             cp_off, te, ldte, te_off, rstd) -> DwnormFwd:
    f = DwnormFwd()
    f.x, f.ldx = ptr(x), ldx
    f.z, f.ldz = ptr(z), ldz
    f.B, f.F, f.C, f.K = B, F, Cc, K
    f.lens = ptr(lens)
    f.w_dw, f.b_dw, f.beta, f.log_scale = ptr(w_dw), ptr(b_dw), ptr(beta), ptr(log_scale)
    f.cproj = None if cproj is None else ptr(cproj) + 4 * cp_off
    f.ldcp, f.Fc, f.up = ldcp, Fc, up
    f.te = None if te is None else ptr(te) + 4 * te_off
    f.ldte = ldte
    f.rstd = ptr(rstd)
    return f


def dwnorm_fwd(x, z, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, cproj=None, ldcp=0, Fc=0,
               up=1, cp_off=0, te=None, ldte=0, te_off=0, rstd=None, z_format: int = 0):
    """z_format: 0 fp32, 1 split-bf16 image in an fp32-typed tensor, 2 z is a bf16 tensor."""
    f = _dw_desc(x, x.stride(0), z, z.stride(0), B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale,
                 cproj, ldcp, Fc, up, cp_off, te, ldte, te_off, rstd)
    f.z_format = z_format
    if GEMM_TIMER is not None:
        # algorithmic bytes: read x, write z, read the condition row once per `up` frames
        nb = 4.0 * B * F * Cc * (2.0 + (1.0 / up if cproj is not None else 0.0))
        GEMM_TIMER.time_hbm(lambda: call("f2g_dwnorm_fwd", C.byref(f)), nb, "dwnorm_fwd")
    else:
        call("f2g_dwnorm_fwd", C.byref(f))
    return z


def dwnorm_bwd(x, gz, du, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, cproj=None, ldcp=0,
               Fc=0, up=1, cp_off=0, te=None, ldte=0, te_off=0, g_cproj=None, g_te=None,
               g_beta=None, g_log_scale=None, g_cproj_store=False):
    """g_cproj_store: the caller zero-filled g_cproj's columns [cp_off, cp_off + Cc) and nothing else
    adds to them -- the kernel then stores the condition gradient instead of read-modify-writing it."""
    d = DwnormBwd()
    d.f = _dw_desc(x, x.stride(0), None, 0, B, F, Cc, K, lens, w_dw, b_dw, beta, log_scale, cproj,
                   ldcp, Fc, up, cp_off, te, ldte, te_off, None)
    d.gz, d.ldgz = ptr(gz), gz.stride(0)
    d.du, d.lddu = ptr(du), du.stride(0)
    d.g_cproj = None if g_cproj is None else ptr(g_cproj) + 4 * cp_off
    d.g_cproj_store = 1 if (g_cproj_store and g_cproj is not None) else 0
    d.g_te = None if g_te is None else ptr(g_te) + 4 * te_off
    d.g_beta, d.g_log_scale = ptr(g_beta), ptr(g_log_scale)
    ws = torch.empty(L.lib.f2g_dwnorm_bwd_workspace(B, F, Cc, up if cproj is not None else 1),
                     device=x.device, dtype=torch.float32)
    d.partials = ptr(ws)
    if GEMM_TIMER is not None:
        # algorithmic bytes: read x and gz, write du; with a condition input its rows are read and its
        # gradient rows written once per `up` frames (the gradient is consumed by the condition path's
        # backward GEMM: required traffic, as the forward kernel's condition read is)
        nb = 4.0 * B * F * Cc * (3.0 + ((1.0 / up if cproj is not None else 0.0)
                                      + (1.0 / up if g_cproj is not None else 0.0)))
        GEMM_TIMER.time_hbm(lambda: call("f2g_dwnorm_bwd", C.byref(d)), nb, "dwnorm_bwd")
    else:
        call("f2g_dwnorm_bwd", C.byref(d))
    return du


def dwconv_bwd(du, x, gx, B, F, Cc, K, lens, w_dw, gres=None, gamma=None, g_w=None, g_b=None,
               g_gamma=None):
    d = DwconvBwd()
    d.du, d.lddu = ptr(du), du.stride(0)
    d.x, d.ldx = ptr(x), x.stride(0)
    d.gx, d.ldgx = ptr(gx), gx.stride(0)
    d.B, d.F, d.C, d.K = B, F, Cc, K
    d.lens = ptr(lens)
    d.w_dw = ptr(w_dw)
    d.gres = ptr(gres)
    d.ldgres = gres.stride(0) if gres is not None else 0
    d.gamma = ptr(gamma)
    d.g_w, d.g_b, d.g_gamma = ptr(g_w), ptr(g_b), ptr(g_gamma)
    ws = torch.empty(L.lib.f2g_dwconv_bwd_workspace(B, F, Cc, K), device=x.device,
                     dtype=torch.float32)
    d.partials = ptr(ws)
    if GEMM_TIMER is not None:   # read du, x, gres; write gx
        GEMM_TIMER.time_hbm(lambda: call("f2g_dwconv_bwd", C.byref(d)), 16.0 * B * F * Cc, "dwconv_bwd")
    else:
        call("f2g_dwconv_bwd", C.byref(d))
    return gx


def biasnorm_fwd(x, y, rows, Cc, beta, log_scale):
    call("f2g_biasnorm_fwd", ptr(x), x.stride(0), ptr(y), y.stride(0), rows, Cc, ptr(beta),
         ptr(log_scale))
    return y


def biasnorm_bwd(x, gy, gx, rows, Cc, beta, log_scale, g_beta, g_log_scale):
    call("f2g_biasnorm_bwd", ptr(x), x.stride(0), ptr(gy), gy.stride(0), ptr(gx), gx.stride(0),
         rows, Cc, ptr(beta), ptr(log_scale), ptr(g_beta), ptr(g_log_scale))
    return gx


# ------------------------------------------------------------------ signal / elementwise
def istft_ola(frames, out, B, F, n_fft, hop, T, window, wbranch, wscale, accumulate):
    call("f2g_istft_ola", ptr(frames), frames.stride(0), ptr(out), B, F, n_fft, hop, T,
         ptr(window), ptr(wbranch), float(wscale), 1 if accumulate else 0)


def istft_ola_multi(entries, out, B, T, wscale, accumulate=False):
    """The overlap-adds of up to four branches in one launch; entries: (frames, F, n_fft, hop, window,
    wbranch or None) in the order their sums are to be added."""
    d = L.OlaMultiDesc()
    d.n = len(entries)
    for i, (frames, F, n_fft, hop, window, wbranch) in enumerate(entries):
        d.frames[i], d.ldf[i] = ptr(frames), frames.stride(0)
        d.F[i], d.n_fft[i], d.hop[i] = F, n_fft, hop
        d.window[i], d.wbranch[i] = ptr(window), ptr(wbranch)
    call("f2g_istft_ola_multi", C.byref(d), ptr(out), B, T, float(wscale), 1 if accumulate else 0)
    return out


def istft_ola_bwd(gout, gframes, B, F, n_fft, hop, T, window, wbranch, wscale):
    call("f2g_istft_ola_bwd", ptr(gout), ptr(gframes), gframes.stride(0), B, F, n_fft, hop, T,
         ptr(window), ptr(wbranch), float(wscale))


def frames_fold(gframes, gx, B, F, n_fft, hop, T, accumulate):
    call("f2g_frames_fold", ptr(gframes), gframes.stride(0), ptr(gx), B, F, n_fft, hop, T,
         1 if accumulate else 0)


def axpby_rows(y, x0, x1, ca=None, cb=None, sa=1.0, sb=1.0):
    rows = x0.shape[0]
    cols = x0.numel() // rows
    call("f2g_axpby_rows", ptr(y), ptr(x0), ptr(x1), ptr(ca), ptr(cb), float(sa), float(sb), rows,
         cols)
    return y


def clamp(y, x, lo, hi):
    call("f2g_clamp", ptr(y), ptr(x), float(lo), float(hi), x.numel())
    return y


def silu(y, x):
    call("f2g_silu", ptr(y), ptr(x), x.numel())
    return y


def silu_bwd(gx, gy, x):
    call("f2g_silu_bwd", ptr(gx), ptr(gy), ptr(x), x.numel())
    return gx


def time_embedding(out, t, dim, scale=1000.0):
    call("f2g_time_embedding", ptr(out), ptr(t), t.numel(), dim, float(scale))
    return out


def mask_rows(x, B, F, Cc, lens):
    call("f2g_mask_rows", ptr(x), x.stride(0), B, F, Cc, ptr(lens))
    return x


def colsum(out, a, rows, cols, b=None, out_offset=0):
    call("f2g_colsum", ptr(out) + 4 * out_offset, ptr(a), a.stride(0), ptr(b),
         b.stride(0) if b is not None else 0, rows, cols)
    return out


def bct_to_rows(out, x, B, Cc, F):
    call("f2g_bct_to_rows", ptr(out), out.stride(0), ptr(x), B, Cc, F)
    return out


def rows_to_bct(out, x, B, Cc, F):
    call("f2g_rows_to_bct", ptr(out), ptr(x), x.stride(0), B, Cc, F)
    return out


def permute4(out, x, dims, strides, in_offset=0):
    n0, n1, n2, n3 = dims
    s0, s1, s2, s3 = strides
    if BATCH is not None and n0 * n1 * n2 * n3 > 0:
        lo = sum(min(0, (n - 1) * st) for n, st in zip(dims, strides))
        hi = sum(max(0, (n - 1) * st) for n, st in zip(dims, strides)) + 1
        BATCH.add(1, ptr(out), 4 * n0 * n1 * n2 * n3, ptr(x) + 4 * (in_offset + lo), 4 * (hi - lo), dims, strides,
                  n0 * n1 * n2 * n3, (out, x), in_ptr=ptr(x) + 4 * in_offset)
        return out
    call("f2g_permute4", ptr(out), ptr(x) + 4 * in_offset, n0, n1, n2, n3, s0, s1, s2, s3)
    return out


def copy3(out, so0, so1, x, si0, si1, n0, n1, n2, accumulate=False, out_offset=0, in_offset=0):
    if BATCH is not None and n0 * n1 * n2 > 0 and min(so0, so1, si0, si1) >= 0:
        BATCH.add(2, ptr(out) + 4 * out_offset, 4 * ((n0 - 1) * so0 + (n1 - 1) * so1 + n2),
                  ptr(x) + 4 * in_offset, 4 * ((n0 - 1) * si0 + (n1 - 1) * si1 + n2),
                  (n0, n1, n2, 1 if accumulate else 0), (so0, so1, si0, si1), n0 * n1 * n2, (out, x),
                  reads_out=bool(accumulate))
        return out
    call("f2g_copy3", ptr(out) + 4 * out_offset, so0, so1, ptr(x) + 4 * in_offset, si0, si1, n0,
         n1, n2, 1 if accumulate else 0)
    return out


def spec_power(out, packed, rows, nb, power):
    call("f2g_spec_power", ptr(out), out.stride(0), ptr(packed), packed.stride(0), rows, nb, power)
    return out


def spec_power_bwd(gpacked, gout, packed, rows, nb, power):
    call("f2g_spec_power_bwd", ptr(gpacked), gpacked.stride(0), ptr(gout), gout.stride(0),
         ptr(packed), rows, nb, power)
    return gpacked


def fm_spec_loss(loss, g_err, s_err, s_gt, B, F, nf, lens, eps, power, lo, hi, inv_denom):
    call("f2g_fm_spec_loss", ptr(loss), ptr(g_err), ptr(s_err), ptr(s_gt), B, F, nf, ptr(lens),
         float(eps), float(power), float(lo), float(hi), float(inv_denom))


def masked_mse(loss, g_err, pred, ref, B, T, lens, inv_denom):
    call("f2g_masked_mse", ptr(loss), ptr(g_err), ptr(pred), ptr(ref), B, T, ptr(lens),
         float(inv_denom))


def l1_loss(loss, gb, a, b, rows, cols, ld, w, clip=0.0, wdev=None, loss_offset=0, off=0):
    """a, b, gb are views of the same layout starting `off` floats into their tensors."""
    call("f2g_l1_loss", None if loss is None else ptr(loss) + 4 * loss_offset,
         None if gb is None else ptr(gb) + 4 * off, ptr(a) + 4 * off, ptr(b) + 4 * off, rows, cols,
         ld, float(w), float(clip), ptr(wdev))


def l1_loss_ab(loss, gb, a, a_off, b, b_off, rows, cols, ld, w, clip=0.0, wdev=None,
               loss_offset=0):
    """Same with independent offsets for a and b (gb follows b)."""
    call("f2g_l1_loss", None if loss is None else ptr(loss) + 4 * loss_offset,
         None if gb is None else ptr(gb) + 4 * b_off, ptr(a) + 4 * a_off, ptr(b) + 4 * b_off, rows,
         cols, ld, float(w), float(clip), ptr(wdev))


def hinge_loss(loss, gs, s, n, sgn, w, wdev=None, loss_offset=0, s_off=0):
    call("f2g_hinge_loss", None if loss is None else ptr(loss) + 4 * loss_offset,
         None if gs is None else ptr(gs) + 4 * s_off, ptr(s) + 4 * s_off, n, float(sgn), float(w),
         ptr(wdev))


def peaknorm_fwd(y, stats, x, rows, T):
    call("f2g_peaknorm_fwd", ptr(y), ptr(stats), ptr(x), rows, T)


def peaknorm_bwd(gx, gy, x, stats, rows, T):
    call("f2g_peaknorm_bwd", ptr(gx), ptr(gy), ptr(x), ptr(stats), rows, T)


def lrelu_bwd(g, y_act, f_real, w, slope, rows, cols, ld, wdev=None, g_off=0, y_off=0, r_off=0):
    call("f2g_lrelu_bwd", ptr(g) + 4 * g_off, ptr(y_act) + 4 * y_off,
         None if f_real is None else ptr(f_real) + 4 * r_off, float(w), ptr(wdev), float(slope),
         rows, cols, ld)


def lrelu_bwd_colsum(g, y_act, f_real, w, slope, rows, Cc, ld, colsum, wdev=None, g_off=0, y_off=0,
                     r_off=0):
    """In-place leaky-ReLU backward of a (rows, Cc) map + column sums of the result (bias grad)."""
    args = ("f2g_lrelu_bwd_colsum", ptr(g) + 4 * g_off, ptr(y_act) + 4 * y_off,
            None if f_real is None else ptr(f_real) + 4 * r_off, float(w), ptr(wdev), float(slope),
            rows, Cc, ld, ptr(colsum))
    if GEMM_TIMER is not None:   # read g, y (+ f_real); write g
        GEMM_TIMER.time_hbm(lambda: call(*args), (12.0 + (4.0 if f_real is not None else 0.0)) * rows * Cc,
                            "lrelu_bwd_colsum")
    else:
        call(*args)


def zeros_many(shapes, device):
    """Several zero-initialised tensors carved from ONE zeroed allocation (one fill launch instead
    of one per tensor); every piece starts on a 64-float boundary."""
    sizes = []
    for sh in shapes:
        n = 1
        for v in sh:
            n *= v
        sizes.append(n)
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    flat = zeros(max(total, 1), device=device)
    return [flat[o:o + n].view(*sh) for o, n, sh in zip(offs, sizes, shapes)]


# ------------------------------------------------------------------ first MPD layer (1 -> 32 channels)
MPD0_DIRECT = opt("mpd0_direct", True)


def _mpd0_desc(x, S, H, Hout, halo, w=None, bias=None, slope=0.0, y=None, x_off=0, y_off=0):
    d = L.Mpd0Desc()
    d.x = None if x is None else ptr(x) + 4 * x_off
    d.S, d.H, d.Hout, d.halo = S, H, Hout, halo
    d.w, d.bias, d.slope = ptr(w), ptr(bias), slope
    d.y = ptr(y) + 4 * y_off
    return d


def mpd0_fwd(x, S, H, Hout, halo, w5, bias, slope, y):
    """y (halo layout (S, Hout + 2*halo, 32)) = lrelu(conv(x (S, H); w5 (32, 5), stride 3, pad 2) + bias)."""
    d = _mpd0_desc(x, S, H, Hout, halo, w5, bias, slope, y)
    _timed_hbm("f2g_mpd0_fwd", 4.0 * S * (Hout * 32 + H), C.byref(d))
    return y


def mpd0_wgrad(x, S, H, Hout, halo, g, gw):
    """gw (32, 5) += weight gradient of the first MPD layer; g = gradient map (halo layout)."""
    d = _mpd0_desc(x, S, H, Hout, halo, y=g)
    _timed_hbm("f2g_mpd0_wgrad", 4.0 * S * (Hout * 32 + H), C.byref(d), ptr(gw))
    return gw


def mpd0_dgrad(g, S, H, Hout, halo, w5, gx, g_off=0):
    """gx (S*H) = data gradient of the first MPD layer from the gradient map g (halo layout)."""
    d = _mpd0_desc(None, S, H, Hout, halo, w5, None, 0.0, g, y_off=g_off)
    _timed_hbm("f2g_mpd0_dgrad", 4.0 * S * (Hout * 32 + H), C.byref(d), ptr(gx))
    return gx


def _mpdpost_desc(y, S, H, halo, w3=None, bias=None, out=None, g=None, g_off=0, y_off=0):
    d = L.MpdPostDesc()
    d.y = ptr(y) + 4 * y_off
    d.S, d.H, d.halo = S, H, halo
    d.w, d.bias, d.out = ptr(w3), ptr(bias), ptr(out)
    d.g = None if g is None else ptr(g) + 4 * g_off
    return d


def mpdpost_fwd(y, S, H, halo, w3, bias, out):
    """out (S*H) = conv_post of an MPD sub-discriminator over the 1024-channel map y (halo layout)."""
    _timed_hbm("f2g_mpdpost_fwd", 4.0 * S * H * (1024 + 1), C.byref(_mpdpost_desc(y, S, H, halo, w3, bias, out)))
    return out


def mpdpost_dgrad(g, S, H, halo, w3, gy, g_off=0, mask=None, fm=None, colsum=None):
    """gy (halo layout (S, H + 2*halo, 1024), halo rows pre-zeroed) = data gradient of conv_post; optionally with
    the leaky-ReLU backward of the layer it lands on (mask = (activation map, float offset, slope); fm =
    (reference map, float offset, weight, device scalar)), that layer's bias gradient (colsum) and -- when gy
    carries x3_reserve storage -- the three-piece image of the result (bf16x6 mode)."""
    d = _mpdpost_desc(gy, S, H, halo, w3, None, None, g, g_off)
    nbytes = 4.0 * S * H * (1024 + 1)
    if mask is not None:
        d.mask_src, d.mask_slope = ptr(mask[0]) + 4 * mask[1], float(mask[2])
        nbytes += 4.0 * S * H * 1024
        if fm is not None:
            d.fm_ref, d.fm_w, d.fm_wdev = ptr(fm[0]) + 4 * fm[1], float(fm[2]), ptr(fm[3])
            nbytes += 4.0 * S * H * 1024
    d.colsum = ptr(colsum)
    buf = getattr(gy, "_f2g_x3_buf", None)
    if buf is not None and not getattr(gy, "_f2g_x3_bad", False):
        d.x3_out = ptr(buf)
        gy._f2g_x3 = buf
        nbytes += 6.0 * S * H * 1024
    _timed_hbm("f2g_mpdpost_dgrad", nbytes, C.byref(d))
    return gy


def mpdpost_wgrad(y, S, H, halo, g, gw):
    """gw (3*1024, tap-major) += weight gradient of conv_post."""
    d = _mpdpost_desc(y, S, H, halo, None, None, None, g)
    _timed_hbm("f2g_mpdpost_wgrad", 4.0 * S * H * (1024 + 1), C.byref(d), ptr(gw))
    return gw


# ------------------------------------------------------------------ LDS-butterfly FFT (n_fft >= 1024)
USE_FFT = opt("fft", True)
# transforms below FFT_MIN stay on the DFT GEMM (the 32- / 64-point mel-reconstruction scales: a
# handful of MFMAs per frame).  F2G_FFT_MIN=1024 restores round 2's split (LDS FFT for n_fft >= 1024 only).
FFT_MIN = opt("fft_min", 128)
_FFT_TABLES = {}


def fft_applies(n_fft: int) -> bool:
    return USE_FFT and FFT_MIN <= n_fft <= 4096 and (n_fft & (n_fft - 1)) == 0


def _fft_tables(n_fft: int, device):
    """(hann window, twiddles (cos, -sin)(2 pi j / N)), computed in float64, cached per device."""
    import math
    key = (n_fft, str(device))
    if key not in _FFT_TABLES:
        j = torch.arange(n_fft // 2, dtype=torch.float64)
        ang = 2.0 * math.pi * j / n_fft
        tw = torch.stack([torch.cos(ang), -torch.sin(ang)], dim=1).float().contiguous().to(device)
        win = torch.hann_window(n_fft, dtype=torch.float32).to(device)
        _FFT_TABLES[key] = (win, tw)
    return _FFT_TABLES[key]


def _spec_flags(spec, interleaved: bool) -> int:
    return (1 if interleaved else 0) | (2 if spec.dtype == torch.bfloat16 else 0)


FFT_REFLECT = opt("fft_reflect", True)   # center / reflect padding inside the FFT kernel


def stft_fft(x, n_fft: int, hop: int, F: int, spec, interleaved: bool = False, zero_pad: bool = False):
    """spec (B*F, ld) = STFT of x (B, T) (center, reflect, periodic hann) through the LDS FFT.  spec
    may be a bf16 tensor (planar rows); zero_pad: the kernel also zeroes columns [n_fft + 2, ld)."""
    B, T = x.shape
    pad = n_fft // 2
    win, tw = _fft_tables(n_fft, x.device)
    d = L.FftDesc()
    if FFT_REFLECT and T > pad and x.stride(1) == 1:
        # the kernel reflects at the signal's ends itself: no padded copy, no extra launch
        xp = x
        d.x, d.x_stride, d.reflect_T = ptr(x), x.stride(0), T
    else:
        Tp = pad4(T + 2 * pad)
        xp = torch.empty(B, Tp, device=x.device, dtype=torch.float32)
        call("f2g_reflect_pad", ptr(xp), ptr(x), B, T, pad, Tp)
        d.x, d.x_stride = ptr(xp), Tp
    d.hop, d.n_fft, d.F, d.rows = hop, n_fft, F, B * F
    d.window, d.twiddle = ptr(win), ptr(tw)
    d.spec, d.ld_spec, d.interleaved = ptr(spec), spec.stride(0), _spec_flags(spec, interleaved)
    d.spec_cols = spec.shape[1] if zero_pad else 0
    call("f2g_fft_frames", C.byref(d), 0)
    d._keep = (xp, win, tw)
    return spec


def stft_fft_adjoint(gspec, n_fft: int, F: int, gframes, interleaved: bool = False):
    """gframes (rows, n_fft) = gradient of the (windowed) frames given the gradient of the stored
    bins gspec (rows, ld): the adjoint of stft_fft; frames_fold scatters it back onto the signal."""
    rows = gframes.shape[0]
    win, tw = _fft_tables(n_fft, gspec.device)
    d = L.FftDesc()
    d.hop, d.n_fft, d.F, d.rows = 0, n_fft, F, rows
    d.window, d.twiddle = ptr(win), ptr(tw)
    d.spec, d.ld_spec, d.interleaved = ptr(gspec), gspec.stride(0), 1 if interleaved else 0
    d.frames, d.ld_frames = ptr(gframes), gframes.stride(0)
    call("f2g_fft_frames", C.byref(d), 1)
    return gframes


def istft_fft(spec, n_fft: int, F: int, frames):
    """frames (rows, n_fft) = the iSTFT's windowed inverse real transform of the planar half spectra
    spec (rows, ld) (f2g_fft_frames mode 2: 1/N, doubled interior bins, Im of DC / Nyquist ignored,
    hann window): what istft_ola overlap-adds (the inverse-DFT GEMM's output)."""
    win, tw = _fft_tables(n_fft, spec.device)
    d = L.FftDesc()
    d.hop, d.n_fft, d.F, d.rows = 0, n_fft, F, frames.shape[0]
    d.window, d.twiddle = ptr(win), ptr(tw)
    d.spec, d.ld_spec, d.interleaved = ptr(spec), spec.stride(0), _spec_flags(spec, False)
    d.frames, d.ld_frames = ptr(frames), frames.stride(0)
    call("f2g_fft_frames", C.byref(d), 2)
    return frames


def istft_fft_adjoint(gframes, n_fft: int, F: int, gspec, zero_pad: bool = False):
    """gspec (rows, ld) columns [0, n_fft + 2) = gradient of the bins given the gradient of the
    windowed inverse-transformed frames (mode 3); columns beyond are zeroed (zero_pad) or left alone."""
    win, tw = _fft_tables(n_fft, gframes.device)
    d = L.FftDesc()
    d.hop, d.n_fft, d.F, d.rows = 0, n_fft, F, gframes.shape[0]
    d.window, d.twiddle = ptr(win), ptr(tw)
    d.spec, d.ld_spec, d.interleaved = ptr(gspec), gspec.stride(0), 0
    d.spec_cols = gspec.shape[1] if zero_pad else 0
    d.frames, d.ld_frames = ptr(gframes), gframes.stride(0)
    call("f2g_fft_frames", C.byref(d), 3)
    return gspec


def stft_frames(x, n_fft: int, hop: int, F: int) -> Operand:
    """Framing operand of an STFT (center=True, reflect): rows = (item, frame m), cols = the n_fft
    samples [m*hop, m*hop + n_fft) of the reflect-padded signal.  The padding is materialised once
    (B x (T + n_fft) floats) so that the frames are plain overlapping rows -- no bounds tests or
    mirroring in the GEMM's K loop (the lean kernel applies)."""
    B, T = x.shape
    pad = n_fft // 2
    Tp = pad4(T + 2 * pad)
    if hop % 4 or n_fft % 32:
        return win1d(x, B, T, 1, F, hop, pad, n_fft, reflect=True)
    xp = torch.empty(B, Tp, device=x.device, dtype=torch.float32)
    call("f2g_reflect_pad", ptr(xp), ptr(x), B, T, pad, Tp)
    return win1d(xp, B, Tp, 1, F, hop, 0, n_fft)


def period_fold(out, x, B, T, p, H):
    call("f2g_period_fold", ptr(out), ptr(x), B, T, p, H)


def period_fold_bwd(gx, gout, B, T, p, H, accumulate):
    call("f2g_period_fold_bwd", ptr(gx), ptr(gout), B, T, p, H, 1 if accumulate else 0)


def limit_grad(g, p, lo, hi):
    call("f2g_limit_grad", ptr(g), ptr(p), float(lo), float(hi), g.numel())
    return g


def log_clip_(x, clip=1e-7):
    call("f2g_log_clip", ptr(x), x.numel(), float(clip))
    return x
