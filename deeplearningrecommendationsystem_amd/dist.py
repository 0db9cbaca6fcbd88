"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).  Nothing like it exists in the reference
(single device).

* ``GradBucket`` -- data-parallel part: the batch is split across ranks; parameters that
  are replicated (dense layers, ml-100k-sized tables) get their gradients averaged with
  ONE all-reduce over a flat bucket (low MBs for every model of the zoo).
* ``ShardedEmbedding`` -- model-parallel part for the 1e6..1e7-row tables: rows are dealt
  round-robin to the ranks (``owner = row % world``, ``local = row // world``: uniform load
  whatever the id skew).  A lookup is: an ``ExchangePlan`` of the id tensor (bucket ids by owner with a HIP
  kernel, all_to_all of the per-rank counts, one host read, all_to_all of the int32 local rows; cached per id
  tensor and shared by every table indexed with it) -> local row gather (HIP kernel) -> all_to_all of the
  rows back -> un-permute into batch order (HIP kernel).  all-to-all is the right xGMI
  primitive: a full mesh of point-to-point links, all seven used concurrently, where a ring
  all-reduce would be bound by a single link.  Backward mirrors it and ends in a scatter-add
  into the local shard's dense gradient.

The compute steps are behind a small backend interface so that the exchange protocol can
be exercised on CPU with gloo (tests inject a numpy backend); the default backend is the
HIP one and needs the MI355X library like everything else in this package.
"""
from __future__ import annotations

import os
import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


_DEBUG = os.environ.get("CTR_DIST_DEBUG", "0") == "1"


class GradBucket:
    """averages the gradients of ``params`` over the ranks with as few collectives as possible.

    The zoo's backward passes carve every dense ``.grad`` out of ONE zero-filled flat buffer
    (``ops.zero_grads``), so after ``backward()`` all gradients usually share a single
    storage: that storage is all-reduced in place -- one collective, no packing copies.
    Gradients that live elsewhere (plain torch modules, CPU tests) take the classic
    pack -> all-reduce -> unpack path through a flat bucket (fixed order)."""

    MAX_STORAGES = 4  # more distinct storages than this: packing is cheaper than the collectives

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        # row shards (ShardedEmbedding.weight) are different rows on every rank: never averaged
        self.params: List[torch.nn.Parameter] = [p for p in params
                                                 if p.requires_grad and not getattr(p, "ctr_row_shard", False)]
        self._flat = None       # packing bucket, allocated on first use
        self._views = None
        # (key of the gradient tensors, flat views of their storages) of the previous step.  The ONLY place a
        # storage view outlives a call: kept when the gradients are static tensors (hipGraph replay) and
        # small; a step with fresh gradient buffers replaces it, so at most one generation is ever pinned
        self._plan = None

    def _pack_bucket(self):
        if self._flat is None:
            total = sum(p.numel() for p in self.params)
            ref = self.params[0]
            self._flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
            self._views, off = [], 0
            for p in self.params:
                self._views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._flat, self._views

    def _shared_storages(self):
        """flat views over the storages holding every gradient, or None when packing is better"""
        seen = {}
        for p in self.params:
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous():
                return None
            st = g.untyped_storage()
            key = st.data_ptr()
            if key not in seen:
                if len(seen) == self.MAX_STORAGES or st.nbytes() % 4:
                    return None
                # a fresh view per call: caching it here would pin every step's gradient buffer for good
                seen[key] = torch.empty(0, dtype=torch.float32, device=g.device).set_(st, 0, (st.nbytes() // 4,))
        return list(seen.values())

    @staticmethod
    def _mean(flat: torch.Tensor, group) -> None:
        world = dist.get_world_size(group)
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)  # RCCL averages in the collective
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / world)

    def all_reduce_mean(self, group=None) -> None:
        """average gradients over the ranks (no-op for a single process)"""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        # hipGraph replay rewrites the same static .grad tensors every step: the storages found once
        # stay valid as long as the first and last gradients are still the same tensors
        first, last = self.params[0].grad, self.params[-1].grad
        key = (id(first), id(last), first.data_ptr() if first is not None else 0)
        if self._plan is not None and self._plan[0] == key:
            shared = self._plan[1]
        else:
            shared = self._shared_storages()
            # (the cached views keep their storage alive: only worth it, and only harmless, when small)
            small = shared is not None and sum(f.numel() for f in shared) <= (1 << 24)
            self._plan = (key, shared) if small else None
        if shared is not None:
            for flat in shared:
                self._mean(flat, group)
            return
        flat, views = self._pack_bucket()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(views, grads)
        self._mean(flat, group)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    @property
    def flat(self) -> torch.Tensor:
        return self._pack_bucket()[0]


# ---------------------------------------------------------------------------
# row-sharded embedding
# ---------------------------------------------------------------------------
class HipShardBackend:
    """the product backend: every step is a libctrhip launch on torch's current stream"""

    @staticmethod
    def bucket(ids: torch.Tensor, world: int, vocab: int):
        """-> (counts (world + 1,) int64 on device -- the last slot counts ids outside [0, vocab) --,
        send (n,) int32 local rows in bucket order, perm (n,), inv (n,))"""
        from . import _lib
        _lib.require_device(ids)
        n = ids.numel()
        dev = ids.device
        counts = torch.empty(world + 1, dtype=torch.int64, device=dev)
        cursor = torch.empty(256 * (world + 1), dtype=torch.int64, device=dev)      # CTR_SHARD_SCRATCH_INT64 (ctrhip.h)
        send = torch.empty(n, dtype=torch.int32, device=dev)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        inv = torch.empty(n, dtype=torch.int64, device=dev)
        rc = _lib.load().ctr_shard_bucket(_lib.ptr(ids) if n else None, n, world, vocab, counts.data_ptr(),
                                          cursor.data_ptr(), _lib.ptr(send) if n else None, _lib.ptr(perm) if n else None,
                                          _lib.ptr(inv) if n else None, _lib.stream_ptr())
        _lib.check(rc, "ctr_shard_bucket")
        return counts, send, perm, inv

    @staticmethod
    def bucket_padded(ids: torch.Tensor, world: int, vocab: int, cap: int):
        """capacity-bounded layout (bucket w = slots [w*cap, (w+1)*cap)) -> (state (4,) int64 on device =
        [overflow, bad ids, n, -n], send (world*cap,) int32 with -1 in unused slots, perm (n,), inv (world*cap,))"""
        from . import _lib
        _lib.require_device(ids)
        n, dev = ids.numel(), ids.device
        state = torch.empty(4, dtype=torch.int64, device=dev)
        cursor = torch.empty(256 * (world + 1), dtype=torch.int64, device=dev)      # CTR_SHARD_SCRATCH_INT64 (ctrhip.h)
        send = torch.empty(world * cap, dtype=torch.int32, device=dev)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        inv = torch.empty(world * cap, dtype=torch.int64, device=dev)
        rc = _lib.load().ctr_shard_bucket_padded(_lib.ptr(ids) if n else None, n, world, vocab, cap, cursor.data_ptr(),
                                                 send.data_ptr(), _lib.ptr(perm) if n else None, inv.data_ptr(),
                                                 state.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "ctr_shard_bucket_padded")
        return state, send, perm, inv

    @staticmethod
    def recv_rows(recv: torch.Tensor, local_rows: int):
        """received wire slots -> (rows (slots,) int64 safe to gather / scatter, valid (slots, 1) float 0/1,
        mark (slots,) int64 with -1 for unused slots)"""
        from . import _lib
        slots, dev = recv.numel(), recv.device
        rows = torch.empty(slots, dtype=torch.int64, device=dev)
        valid = torch.empty((slots, 1), dtype=torch.float32, device=dev)
        mark = torch.empty(slots, dtype=torch.int64, device=dev)
        rc = _lib.load().ctr_shard_recv_rows(_lib.ptr(recv) if slots else None, slots, local_rows, _lib.ptr(rows) if slots else None,
                                             _lib.ptr(valid) if slots else None, _lib.ptr(mark) if slots else None,
                                             _lib.stream_ptr())
        _lib.check(rc, "ctr_shard_recv_rows")
        return rows, valid, mark

    @staticmethod
    def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """out[i] = table[idx[i]] (bit-exact copy)"""
        from . import ops
        from ._lib import FIELD_ID_I64
        n, dim = idx.numel(), table.shape[1]
        out = torch.empty((n, dim), dtype=table.dtype, device=table.device)
        if n:
            ops.embed_fwd([ops.FieldSpec(FIELD_ID_I64, dim, 0, table=table, idx=idx)], None, n, out)
        return out

    @staticmethod
    def scatter_add_rows(grad: torch.Tensor, idx: torch.Tensor, rows: torch.Tensor) -> None:
        """grad[idx[i]] += rows[i]"""
        from . import ops
        from ._lib import FIELD_ID_I64
        n, dim = idx.numel(), grad.shape[1]
        if n:
            spec = ops.FieldSpec(FIELD_ID_I64, dim, 0, table=grad, idx=idx)
            ops.embed_bwd([spec], None, n, rows, {id(grad): grad})

    @staticmethod
    def zero_rows(grad: torch.Tensor, idx: torch.Tensor) -> None:
        """grad[idx[i]] = 0"""
        from . import _lib
        if idx.numel():
            rc = _lib.load().ctr_rows_zero(grad.data_ptr(), grad.stride(0), grad.shape[0], grad.shape[1], _lib.ptr(idx),
                                           idx.numel(), _lib.stream_ptr())
            _lib.check(rc, "ctr_rows_zero")


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo has no device all-to-all: rehearsals with several ranks on one GPU go through the host"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _InFlight:
    """an all-to-all that has been issued (``async_op``: RCCL runs it on the group's own stream) and whose result is
    not needed yet: the caller keeps launching -- the next table's gather, the previous table's interaction math --
    and ``wait()`` only orders torch's current stream behind the collective (no host block with RCCL)."""

    __slots__ = ("work", "out", "keep", "device")

    def __init__(self, work, out, keep, device):
        self.work, self.out, self.keep, self.device = work, out, keep, device

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        if self.device is not None:
            self.out, self.device = self.out.to(self.device), None
        return self.out


def _exchange_start(send: torch.Tensor, send_counts: Optional[List[int]], recv_counts: Optional[List[int]], group) -> _InFlight:
    """all_to_all_single with per-rank row counts (``None``: equal splits, the capacity-bounded layout); trailing dims
    are kept"""
    rows = send.shape[0] if recv_counts is None else sum(recv_counts)
    shape = (rows,) + tuple(send.shape[1:])
    staged = _host_staged(send, group)
    src = send.contiguous().cpu() if staged else send.contiguous()
    out = torch.empty(shape, dtype=send.dtype, device=src.device)
    work = dist.all_to_all_single(out, src, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group,
                                  async_op=True)
    return _InFlight(work, out, src, send.device if staged else None)


def _exchange(send, send_counts, recv_counts, group) -> torch.Tensor:
    return _exchange_start(send, send_counts, recv_counts, group).wait()


class ExchangePlan:
    """everything about ONE id tensor's lookup that does not depend on table contents: who owns which id and which of
    this rank's rows the peers asked for.  Two layouts:

    * exact -- buckets back to back: bucket kernel, count exchange, ONE host read of 2*world + 1 integers (the only
      sync), the id exchange (int32 local rows on the wire);
    * capacity-bounded (``ShardedEmbedding(capacity_factor=...)``) -- every bucket has ``cap`` slots, so the splits are
      host-known: a FRESH id tensor's lookup is enqueued without any host read; whether a bucket overflowed (or an id
      was outside the table) is MAX-all-reduced on the device, copied to pinned memory and looked at only after the
      whole lookup has been enqueued (``verify``), while the GPU works through it.  An overflow makes every rank
      rebuild the plan in the exact layout -- slower, never wrong.

    A plan is reused by every table looked up with those ids (FFM: two field-aware tables per id column) and by
    every later step that passes the same, unmodified tensor (the reference trains full-batch on the same tensors
    every epoch, scripts/din.py:93-96): those steps run no bucketing and no id exchange, only the two row exchanges."""

    __slots__ = ("send_counts", "recv_counts", "perm", "inv", "recv_ids", "n", "cap", "valid", "mark", "_state", "_event",
                 "verified", "_verdict")

    def verify(self) -> Optional[str]:
        """None: the plan is good.  'overflow': every rank must rebuild it in the exact layout.  Raises IndexError (on
        every rank) for ids outside the table.  Host-blocks only until the 32-byte state copy, issued
        before the lookup's gathers and exchanges, has landed."""
        if self.verified:
            return None
        if self._state is not None:
            if self._event is not None:
                self._event.synchronize()
            over, bad, n_max, n_negmax = (int(v) for v in self._state.tolist())
            self._state = self._event = None
            self._verdict = ("bad", bad) if bad else "overflow" if over else None
            self.verified = self._verdict is None
        if isinstance(self._verdict, tuple):        # (every time the cached plan is used again, on every rank)
            raise IndexError(f"index out of range in self ({self._verdict[1]} ids outside the table on a rank of this group)")
        return self._verdict


# id(key tensor) -> (weak reference to it, {sub-key: (ExchangePlan, weak references of the other key tensors)}).  Keyed
# by identity with a liveness check (a WeakKeyDictionary would compare tensors with ==); the entry goes away with the
# tensor.
_PLANS: dict = {}


def _plans_of(anchor: torch.Tensor, create: bool):
    entry = _PLANS.get(id(anchor))
    if entry is not None and entry[0]() is anchor:
        return entry[1]
    if not create:
        return None
    key = id(anchor)
    plans: dict = {}
    _PLANS[key] = (weakref.ref(anchor, lambda _r, key=key: _PLANS.pop(key, None)), plans)
    return plans


def _split_key(ids: torch.Tensor, key):
    """-> (anchor tensor, other key tensors, hashable part).  ``key`` = (anchor, extra): ``extra`` is a tensor, a tuple
    of tensors and hashables, or a hashable.  EVERY tensor of a key is identified by ``id()`` + ``_version`` AND held
    by a weak reference that must still point at it when the plan is found again: CPython hands the id of a freed
    temporary to the next tensor of the same size, and a fresh tensor's version is 0 again."""
    if key is None:
        return ids, (), None
    anchor, extra = key
    parts = extra if isinstance(extra, tuple) else (extra,)
    others = tuple(p for p in parts if isinstance(p, torch.Tensor))
    plain = tuple(("t", id(p), p._version) if isinstance(p, torch.Tensor) else p for p in parts)
    return anchor, others, plain


def _exact_plan(flat: torch.Tensor, module: "ShardedEmbedding") -> ExchangePlan:
    be, group, world = module.backend, module.group, module.world
    counts, send_ids, perm, inv = be.bucket(flat, world, module.num_embeddings)
    staged = _host_staged(counts, group)
    # per peer: (ids I send it, ids of mine outside the table) -- the second column lets EVERY rank learn that some rank
    # saw a bad id from the one count exchange, so that all of them raise together instead of one raising and the
    # others blocking in the next collective until the RCCL timeout
    mine = torch.stack([counts[:world], counts[world:world + 1].expand(world)], dim=1).contiguous()
    mine = mine.cpu() if staged else mine
    theirs = torch.empty_like(mine)
    dist.all_to_all_single(theirs, mine, group=group)
    both = torch.cat([mine[:, 0], theirs.reshape(-1)]).tolist()                # the one host read of the lookup
    send_counts = [int(c) for c in both[:world]]
    recv_counts = [int(c) for c in both[world::2]]
    bad = sum(int(c) for c in both[world + 1::2])                              # over all ranks (mine is in the diagonal)
    if bad:
        raise IndexError(f"index out of range in self ({bad} ids outside [0, {module.num_embeddings}) on the ranks "
                         f"of this group)")
    plan = ExchangePlan()
    plan.send_counts, plan.recv_counts, plan.perm, plan.inv, plan.n = send_counts, recv_counts, perm, inv, flat.numel()
    plan.recv_ids = _exchange(send_ids, send_counts, recv_counts, group).long()   # local rows the peers want
    plan.cap, plan.valid, plan.mark, plan._state, plan._event, plan.verified = None, None, plan.recv_ids, None, None, True
    plan._verdict = None
    return plan


def _padded_plan(flat: torch.Tensor, module: "ShardedEmbedding") -> ExchangePlan:
    """no host read: see ExchangePlan.  The wire size must be the same on every rank WITHOUT talking about it: it is
    derived from ``module.capacity_ids`` when that is set (any n up to it is fine, on any rank), else from this call's
    n -- which then has to be the same on every rank (the equal split of a global batch)."""
    be, group, world = module.backend, module.group, module.world
    n = flat.numel()
    if module.capacity_ids is not None and n > module.capacity_ids:
        raise ValueError(f"lookup of {n} ids on a ShardedEmbedding built for capacity_ids={module.capacity_ids}")
    cap = module.capacity(module.capacity_ids if module.capacity_ids is not None else n)
    state, send, perm, inv = be.bucket_padded(flat, world, module.num_embeddings, cap)
    staged = _host_staged(state, group)
    state = state.cpu() if staged else state
    dist.all_reduce(state, op=dist.ReduceOp.MAX, group=group)
    plan = ExchangePlan()
    if state.is_cuda:
        host = torch.empty(4, dtype=torch.int64, pin_memory=True)
        host.copy_(state, non_blocking=True)
        plan._state, plan._event = host, torch.cuda.Event()
        plan._event.record()
    else:
        plan._state, plan._event = state, None
    recv = _exchange(send, None, None, group)                                   # world*cap int32, -1 = unused slot
    plan.recv_ids, plan.valid, plan.mark = be.recv_rows(recv, module.weight.shape[0])
    plan.send_counts = plan.recv_counts = None
    plan.perm, plan.inv, plan.n, plan.cap, plan.verified, plan._verdict = perm, inv, n, cap, False, None
    return plan


def exchange_plan(ids: torch.Tensor, module: "ShardedEmbedding", key=None, exact: bool = False) -> ExchangePlan:
    """``key`` = (tensor whose identity and version stand for the ids, extra) when ``ids`` itself is a temporary (FFM:
    ``x[:, 0].long()`` of the feature matrix ``x``; DIN: ``cat(hist, target)`` keyed by ``(hist, target)``); default:
    the id tensor itself.  ``exact``: build (and cache) the exact layout even if the module is capacity-bounded -- the
    fallback after an overflow.

    The cache decides per rank, so every rank must present the same sequence of (cached / fresh) lookups: a rank that
    hits while a peer misses would skip the collectives the peer enters.  Identical id-tensor lifetimes on all ranks
    (the usual SPMD loop) guarantee it; ``CTR_DIST_DEBUG=1`` all-reduces the decision and raises on a mismatch.
    ``_version`` does not see writes through raw pointers (this library's kernels): an id tensor filled by a kernel
    must be passed as a new tensor object, or the plan dropped with ``forget_plans(tensor)``."""
    anchor, others, plain = _split_key(ids, key)
    sub = (anchor._version, plain, id(module.group), module.world, module.num_embeddings, tuple(ids.shape),
           module.capacity_factor, module.capacity_ids, module.weight.shape[0])
    plans = _plans_of(anchor, False)
    hit = None
    if plans is not None and sub in plans:
        plan, refs = plans[sub]
        if all(r() is t for r, t in zip(refs, others)):
            hit = plan if (not exact or plan.cap is None) else None     # (exact: only an exact plan will do)
        else:
            del plans[sub]      # an id() recycled by a new tensor: the old plan is for other ids
    group = module.group
    if _DEBUG:
        mine = torch.tensor([1 if hit is not None else 0, 1], dtype=torch.int64)
        dist.all_reduce(mine, group=group)
        if int(mine[0]) not in (0, int(mine[1])):
            raise RuntimeError("exchange_plan: some ranks found a cached plan and others did not -- the ranks' "
                               "collective sequences would diverge (keep id-tensor lifetimes identical on all ranks)")
    if hit is not None:
        return hit
    flat = ids.reshape(-1).contiguous()
    if module.capacity_factor is not None and not exact:
        plan = _padded_plan(flat, module)
    else:
        plan = _exact_plan(flat, module)
    plans = _plans_of(anchor, True)
    if len(plans) >= 8:
        plans.clear()   # an id tensor modified in place over and over: keep the newest versions only
    plans[sub] = (plan, tuple(weakref.ref(t) for t in others))
    return plan


def forget_plans(anchor: torch.Tensor) -> None:
    """drop every cached plan keyed by ``anchor`` (after writing new ids into it through a raw pointer)"""
    _PLANS.pop(id(anchor), None)


# gradient exchanges issued by this backward pass and not finished yet: (module, weight, plan, exchange in flight).
# Drained by ONE engine callback at the end of the pass (what DDP's reducer uses to finalise), so that the all-to-alls of
# several tables queue up on the collective stream while the compute stream still permutes the next table's rows.
_DEFERRED: list = []


def _drain_deferred() -> None:
    jobs, _DEFERRED[:] = list(_DEFERRED), []
    for module, weight, plan, flight in jobs:
        module._land_gradient(weight, plan, flight.wait())


class _ShardedFinish(torch.autograd.Function):
    """second half of a lookup: rows that came back from the owners -> batch order.  Its backward sends the gradient
    rows to the owners; they land in ``weight.grad`` (or the sparse-mode buffer) at the end of the backward pass, so the
    function itself returns no gradient for ``weight``."""

    @staticmethod
    def forward(ctx, weight, module, plan, shape, flight):
        out = module.backend.gather_rows(flight.wait(), plan.perm)                  # batch order
        ctx.module, ctx.plan, ctx.weight = module, plan, weight
        return out.view(tuple(shape) + (weight.shape[1],))

    @staticmethod
    def backward(ctx, gout):
        module, plan, weight = ctx.module, ctx.plan, ctx.weight
        g = gout.reshape(-1, weight.shape[1]).contiguous()
        if plan.n == 0:      # (capacity-bounded slots of a rank that looked nothing up: nothing to read from)
            g_bucketed = torch.zeros((plan.inv.numel(), weight.shape[1]), dtype=g.dtype, device=g.device)
        else:
            g_bucketed = module.backend.gather_rows(g, plan.inv)                   # slot order
        flight = _exchange_start(g_bucketed, plan.send_counts, plan.recv_counts, module.group)   # to the owners
        _DEFERRED.append((module, weight, plan, flight))
        # (one callback per entry, the first drains them all: a pass that died before its callbacks ran cannot leave
        # the list in a state where nobody drains it)
        torch.autograd.Variable._execution_engine.queue_callback(_drain_deferred)
        return None, None, None, None, None


class Lookup:
    """a lookup whose rows are on their way: ``wait()`` returns them in batch order (differentiable w.r.t. the shard)"""

    __slots__ = ("module", "ids", "key", "plan", "shape", "flight")

    def __init__(self, module, ids, key, plan):
        self.module, self.ids, self.key, self.plan, self.shape = module, ids, key, plan, ids.shape
        rows = module.backend.gather_rows(module.weight.detach(), plan.recv_ids)    # this shard's rows
        self.flight = _exchange_start(rows, plan.recv_counts, plan.send_counts, module.group)

    def wait(self) -> torch.Tensor:
        module, plan = self.module, self.plan
        out = _ShardedFinish.apply(module.weight, module, plan, self.shape, self.flight)
        why = plan.verify() if module.verify else None     # (after everything above has been enqueued)
        if why is None:
            return out
        module.fallbacks += 1
        exact = exchange_plan(self.ids, module, self.key, exact=True)
        return Lookup(module, self.ids, self.key, exact).wait()


class ShardedEmbedding(torch.nn.Module):
    """``nn.Embedding(num_embeddings, dim)`` whose rows live on ``world`` ranks.

    ``weight`` is this rank's shard: global row r is local row ``r // world`` on rank
    ``r % world``.  ``forward(ids)`` returns the same values as the full table would
    (bit-exact: rows are only copied).

    ``capacity_factor`` (default: ``CTR_SHARD_CAPACITY`` or None = exact layout): every peer gets
    ``n / world * factor + 6 sqrt(n / world) + 32`` wire slots per lookup of n ids (``capacity``), and a fresh id tensor
    is looked up without a host read on the GPU's critical path (ExchangePlan).  Unused slots cost gather, wire and
    scatter work like real ones: 1.0 suits ids without hot spots, a hot id that is a fraction f of the batch needs
    ``1 + f * world``.  The ranks never talk about the wire size: either every rank passes the same number of ids per
    lookup (the equal split of a global batch), or ``capacity_ids`` names the largest lookup any rank will make and the
    wire is sized for that (ragged last batches are then fine).  ``owner = row % world`` spreads any id distribution except a single
    id repeated through much of the batch; if a bucket overflows anyway, the lookup is redone in the exact layout
    (``fallbacks`` counts them).  ``verify=False`` skips even the late look at the overflow state (for callers that read
    ``plan.verify()`` themselves, e.g. once per epoch).

    Dense gradient: the shard's gradient lives in ONE persistent buffer that the backward scatters into and hands out as
    ``weight.grad``; when ``optimizer.zero_grad()`` has dropped it, the next backward clears the rows the previous step
    touched (or the whole buffer, whichever is fewer bytes) instead of allocating and zero-filling a shard-sized tensor
    per lookup.  ``start(ids)`` issues a lookup and returns at once (``Lookup.wait()``): a model with several sharded
    tables starts them all, then waits for each where its rows are used, so that table k+1's rows travel while table k's
    interaction math runs."""

    def __init__(self, num_embeddings: int, embedding_dim: int, group=None, backend=None, device=None,
                 average: bool = True, capacity_factor: Optional[float] = None, capacity_ids: Optional[int] = None,
                 verify: bool = True):
        super().__init__()
        self.average = average
        if not dist.is_initialized():
            raise RuntimeError("ShardedEmbedding needs torch.distributed to be initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.backend = backend if backend is not None else HipShardBackend()
        if capacity_factor is None and os.environ.get("CTR_SHARD_CAPACITY"):
            capacity_factor = float(os.environ["CTR_SHARD_CAPACITY"])
        if capacity_factor is not None and capacity_factor < 1.0:
            raise ValueError("capacity_factor must be >= 1 (1.25: a quarter more wire slots than the even split)")
        self.capacity_factor, self.capacity_ids, self.verify, self.fallbacks = capacity_factor, capacity_ids, verify, 0
        local_rows = (num_embeddings - self.rank + self.world - 1) // self.world
        self.weight = torch.nn.Parameter(torch.empty(max(local_rows, 1), embedding_dim, device=device))
        self.weight.ctr_row_shard = True  # GradBucket leaves it alone
        torch.nn.init.normal_(self.weight, std=(2.0 / (num_embeddings + embedding_dim)) ** 0.5)  # xavier_normal_ of the full table
        self._gbuf = None          # the persistent dense gradient buffer
        self._touched: list = []   # row lists scattered into it since it was last clean
        self._touched_rows = 0

    def capacity(self, n: int) -> int:
        """wire slots per peer for a lookup of n ids: the even share times the factor, plus six standard deviations of
        a uniform draw (so that ``capacity_factor=1.0`` already never overflows on ids without hot spots) and a
        constant for tiny batches"""
        per = -(-n // self.world)
        return ((int(per * self.capacity_factor + 6.0 * per ** 0.5) + 32 + 7) // 8) * 8

    @torch.no_grad()
    def load_full_table(self, full: torch.Tensor) -> None:
        """take this rank's rows out of a full (num_embeddings, dim) table"""
        mine = full[self.rank::self.world]
        self.weight[:mine.shape[0]].copy_(mine)

    def sparse_grads(self, enable: bool = True):
        """opt-in sparse mode of the shard's gradient (sparse.py): backward lists the touched local rows instead of
        returning a shard-sized dense gradient; train with this package's ``optim.Adam``"""
        from . import sparse
        if enable and sparse.state_of(self.weight) is None:
            self.weight._ctr_sparse = sparse.SparseRows(self.weight)
        elif not enable and sparse.state_of(self.weight) is not None:
            del self.weight._ctr_sparse
        return self

    def _clean_buffer(self) -> torch.Tensor:
        weight = self.weight
        if self._gbuf is None or self._gbuf.shape != weight.shape or self._gbuf.device != weight.device:
            self._gbuf = torch.zeros_like(weight)
        elif self._touched_rows < 0 or self._touched_rows >= weight.shape[0]:
            self._gbuf.zero_()                                   # more row writes than the shard has rows
        else:
            for rows in self._touched:
                self.backend.zero_rows(self._gbuf, rows)
        self._touched, self._touched_rows = [], 0
        return self._gbuf

    @torch.no_grad()
    def _land_gradient(self, weight, plan, g_owner: torch.Tensor) -> None:
        """gradient rows that arrived from the ranks -> this shard's gradient"""
        # every rank's loss is a mean over ITS samples and the replicated parameters are averaged over the ranks
        # (GradBucket): the shard sums contributions of all ranks' samples, so the same global-mean gradient needs the
        # 1/world -- applied to the exchanged rows, not to the whole shard.  Unused wire slots of the capacity-bounded
        # layout carry arbitrary rows: the same pass multiplies them by 0.
        scale = 1.0 / self.world if self.average and self.world > 1 else 1.0
        if plan.valid is not None:
            g_owner.mul_(plan.valid if scale == 1.0 else plan.valid * scale)
        elif scale != 1.0:
            g_owner.mul_(scale)
        from . import sparse
        st = sparse.state_of(weight) if weight.is_cuda else None
        if st is not None:
            # sparse mode: into the shard's persistent accumulation buffer, pending rows listed, no dense gradient
            self.backend.scatter_add_rows(st.grad, plan.recv_ids, g_owner)
            sparse.mark([(weight, plan.mark)])
            return
        # dense semantics (the reference's optimizer sweeps whole tables)
        if weight.grad is None:
            weight.grad = self._clean_buffer()
        self.backend.scatter_add_rows(weight.grad, plan.recv_ids, g_owner)
        if weight.grad is self._gbuf and self._touched_rows >= 0:
            self._touched.append(plan.recv_ids)
            self._touched_rows += plan.recv_ids.numel()
            if self._touched_rows >= weight.shape[0]:
                self._touched, self._touched_rows = [], -1     # a full clear is cheaper: stop listing

    def start(self, ids: torch.Tensor, plan_key=None) -> Lookup:
        return Lookup(self, ids, plan_key, exchange_plan(ids, self, plan_key))

    def forward(self, ids: torch.Tensor, plan_key=None) -> torch.Tensor:
        return self.start(ids, plan_key).wait()
