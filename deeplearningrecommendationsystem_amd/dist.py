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
        cursor = torch.empty(world, dtype=torch.int64, device=dev)
        send = torch.empty(n, dtype=torch.int32, device=dev)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        inv = torch.empty(n, dtype=torch.int64, device=dev)
        rc = _lib.load().ctr_shard_bucket(_lib.ptr(ids) if n else None, n, world, vocab, counts.data_ptr(),
                                          cursor.data_ptr(), _lib.ptr(send) if n else None, _lib.ptr(perm) if n else None,
                                          _lib.ptr(inv) if n else None, _lib.stream_ptr())
        _lib.check(rc, "ctr_shard_bucket")
        return counts, send, perm, inv

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


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo has no device all-to-all: rehearsals with several ranks on one GPU go through the host"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _exchange(send: torch.Tensor, send_counts: List[int], recv_counts: List[int], group) -> torch.Tensor:
    """all_to_all_single with per-rank row counts; trailing dims are kept"""
    shape = (sum(recv_counts),) + tuple(send.shape[1:])
    if _host_staged(send, group):
        out = torch.empty(shape, dtype=send.dtype)
        dist.all_to_all_single(out, send.contiguous().cpu(), output_split_sizes=recv_counts,
                               input_split_sizes=send_counts, group=group)
        return out.to(send.device)
    out = torch.empty(shape, dtype=send.dtype, device=send.device)
    dist.all_to_all_single(out, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts,
                           group=group)
    return out


class ExchangePlan:
    """everything about ONE id tensor's lookup that does not depend on table contents: who owns which id
    (bucket order, counts per rank) and which of this rank's rows the peers asked for.  Built once per id tensor
    -- bucket kernel, count exchange, ONE host read of 2*world + 1 integers (the only sync), the id exchange
    (int32 local rows on the wire) -- and reused by every table looked up with those ids (FFM: two field-aware
    tables per id column) and by every later step that passes the same, unmodified tensor (the reference trains
    full-batch on the same tensors every epoch, scripts/din.py:93-96): those steps run no bucketing, no id exchange
    and no host sync at all, only the two row exchanges."""

    __slots__ = ("send_counts", "recv_counts", "perm", "inv", "recv_ids", "n")


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


def exchange_plan(ids: torch.Tensor, module: "ShardedEmbedding", key=None) -> ExchangePlan:
    """``key`` = (tensor whose identity and version stand for the ids, extra) when ``ids`` itself is a temporary (FFM:
    ``x[:, 0].long()`` of the feature matrix ``x``; DIN: ``cat(hist, target)`` keyed by ``(hist, target)``); default:
    the id tensor itself.

    The cache decides per rank, so every rank must present the same sequence of (cached / fresh) lookups: a rank that
    hits while a peer misses would skip the collectives the peer enters.  Identical id-tensor lifetimes on all ranks
    (the usual SPMD loop) guarantee it; ``CTR_DIST_DEBUG=1`` all-reduces the decision and raises on a mismatch.
    ``_version`` does not see writes through raw pointers (this library's kernels): an id tensor filled by a kernel
    must be passed as a new tensor object, or the plan dropped with ``forget_plans(tensor)``."""
    anchor, others, plain = _split_key(ids, key)
    sub = (anchor._version, plain, id(module.group), module.world, module.num_embeddings, tuple(ids.shape))
    plans = _plans_of(anchor, False)
    hit = None
    if plans is not None and sub in plans:
        plan, refs = plans[sub]
        if all(r() is t for r, t in zip(refs, others)):
            hit = plan
        else:
            del plans[sub]      # an id() recycled by a new tensor: the old plan is for other ids
    be, group, world = module.backend, module.group, module.world
    if _DEBUG:
        mine = torch.tensor([1 if hit is not None else 0, 1], dtype=torch.int64)
        dist.all_reduce(mine, group=group)
        if int(mine[0]) not in (0, int(mine[1])):
            raise RuntimeError("exchange_plan: some ranks found a cached plan and others did not -- the ranks' "
                               "collective sequences would diverge (keep id-tensor lifetimes identical on all ranks)")
    if hit is not None:
        return hit
    flat = ids.reshape(-1).contiguous()
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
    plans = _plans_of(anchor, True)
    if len(plans) >= 8:
        plans.clear()   # an id tensor modified in place over and over: keep the newest versions only
    plans[sub] = (plan, tuple(weakref.ref(t) for t in others))
    return plan


def forget_plans(anchor: torch.Tensor) -> None:
    """drop every cached plan keyed by ``anchor`` (after writing new ids into it through a raw pointer)"""
    _PLANS.pop(id(anchor), None)


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, module, plan, shape):
        be, group = module.backend, module.group
        rows = be.gather_rows(weight, plan.recv_ids)                                # this shard's rows
        back = _exchange(rows, plan.recv_counts, plan.send_counts, group)          # my rows, bucket order
        out = be.gather_rows(back, plan.perm)                                       # batch order
        ctx.module, ctx.plan = module, plan
        ctx.save_for_backward(weight)
        return out.view(tuple(shape) + (weight.shape[1],))

    @staticmethod
    def backward(ctx, gout):
        (weight,) = ctx.saved_tensors
        module, plan = ctx.module, ctx.plan
        be, group = module.backend, module.group
        g = gout.reshape(-1, weight.shape[1]).contiguous()
        g_bucketed = be.gather_rows(g, plan.inv)                                         # bucket order
        g_owner = _exchange(g_bucketed, plan.send_counts, plan.recv_counts, group)      # to the owners
        if module.average:
            # every rank's loss is a mean over ITS samples and the replicated parameters are averaged over the
            # ranks (GradBucket): the shard sums contributions of all ranks' samples, so the same global-mean
            # gradient needs the 1/world -- applied to the exchanged rows, not to the whole shard
            g_owner.mul_(1.0 / module.world)
        from . import sparse
        st = sparse.state_of(weight) if weight.is_cuda else None
        if st is not None:
            # sparse mode: into the shard's persistent accumulation buffer, pending rows listed, no dense gradient
            be.scatter_add_rows(st.grad, plan.recv_ids, g_owner)
            sparse.mark([(weight, plan.recv_ids)])
            return None, None, None, None
        grad = torch.zeros_like(weight)   # dense semantics (the reference's optimizer sweeps whole tables)
        be.scatter_add_rows(grad, plan.recv_ids, g_owner)
        return grad, None, None, None


class ShardedEmbedding(torch.nn.Module):
    """``nn.Embedding(num_embeddings, dim)`` whose rows live on ``world`` ranks.

    ``weight`` is this rank's shard: global row r is local row ``r // world`` on rank
    ``r % world``.  ``forward(ids)`` returns the same values as the full table would
    (bit-exact: rows are only copied)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, group=None, backend=None, device=None,
                 average: bool = True):
        super().__init__()
        self.average = average
        if not dist.is_initialized():
            raise RuntimeError("ShardedEmbedding needs torch.distributed to be initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.backend = backend if backend is not None else HipShardBackend()
        local_rows = (num_embeddings - self.rank + self.world - 1) // self.world
        self.weight = torch.nn.Parameter(torch.empty(max(local_rows, 1), embedding_dim, device=device))
        self.weight.ctr_row_shard = True  # GradBucket leaves it alone
        torch.nn.init.normal_(self.weight, std=(2.0 / (num_embeddings + embedding_dim)) ** 0.5)  # xavier_normal_ of the full table

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

    def forward(self, ids: torch.Tensor, plan_key=None) -> torch.Tensor:
        plan = exchange_plan(ids, self, plan_key)
        return _ShardedLookup.apply(self.weight, self, plan, ids.shape)
