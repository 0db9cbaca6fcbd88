"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).  Nothing like it exists in the reference
(single device).

* ``GradBucket`` -- data-parallel part: the batch is split across ranks; parameters that
  are replicated (dense layers, ml-100k-sized tables) get their gradients averaged with
  ONE all-reduce over a flat bucket (low MBs for every model of the zoo).
* ``ShardedEmbedding`` -- model-parallel part for the 1e6..1e7-row tables: rows are dealt
  round-robin to the ranks (``owner = row % world``, ``local = row // world``: uniform load
  whatever the id skew).  A lookup is: bucket ids by owner (HIP kernel) -> all_to_all of the
  per-rank counts and of the ids -> local row gather (HIP kernel) -> all_to_all of the
  rows back -> un-permute into batch order (HIP kernel).  all-to-all is the right xGMI
  primitive: a full mesh of point-to-point links, all seven used concurrently, where a ring
  all-reduce would be bound by a single link.  Backward mirrors it and ends in a scatter-add
  into the local shard's dense gradient.

The compute steps are behind a small backend interface so that the exchange protocol can
be exercised on CPU with gloo (tests inject a numpy backend); the default backend is the
HIP one and needs the MI355X library like everything else in this package.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


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
    def bucket(ids: torch.Tensor, world: int):
        """-> (counts (world,) int64 on device, send (n,), perm (n,), inv (n,))"""
        from . import _lib
        _lib.require_device(ids)
        n = ids.numel()
        dev = ids.device
        counts = torch.empty(world, dtype=torch.int64, device=dev)
        cursor = torch.empty(world, dtype=torch.int64, device=dev)
        send = torch.empty(n, dtype=torch.int64, device=dev)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        inv = torch.empty(n, dtype=torch.int64, device=dev)
        rc = _lib.load().ctr_shard_bucket(_lib.ptr(ids) if n else None, n, world, counts.data_ptr(), cursor.data_ptr(),
                                          _lib.ptr(send) if n else None, _lib.ptr(perm) if n else None,
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


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids, module):
        be, group, world = module.backend, module.group, module.world
        flat = ids.reshape(-1).contiguous()
        counts, send_ids, perm, inv = be.bucket(flat, world)
        send_counts = [int(c) for c in counts.tolist()]                  # one host sync per lookup
        if _host_staged(counts, group):
            recv_counts_t = torch.empty(world, dtype=counts.dtype)
            dist.all_to_all_single(recv_counts_t, counts.cpu(), group=group)
        else:
            recv_counts_t = torch.empty_like(counts)
            dist.all_to_all_single(recv_counts_t, counts, group=group)
        recv_counts = [int(c) for c in recv_counts_t.tolist()]
        recv_ids = _exchange(send_ids, send_counts, recv_counts, group)   # local rows other ranks want
        rows = be.gather_rows(weight, recv_ids)                           # this shard's rows
        back = _exchange(rows, recv_counts, send_counts, group)          # my rows, bucket order
        out = be.gather_rows(back, perm)                                  # batch order
        ctx.module = module
        ctx.send_counts, ctx.recv_counts = send_counts, recv_counts
        ctx.save_for_backward(weight, recv_ids, inv)
        return out.view(tuple(ids.shape) + (weight.shape[1],))

    @staticmethod
    def backward(ctx, gout):
        weight, recv_ids, inv = ctx.saved_tensors
        be, group = ctx.module.backend, ctx.module.group
        g = gout.reshape(-1, weight.shape[1]).contiguous()
        g_bucketed = be.gather_rows(g, inv)                                            # bucket order
        g_owner = _exchange(g_bucketed, ctx.send_counts, ctx.recv_counts, group)      # to the owners
        grad = torch.zeros_like(weight)
        be.scatter_add_rows(grad, recv_ids, g_owner)
        if ctx.module.average:
            # every rank's loss is a mean over ITS samples and the replicated parameters are
            # averaged over the ranks (GradBucket): the shard sums contributions of all ranks'
            # samples, so the same global-mean gradient needs the 1/world here
            grad.mul_(1.0 / ctx.module.world)
        return grad, None, None


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

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        return _ShardedLookup.apply(self.weight, ids, self)
