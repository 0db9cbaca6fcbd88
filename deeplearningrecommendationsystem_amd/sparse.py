"""Opt-in sparse mode of the embedding gradient and the optimizer.

The reference trains every table with ``optim.Adam(model.parameters(), lr, weight_decay=1e-5)``
(scripts/din.py:87, trainer/trainer.py:39): a dense (V,E) gradient per table and Adam + L2 over every
row, 7 x the table bytes per step, although a batch touches at most ``batch`` rows per table.  With
``model.sparse_grads(True)`` the big tables keep a persistent gradient ACCUMULATION buffer that is clean
outside the rows a batch touched (csrc/sparse_rows.hip):

* backward scatters into that buffer with the unchanged kernels (no zero-fill) and appends the rows touched
  for the first time to the table's pending-row list; ``param.grad`` stays ``None`` for such a table;
* ``optim.Adam.step()`` (the mirror in this package) updates only the pending rows, with the same update
  rule -- L2 decay and moment decay happen on touched rows only ("lazy") -- and leaves buffer, flags and
  list clean; ``zero_grad`` discards pending rows.

Semantics therefore differ from the reference's dense Adam for rows a step does not touch, which is why
the mode is opt-in; parity is tested against a torch restatement of exactly this rule
(``lazy_adam_rows`` in the test infrastructure).  Everything is device-side with fixed launch geometry: a step in
sparse mode can be captured in a hipGraph like the dense one."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib


class SparseRows:
    """persistent sparse-mode state of one table parameter (attached as ``param._ctr_sparse``)"""

    def __init__(self, param: torch.Tensor):
        _lib.require_device(param)
        if param.dim() != 2 or param.dtype != torch.float32 or not param.is_contiguous():
            raise ValueError("sparse mode needs a contiguous float32 (vocab, dim) table")
        if param.shape[0] >= 2 ** 31:
            raise ValueError("sparse mode keeps int32 row lists: vocab must be below 2^31")
        dev = param.device
        self.vocab, self.dim = param.shape
        self.grad = torch.zeros_like(param)                       # accumulation buffer, clean outside pending rows
        self.flags = torch.zeros(self.vocab, dtype=torch.int32, device=dev)
        self.rows = torch.empty(self.vocab, dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.done = torch.zeros(1, dtype=torch.int32, device=dev)

    def pending(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sorted pending rows, their accumulated gradient rows) -- for inspection / tests (syncs)"""
        n = int(self.count.item())
        rows = torch.sort(self.rows[:n].long()).values
        return rows, self.grad[rows]

    def table_struct(self, param=None, exp_avg=None, exp_avg_sq=None) -> "_lib.RowsTable":
        return _lib.RowsTable(_lib.ptr(param), self.grad.data_ptr(), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                              self.flags.data_ptr(), self.rows.data_ptr(), self.count.data_ptr(), self.done.data_ptr(),
                              self.dim, 0)


def state_of(t: torch.Tensor) -> Optional[SparseRows]:
    return getattr(t, "_ctr_sparse", None)


def mark(jobs: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> None:
    """``jobs``: (table parameter in sparse mode, id tensor) pairs -- the ids whose gradient rows the backward has
    just scattered.  An id tensor is int64 or float32, 1-D with any element stride (a column of a matrix)."""
    todo = [(state_of(p), ids) for p, ids in jobs if state_of(p) is not None and ids.numel() > 0]
    lib = _lib.load()
    for base in range(0, len(todo), _lib.CTR_MAX_FIELDS):
        part = todo[base:base + _lib.CTR_MAX_FIELDS]
        arr = (_lib.RowsMark * len(part))()
        for k, (st, ids) in enumerate(part):
            _lib.require_device(ids)
            if ids.dim() != 1 or ids.dtype not in (torch.int64, torch.float32):
                raise ValueError("sparse.mark: ids must be a 1-D int64 / float32 tensor (a matrix column is fine)")
            arr[k] = _lib.RowsMark(ids.data_ptr(), ids.stride(0) if ids.numel() > 1 else 1, ids.numel(), st.vocab,
                                   st.flags.data_ptr(), st.rows.data_ptr(), st.count.data_ptr(),
                                   int(ids.dtype == torch.float32), 0)
        _lib.check(lib.ctr_rows_mark(arr, len(part), _lib.stream_ptr()), "ctr_rows_mark")


def discard(params: Iterable[torch.Tensor]) -> None:
    """drop pending gradient rows of every sparse-mode table in ``params`` (zero_grad)"""
    sts = [state_of(p) for p in params if state_of(p) is not None]
    lib = _lib.load()
    for base in range(0, len(sts), _lib.CTR_MAX_FIELDS):
        part = sts[base:base + _lib.CTR_MAX_FIELDS]
        arr = (_lib.RowsTable * len(part))(*[st.table_struct() for st in part])
        _lib.check(lib.ctr_rows_discard(arr, len(part), _lib.stream_ptr()), "ctr_rows_discard")


def adam_rows(items: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], lr, betas, eps, weight_decay, step) -> None:
    """row-wise Adam on the pending rows of ``items`` = [(param, exp_avg, exp_avg_sq)] (all in sparse mode)"""
    lib = _lib.load()
    for base in range(0, len(items), _lib.CTR_MAX_FIELDS):
        part = items[base:base + _lib.CTR_MAX_FIELDS]
        arr = (_lib.RowsTable * len(part))(*[state_of(p).table_struct(p, m, v) for p, m, v in part])
        _lib.check(lib.ctr_adam_rows(arr, len(part), lr, betas[0], betas[1], eps, weight_decay, step, _lib.stream_ptr()),
                   "ctr_adam_rows")
