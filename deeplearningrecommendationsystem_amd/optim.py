"""``Adam`` -- drop-in for the ``torch.optim.Adam(model.parameters(), lr, weight_decay=1e-5)``
of the reference's scripts (e.g. scripts/pnn.py:55; no amsgrad, no maximize): identical update
rule, every parameter tensor updated by ONE libctrhip launch (torch's foreach path makes
several passes over the 64 MB .. 2.5 GB tables that dense Adam + L2 decay touches whole)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, sparse


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            todo = []
            rows = []   # tables in sparse mode (sparse.py): Adam on their pending rows only
            step = None
            for p in group["params"]:
                if sparse.state_of(p) is not None:
                    st = self.state[p]
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    st["step"] += 1
                    step = st["step"] if step is None else step
                    if st["step"] != step:
                        raise RuntimeError("parameters of one group must share a step count")
                    rows.append((p, st["exp_avg"], st["exp_avg_sq"]))
                    continue
                if p.grad is None:
                    continue
                _lib.require_device(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("Adam (libctrhip) handles contiguous float32 parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                step = st["step"] if step is None else step
                if st["step"] != step:
                    raise RuntimeError("parameters of one group must share a step count")
                todo.append((p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            if rows:
                sparse.adam_rows(rows, group["lr"], group["betas"], group["eps"], group["weight_decay"], step)
            if not todo:
                continue
            arr = (_lib.AdamTensor * len(todo))()
            for k, (p, g, m, v) in enumerate(todo):
                arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
                arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].numel = m.data_ptr(), v.data_ptr(), p.numel()
            rc = lib.ctr_adam_step(arr, len(todo), group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                                   group["weight_decay"], step, _lib.stream_ptr())
            _lib.check(rc, "ctr_adam_step")
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """also drops the pending gradient rows of tables in sparse mode"""
        super().zero_grad(set_to_none)
        sparse.discard(p for group in self.param_groups for p in group["params"])
