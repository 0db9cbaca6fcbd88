"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

Data-parallel part: the batch is split across ranks; parameters that are
replicated (dense layers and ml-100k-sized tables) get their gradients averaged
with ONE all-reduce over a flat bucket after backward -- the only collective a
replicated model needs, sized in the low MBs for every model of the zoo."""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class GradBucket:
    """flat fp32 bucket holding the gradients of ``params`` (fixed order)"""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def all_reduce_mean(self, group=None) -> None:
        """average gradients over the ranks (no-op for a single process)"""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / dist.get_world_size(group))
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
