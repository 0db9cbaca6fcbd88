"""``BCELoss`` -- drop-in for the ``torch.nn.BCELoss()`` every reference script builds
(e.g. scripts/pnn.py:54): mean reduction, log terms clamped at -100.  Forward is one pass +
a fixed-order reduction of <= 256 partials (torch: elementwise kernel + a single-workgroup
mean), backward one pass."""
from __future__ import annotations

import torch

from . import _lib


class _BCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target):
        _lib.require_device(prob, target)
        p, t = prob.reshape(-1), target.reshape(-1)
        if p.dtype != torch.float32 or t.dtype != torch.float32 or p.numel() != t.numel():
            raise ValueError("BCELoss expects float32 input and target of the same size")
        loss = torch.empty((), dtype=torch.float32, device=prob.device)
        ws = torch.empty(256, dtype=torch.float32, device=prob.device)
        rc = _lib.load().ctr_bce_fwd(p.data_ptr(), p.stride(0) if p.numel() > 1 else 1, t.data_ptr(),
                                     t.stride(0) if t.numel() > 1 else 1, p.numel(), loss.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "ctr_bce_fwd")
        ctx.save_for_backward(p, t)
        ctx.shape = prob.shape
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, t = ctx.saved_tensors
        gp = torch.empty(p.numel(), dtype=torch.float32, device=p.device)
        g = gloss.contiguous()
        rc = _lib.load().ctr_bce_bwd(p.data_ptr(), p.stride(0) if p.numel() > 1 else 1, t.data_ptr(),
                                     t.stride(0) if t.numel() > 1 else 1, p.numel(), g.data_ptr(), gp.data_ptr(), 1,
                                     _lib.stream_ptr())
        _lib.check(rc, "ctr_bce_bwd")
        return gp.view(ctx.shape), None


class BCELoss(torch.nn.Module):
    """``BCELoss()(prob, target)`` -> scalar mean loss, like ``torch.nn.BCELoss()``"""

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _BCEFunction.apply(input, target)
