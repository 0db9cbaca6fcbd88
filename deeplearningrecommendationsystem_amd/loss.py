"""``BCELoss`` -- drop-in for the ``torch.nn.BCELoss()`` every reference script builds
(e.g. scripts/pnn.py:54): mean reduction, log terms clamped at -100.  Forward is ONE launch: a pass
over the samples whose last workgroup sums the <= 256 partials in a fixed order (torch: elementwise
kernel + a single-workgroup mean), backward one pass."""
from __future__ import annotations

import torch

from . import _lib


_tickets = {}


def _ticket(device: torch.device) -> torch.Tensor:
    """the zeroed device word ctr_bce_fwd finds its last workgroup with; the kernel leaves it zero.
    One per device: losses of one device are issued from one stream at a time here (the trainer's
    stream, or the hipGraph capture stream after its warm-up) -- a second stream computing a loss
    CONCURRENTLY would need its own word.  Created outside any capture (a tensor made during
    capture would add a fill to every replay)."""
    t = _tickets.get(device.index)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("BCELoss: run one step before capturing (GraphedStep does) so its ticket exists")
        t = _tickets[device.index] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


_units = {}


def unit_grad(device: torch.device) -> torch.Tensor:
    """THE scalar 1.0 of a device: pass it as ``loss.backward(unit_grad(dev))`` (GraphedStep does) and
    BCELoss' backward recognises it by address and returns the gradient its forward launch already
    wrote, instead of launching the scaling kernel.  Never write to it."""
    t = _units.get(device.index)
    if t is None:
        t = _units[device.index] = torch.ones((), dtype=torch.float32, device=device)
    return t


class _BCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target):
        _lib.require_device(prob, target)
        p, t = prob.reshape(-1), target.reshape(-1)
        if p.dtype != torch.float32 or t.dtype != torch.float32 or p.numel() != t.numel():
            raise ValueError("BCELoss expects float32 input and target of the same size")
        loss = torch.empty((), dtype=torch.float32, device=prob.device)
        ws = torch.empty(256, dtype=torch.float32, device=prob.device)
        # d loss / d prob for an upstream gradient of exactly 1, written by the same launch
        gp1 = torch.empty(p.numel(), dtype=torch.float32, device=prob.device) if ctx.needs_input_grad[0] else None
        rc = _lib.load().ctr_bce_fwd(p.data_ptr(), p.stride(0) if p.numel() > 1 else 1, t.data_ptr(),
                                     t.stride(0) if t.numel() > 1 else 1, p.numel(), loss.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _ticket(prob.device).data_ptr(), _lib.ptr(gp1),
                                     _lib.stream_ptr())
        _lib.check(rc, "ctr_bce_fwd")
        ctx.save_for_backward(p, t)
        ctx.shape = prob.shape
        ctx.gp1 = gp1
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, t = ctx.saved_tensors
        unit = _units.get(p.device.index)
        if ctx.gp1 is not None and unit is not None and gloss.data_ptr() == unit.data_ptr():
            return ctx.gp1.view(ctx.shape), None  # upstream gradient is THE 1.0: forward wrote this already
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
