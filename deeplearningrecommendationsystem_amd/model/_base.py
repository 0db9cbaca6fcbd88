"""shared host-side pieces of the model mirrors"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import _lib


class CtrModule(nn.Module):
    """nn.Module whose forward/backward are libctrhip kernels.

    Index validation: kernels never fault on a bad id (the row is read as row 0)
    and raise a device flag; with ``CTRHIP_CHECK_INDEX=1`` (or
    ``self.check_index = True``) the flag is read back after the forward and an
    ``IndexError`` is raised like ``nn.Embedding`` does on CPU (costs a sync).
    """

    check_index = os.environ.get("CTRHIP_CHECK_INDEX", "0") == "1"

    def _err_flag(self, device):
        flag = getattr(self, "_err", None)
        if flag is None or flag.device != device:
            flag = torch.zeros(1, dtype=torch.int32, device=device)
            object.__setattr__(self, "_err", flag)
        return flag

    def _raise_if_bad_index(self):
        if self.check_index and getattr(self, "_err", None) is not None:
            if int(self._err.item()) != 0:
                self._err.zero_()
                raise IndexError("index out of range in self")

    @staticmethod
    def _need_device(*tensors):
        _lib.require_device(*tensors)


def topk_rows(scores: torch.Tensor, k: int) -> np.ndarray:
    return torch.topk(scores, k, dim=-1).indices.cpu().numpy()


class _ModelFunction(torch.autograd.Function):
    """one autograd node per model: ``impl.run_forward(inputs, params)`` returns
    ``(output, state)``; ``impl.run_backward(state, inputs, params, gout)``
    returns one gradient (or None) per parameter.  Forward and backward are
    straight sequences of libctrhip launches on torch's current stream."""

    @staticmethod
    def forward(ctx, impl, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        out, state = impl.run_forward(inputs, params)
        ctx.impl, ctx.state, ctx.n_inputs = impl, state, n_inputs
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, gout):
        tensors = ctx.saved_tensors
        inputs, params = tensors[:ctx.n_inputs], tensors[ctx.n_inputs:]
        grads = ctx.impl.run_backward(ctx.state, inputs, params, gout.contiguous())
        ctx.state = None
        return (None, None) + (None,) * ctx.n_inputs + tuple(grads)


class FeatureModel(CtrModule):
    """models fed by the (B,45) float feature matrix of data/reader.py:98-112"""

    def _run_model(self, x, params):
        self._need_device(x, params[0])
        if x.dim() != 2 or x.shape[1] != 45 or x.dtype != torch.float32:
            raise ValueError(f"expected a (B,45) float32 feature matrix, got {tuple(x.shape)} {x.dtype}")
        x = x if x.stride(1) == 1 else x.contiguous()
        object.__setattr__(self, "_flag", self._err_flag(x.device))
        out = _ModelFunction.apply(self, 1, x, *params)
        self._raise_if_bad_index()
        return out

    def _aligned_weight(self, w, refresh=True):
        """weights whose rows are not a multiple of 4 floats long: the GEMM kernels stage 16-byte
        aligned rows several times faster, so they read a copy padded to a multiple of 4 columns
        (refreshed on every forward -- the optimizer updates ``w`` in place)."""
        k = w.shape[1]
        if k % 4 == 0:
            return w
        pad = getattr(self, "_padded", None)
        if pad is None:
            pad = {}
            object.__setattr__(self, "_padded", pad)
        buf = pad.get(id(w))
        if buf is None or buf.device != w.device or buf.shape[0] != w.shape[0]:
            buf = torch.zeros((w.shape[0], (k + 3) // 4 * 4), dtype=w.dtype, device=w.device)
            pad[id(w)] = buf
        if refresh:
            buf[:, :k].copy_(w.detach())
        return buf[:, :k]

    @staticmethod
    def _padded_rows(rows, width, device):
        """(rows, width) buffer whose row stride is a multiple of 4 floats"""
        return torch.empty((rows, (width + 3) // 4 * 4), dtype=torch.float32, device=device)[:, :width]

    def _rank_users(self, num_users, user_item, k):
        """reference recommendation(): per-user scoring of the rows of the pandas
        frame ``user_item`` (e.g. model/pnn.py:133-143)"""
        rows = []
        dev = next(self.parameters()).device
        with torch.no_grad():
            for u in range(num_users):
                feats = torch.tensor(user_item[user_item['user_id'] == u].values, dtype=torch.float32, device=dev)
                scores = self.forward(feats)
                rows.append(torch.topk(scores, k, dim=0).indices.view(1, -1).tolist()[0])
        return np.array(rows)
