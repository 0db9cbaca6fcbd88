"""shared host-side pieces of the model mirrors"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import _lib, ops, sparse


class CtrModule(nn.Module):
    """nn.Module whose forward/backward are libctrhip kernels.

    Index validation: kernels never fault on a bad id -- the forward reads it as row 0 and
    raises a device flag, the backward adds its gradient to no row.  The flag is read where
    the host syncs anyway: ``Trainer.model_eval`` (next to its ``loss.item()``), the sharded
    lookup's count exchange, or ``check_bad_index()`` called by hand -- each raises the
    ``IndexError`` ``nn.Embedding`` raises on CPU.  With ``CTRHIP_CHECK_INDEX=1`` (or
    ``self.check_index = True``) it is read after every forward (one sync per call).
    """

    check_index = os.environ.get("CTRHIP_CHECK_INDEX", "0") == "1"

    def _err_flag(self, device):
        flag = getattr(self, "_err", None)
        if flag is None or flag.device != device:
            flag = torch.zeros(1, dtype=torch.int32, device=device)
            object.__setattr__(self, "_err", flag)
        return flag

    def check_bad_index(self):
        """read the device flag (one sync) and raise IndexError if any lookup since the last check saw an
        id outside its table"""
        flag = getattr(self, "_err", None)
        if flag is not None and int(flag.item()) != 0:
            flag.zero_()
            raise IndexError("index out of range in self")

    def _raise_if_bad_index(self):
        if self.check_index:
            self.check_bad_index()

    @staticmethod
    def _need_device(*tensors):
        _lib.require_device(*tensors)

    # ---- opt-in sparse mode of the big tables' gradients (sparse.py)
    def sparse_ids(self, inputs):
        """{position in the autograd node's parameter list: [id tensors scattered into that table]} -- models
        whose tables can run in sparse mode override this"""
        return {}

    def _node_params(self):
        return self._params()

    def sparse_grads(self, enable: bool = True, min_rows: int = 65536):
        """switch the tables with at least ``min_rows`` rows to the sparse gradient mode (or back).  Call after the
        module is on its device; train with ``deeplearningrecommendationsystem_amd.optim.Adam`` (it updates the
        pending rows; a stock torch optimizer would see ``grad is None`` and skip these tables)."""
        params = self._node_params()
        positions = sorted(self.sparse_ids(None))
        if enable and not positions:
            raise NotImplementedError(f"{type(self).__name__} has no sparse-mode tables")
        for k in positions:
            p = params[k]
            if enable and p.shape[0] >= min_rows:
                if sparse.state_of(p) is None:
                    p._ctr_sparse = sparse.SparseRows(p)
            elif sparse.state_of(p) is not None:
                del p._ctr_sparse
        return self

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)
        sparse.discard(self.parameters())


def topk_rows(scores: torch.Tensor, k: int) -> np.ndarray:
    return ops.topk_rows(scores, k).cpu().numpy()


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
        if any(sparse.state_of(p) is not None for p in params):
            # sparse mode: the scatter went into the tables' persistent buffers; list the rows it touched and
            # hand autograd no dense gradient for those tables
            grads = list(grads)
            jobs = []
            for k, id_list in ctx.impl.sparse_ids(inputs).items():
                if sparse.state_of(params[k]) is not None:
                    jobs += [(params[k], ids) for ids in id_list]
                    grads[k] = None
            sparse.mark(jobs)
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

    def _run_fields(self, idx, params):
        """N-id-field generalisation: ``idx`` (B, F) int64 on the device"""
        self._need_device(idx, params[0])
        object.__setattr__(self, "_flag", self._err_flag(idx.device))
        out = _ModelFunction.apply(self, 1, idx, *params)
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

    def _rank_users(self, num_users, user_item, k, chunk: int = 1 << 18):
        """reference recommendation() (e.g. model/pnn.py:133-143): for every user, score the rows of the pandas
        frame ``user_item`` that carry its id and return the positions (within those rows) of the k best.  The
        reference filters the frame and calls forward once per user (943 host->device copies and forward calls);
        here the frame goes to the device once, the rows are grouped by user with one stable sort, scored in
        ``chunk``-row forward calls under no_grad, and ranked by one batched top-k."""
        dev = next(self.parameters()).device
        feats = torch.as_tensor(user_item.values, dtype=torch.float32).to(dev)
        uid = feats[:, 0].long()
        order = torch.argsort(uid, stable=True)                 # rows of a user keep their frame order
        feats = feats[order].contiguous()
        counts = torch.bincount(uid, minlength=num_users)[:num_users]
        scores = torch.empty(feats.shape[0], dtype=torch.float32, device=dev)
        with torch.no_grad():
            for lo in range(0, feats.shape[0], chunk):
                scores[lo:lo + chunk] = self.forward(feats[lo:lo + chunk]).view(-1)
        return _segment_topk(scores, counts, k)


def _segment_topk(scores: torch.Tensor, counts: torch.Tensor, k: int) -> np.ndarray:
    """top-k positions inside consecutive segments of ``scores`` (segment u has ``counts[u]`` entries)"""
    n_seg, longest = counts.numel(), int(counts.max()) if counts.numel() else 0
    if k > int(counts.min()):
        raise RuntimeError(f"selected index k out of range: a user has {int(counts.min())} candidate rows, k = {k}")
    if int(counts.min()) == longest:                                     # the usual case: every user x every item
        grid = scores.view(n_seg, longest)
    else:
        starts = torch.cumsum(counts, 0) - counts
        pos = torch.arange(longest, device=scores.device).unsqueeze(0)
        valid = pos < counts.unsqueeze(1)
        grid = torch.full((n_seg, longest), float("-inf"), device=scores.device)
        grid[valid] = scores[(starts.unsqueeze(1) + pos)[valid]]
    return ops.topk_rows(grid, k).cpu().numpy()
