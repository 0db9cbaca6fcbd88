"""The fused embedding stage on its own: F single-id fields -> one (B, F*E) matrix.

Not a reference class: it is the generalisation the BASELINE "roofline shape" needs
(26 fields x 1e6 rows x E=16, batch 65536) -- what
``torch.cat([emb_f(idx[:, f]) for f in range(F)], 1)`` (the pattern of model/pnn.py:113-121,
model/deepfm.py:45-54) becomes with more fields than the reference hard-codes.  One
``ctr_embed_fwd`` launch gathers all fields; backward leaves a dense ``.grad`` per table."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import FieldSpec
from .._lib import FIELD_ID_I64
from ._base import CtrModule, _ModelFunction


class EmbeddingStage(CtrModule):
    """``forward(idx)`` with ``idx`` (B, F) int64 returns (B, F*E) fp32: columns
    [f*E, (f+1)*E) hold ``tables[f][idx[:, f]]`` (bit-exact row copies)."""

    def __init__(self, num_fields: int, vocab: int, dim: int):
        super().__init__()
        self.num_fields, self.vocab, self.dim = num_fields, vocab, dim
        self.tables = nn.ParameterList([nn.Parameter(torch.empty(vocab, dim)) for _ in range(num_fields)])
        for t in self.tables:
            xavier_normal_(t.data)

    def _specs(self, idx, tables):
        f, e = self.num_fields, self.dim
        return [FieldSpec(FIELD_ID_I64, e, k * e, table=tables[k], idx=idx[:, k], idx_stride=f) for k in range(f)]

    def _node_params(self):
        return list(self.tables)

    def sparse_ids(self, inputs):
        if inputs is None:
            return {k: [] for k in range(self.num_fields)}
        return {k: [inputs[0][:, k]] for k in range(self.num_fields)}

    def run_forward(self, inputs, params):
        (idx,) = inputs
        out = torch.empty((idx.shape[0], self.num_fields * self.dim), dtype=torch.float32, device=params[0].device)
        ops.embed_fwd(self._specs(idx, params), None, idx.shape[0], out, self._flag)
        return out, None

    def run_backward(self, state, inputs, params, gout):
        (idx,) = inputs
        zeros = ops.zero_grads(list(params))
        ops.embed_bwd(self._specs(idx, params), None, idx.shape[0], gout, zeros)
        return [zeros[id(p)] for p in params]

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        params = list(self.tables)
        self._need_device(idx, params[0])
        if idx.dim() != 2 or idx.shape[1] != self.num_fields or idx.dtype != torch.int64:
            raise ValueError(f"expected a (B,{self.num_fields}) int64 index matrix, got {tuple(idx.shape)} {idx.dtype}")
        object.__setattr__(self, "_flag", self._err_flag(idx.device))
        out = _ModelFunction.apply(self, 1, idx.contiguous(), *params)
        self._raise_if_bad_index()
        return out
