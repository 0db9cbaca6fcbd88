"""Matrix factorisation -- counterpart of the reference's model/mf.py:10-35."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ._base import CtrModule


class _MFFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user_table, item_table, user_idx, item_idx, err_flag):
        prob = ops.mf_fwd(user_table, item_table, user_idx, item_idx, err_flag)
        ctx.save_for_backward(user_table, item_table, user_idx, item_idx, prob)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        user_table, item_table, user_idx, item_idx, prob = ctx.saved_tensors
        zeros = ops.zero_grads([user_table, item_table])
        gu = zeros[id(user_table)] if ctx.needs_input_grad[0] else None
        gi = zeros[id(item_table)] if ctx.needs_input_grad[1] else None
        ops.mf_bwd(user_table, item_table, user_idx, item_idx, prob, gprob.contiguous(), gu, gi)
        return gu, gi, None, None, None


class MatrixFactorization(CtrModule):
    """``MatrixFactorization(num_users, num_items, embedding_size)``;
    ``forward(user_indices, item_indices) -> (B,)`` probabilities
    (reference model/mf.py:12-26)."""

    def __init__(self, num_users: int, num_items: int, embedding_size: int):
        super().__init__()
        self.user_embeddings = nn.Embedding(num_users, embedding_size)
        self.item_embeddings = nn.Embedding(num_items, embedding_size)
        xavier_normal_(self.user_embeddings.weight.data)
        xavier_normal_(self.item_embeddings.weight.data)

    def forward(self, user_indices: torch.Tensor, item_indices: torch.Tensor) -> torch.Tensor:
        w_u, w_i = self.user_embeddings.weight, self.item_embeddings.weight
        self._need_device(w_u, user_indices, item_indices)
        out = _MFFunction.apply(w_u, w_i, user_indices.contiguous(), item_indices.contiguous(),
                                self._err_flag(w_u.device))
        self._raise_if_bad_index()
        return out

    def recommendation(self, num_users, num_items):
        """full-catalogue ranking (reference model/mf.py:28-35)"""
        with torch.no_grad():
            dev = self.user_embeddings.weight.device
            users = self.user_embeddings.weight[:num_users]
            items = self.item_embeddings.weight[:num_items]
            scores = ops.linear_fwd(users.contiguous(), items.contiguous(), None)
            return ops.topk_rows(scores, num_items).cpu().numpy()
