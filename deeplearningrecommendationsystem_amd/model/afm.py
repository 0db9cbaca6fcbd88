"""AFM -- counterpart of the reference's model/afm.py:7-83."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, FieldSpec
from .._lib import FIELD_BAG, FIELD_ID_F32
from ._base import FeatureModel

NVEC, NPAIRS = 6, 15


class AFM(FeatureModel):
    """``AFM(num_users, num_items, embedding_dim, attention_dim)``; ``forward(x: (B,45)) -> (B,1)``.

    Vectors [user, item, age broadcast to E, gender, occupation, movie] in one embedding-stage
    launch (the age broadcast is a 1-row bag over a constant table of ones) -> the 15 pair
    products as a (B*15, E) operand -> attention net ``relu(P W + b) h`` on the matrix cores ->
    softmax over the pairs and weighted sum (the DIN pooling kernel with L = 15) ->
    ``output_layer`` with the linear part as the residual of its sigmoid."""

    def __init__(self, num_users, num_items, embedding_dim, attention_dim):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, embedding_dim)
        self.item_embedding = nn.Embedding(num_items, embedding_dim)
        self.gender_embedding = nn.Embedding(2, embedding_dim)
        self.occupation_embedding = nn.Embedding(21, embedding_dim)
        self.movie_embedding = nn.Embedding(19, embedding_dim)
        self.attention_W = nn.Parameter(torch.randn(embedding_dim, attention_dim))
        self.attention_b = nn.Parameter(torch.randn(attention_dim))
        self.attention_h = nn.Parameter(torch.randn(attention_dim, 1))
        self.output_layer = nn.Linear(embedding_dim, 1)
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.linear = nn.Linear(1 + 2 + 21 + 19, 1)
        for emb in (self.user_embedding, self.item_embedding, self.gender_embedding, self.occupation_embedding,
                    self.movie_embedding, self.user, self.item):
            xavier_normal_(emb.weight.data)

    def _params(self):
        return [self.user_embedding.weight, self.item_embedding.weight, self.gender_embedding.weight,
                self.occupation_embedding.weight, self.movie_embedding.weight, self.attention_W, self.attention_b,
                self.attention_h, self.output_layer.weight, self.output_layer.bias, self.user.weight,
                self.item.weight, self.linear.weight, self.linear.bias]

    def forward(self, x):
        return self._run_model(x, self._params())

    def _ones(self, e, device):
        ones = getattr(self, "_age_ones", None)
        if ones is None or ones.device != device or ones.shape[1] != e:
            ones = torch.ones((1, e), dtype=torch.float32, device=device)
            object.__setattr__(self, "_age_ones", ones)
        return ones

    def _specs(self, tables, e, device):
        user, item, gender, occ, movie = tables
        return [
            FieldSpec(FIELD_ID_F32, e, 0 * e, table=user, src_col=0),
            FieldSpec(FIELD_ID_F32, e, 1 * e, table=item, src_col=1),
            FieldSpec(FIELD_BAG, e, 2 * e, table=self._ones(e, device), src_col=2, bag_size=1),   # age * ones(E)
            FieldSpec(FIELD_BAG, e, 3 * e, table=gender, src_col=3, bag_size=2),
            FieldSpec(FIELD_BAG, e, 4 * e, table=occ, src_col=5, bag_size=21),
            FieldSpec(FIELD_BAG, e, 5 * e, table=movie, src_col=26, bag_size=19),
        ]

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables = params[:5]
        att_w, att_b, att_h, out_w, out_b, user1, item1, lin_w, lin_b = params[5:14]
        batch, e, dev = x.shape[0], tables[0].shape[1], x.device
        emb = torch.empty((batch, NVEC * e), dtype=torch.float32, device=dev)
        ops.embed_fwd(self._specs(tables, e, dev), x, batch, emb, self._flag)
        pairs = self._padded_rows(batch * NPAIRS, e, dev)
        ops.pairprod_fwd(emb, NVEC, e, pairs)
        wt = att_w.t().contiguous()                       # (A, E): nn.Linear layout of the attention map
        hidden = ops.linear_fwd(pairs, self._aligned_weight(wt), att_b, ACT_RELU)
        ht = att_h.t().contiguous()                       # (1, A)
        score = ops.linear_fwd(hidden, ht, None, ACT_NONE)
        attn = torch.empty((batch, NPAIRS), dtype=torch.float32, device=dev)
        pooled = self._padded_rows(batch, e, dev)
        ops.din_pool_fwd(score, pairs, batch, NPAIRS, e, attn, pooled, summed=True)
        wide = torch.empty((batch, 1), dtype=torch.float32, device=dev)
        ops.fm_wide_fwd(emb[:, :e], 1, e, x, user1, item1, lin_w, lin_b, wide, self._flag)
        prob = ops.linear_fwd(pooled, out_w, out_b, ACT_SIGMOID, residual=wide)
        return prob, (emb, pairs, wt, ht, hidden, attn, pooled, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, pairs, wt, ht, hidden, attn, pooled, prob = state
        tables = params[:5]
        att_w, att_b, att_h, out_w, out_b, user1, item1, lin_w, lin_b = params[5:14]
        batch, e, dev = x.shape[0], tables[0].shape[1], x.device
        zeros = ops.zero_grads(params)
        # sigmoid(wide + pooled W^T + b): the residual's gradient is gz itself
        gz = torch.empty_like(prob)
        ops.act_bwd(prob, gprob, ACT_SIGMOID, gz, accumulate=False)
        gpooled = self._padded_rows(batch, e, dev)
        ops.linear_bwd(pooled, out_w, None, gz, ACT_NONE, gpooled, zeros[id(out_w)], zeros[id(out_b)])
        ops.fm_wide_bwd(emb[:, :e], 1, e, x, user1, item1, lin_w, lin_b, gz, zeros[id(user1)], zeros[id(item1)],
                        zeros[id(lin_w)], zeros[id(lin_b)], None, accumulate=False)
        gscore = torch.empty((batch * NPAIRS, 1), dtype=torch.float32, device=dev)
        ops.din_pool_bwd(attn, pairs, batch, NPAIRS, e, gpooled, True, gscore)
        ghidden = torch.empty_like(hidden)
        ght = torch.zeros_like(ht)
        ops.linear_bwd(hidden, ht, None, gscore, ACT_NONE, ghidden, ght, None)
        gpairs = self._padded_rows(batch * NPAIRS, e, dev)
        gwt = torch.zeros_like(wt)
        ops.linear_bwd(pairs, self._aligned_weight(wt, refresh=False), hidden, ghidden, ACT_RELU, gpairs, gwt,
                       zeros[id(att_b)])
        zeros[id(att_w)].copy_(gwt.t())
        zeros[id(att_h)].copy_(ght.t())
        gemb = torch.empty_like(emb)
        ops.pairprod_bwd(emb, NVEC, e, gpairs, attn, gpooled, gemb, accumulate=False)
        ops.embed_bwd(self._specs(tables, e, dev), x, batch, gemb, zeros)
        return [zeros[id(p)] for p in params]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
