"""Deep Crossing -- counterpart of the reference's model/deepcrossing.py:8-92."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_RELU, ACT_SIGMOID, FieldSpec
from .._lib import FIELD_BAG, FIELD_DENSE, FIELD_ID_F32
from ._base import FeatureModel


class ResidualBlock(nn.Module):
    """parameter container of one residual unit ``relu(linear2(relu(linear1(x))) + x)``
    (reference model/deepcrossing.py:8-27)"""

    def __init__(self, hidden_unit, dim_stack):
        super().__init__()
        self.linear1 = nn.Linear(dim_stack, hidden_unit)
        self.linear2 = nn.Linear(hidden_unit, dim_stack)
        self.relu = nn.ReLU()


class DeepCrossing(FeatureModel):
    """``DeepCrossing(num_user, num_item, num_feature, hidden_units)``;
    ``forward(x: (B,45)) -> (B,1)``.  The stacking layer
    [user, item, age(1), gender, occupation, movie] is one embedding-stage launch
    into a (B, 5E+1) buffer whose rows are padded to a multiple of 4 floats; every
    residual add + ReLU is the epilogue of the block's second GEMM."""

    def __init__(self, num_user, num_item, num_feature, hidden_units):
        super().__init__()
        self.user_embedding = nn.Embedding(num_user, num_feature)
        self.item_embedding = nn.Embedding(num_item, num_feature)
        self.gender_embedding = nn.Embedding(2, num_feature)
        self.occupation_embedding = nn.Embedding(21, num_feature)
        self.movie_embedding = nn.Embedding(19, num_feature)
        for emb in (self.user_embedding, self.item_embedding, self.gender_embedding, self.occupation_embedding,
                    self.movie_embedding):
            xavier_normal_(emb.weight.data)
        dim_stack = num_feature * 5 + 1
        self.res_layers = nn.ModuleList([ResidualBlock(unit, dim_stack) for unit in hidden_units])
        self.linear = nn.Linear(dim_stack, 1)

    def _params(self):
        p = [e.weight for e in (self.user_embedding, self.item_embedding, self.gender_embedding,
                                self.occupation_embedding, self.movie_embedding)]
        p += [self.linear.weight, self.linear.bias]
        for blk in self.res_layers:
            p += [blk.linear1.weight, blk.linear1.bias, blk.linear2.weight, blk.linear2.bias]
        return p

    def forward(self, feature_vector):
        return self._run_model(feature_vector, self._params())

    @staticmethod
    def _specs(tables, e):
        user, item, gender, occ, movie = tables
        return [
            FieldSpec(FIELD_ID_F32, e, 0, table=user, src_col=0),
            FieldSpec(FIELD_ID_F32, e, e, table=item, src_col=1),
            FieldSpec(FIELD_DENSE, 1, 2 * e, src_col=2),
            FieldSpec(FIELD_BAG, e, 2 * e + 1, table=gender, src_col=3, bag_size=2),
            FieldSpec(FIELD_BAG, e, 3 * e + 1, table=occ, src_col=5, bag_size=21),
            FieldSpec(FIELD_BAG, e, 4 * e + 1, table=movie, src_col=26, bag_size=19),
        ]

    @staticmethod
    def _stack_buffer(batch, width, device):
        padded = (width + 3) // 4 * 4
        return torch.empty((batch, padded), dtype=torch.float32, device=device)[:, :width]

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables, (lin_w, lin_b) = params[:5], params[5:7]
        batch, e = x.shape[0], tables[0].shape[1]
        width = 5 * e + 1
        r = self._stack_buffer(batch, width, x.device)
        ops.embed_fwd(self._specs(tables, e), x, batch, r, self._flag)
        rs, hs = [r], []
        for k in range(len(self.res_layers)):
            w1, b1, w2, b2 = params[7 + 4 * k: 11 + 4 * k]
            h = ops.linear_fwd(rs[-1], self._aligned_weight(w1), b1, ACT_RELU)
            out = self._stack_buffer(batch, width, x.device)
            ops.linear_fwd(h, w2, b2, ACT_RELU, out=out, residual=rs[-1])
            hs.append(h)
            rs.append(out)
        prob = ops.linear_fwd(rs[-1], self._aligned_weight(lin_w), lin_b, ACT_SIGMOID)
        return prob, (rs, hs, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        rs, hs, prob = state
        tables, (lin_w, lin_b) = params[:5], params[5:7]
        batch, e = x.shape[0], tables[0].shape[1]
        width = 5 * e + 1
        zeros = ops.zero_grads(params)
        g_lin_w, g_lin_b = zeros[id(lin_w)], zeros[id(lin_b)]
        gr = self._stack_buffer(batch, width, x.device)
        ops.linear_bwd(rs[-1], self._aligned_weight(lin_w, refresh=False), prob, gprob, ACT_SIGMOID, gr, g_lin_w, g_lin_b)
        block_grads = []
        for k in range(len(self.res_layers) - 1, -1, -1):
            w1, b1, w2, b2 = params[7 + 4 * k: 11 + 4 * k]
            r_in, r_out, h = rs[k], rs[k + 1], hs[k]
            gw1, gb1, gw2, gb2 = (zeros[id(t)] for t in (w1, b1, w2, b2))
            gh = torch.empty_like(h)
            ops.linear_bwd(h, w2, r_out, gr, ACT_RELU, gh, gw2, gb2)            # through relu(linear2(h)+r)
            gr_in = self._stack_buffer(batch, width, x.device)
            ops.linear_bwd(r_in, self._aligned_weight(w1, refresh=False), h, gh, ACT_RELU, gr_in, gw1, gb1)  # relu(linear1(r))
            ops.act_bwd(r_out, gr, ACT_RELU, gr_in, accumulate=True)            # the skip connection
            gr = gr_in
            block_grads.append((gw1, gb1, gw2, gb2))
        tgrads = zeros
        ops.embed_bwd(self._specs(tables, e), x, batch, gr, tgrads)
        grads = [tgrads[id(t)] for t in tables] + [g_lin_w, g_lin_b]
        for bg in reversed(block_grads):
            grads += list(bg)
        return grads

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
