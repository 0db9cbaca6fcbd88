"""Deep & Cross (DCN) -- counterpart of the reference's model/deepcross.py:7-89."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID
from ._base import FeatureModel
from .deepcrossing import DeepCrossing


class CrossNetwork(nn.Module):
    """parameter container of the cross layers ``x_{l+1} = x0 * (W_l x_l) + b_l + x_l``
    (reference model/deepcross.py:7-18)"""

    def __init__(self, input_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.cross_weights = nn.ModuleList([nn.Linear(input_dim, input_dim, bias=False) for _ in range(num_layers)])
        self.cross_biases = nn.ParameterList([nn.Parameter(torch.zeros(input_dim)) for _ in range(num_layers)])


class DeepNetwork(nn.Module):
    """parameter container of the deep tower: Linear + ReLU after EVERY layer
    (reference model/deepcross.py:21-31)"""

    def __init__(self, input_dim, hidden_units):
        super().__init__()
        layers = []
        for a, b in zip([input_dim] + hidden_units[:-1], hidden_units):
            layers += [nn.Linear(a, b), nn.ReLU()]
        self.network = nn.Sequential(*layers)


class DeepCross(FeatureModel):
    """``DeepCross(num_users, num_items, cross_layers, deep_hidden_units, embedding_dim)``;
    ``forward(x: (B,45)) -> (B,1)``.

    One embedding-stage launch builds the (B, 5E+1) stack; every cross layer is a d x d GEMM
    on the matrix cores plus one streaming combine kernel (csrc/cross.hip); the last cross
    output and the last deep activation are written side by side into the operand of the
    output layer (no torch.cat)."""

    def __init__(self, num_users, num_items, cross_layers, deep_hidden_units, embedding_dim):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, embedding_dim)
        self.item_embedding = nn.Embedding(num_items, embedding_dim)
        self.gender_embedding = nn.Embedding(2, embedding_dim)
        self.occupation_embedding = nn.Embedding(21, embedding_dim)
        self.movie_embedding = nn.Embedding(19, embedding_dim)
        d = embedding_dim * 5 + 1
        self.cross_network = CrossNetwork(d, cross_layers)
        self.deep_network = DeepNetwork(d, list(deep_hidden_units))
        self.output_layer = nn.Linear(d + deep_hidden_units[-1], 1)
        for emb in (self.user_embedding, self.item_embedding, self.gender_embedding, self.occupation_embedding,
                    self.movie_embedding):
            xavier_normal_(emb.weight.data)

    def _deep_linears(self):
        return [m for m in self.deep_network.network if isinstance(m, nn.Linear)]

    def _params(self):
        p = [e.weight for e in (self.user_embedding, self.item_embedding, self.gender_embedding,
                                self.occupation_embedding, self.movie_embedding)]
        p += [lin.weight for lin in self.cross_network.cross_weights]
        p += list(self.cross_network.cross_biases)
        for lin in self._deep_linears():
            p += [lin.weight, lin.bias]
        p += [self.output_layer.weight, self.output_layer.bias]
        return p

    def forward(self, x):
        return self._run_model(x, self._params())

    def _split(self, params):
        nc, nd = self.cross_network.num_layers, len(self._deep_linears())
        tables = params[:5]
        cw, cb = params[5:5 + nc], params[5 + nc:5 + 2 * nc]
        deep = params[5 + 2 * nc:5 + 2 * nc + 2 * nd]
        out_w, out_b = params[-2:]
        return tables, cw, cb, [(deep[2 * k], deep[2 * k + 1]) for k in range(nd)], out_w, out_b

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables, cw, cb, deep, out_w, out_b = self._split(params)
        batch, e = x.shape[0], tables[0].shape[1]
        d, dev = 5 * e + 1, x.device
        hl = deep[-1][0].shape[0]
        x0 = self._padded_rows(batch, d, dev)
        ops.embed_fwd(DeepCrossing._specs(tables, e), x, batch, x0, self._flag)
        comb = self._padded_rows(batch, d + hl, dev)        # [x_L | deep_out]: operand of the output layer
        xs, us = [x0], []
        for l in range(len(cw)):
            u = ops.linear_fwd(xs[-1], self._aligned_weight(cw[l]), None, ACT_NONE, out=self._padded_rows(batch, d, dev))
            out = comb[:, :d] if l == len(cw) - 1 else self._padded_rows(batch, d, dev)
            ops.cross_fwd(x0, u, xs[-1], cb[l], out)
            us.append(u)
            xs.append(out)
        if not cw:
            comb[:, :d].copy_(x0)
        hs = [x0]
        for k, (w, b) in enumerate(deep):
            out = comb[:, d:] if k == len(deep) - 1 else None
            hs.append(ops.linear_fwd(hs[-1], self._aligned_weight(w) if k == 0 else w, b, ACT_RELU, out=out))
        prob = ops.linear_fwd(comb, self._aligned_weight(out_w), out_b, ACT_SIGMOID)
        return prob, (xs, us, hs, comb, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        xs, us, hs, comb, prob = state
        tables, cw, cb, deep, out_w, out_b = self._split(params)
        batch, e = x.shape[0], tables[0].shape[1]
        d, dev = 5 * e + 1, x.device
        hl = deep[-1][0].shape[0]
        zeros = ops.zero_grads(params)
        gcomb = self._padded_rows(batch, d + hl, dev)
        ops.linear_bwd(comb, self._aligned_weight(out_w, refresh=False), prob, gprob, ACT_SIGMOID, gcomb,
                       zeros[id(out_w)], zeros[id(out_b)])
        # deep tower: its input gradient starts the accumulator of d loss / d x0
        gh = gcomb[:, d:]
        for k in range(len(deep) - 1, -1, -1):
            w, b = deep[k]
            gin = self._padded_rows(batch, hs[k].shape[1], dev)
            ops.linear_bwd(hs[k], self._aligned_weight(w, refresh=False) if k == 0 else w, hs[k + 1], gh, ACT_RELU, gin,
                           zeros[id(w)], zeros[id(b)])
            gh = gin
        gx0 = gh
        # cross layers, last to first: gx holds d loss / d x_{l+1}, becomes d loss / d x_l in place
        gx = gcomb[:, :d]
        for l in range(len(cw) - 1, -1, -1):
            gu = self._padded_rows(batch, d, dev)
            ops.cross_bwd(xs[0], us[l], gx, gu, gx0, zeros[id(cb[l])])
            ops.linear_bwd(xs[l], self._aligned_weight(cw[l], refresh=False), None, gu, ACT_NONE, gx, zeros[id(cw[l])],
                           None, accumulate_gx=True)
        ops.act_bwd(gx, gx, ACT_NONE, gx0, accumulate=True)   # x_0 is x0 itself: gx0 += gx
        ops.embed_bwd(DeepCrossing._specs(tables, e), x, batch, gx0, zeros)
        return [zeros[id(p)] for p in params]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
