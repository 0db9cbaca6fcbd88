"""NFM -- counterpart of the reference's model/nfm.py:8-84."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID
from ._base import FeatureModel
from .deepfm import six_field_specs


class NFM(FeatureModel):
    """``NFM(num_users, num_items, hidden_units, embedding_dim)``; ``forward(x: (B,45)) -> (B,1)``.

    The six vectors of DeepFM (one embedding-stage launch) -> bi-interaction pooling
    ``sum_{i<j} f_i * f_j`` (csrc/interact.hip) -> ``linear`` (no activation, nfm.py:62) -> Linear+ReLU
    per pair; wide part from the DeepFM wide kernel on a single vector (FM term identically
    zero); both write the columns of the (B, 1+H_last) operand of ``output``."""

    def __init__(self, num_users, num_items, hidden_units, embedding_dim):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, embedding_dim)
        self.item_embedding = nn.Embedding(num_items, embedding_dim)
        self.age_embedding = nn.Embedding(1, embedding_dim)
        self.gender_embedding = nn.Embedding(2, embedding_dim)
        self.occupation_embedding = nn.Embedding(21, embedding_dim)
        self.movie_embedding = nn.Embedding(19, embedding_dim)
        self.linear = nn.Linear(embedding_dim, hidden_units[0])
        self.dnn_network = nn.ModuleList([nn.Linear(a, b) for a, b in zip(hidden_units[:-1], hidden_units[1:])])
        self.relu = nn.ReLU()
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.wide = nn.Linear(1 + 2 + 21 + 19, 1)
        self.output = nn.Linear(2, 1)
        for emb in (self.user_embedding, self.item_embedding, self.age_embedding, self.gender_embedding,
                    self.occupation_embedding, self.movie_embedding, self.user, self.item):
            xavier_normal_(emb.weight.data)

    def _params(self):
        p = [e.weight for e in (self.user_embedding, self.item_embedding, self.age_embedding,
                                self.gender_embedding, self.occupation_embedding, self.movie_embedding)]
        p += [self.user.weight, self.item.weight, self.wide.weight, self.wide.bias,
              self.output.weight, self.output.bias, self.linear.weight, self.linear.bias]
        for lin in self.dnn_network:
            p += [lin.weight, lin.bias]
        return p

    def forward(self, x):
        return self._run_model(x, self._params())

    def _deep(self, params):
        layers = [(params[12], params[13], ACT_NONE)]
        for k in range(len(self.dnn_network)):
            layers.append((params[14 + 2 * k], params[15 + 2 * k], ACT_RELU))
        return layers

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables = params[:6]
        user1, item1, wide_w, wide_b, out_w, out_b = params[6:12]
        batch, e, dev = x.shape[0], tables[0].shape[1], x.device
        emb = torch.empty((batch, 6 * e), dtype=torch.float32, device=dev)
        ops.embed_fwd(six_field_specs(tables, e), x, batch, emb, self._flag)
        cross = self._padded_rows(batch, e, dev)
        ops.biinteract_fwd(emb, 6, e, cross)
        deep = self._deep(params)
        comb = torch.empty((batch, 1 + deep[-1][0].shape[0]), dtype=torch.float32, device=dev)
        hs = [cross]
        for k, (w, b, act) in enumerate(deep):
            out = comb[:, 1:] if k == len(deep) - 1 else None
            hs.append(ops.linear_fwd(hs[-1], self._aligned_weight(w) if k == 0 else w, b, act, out=out))
        ops.fm_wide_fwd(emb[:, :e], 1, e, x, user1, item1, wide_w, wide_b, comb[:, 0:1], self._flag)
        prob = ops.linear_fwd(comb, out_w, out_b, ACT_SIGMOID)
        return prob, (emb, hs, comb, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, hs, comb, prob = state
        tables = params[:6]
        user1, item1, wide_w, wide_b, out_w, out_b = params[6:12]
        batch, e, dev = x.shape[0], tables[0].shape[1], x.device
        deep = self._deep(params)
        zeros = ops.zero_grads(params)
        gcomb = torch.empty_like(comb)
        ops.linear_bwd(comb, out_w, prob, gprob, ACT_SIGMOID, gcomb, zeros[id(out_w)], zeros[id(out_b)])
        gh = gcomb[:, 1:]
        for k in range(len(deep) - 1, -1, -1):
            w, b, act = deep[k]
            gin = self._padded_rows(batch, hs[k].shape[1], dev)
            ops.linear_bwd(hs[k], self._aligned_weight(w, refresh=False) if k == 0 else w, hs[k + 1], gh, act, gin,
                           zeros[id(w)], zeros[id(b)])
            gh = gin
        gemb = torch.empty_like(emb)
        ops.biinteract_bwd(emb, 6, e, gh, gemb, accumulate=False)
        ops.fm_wide_bwd(emb[:, :e], 1, e, x, user1, item1, wide_w, wide_b, gcomb[:, 0:1], zeros[id(user1)],
                        zeros[id(item1)], zeros[id(wide_w)], zeros[id(wide_b)], None, accumulate=False)
        ops.embed_bwd(six_field_specs(tables, e), x, batch, gemb, zeros)
        return [zeros[id(p)] for p in params]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
