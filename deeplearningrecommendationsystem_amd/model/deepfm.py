"""DeepFM -- counterpart of the reference's model/deepfm.py:8-95."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, FieldSpec, Layer
from .._lib import FIELD_BAG, FIELD_ID_F32
from ._base import FeatureModel


def six_field_specs(tables, dim):
    """[user, item, age, gender, occupation, movie] vectors at columns f*dim
    (reference model/deepfm.py:45-54, model/pnn.py:113-121)"""
    user, item, age, gender, occ, movie = tables
    return [
        FieldSpec(FIELD_ID_F32, dim, 0 * dim, table=user, src_col=0),
        FieldSpec(FIELD_ID_F32, dim, 1 * dim, table=item, src_col=1),
        FieldSpec(FIELD_BAG, dim, 2 * dim, table=age, src_col=2, bag_size=1),
        FieldSpec(FIELD_BAG, dim, 3 * dim, table=gender, src_col=3, bag_size=2),
        FieldSpec(FIELD_BAG, dim, 4 * dim, table=occ, src_col=5, bag_size=21),
        FieldSpec(FIELD_BAG, dim, 5 * dim, table=movie, src_col=26, bag_size=19),
    ]


class DeepFM(FeatureModel):
    """``DeepFM(num_users, num_items, hidden_units, embedding_dim)``;
    ``forward(x: (B,45)) -> (B,1)``.

    Buffers: emb (B,6E) feeds both the deep MLP and the FM term; the last deep
    layer and the wide+FM kernel write the two columns of ``comb`` (B,1+H_last)
    that ``output`` consumes (reference's torch.cat at deepfm.py:80)."""

    def __init__(self, num_users, num_items, hidden_units, embedding_dim):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, embedding_dim)
        self.item_embedding = nn.Embedding(num_items, embedding_dim)
        self.age_embedding = nn.Embedding(1, embedding_dim)
        self.gender_embedding = nn.Embedding(2, embedding_dim)
        self.occupation_embedding = nn.Embedding(21, embedding_dim)
        self.movie_embedding = nn.Embedding(19, embedding_dim)
        self.linear = nn.Linear(embedding_dim * 6, hidden_units[0])
        self.dnn_network = nn.ModuleList([nn.Linear(a, b) for a, b in zip(hidden_units[:-1], hidden_units[1:])])
        self.relu = nn.ReLU()
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.wide = nn.Linear(1 + 2 + 21 + 19, 1)
        self.output = nn.Linear(2, 1)
        for emb in (self.user_embedding, self.item_embedding, self.age_embedding, self.gender_embedding,
                    self.occupation_embedding, self.movie_embedding, self.user, self.item):
            xavier_normal_(emb.weight.data)

    # parameter order handed to the autograd node
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

    def _layers(self, params):
        layers = [Layer(params[12], params[13], ACT_NONE)]
        for k in range(len(self.dnn_network)):
            layers.append(Layer(params[14 + 2 * k], params[15 + 2 * k], ACT_RELU))
        return layers

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables = params[:6]
        user1, item1, wide_w, wide_b, out_w, out_b = params[6:12]
        batch, dim = x.shape[0], tables[0].shape[1]
        emb = torch.empty((batch, 6 * dim), dtype=torch.float32, device=x.device)
        ops.embed_fwd(six_field_specs(tables, dim), x, batch, emb, self._flag)
        layers = self._layers(params)
        comb = torch.empty((batch, 1 + layers[-1].weight.shape[0]), dtype=torch.float32, device=x.device)
        acts = ops.mlp_fwd(emb, layers, last_out=comb[:, 1:])
        ops.fm_wide_fwd(emb, 6, dim, x, user1, item1, wide_w, wide_b, comb[:, 0:1], self._flag)
        prob = ops.linear_fwd(comb, out_w, out_b, ACT_SIGMOID)
        return prob, (emb, comb, acts, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, comb, acts, prob = state
        tables = params[:6]
        user1, item1, wide_w, wide_b, out_w, out_b = params[6:12]
        batch, dim = x.shape[0], tables[0].shape[1]
        layers = self._layers(params)
        zeros = ops.zero_grads(params)
        gcomb = torch.empty_like(comb)
        g_out_w, g_out_b = zeros[id(out_w)], zeros[id(out_b)]
        ops.linear_bwd(comb, out_w, prob, gprob, ACT_SIGMOID, gcomb, g_out_w, g_out_b)
        gemb = torch.empty_like(emb)
        layer_grads, _ = ops.mlp_bwd(acts, layers, gcomb[:, 1:], gemb, zeros=zeros)
        g_user1, g_item1, g_wide_w, g_wide_b = (zeros[id(t)] for t in (user1, item1, wide_w, wide_b))
        ops.fm_wide_bwd(emb, 6, dim, x, user1, item1, wide_w, wide_b, gcomb[:, 0:1], g_user1, g_item1,
                        g_wide_w, g_wide_b, gemb, accumulate=True)
        tgrads = zeros
        ops.embed_bwd(six_field_specs(tables, dim), x, batch, gemb, tgrads)
        grads = [tgrads[id(t)] for t in tables] + [g_user1, g_item1, g_wide_w, g_wide_b, g_out_w, g_out_b]
        for gw, gb in layer_grads:
            grads += [gw, gb]
        return grads

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
