"""FFM -- counterpart of the reference's model/ffm.py:7-98."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import FieldSpec
from .._lib import FIELD_BAG, FIELD_ID_F32
from ._base import FeatureModel

# the 12 field-aware vectors in buffer order, and the reference's 15 dot products
# (model/ffm.py:62-80) as index pairs into that order
VECTORS = ("age_user", "age_item", "gender_user", "gender_item", "occupation_user", "occupation_item",
           "movie_user", "movie_item", "userid_user", "userid_item", "itemid_user", "itemid_item")
_PAIR_NAMES = (
    ("age_user", "gender_user"), ("age_user", "occupation_user"), ("age_item", "movie_user"),
    ("age_user", "userid_user"), ("age_item", "itemid_user"),
    ("gender_user", "occupation_user"), ("gender_item", "movie_user"),
    ("gender_user", "userid_user"), ("gender_item", "itemid_user"),
    ("occupation_item", "movie_user"), ("occupation_user", "userid_user"), ("occupation_item", "itemid_user"),
    ("movie_user", "userid_item"), ("movie_item", "itemid_item"),
    ("userid_item", "itemid_user"),
)
PAIRS = tuple((VECTORS.index(a), VECTORS.index(b)) for a, b in _PAIR_NAMES)
_SOURCE = {"age": (FIELD_BAG, 2, 1), "gender": (FIELD_BAG, 3, 2), "occupation": (FIELD_BAG, 5, 21),
           "movie": (FIELD_BAG, 26, 19), "userid": (FIELD_ID_F32, 0, 0), "itemid": (FIELD_ID_F32, 1, 0)}


class FFM(FeatureModel):
    """``FFM(num_feature, num_vector)``; ``forward(x: (B,45)) -> (B,1)``.
    ``num_users`` / ``num_items`` (keyword-only, reference values hard-coded at
    model/ffm.py:19-26) allow the larger id vocabularies of BASELINE configs[3]."""

    def __init__(self, num_feature: int, num_vector: int, *, num_users: int = 943, num_items: int = 1682):
        super().__init__()
        vocab = {"age": 1, "gender": 2, "occupation": 21, "movie": 19, "userid": num_users, "itemid": num_items}
        for name in VECTORS:
            setattr(self, name, nn.Embedding(vocab[name.split("_")[0]], num_vector))
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.linear = nn.Linear(num_feature, 1, True)
        for name in VECTORS + ("user", "item"):
            xavier_normal_(getattr(self, name).weight.data)

    def _params(self):
        return [getattr(self, n).weight for n in VECTORS] + [self.user.weight, self.item.weight,
                                                              self.linear.weight, self.linear.bias]

    def forward(self, feature_vector):
        return self._run_model(feature_vector, self._params())

    @staticmethod
    def _specs(tables, dim):
        specs = []
        for k, (name, table) in enumerate(zip(VECTORS, tables)):
            kind, col, bag = _SOURCE[name.split("_")[0]]
            specs.append(FieldSpec(kind, dim, k * dim, table=table, src_col=col, bag_size=bag))
        return specs

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables, (user1, item1, lin_w, lin_b) = params[:12], params[12:16]
        batch, dim = x.shape[0], tables[0].shape[1]
        emb = torch.empty((batch, 12 * dim), dtype=torch.float32, device=x.device)
        ops.embed_fwd(self._specs(tables, dim), x, batch, emb, self._flag)
        prob = torch.empty((batch, 1), dtype=torch.float32, device=x.device)
        ops.ffm_head_fwd(emb, 12, dim, PAIRS, x, user1, item1, lin_w, lin_b, prob, self._flag)
        return prob, (emb, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, prob = state
        tables, (user1, item1, lin_w, lin_b) = params[:12], params[12:16]
        batch, dim = x.shape[0], tables[0].shape[1]
        gemb = torch.empty_like(emb)
        zeros = ops.zero_grads(params)
        g_user1, g_item1, g_w, g_b = (zeros[id(t)] for t in (user1, item1, lin_w, lin_b))
        ops.ffm_head_bwd(emb, 12, dim, PAIRS, x, user1, item1, lin_w, lin_b, prob, gprob.view(batch, 1),
                         g_user1, g_item1, g_w, g_b, gemb)
        tgrads = zeros
        ops.embed_bwd(self._specs(tables, dim), x, batch, gemb, tgrads)
        return [tgrads[id(t)] for t in tables] + [g_user1, g_item1, g_w, g_b]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
