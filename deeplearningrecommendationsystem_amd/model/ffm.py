"""FFM -- counterpart of the reference's model/ffm.py:7-98."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import FieldSpec
from .._lib import FIELD_BAG, FIELD_ID_F32, FIELD_ID_I64
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


SHARDED = ("userid_user", "userid_item", "itemid_user", "itemid_item")  # the tables indexed by user / item id


class FFM(FeatureModel):
    """``FFM(num_feature, num_vector)``; ``forward(x: (B,45)) -> (B,1)``.
    Keyword-only extensions: ``num_users`` / ``num_items`` (reference values hard-coded at
    model/ffm.py:19-26) allow the larger id vocabularies of BASELINE configs[3];
    ``sharded=True`` deals the rows of the four field-aware id tables round-robin to the ranks
    of ``group`` (dist.ShardedEmbedding: lookups through all-to-all)."""

    def __init__(self, num_feature: int, num_vector: int, *, num_users: int = 943, num_items: int = 1682,
                 sharded: bool = False, group=None):
        super().__init__()
        self.sharded = bool(sharded)
        vocab = {"age": 1, "gender": 2, "occupation": 21, "movie": 19, "userid": num_users, "itemid": num_items}
        for name in VECTORS:
            if sharded and name in SHARDED:
                from ..dist import ShardedEmbedding
                setattr(self, name, ShardedEmbedding(vocab[name.split("_")[0]], num_vector, group=group))
            else:
                setattr(self, name, nn.Embedding(vocab[name.split("_")[0]], num_vector))
                xavier_normal_(getattr(self, name).weight.data)
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.linear = nn.Linear(num_feature, 1, True)
        for name in ("user", "item"):
            xavier_normal_(getattr(self, name).weight.data)

    fused_forward = True   # False: the two-launch forward of round 1 (embed_fwd + ffm_head_fwd), kept for A/B

    def _params(self):
        return [getattr(self, n).weight for n in VECTORS] + [self.user.weight, self.item.weight,
                                                              self.linear.weight, self.linear.bias]

    def forward(self, feature_vector):
        params = self._params()
        if self.sharded:
            # rows of this batch through the exchange; the kernels then see each of them as a
            # (B, k) table indexed 0..B-1, and its gradient flows back through the exchange
            self._need_device(feature_vector, params[12])
            uid, iid = feature_vector[:, 0].long(), feature_vector[:, 1].long()
            # every table's rows are requested first and waited for afterwards: table k+1's all-to-all runs on the
            # collective stream while table k's rows are put back into batch order.  The ids are a temporary of the
            # feature matrix: its identity / version key the exchange plan, which the two field-aware tables of an id
            # column share (one id exchange per column)
            flights = {}
            for k, name in enumerate(VECTORS):
                if name in SHARDED:
                    user = name.startswith("userid")
                    flights[k] = getattr(self, name).start(uid if user else iid, plan_key=(feature_vector, 0 if user else 1))
            for k, flight in flights.items():
                params[k] = flight.wait()
        return self._run_model(feature_vector, params)

    def _specs(self, tables, dim):
        specs = []
        for k, (name, table) in enumerate(zip(VECTORS, tables)):
            kind, col, bag = _SOURCE[name.split("_")[0]]
            if self.sharded and name in SHARDED:
                pos = torch.arange(table.shape[0], device=table.device, dtype=torch.int64)
                specs.append(FieldSpec(FIELD_ID_I64, dim, k * dim, table=table, idx=pos))
            else:
                specs.append(FieldSpec(kind, dim, k * dim, table=table, src_col=col, bag_size=bag))
        return specs

    def run_forward(self, inputs, params):
        (x,) = inputs
        tables, (user1, item1, lin_w, lin_b) = params[:12], params[12:16]
        batch, dim = x.shape[0], tables[0].shape[1]
        emb = torch.empty((batch, 12 * dim), dtype=torch.float32, device=x.device)
        prob = torch.empty((batch, 1), dtype=torch.float32, device=x.device)
        if self.fused_forward and not self.sharded and dim in (8, 16, 32, 64):
            # gather, bags, the 15 dots and the head in one launch (csrc/ffm_fused.hip)
            ops.ffm_fused_fwd(x, tables, user1, item1, lin_w, lin_b, emb, prob, self._flag)
            return prob, (emb, prob)
        ops.embed_fwd(self._specs(tables, dim), x, batch, emb, self._flag)
        ops.ffm_head_fwd(emb, 12, dim, PAIRS, x, user1, item1, lin_w, lin_b, prob, self._flag)
        return prob, (emb, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, prob = state
        tables, (user1, item1, lin_w, lin_b) = params[:12], params[12:16]
        batch, dim = x.shape[0], tables[0].shape[1]
        gemb = torch.empty_like(emb)
        if self.sharded:
            # exchanged rows are activations: their gradients stay out of the flat buffer that the
            # data-parallel all-reduce works on
            rows = [k for k, name in enumerate(VECTORS) if name in SHARDED]
            zeros = ops.zero_grads([p for k, p in enumerate(params) if k not in rows])
            for k in rows:
                zeros[id(params[k])] = torch.zeros_like(params[k])
        else:
            zeros = ops.zero_grads(params)
        g_user1, g_item1, g_w, g_b = (zeros[id(t)] for t in (user1, item1, lin_w, lin_b))
        if self.fused_forward and not self.sharded and dim in (8, 16, 32, 64):
            ops.ffm_fused_bwd(x, emb, user1.shape[0], item1.shape[0], lin_w, prob, gprob.view(batch, 1), g_user1, g_item1,
                              g_w, g_b, gemb)
        else:
            ops.ffm_head_bwd(emb, 12, dim, PAIRS, x, user1, item1, lin_w, lin_b, prob, gprob.view(batch, 1),
                             g_user1, g_item1, g_w, g_b, gemb)
        tgrads = zeros
        ops.embed_bwd(self._specs(tables, dim), x, batch, gemb, tgrads)
        return [tgrads[id(t)] for t in tables] + [g_user1, g_item1, g_w, g_b]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
