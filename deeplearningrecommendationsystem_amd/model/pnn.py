"""PNN -- counterpart of the reference's model/pnn.py:8-143 (DNN, ProductLayers, PNN)."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, Layer
from ._base import FeatureModel
from .deepfm import FieldsInput, field_vocabs, six_field_specs
from .._lib import FIELD_ID_I64


class DNN(nn.Module):
    """parameter container for ``Linear+ReLU`` per consecutive pair of
    ``hidden_units`` (reference model/pnn.py:8-23); the math runs in PNN's
    fused forward/backward."""

    def __init__(self, hidden_units):
        super().__init__()
        self.dnn_network = nn.ModuleList([nn.Linear(a, b) for a, b in zip(hidden_units[:-1], hidden_units[1:])])
        self.relu = nn.ReLU()


class ProductLayers(nn.Module):
    """parameter container of the product layer (reference model/pnn.py:27-79)"""

    def __init__(self, num_feature, embed_dim, hidden_units, model="in"):
        super().__init__()
        self.model = model
        self.linear1 = nn.Linear(num_feature * embed_dim, hidden_units[0])
        if model == "in":
            self.linear2 = nn.Linear(int(num_feature * (num_feature - 1) / 2), hidden_units[0])
        elif model == "out":
            self.linear2 = nn.Linear(embed_dim, hidden_units[0])


class PNN(FeatureModel):
    """``PNN(embed_dim, hidden_units, model="in")``; ``forward(x: (B,45)) -> (B,1)``.

    inner mode: h0 = linear1(emb) + linear2(allpairs(emb)); the second GEMM takes
    the first one's output as its fused residual.  outer mode (reference
    pnn.py:67-72) reduces over the batch and only broadcasts when B == embed_dim;
    it is built from the same GEMM kernels (p = S^T S is the dW form)."""

    def __init__(self, embed_dim, hidden_units, model="in", *, num_users=943, num_items=1682, num_fields=None,
                 vocab=None):
        """keyword-only generalisation (BASELINE configs[2], "26 fields x 1e6 vocab"; the reference hard-codes six
        fields, pnn.py:87-92): ``num_fields=F, vocab=V`` builds F single-id fields ``embeddings.f`` (V, E) and
        ``forward(x: (B,F) ids)``; the product layer then has F(F-1)/2 inner products (325 at F = 26).  Inner
        mode only.  Defaults leave the reference model unchanged."""
        super().__init__()
        self.num_fields = num_fields
        if num_fields is not None:
            if model != "in":
                raise ValueError("the N-field generalisation implements the inner-product mode")
            self.vocabs = field_vocabs(num_fields, vocab)
            self.embeddings = nn.ModuleList([nn.Embedding(v, embed_dim) for v in self.vocabs])
            for emb in self.embeddings:
                xavier_normal_(emb.weight.data)
            self.product = ProductLayers(num_fields, embed_dim, hidden_units, model)
            self.dnn = DNN(hidden_units)
            self.output = nn.Linear(hidden_units[-1], 1)
            return
        self.user_embed = nn.Embedding(num_users, embed_dim)
        self.item_embed = nn.Embedding(num_items, embed_dim)
        self.age_embed = nn.Embedding(1, embed_dim)
        self.gender_embed = nn.Embedding(2, embed_dim)
        self.occupation_embed = nn.Embedding(21, embed_dim)
        self.movie_embed = nn.Embedding(19, embed_dim)
        for emb in (self.user_embed, self.item_embed, self.age_embed, self.gender_embed, self.occupation_embed,
                    self.movie_embed):
            xavier_normal_(emb.weight.data)
        self.product = ProductLayers(6, embed_dim, hidden_units, model)
        self.dnn = DNN(hidden_units)
        self.output = nn.Linear(hidden_units[-1], 1)
        if model == "out":
            # constant selector: S = sum_f v_f as one exact GEMM (products with 0/1)
            self.register_buffer("_sum_selector", torch.eye(embed_dim).repeat(1, 6), persistent=False)

    def _params(self):
        if self.num_fields is not None:
            p = [e.weight for e in self.embeddings]
        else:
            p = [e.weight for e in (self.user_embed, self.item_embed, self.age_embed, self.gender_embed,
                                    self.occupation_embed, self.movie_embed)]
        p += [self.product.linear1.weight, self.product.linear1.bias, self.product.linear2.weight,
              self.product.linear2.bias, self.output.weight, self.output.bias]
        for lin in self.dnn.dnn_network:
            p += [lin.weight, lin.bias]
        return p

    def sparse_ids(self, inputs):
        if self.num_fields is not None:
            return {f: ([] if inputs is None else [inputs[0][:, f]]) for f in range(self.num_fields)}
        return {c: ([] if inputs is None else [inputs[0][:, c]]) for c in (0, 1)}

    def forward(self, x):
        if self.num_fields is not None:
            return self._run_fields(FieldsInput.ids(x, self.num_fields), self._params())
        return self._run_model(x, self._params())

    def _nvec(self):
        return 6 if self.num_fields is None else self.num_fields

    def _tail(self, params):
        n = self._nvec()
        layers = [Layer(params[n + 6 + 2 * k], params[n + 7 + 2 * k], ACT_RELU) for k in range(len(self.dnn.dnn_network))]
        return layers + [Layer(params[n + 4], params[n + 5], ACT_SIGMOID)]

    def _specs(self, x, tables, dim):
        """the embedding stage's field list: the reference's six fields out of the (B,45) matrix, or F id columns"""
        if self.num_fields is None:
            return six_field_specs(tables, dim), x
        nf = self.num_fields
        return [ops.FieldSpec(FIELD_ID_I64, dim, f * dim, table=tables[f], idx=x[:, f], idx_stride=x.stride(0))
                for f in range(nf)], None

    def run_forward(self, inputs, params):
        (x,) = inputs
        n = self._nvec()
        npairs = n * (n - 1) // 2
        tables = params[:n]
        w1, b1, w2, b2 = params[n:n + 4]
        batch, dim = x.shape[0], tables[0].shape[1]
        emb = torch.empty((batch, n * dim), dtype=torch.float32, device=x.device)
        specs, xin = self._specs(x, tables, dim)
        ops.embed_fwd(specs, xin, batch, emb, self._flag)
        if self.product.model == "in":
            prod = ops.allpairs_fwd(emb, n, dim, out=self._padded_rows(batch, npairs, x.device))
            lz = ops.linear_fwd(emb, w1, b1)
            h0 = ops.linear_fwd(prod, self._aligned_weight(w2), b2, residual=lz)
            extra = (prod,)
        else:
            if batch != dim:
                raise RuntimeError(f"PNN outer product: lz (1,{batch},H) and lp ({dim},H) only broadcast when "
                                   f"batch == embed_dim (reference model/pnn.py:75-77)")
            s = ops.linear_fwd(emb, self._sum_selector, None)                       # (B,E) = sum_f v_f
            prod = torch.zeros((dim, dim), dtype=torch.float32, device=x.device)
            ops.linear_bwd(s, prod, None, s, ACT_NONE, None, prod, None)           # p = S^T S
            lp = ops.linear_fwd(prod, w2, b2)                                       # (E,H0)
            h0 = ops.linear_fwd(emb, w1, b1, residual=lp)
            extra = (prod, s)
        acts = ops.mlp_fwd(h0, self._tail(params))
        return acts[-1].view(-1, 1), (emb, acts, extra)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        emb, acts, extra = state
        n = self._nvec()
        npairs = n * (n - 1) // 2
        tables = params[:n]
        w1, b1, w2, b2 = params[n:n + 4]
        batch, dim = x.shape[0], tables[0].shape[1]
        tail = self._tail(params)
        zeros = ops.zero_grads(params)
        tail_grads, gh0 = ops.mlp_bwd(acts, tail, gprob.view(batch, 1), None, zeros=zeros)
        gw1, gb1, gw2, gb2 = (zeros[id(t)] for t in (w1, b1, w2, b2))
        gemb = torch.empty_like(emb)
        ops.linear_bwd(emb, w1, None, gh0, ACT_NONE, gemb, gw1, gb1)
        if self.product.model == "in":
            (prod,) = extra
            gprod = self._padded_rows(batch, npairs, x.device)
            ops.linear_bwd(prod, self._aligned_weight(w2, refresh=False), None, gh0, ACT_NONE, gprod, gw2, gb2)
            ops.allpairs_bwd(emb, n, dim, gprod, gemb, accumulate=True)
        else:
            prod, s = extra
            gprod = torch.empty_like(prod)
            ops.linear_bwd(prod, w2, None, gh0, ACT_NONE, gprod, gw2, gb2)          # glp = gh0 as (E,H0)
            gs = ops.linear_fwd(s, gprod, None)                                     # S gp^T
            ops.linear_bwd(s, gprod, None, s, ACT_NONE, gs, None, None, accumulate_gx=True)   # += S gp
            ops.linear_bwd(emb, self._sum_selector, None, gs, ACT_NONE, gemb, None, None, accumulate_gx=True)
        tgrads = zeros
        specs, xin = self._specs(x, tables, dim)
        ops.embed_bwd(specs, xin, batch, gemb, tgrads)
        grads = [tgrads[id(t)] for t in tables] + [gw1, gb1, gw2, gb2, tail_grads[-1][0], tail_grads[-1][1]]
        for gw, gb in tail_grads[:-1]:
            grads += [gw, gb]
        return grads

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
