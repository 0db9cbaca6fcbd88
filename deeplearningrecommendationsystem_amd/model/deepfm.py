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


def field_vocabs(num_fields, vocab):
    """keyword-only generalisation shared by DeepFM / PNN: ``num_fields`` single-id fields with ``vocab`` rows
    each (an int, or one int per field)"""
    if vocab is None:
        raise ValueError("num_fields=... needs vocab=... (rows per field: an int or one per field)")
    vocabs = [int(vocab)] * num_fields if isinstance(vocab, int) else [int(v) for v in vocab]
    if len(vocabs) != num_fields or num_fields < 2 or num_fields > 32 or min(vocabs) < 1:
        raise ValueError(f"expected 2..32 fields with one positive vocabulary size each, got {num_fields} / {vocab}")
    return vocabs


class FieldsInput:
    """input handling of the N-id-field models: ``x`` is (B, F) int64 ids, or float32 carrying ids as the
    reference's feature matrix does in its two id columns (converted like ``x[:, c].long()``)"""

    @staticmethod
    def ids(x, num_fields):
        if x.dim() != 2 or x.shape[1] != num_fields or x.dtype not in (torch.int64, torch.float32):
            raise ValueError(f"expected a (B,{num_fields}) int64 (or float32) id matrix, got {tuple(x.shape)} {x.dtype}")
        x = x if x.dtype == torch.int64 else x.long()
        return x if x.stride(1) == 1 else x.contiguous()


class DeepFM(FeatureModel):
    """``DeepFM(num_users, num_items, hidden_units, embedding_dim)``;
    ``forward(x: (B,45)) -> (B,1)``.

    Buffers: emb (B,6E) feeds both the deep MLP and the FM term; the last deep
    layer and the wide+FM kernel write the two columns of ``comb`` (B,1+H_last)
    that ``output`` consumes (reference's torch.cat at deepfm.py:80).

    Keyword-only generalisation (BASELINE configs[2], "26 fields x 1e6 vocab"; the reference hard-codes its
    six fields, deepfm.py:14-17): ``DeepFM(None, None, hidden_units, embedding_dim, num_fields=F, vocab=V)``
    builds F single-id fields -- ``embeddings.f`` (V, E) and first-order ``first_order.f`` (V, 1) per field,
    ``first_order_bias`` in place of the ``wide`` Linear (no dense columns) -- and ``forward(x: (B,F) ids)``
    runs the same deep / FM / output structure with the gather fused into the FM kernel
    (``ctr_fields_fm_fwd``).  Defaults leave the reference model unchanged."""

    def __init__(self, num_users, num_items, hidden_units, embedding_dim, *, num_fields=None, vocab=None):
        super().__init__()
        self.num_fields = num_fields
        if num_fields is not None:
            self.vocabs = field_vocabs(num_fields, vocab)
            self.embeddings = nn.ModuleList([nn.Embedding(v, embedding_dim) for v in self.vocabs])
            self.first_order = nn.ModuleList([nn.Embedding(v, 1) for v in self.vocabs])
            self.first_order_bias = nn.Parameter(torch.zeros(1))
            self.linear = nn.Linear(embedding_dim * num_fields, hidden_units[0])
            self.dnn_network = nn.ModuleList([nn.Linear(a, b) for a, b in zip(hidden_units[:-1], hidden_units[1:])])
            self.relu = nn.ReLU()
            self.output = nn.Linear(2, 1)
            for emb in list(self.embeddings) + list(self.first_order):
                xavier_normal_(emb.weight.data)
            return
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
        if self.num_fields is not None:
            p = [e.weight for e in self.embeddings] + [e.weight for e in self.first_order]
            p += [self.first_order_bias, self.output.weight, self.output.bias, self.linear.weight, self.linear.bias]
            for lin in self.dnn_network:
                p += [lin.weight, lin.bias]
            return p
        p = [e.weight for e in (self.user_embedding, self.item_embedding, self.age_embedding,
                                self.gender_embedding, self.occupation_embedding, self.movie_embedding)]
        p += [self.user.weight, self.item.weight, self.wide.weight, self.wide.bias,
              self.output.weight, self.output.bias, self.linear.weight, self.linear.bias]
        for lin in self.dnn_network:
            p += [lin.weight, lin.bias]
        return p

    def sparse_ids(self, inputs):
        """sparse mode: the id tables and their first-order (V,1) companions"""
        if self.num_fields is not None:
            nf = self.num_fields
            cols = [[] if inputs is None else [inputs[0][:, f]] for f in range(nf)]
            return {k: cols[k % nf] for k in range(2 * nf)}
        cols = [[] if inputs is None else [inputs[0][:, c]] for c in (0, 1)]   # float id columns of the (B,45) matrix
        return {0: cols[0], 1: cols[1], 6: cols[0], 7: cols[1]}

    def forward(self, x):
        if self.num_fields is not None:
            return self._run_fields(FieldsInput.ids(x, self.num_fields), self._params())
        return self._run_model(x, self._params())

    def _layers(self, params, first=12):
        layers = [Layer(params[first], params[first + 1], ACT_NONE)]
        for k in range(len(self.dnn_network)):
            layers.append(Layer(params[first + 2 + 2 * k], params[first + 3 + 2 * k], ACT_RELU))
        return layers

    # ---- N id fields: gather fused with the FM term (csrc/fields.hip)
    def _fields_forward(self, idx, params):
        nf = self.num_fields
        tables, firsts = params[:nf], params[nf:2 * nf]
        bias, out_w, out_b = params[2 * nf:2 * nf + 3]
        batch, dim = idx.shape[0], tables[0].shape[1]
        layers = self._layers(params, 2 * nf + 3)
        emb = torch.empty((batch, nf * dim), dtype=torch.float32, device=idx.device)
        comb = torch.empty((batch, 1 + layers[-1].weight.shape[0]), dtype=torch.float32, device=idx.device)
        ops.fields_fm_fwd(idx, tables, firsts, bias, emb, comb[:, 0:1], self._flag)
        acts = ops.mlp_fwd(emb, layers, last_out=comb[:, 1:])
        prob = ops.linear_fwd(comb, out_w, out_b, ACT_SIGMOID)
        return prob, (emb, comb, acts, prob)

    def _fields_backward(self, state, idx, params, gprob):
        nf = self.num_fields
        emb, comb, acts, prob = state
        tables, firsts = params[:nf], params[nf:2 * nf]
        bias, out_w, out_b = params[2 * nf:2 * nf + 3]
        layers = self._layers(params, 2 * nf + 3)
        zeros = ops.zero_grads(params)
        gcomb = torch.empty_like(comb)
        ops.linear_bwd(comb, out_w, prob, gprob, ACT_SIGMOID, gcomb, zeros[id(out_w)], zeros[id(out_b)])
        gemb = torch.empty_like(emb)
        layer_grads, _ = ops.mlp_bwd(acts, layers, gcomb[:, 1:], gemb, zeros=zeros)
        ops.fields_fm_bwd(idx, self.vocabs, tables[0].shape[1], emb, gemb, gcomb[:, 0:1],
                          [zeros[id(t)] for t in tables], [zeros[id(t)] for t in firsts], zeros[id(bias)])
        grads = [zeros[id(t)] for t in tables] + [zeros[id(t)] for t in firsts]
        grads += [zeros[id(bias)], zeros[id(out_w)], zeros[id(out_b)]]
        for gw, gb in layer_grads:
            grads += [gw, gb]
        return grads

    def run_forward(self, inputs, params):
        (x,) = inputs
        if self.num_fields is not None:
            return self._fields_forward(x, params)
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
        if self.num_fields is not None:
            return self._fields_backward(state, x, params, gprob)
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
