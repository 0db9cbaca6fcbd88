"""DIEN -- counterpart of the reference's model/dien.py:8-81."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, Layer
from .din import (SequenceModel, _zero_grads, attention_layers, fc_layers, fold_attention_weight,
                  unfold_attention_grad)

# the GRU kernels form the input projection themselves (ctr_gru_fused_fwd / _bwd); CTR_DIEN_FUSED_GRU=0 keeps the
# gi GEMM + recurrence + three gradient GEMMs for A/B
FUSED_GRU = os.environ.get("CTR_DIEN_FUSED_GRU", "1") != "0"


class DIN(nn.Module):
    """parameter container of DIEN's attention unit (reference model/dien.py:8-39:
    attention MLP 3E -> 64 -> 32 -> 1, returns the un-summed weighted history)"""

    def __init__(self, num_items, embed_size, table=None):
        super().__init__()
        if table is None:
            table = nn.Embedding(num_items, embed_size)
            xavier_normal_(table.weight.data)
        self.item_embedding = table
        self.attention = nn.Sequential(nn.Linear(embed_size * 3, 64), nn.ReLU(), nn.Linear(64, 32), nn.ReLU(),
                                       nn.Linear(32, 1))


class DIEN(SequenceModel):
    """``DIEN(num_items, embed_size)``; ``forward(hist, target_item) -> (B,1)``.

    attention (as DIN, un-summed) -> gi = seq W_ih^T + b_ih as ONE GEMM over all
    B*L rows -> sequential GRU kernel (h0 = 0) -> hidden[-1] lands in the left half
    of the fc input next to t -> fc MLP + sigmoid."""

    def __init__(self, num_items, embed_size, *, sharded=False, group=None):
        super().__init__()
        self.sharded = bool(sharded)
        self.din = DIN(num_items, embed_size, self._make_table(num_items, embed_size, sharded, group))
        self.interest_evolution = nn.GRU(embed_size, embed_size, batch_first=True)
        self.fc = nn.Sequential(nn.Linear(embed_size * 2, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                nn.Linear(64, 1), nn.Sigmoid())

    def _params(self):
        p = [self.din.item_embedding.weight]
        for k in (0, 2, 4):
            p += [self.din.attention[k].weight, self.din.attention[k].bias]
        for k in (0, 2, 4):
            p += [self.fc[k].weight, self.fc[k].bias]
        g = self.interest_evolution
        p += [g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0]
        return p

    def forward(self, hist, target_item):
        params = self._params()
        if self.sharded:
            self._need_device(hist, target_item, params[0])
            rows, hist, target_item = self._lookup_sharded(self.din.item_embedding, hist, target_item)
            params[0] = rows
        return self._run_sequence(hist, target_item, params)

    def run_forward(self, inputs, params):
        hist, target = inputs
        table = params[0]
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        w_ih, w_hh, b_ih, b_hh = params[13:17]
        batch, length = hist.shape
        dim = table.shape[1]
        dev = table.device
        w1f = fold_attention_weight(att[0].weight, dim)  # [h, t] operand instead of [h, h-t, t]
        att[0] = Layer(w1f, att[0].bias, att[0].act)
        c = torch.empty((batch * length, 2 * dim), dtype=torch.float32, device=dev)
        fcin = torch.empty((batch, 2 * dim), dtype=torch.float32, device=dev)
        ops.din_concat_fwd(table, hist, target, c, fcin[:, dim:], self._flag, pair=True)
        att_acts = ops.mlp_fwd(c, att)
        attn = torch.empty((batch, length), dtype=torch.float32, device=dev)
        seq = torch.empty((batch * length, dim), dtype=torch.float32, device=dev)
        ops.din_pool_fwd(att_acts[-1], c, batch, length, dim, attn, seq, summed=False)
        hbuf = torch.empty((batch * (length + 1), dim), dtype=torch.float32, device=dev)
        # the recurrence kernel forms the input projection itself where it can (E = 16): no (B*L, 3E) gi
        gi = None
        if not (FUSED_GRU and ops.gru_fused_fwd(seq, w_ih, b_ih, w_hh, b_hh, batch, length, dim, hbuf, fcin[:, :dim])):
            gi = ops.linear_fwd(seq, w_ih, b_ih)
            ops.gru_fwd(gi, w_hh, b_hh, batch, length, dim, hbuf, fcin[:, :dim])
        fc_acts = ops.mlp_fwd(fcin, fc)
        return fc_acts[-1], (att_acts, attn, seq, gi, hbuf, fc_acts, w1f)

    def run_backward(self, state, inputs, params, gprob):
        hist, target = inputs
        att_acts, attn, seq, gi, hbuf, fc_acts, w1f = state
        table = params[0]
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        w1 = att[0].weight
        att[0] = Layer(w1f, att[0].bias, att[0].act)
        w_ih, w_hh, b_ih, b_hh = params[13:17]
        batch, length = hist.shape
        dim = table.shape[1]
        dev = table.device
        c = att_acts[0]
        zeros = _zero_grads(self, params)
        zeros[id(w1f)] = torch.zeros_like(w1f)
        fc_grads, gfcin = ops.mlp_bwd(fc_acts, fc, gprob, None, zeros=zeros)
        g_w_ih, g_w_hh, g_b_ih, g_b_hh = (zeros[id(t)] for t in (w_ih, w_hh, b_ih, b_hh))
        gseq = torch.empty_like(seq)
        if gi is None:
            # input gradient and the four parameter gradients straight out of the recurrence's backward
            ops.gru_fused_bwd(seq, w_ih, b_ih, w_hh, b_hh, hbuf, batch, length, dim, gfcin[:, :dim], gseq, g_w_ih,
                              g_b_ih, g_w_hh, g_b_hh)
        else:
            dgi = torch.empty((batch * length, 3 * dim), dtype=torch.float32, device=dev)
            dgh = torch.empty((batch * (length + 1), 3 * dim), dtype=torch.float32, device=dev)
            ops.gru_bwd(gi, w_hh, b_hh, hbuf, batch, length, dim, gfcin[:, :dim], dgi, dgh)
            ops.linear_bwd(seq, w_ih, None, dgi, ACT_NONE, gseq, g_w_ih, g_b_ih)
            # dW_hh = sum_{b,t} dgh_t (x) h_{t-1}: rows r+1 of dgh against rows r of hbuf; the
            # zero row at the head of every sample makes the pairs that straddle samples vanish
            rows = batch * (length + 1) - 1
            if rows > 0:
                ops.linear_bwd(hbuf[:rows], w_hh, None, dgh[1:], ACT_NONE, None, g_w_hh, g_b_hh)
        gscore = torch.empty((batch * length, 1), dtype=torch.float32, device=dev)
        ops.din_pool_bwd(attn, c, batch, length, dim, gseq, False, gscore)
        att_grads, gc = ops.mlp_bwd(att_acts, att, gscore, None, zeros=zeros)
        gtable = zeros[id(table)]
        ops.din_concat_bwd(hist, target, table.shape[0], dim, gc, attn, gseq, False, gfcin[:, dim:], gtable,
                           pair=True)
        unfold_attention_grad(att_grads[0][0], zeros[id(w1)], dim)
        att_grads[0] = (zeros[id(w1)], att_grads[0][1])
        grads = [gtable]
        for gw, gb in att_grads + fc_grads:
            grads += [gw, gb]
        grads += [g_w_ih, g_w_hh, g_b_ih, g_b_hh]
        return grads

    def recommendation(self, num_users, num_items, hist_list, k):
        return self._rank_histories(num_users, num_items, hist_list, k)
