"""DIN -- counterpart of the reference's model/din.py:9-66."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, Layer
from ._base import CtrModule, _ModelFunction

# the history gradient of the E-wide attention leaves the first layer's dX GEMM as atomics on the table gradient
# (ctr_linear_dx_scatter); CTR_DIN_FUSED_SCATTER=0 keeps the two-pass form (dX to memory, seq_scatter) for A/B
FUSED_SCATTER = os.environ.get("CTR_DIN_FUSED_SCATTER", "1") != "0"


def attention_layers(p):
    """(w0,b0,w2,b2,w4,b4) -> Linear+ReLU, Linear+ReLU, Linear"""
    return [Layer(p[0], p[1], ACT_RELU), Layer(p[2], p[3], ACT_RELU), Layer(p[4], p[5], ACT_NONE)]


def fold_attention_weight(w1, dim):
    """``W1 @ [h, h-t, t] == [Wa+Wb, Wc-Wb] @ [h, t]`` (W1 = [Wa | Wb | Wc], model/din.py:39-42):
    the first attention layer on a 2E-wide operand instead of the reference's 3E -- two thirds
    of the bytes and flops of the largest GEMM of a step, same layer output up to fp32 rounding.
    Returns the folded (n, 2E) weight (two small elementwise launches)."""
    folded = torch.empty((w1.shape[0], 2 * dim), dtype=w1.dtype, device=w1.device)
    torch.add(w1[:, :dim], w1[:, dim:2 * dim], out=folded[:, :dim])
    torch.sub(w1[:, 2 * dim:], w1[:, dim:2 * dim], out=folded[:, dim:])
    return folded


def unfold_attention_grad(gfolded, gw1, dim):
    """chain rule of ``fold_attention_weight``: dWa = g1, dWb = g1 - g2, dWc = g2"""
    g1, g2 = gfolded[:, :dim], gfolded[:, dim:]
    gw1[:, :dim].copy_(g1)
    torch.sub(g1, g2, out=gw1[:, dim:2 * dim])
    gw1[:, 2 * dim:].copy_(g2)


def _lib_field_id():
    from .._lib import FIELD_ID_I64
    return FIELD_ID_I64


def fc_layers(p):
    return [Layer(p[0], p[1], ACT_RELU), Layer(p[2], p[3], ACT_RELU), Layer(p[4], p[5], ACT_SIGMOID)]


def _zero_grads(model, params):
    """one flat zero buffer for the parameters' gradients; the exchanged rows of a sharded
    table are an activation, not a replicated parameter: their gradient gets its own buffer
    so that the data-parallel all-reduce of the flat one does not carry it"""
    if not getattr(model, "sharded", False):
        return ops.zero_grads(params)
    zeros = ops.zero_grads(params[1:])
    zeros[id(params[0])] = torch.zeros_like(params[0])
    return zeros


class SequenceModel(CtrModule):
    """shared plumbing of the (hist, target) models"""

    def _make_table(self, num_items, embed_size, sharded, group):
        """``nn.Embedding`` like the reference, or -- keyword-only extension for the 1e7-row
        multi-GPU configs -- its row-sharded counterpart (dist.ShardedEmbedding)"""
        if not sharded:
            emb = nn.Embedding(num_items, embed_size)
            xavier_normal_(emb.weight.data)
            return emb
        from ..dist import ShardedEmbedding
        return ShardedEmbedding(num_items, embed_size, group=group)

    def sparse_ids(self, inputs):
        """sparse mode: the item table (position 0) is touched by every history id and the target ids"""
        if getattr(self, "sharded", False):
            return {}
        return {0: [] if inputs is None else [inputs[0].reshape(-1), inputs[1]]}

    def _lookup_sharded(self, table_module, hist, target):
        """rows of every (hist, target) id through the all-to-all exchange, then the model runs
        on them as if they were a (B*L+B)-row table indexed 0..B*L+B-1: the kernels are unchanged,
        their table gradient becomes the gradient of the exchanged rows and flows back through
        the exchange into the owners' shards"""
        batch, length = hist.shape
        ids = torch.cat([hist.reshape(-1), target.reshape(-1)])
        # the concatenation is a temporary: the exchange plan is keyed by the caller's tensors (same, unmodified
        # (hist, target) next epoch -> no bucketing, no id exchange, no host sync)
        rows = table_module(ids, plan_key=(hist, target))   # both held by weak reference: dist._split_key
        pos = torch.arange(batch * (length + 1), device=hist.device, dtype=torch.int64)
        return rows, pos[:batch * length].view(batch, length), pos[batch * length:]

    def _run_sequence(self, hist, target, params):
        self._need_device(hist, target, params[0])
        if hist.dim() != 2 or target.dim() != 1 or hist.shape[0] != target.shape[0]:
            raise ValueError(f"expected hist (B,L) and target (B,), got {tuple(hist.shape)} {tuple(target.shape)}")
        object.__setattr__(self, "_flag", self._err_flag(hist.device))
        out = _ModelFunction.apply(self, 2, hist.contiguous(), target.contiguous(), *params)
        self._raise_if_bad_index()
        return out

    def _rank_histories(self, num_users, num_items, hist_list, k, max_positions: int = 1 << 22):
        """reference recommendation() (model/din.py:55-66): every user's WHOLE history (any length) against every
        item.  The reference builds ``hist.repeat(num_items, 1)`` on the host and calls forward once per user;
        here users are grouped by history length, a group's (user x item) samples go through forward together
        (up to ``max_positions`` history positions per call), and the ranking is one top-k per group."""
        dev = next(self.parameters()).device
        lengths = [len(hist_list[u]) for u in range(num_users)]
        out = np.empty((num_users, k), dtype=np.int64)
        targets = torch.arange(0, num_items, device=dev)
        by_len = {}
        for u, n in enumerate(lengths):
            by_len.setdefault(n, []).append(u)
        with torch.no_grad():
            for n, users in by_len.items():
                per_call = max(1, max_positions // max(1, n * num_items))
                for lo in range(0, len(users), per_call):
                    part = users[lo:lo + per_call]
                    hist = torch.as_tensor(np.stack([np.asarray(hist_list[u], dtype=np.int64) for u in part]), device=dev)
                    hist = hist.repeat_interleave(num_items, 0)                     # (users * items, n)
                    scores = self.forward(hist, targets.repeat(len(part))).view(len(part), num_items)
                    out[part] = ops.topk_rows(scores, k).cpu().numpy()
        return out


class DIN(SequenceModel):
    """``DIN(num_items, embed_size)``; ``forward(hist (B,L) int64, target_item (B,) int64) -> (B,1)``.

    Pipeline: gather + [h, h-t, t] operand (one kernel) -> attention MLP on the
    matrix cores over all B*L positions -> softmax over L + weighted sum written
    straight into the left half of the fc input, whose right half (t) the gather
    kernel already filled -> fc MLP + sigmoid."""

    def __init__(self, num_items, embed_size, *, sharded=False, group=None):
        super().__init__()
        self.sharded = bool(sharded)
        self.item_embedding = self._make_table(num_items, embed_size, sharded, group)
        self.attention = nn.Sequential(nn.Linear(embed_size * 3, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                       nn.Linear(64, 1))
        self.fc = nn.Sequential(nn.Linear(embed_size * 2, 256), nn.ReLU(), nn.Linear(256, 128), nn.ReLU(),
                                nn.Linear(128, 1), nn.Sigmoid())

    def _params(self):
        p = [self.item_embedding.weight]
        for seq in (self.attention, self.fc):
            for k in (0, 2, 4):
                p += [seq[k].weight, seq[k].bias]
        return p

    def forward(self, hist, target_item):
        params = self._params()
        if self.sharded:
            self._need_device(hist, target_item, params[0])
            rows, hist, target_item = self._lookup_sharded(self.item_embedding, hist, target_item)
            params[0] = rows
        return self._run_sequence(hist, target_item, params)

    # ---- attention on the E-wide operand: W1 [h, h-t, t] + b1 = (Wa+Wb) h + u[b],  u[b] = (Wc-Wb) t_b + b1
    # (model/din.py:39-44).  The GEMM over all B*L positions contracts E columns instead of 3E (reference) or 2E
    # (the [h, t] operand below); the per-sample term rides in the GEMM epilogue, its gradient comes out of the
    # layer-2 input-gradient epilogue as per-sample column sums.
    e_wide = True

    def _use_e_wide(self, hist, dim):
        n1 = self.attention[0].weight.shape[0]
        return (self.e_wide and dim >= 4 and dim <= 256 and (dim & (dim - 1)) == 0 and hist.shape[1] >= 32 and n1 <= 128
                and self.attention[2].weight.shape[0] >= 4 and hist.numel() < 2 ** 32)

    def _forward_e_wide(self, hist, target, params):
        table = params[0]
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        batch, length = hist.shape
        dim, dev = table.shape[1], table.device
        w1 = att[0].weight
        wf = torch.empty((2, w1.shape[0], dim), dtype=torch.float32, device=dev)
        torch.add(w1[:, :dim], w1[:, dim:2 * dim], out=wf[0])        # Wh = Wa + Wb
        torch.sub(w1[:, 2 * dim:], w1[:, dim:2 * dim], out=wf[1])    # Wu = Wc - Wb
        hrows = torch.empty((batch * length, dim), dtype=torch.float32, device=dev)
        fcin = torch.empty((batch, 2 * dim), dtype=torch.float32, device=dev)
        ops.din_concat_fwd(table, hist, target, hrows, fcin[:, dim:], self._flag, h_only=True)
        u = ops.linear_fwd(fcin[:, dim:], wf[1], att[0].bias)                       # (B, n1)
        # the ReLU sign bits of z1 travel to the backward as 1 bit per element when the width allows whole words
        n1 = w1.shape[0]
        bits = torch.empty((batch * length, n1 // 32), dtype=torch.int32, device=dev) if n1 % 32 == 0 else None
        z1 = ops.linear_group_fwd(hrows, wf[0], None, u, length, ACT_RELU, sign_bits=bits)   # (B*L, n1)
        if att[1].weight.shape[0] <= 128 and att[2].weight.shape[0] == 1 and att[2].act == ACT_NONE:
            # the score layer rides in layer 2's epilogue (its rows are whole in a workgroup)
            h2, score = ops.linear_fwd_dot(z1, att[1].weight, att[1].bias, ACT_RELU, att[2].weight, att[2].bias)
        else:
            h2 = ops.linear_fwd(z1, att[1].weight, att[1].bias, ACT_RELU)
            score = ops.linear_fwd(h2, att[2].weight, att[2].bias, ACT_NONE)
        attn = torch.empty((batch, length), dtype=torch.float32, device=dev)
        ops.din_pool_fwd(score, hrows, batch, length, dim, attn, fcin[:, :dim], summed=True)
        fc_acts = ops.mlp_fwd(fcin, fc)
        return fc_acts[-1], ("e", hrows, z1, h2, attn, fc_acts, wf, bits, [False])

    def _backward_e_wide(self, state, hist, target, params, gprob):
        _, hrows, z1, h2, attn, fc_acts, wf, bits, spent = state
        if spent[0]:
            raise RuntimeError("DIN backward overwrites its saved activations: run the forward again before a second "
                               "backward (retain_graph is not supported)")
        spent[0] = True
        table = params[0]
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        w1, b1 = att[0].weight, att[0].bias
        batch, length = hist.shape
        dim, dev = table.shape[1], table.device
        n1 = w1.shape[0]
        fcin = fc_acts[0]
        zeros = _zero_grads(self, params)
        fc_grads, gfcin = ops.mlp_bwd(fc_acts, fc, gprob, None, zeros=zeros)
        gscore = torch.empty((batch * length, 1), dtype=torch.float32, device=dev)
        ops.din_pool_bwd(attn, hrows, batch, length, dim, gfcin[:, :dim], True, gscore)
        # layer 3 (n2 -> 1) with layer 2's ReLU derivative folded in: h2 is overwritten by the pre-activation
        # gradient of layer 2, so neither layer-2 kernel below reads an activation to mask with
        gz2 = h2
        ops.linear_n1_bwd_masked(h2, att[2].weight, gscore, ACT_RELU, gz2, zeros[id(att[2].weight)],
                                 zeros[id(att[2].bias)])
        ops.linear_bwd(z1, att[1].weight, None, gz2, ACT_NONE, None, zeros[id(att[1].weight)], zeros[id(att[1].bias)])
        gz1 = torch.empty_like(z1)
        gu = torch.zeros((batch, n1), dtype=torch.float32, device=dev)
        ops.linear_dx_masked(att[1].weight, None, gz2, ACT_NONE, z1, ACT_RELU, gz1, gu, length, sign_bits=bits)
        # layer 1 on the E-wide operand (gz1 already carries relu'(z1))
        gwf = torch.zeros_like(wf)
        gtable = zeros[id(table)]
        fused_scatter = FUSED_SCATTER and dim % 32 == 0 and dim <= 128 and length >= 32 and table.shape[0] < 2 ** 31
        if fused_scatter:
            # dWh alone, then the input gradient added straight to the history rows of the table gradient (with
            # the pooling's share): the (B*L, E) gradient and its scatter pass never exist
            ops.linear_bwd(hrows, wf[0], None, gz1, ACT_NONE, None, gwf[0], zeros[id(b1)])
            ops.linear_dx_scatter(wf[0], gz1, hist.reshape(-1), attn.reshape(-1), gfcin[:, :dim], length, gtable)
        else:
            ghrows = torch.empty_like(hrows)
            ops.linear_bwd(hrows, wf[0], None, gz1, ACT_NONE, ghrows, gwf[0], zeros[id(b1)])
        # u = t Wu^T + b1: dWu = gu^T t, and the target's gradient gains gu Wu on top of the fc input's share
        gt = gfcin[:, dim:]
        ops.linear_bwd(fcin[:, dim:], wf[1], None, gu, ACT_NONE, gt, gwf[1], None, accumulate_gx=True)
        gw1 = zeros[id(w1)]                                   # dWa = dWh, dWb = dWh - dWu, dWc = dWu
        gw1[:, :dim].copy_(gwf[0])
        torch.sub(gwf[0], gwf[1], out=gw1[:, dim:2 * dim])
        gw1[:, 2 * dim:].copy_(gwf[1])
        if not fused_scatter:
            ops.din_scatter_bwd(hist, table.shape[0], dim, ghrows, attn, gfcin[:, :dim], True, gtable)
        ops.embed_bwd([ops.FieldSpec(_lib_field_id(), dim, dim, table=table, idx=target)], None, batch, gfcin,
                      {id(table): gtable})
        grads = [gtable]
        for layer in att:
            grads += [zeros[id(layer.weight)], zeros[id(layer.bias)]]
        for gw, gb in fc_grads:
            grads += [gw, gb]
        return grads

    def run_forward(self, inputs, params):
        hist, target = inputs
        table = params[0]
        if self._use_e_wide(hist, table.shape[1]):
            return self._forward_e_wide(hist, target, params)
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        batch, length = hist.shape
        dim = table.shape[1]
        dev = table.device
        w1f = fold_attention_weight(att[0].weight, dim)
        att[0] = Layer(w1f, att[0].bias, att[0].act)
        c = torch.empty((batch * length, 2 * dim), dtype=torch.float32, device=dev)
        fcin = torch.empty((batch, 2 * dim), dtype=torch.float32, device=dev)
        ops.din_concat_fwd(table, hist, target, c, fcin[:, dim:], self._flag, pair=True)
        att_acts = ops.mlp_fwd(c, att)
        attn = torch.empty((batch, length), dtype=torch.float32, device=dev)
        ops.din_pool_fwd(att_acts[-1], c, batch, length, dim, attn, fcin[:, :dim], summed=True)
        fc_acts = ops.mlp_fwd(fcin, fc)
        return fc_acts[-1], (att_acts, attn, fc_acts, w1f)

    def run_backward(self, state, inputs, params, gprob):
        hist, target = inputs
        if state[0] == "e":
            return self._backward_e_wide(state, hist, target, params, gprob)
        att_acts, attn, fc_acts, w1f = state
        table = params[0]
        att, fc = attention_layers(params[1:7]), fc_layers(params[7:13])
        w1 = att[0].weight
        att[0] = Layer(w1f, att[0].bias, att[0].act)
        batch, length = hist.shape
        dim = table.shape[1]
        c, fcin = att_acts[0], fc_acts[0]
        zeros = _zero_grads(self, params)
        zeros[id(w1f)] = torch.zeros_like(w1f)
        fc_grads, gfcin = ops.mlp_bwd(fc_acts, fc, gprob, None, zeros=zeros)
        gscore = torch.empty((batch * length, 1), dtype=torch.float32, device=table.device)
        ops.din_pool_bwd(attn, c, batch, length, dim, gfcin[:, :dim], True, gscore)
        att_grads, gc = ops.mlp_bwd(att_acts, att, gscore, None, zeros=zeros)
        gtable = zeros[id(table)]
        ops.din_concat_bwd(hist, target, table.shape[0], dim, gc, attn, gfcin[:, :dim], True, gfcin[:, dim:], gtable,
                           pair=True)
        unfold_attention_grad(att_grads[0][0], zeros[id(w1)], dim)
        att_grads[0] = (zeros[id(w1)], att_grads[0][1])
        grads = [gtable]
        for gw, gb in att_grads + fc_grads:
            grads += [gw, gb]
        return grads

    def recommendation(self, num_users, num_items, hist_list, k):
        return self._rank_histories(num_users, num_items, hist_list, k)
