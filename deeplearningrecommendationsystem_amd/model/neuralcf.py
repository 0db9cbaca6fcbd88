"""NeuralCF -- counterpart of the reference's model/neuralcf.py:7-72."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, FieldSpec, Layer
from .._lib import FIELD_ID_I64, FIELD_PROD_I64
from ._base import CtrModule

# The gather of the four embedding rows inside the tower's forward kernel (ctr_embed_mlp_head_fwd) where the library
# has that kernel (the BASELINE shape); CTR_NCF_FUSED_GATHER=0 keeps the gather launch + tower launch for A/B.
FUSED_GATHER = os.environ.get("CTR_NCF_FUSED_GATHER", "1") != "0"
# ... and gathered AGAIN by the tower's backward kernel (ctr_embed_mlp_head_bwd), so that the forward never writes the
# (B, 128) tower input and the backward never reads it (the tables of the BASELINE shape sit in L2);
# CTR_NCF_REGATHER=0: the forward writes it, the backward reads it.
REGATHER = os.environ.get("CTR_NCF_REGATHER", "1") != "0"


# Vocabularies much smaller than the batch (BASELINE configs[1]: 943 + 1682 rows, batch 65536): the first tower layer
# and its backward on the TABLE ROWS instead of the samples (csrc/ncf_proj.hip, ops.NcfProj).  CTR_NCF_PROJ=0 keeps the
# per-sample kernels for A/B; they are also what every other shape runs (and the cross-check of this path in the tests).
PROJECT_TABLES = os.environ.get("CTR_NCF_PROJ", "1") != "0"


class _NeuralCFProjFunction(torch.autograd.Function):
    """inputs as ``_NeuralCFFunction``; one ``ctr_ncf_proj_fwd`` forward, one ``ctr_ncf_proj_bwd`` backward"""

    @staticmethod
    def forward(ctx, user_idx, item_idx, err_flag, n_hidden, gmf_u, gmf_i, mlp_u, mlp_i, *dense):
        n_hidden, grad_on, counts = n_hidden                 # (the caller's grad mode: it is off in here; the model's counters)
        hidden = [Layer(dense[2 * k], dense[2 * k + 1], ACT_RELU) for k in range(n_hidden)]
        proj = (dense[2 * n_hidden], dense[2 * n_hidden + 1])
        head = (dense[2 * n_hidden + 2], dense[2 * n_hidden + 3])
        # needs_input_grad is True for parameters under torch.no_grad() too; without a graph there will be no backward, and
        # the forward then takes no ranks (the backward's bucketing needs them; they are returning atomics:
        # profiles/r03_rank_atomics.txt)
        training = grad_on and any(ctx.needs_input_grad[4:])
        run = ops.NcfProj(user_idx, item_idx, (gmf_u, gmf_i, mlp_u, mlp_i), hidden, proj, head, err_flag, training, counts)
        prob = run.forward()
        if prob is None:
            raise _lib_error("ctr_ncf_proj_fwd refused a shape NcfProj.supported() accepted")
        ctx.run = run if training else None
        ctx.save_for_backward(gmf_u, gmf_i, mlp_u, mlp_i, *dense)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        run, params = ctx.run, ctx.saved_tensors
        if run is None:
            raise RuntimeError("NeuralCF backward without a training forward")
        ctx.run = None
        zeros = ops.zero_grads(list(params), lazy=True)   # cleared by the backward's first launch
        flat = zeros.pop("flat")
        run.backward(gprob.contiguous(), zeros, flat)
        return (None, None, None, None) + tuple(zeros[id(p)] for p in params)


class _NeuralCFRowsFunction(torch.autograd.Function):
    """The same move for ANY tower (the reference script's NeuralCF(943, 1682, 256, [512, 256, 128, 64, 32]),
    scripts/neuralcf.py:60): the first layer on the table rows, composed from library calls --
        P_U = MLP_U W0[:, :h]^T, P_I = MLP_I W0[:, h:]^T + b0           two ctr_linear_fwd over U + I rows
        a0  = relu(P_U[u] + P_I[i])                                      ctr_rows_sum_act_fwd
        tower layers 1.., folded head, GMF product                       as in _NeuralCFFunction
    backward: the gradient of a0, masked (ctr_act_mask_bwd), is the gradient of BOTH projected rows: ctr_embed_bwd sums
    it by user and by item (S_U, S_I), and two ctr_linear_bwd over the table rows give dMLP = S W0half,
    dW0half = S^T MLP, db0 = column sums.  Three quarters of the tower's matrix work leave the batch."""

    @staticmethod
    def forward(ctx, user_idx, item_idx, err_flag, n_hidden, gmf_u, gmf_i, mlp_u, mlp_i, *dense):
        batch = user_idx.numel()
        mf, half = gmf_u.shape[1], mlp_u.shape[1]
        hidden = [Layer(dense[2 * k], dense[2 * k + 1], ACT_RELU) for k in range(n_hidden)]
        proj_w, proj_b = dense[2 * n_hidden], dense[2 * n_hidden + 1]
        head_w, head_b = dense[2 * n_hidden + 2], dense[2 * n_hidden + 3]
        w0, b0 = hidden[0].weight, hidden[0].bias
        n0, kh = w0.shape[0], proj_w.shape[1]
        dev = gmf_u.device
        p_u = ops.linear_fwd(mlp_u, w0[:, :half], None)
        p_i = ops.linear_fwd(mlp_i, w0[:, half:], b0)
        buf = torch.empty((batch, n0 + mf + kh), dtype=torch.float32, device=dev)   # [a0 | gmf | h]
        ops.rows_sum_act_fwd(p_u, user_idx, p_i, item_idx, ACT_RELU, buf[:, :n0], err_flag)
        ops.embed_fwd([FieldSpec(FIELD_PROD_I64, mf, n0, table=gmf_u, idx=user_idx, table2=gmf_i, idx2=item_idx)], None,
                      batch, buf, err_flag)
        wfold, cfold = ops.fold_head_fwd(head_w, mf, proj_w, proj_b, head_b)
        head = ops.Head(buf[:, n0:n0 + mf], wfold, cfold, ACT_SIGMOID)
        acts = ops.mlp_fwd(buf[:, :n0], hidden[1:], last_out=buf[:, n0 + mf:], head=head)
        prob = head.out
        ctx.n_hidden = n_hidden
        ctx.save_for_backward(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i, buf, prob, wfold, p_u, p_i, *acts[1:-1], *dense)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        n_hidden = ctx.n_hidden
        saved = ctx.saved_tensors
        user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i, buf, prob, wfold, p_u, p_i = saved[:11]
        nmid = n_hidden - 2
        mids = list(saved[11:11 + nmid])
        dense = saved[11 + nmid:]
        batch = user_idx.numel()
        mf, half = gmf_u.shape[1], mlp_u.shape[1]
        hidden = [Layer(dense[2 * k], dense[2 * k + 1], ACT_RELU) for k in range(n_hidden)]
        proj_w, proj_b = dense[2 * n_hidden], dense[2 * n_hidden + 1]
        head_w, head_b = dense[2 * n_hidden + 2], dense[2 * n_hidden + 3]
        w0, n0 = hidden[0].weight, hidden[0].weight.shape[0]
        tables = (gmf_u, gmf_i, mlp_u, mlp_i)
        zeros = ops.zero_grads(list(tables) + list(dense) + [wfold, head_b.new_empty(4), p_u, p_i])
        gwfold, gcfold = zeros[id(wfold)], zeros[id(wfold)].new_zeros(1)
        s_u, s_i = zeros[id(p_u)], zeros[id(p_i)]                 # the row sums of the first layer's gradient
        gbuf = torch.empty_like(buf)
        acts = [buf[:, :n0]] + mids + [buf[:, n0 + mf:]]
        # the head as a single-unit layer on [gmf | h], then the tower down to a0
        ops.linear_bwd(buf[:, n0:], wfold, prob, gprob.contiguous(), ACT_SIGMOID, gbuf[:, n0:], gwfold, gcfold)
        layer_grads, _ = ops.mlp_bwd(acts, hidden[1:], gbuf[:, n0 + mf:], gbuf[:, :n0], zeros=zeros)
        ops.act_mask_bwd(gbuf[:, :n0], buf[:, :n0], ACT_RELU)
        ops.fold_head_bwd(head_w, mf, proj_w, proj_b, gwfold, gcfold, zeros[id(head_w)], zeros[id(proj_w)],
                          zeros[id(proj_b)], zeros[id(head_b)])
        specs = [FieldSpec(FIELD_ID_I64, n0, 0, table=p_u, idx=user_idx),
                 FieldSpec(FIELD_ID_I64, n0, 0, table=p_i, idx=item_idx),
                 FieldSpec(FIELD_PROD_I64, mf, n0, table=gmf_u, idx=user_idx, table2=gmf_i, idx2=item_idx)]
        ops.embed_bwd(specs, None, batch, gbuf, zeros)
        g_w0 = zeros[id(w0)]
        ops.linear_bwd(mlp_u, w0[:, :half], None, s_u, ACT_NONE, zeros[id(mlp_u)], g_w0[:, :half], None, accumulate_gx=True)
        ops.linear_bwd(mlp_i, w0[:, half:], None, s_i, ACT_NONE, zeros[id(mlp_i)], g_w0[:, half:], zeros[id(hidden[0].bias)],
                       accumulate_gx=True)
        out = [None, None, None, None] + [zeros[id(t)] for t in tables]
        out += [g_w0, zeros[id(hidden[0].bias)]]
        for gw, gb in layer_grads:
            out += [gw, gb]
        out += [zeros[id(proj_w)], zeros[id(proj_b)], zeros[id(head_w)], zeros[id(head_b)]]
        return tuple(out)


def _rows_path_ok(tables, hidden, batch) -> bool:
    """the composed table-row path: a tower of at least two layers whose first layer takes cat(MLP_U[u], MLP_I[i]), row
    counts far below the batch, widths the library's row kernels take"""
    gmf_u, gmf_i, mlp_u, mlp_i = tables
    rows = gmf_u.shape[0] + gmf_i.shape[0]
    if len(hidden) < 2 or hidden[0].bias is None:
        return False
    n0, k0 = hidden[0].weight.shape
    half = mlp_u.shape[1]
    return (k0 == 2 * half and half % 4 == 0 and n0 % 4 == 0 and n0 <= 256 and gmf_u.shape[1] % 4 == 0 and
            batch >= 4096 and batch >= 4 * rows)


def _lib_error(msg):
    from .._lib import CtrHipError
    return CtrHipError(msg)


class _NeuralCFFunction(torch.autograd.Function):
    """inputs: user_idx, item_idx, err_flag, n_hidden, then parameters in the
    order GMF_U, GMF_I, MLP_U, MLP_I, (W,b) x n_hidden, linear W,b, linear2 W,b.

    ``linear`` (h -> mf_dim, no activation) feeds only ``linear2`` (reference model/neuralcf.py:50-56), so
    the pair is one k-wide dot product per sample: ``[gmf | linear(h)] . w2 + b2 == [gmf | h] . wfold + c``
    with ``wfold = [w2[:mf] | W_l^T w2[mf:]]``, ``c = b_l . w2[mf:] + b2`` (ops.fold_head_fwd, O(mf*k) per
    step).  The (B, mf_dim) output of ``linear`` and its gradient never exist; the gradients of
    ``linear`` / ``linear2`` come back through the chain rule (ops.fold_head_bwd) from the k + mf sums the
    head's backward produces anyway.  Same values up to fp32 rounding (parity tests unchanged).

    Buffer layout (one (B, L0 + mf + k) matrix, no torch.cat anywhere):
        [ MLP_U[u] | MLP_I[i] |  GMF_U[u]*GMF_I[i] | h = tower(x0) ]
          `-- MLP input x0 --'   `--- input of the folded head ---'
    """

    @staticmethod
    def forward(ctx, user_idx, item_idx, err_flag, n_hidden, gmf_u, gmf_i, mlp_u, mlp_i, *dense):
        batch = user_idx.numel()
        mf, half = gmf_u.shape[1], mlp_u.shape[1]
        l0 = 2 * half
        hidden = [Layer(dense[2 * k], dense[2 * k + 1], ACT_RELU) for k in range(n_hidden)]
        proj_w, proj_b = dense[2 * n_hidden], dense[2 * n_hidden + 1]
        head_w, head_b = dense[2 * n_hidden + 2], dense[2 * n_hidden + 3]
        kh = proj_w.shape[1]  # width of h
        buf = torch.empty((batch, l0 + mf + (kh if n_hidden else 0)), dtype=torch.float32, device=gmf_u.device)
        specs = _specs(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i)
        acts = None
        wfold = cfold = None
        if n_hidden and FUSED_GATHER:
            # gather + head fold + tower + folded head in one launch; None: the library has no such kernel for this shape
            wfold = torch.empty((1, mf + kh), dtype=torch.float32, device=buf.device)
            cfold = torch.empty(1, dtype=torch.float32, device=buf.device)
            head = ops.Head(buf[:, l0:l0 + mf], wfold, cfold, ACT_SIGMOID)
            regather = REGATHER and any(ctx.needs_input_grad[4:])
            acts = ops.embed_mlp_head_fwd(specs, batch, buf, l0, hidden, head, buf[:, l0 + mf:], err_flag,
                                          write_x=not regather, fold=(head_w, proj_w, proj_b, head_b))
            ctx.regather = regather and acts is not None
        if acts is None:
            wfold, cfold = ops.fold_head_fwd(head_w, mf, proj_w, proj_b, head_b,
                                             out=(wfold, cfold) if wfold is not None else None)
        if acts is not None:
            prob = head.out
        elif n_hidden:
            ops.embed_fwd(specs, None, batch, buf, err_flag)
            # tower + folded head in one launch: the head's dot product runs on the tile's last
            # activations while they are still in LDS
            head = ops.Head(buf[:, l0:l0 + mf], wfold, cfold, ACT_SIGMOID)
            acts = ops.mlp_fwd(buf[:, :l0], hidden, last_out=buf[:, l0 + mf:], head=head)
            prob = head.out
        else:
            ops.embed_fwd(specs, None, batch, buf, err_flag)
            # no tower: h is x0 itself, which sits in FRONT of the GMF columns
            acts = [buf[:, :l0]]
            wf = torch.cat([wfold[:, mf:], wfold[:, :mf]], dim=1)
            prob = ops.linear_fwd(buf, wf, cfold, ACT_SIGMOID)
        ctx.n_hidden = n_hidden
        if not hasattr(ctx, "regather"):
            ctx.regather = False
        ctx.save_for_backward(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i, buf, prob, wfold, *acts[1:-1], *dense)
        return prob

    @staticmethod
    def backward(ctx, gprob):
        n_hidden = ctx.n_hidden
        saved = ctx.saved_tensors
        user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i, buf, prob, wfold = saved[:9]
        nmid = max(n_hidden - 1, 0)
        mids = list(saved[9:9 + nmid])
        dense = saved[9 + nmid:]
        batch = user_idx.numel()
        mf, l0 = gmf_u.shape[1], 2 * mlp_u.shape[1]
        hidden = [Layer(dense[2 * k], dense[2 * k + 1], ACT_RELU) for k in range(n_hidden)]
        proj_w, proj_b = dense[2 * n_hidden], dense[2 * n_hidden + 1]
        head_w, head_b = dense[2 * n_hidden + 2], dense[2 * n_hidden + 3]

        tables = (gmf_u, gmf_i, mlp_u, mlp_i)
        cfold_like = head_b  # (1,)
        # (with the gathering backward the flat gradient buffer is cleared by that call's first launch, not by a fill)
        zeros = ops.zero_grads(list(tables) + list(dense) + [wfold, cfold_like.new_empty(4)],
                               lazy=bool(ctx.regather and n_hidden))
        flat = zeros.pop("flat", None)
        gwfold, gcfold = zeros[id(wfold)], list(zeros.values())[-1][:1]
        gbuf = torch.empty_like(buf)
        g_proj_w, g_proj_b = zeros[id(proj_w)], zeros[id(proj_b)]
        g_head_w, g_head_b = zeros[id(head_w)], zeros[id(head_b)]
        folded = False
        if n_hidden:
            acts = [buf[:, :l0]] + mids + [buf[:, l0 + mf:]]
            # head backward + tower backward in one launch where the library has it (the BASELINE tower) ...
            head = ops.Head(buf[:, l0:l0 + mf], wfold, None, ACT_SIGMOID)
            layer_grads = ops.mlp_head_bwd(acts, hidden, head, prob, gprob.contiguous(), gbuf[:, l0:l0 + mf], gwfold,
                                           gcfold, gbuf[:, :l0], zeros,
                                           gather_specs=_specs(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i)
                                           if ctx.regather else None,
                                           # ... and fold_head_bwd in that call's reduction launch
                                           fold_grad=(head_w, proj_w, proj_b, g_head_w, g_proj_w, g_proj_b, g_head_b)
                                           if ctx.regather else None, zero=flat)
            folded = ctx.regather
            if layer_grads is None:
                # ... else the head as a single-unit layer on [gmf | h], then the tower
                ops.linear_bwd(buf[:, l0:], wfold, prob, gprob.contiguous(), ACT_SIGMOID, gbuf[:, l0:], gwfold, gcfold)
                layer_grads, _ = ops.mlp_bwd(acts, hidden, gbuf[:, l0 + mf:], gbuf[:, :l0], zeros=zeros)
        else:
            wf = torch.cat([wfold[:, mf:], wfold[:, :mf]], dim=1)
            gwf = torch.zeros_like(wf)
            ops.linear_bwd(buf, wf, prob, gprob.contiguous(), ACT_SIGMOID, gbuf, gwf, gcfold)
            gwfold.copy_(torch.cat([gwf[:, l0:], gwf[:, :l0]], dim=1))
            layer_grads = []
        if not folded:
            ops.fold_head_bwd(head_w, mf, proj_w, proj_b, gwfold, gcfold, g_head_w, g_proj_w, g_proj_b, g_head_b)
        tgrads = zeros
        ops.embed_bwd(_specs(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i), None, batch, gbuf, tgrads)
        out = [None, None, None, None] + [tgrads[id(t)] for t in tables]
        for gw, gb in layer_grads:
            out += [gw, gb]
        out += [g_proj_w, g_proj_b, g_head_w, g_head_b]
        return tuple(out)


def _specs(user_idx, item_idx, gmf_u, gmf_i, mlp_u, mlp_i):
    mf, half = gmf_u.shape[1], mlp_u.shape[1]
    l0 = 2 * half
    return [
        FieldSpec(FIELD_ID_I64, half, 0, table=mlp_u, idx=user_idx),
        FieldSpec(FIELD_ID_I64, half, half, table=mlp_i, idx=item_idx),
        FieldSpec(FIELD_PROD_I64, mf, l0, table=gmf_u, idx=user_idx, table2=gmf_i, idx2=item_idx),
    ]


class NeuralCF(CtrModule):
    """``NeuralCF(num_user, num_item, mf_dim, layers)``;
    ``forward(user_indices, item_indices) -> (B,1)`` (reference
    model/neuralcf.py:8-59)."""

    def __init__(self, num_user, num_item, mf_dim, layers):
        super().__init__()
        self.GMF_Embedding_User = nn.Embedding(num_user, mf_dim)
        self.GMF_Embedding_Item = nn.Embedding(num_item, mf_dim)
        self.MLP_Embedding_User = nn.Embedding(num_user, int(layers[0] / 2))
        self.MLP_Embedding_Item = nn.Embedding(num_item, int(layers[0] / 2))
        for emb in (self.GMF_Embedding_User, self.GMF_Embedding_Item, self.MLP_Embedding_User,
                    self.MLP_Embedding_Item):
            xavier_normal_(emb.weight.data)
        self.dnn_network = nn.ModuleList([nn.Linear(a, b) for a, b in zip(layers[:-1], layers[1:])])
        self.relu = nn.ReLU()
        self.linear = nn.Linear(layers[-1], mf_dim)
        self.linear2 = nn.Linear(2 * mf_dim, 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, user_indices, item_indices):
        w = self.GMF_Embedding_User.weight
        self._need_device(w, user_indices, item_indices)
        dense = []
        for lin in list(self.dnn_network) + [self.linear, self.linear2]:
            dense += [lin.weight, lin.bias]
        tables = (w, self.GMF_Embedding_Item.weight, self.MLP_Embedding_User.weight, self.MLP_Embedding_Item.weight)
        hidden = [Layer(lin.weight, lin.bias, ACT_RELU) for lin in self.dnn_network]
        fn = _NeuralCFFunction
        if PROJECT_TABLES and user_indices.dim() == 1 and ops.NcfProj.supported(
                tables, hidden, (self.linear.weight, self.linear.bias), user_indices.numel()) and not any(
                getattr(t, "_ctr_sparse", None) is not None for t in tables):
            fn = _NeuralCFProjFunction
        elif PROJECT_TABLES and user_indices.dim() == 1 and _rows_path_ok(tables, hidden, user_indices.numel()) and not any(
                getattr(t, "_ctr_sparse", None) is not None for t in tables):
            fn = _NeuralCFRowsFunction
        n_hidden = len(self.dnn_network)
        if fn is _NeuralCFProjFunction:
            if getattr(self, "_ncf_counts", None) is None:
                object.__setattr__(self, "_ncf_counts", ops.NcfCounts())
            n_hidden = (n_hidden, torch.is_grad_enabled(), self._ncf_counts)
        out = fn.apply(user_indices.contiguous(), item_indices.contiguous(), self._err_flag(w.device), n_hidden, *tables, *dense)
        self._raise_if_bad_index()
        return out

    def recommendation(self, num_users, num_items, chunk: int = 1 << 20):
        """score every (user, item) pair and rank: the result of the reference's per-user loop
        (model/neuralcf.py:61-72; 943 forward calls of 1682 samples) from a few forward calls over
        the whole user x item grid, ``chunk`` pairs at a time, ranked on the device"""
        dev = self.GMF_Embedding_User.weight.device
        users = torch.arange(num_users, device=dev).repeat_interleave(num_items)
        items = torch.arange(num_items, device=dev).repeat(num_users)
        scores = torch.empty(num_users * num_items, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for lo in range(0, users.numel(), chunk):
                scores[lo:lo + chunk] = self.forward(users[lo:lo + chunk], items[lo:lo + chunk]).view(-1)
        return ops.topk_rows(scores.view(num_users, num_items), num_items).cpu().numpy()
