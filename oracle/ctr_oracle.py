"""CPU oracle for the CTR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The shipped package
(``deeplearningrecommendationsystem_amd``) never imports it and has no CPU
fallback: it raises when the HIP library is missing.

What this is: a functional, dependency-free (numpy + torch CPU) restatement of
the arithmetic of the reference's model zoo.  Every function takes the
parameters as a plain ``dict`` keyed exactly like the reference module's
``state_dict()`` so that fixtures generated from the reference load unchanged.
Index work (row gather, scatter-add) is restated with integer indexing and is
bit-exact; floating point follows the reference's operation order op by op.

Pinning: the reference holds no tests or golden vectors.
The oracle is pinned by ``tests/golden/*.npz``, produced by
``oracle/make_golden.py`` which imports the reference's own classes from
``/root/reference`` in the build container (seeded), and checked in
``tests/test_oracle_golden.py``.

Reference citations are ``file:line`` into the reference repository.
"""
from __future__ import annotations

from typing import Callable, Dict, Sequence, Tuple

import numpy as np
import torch

Params = Dict[str, torch.Tensor]

# (B,45) feature layout defined by data/reader.py:98-112 and used by every
# feature-vector model: col 0 user id, col 1 item id (both stored as floats),
# col 2 age in [0,1], 3:5 gender one-hot, 5:26 occupation one-hot,
# 26:45 genre multi-hot.
COL_USER, COL_ITEM, COL_AGE = 0, 1, 2
SL_GENDER = slice(3, 5)
SL_OCC = slice(5, 26)
SL_GENRE = slice(26, 45)
NUM_FEATURE_COLS = 45


# --------------------------------------------------------------------------
# index primitives (bit-exact)
# --------------------------------------------------------------------------
def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """K1/K3: ``out[..., :] = table[idx[...], :]`` -- what ``nn.Embedding``
    computes at e.g. model/mf.py:24-25, model/din.py:35-36."""
    t = table.detach().cpu().numpy()
    i = idx.detach().cpu().numpy().astype(np.int64)
    if i.size and (i.min() < 0 or i.max() >= t.shape[0]):
        raise IndexError("index out of range in gather_rows")
    return torch.from_numpy(t[i].copy())


def ids_from_float(col: torch.Tensor) -> torch.Tensor:
    """``x[:, c].long()`` (truncation toward zero), model/pnn.py:113-114."""
    return col.to(torch.int64)


def scatter_add_rows(num_rows: int, idx: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """K11: dense ``(V,E)`` gradient of a row gather, accumulated in batch
    order (the order ``embedding_dense_backward`` uses on one CPU thread)."""
    i = idx.reshape(-1).cpu().numpy().astype(np.int64)
    gg = g.reshape(i.shape[0], -1).cpu().numpy()
    out = np.zeros((num_rows, gg.shape[1]), dtype=gg.dtype)
    np.add.at(out, i, gg)
    return torch.from_numpy(out)


def bag_pool(weights: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """K2: the reference's "multi-hot matmul" pooling
    ``torch.matmul(x[:, a:b], table.weight)`` (model/pnn.py:115-118,
    model/ffm.py:48-55, model/deepfm.py:47-51, model/deepcrossing.py:66-68)."""
    return torch.matmul(weights, table)


def _emb(p: Params, name: str, idx: torch.Tensor) -> torch.Tensor:
    # differentiable gather (autograd supplies K11); values equal gather_rows
    return p[name + ".weight"][idx]


def _lin(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.linear(x, p[name + ".weight"], p[name + ".bias"])


def _count(p: Params, prefix: str) -> int:
    n = 0
    while f"{prefix}.{n}.weight" in p:
        n += 1
    return n


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def mf_forward(p: Params, user: torch.Tensor, item: torch.Tensor) -> torch.Tensor:
    """model/mf.py:23-26 -> (B,)"""
    u = _emb(p, "user_embeddings", user)
    v = _emb(p, "item_embeddings", item)
    return torch.sigmoid((u * v).sum(dim=1))


def neuralcf_forward(p: Params, user: torch.Tensor, item: torch.Tensor) -> torch.Tensor:
    """model/neuralcf.py:33-59 -> (B,1)"""
    gmf = _emb(p, "GMF_Embedding_User", user) * _emb(p, "GMF_Embedding_Item", item)
    h = torch.cat([_emb(p, "MLP_Embedding_User", user), _emb(p, "MLP_Embedding_Item", item)], dim=1)
    for k in range(_count(p, "dnn_network")):
        h = torch.relu(_lin(p, f"dnn_network.{k}", h))
    mlp = _lin(p, "linear", h)
    return torch.sigmoid(_lin(p, "linear2", torch.cat([gmf, mlp], dim=1)))


# the 15 field-aware pairs of model/ffm.py:62-80, in the reference's order
FFM_PAIRS: Tuple[Tuple[str, str], ...] = (
    ("age_user", "gender_user"), ("age_user", "occupation_user"), ("age_item", "movie_user"),
    ("age_user", "userid_user"), ("age_item", "itemid_user"),
    ("gender_user", "occupation_user"), ("gender_item", "movie_user"),
    ("gender_user", "userid_user"), ("gender_item", "itemid_user"),
    ("occupation_item", "movie_user"), ("occupation_user", "userid_user"),
    ("occupation_item", "itemid_user"),
    ("movie_user", "userid_item"), ("movie_item", "itemid_item"),
    ("userid_item", "itemid_user"),
)


def ffm_vectors(p: Params, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    """the 12 field-aware vectors of model/ffm.py:48-59"""
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    age = x[:, COL_AGE].unsqueeze(1)
    v = {}
    for f in ("user", "item"):
        v[f"age_{f}"] = bag_pool(age, p[f"age_{f}.weight"])
        v[f"gender_{f}"] = bag_pool(x[:, SL_GENDER], p[f"gender_{f}.weight"])
        v[f"occupation_{f}"] = bag_pool(x[:, SL_OCC], p[f"occupation_{f}.weight"])
        v[f"movie_{f}"] = bag_pool(x[:, SL_GENRE], p[f"movie_{f}.weight"])
        v[f"userid_{f}"] = _emb(p, f"userid_{f}", uid)
        v[f"itemid_{f}"] = _emb(p, f"itemid_{f}", iid)
    return v


def ffm_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/ffm.py:46-86 -> (B,1).  Keeps the reference's quirk of adding the
    cross scalar to all 43 dense inputs before the linear layer (:84-86)."""
    v = ffm_vectors(p, x)
    cross = None
    for a, b in FFM_PAIRS:  # left-to-right sum as in model/ffm.py:82
        d = (v[a] * v[b]).sum(dim=1)
        cross = d if cross is None else cross + d
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    lin = _lin(p, "linear", x[:, 2:] + cross.unsqueeze(1))
    return torch.sigmoid(_emb(p, "user", uid) + _emb(p, "item", iid) + lin)


def six_field_vectors(p: Params, x: torch.Tensor, names: Sequence[str]):
    """[user, item, age, gender, occupation, movie] vectors shared by
    model/pnn.py:113-118 and model/deepfm.py:45-51."""
    nu, ni, na, ng, no, nm = names
    return [
        _emb(p, nu, ids_from_float(x[:, COL_USER])),
        _emb(p, ni, ids_from_float(x[:, COL_ITEM])),
        bag_pool(x[:, COL_AGE].unsqueeze(1), p[na + ".weight"]),
        bag_pool(x[:, SL_GENDER], p[ng + ".weight"]),
        bag_pool(x[:, SL_OCC], p[no + ".weight"]),
        bag_pool(x[:, SL_GENRE], p[nm + ".weight"]),
    ]


def pnn_inner_products(f: Sequence[torch.Tensor]) -> torch.Tensor:
    """model/pnn.py:59-66: p[:, idx(i,j)] = <f_i, f_j>, i<j lexicographic"""
    cols = []
    for i in range(len(f)):
        for j in range(i + 1, len(f)):
            cols.append((f[i] * f[j]).sum(dim=1, keepdim=True))
    return torch.cat(cols, dim=1)


def pnn_forward(p: Params, x: torch.Tensor, mode: str = "in") -> torch.Tensor:
    """model/pnn.py:111-131 (+ ProductLayers :50-79, DNN :18-23) -> (B,1)"""
    f = six_field_vectors(p, x, ("user_embed", "item_embed", "age_embed", "gender_embed",
                                 "occupation_embed", "movie_embed"))
    z = torch.cat(f, dim=1).unsqueeze(0)                      # (1,B,6E)  pnn.py:55
    if mode == "in":
        prod = pnn_inner_products(f)                          # (B,15)
    elif mode == "out":
        s = torch.stack(f).sum(dim=0)                         # (B,E)     pnn.py:69-71
        prod = torch.matmul(s.T, s)                           # (E,E): reduces over batch
    else:
        raise ValueError(mode)
    h = _lin(p, "product.linear1", z) + _lin(p, "product.linear2", prod)
    for k in range(_count(p, "dnn.dnn_network")):
        h = torch.relu(_lin(p, f"dnn.dnn_network.{k}", h))
    return torch.sigmoid(_lin(p, "output", h)).view(-1, 1)


def deepcrossing_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/deepcrossing.py:61-79 (+ ResidualBlock :22-27) -> (B,1)"""
    r = torch.cat([
        _emb(p, "user_embedding", ids_from_float(x[:, COL_USER])),
        _emb(p, "item_embedding", ids_from_float(x[:, COL_ITEM])),
        x[:, COL_AGE].unsqueeze(1),
        bag_pool(x[:, SL_GENDER], p["gender_embedding.weight"]),
        bag_pool(x[:, SL_OCC], p["occupation_embedding.weight"]),
        bag_pool(x[:, SL_GENRE], p["movie_embedding.weight"]),
    ], dim=1)
    for k in range(_count_res(p)):
        h = torch.relu(_lin(p, f"res_layers.{k}.linear1", r))
        r = torch.relu(_lin(p, f"res_layers.{k}.linear2", h) + r)
    return torch.sigmoid(_lin(p, "linear", r))


def stack_5e1(p: Params, x: torch.Tensor) -> torch.Tensor:
    """cat[user, item, age(1), gender, occupation, movie] -> (B, 5E+1): the stack shared by
    model/deepcrossing.py:63-71, model/deepcross.py:61-70 and model/widedeep.py:44-52"""
    return torch.cat([
        _emb(p, "user_embedding", ids_from_float(x[:, COL_USER])),
        _emb(p, "item_embedding", ids_from_float(x[:, COL_ITEM])),
        x[:, COL_AGE].unsqueeze(1),
        bag_pool(x[:, SL_GENDER], p["gender_embedding.weight"]),
        bag_pool(x[:, SL_OCC], p["occupation_embedding.weight"]),
        bag_pool(x[:, SL_GENRE], p["movie_embedding.weight"]),
    ], dim=1)


def deepcross_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/deepcross.py:59-77 (Deep & Cross): CrossNetwork :7-18
    ``x_{l+1} = x0 * (W_l x_l) + b_l + x_l`` (W_l a bias-free d x d Linear), DeepNetwork
    :21-31 (Linear+ReLU after EVERY layer), output Linear(d + H_last, 1) + sigmoid -> (B,1)"""
    x0 = stack_5e1(p, x)
    xl = x0
    n = 0
    while f"cross_network.cross_weights.{n}.weight" in p:
        xl = x0 * (xl @ p[f"cross_network.cross_weights.{n}.weight"].T) + p[f"cross_network.cross_biases.{n}"] + xl
        n += 1
    h = x0
    k = 0
    while f"deep_network.network.{2 * k}.weight" in p:
        h = torch.relu(_lin(p, f"deep_network.network.{2 * k}", h))
        k += 1
    return torch.sigmoid(_lin(p, "output_layer", torch.cat([xl, h], dim=1)))


def widedeep_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/widedeep.py:41-66: deep = Linear(5E+1, H0) WITHOUT activation (:55) then
    Linear+ReLU per pair (:56-58); wide = user(u) + item(i) + Linear(43,1)(x[:,2:]) (:61);
    sigmoid(Linear(2,1)(cat(wide, deep))) -> (B,1)"""
    deep = _lin(p, "linear", stack_5e1(p, x))
    for k in range(_count(p, "dnn_network")):
        deep = torch.relu(_lin(p, f"dnn_network.{k}", deep))
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    wide = _emb(p, "user", uid) + _emb(p, "item", iid) + _lin(p, "wide", x[:, 2:])
    return torch.sigmoid(_lin(p, "output", torch.cat([wide, deep], dim=1)))


def nfm_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/nfm.py:43-73: six vectors as DeepFM; bi-interaction ``sum_{i<j} f_i * f_j`` (:58-61,
    element products summed in pair order) -> ``linear`` WITHOUT activation (:62) -> Linear+ReLU
    per pair (:63-65); wide = user(u) + item(i) + Linear(43,1)(x[:,2:]) (:54);
    sigmoid(Linear(2,1)(cat(wide, deep))) -> (B,1)"""
    f = six_field_vectors(p, x, ("user_embedding", "item_embedding", "age_embedding", "gender_embedding",
                                 "occupation_embedding", "movie_embedding"))
    cross = 0.0
    for i in range(len(f)):
        for j in range(i + 1, len(f)):
            cross = cross + f[i] * f[j]
    deep = _lin(p, "linear", cross)
    for k in range(_count(p, "dnn_network")):
        deep = torch.relu(_lin(p, f"dnn_network.{k}", deep))
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    wide = _emb(p, "user", uid) + _emb(p, "item", iid) + _lin(p, "wide", x[:, 2:])
    return torch.sigmoid(_lin(p, "output", torch.cat([wide, deep], dim=1)))


def afm_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/afm.py:41-71: vectors [user, item, age broadcast to E (:54), gender, occupation, movie];
    the 15 element products stacked (B,15,E) (:56-60); scores = relu(P W + b) h (:63-64), softmax over
    the pairs, weighted sum (:65), Linear(E,1) (:66); sigmoid(linear part + cross part) -> (B,1)"""
    e = p["user_embedding.weight"].shape[1]
    f = [
        _emb(p, "user_embedding", ids_from_float(x[:, COL_USER])),
        _emb(p, "item_embedding", ids_from_float(x[:, COL_ITEM])),
        x[:, COL_AGE].unsqueeze(1).expand(-1, e),
        bag_pool(x[:, SL_GENDER], p["gender_embedding.weight"]),
        bag_pool(x[:, SL_OCC], p["occupation_embedding.weight"]),
        bag_pool(x[:, SL_GENRE], p["movie_embedding.weight"]),
    ]
    pairs = torch.stack([f[i] * f[j] for i in range(6) for j in range(i + 1, 6)], dim=1)   # (B,15,E)
    scores = torch.relu(torch.matmul(pairs, p["attention_W"]) + p["attention_b"])
    weights = torch.softmax(torch.matmul(scores, p["attention_h"]), dim=1)
    pooled = torch.sum(weights * pairs, dim=1)
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    linear = _emb(p, "user", uid) + _emb(p, "item", iid) + _lin(p, "linear", x[:, 2:])
    return torch.sigmoid(linear + _lin(p, "output_layer", pooled))


def autorec_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/autorec.py:11-14: sigmoid(decoder(sigmoid(encoder(x)))) on rows of the rating matrix"""
    return torch.sigmoid(_lin(p, "decoder", torch.sigmoid(_lin(p, "encoder", x))))


def lr_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/lr.py:24-25: sigmoid(user(u) + item(i) + Linear(43,1)(x[:,2:])) -> (B,1)"""
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    return torch.sigmoid(_emb(p, "user", uid) + _emb(p, "item", iid) + _lin(p, "linear", x[:, 2:]))


def _count_res(p: Params) -> int:
    n = 0
    while f"res_layers.{n}.linear1.weight" in p:
        n += 1
    return n


def fm_second_order(f: Sequence[torch.Tensor]) -> torch.Tensor:
    """model/deepfm.py:71-76: 0.5*sum_e[(sum_f v)^2 - sum_f v^2] -> (B,)"""
    feats = torch.stack(list(f), dim=1)
    return 0.5 * torch.sum(torch.sum(feats, dim=1) ** 2 - torch.sum(feats ** 2, dim=1), dim=1)


def deepfm_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/deepfm.py:43-83 -> (B,1).  Note: first deep layer has no
    activation (:57) and the last (1-unit) deep layer has a ReLU (:58-60)."""
    f = six_field_vectors(p, x, ("user_embedding", "item_embedding", "age_embedding",
                                 "gender_embedding", "occupation_embedding", "movie_embedding"))
    h = _lin(p, "linear", torch.cat(f, dim=1))
    for k in range(_count(p, "dnn_network")):
        h = torch.relu(_lin(p, f"dnn_network.{k}", h))
    uid, iid = ids_from_float(x[:, COL_USER]), ids_from_float(x[:, COL_ITEM])
    wide = _emb(p, "user", uid) + _emb(p, "item", iid) + _lin(p, "wide", x[:, 2:])
    wide = wide + fm_second_order(f).unsqueeze(1)
    return torch.sigmoid(_lin(p, "output", torch.cat([wide, h], dim=1)))


# --------------------------------------------------------------------------
# generalised N-field DeepFM / PNN (BASELINE configs[2]: "26 fields x 1e6 vocab").  The reference hard-codes six
# fields (model/deepfm.py:14-17, model/pnn.py:87-92); these restate the SAME forward with the six vectors replaced
# by F single-id lookups ``embeddings.f[x[:, f].long()]`` -- the pattern of deepfm.py:45-46 / pnn.py:113-114 applied
# to every column -- and everything downstream unchanged, built from the same pinned pieces (_emb, _lin,
# fm_second_order, pnn_inner_products).  The reference cannot build such a model, so no reference fixture exists
# at F != 6; the restatements are pinned THROUGH the pinned six-field oracle instead
# (tests/test_oracle_golden.py::test_*_fields_oracle_reduces_to_the_pinned_six_field_oracle: with the bag tables
# zeroed the reference model is a two-id-field model padded with zero vectors, which the F = 6 restatement must
# reproduce on the reference fixture's own parameters: outputs, loss, gradients).
# --------------------------------------------------------------------------
def field_vectors(p: Params, x: torch.Tensor) -> Sequence[torch.Tensor]:
    """F id lookups: column f of ``x`` (ids as floats or int64) into ``embeddings.f``"""
    n = _count(p, "embeddings")
    ids = x if not x.is_floating_point() else None
    return [_emb(p, f"embeddings.{f}", ids[:, f] if ids is not None else ids_from_float(x[:, f])) for f in range(n)]


def deepfm_fields_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/deepfm.py:43-83 over F id fields: first order = sum_f first_order.f[id_f] + first_order_bias (the
    id terms of deepfm.py:63; with no dense columns the ``wide`` Linear reduces to its bias)"""
    f = field_vectors(p, x)
    h = _lin(p, "linear", torch.cat(f, dim=1))
    for k in range(_count(p, "dnn_network")):
        h = torch.relu(_lin(p, f"dnn_network.{k}", h))
    ids = x if not x.is_floating_point() else torch.stack([ids_from_float(x[:, c]) for c in range(x.shape[1])], 1)
    wide = _emb(p, "first_order.0", ids[:, 0])
    for c in range(1, len(f)):
        wide = wide + _emb(p, f"first_order.{c}", ids[:, c])
    wide = wide + p["first_order_bias"]
    wide = wide + fm_second_order(f).unsqueeze(1)
    return torch.sigmoid(_lin(p, "output", torch.cat([wide, h], dim=1)))


def pnn_fields_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model/pnn.py:111-131 (inner mode) over F id fields: F(F-1)/2 inner products, i<j lexicographic"""
    f = field_vectors(p, x)
    z = torch.cat(f, dim=1).unsqueeze(0)
    h = _lin(p, "product.linear1", z) + _lin(p, "product.linear2", pnn_inner_products(f))
    for k in range(_count(p, "dnn.dnn_network")):
        h = torch.relu(_lin(p, f"dnn.dnn_network.{k}", h))
    return torch.sigmoid(_lin(p, "output", h)).view(-1, 1)


def din_attention(p: Params, prefix: str, hist: torch.Tensor, target: torch.Tensor):
    """model/din.py:35-44 / model/dien.py:25-34: returns (H, t, a) with
    a = softmax over L of MLP([h, h-t, t]); no padding mask (pad id 0 is a
    real row)."""
    table = prefix + "item_embedding"
    t = _emb(p, table, target)                                   # (B,E)
    h = _emb(p, table, hist)                                     # (B,L,E)
    te = t.unsqueeze(1).expand_as(h)
    s = torch.cat([h, h - te, te], dim=-1)
    s = torch.relu(_lin(p, prefix + "attention.0", s))
    s = torch.relu(_lin(p, prefix + "attention.2", s))
    s = _lin(p, prefix + "attention.4", s).squeeze(-1)           # (B,L)
    return h, t, torch.softmax(s, dim=-1)


def _fc3_sigmoid(p: Params, x: torch.Tensor) -> torch.Tensor:
    x = torch.relu(_lin(p, "fc.0", x))
    x = torch.relu(_lin(p, "fc.2", x))
    return torch.sigmoid(_lin(p, "fc.4", x))


def din_forward(p: Params, hist: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """model/din.py:33-53 -> (B,1)"""
    h, t, a = din_attention(p, "", hist, target)
    pooled = (h * a.unsqueeze(-1)).sum(dim=1)
    return _fc3_sigmoid(p, torch.cat([pooled, t], dim=1))


def gru_last_hidden(w_ih, w_hh, b_ih, b_hh, x: torch.Tensor) -> torch.Tensor:
    """single-layer batch_first ``nn.GRU`` with h0 = 0, PyTorch gate order
    (r, z, n): n = tanh(W_in x + b_in + r*(W_hn h + b_hn)),
    h' = (1-z)*n + z*h.  Replaces model/dien.py:47,61; returns hidden[-1]."""
    bsz, steps, _ = x.shape
    hid = w_hh.shape[1]
    h = x.new_zeros(bsz, hid)
    for s in range(steps):
        gi = torch.nn.functional.linear(x[:, s], w_ih, b_ih)
        gh = torch.nn.functional.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
        z = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
        n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
        h = (1.0 - z) * n + z * h
    return h


def dien_forward(p: Params, hist: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """model/dien.py:57-68 (+ :23-39) -> (B,1)"""
    h, t, a = din_attention(p, "din.", hist, target)
    seq = h * a.unsqueeze(-1)                                    # un-summed, dien.py:37
    last = gru_last_hidden(p["interest_evolution.weight_ih_l0"], p["interest_evolution.weight_hh_l0"],
                           p["interest_evolution.bias_ih_l0"], p["interest_evolution.bias_hh_l0"], seq)
    return _fc3_sigmoid(p, torch.cat([last, t], dim=-1))


def bce_loss(prob: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``torch.nn.BCELoss()`` (mean) as every script uses it
    (e.g. scripts/pnn.py:54): log terms clamped at -100."""
    lp = torch.clamp(torch.log(prob), min=-100.0)
    l1p = torch.clamp(torch.log(1.0 - prob), min=-100.0)
    return -(y * lp + (1.0 - y) * l1p).mean()


def adam_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float = 1e-3,
                betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, rows: torch.Tensor = None) -> None:
    """``torch.optim.Adam`` (no amsgrad) restated, in place: the update every reference script applies
    (``optim.Adam(model.parameters(), lr, weight_decay=1e-5)``, e.g. scripts/pnn.py:55).  With ``rows`` it is the
    LAZY row-wise rule of the build's opt-in sparse mode: only the listed rows are decayed and
    updated, with the parameter's global step count in the bias corrections; every other row of p, m, v is
    untouched.  Not a reference semantics: pinned by construction (same formula on a row subset)."""
    b1, b2 = betas
    sel = slice(None) if rows is None else rows
    gr = g[sel] + weight_decay * p[sel]
    m[sel] = m[sel] + (1.0 - b1) * (gr - m[sel])
    v[sel] = v[sel] * b2 + (1.0 - b2) * gr * gr
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    p[sel] = p[sel] - (lr / bc1) * (m[sel] / (v[sel].sqrt() / bc2 ** 0.5 + eps))


# ---------------------------------------------------------------------------------------------------------------
# recommendation(): the reference's per-user ranking loops, restated.  Returns (ids (users, k), scores (users, n)).
# ---------------------------------------------------------------------------------------------------------------
def recommend_ids(model: str, p: Params, num_users: int, num_items: int):
    """model/mf.py:28-35 (ranks the raw dot products, k = num_items) and model/neuralcf.py:61-72 (one forward per
    user over every item, k = num_items)"""
    if model == "mf":
        scores = torch.matmul(p["user_embeddings.weight"][:num_users], p["item_embeddings.weight"][:num_items].T)
    else:
        items = torch.arange(num_items)
        scores = torch.stack([FORWARDS[model](p, torch.full((num_items,), u), items).view(-1) for u in range(num_users)])
    return torch.topk(scores, num_items, dim=1).indices, scores


def recommend_frame(model: str, p: Params, num_users: int, frame: torch.Tensor, k: int):
    """model/pnn.py:133-143, model/deepfm.py:85-95: the rows of the (pairs, 45) user_item frame that carry user u,
    scored by one forward, ranked by position within those rows"""
    ids, scores = [], []
    for u in range(num_users):
        rows = frame[frame[:, 0] == u]
        s = FORWARDS[model](p, rows).view(-1)
        scores.append(s)
        ids.append(torch.topk(s, k).indices)
    return torch.stack(ids), torch.stack(scores)


def recommend_hist(model: str, p: Params, num_users: int, num_items: int, hist_list, k: int):
    """model/din.py:55-66, model/dien.py:70-81: the user's whole history repeated against every item"""
    ids, scores = [], []
    targets = torch.arange(num_items)
    for u in range(num_users):
        hist = torch.tensor(hist_list[u]).repeat(num_items, 1)
        s = FORWARDS[model](p, hist, targets).view(-1)
        scores.append(s)
        ids.append(torch.topk(s, k).indices)
    return torch.stack(ids), torch.stack(scores)


FORWARDS: Dict[str, Callable[..., torch.Tensor]] = {
    "mf": mf_forward, "neuralcf": neuralcf_forward, "ffm": ffm_forward, "pnn": pnn_forward,
    "deepcrossing": deepcrossing_forward, "deepfm": deepfm_forward, "din": din_forward,
    "dien": dien_forward, "deepfm_fields": deepfm_fields_forward, "pnn_fields": pnn_fields_forward, "deepcross": deepcross_forward, "widedeep": widedeep_forward, "lr": lr_forward, "nfm": nfm_forward, "afm": afm_forward, "autorec": autorec_forward,
}


def step(model: str, params: Params, inputs: Sequence[torch.Tensor], y: torch.Tensor,
         dtype: torch.dtype = torch.float32, **kw):
    """One Trainer.train_loop body without the optimizer (trainer/trainer.py:
    30-38): forward, BCELoss, backward.  Returns (prob, loss, grads-by-name).
    ``dtype=torch.float64`` gives the tie-break reference."""
    leaf = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    ins = [t.to(dtype) if t.is_floating_point() else t for t in inputs]
    prob = FORWARDS[model](leaf, *ins, **kw)
    loss = bce_loss(prob, y.to(dtype))
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in leaf.items()}
    return prob.detach(), loss.detach(), grads
