"""CPU: pin the oracle (oracle/ctr_oracle.py) to outputs of the reference's own
classes (tests/golden/*.npz, written by oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import ctr_oracle as orc


def _kw(meta):
    if meta["model"] == "pnn" and meta["kwargs"].get("model") == "out":
        return {"mode": "out"}
    return {}


@pytest.mark.parametrize("name", gu.names())
def test_oracle_matches_reference_forward_loss_grads(name):
    g = gu.load(name)
    torch.set_num_threads(1)
    prob, loss, grads = orc.step(g["meta"]["model"], g["params"], g["inputs"], g["y"], **_kw(g["meta"]))
    # same ATen ops in the same order as the reference => equal to the last bit
    # on this build; keep a 1e-6 relative guard for other torch CPU builds
    torch.testing.assert_close(prob, g["prob"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-6, atol=1e-7)
    assert set(grads) == set(g["grads"])
    for k in grads:
        torch.testing.assert_close(grads[k], g["grads"][k], rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("name", gu.names())
def test_fp64_oracle_agrees_with_fp32_reference(name):
    g = gu.load(name)
    prob64, loss64, _ = orc.step(g["meta"]["model"], g["params"], g["inputs"], g["y"],
                                 dtype=torch.float64, **_kw(g["meta"]))
    torch.testing.assert_close(prob64.float(), g["prob"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss64.float(), g["loss"], rtol=1e-5, atol=1e-6)


def test_gather_is_bit_exact_and_checks_range():
    t = torch.randn(50, 7)
    idx = torch.randint(0, 50, (4, 9))
    out = orc.gather_rows(t, idx)
    assert out.shape == (4, 9, 7)
    assert np.array_equal(out.numpy(), t.numpy()[idx.numpy()])
    with pytest.raises(IndexError):
        orc.gather_rows(t, torch.tensor([50]))
    assert orc.gather_rows(t, torch.zeros(0, dtype=torch.int64)).shape == (0, 7)


def test_one_hot_bag_equals_row_gather_bitwise():
    # a one-hot slice through the matmul reproduces the row
    t = torch.randn(21, 16)
    ids = torch.randint(0, 21, (300,))
    onehot = torch.zeros(300, 21)
    onehot[torch.arange(300), ids] = 1.0
    assert torch.equal(orc.bag_pool(onehot, t), orc.gather_rows(t, ids))


def test_scatter_add_matches_autograd_dense_grad():
    torch.set_num_threads(1)
    w = torch.randn(13, 5, requires_grad=True)
    idx = torch.randint(0, 13, (200,))
    g = torch.randn(200, 5)
    w[idx].backward(g)
    torch.testing.assert_close(orc.scatter_add_rows(13, idx, g), w.grad, rtol=1e-6, atol=1e-6)


def test_gru_restatement_equals_nn_gru():
    torch.manual_seed(0)
    gru = torch.nn.GRU(6, 6, batch_first=True)
    x = torch.randn(5, 11, 6)
    _, h = gru(x)
    mine = orc.gru_last_hidden(gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0, x)
    torch.testing.assert_close(mine, h[-1], rtol=1e-5, atol=1e-6)


def test_bce_clamps_log_at_minus_100():
    p = torch.tensor([0.0, 1.0, 0.5])
    y = torch.tensor([1.0, 0.0, 1.0])
    assert torch.equal(orc.bce_loss(p, y), torch.nn.BCELoss()(p, y))


# ---------------------------------------------------------------------------
# N-field generalisation (BASELINE configs[2]): the reference cannot build such a model, so the F-field
# restatements are pinned THROUGH the pinned six-field oracle: with the four bag tables (and the dense part of
# the wide Linear) zeroed, the reference model IS a two-id-field model padded with four zero vectors, and the
# F = 6 field restatement with one-row zero tables in fields 2..5 must reproduce it -- outputs, loss and the
# gradients of every shared parameter -- on the reference fixture's own parameters and inputs.
# ---------------------------------------------------------------------------
def _zero_bags(params, names):
    p = {k: v.clone() for k, v in params.items()}
    for n in names:
        p[n + ".weight"].zero_()
    return p


@pytest.mark.parametrize("name", ["deepfm_s0", "deepfm_b37"])
def test_deepfm_fields_oracle_reduces_to_the_pinned_six_field_oracle(name):
    g = gu.load(name)
    x, y = g["inputs"][0], g["y"]
    bags = ["age_embedding", "gender_embedding", "occupation_embedding", "movie_embedding"]
    p6 = _zero_bags(g["params"], bags + ["wide"])          # wide.weight zeroed, wide.bias kept
    prob6, loss6, grads6 = orc.step("deepfm", p6, [x], y)
    dim = p6["user_embedding.weight"].shape[1]
    pf = {"embeddings.0.weight": p6["user_embedding.weight"], "embeddings.1.weight": p6["item_embedding.weight"],
          "first_order.0.weight": p6["user.weight"], "first_order.1.weight": p6["item.weight"],
          "first_order_bias": p6["wide.bias"]}
    for f in range(2, 6):
        pf[f"embeddings.{f}.weight"] = torch.zeros(1, dim)
        pf[f"first_order.{f}.weight"] = torch.zeros(1, 1)
    for k, v in p6.items():
        if k.startswith(("linear.", "dnn_network.", "output.")):
            pf[k] = v
    ids = torch.cat([x[:, :2].long(), torch.zeros(x.shape[0], 4, dtype=torch.int64)], 1)
    probf, lossf, gradsf = orc.step("deepfm_fields", pf, [ids], y)
    torch.testing.assert_close(probf, prob6, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(lossf, loss6, rtol=1e-6, atol=1e-7)
    same = {"embeddings.0.weight": "user_embedding.weight", "embeddings.1.weight": "item_embedding.weight",
            "first_order.0.weight": "user.weight", "first_order.1.weight": "item.weight", "first_order_bias": "wide.bias"}
    same.update({k: k for k in pf if k.startswith(("linear.", "dnn_network.", "output."))})
    for kf, k6 in same.items():
        torch.testing.assert_close(gradsf[kf], grads6[k6], rtol=1e-5, atol=1e-7, msg=lambda m, kf=kf: f"{kf}: {m}")
    # ids carried as floats (the reference's own convention for its two id columns) give the same result
    probx, _, _ = orc.step("deepfm_fields", pf, [ids.float()], y)
    assert torch.equal(probx, probf)


@pytest.mark.parametrize("name", ["pnn_s0", "pnn_b37"])
def test_pnn_fields_oracle_reduces_to_the_pinned_six_field_oracle(name):
    g = gu.load(name)
    x, y = g["inputs"][0], g["y"]
    p6 = _zero_bags(g["params"], ["age_embed", "gender_embed", "occupation_embed", "movie_embed"])
    prob6, loss6, grads6 = orc.step("pnn", p6, [x], y)
    dim = p6["user_embed.weight"].shape[1]
    pf = {"embeddings.0.weight": p6["user_embed.weight"], "embeddings.1.weight": p6["item_embed.weight"]}
    for f in range(2, 6):
        pf[f"embeddings.{f}.weight"] = torch.zeros(1, dim)
    for k, v in p6.items():
        if k.startswith(("product.", "dnn.", "output.")):
            pf[k] = v
    ids = torch.cat([x[:, :2].long(), torch.zeros(x.shape[0], 4, dtype=torch.int64)], 1)
    probf, lossf, gradsf = orc.step("pnn_fields", pf, [ids], y)
    torch.testing.assert_close(probf, prob6, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(lossf, loss6, rtol=1e-6, atol=1e-7)
    same = {"embeddings.0.weight": "user_embed.weight", "embeddings.1.weight": "item_embed.weight"}
    same.update({k: k for k in pf if k.startswith(("product.", "dnn.", "output."))})
    for kf, k6 in same.items():
        torch.testing.assert_close(gradsf[kf], grads6[k6], rtol=1e-5, atol=1e-7, msg=lambda m, kf=kf: f"{kf}: {m}")


@pytest.mark.parametrize("name", gu.rec_names())
def test_oracle_recommendation_matches_the_reference(name):
    """the reference's own recommendation() output (ids) and the scores it ranked, against the oracle's
    restatement of the per-user loops on the fixture's state_dict"""
    g = gu.load_rec(name)
    model, nu, ni = g["meta"]["model"], g["num_users"], g["num_items"]
    with torch.no_grad():
        if g["meta"]["kind"] == "ids":
            ids, scores = orc.recommend_ids(model, g["params"], nu, ni)
        elif g["meta"]["kind"] == "frame":
            ids, scores = orc.recommend_frame(model, g["params"], nu, torch.from_numpy(g["frame"]), g["k"])
        else:
            ids, scores = orc.recommend_hist(model, g["params"], nu, ni, g["hist_list"], g["k"])
    torch.testing.assert_close(scores, torch.from_numpy(g["scores"]), rtol=1e-6, atol=1e-7)
    gu.assert_same_ranking(ids.numpy(), g["topk"], g["scores"])
