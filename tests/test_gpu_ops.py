"""GPU: op-level parity of libctrhip kernels (called through the C ABI) against
the CPU oracle / an fp64 torch reference.

Tolerances: index work (row gather, one-hot bags, dense copies) is bit-exact;
fp32 sums are compared at rtol 1e-5 (north_star: "within 1e-5 relative")."""
import os
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from deeplearningrecommendationsystem_amd import ops as o
    return o


def _lib():
    from deeplearningrecommendationsystem_amd import _lib
    return _lib


def _feature_specs(ops, tabs, widths_e, ldo_cols=True):
    L = _lib()
    e = widths_e
    return [
        ops.FieldSpec(L.FIELD_ID_F32, e, 0 * e, table=tabs["user"], src_col=0),
        ops.FieldSpec(L.FIELD_ID_F32, e, 1 * e, table=tabs["item"], src_col=1),
        ops.FieldSpec(L.FIELD_BAG, e, 2 * e, table=tabs["age"], src_col=2, bag_size=1),
        ops.FieldSpec(L.FIELD_BAG, e, 3 * e, table=tabs["gender"], src_col=3, bag_size=2),
        ops.FieldSpec(L.FIELD_BAG, e, 4 * e, table=tabs["occ"], src_col=5, bag_size=21),
        ops.FieldSpec(L.FIELD_BAG, e, 5 * e, table=tabs["genre"], src_col=26, bag_size=19),
    ]


@pytest.mark.parametrize("e,batch", [(16, 1000), (8, 37), (4, 1), (64, 4096), (6, 129), (1, 77), (128, 2048), (256, 300)])
def test_embed_stage_forward_and_backward(ops, e, batch):
    from deeplearningrecommendationsystem_amd import synth
    g = synth.generator(e * 1000 + batch)
    x = synth.feature_batch(batch, 50, 70, g, zero_genre_rows=min(3, batch))
    tabs = {k: torch.randn(v, e, generator=g) for k, v in
            dict(user=50, item=70, age=1, gender=2, occ=21, genre=19).items()}
    dt = {k: v.to(DEV) for k, v in tabs.items()}
    specs = _feature_specs(ops, dt, e)
    out = torch.full((batch, 6 * e), float("nan"), device=DEV)
    ops.embed_fwd(specs, x.to(DEV), batch, out)
    out = out.cpu()
    uid, iid = x[:, 0].long(), x[:, 1].long()
    # id rows and one-hot bags: bit-exact
    assert torch.equal(out[:, 0:e], orc.gather_rows(tabs["user"], uid))
    assert torch.equal(out[:, e:2 * e], orc.gather_rows(tabs["item"], iid))
    assert torch.equal(out[:, 3 * e:4 * e], orc.gather_rows(tabs["gender"], x[:, 3:5].argmax(1)))
    assert torch.equal(out[:, 4 * e:5 * e], orc.gather_rows(tabs["occ"], x[:, 5:26].argmax(1)))
    # real-weighted / multi-hot bags: fp32 sums
    torch.testing.assert_close(out[:, 2 * e:3 * e], orc.bag_pool(x[:, 2:3], tabs["age"]), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(out[:, 5 * e:6 * e], orc.bag_pool(x[:, 26:45], tabs["genre"]), rtol=1e-6, atol=1e-6)

    # backward: dense grads of every table vs autograd of the oracle restatement
    gout = torch.randn(batch, 6 * e, generator=g)
    # (fp64 reference: the bag tables' gradients are sums over the whole batch)
    leaf = {k: v.double().requires_grad_(True) for k, v in tabs.items()}
    xd = x.double()
    ref = torch.cat([leaf["user"][uid], leaf["item"][iid], xd[:, 2:3] @ leaf["age"], xd[:, 3:5] @ leaf["gender"],
                     xd[:, 5:26] @ leaf["occ"], xd[:, 26:45] @ leaf["genre"]], dim=1)
    ref.backward(gout.double())
    grads = {id(dt[k]): torch.zeros_like(dt[k]) for k in dt}
    ops.embed_bwd(specs, x.to(DEV), batch, gout.to(DEV), grads)
    for k in tabs:
        torch.testing.assert_close(grads[id(dt[k])].cpu(), leaf[k].grad.float(), rtol=1e-5,
                                   atol=1e-6 + 3e-6 * batch ** 0.5, msg=lambda m, k=k: f"grad of {k}: {m}")


def test_embed_id_i64_prod_dense_and_strided_output(ops):
    L = _lib()
    g = torch.Generator().manual_seed(5)
    batch, e = 333, 12
    t1, t2 = torch.randn(40, e, generator=g), torch.randn(30, e, generator=g)
    i1 = torch.randint(0, 40, (batch,), generator=g)
    i2 = torch.randint(0, 30, (batch,), generator=g)
    x = torch.randn(batch, 7, generator=g)
    d1, d2, di1, di2 = t1.to(DEV), t2.to(DEV), i1.to(DEV), i2.to(DEV)
    specs = [
        ops.FieldSpec(L.FIELD_ID_I64, e, 0, table=d1, idx=di1),
        ops.FieldSpec(L.FIELD_DENSE, 3, e, src_col=2),
        ops.FieldSpec(L.FIELD_PROD_I64, e, e + 3, table=d1, idx=di1, table2=d2, idx2=di2),
    ]
    wide = torch.zeros(batch, 2 * e + 3 + 5, device=DEV)  # wider than the fields: ld > used columns
    ops.embed_fwd(specs, x.to(DEV), batch, wide)
    w = wide.cpu()
    assert torch.equal(w[:, :e], t1[i1])
    assert torch.equal(w[:, e:e + 3], x[:, 2:5])
    assert torch.equal(w[:, e + 3:2 * e + 3], t1[i1] * t2[i2])
    assert torch.equal(w[:, 2 * e + 3:], torch.zeros(batch, 5))

    gout = torch.randn(batch, 2 * e + 8, generator=g)
    l1, l2 = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
    (torch.cat([l1[i1], x[:, 2:5], l1[i1] * l2[i2]], 1) * gout[:, :2 * e + 3]).sum().backward()
    grads = {id(d1): torch.zeros_like(d1), id(d2): torch.zeros_like(d2)}
    ops.embed_bwd(specs, x.to(DEV), batch, gout.to(DEV), grads)
    torch.testing.assert_close(grads[id(d1)].cpu(), l1.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(grads[id(d2)].cpu(), l2.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("batch,e", [(20000, 32), (65536, 64), (4099, 4)])
def test_embed_backward_small_tables_sorted_path(ops, batch, e):
    # tables much smaller than the batch take the counting-sort + run-reduce backward
    # (embed_sorted.hip); same contract as the atomic scatter: dense grads, fp32 sums
    L = _lib()
    g = torch.Generator().manual_seed(batch + e)
    v1, v2 = 300, 500
    t1, t2 = torch.randn(v1, e, generator=g), torch.randn(v2, e, generator=g)
    p1, p2 = torch.randn(v1, e, generator=g), torch.randn(v2, e, generator=g)
    # skewed ids (a few very long runs) incl. both ends of the range
    i1 = (torch.rand(batch, generator=g) ** 3 * v1).long().clamp_(0, v1 - 1)
    i2 = torch.randint(0, v2, (batch,), generator=g)
    i1[0], i1[1], i2[0], i2[1] = 0, v1 - 1, 0, v2 - 1
    dev = {k: v.to(DEV) for k, v in dict(t1=t1, t2=t2, p1=p1, p2=p2, i1=i1, i2=i2).items()}
    specs = [
        ops.FieldSpec(L.FIELD_ID_I64, e, 0, table=dev["t1"], idx=dev["i1"]),
        ops.FieldSpec(L.FIELD_ID_I64, e, e, table=dev["t2"], idx=dev["i2"]),
        ops.FieldSpec(L.FIELD_PROD_I64, e, 2 * e, table=dev["p1"], idx=dev["i1"], table2=dev["p2"], idx2=dev["i2"]),
    ]
    gout = torch.randn(batch, 3 * e, generator=g)
    leaf = {k: v.double().requires_grad_(True) for k, v in dict(t1=t1, t2=t2, p1=p1, p2=p2).items()}
    ref = torch.cat([leaf["t1"][i1], leaf["t2"][i2], leaf["p1"][i1] * leaf["p2"][i2]], 1)
    ref.backward(gout.double())
    grads = {id(dev[k]): torch.zeros_like(dev[k]) for k in ("t1", "t2", "p1", "p2")}
    ops.embed_bwd(specs, None, batch, gout.to(DEV), grads)
    for k in ("t1", "t2", "p1", "p2"):
        torch.testing.assert_close(grads[id(dev[k])].cpu(), leaf[k].grad.float(), rtol=1e-5,
                                   atol=1e-6 + 3e-6 * batch ** 0.5, msg=lambda m, k=k: f"grad of {k}: {m}")


def test_embed_backward_sorted_path_at_the_largest_vocab(ops):
    # 8192 rows: the largest vocabulary the counting sort takes (LDS histogram and cursors of 32 KB)
    L = _lib()
    g = torch.Generator().manual_seed(99)
    batch, e, v = 70000, 8, 8192
    t = torch.randn(v, e, generator=g)
    idx = torch.randint(0, v, (batch,), generator=g)
    idx[:3] = torch.tensor([0, v - 1, v - 1])
    gout = torch.randn(batch, e, generator=g)
    want = torch.zeros(v, e, dtype=torch.float64).index_add_(0, idx, gout.double())
    dt = t.to(DEV)
    grads = {id(dt): torch.zeros_like(dt)}
    ops.embed_bwd([ops.FieldSpec(L.FIELD_ID_I64, e, 0, table=dt, idx=idx.to(DEV))], None, batch, gout.to(DEV), grads)
    torch.testing.assert_close(grads[id(dt)].cpu(), want.float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("batch,e", [(1000, 64), (77, 16), (5, 4)])
def test_embed_forward_row_fields_from_separate_id_columns(ops, batch, e):
    # NeuralCF's field set (model/neuralcf.py:35-38): two id fields and the product of two rows, ids in separate
    # int64 vectors -- bit-exact copies / products, out-of-range ids of either factor flagged and read as row 0
    L = _lib()
    g = torch.Generator().manual_seed(batch)
    v1, v2 = 40, 70
    t1, t2 = torch.randn(v1, e, generator=g), torch.randn(v2, e, generator=g)
    p1, p2 = torch.randn(v1, e, generator=g), torch.randn(v2, e, generator=g)
    i1, i2 = torch.randint(0, v1, (batch,), generator=g), torch.randint(0, v2, (batch,), generator=g)
    dev = {k: x.to(DEV) for k, x in dict(t1=t1, t2=t2, p1=p1, p2=p2, i1=i1, i2=i2).items()}

    def run(a, b):
        specs = [ops.FieldSpec(L.FIELD_ID_I64, e, 0, table=dev["t1"], idx=a),
                 ops.FieldSpec(L.FIELD_ID_I64, e, e, table=dev["t2"], idx=b),
                 ops.FieldSpec(L.FIELD_PROD_I64, e, 2 * e, table=dev["p1"], idx=a, table2=dev["p2"], idx2=b)]
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        out = torch.full((batch, 3 * e + 4), float("nan"), device=DEV)
        ops.embed_fwd(specs, None, batch, out[:, :3 * e], flag)
        return out.cpu(), int(flag.item())

    out, flag = run(dev["i1"], dev["i2"])
    assert flag == 0
    assert torch.equal(out[:, :3 * e], torch.cat([t1[i1], t2[i2], p1[i1] * p2[i2]], 1))
    assert torch.isnan(out[:, 3 * e:]).all()
    b1, b2 = i1.clone(), i2.clone()
    b1[0], b2[batch - 1] = v1, -3
    out, flag = run(b1.to(DEV), b2.to(DEV))
    assert flag == 1
    c1, c2 = b1.clone(), b2.clone()
    c1[0], c2[batch - 1] = 0, 0
    assert torch.equal(out[:, :3 * e], torch.cat([t1[c1], t2[c2], p1[c1] * p2[c2]], 1))


def test_embed_sequence_gather_is_bit_exact(ops):
    # K3: (B,L) history gathered column by column through idx_stride
    L = _lib()
    g = torch.Generator().manual_seed(6)
    batch, length, e = 50, 7, 8
    table = torch.randn(100, e, generator=g)
    hist = torch.randint(0, 100, (batch, length), generator=g)
    dt, dh = table.to(DEV), hist.to(DEV)
    specs = [ops.FieldSpec(L.FIELD_ID_I64, e, l * e, table=dt, idx=dh[:, l], idx_stride=length)
             for l in range(length)]
    out = torch.empty(batch, length * e, device=DEV)
    ops.embed_fwd(specs, None, batch, out)
    assert torch.equal(out.cpu().view(batch, length, e), orc.gather_rows(table, hist))


def test_embed_out_of_range_index_sets_flag_and_does_not_fault(ops):
    L = _lib()
    table = torch.randn(10, 4).to(DEV)
    idx = torch.tensor([1, 10, -1, 3]).to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.empty(4, 4, device=DEV)
    ops.embed_fwd([ops.FieldSpec(L.FIELD_ID_I64, 4, 0, table=table, idx=idx)], None, 4, out, flag)
    assert flag.item() == 1
    assert torch.equal(out[1], table[0]) and torch.equal(out[0], table[1])


def test_embed_empty_batch(ops):
    L = _lib()
    table = torch.randn(10, 4).to(DEV)
    idx = torch.zeros(0, dtype=torch.int64, device=DEV)
    out = torch.empty(0, 4, device=DEV)
    ops.embed_fwd([ops.FieldSpec(L.FIELD_ID_I64, 4, 0, table=table, idx=idx)], None, 0, out)


LINEAR_SHAPES = [
    # (m, n, k)
    (1, 1, 1), (37, 5, 3), (64, 32, 32), (129, 33, 31), (1000, 64, 128), (513, 1, 128),
    (300, 128, 96), (2048, 161, 256), (700, 256, 161), (257, 512, 48), (4096, 8, 16),
    # few units on both sides over a long batch: the streaming kernels of linear_skinny.hip
    (70001, 48, 16), (65536, 64, 32), (66000, 8, 8), (65537, 4, 16),
    # direct-to-LDS kernels (gemm_dlds*.hip): sizes that are not multiples of the 16-float step / 4-float
    # chunk on either side (shifted last chunks, fragment masks), both dW orientations, split launches
    (4200, 641, 641), (4160, 256, 161), (4100, 385, 193),   # widths just above a multiple of 128: main + tail launches
    (4096, 256, 256), (4500, 512, 416), (4100, 256, 272), (8000, 768, 48),   # 256 x 256 macro tile (gemm_wide.hip): n % 256 == 0, k % 16 == 0, k >= 256
    (4500, 130, 131), (4100, 97, 35), (4200, 33, 103), (4097, 19, 30), (5000, 200, 17), (4099, 161, 289), (3000, 256, 15), (3000, 6, 5), (2000, 40, 4),
    # single-unit layers over a long batch: four rows in flight per lane group (linear_n1.hip)
    (262200, 1, 64), (270001, 1, 20), (262144, 1, 3),
]


@pytest.mark.parametrize("m,n,k", LINEAR_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_forward(ops, m, n, k, act):
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k + act)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), act).cpu()
    ref = x.double() @ w.double().T + b.double()
    ref = [ref, torch.relu(ref), torch.sigmoid(ref)][act]
    torch.testing.assert_close(y, ref.float(), rtol=1e-5, atol=4e-6)  # fp32 sum over k terms in tile order: a few ulp of the largest partial sum


@pytest.mark.parametrize("m,n,k", LINEAR_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_backward(ops, m, n, k, act):
    g = torch.Generator().manual_seed(m * 11 + n * 5 + k + act)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    z = xd @ wd.T + bd
    yd = [z, torch.relu(z), torch.sigmoid(z)][act]
    yd.backward(gy.double())
    dx, dw, dy, dgy = x.to(DEV), w.to(DEV), yd.detach().float().to(DEV), gy.to(DEV)
    gx = torch.full((m, k), float("nan"), device=DEV)
    gw, gb = torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)
    ops.linear_bwd(dx, dw, dy, dgy, act, gx, gw, gb)
    scale = max(1.0, m ** 0.5)
    torch.testing.assert_close(gx.cpu(), xd.grad.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gw.cpu(), wd.grad.float(), rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(gb.cpu(), bd.grad.float(), rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("m,n,k,act", [(4096, 256, 256, 1), (4500, 512, 416, 0), (4100, 256, 272, 2), (70000, 256, 512, 1)])
def test_linear_forward_wide_macro_tile(m, n, k, act):
    """csrc/gemm_wide.hip (256 x 256 macro tile; opt-in: CTR_GEMM_WIDE=1) in a process of its own, against float64"""
    import subprocess
    import sys
    code = f"""
import torch
from deeplearningrecommendationsystem_amd import ops
g = torch.Generator().manual_seed({m + n + k})
x, w, b = torch.randn({m}, {k}, generator=g), torch.randn({n}, {k}, generator=g) / {k} ** 0.5, torch.randn({n}, generator=g)
y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), {act}).cpu()
ref = x.double() @ w.double().T + b.double()
ref = [ref, torch.relu(ref), torch.sigmoid(ref)][{act}]
torch.testing.assert_close(y, ref.float(), rtol=1e-5, atol=4e-6)
print("ok")
"""
    env = dict(os.environ, CTR_GEMM_WIDE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_linear_strided_views_residual_and_accumulate(ops):
    g = torch.Generator().manual_seed(9)
    m, n, k = 200, 24, 40
    big = torch.randn(m, 100, generator=g)
    x = big[:, 10:10 + k]                       # column slice: ld = 100
    w, b = torch.randn(n, k, generator=g), torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    dbig = big.to(DEV)
    outbuf = torch.zeros(m, 64, device=DEV)
    ops.linear_fwd(dbig[:, 10:10 + k], w.to(DEV), b.to(DEV), 1, out=outbuf[:, 8:8 + n], residual=res.to(DEV))
    ref = torch.relu(x.double() @ w.double().T + b.double() + res.double()).float()
    torch.testing.assert_close(outbuf[:, 8:8 + n].cpu(), ref, rtol=1e-5, atol=2e-6)
    assert torch.equal(outbuf[:, :8].cpu(), torch.zeros(m, 8))
    assert torch.equal(outbuf[:, 8 + n:].cpu(), torch.zeros(m, 64 - 8 - n))
    # accumulate_gx adds to what is there
    gy = torch.randn(m, n, generator=g)
    gx = torch.ones(m, k, device=DEV)
    ops.linear_bwd(dbig[:, 10:10 + k], w.to(DEV), None, gy.to(DEV), 0, gx, None, None, accumulate_gx=True)
    torch.testing.assert_close(gx.cpu(), (1.0 + gy.double() @ w.double()).float(), rtol=1e-5, atol=1e-5)


def test_linear_direct_to_lds_odd_leading_dimensions(ops):
    """operands cut out of wider buffers with ODD row strides (rows only 4-byte aligned), residual,
    accumulate_gx, at sizes the direct-to-LDS kernels take (k, n >= 16; m >= 4096 for dW)"""
    g = torch.Generator().manual_seed(21)
    m, n, k = 4300, 161, 163
    xb, wb = torch.randn(m, k + 6, generator=g), torch.randn(n, k + 2, generator=g) / k ** 0.5
    x, w, b = xb[:, 3:3 + k], wb[:, 1:1 + k], torch.randn(n, generator=g)
    resb, gyb = torch.randn(m, n + 4, generator=g), torch.randn(m, n + 2, generator=g)
    res, gy = resb[:, 2:2 + n], gyb[:, 1:1 + n]
    dxb, dwb, dresb, dgyb = xb.to(DEV), wb.to(DEV), resb.to(DEV), gyb.to(DEV)
    dx, dw, dres, dgy = dxb[:, 3:3 + k], dwb[:, 1:1 + k], dresb[:, 2:2 + n], dgyb[:, 1:1 + n]
    ybuf = torch.full((m, n + 3), 7.0, device=DEV)
    ops.linear_fwd(dx, dw, b.to(DEV), 1, out=ybuf[:, 1:1 + n], residual=dres)
    z = x.double() @ w.double().T + b.double() + res.double()
    torch.testing.assert_close(ybuf[:, 1:1 + n].cpu(), torch.relu(z).float(), rtol=1e-5, atol=4e-6)
    assert torch.equal(ybuf[:, :1].cpu(), torch.full((m, 1), 7.0)) and torch.equal(ybuf[:, 1 + n:].cpu(), torch.full((m, 2), 7.0))
    # backward through the relu, gX accumulated into a strided view
    yd = torch.relu(z)
    gz = gy.double() * (yd > 0)
    gxbuf = torch.ones(m, k + 5, device=DEV)
    gw, gb = torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)
    ops.linear_bwd(dx, dw, ybuf[:, 1:1 + n], dgy, 1, gxbuf[:, 2:2 + k], gw, gb, accumulate_gx=True)
    torch.testing.assert_close(gxbuf[:, 2:2 + k].cpu(), (1.0 + gz @ w.double()).float(), rtol=1e-5, atol=1e-5)
    assert torch.equal(gxbuf[:, :2].cpu(), torch.ones(m, 2)) and torch.equal(gxbuf[:, 2 + k:].cpu(), torch.ones(m, 3))
    scale = m ** 0.5
    torch.testing.assert_close(gw.cpu(), (gz.T @ x.double()).float(), rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(gb.cpu(), gz.sum(0).float(), rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("dims,p", [([128, 64, 32, 16, 8], 64), ([16, 8], 8), ([24, 16, 8], 0), ([32, 16], 40)])
def test_mlp_forward_with_fused_head(ops, dims, p):
    """ctr_mlp_head_fwd: the single-unit layer on [x_extra | last activations] from the stack's epilogue
    (pinned NeuralCF tower, generic stacks, no extra columns, ragged batch)"""
    g = torch.Generator().manual_seed(sum(dims) + p)
    m = 1000 + p
    buf = torch.randn(m, dims[0] + p + dims[-1] + 4, generator=g)  # [x0 | x_extra | room for y_last | pad]
    layers = [ops.Layer((torch.randn(n, k, generator=g) / k ** 0.5).to(DEV), (torch.randn(n, generator=g) * 0.1).to(DEV), 1)
              for k, n in zip(dims[:-1], dims[1:])]
    w = torch.randn(1, p + dims[-1], generator=g)
    c = torch.randn(1, generator=g)
    dbuf = buf.to(DEV)
    x0, xe = dbuf[:, :dims[0]], (dbuf[:, dims[0]:dims[0] + p] if p else None)
    head = ops.Head(xe, w.to(DEV), c.to(DEV), 2)
    acts = ops.mlp_fwd(x0, layers, last_out=dbuf[:, dims[0] + p:dims[0] + p + dims[-1]], head=head)
    h = buf[:, :dims[0]].double()
    for layer in layers:
        h = torch.relu(h @ layer.weight.cpu().double().T + layer.bias.cpu().double())
    torch.testing.assert_close(acts[-1].cpu(), h.float(), rtol=1e-5, atol=2e-6)
    operand = torch.cat([buf[:, dims[0]:dims[0] + p].double(), h], dim=1)
    ref = torch.sigmoid(operand @ w.double().T + c.double())
    assert head.out.shape == (m, 1)
    torch.testing.assert_close(head.out.cpu(), ref.float(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("m", [4096, 1000, 37, 1029, 20011, 70001])
def test_mlp_backward_with_fused_head_matches_autograd(ops, m):
    """ctr_mlp_head_bwd (pinned NeuralCF tower, 64 extra columns): gradients of every layer, of the stack
    input, of the extra columns and of the head's weights / bias against fp64 autograd.  1029: a ragged last group of
    sixteen and waves without a group; 20011 / 70001: several groups per wave with a ragged last one (ncf16_bwd_kernel
    clamps the rows past m and gives them a zero gradient)"""
    dims, p = [128, 64, 32, 16, 8], 64
    g = torch.Generator().manual_seed(m)
    x0 = torch.randn(m, dims[0], generator=g)
    xe = torch.randn(m, p, generator=g)
    ws = [torch.randn(n, k, generator=g) / k ** 0.5 for k, n in zip(dims[:-1], dims[1:])]
    bs = [torch.randn(n, generator=g) * 0.1 for n in dims[1:]]
    wh, ch = torch.randn(1, p + dims[-1], generator=g) * 0.3, torch.randn(1, generator=g)
    gprob = torch.randn(m, 1, generator=g)
    # fp64 reference
    leaves = [t.double().requires_grad_(True) for t in [x0, xe, wh, ch] + ws + bs]
    rx0, rxe, rwh, rch = leaves[:4]
    rws, rbs = leaves[4:4 + len(ws)], leaves[4 + len(ws):]
    hcur = rx0
    borderline = torch.zeros(m, dtype=torch.bool)
    for w_, b_ in zip(rws, rbs):
        z = hcur @ w_.T + b_
        borderline |= (z.detach().abs() < 1e-5).any(dim=1)
        hcur = torch.relu(z)
    # A unit within fp32 rounding of zero may sit on the other side of the ReLU on the device (8.4 M activations at
    # m = 70001: it happens).  Both sides are right for their own forward; such samples get a zero upstream gradient so
    # that the comparison stays exact for everything else.
    gprob[borderline] = 0.0
    prob_ref = torch.sigmoid(torch.cat([rxe, hcur], dim=1) @ rwh.T + rch)
    prob_ref.backward(gprob.double())
    # device: forward with the fused head, then the fused backward
    buf = torch.zeros(m, dims[0] + p + dims[-1], device=DEV)
    buf[:, :dims[0]] = x0.to(DEV)
    buf[:, dims[0]:dims[0] + p] = xe.to(DEV)
    layers = [ops.Layer(w_.to(DEV), b_.to(DEV), 1) for w_, b_ in zip(ws, bs)]
    dwh, dch = wh.to(DEV), ch.to(DEV)
    head = ops.Head(buf[:, dims[0]:dims[0] + p], dwh, dch, 2)
    acts = ops.mlp_fwd(buf[:, :dims[0]], layers, last_out=buf[:, dims[0] + p:], head=head)
    torch.testing.assert_close(head.out.cpu(), prob_ref.detach().float(), rtol=1e-5, atol=2e-6)
    params = [l.weight for l in layers] + [l.bias for l in layers]
    zeros = ops.zero_grads(params + [dwh, dch.new_empty(4)])
    gwh, gch = zeros[id(dwh)], list(zeros.values())[-1][:1]
    gbuf = torch.full_like(buf, float("nan"))
    grads = ops.mlp_head_bwd(acts, layers, head, head.out, gprob.to(DEV), gbuf[:, dims[0]:dims[0] + p], gwh, gch,
                             gbuf[:, :dims[0]], zeros)
    if m < 1024:  # ops fuses stacks from 1024 rows up: below that the caller runs the head and the stack separately
        assert grads is None
        return
    assert grads is not None, "the pinned tower must take the fused path"
    scale = max(1.0, m ** 0.5)
    torch.testing.assert_close(gbuf[:, :dims[0]].cpu(), rx0.grad.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gbuf[:, dims[0]:dims[0] + p].cpu(), rxe.grad.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gwh.cpu(), rwh.grad.float(), rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(gch.cpu(), rch.grad.float(), rtol=1e-5, atol=2e-6 * scale)
    for (gw, gb), rw, rb in zip(grads, rws, rbs):
        torch.testing.assert_close(gw.cpu(), rw.grad.float(), rtol=1e-5, atol=2e-6 * scale)
        torch.testing.assert_close(gb.cpu(), rb.grad.float(), rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("m", [4096, 1037, 70001])
def test_gather_inside_the_tower_forward_matches_the_two_launches(ops, m):
    """ctr_embed_mlp_head_fwd (NeuralCF at BASELINE configs[1]: ids -> four table rows -> tower -> folded head in one
    launch) against ctr_embed_fwd + ctr_mlp_head_fwd: the gathered columns bit for bit, activations and probabilities
    within the forward tolerance; one id outside its table raises the flag and reads row 0, as the gather kernel does"""
    from deeplearningrecommendationsystem_amd._lib import FIELD_ID_I64, FIELD_PROD_I64
    g = torch.Generator().manual_seed(m)
    nu, ni, half, mf = 943, 1682, 64, 64
    dims = [128, 64, 32, 16, 8]
    tabs = [torch.randn(v, 64, generator=g).to(DEV) for v in (nu, ni, nu, ni)]   # mlp_u, mlp_i, gmf_u, gmf_i
    u = torch.randint(0, nu, (m,), generator=g)
    i = torch.randint(0, ni, (m,), generator=g)
    u[5] = nu + 3                                     # outside the table
    u, i = u.to(DEV), i.to(DEV)
    layers = [ops.Layer((torch.randn(n, k, generator=g) / k ** 0.5).to(DEV), (torch.randn(n, generator=g) * 0.1).to(DEV), 1)
              for k, n in zip(dims[:-1], dims[1:])]
    w = (torch.randn(1, mf + dims[-1], generator=g) * 0.3).to(DEV)
    c = torch.randn(1, generator=g).to(DEV)
    specs = [ops.FieldSpec(FIELD_ID_I64, half, 0, table=tabs[0], idx=u),
             ops.FieldSpec(FIELD_ID_I64, half, half, table=tabs[1], idx=i),
             ops.FieldSpec(FIELD_PROD_I64, mf, 2 * half, table=tabs[2], idx=u, table2=tabs[3], idx2=i)]
    width = 2 * half + mf + dims[-1]

    def run(fused):
        buf = torch.full((m, width), float("nan"), device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        head = ops.Head(buf[:, 128:192], w, c, 2)
        if fused:
            acts = ops.embed_mlp_head_fwd(specs, m, buf, 128, layers, head, buf[:, 192:], flag)
            assert acts is not None, "the BASELINE pattern must take the fused launch"
        else:
            ops.embed_fwd(specs, None, m, buf, flag)
            acts = ops.mlp_fwd(buf[:, :128], layers, last_out=buf[:, 192:], head=head)
        return buf, acts, head.out, flag

    bf, af, pf, ff = run(True)
    bs, as_, ps, fs = run(False)
    assert int(ff.item()) == 1 and int(fs.item()) == 1
    assert torch.equal(bf[:, :192].cpu(), bs[:, :192].cpu())
    for a, b in zip(af[1:], as_[1:]):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(pf.cpu(), ps.cpu(), rtol=1e-5, atol=2e-6)
    assert not torch.isnan(bf).any()

    # fold=...: ctr_fold_head_fwd's map inside the same launch -- the head's weights are outputs then
    u_full = (torch.randn(1, 128, generator=g) * 0.3).to(DEV)
    pw, pb, b2 = (torch.randn(64, 8, generator=g) / 3).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV), c
    wref, cref = ops.fold_head_fwd(u_full, 64, pw, pb, b2)
    buf = torch.full((m, width), float("nan"), device=DEV)
    wout, cout = torch.full_like(wref, float("nan")), torch.full_like(cref, float("nan"))
    head = ops.Head(buf[:, 128:192], wout, cout, 2)
    assert ops.embed_mlp_head_fwd(specs, m, buf, 128, layers, head, buf[:, 192:], None, fold=(u_full, pw, pb, b2)) is not None
    torch.testing.assert_close(wout.cpu(), wref.cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cout.cpu(), cref.cpu(), rtol=1e-5, atol=1e-6)
    buf2 = torch.full((m, width), float("nan"), device=DEV)
    head2 = ops.Head(buf2[:, 128:192], wref, cref, 2)
    ops.embed_mlp_head_fwd(specs, m, buf2, 128, layers, head2, buf2[:, 192:], None)
    torch.testing.assert_close(head.out.cpu(), head2.out.cpu(), rtol=1e-5, atol=2e-6)

    # write_x=False: the tower input is never written, and ctr_embed_mlp_head_bwd gathers it again -- same gradients as
    # the backward that reads the written columns
    buf = torch.full((m, width), float("nan"), device=DEV)
    head = ops.Head(buf[:, 128:192], w, c, 2)
    acts = ops.embed_mlp_head_fwd(specs, m, buf, 128, layers, head, buf[:, 192:], None, write_x=False)
    assert acts is not None and torch.isnan(buf[:, :128]).all()
    assert torch.equal(buf[:, 128:192].cpu(), bf[:, 128:192].cpu()) and torch.equal(head.out.cpu(), pf.cpu())
    gprob = torch.randn(m, 1, generator=g).to(DEV)
    wz = w.new_zeros(w.shape)

    def backward(acts_, head_, prob_, gather):
        params = [l.weight for l in layers] + [l.bias for l in layers]
        zeros = ops.zero_grads(params + [wz, c.new_empty(4)])
        gwh, gch = zeros[id(wz)], list(zeros.values())[-1][:1]
        gbuf = torch.full((m, 192), float("nan"), device=DEV)
        grads = ops.mlp_head_bwd(acts_, layers, head_, prob_, gprob, gbuf[:, 128:192], gwh, gch, gbuf[:, :128], zeros,
                                 gather_specs=specs if gather else None)
        assert grads is not None
        return [gbuf, gwh, gch] + [t for pair in grads for t in pair]

    ref = backward(af, ops.Head(bf[:, 128:192], w, c, 2), pf, False)
    got = backward(acts, head, head.out, True)
    scale = m ** 0.5
    for a, b in zip(got, ref):
        assert not torch.isnan(a).any()
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-5, atol=2e-6 * scale)

    # fold_grad=...: ctr_fold_head_bwd inside the reduction launch of the same call, against the separate launch
    def fold_grads(fused):
        params = [l.weight for l in layers] + [l.bias for l in layers]
        zeros = ops.zero_grads(params + [wz, c.new_empty(4), u_full, pw, pb, b2], lazy=fused)
        flat = zeros.pop("flat", None)     # fused: NOT cleared here -- the call's first launch clears it (zero=)
        if flat is not None:
            flat.fill_(float("nan"))
        gwh, gch = zeros[id(wz)], list(zeros.values())[2 * len(layers) + 1][:1]   # (the c.new_empty(4) slot)
        gu, gpw, gpb, gb2 = zeros[id(u_full)], zeros[id(pw)], zeros[id(pb)], zeros[id(b2)]
        gbuf = torch.empty((m, 192), device=DEV)
        fg = (u_full, pw, pb, gu, gpw, gpb, gb2) if fused else None
        assert ops.mlp_head_bwd(acts, layers, head, head.out, gprob, gbuf[:, 128:192], gwh, gch, gbuf[:, :128], zeros,
                                gather_specs=specs, fold_grad=fg, zero=flat) is not None
        if not fused:
            ops.fold_head_bwd(u_full, 64, pw, pb, gwh, gch, gu, gpw, gpb, gb2)
        return [gwh, gch, gu, gpw, gpb, gb2]

    for a, b in zip(fold_grads(True), fold_grads(False)):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-5, atol=2e-6 * scale)


def test_tiled_tower_kernels_behind_the_switch():
    """CTR_MLP_16=0 selects the tiled fused-MLP kernels for the pinned NeuralCF tower (the A/B partner of the
    operand-layout kernels, read once per process): the same parity cases in one child process"""
    import subprocess
    import sys
    env = dict(os.environ, CTR_MLP_16="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-x", "-m", "gpu",
                          "-k", "fused_head and not behind_the_switch", "-p", "no:cacheprovider"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_fold_head_matches_the_unfolded_pair(ops):
    """(h W^T + b).u + b2 == h.v + c and its chain rule (ctr_fold_head_fwd/bwd) against autograd on the
    unfolded expression"""
    g = torch.Generator().manual_seed(5)
    for p_, n, k in [(64, 64, 8), (0, 5, 3), (7, 130, 33)]:
        u = torch.randn(1, p_ + n, generator=g).double().requires_grad_(True)
        w = torch.randn(n, k, generator=g).double().requires_grad_(True)
        b = torch.randn(n, generator=g).double().requires_grad_(True)
        b2 = torch.randn(1, generator=g).double().requires_grad_(True)
        wfold_ref = torch.cat([u[:, :p_], u[:, p_:] @ w], dim=1)
        c_ref = (b * u[0, p_:]).sum() + b2[0]
        du, dw, db, db2 = (t.detach().float().to(DEV) for t in (u, w, b, b2))
        wfold, cfold = ops.fold_head_fwd(du, p_, dw, db, db2)
        torch.testing.assert_close(wfold.cpu(), wfold_ref.detach().float(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(cfold.cpu(), c_ref.detach().float().reshape(1), rtol=1e-5, atol=1e-6)
        gwf, gc = torch.randn(1, p_ + k, generator=g), torch.randn(1, generator=g)
        ((wfold_ref * gwf.double()).sum() + c_ref * gc.double()[0]).backward()
        gu, gw, gb, gb2 = (torch.ones_like(t) for t in (du, dw, db, db2))  # accumulated into
        ops.fold_head_bwd(du, p_, dw, db, gwf.to(DEV), gc.to(DEV), gu, gw, gb, gb2)
        for got, ref in ((gu, u), (gw, w), (gb, b), (gb2, b2)):
            torch.testing.assert_close(got.cpu(), 1.0 + ref.grad.float(), rtol=1e-5, atol=1e-5)


def test_mf_fused_kernels(ops):
    g = torch.Generator().manual_seed(3)
    for dim, batch in [(64, 1024), (12, 37), (5, 3)]:
        ut, it = torch.randn(30, dim, generator=g), torch.randn(40, dim, generator=g)
        u, i = torch.randint(0, 30, (batch,), generator=g), torch.randint(0, 40, (batch,), generator=g)
        prob = ops.mf_fwd(ut.to(DEV), it.to(DEV), u.to(DEV), i.to(DEV))
        ref = orc.mf_forward({"user_embeddings.weight": ut, "item_embeddings.weight": it}, u, i)
        torch.testing.assert_close(prob.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("nvec,dim,batch", [(6, 16, 1000), (6, 8, 37), (3, 5, 65), (26, 16, 300), (6, 256, 70)])
def test_allpairs_inner_products(ops, nvec, dim, batch):
    g = torch.Generator().manual_seed(nvec * 100 + dim + batch)
    emb = torch.randn(batch, nvec * dim, generator=g)
    vecs = [emb[:, f * dim:(f + 1) * dim] for f in range(nvec)]
    ref = orc.pnn_inner_products(vecs)
    out = ops.allpairs_fwd(emb.to(DEV), nvec, dim).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    gp = torch.randn(batch, ref.shape[1], generator=g)
    leaf = emb.double().requires_grad_(True)
    orc.pnn_inner_products([leaf[:, f * dim:(f + 1) * dim] for f in range(nvec)]).backward(gp.double())
    gemb = torch.ones(batch, nvec * dim, device=DEV)
    ops.allpairs_bwd(emb.to(DEV), nvec, dim, gp.to(DEV), gemb, accumulate=True)
    torch.testing.assert_close(gemb.cpu(), (1.0 + leaf.grad).float(), rtol=1e-5, atol=1e-4)
    ops.allpairs_bwd(emb.to(DEV), nvec, dim, gp.to(DEV), gemb, accumulate=False)
    torch.testing.assert_close(gemb.cpu(), leaf.grad.float(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dim,batch", [(16, 1000), (8, 37), (5, 3), (128, 200)])
def test_fm_wide_forward_backward(ops, dim, batch):
    from deeplearningrecommendationsystem_amd import synth
    g = synth.generator(dim + batch)
    x = synth.feature_batch(batch, 30, 40, g)
    emb = torch.randn(batch, 6 * dim, generator=g)
    user1, item1 = torch.randn(30, 1, generator=g), torch.randn(40, 1, generator=g)
    w, b = torch.randn(1, 43, generator=g), torch.randn(1, generator=g)

    def ref(embt, u1, i1, wt, bt, xx):
        vecs = [embt[:, f * dim:(f + 1) * dim] for f in range(6)]
        return (u1[xx[:, 0].long()] + i1[xx[:, 1].long()] + xx[:, 2:] @ wt.T + bt
                + orc.fm_second_order(vecs).unsqueeze(1))

    out = torch.zeros(batch, 2, device=DEV)
    ops.fm_wide_fwd(emb.to(DEV), 6, dim, x.to(DEV), user1.to(DEV), item1.to(DEV), w.to(DEV), b.to(DEV), out[:, 0:1])
    torch.testing.assert_close(out[:, 0:1].cpu(), ref(emb, user1, item1, w, b, x), rtol=1e-5, atol=1e-5)
    assert torch.equal(out[:, 1].cpu(), torch.zeros(batch))

    leaves = [t.double().requires_grad_(True) for t in (emb, user1, item1, w, b)]
    gout = torch.randn(batch, 1, generator=g)
    ref(*leaves, x.double()).backward(gout.double())
    gs = [torch.zeros_like(t).to(DEV) for t in (user1, item1, w, b)]
    gemb = torch.zeros(batch, 6 * dim, device=DEV)
    ops.fm_wide_bwd(emb.to(DEV), 6, dim, x.to(DEV), user1.to(DEV), item1.to(DEV), w.to(DEV), b.to(DEV),
                    gout.to(DEV), *gs, gemb, accumulate=False)
    atol = 1e-5 + 3e-6 * batch ** 0.5
    torch.testing.assert_close(gemb.cpu(), leaves[0].grad.float(), rtol=1e-5, atol=1e-4)
    for got, leaf in zip(gs, leaves[1:]):
        torch.testing.assert_close(got.cpu(), leaf.grad.float(), rtol=1e-5, atol=atol)


def test_act_bwd(ops):
    g = torch.Generator().manual_seed(1)
    y, gy = torch.randn(77, 13, generator=g), torch.randn(77, 13, generator=g)
    out = torch.ones(77, 16, device=DEV)
    ops.act_bwd(y.to(DEV), gy.to(DEV), 1, out[:, :13], accumulate=True)
    torch.testing.assert_close(out[:, :13].cpu(), 1.0 + gy * (y > 0).float())
    assert torch.equal(out[:, 13:].cpu(), torch.ones(77, 3))


@pytest.mark.parametrize("dim,length,batch", [(64, 100, 300), (8, 10, 64), (4, 1, 37), (16, 130, 5), (6, 7, 9), (8, 600, 3)])
def test_din_attention_pieces(ops, dim, length, batch):
    from deeplearningrecommendationsystem_amd import synth
    g = synth.generator(dim * 7 + length)
    vocab = 50
    table = torch.randn(vocab, dim, generator=g)
    hist, target = synth.hist_batch(batch, length, vocab, g)
    dt, dh, dtg = table.to(DEV), hist.to(DEV), target.to(DEV)
    c = torch.empty(batch * length, 3 * dim, device=DEV)
    fcin = torch.zeros(batch, 2 * dim, device=DEV)
    ops.din_concat_fwd(dt, dh, dtg, c, fcin[:, dim:])
    h, t = orc.gather_rows(table, hist), orc.gather_rows(table, target)
    te = t.unsqueeze(1).expand_as(h)
    assert torch.equal(c.cpu().view(batch, length, 3 * dim), torch.cat([h, h - te, te], dim=-1))
    assert torch.equal(fcin[:, dim:].cpu(), t)

    score = torch.randn(batch, length, generator=g)
    for summed in (True, False):
        attn = torch.empty(batch, length, device=DEV)
        out = torch.zeros((batch, 2 * dim) if summed else (batch * length, dim), device=DEV)
        ops.din_pool_fwd(score.to(DEV), c, batch, length, dim, attn, out[:, :dim], summed)
        a = torch.softmax(score.double(), dim=-1)
        torch.testing.assert_close(attn.cpu(), a.float(), rtol=1e-5, atol=1e-7)
        ref = h.double() * a.unsqueeze(-1)
        ref = ref.sum(1) if summed else ref.reshape(batch * length, dim)
        torch.testing.assert_close(out[:, :dim].cpu(), ref.float(), rtol=1e-5, atol=1e-6)
        # backward of pooling + softmax w.r.t. the scores
        gout = torch.randn(out[:, :dim].shape, generator=g)
        sl = score.double().requires_grad_(True)
        o = h.double() * torch.softmax(sl, dim=-1).unsqueeze(-1)
        o = o.sum(1) if summed else o.reshape(batch * length, dim)
        o.backward(gout.double())
        gscore = torch.empty(batch * length, 1, device=DEV)
        ops.din_pool_bwd(attn, c, batch, length, dim, gout.to(DEV), summed, gscore)
        torch.testing.assert_close(gscore.cpu().view(batch, length), sl.grad.float(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dim,length,batch", [(16, 20, 100), (64, 7, 33), (5, 3, 4)])
def test_din_pair_operand_and_folded_first_layer(ops, dim, length, batch):
    # [h, t] operand (CTR_DIN_PAIR) with the first attention layer's weight columns folded:
    #   W @ [h, h-t, t] == [Wa+Wb, Wc-Wb] @ [h, t]      (model/din.py:39-42)
    # forward copies are bit-exact; the scatter of the pair gradient equals autograd of the
    # reference expression with the unfolded weight
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model.din import fold_attention_weight, unfold_attention_grad
    g = synth.generator(dim * 11 + length)
    vocab = 40
    table = torch.randn(vocab, dim, generator=g)
    hist, target = synth.hist_batch(batch, length, vocab, g)
    dt, dh, dtg = table.to(DEV), hist.to(DEV), target.to(DEV)
    c = torch.empty(batch * length, 2 * dim, device=DEV)
    tv = torch.zeros(batch, dim, device=DEV)
    ops.din_concat_fwd(dt, dh, dtg, c, tv, pair=True)
    h, t = orc.gather_rows(table, hist), orc.gather_rows(table, target)
    te = t.unsqueeze(1).expand_as(h)
    assert torch.equal(c.cpu().view(batch, length, 2 * dim), torch.cat([h, te], dim=-1))
    assert torch.equal(tv.cpu(), t)

    n = 12
    w = torch.randn(n, 3 * dim, generator=g)
    wf = fold_attention_weight(w.to(DEV), dim).cpu()
    z_ref = torch.cat([h, h - te, te], -1).double() @ w.double().T
    z_pair = torch.cat([h, te], -1).double() @ wf.double().T
    torch.testing.assert_close(z_pair, z_ref, rtol=1e-5, atol=1e-5)

    # backward: gradient of sum(gz * z) w.r.t. the table through the pair operand ...
    gz = torch.randn(batch, length, n, generator=g)
    gc = (gz.double() @ wf.double()).float().reshape(batch * length, 2 * dim)     # d/d[h, t]
    attn = torch.rand(batch, length, generator=g)
    gout = torch.randn(batch, dim, generator=g)
    gtable = torch.zeros(vocab, dim, device=DEV)
    ops.din_concat_bwd(dh, dtg, vocab, dim, gc.to(DEV), attn.to(DEV), gout.to(DEV), True, None, gtable, pair=True)
    # ... equals autograd of the reference's triple expression (+ the pooled-output path a_l * gout)
    leaf = table.double().requires_grad_(True)
    hh, tt = leaf[hist], leaf[target].unsqueeze(1).expand(batch, length, dim)
    z = torch.cat([hh, hh - tt, tt], -1) @ w.double().T
    ((z * gz.double()).sum() + (hh * attn.double().unsqueeze(-1) * gout.double().unsqueeze(1)).sum()).backward()
    torch.testing.assert_close(gtable.cpu(), leaf.grad.float(), rtol=1e-4, atol=1e-4)
    # weight gradient chain rule of the fold
    gwf = torch.randn(n, 2 * dim, generator=g)
    gw = torch.zeros(n, 3 * dim, device=DEV)
    unfold_attention_grad(gwf.to(DEV), gw, dim)
    wl = w.double().requires_grad_(True)
    folded = torch.cat([wl[:, :dim] + wl[:, dim:2 * dim], wl[:, 2 * dim:] - wl[:, dim:2 * dim]], 1)
    (folded * gwf.double()).sum().backward()
    torch.testing.assert_close(gw.cpu(), wl.grad.float(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("batch,nvec,dim", [(1000, 6, 8), (37, 6, 128), (5, 2, 3), (4096, 12, 16)])
def test_bi_interaction_pooling(ops, batch, nvec, dim):
    # model/nfm.py:56-61: sum of the element products of every pair of vectors, and its backward
    g = torch.Generator().manual_seed(batch + nvec + dim)
    emb = torch.randn(batch, nvec * dim, generator=g)
    out = torch.empty(batch, dim, device=DEV)
    ops.biinteract_fwd(emb.to(DEV), nvec, dim, out)
    leaf = emb.double().requires_grad_(True)
    v = leaf.view(batch, nvec, dim)
    ref = sum(v[:, i] * v[:, j] for i in range(nvec) for j in range(i + 1, nvec))
    torch.testing.assert_close(out.cpu(), ref.float(), rtol=1e-5, atol=1e-5)
    gout = torch.randn(batch, dim, generator=g)
    ref.backward(gout.double())
    gemb = torch.full((batch, nvec * dim), float("nan"), device=DEV)
    ops.biinteract_bwd(emb.to(DEV), nvec, dim, gout.to(DEV), gemb, accumulate=False)
    torch.testing.assert_close(gemb.cpu(), leaf.grad.float(), rtol=1e-5, atol=1e-5)
    ops.biinteract_bwd(emb.to(DEV), nvec, dim, gout.to(DEV), gemb, accumulate=True)
    torch.testing.assert_close(gemb.cpu(), 2 * leaf.grad.float(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("batch,nvec,dim", [(500, 6, 8), (33, 6, 128), (7, 3, 5), (2048, 4, 16)])
def test_pair_products(ops, batch, nvec, dim):
    # model/afm.py:56-65: the stacked pair products, and their backward incl. the attention-weighted sum
    g = torch.Generator().manual_seed(batch * 3 + nvec + dim)
    npairs = nvec * (nvec - 1) // 2
    emb = torch.randn(batch, nvec * dim, generator=g)
    out = torch.empty(batch * npairs, dim, device=DEV)
    ops.pairprod_fwd(emb.to(DEV), nvec, dim, out)
    leaf = emb.double().requires_grad_(True)
    v = leaf.view(batch, nvec, dim)
    ref = torch.stack([v[:, i] * v[:, j] for i in range(nvec) for j in range(i + 1, nvec)], dim=1)
    assert torch.equal(out.cpu().view(batch, npairs, dim), ref.float())     # single products: exact
    gp = torch.randn(batch, npairs, dim, generator=g)
    attn = torch.rand(batch, npairs, generator=g)
    gpool = torch.randn(batch, dim, generator=g)
    ((ref * gp.double()).sum() + ((ref * attn.double().unsqueeze(-1)).sum(1) * gpool.double()).sum()).backward()
    gemb = torch.empty(batch, nvec * dim, device=DEV)
    ops.pairprod_bwd(emb.to(DEV), nvec, dim, gp.view(batch * npairs, dim).to(DEV), attn.to(DEV), gpool.to(DEV), gemb,
                     accumulate=False)
    torch.testing.assert_close(gemb.cpu(), leaf.grad.float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,d", [(1000, 41), (257, 64), (3, 5), (4096, 641)])
def test_cross_layer_combine(ops, m, d):
    # model/deepcross.py:14-17: x_{l+1} = x0 * u + b + x_l and its backward pieces
    g = torch.Generator().manual_seed(m + d)
    x0, u, xl, gy = (torch.randn(m, d, generator=g) for _ in range(4))
    b = torch.randn(d, generator=g)
    pad = (d + 3) // 4 * 4
    dev = lambda t: torch.zeros(m, pad, device=DEV)[:, :d].copy_(t)  # noqa: E731  (16-byte aligned rows)
    out = torch.empty(m, pad, device=DEV)[:, :d]
    ops.cross_fwd(dev(x0), dev(u), dev(xl), b.to(DEV), out)
    torch.testing.assert_close(out.cpu(), x0 * u + b + xl, rtol=1e-6, atol=1e-6)
    # unaligned operands take the scalar path
    out2 = torch.empty(m, d, device=DEV)
    ops.cross_fwd(x0.to(DEV), u.to(DEV), xl.to(DEV), b.to(DEV), out2)
    torch.testing.assert_close(out2.cpu(), x0 * u + b + xl, rtol=1e-6, atol=1e-6)
    gu = torch.empty(m, d, device=DEV)
    gx0_start = torch.randn(m, d, generator=g)
    gx0 = gx0_start.to(DEV)
    gb = torch.zeros(d, device=DEV)
    ops.cross_bwd(x0.to(DEV), u.to(DEV), gy.to(DEV), gu, gx0, gb)
    torch.testing.assert_close(gu.cpu(), gy * x0, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gx0.cpu(), gx0_start + gy * u, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gb.cpu(), gy.double().sum(0).float(), rtol=1e-5, atol=1e-5 * m ** 0.5)


def test_embed_backward_padding_row_and_large_table(ops):
    # a 100k-row table (atomic scatter path) where 30 % of the ids are 0, the padding id of the
    # behaviour sequences: row 0 is pre-reduced per workgroup instead of 20k same-address atomics
    L = _lib()
    g = torch.Generator().manual_seed(77)
    vocab, e, batch = 100_000, 16, 60_000
    ids = torch.randint(0, vocab, (batch,), generator=g)
    ids[torch.rand(batch, generator=g) < 0.3] = 0
    table = torch.randn(vocab, e, generator=g)
    gout = torch.randn(batch, e, generator=g)
    dtab, dids = table.to(DEV), ids.to(DEV)
    grads = {id(dtab): torch.zeros_like(dtab)}
    ops.embed_bwd([ops.FieldSpec(L.FIELD_ID_I64, e, 0, table=dtab, idx=dids)], None, batch, gout.to(DEV), grads)
    ref = torch.zeros(vocab, e, dtype=torch.float64)
    ref.index_add_(0, ids, gout.double())
    torch.testing.assert_close(grads[id(dtab)].cpu(), ref.float(), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(grads[id(dtab)][1:].cpu(), ref[1:].float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dim,length,batch", [(16, 20, 100), (4, 1, 5), (8, 33, 70), (64, 5, 9), (5, 3, 3),
                                              (16, 50, 1024), (16, 7, 6), (16, 1, 4)])
def test_gru_recurrence(ops, dim, length, batch):
    g = torch.Generator().manual_seed(dim + length + batch)
    gru = torch.nn.GRU(dim, dim, batch_first=True)
    x = torch.randn(batch, length, dim, generator=g)
    w_ih, w_hh, b_ih, b_hh = (p.detach() for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0))
    leaves = [t.double().requires_grad_(True) for t in (w_ih, w_hh, b_ih, b_hh, x)]
    last = orc.gru_last_hidden(*leaves)
    glast = torch.randn(batch, dim, generator=g)
    last.backward(glast.double())

    dx = x.to(DEV).view(batch * length, dim)
    gi = ops.linear_fwd(dx, w_ih.to(DEV), b_ih.to(DEV))
    hbuf = torch.empty(batch * (length + 1), dim, device=DEV)
    out = torch.zeros(batch, 2 * dim, device=DEV)
    ops.gru_fwd(gi, w_hh.to(DEV), b_hh.to(DEV), batch, length, dim, hbuf, out[:, :dim])
    torch.testing.assert_close(out[:, :dim].cpu(), last.detach().float(), rtol=1e-5, atol=1e-6)
    assert torch.equal(hbuf.view(batch, length + 1, dim)[:, 0].cpu(), torch.zeros(batch, dim))
    assert torch.equal(hbuf.view(batch, length + 1, dim)[:, -1], out[:, :dim])

    dgi = torch.empty(batch * length, 3 * dim, device=DEV)
    dgh = torch.empty(batch * (length + 1), 3 * dim, device=DEV)
    ops.gru_bwd(gi, w_hh.to(DEV), b_hh.to(DEV), hbuf, batch, length, dim, glast.to(DEV), dgi, dgh)
    g_w_ih, g_w_hh = torch.zeros(3 * dim, dim, device=DEV), torch.zeros(3 * dim, dim, device=DEV)
    g_b_ih, g_b_hh = torch.zeros(3 * dim, device=DEV), torch.zeros(3 * dim, device=DEV)
    gx = torch.empty(batch * length, dim, device=DEV)
    ops.linear_bwd(dx, w_ih.to(DEV), None, dgi, 0, gx, g_w_ih, g_b_ih)
    rows = batch * (length + 1) - 1
    ops.linear_bwd(hbuf[:rows], w_hh.to(DEV), None, dgh[1:], 0, None, g_w_hh, g_b_hh)
    tol = dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gx.cpu().view(batch, length, dim), leaves[4].grad.float(), **tol)
    torch.testing.assert_close(g_w_ih.cpu(), leaves[0].grad.float(), **tol)
    torch.testing.assert_close(g_w_hh.cpu(), leaves[1].grad.float(), **tol)
    torch.testing.assert_close(g_b_ih.cpu(), leaves[2].grad.float(), **tol)
    torch.testing.assert_close(g_b_hh.cpu(), leaves[3].grad.float(), **tol)


FUSED_STACKS = [
    # (m, dims, activations)  dims[0] = input width
    (65536, [128, 64, 32, 16, 8, 64], [1, 1, 1, 1, 0]),     # NeuralCF tower at BASELINE configs[1]
    (1500, [64, 128, 8, 40, 1], [1, 0, 2, 2]),               # odd batch, sigmoid, n = 1 head, n not % 8
    (1024, [8, 8], [1]),                                     # single layer is refused (falls back): still right
    (4097, [24, 56, 104, 16], [1, 1, 1]),                    # 24-wide remainder chunks
]


@pytest.mark.parametrize("m,dims,acts_", FUSED_STACKS)
def test_fused_mlp_matches_layerwise_and_fp64(ops, m, dims, acts_):
    g = torch.Generator().manual_seed(m + len(dims))
    x = torch.randn(m, dims[0] + 8, generator=g)[:, 4:4 + dims[0]]      # column slice: ld != k
    ws = [torch.randn(n, k, generator=g) / k ** 0.5 for k, n in zip(dims[:-1], dims[1:])]
    bs = [torch.randn(n, generator=g) for n in dims[1:]]
    gy = torch.randn(m, dims[-1], generator=g)

    def run(fused):
        ops.FUSED_MLP = fused
        big = torch.zeros(m, dims[0] + 8, device=DEV)
        big[:, 4:4 + dims[0]] = x.to(DEV)
        layers = [ops.Layer(w.to(DEV), b.to(DEV), a) for w, b, a in zip(ws, bs, acts_)]
        acts = ops.mlp_fwd(big[:, 4:4 + dims[0]], layers)
        grads, gx = ops.mlp_bwd(acts, layers, gy.to(DEV), None)
        return acts, grads, gx
    try:
        acts_f, grads_f, gx_f = run(True)
        acts_l, grads_l, gx_l = run(False)
    finally:
        ops.FUSED_MLP = True

    # fp64 forward reference
    h = x.double()
    for w, b, a in zip(ws, bs, acts_):
        z = h @ w.double().T + b.double()
        h = [z, torch.relu(z), torch.sigmoid(z)][a]
    torch.testing.assert_close(acts_f[-1].cpu(), h.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(acts_f[-1], acts_l[-1], rtol=1e-5, atol=1e-5)

    # fp64 backward reference through the activations the kernel saved: a ReLU unit whose
    # pre-activation is within rounding of 0 may be on in fp32 and off in fp64, which would
    # change a whole row of dW -- the derivative is taken where the forward actually went
    saved = [a.cpu().double() for a in acts_f]
    gcur = gy.double()
    ref = []
    for k in range(len(ws) - 1, -1, -1):
        y = saved[k + 1]
        gz = gcur * [torch.ones_like(y), (y > 0).double(), y * (1 - y)][acts_[k]]
        ref.append((gz.T @ saved[k], gz.sum(0)))
        gcur = gz @ ws[k].double()
    ref.reverse()
    scale = max(1.0, m ** 0.5)
    torch.testing.assert_close(gx_f.cpu(), gcur.float(), rtol=1e-4, atol=1e-5)
    for (gw, gb), (rw, rb) in zip(grads_f, ref):
        torch.testing.assert_close(gw.cpu(), rw.float(), rtol=1e-4, atol=1e-5 * scale)
        torch.testing.assert_close(gb.cpu(), rb.float(), rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("n,world", [(100000, 8), (37, 2), (1, 3), (0, 4), (5000, 1)])
def test_shard_bucket_kernel(n, world):
    from deeplearningrecommendationsystem_amd.dist import HipShardBackend
    g = torch.Generator().manual_seed(n + world)
    vocab = 1_000_000
    ids = torch.randint(0, vocab, (n,), generator=g)
    nbad = 0
    if n >= 37:
        ids[3], ids[11] = vocab + 5, -2                                      # out of range: counted, sent as row 0
        nbad = 2
    counts, send, perm, inv = HipShardBackend.bucket(ids.to(DEV), world, vocab)
    counts, send, perm, inv = counts.cpu(), send.cpu(), perm.cpu(), inv.cpu()
    good = torch.where((ids < 0) | (ids >= vocab), torch.zeros_like(ids), ids)
    assert send.dtype == torch.int32                                     # the wire format
    assert counts.tolist() == torch.bincount(good % world, minlength=world).tolist() + [nbad]
    counts = counts[:world]
    assert sorted(perm.tolist()) == list(range(n))                       # a permutation
    assert torch.equal(inv[perm], torch.arange(n))                       # inv is its inverse
    assert torch.equal(send[perm].long(), good // world)                 # slot holds the local row
    owner_of_slot = torch.bucketize(torch.arange(n), torch.cumsum(counts, 0), right=True)
    assert torch.equal(owner_of_slot[perm], good % world)                # buckets are in rank order


@pytest.mark.parametrize("n,world,cap", [(100000, 8, 12800), (100000, 8, 12000), (37, 2, 24), (1, 3, 8), (0, 4, 8),
                                         (5000, 1, 5000)])
def test_shard_bucket_padded_kernel(n, world, cap):
    """the capacity-bounded layout: bucket w = slots [w*cap, (w+1)*cap), -1 in unused slots, overflow reported"""
    from deeplearningrecommendationsystem_amd.dist import HipShardBackend
    g = torch.Generator().manual_seed(n + world)
    vocab = 1_000_000
    ids = torch.randint(0, vocab, (n,), generator=g)
    nbad = 0
    if n >= 37:
        ids[3], ids[11] = vocab + 5, -2
        nbad = 2
    state, send, perm, inv = (t.cpu() for t in HipShardBackend.bucket_padded(ids.to(DEV), world, vocab, cap))
    good = torch.where((ids < 0) | (ids >= vocab), torch.zeros_like(ids), ids)
    counts = torch.bincount(good % world, minlength=world)
    assert state.tolist() == [int((counts > cap).any()), nbad, n, -n]
    assert send.dtype == torch.int32 and send.numel() == world * cap and inv.numel() == world * cap
    if n:
        assert int(perm.min()) >= 0 and int(perm.max()) < world * cap            # in bounds even when overflowing
        assert torch.equal(perm // cap, good % world)                         # every id sits in its owner's bucket
        assert int(inv.min()) >= 0 and int(inv.max()) < n                     # unused slots name SOME id of the batch
    for w in range(world):
        bucket = send[w * cap:(w + 1) * cap]
        used = min(int(counts[w]), cap)
        assert int((bucket >= 0).sum()) == used and (bucket[used:] == -1).all()   # placed in arrival order, rest unused
    if not state[0]:
        assert torch.equal(send[perm].long(), good // world)                  # slot holds the local row
        assert torch.equal(inv[perm], torch.arange(n))                        # inv is perm's inverse on the used slots
    # the owner's side of it and the row clearing of the persistent gradient buffer
    rows, valid, mark = (t.cpu() for t in HipShardBackend.recv_rows(send.to(DEV), 1000))
    ok = (send >= 0) & (send < 1000)
    assert torch.equal(valid.view(-1), ok.float()) and torch.equal(mark, torch.where(ok, send.long(), torch.full_like(rows, -1)))
    assert torch.equal(rows[ok], send[ok].long()) and torch.equal(rows[~ok], (torch.arange(send.numel()) % 1000)[~ok])
    for dim in (16, 7):
        table = torch.randn(1000, dim, generator=g)
        dev = table.to(DEV)
        HipShardBackend.zero_rows(dev, rows.to(DEV)[: max(1, n // 3)])
        want = table.clone()
        want[rows[: max(1, n // 3)]] = 0.0
        assert torch.equal(dev.cpu(), want)


@pytest.mark.parametrize("capacity", [None, 1.25])
def test_sharded_embedding_single_rank_on_gpu(capacity):
    import os
    import torch.distributed as dist
    from deeplearningrecommendationsystem_amd.dist import ShardedEmbedding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(0)
        full = torch.randn(5000, 16)
        emb = ShardedEmbedding(5000, 16, device=DEV, capacity_factor=capacity)
        emb.load_full_table(full.to(DEV))
        ids = torch.randint(0, 5000, (300, 7))
        got = emb(ids.to(DEV))
        assert torch.equal(got.cpu(), full[ids])
        gout = torch.randn(300, 7, 16)
        got.backward(gout.to(DEV))
        ref = torch.zeros(5000, 16).index_put_((ids.reshape(-1),), gout.reshape(-1, 16), accumulate=True)
        torch.testing.assert_close(emb.weight.grad.cpu(), ref, rtol=1e-5, atol=1e-5)
        # fresh id tensors step after step over RCCL: the gradient lives in one buffer, stale rows are cleared, the
        # capacity-bounded layout never falls back (one rank owns every id: capacity >= n)
        buf = emb.weight.grad.data_ptr()
        for step in range(3):
            emb.weight.grad = None
            ids = torch.randint(0, 5000, (40 + step, 3))
            got = emb(ids.to(DEV))
            assert torch.equal(got.cpu(), full[ids])
            gout = torch.randn(40 + step, 3, 16)
            got.backward(gout.to(DEV))
            ref = torch.zeros(5000, 16).index_put_((ids.reshape(-1),), gout.reshape(-1, 16), accumulate=True)
            torch.testing.assert_close(emb.weight.grad.cpu(), ref, rtol=1e-5, atol=1e-5)
            assert emb.weight.grad.data_ptr() == buf
        assert emb.fallbacks == 0
        with pytest.raises(IndexError):
            emb(torch.tensor([1, 5000], device=DEV))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,n,k", [(943, 1682, 1682), (7, 1682, 10), (3, 1, 1), (5, 4096, 50), (4, 4097, 4096),
                                      (6, 100000, 20), (2, 1000003, 100), (3, 20000, 1)])
def test_topk_rows_ranks_like_torch_with_ties_by_index(rows, n, k):
    """csrc/topk.hip against torch.topk on CPU (the reference's ranking call, model/mf.py:35): same scores position
    by position; where scores tie, ascending index (torch leaves the order of ties open)"""
    from deeplearningrecommendationsystem_amd import ops
    g = torch.Generator().manual_seed(rows * 31 + n + k)
    scores = torch.randn(rows, n, generator=g)
    if n >= 1682:
        scores[0] = torch.randint(0, 7, (n,), generator=g).float()          # heavy ties: only 7 distinct scores
        scores[1, 5], scores[1, 900] = float("inf"), float("-inf")
        scores[2 % rows, 17] = float("nan")                                  # NaN ranks first, as in torch
        scores[-1] = 0.25                                                    # one value everywhere: index order
    got = ops.topk_rows(scores.to(DEV), k).cpu()
    assert got.shape == (rows, k) and got.dtype == torch.int64
    # the exact expected order: stable sort by descending score, NaN first
    key = torch.where(torch.isnan(scores), torch.full_like(scores, float("inf")), scores)
    nan_first = torch.isnan(scores).double() * 1e30
    want = torch.argsort(-(key.double().clamp(-1e300, 1e300).nan_to_num(posinf=1e29, neginf=-1e29) + nan_first), dim=1, stable=True)[:, :k]
    assert torch.equal(got, want)
    ref = torch.topk(scores, k, dim=1)
    torch.testing.assert_close(torch.gather(scores, 1, got), ref.values, rtol=0, atol=0, equal_nan=True)
    # a column-major view (AutoRec's item-based ranking: topk along dim 0) and a strided row view
    view = scores.to(DEV).t().contiguous().t()                               # same values, column stride != 1
    assert torch.equal(ops.topk_rows(view, k).cpu(), want)
    kk = min(k, rows)
    cols = ops.topk_rows(scores.to(DEV), kk, dim=0).cpu()
    assert cols.shape == (kk, n)
    torch.testing.assert_close(torch.gather(scores, 0, cols), torch.topk(scores, kk, dim=0).values, rtol=0, atol=0,
                               equal_nan=True)
    with pytest.raises(RuntimeError):
        ops.topk_rows(scores.to(DEV), n + 1)


@pytest.mark.parametrize("batch,nu,ni,with_prob", [(65536, 943, 1682, True), (5000, 7, 3, False), (4096, 16000, 16000, True),
                                                    (70001, 300, 1, False)])
def test_rows1_scatter_sums_first_order_gradients_in_lds(batch, nu, ni, with_prob):
    """csrc/rows_sum.hip: the (V, 1) first-order gradients of small tables, against a float64 scatter on the host"""
    from deeplearningrecommendationsystem_amd import ops
    g = torch.Generator().manual_seed(batch + nu)
    x = torch.rand(batch, 45, generator=g)
    x[:, 0] = torch.randint(0, nu, (batch,), generator=g).float()
    x[:, 1] = torch.randint(0, ni, (batch,), generator=g).float()
    x[5, 0], x[6, 1] = nu + 3.0, -1.0                                   # ids outside their table add nothing
    gv = torch.randn(batch, 1, generator=g)
    prob = torch.rand(batch, 1, generator=g) if with_prob else None
    v = (gv * prob * (1 - prob) if with_prob else gv).double().view(-1)
    u, it = x[:, 0].long(), x[:, 1].long()
    want_u = torch.zeros(nu, dtype=torch.float64).index_put_((u[(u >= 0) & (u < nu)],), v[(u >= 0) & (u < nu)], accumulate=True)
    want_i = torch.zeros(ni, dtype=torch.float64).index_put_((it[(it >= 0) & (it < ni)],), v[(it >= 0) & (it < ni)], accumulate=True)
    gu, gi = torch.full((nu, 1), 0.5, device=DEV), torch.full((ni, 1), -0.25, device=DEV)        # (+=)
    ops.rows1_scatter(x.to(DEV), gv.to(DEV), prob.to(DEV) if with_prob else None, gu, gi)
    scale = max(1.0, (batch / min(nu, ni)) ** 0.5)
    torch.testing.assert_close(gu.cpu().view(-1), (want_u + 0.5).float(), rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(gi.cpu().view(-1), (want_i - 0.25).float(), rtol=1e-5, atol=2e-6 * scale)
    only = torch.zeros((nu, 1), device=DEV)
    ops.rows1_scatter(x.to(DEV), gv.to(DEV), prob.to(DEV) if with_prob else None, only, None)   # one table alone
    torch.testing.assert_close(only.cpu().view(-1), want_u.float(), rtol=1e-5, atol=2e-6 * scale)


def test_bce_loss_matches_torch():
    from deeplearningrecommendationsystem_amd.loss import BCELoss
    g = torch.Generator().manual_seed(2)
    for n in (1, 37, 65536):
        p = torch.rand(n, 1, generator=g)
        if n > 4:
            p[0], p[1] = 0.0, 1.0                      # the -100 clamp and the 1e-12 floor
        y = (torch.rand(n, 1, generator=g) < 0.5).float()
        a = p.clone().requires_grad_(True)
        ref = torch.nn.BCELoss()(a, y)
        ref.backward()
        b = p.to(DEV).requires_grad_(True)
        got = BCELoss()(b, y.to(DEV))
        got.backward()
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(b.grad.cpu(), a.grad, rtol=1e-5, atol=1e-7)


def test_adam_matches_torch_optim():
    from deeplearningrecommendationsystem_amd.optim import Adam
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 16), (33,), (7, 5), (1,), (257, 64)]
    ref_p = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    my_p = [t.detach().clone().to(DEV).requires_grad_(True) for t in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-2, weight_decay=1e-5)
    my_opt = Adam(my_p, lr=1e-2, weight_decay=1e-5)
    for step in range(6):
        for a, b in zip(ref_p, my_p):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.to(DEV)
        ref_opt.step()
        my_opt.step()
    for a, b in zip(ref_p, my_p):
        torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(my_opt.state[b]["exp_avg_sq"].cpu(), ref_opt.state[a]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("m,n,k,group", [(1000, 128, 64, 100), (4096, 64, 16, 32), (777, 128, 64, 37), (130, 32, 8, 130)])
def test_linear_group_fwd_adds_one_residual_row_per_group(ops, m, n, k, group):
    # DIN's first attention layer on the E-wide operand: y = relu(x W^T + u[row // L])
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    u = torch.randn((m + group - 1) // group, n, generator=g)
    want = torch.relu(x.double() @ w.double().T + u.double().repeat_interleave(group, 0)[:m]).float()
    got = ops.linear_group_fwd(x.to(DEV), w.to(DEV), None, u.to(DEV), group, ops.ACT_RELU).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # the optional sign bits are exactly (output > 0), bit (j & 31) of word j // 32
    bits = torch.full((m, n // 32), -1, dtype=torch.int32, device=DEV)
    got2 = ops.linear_group_fwd(x.to(DEV), w.to(DEV), None, u.to(DEV), group, ops.ACT_RELU, sign_bits=bits).cpu()
    assert torch.equal(got2, got)
    words = bits.cpu().numpy().view(np.uint32)
    unpacked = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(m, n).astype(bool)
    assert np.array_equal(unpacked, (got2 > 0).numpy())


@pytest.mark.parametrize("m,n,k,group", [(1000, 64, 128, 100), (4096, 32, 64, 32), (777, 64, 128, 37), (260, 16, 32, 130)])
def test_linear_dx_masked_and_group_sums(ops, m, n, k, group):
    # gx = ((gy * relu'(y)) W) * relu'(xin), gsum[row // group] += gx[row]
    g = torch.Generator().manual_seed(m + k)
    w = torch.randn(n, k, generator=g) / n ** 0.5
    y = torch.relu(torch.randn(m, n, generator=g))
    xin = torch.relu(torch.randn(m, k, generator=g))
    gy = torch.randn(m, n, generator=g)
    want = ((gy.double() * (y > 0)) @ w.double()) * (xin > 0)
    groups = (m + group - 1) // group
    wsum = torch.zeros(groups, k, dtype=torch.float64).index_add_(0, torch.arange(m) // group, want)
    gx = torch.full((m, k), float("nan"), device=DEV)
    gsum = torch.zeros(groups, k, device=DEV)
    ops.linear_dx_masked(w.to(DEV), y.to(DEV), gy.to(DEV), ops.ACT_RELU, xin.to(DEV), ops.ACT_RELU, gx, gsum, group)
    torch.testing.assert_close(gx.cpu(), want.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gsum.cpu(), wsum.float(), rtol=1e-4, atol=1e-4)
    # the mask as sign bits (what linear_group_fwd writes) instead of the activations themselves
    pos = (xin > 0).numpy().reshape(m, k // 32, 32).astype(np.uint32)
    words = (pos << np.arange(32, dtype=np.uint32)).sum(-1, dtype=np.uint32).view(np.int32)
    gx3 = torch.full((m, k), float("nan"), device=DEV)
    gsum3 = torch.zeros(groups, k, device=DEV)
    ops.linear_dx_masked(w.to(DEV), y.to(DEV), gy.to(DEV), ops.ACT_RELU, None, ops.ACT_RELU, gx3, gsum3, group,
                         sign_bits=torch.from_numpy(words).to(DEV))
    assert torch.equal(gx3, gx)
    torch.testing.assert_close(gsum3.cpu(), wsum.float(), rtol=1e-4, atol=1e-4)
    # without the optional parts it is the plain input gradient
    gx2 = torch.empty((m, k), device=DEV)
    ops.linear_dx_masked(w.to(DEV), None, gy.to(DEV), ops.ACT_NONE, None, ops.ACT_NONE, gx2)
    torch.testing.assert_close(gx2.cpu(), (gy.double() @ w.double()).float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("batch,length", [(64, 100), (8, 1), (12, 7), (260, 33), (5, 3), (17, 10), (1, 6)])
def test_gru_with_the_input_projection_inside_matches_torch_gru(ops, batch, length):
    # DIEN interest evolution (model/dien.py:47,61: nn.GRU(E, E, batch_first=True), h0 = 0) at E = 16: forward
    # states, last state, input gradient and the four parameter gradients against torch autograd on the CPU
    dim = 16
    g = torch.Generator().manual_seed(batch + length)
    gru = torch.nn.GRU(dim, dim, batch_first=True)
    with torch.no_grad():
        for prm in gru.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)
    x = torch.randn(batch, length, dim, generator=g, requires_grad=True)
    out, hn = gru(x)
    glast = torch.randn(batch, dim, generator=g)
    (hn[0] * glast).sum().backward()
    w_ih, w_hh, b_ih, b_hh = (t.detach().to(DEV) for t in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0))
    xd = x.detach().reshape(batch * length, dim).to(DEV)
    hbuf = torch.full((batch * (length + 1), dim), float("nan"), device=DEV)
    last = torch.empty(batch, dim, device=DEV)
    assert ops.gru_fused_fwd(xd, w_ih, b_ih, w_hh, b_hh, batch, length, dim, hbuf, last)
    hb = hbuf.view(batch, length + 1, dim).cpu()
    assert torch.equal(hb[:, 0], torch.zeros(batch, dim))
    torch.testing.assert_close(hb[:, 1:], out.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(last.cpu(), hn[0].detach(), rtol=1e-5, atol=1e-5)
    gx = torch.full((batch * length, dim), float("nan"), device=DEV)
    gw_ih, gw_hh = torch.zeros_like(w_ih), torch.zeros_like(w_hh)
    gb_ih, gb_hh = torch.zeros_like(b_ih), torch.zeros_like(b_hh)
    ops.gru_fused_bwd(xd, w_ih, b_ih, w_hh, b_hh, hbuf, batch, length, dim, glast.to(DEV), gx, gw_ih, gb_ih, gw_hh, gb_hh)
    torch.testing.assert_close(gx.cpu().view(batch, length, dim), x.grad, rtol=1e-4, atol=1e-5)
    for got, want in ((gw_ih, gru.weight_ih_l0), (gw_hh, gru.weight_hh_l0), (gb_ih, gru.bias_ih_l0), (gb_hh, gru.bias_hh_l0)):
        torch.testing.assert_close(got.cpu(), want.grad, rtol=1e-4, atol=1e-4 * max(1.0, (batch * length) ** 0.5 / 30))
    if batch % 4 == 0:
        # rows that are not 16-byte aligned (a column slice of a 17-wide buffer) are refused with nothing enqueued:
        # the caller then forms gi with ctr_linear_fwd and runs ctr_gru_fwd / ctr_gru_bwd (model/dien.py does)
        wide = torch.zeros(batch * length, 17, device=DEV)
        wide[:, 1:] = xd
        hbuf2 = torch.full((batch * (length + 1), dim), float("nan"), device=DEV)
        assert not ops.gru_fused_fwd(wide[:, 1:], w_ih, b_ih, w_hh, b_hh, batch, length, dim, hbuf2, None)
        assert bool(torch.isnan(hbuf2).all())
    # other widths are refused with nothing enqueued
    assert not ops.gru_fused_fwd(torch.zeros(8, 8, device=DEV), torch.zeros(24, 8, device=DEV), torch.zeros(24, device=DEV),
                                 torch.zeros(24, 8, device=DEV), torch.zeros(24, device=DEV), 4, 2, 8,
                                 torch.zeros(12, 8, device=DEV), None)


@pytest.mark.parametrize("m,n,k,act", [(1000, 64, 128, "ACT_RELU"), (333, 128, 64, "ACT_NONE"), (4097, 40, 24, "ACT_SIGMOID"),
                                       (65, 8, 16, "ACT_RELU")])
def test_linear_fwd_dot_forms_the_single_unit_layer_in_the_epilogue(ops, m, n, k, act):
    # DIN attention layers 2 + 3 (model/din.py:45-46): y = act(x W^T + b) stored, out = y u^T + c from the same tile
    g = torch.Generator().manual_seed(m + n)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    u, c = torch.randn(1, n, generator=g), torch.randn(1, generator=g)
    f = {"ACT_RELU": torch.relu, "ACT_NONE": lambda t: t, "ACT_SIGMOID": torch.sigmoid}[act]
    want_y = f(x.double() @ w.double().T + b.double())
    want_o = want_y @ u.double().T + c.double()
    y, out = ops.linear_fwd_dot(x.to(DEV), w.to(DEV), b.to(DEV), getattr(ops, act), u.to(DEV), c.to(DEV))
    torch.testing.assert_close(y.cpu(), want_y.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.cpu(), want_o.float(), rtol=1e-5, atol=2e-5)
    # (the plain entry point picks narrower column tiles for a few rows: another summation order, same values to fp32)
    torch.testing.assert_close(y.cpu(), ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), getattr(ops, act)).cpu(),
                               rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("batch,length,n,k,vocab", [(40, 100, 128, 64, 500), (7, 33, 64, 32, 50), (3, 130, 16, 128, 1000)])
def test_linear_dx_scatter_adds_the_input_gradient_where_the_rows_came_from(ops, batch, length, n, k, vocab):
    # DIN's history gradient (model/din.py:35-44 backward): table[hist[i]] += gy[i] W + attn[i] * gpool[i // L];
    # a quarter of the positions are the padding id 0 (summed per workgroup), a few ids are out of range (dropped)
    g = torch.Generator().manual_seed(batch * length + k)
    m = batch * length
    w = torch.randn(n, k, generator=g) / n ** 0.5
    gy = torch.randn(m, n, generator=g)
    attn = torch.rand(batch, length, generator=g)
    gpool = torch.randn(batch, k + 8, generator=g)[:, :k]          # a column slice: ldgp > k
    hist = torch.randint(1, vocab, (batch, length), generator=g)
    hist[torch.rand(batch, length, generator=g) < 0.25] = 0
    hist[0, 1], hist[-1, -1] = vocab, -5
    gx = gy.double() @ w.double() + attn.double().reshape(m, 1) * gpool.double().repeat_interleave(length, 0)
    flat = hist.reshape(-1)
    ok = (flat >= 0) & (flat < vocab)
    want = torch.zeros(vocab, k, dtype=torch.float64).index_add_(0, flat[ok], gx[ok])
    base = torch.randn(vocab, k, generator=g)
    table = base.clone().to(DEV)
    ops.linear_dx_scatter(w.to(DEV), gy.to(DEV), hist.to(DEV).reshape(-1), attn.to(DEV).reshape(-1), gpool.to(DEV),
                          length, table)
    torch.testing.assert_close(table.cpu(), (want + base.double()).float(), rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("m,k", [(5000, 64), (333, 16), (70001, 36), (9, 8)])
def test_linear_n1_bwd_masked_in_place(ops, m, k):
    # DIN's score layer (model/din.py:46): gx = (gy w) * relu'(x) written over x; gw += gy^T x; gb += sum gy
    g = torch.Generator().manual_seed(m + k)
    x = torch.relu(torch.randn(m, k, generator=g))
    w = torch.randn(1, k, generator=g)
    gy = torch.randn(m, 1, generator=g)
    want_gx = (gy.double() @ w.double()) * (x > 0)
    want_gw = gy.double().T @ x.double()
    buf = x.clone().to(DEV)
    gw = torch.zeros(1, k, device=DEV)
    gb = torch.zeros(1, device=DEV)
    ops.linear_n1_bwd_masked(buf, w.to(DEV), gy.to(DEV), ops.ACT_RELU, buf, gw, gb)
    torch.testing.assert_close(buf.cpu(), want_gx.float(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gw.cpu(), want_gw.float(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gb.cpu(), gy.double().sum().float().reshape(1), rtol=1e-4, atol=1e-4)
    # out of place, no weight gradients
    out = torch.full((m, k), float("nan"), device=DEV)
    ops.linear_n1_bwd_masked(x.to(DEV), w.to(DEV), gy.to(DEV), ops.ACT_RELU, out)
    torch.testing.assert_close(out.cpu(), want_gx.float(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("batch,length,dim,summed", [(50, 100, 64, True), (33, 7, 16, False), (300, 40, 8, True)])
def test_din_scatter_bwd_matches_index_add(ops, batch, length, dim, summed):
    g = torch.Generator().manual_seed(batch)
    vocab = 500
    hist = torch.randint(0, vocab, (batch, length), generator=g)
    hist[:, : length // 4] = 0                                  # padding id: the LDS-reduced row
    gh = torch.randn(batch * length, dim, generator=g)
    attn = torch.rand(batch, length, generator=g)
    gpool = torch.randn(batch if summed else batch * length, dim, generator=g)
    gp = gpool.repeat_interleave(length, 0) if summed else gpool
    want = torch.zeros(vocab, dim, dtype=torch.float64)
    want.index_add_(0, hist.reshape(-1), gh.double() + attn.reshape(-1, 1).double() * gp.double())
    gt = torch.zeros(vocab, dim, device=DEV)
    ops.din_scatter_bwd(hist.to(DEV), vocab, dim, gh.to(DEV), attn.to(DEV), gpool.to(DEV), summed, gt)
    torch.testing.assert_close(gt.cpu(), want.float(), rtol=1e-4, atol=1e-4)


def test_device_negative_sampling_properties():
    """sampler/sampler.py:16-48 on the device: num_negatives per user, none of them an observed pair, uniform over
    the user's free items, reproducible per seed, accumulating across calls like the reference's instance"""
    from deeplearningrecommendationsystem_amd.sampler import Sampler
    nu, ni, neg = 50, 300, 2000
    rng = np.random.default_rng(0)
    excluded = set()
    for u in range(nu):
        for i in rng.choice(ni, size=100 + u, replace=False):
            excluded.add((u, int(i)))
    for i in range(ni - 3):
        excluded.add((7, i))                                   # user 7 has three free items
    s = Sampler(seed=3)
    users, items, ratings = s.negative_sampling(nu, ni, excluded, neg, DEV)
    assert users.shape == items.shape == ratings.shape == (nu * neg,) and not bool(ratings.any())
    u, i = users.cpu().numpy(), items.cpu().numpy()
    assert np.array_equal(u, np.repeat(np.arange(nu), neg))
    assert not any((int(a), int(b)) in excluded for a, b in zip(u[::7], i[::7]))
    ex = np.zeros((nu, ni), dtype=bool)
    for a, b in excluded:
        ex[a, b] = True
    assert not ex[u, i].any()
    assert set(i[u == 7]) == {ni - 3, ni - 2, ni - 1}
    # uniform over the free items: chi-square of user 3's draws against a flat law
    free = np.flatnonzero(~ex[3])
    cnt = np.bincount(i[u == 3], minlength=ni)[free]
    chi2 = ((cnt - neg / free.size) ** 2 / (neg / free.size)).sum()
    assert chi2 < free.size + 6 * (2 * free.size) ** 0.5, chi2
    # same seed -> same sample; the instance accumulates over calls (sampler.py:13-14)
    u2, i2, _ = Sampler(seed=3).negative_sampling(nu, ni, excluded, neg, DEV)
    assert torch.equal(i2, items)
    u3, i3, r3 = s.negative_sampling(nu, ni, excluded, 5, DEV)
    assert u3.numel() == nu * (neg + 5) and torch.equal(i3[:nu * neg], items)
    df = Sampler(seed=4).negative_sampling2(nu, ni, excluded, 4, DEV)
    assert list(df.columns) == ['user_id', 'item_id', 'rating'] and len(df) == nu * 4 and int(df['rating'].sum()) == 0
    # default-constructed samplers (what the reference's scripts build for train / valid / test,
    # scripts/neuralcf.py:27-47) draw from different streams: equal arguments, different negatives
    a1 = Sampler().negative_sampling(nu, ni, excluded, 20, DEV)[1]
    a2 = Sampler().negative_sampling(nu, ni, excluded, 20, DEV)[1]
    assert not torch.equal(a1, a2)


def test_device_feature_assembly_equals_the_pandas_merges():
    """data/reader.py:98-101 feature(): merge on user_id then on item_id (inner joins keep the left order)"""
    import pandas as pd
    from deeplearningrecommendationsystem_amd.data import FeatureAssembler
    rng = np.random.default_rng(1)
    nu, ni, n = 40, 60, 5000
    user_data = pd.DataFrame(np.column_stack([np.arange(nu), rng.random(nu), rng.integers(0, 2, (nu, 23))]),
                             columns=['user_id', 'age'] + [f'u{c}' for c in range(23)])
    item_data = pd.DataFrame(np.column_stack([np.arange(ni), rng.integers(0, 2, (ni, 19))]),
                             columns=['item_id'] + [f'g{c}' for c in range(19)])
    user_data, item_data = user_data.sample(frac=1, random_state=0), item_data.sample(frac=1, random_state=1)  # any row order
    pairs = pd.DataFrame({'user_id': rng.integers(0, nu, n), 'item_id': rng.integers(0, ni, n), 'rating': 1})
    want = pd.merge(pd.merge(pairs, user_data, on='user_id'), item_data, on='item_id').drop('rating', axis=1)
    fa = FeatureAssembler.from_frames(user_data, item_data, DEV)
    got = fa.feature(torch.from_numpy(pairs['user_id'].values).to(DEV), torch.from_numpy(pairs['item_id'].values).to(DEV))
    assert got.shape == (n, 45)
    assert np.array_equal(got.cpu().numpy(), want.values.astype(np.float32))
    fa.check_bad_index()


@pytest.mark.parametrize("nvec,dim,batch", [(26, 16, 1000), (12, 8, 333), (9, 32, 65), (26, 16, 1), (20, 64, 130)])
def test_allpairs_many_vectors_lane_group_kernels(ops, nvec, dim, batch):
    # PNN inner products over F id fields (model/pnn.py:59-66 pairing, i < j lexicographic) and their backward
    g = torch.Generator().manual_seed(nvec * dim + batch)
    emb = torch.randn(batch, nvec * dim, generator=g)
    v = emb.double().view(batch, nvec, dim)
    iu = torch.triu_indices(nvec, nvec, 1)
    want = (v[:, iu[0]] * v[:, iu[1]]).sum(-1)
    npairs = nvec * (nvec - 1) // 2
    out = torch.full((batch, (npairs + 3) // 4 * 4), float("nan"), device=DEV)[:, :npairs]
    ops.allpairs_fwd(emb.to(DEV), nvec, dim, out=out)
    torch.testing.assert_close(out.cpu(), want.float(), rtol=1e-5, atol=1e-5)
    gp = torch.randn(batch, npairs, generator=g)
    gwant = torch.zeros(batch, nvec, dim, dtype=torch.float64)
    gwant.index_add_(1, iu[0], gp.double().unsqueeze(-1) * v[:, iu[1]])
    gwant.index_add_(1, iu[1], gp.double().unsqueeze(-1) * v[:, iu[0]])
    base = torch.randn(batch, nvec * dim, generator=g)
    # the coefficients once densely packed and once in a row padded to 16 bytes (what the models pass; the pinned
    # 26 x 16 shape takes its register-resident kernel only then)
    gp_pad = torch.full((batch, (npairs + 3) // 4 * 4), float("nan"), device=DEV)[:, :npairs]
    gp_pad.copy_(gp)
    for accumulate in (False, True):
        for coef in (gp.to(DEV), gp_pad):
            gemb = base.clone().to(DEV)
            ops.allpairs_bwd(emb.to(DEV), nvec, dim, coef, gemb, accumulate)
            ref = gwant.view(batch, -1) + (base.double() if accumulate else 0)
            torch.testing.assert_close(gemb.cpu(), ref.float(), rtol=1e-5, atol=1e-4)
