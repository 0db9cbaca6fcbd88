"""GPU: the BASELINE configs at their FULL sizes (tables above 2^31 bytes, batch 65536 / 32768).

The CPU oracle cannot run these shapes in seconds, so parity is established through
size-independent properties:
  * tables are filled with a closed-form function of (row, column) that fp32 holds exactly, so the
    expected gather is computed with integer arithmetic on the host -- independent of any GPU gather;
  * scatter: rows that received a gradient == the unique ids (exact), the sum of everything
    scattered == the sum of gout (fp64), and a sample of rows (first, last, above 2^31 bytes,
    duplicates) against fp64 sums built from gout on the host;
  * DIN / DIEN forward is per-sample independent: the full-size batch is compared with the CPU oracle on
    a 256-sample slice of the same batch and the same parameters.
"""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
MOD = 1 << 24  # integers below 2^24 are exact in fp32


def _closed_form_table(vocab, dim, salt):
    """table[r, c] = ((r * dim + c) * 3 + salt) mod 2^24, as fp32, built on the device in chunks"""
    t = torch.empty(vocab, dim, dtype=torch.float32, device=DEV)
    step = 1 << 20
    cols = torch.arange(dim, device=DEV, dtype=torch.int64)
    for r0 in range(0, vocab, step):
        r = torch.arange(r0, min(vocab, r0 + step), device=DEV, dtype=torch.int64)
        t[r0:r0 + r.numel()] = (((r[:, None] * dim + cols[None, :]) * 3 + salt) % MOD).float()
    return t


def _closed_form_rows(rows: np.ndarray, dim, salt):
    c = np.arange(dim, dtype=np.int64)
    return (((rows.astype(np.int64)[:, None] * dim + c[None, :]) * 3 + salt) % MOD).astype(np.float32)


def _ids_with_edges(n, vocab, gen):
    """uniform ids plus the edge cases: first / last row, rows above 2^31 bytes, duplicates"""
    ids = torch.randint(0, vocab, (n,), generator=gen)
    ids[0], ids[1], ids[2] = 0, vocab - 1, vocab - 1           # first, last, last again
    ids[3:7] = ids[7]                                          # a 5-fold duplicate
    return ids


def _check_scatter(grad, ids, gout, dim, extra_rows):
    """`grad` (V, dim) on the device after scattering gout rows by ids (host tensors)"""
    touched = torch.unique(ids)
    nz = (grad != 0).any(dim=1).nonzero().flatten().cpu()
    assert torch.equal(nz, touched), "rows with a gradient differ from the unique ids"
    total = float(grad.double().sum())
    want = float(gout.double().sum())
    assert abs(total - want) <= 1e-6 * max(1.0, float(gout.double().abs().sum())), (total, want)
    rows = torch.unique(torch.cat([ids[:16], torch.as_tensor(extra_rows)]))
    got = grad[rows.to(DEV)].cpu()
    for k, r in enumerate(rows.tolist()):
        ref = gout[ids == r].double().sum(0)
        torch.testing.assert_close(got[k].double(), ref, rtol=1e-5, atol=1e-5, msg=lambda m, r=r: f"grad row {r}: {m}")


def test_gather_and_scatter_on_a_2_56_gb_table():
    """BASELINE configs[4] item table: (1e7, 64) fp32 = 2.56 GB, byte offsets above 2^31"""
    from deeplearningrecommendationsystem_amd import _lib, ops
    vocab, dim, n = 10_000_000, 64, 200_000
    table = _closed_form_table(vocab, dim, 5)
    gen = torch.Generator().manual_seed(99)
    ids = _ids_with_edges(n, vocab, gen)
    ids[8] = (1 << 31) // (dim * 4) + 1        # first row past 2^31 bytes
    ids[9] = 9_000_001                         # well past 2^31 bytes (element offset > 2^29)
    assert int(ids.max()) < vocab and int(ids[8]) * dim * 4 > (1 << 31) and int(ids[9]) * dim * 4 > (1 << 31)
    d_ids = ids.to(DEV)
    spec = [ops.FieldSpec(_lib.FIELD_ID_I64, dim, 0, table=table, idx=d_ids)]
    out = torch.full((n, dim), float("nan"), device=DEV)
    ops.embed_fwd(spec, None, n, out)
    assert np.array_equal(out.cpu().numpy(), _closed_form_rows(ids.numpy(), dim, 5)), "gather is not bit-exact"
    gout = torch.randn(n, dim, generator=gen)
    grad = torch.zeros_like(table)
    ops.embed_bwd(spec, None, n, gout.to(DEV), {id(table): grad})
    _check_scatter(grad, ids, gout, dim, [0, vocab - 1, int(ids[8]), int(ids[9]), int(ids[7])])


def test_embedding_stage_26_fields_1e6_rows_batch_65536():
    """BASELINE configs[2] as worded / SURVEY 8(d) cfg3b: 26 fields x 1e6 rows x emb 16, batch 65536"""
    from deeplearningrecommendationsystem_amd.model import EmbeddingStage
    fields, vocab, dim, batch = 26, 1_000_000, 16, 65536
    with torch.device(DEV):
        stage = EmbeddingStage(fields, vocab, dim)
    with torch.no_grad():
        for f, t in enumerate(stage.tables):
            t.copy_(_closed_form_table(vocab, dim, 7 * f + 1))
    gen = torch.Generator().manual_seed(26)
    idx = torch.stack([_ids_with_edges(batch, vocab, gen) for _ in range(fields)], 1).contiguous()
    out = stage(idx.to(DEV))
    want = np.concatenate([_closed_form_rows(idx[:, f].numpy(), dim, 7 * f + 1) for f in range(fields)], 1)
    assert np.array_equal(out.detach().cpu().numpy(), want), "26-field gather is not bit-exact"
    gout = torch.randn(batch, fields * dim, generator=gen)
    out.backward(gout.to(DEV))
    for f in (0, 13, 25):
        _check_scatter(stage.tables[f].grad, idx[:, f], gout[:, f * dim:(f + 1) * dim], dim, [0, vocab - 1])


def test_embedding_stage_scatter_under_a_zipf_law_hot_rows_summed_in_lds():
    """the cfg3b scatter with Zipf ids (rank r with P(rank <= r) = log r / log V): the rows a workgroup hits repeatedly
    are summed in an LDS cache before they reach the table gradient (embed_ids_hot_bwd_kernel) -- every row against
    fp64 sums built on the host, for the hottest rows (thousands of contributions) and for rows hit once"""
    from deeplearningrecommendationsystem_amd.model import EmbeddingStage
    fields, vocab, dim, batch = 26, 1_000_000, 16, 65536
    with torch.device(DEV):
        stage = EmbeddingStage(fields, vocab, dim)
    gen = torch.Generator().manual_seed(27)
    u = torch.rand(batch, fields, generator=gen, dtype=torch.float64)
    idx = (float(vocab) ** u - 1.0).long().clamp_(0, vocab - 1)
    idx[0, :], idx[1, :] = vocab - 1, 0
    out = stage(idx.to(DEV))
    gout = torch.randn(batch, fields * dim, generator=gen)
    out.backward(gout.to(DEV))
    for f in (0, 7, 25):
        ids, g = idx[:, f], gout[:, f * dim:(f + 1) * dim]
        counts = torch.bincount(ids, minlength=8)
        assert int(counts[1]) > 500, "the law must produce hot rows"
        grad = stage.tables[f].grad
        touched = torch.unique(ids)
        nz = (grad != 0).any(dim=1).nonzero().flatten().cpu()
        assert torch.equal(nz, touched), "rows with a gradient differ from the unique ids"
        want = torch.zeros(vocab, dim, dtype=torch.float64)
        want.index_add_(0, ids, g.double())
        got = grad[touched.to(DEV)].cpu().double()
        # a hot row sums thousands of N(0,1) terms in an order that changes from run to run
        scale = counts[touched].double().sqrt().unsqueeze(1)
        err = ((got - want[touched]).abs() / scale).max()
        assert float(err) < 2e-5, float(err)


@pytest.mark.parametrize("name,dim", [("din", 64), ("dien", 16)])
def test_sequence_models_config5_full_size_forward_against_oracle_slice(name, dim):
    """BASELINE configs[4] on one GPU: item vocab 1e7, L = 100, batch 32768.  DIN/DIEN score every sample
    independently, so the oracle's forward on a 256-sample slice must equal the same rows of the
    full-size launch (logits/probabilities at 1e-5 relative, north_star)."""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import DIEN, DIN
    vocab, length, batch = 10_000_000, 100, 32768
    torch.manual_seed(50 + dim)
    with torch.device(DEV):
        module = (DIN if name == "din" else DIEN)(vocab, dim)
    # xavier_normal_ on a 1e7-row table gives rows of ~5e-4: every sample would score the same to 1e-4 and a
    # wrong row would hide inside the tolerance.  N(0, 0.5) rows make the output depend on every gathered row
    with torch.no_grad():
        next(module.parameters()).normal_(0.0, 0.5)
    assert next(module.parameters()).shape == (vocab, dim)
    gen = synth.generator(77)
    hist, target = synth.hist_batch(batch, length, vocab, gen)
    # make the slice see the table's edges: last row, a row above 2^31 bytes (DIN), padding id 0
    hist[0, 0], hist[0, 1], target[1] = vocab - 1, (1 << 31) // (dim * 4) + 5 if dim == 64 else vocab - 2, vocab - 1
    module.eval()
    with torch.no_grad():
        prob = module(hist.to(DEV), target.to(DEV)).cpu()
    assert prob.shape == (batch, 1)
    sl = torch.cat([torch.arange(0, 128), torch.arange(batch - 128, batch)])   # both ends of the batch
    params = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    with torch.no_grad():
        ref = orc.FORWARDS[name](params, hist[sl], target[sl])
    assert float(ref.std()) > 0.003, "degenerate case: the scores do not depend on the rows"
    torch.testing.assert_close(prob[sl], ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["deepfm", "pnn"])
def test_n_field_models_config3_as_worded_full_size_against_oracle(name):
    """BASELINE configs[2] as worded: DeepFM / PNN with 26 fields x 1e6 vocab, emb 16, batch 65536 -- one whole
    train-loop body (forward, BCELoss, backward) against the CPU oracle: prob / loss at 1e-5, every gradient
    (26 dense (1e6,16) tables included) at the repo's gradient tolerance."""
    from deeplearningrecommendationsystem_amd.model import DeepFM, PNN
    fields, vocab, dim, batch = 26, 1_000_000, 16, 65536
    torch.manual_seed(3)
    if name == "deepfm":
        m = DeepFM(None, None, [512, 256, 128, 1], dim, num_fields=fields, vocab=vocab)
    else:
        m = PNN(dim, [256, 128, 64, 32], num_fields=fields, vocab=vocab)
    with torch.no_grad():   # xavier rows of a 1e6-row table are ~1e-3: scale them up so the output depends on them
        for k, p in m.named_parameters():
            if k.startswith(("embeddings.", "first_order.")):
                p.mul_(300.0)
    gen = torch.Generator().manual_seed(5)
    ids = torch.stack([_ids_with_edges(batch, vocab, gen) for _ in range(fields)], 1).contiguous()
    y = (torch.rand(batch, 1, generator=gen) < 0.5).float()
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    prob_ref, loss_ref, grads_ref = orc.step(name + "_fields", params, [ids], y)
    assert float(prob_ref.std()) > 0.001, "degenerate case: the output must depend on the gathered rows"
    m = m.to(DEV)
    m.train()
    prob = m(ids.to(DEV))
    loss = torch.nn.BCELoss()(prob, y.to(DEV))
    loss.backward()
    torch.testing.assert_close(prob.detach().cpu(), prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.detach().cpu(), loss_ref, rtol=1e-5, atol=1e-6)
    for k, p in m.named_parameters():
        want = grads_ref[k]
        floor = 1e-6 + 1e-5 * float(want.abs().max())
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-4, atol=floor, msg=lambda s, k=k: f"grad {k}: {s}")


@pytest.mark.parametrize("batch", [131072, 16384])
def test_ffm_config3_shape_whole_step_against_oracle(batch):
    """BASELINE configs[3] on one GPU: FFM k = 32 with 1e6-row user / item id tables (the HBM-resident-row path of
    ffm_fused.hip), global batch 131072 and the per-rank 16384 -- one train-loop body against the CPU oracle"""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import FFM
    from deeplearningrecommendationsystem_amd.model.ffm import SHARDED
    nu = ni = 1_000_000
    torch.manual_seed(7)
    m = FFM(43, 32, num_users=nu, num_items=ni)
    with torch.no_grad():   # xavier rows of a 1e6-row table are ~1e-3: scale them up so the output depends on them
        for k, p in m.named_parameters():
            if k.split(".")[0] in SHARDED + ("user", "item"):
                p.mul_(300.0)
    gen = synth.generator(11)
    x = synth.feature_batch(batch, nu, ni, gen)
    x[0, 0], x[1, 1], x[2, 0], x[2, 1] = nu - 1, ni - 1, 0, 0   # table edges
    x[3:7, 0] = x[7, 0]                                          # a 5-fold duplicate
    y = synth.labels(batch, True, gen)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    prob_ref, loss_ref, grads_ref = orc.step("ffm", params, [x], y)
    assert float(prob_ref.std()) > 0.01, "degenerate case: the output must depend on the gathered rows"
    m = m.to(DEV)
    m.train()
    prob = m(x.to(DEV))
    loss = torch.nn.BCELoss()(prob, y.to(DEV))
    loss.backward()
    torch.testing.assert_close(prob.detach().cpu(), prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.detach().cpu(), loss_ref, rtol=1e-5, atol=1e-6)
    for k, p in m.named_parameters():
        want = grads_ref[k]
        floor = 1e-6 + 1e-5 * float(want.abs().max())
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-4, atol=floor, msg=lambda s, k=k: f"grad {k}: {s}")


@pytest.mark.parametrize("name,dim", [("din", 64), ("dien", 16)])
def test_sequence_models_config5_backward_against_oracle_on_a_slice(name, dim):
    """BASELINE configs[4] tables (1e7 rows; DIN's is 2.56 GB with rows above 2^31 bytes), L = 100: a whole train-loop
    body on a 256-sample batch that touches the table's edges, against the CPU oracle -- every dense-layer gradient and
    every touched table row (the padding row 0, the last row, a row above 2^31 bytes among them); rows nobody touched
    must have no gradient.  The oracle sees the SAME rows through a compacted table (the distinct ids of the batch
    renumbered 0..n-1: gather and scatter are row copies / row sums, so the arithmetic is unchanged) instead of
    cloning 2.56 GB and building a dense 1e7-row gradient on the host."""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import DIEN, DIN
    vocab, length, batch = 10_000_000, 100, 256
    torch.manual_seed(60 + dim)
    with torch.device(DEV):
        module = (DIN if name == "din" else DIEN)(vocab, dim)
    table = next(module.parameters())
    assert table.shape == (vocab, dim)
    with torch.no_grad():
        table.normal_(0.0, 0.5)      # (xavier rows of a 1e7-row table are ~5e-4: the output would not depend on them)
    gen = synth.generator(78)
    hist, target = synth.hist_batch(batch, length, vocab, gen)
    far = (1 << 31) // (dim * 4) + 5 if dim == 64 else vocab - 2     # DIN: first rows past 2^31 bytes
    hist[0, 0], hist[0, 1], target[1], hist[2, 5] = vocab - 1, far, vocab - 1, far
    hist[3, :] = 0                                                    # an all-padding history
    y = synth.labels(batch, True, gen)
    # ---- oracle on the compacted table
    uniq, inv = torch.unique(torch.cat([hist.reshape(-1), target]), return_inverse=True)
    assert int(uniq[0]) == 0 and int(uniq[-1]) == vocab - 1 and far in uniq.tolist()
    tname = "item_embedding.weight" if name == "din" else "din.item_embedding.weight"
    params = {k: v.detach().cpu().clone() for k, v in module.state_dict().items() if k != tname}
    params[tname] = table.detach()[uniq.to(DEV)].cpu()
    chist, ctarget = inv[:batch * length].view(batch, length), inv[batch * length:]
    prob_ref, loss_ref, grads_ref = orc.step(name, params, [chist, ctarget], y)
    assert float(prob_ref.std()) > 0.003, "degenerate case: the scores do not depend on the rows"
    # ---- the HIP step on the full table
    module.train()
    prob = module(hist.to(DEV), target.to(DEV))
    loss = torch.nn.BCELoss()(prob, y.to(DEV))
    loss.backward()
    torch.testing.assert_close(prob.detach().cpu(), prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.detach().cpu(), loss_ref, rtol=1e-5, atol=1e-6)
    for k, p in module.named_parameters():
        want = grads_ref[k]
        floor = 1e-6 + 1e-5 * float(want.abs().max())
        if k == tname:
            got = p.grad[uniq.to(DEV)].cpu()                 # every touched row, in the compacted order
            torch.testing.assert_close(got, want, rtol=1e-4, atol=floor, msg=lambda s: f"touched table rows: {s}")
            assert float(want[0].abs().sum()) > 0 and float(want[-1].abs().sum()) > 0   # pad row, last row
            # rows nobody touched: the dense gradient is zero there (sum of |g| over the table == over the touched rows)
            total, touched = float(p.grad.double().abs().sum()), float(p.grad[uniq.to(DEV)].double().abs().sum())
            assert abs(total - touched) <= 1e-9 * max(1.0, total), (total, touched)
        else:
            torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-4, atol=floor, msg=lambda s, k=k: f"grad {k}: {s}")


def test_gru_config5_full_size_matrix_core_kernels_against_the_gemm_decomposition():
    # DIEN's interest evolution at cfg5 (32768 x 100 steps, E = 16): the recurrence with the input projection inside
    # (sixteen samples per wave on the matrix cores) against the independent decomposition the library also has --
    # gi as a GEMM, the DPP recurrence kernels, dX / dW_ih / dW_hh as GEMMs over dgi / dgh (model/dien.py:47,61)
    from deeplearningrecommendationsystem_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    batch, length, dim = 32768, 100, 16
    x = torch.randn(batch * length, dim, device=DEV, generator=g) * 0.5
    w_ih, w_hh = (torch.randn(3 * dim, dim, device=DEV, generator=g) * 0.3 for _ in range(2))
    b_ih, b_hh = (torch.randn(3 * dim, device=DEV, generator=g) * 0.1 for _ in range(2))
    glast = torch.randn(batch, dim, device=DEV, generator=g)
    # fused
    hbuf = torch.empty(batch * (length + 1), dim, device=DEV)
    last = torch.empty(batch, dim, device=DEV)
    assert ops.gru_fused_fwd(x, w_ih, b_ih, w_hh, b_hh, batch, length, dim, hbuf, last)
    gx = torch.empty_like(x)
    gw_ih, gw_hh, gb_ih, gb_hh = (torch.zeros_like(t) for t in (w_ih, w_hh, b_ih, b_hh))
    ops.gru_fused_bwd(x, w_ih, b_ih, w_hh, b_hh, hbuf, batch, length, dim, glast, gx, gw_ih, gb_ih, gw_hh, gb_hh)
    # decomposition
    gi = ops.linear_fwd(x, w_ih, b_ih)
    hbuf2 = torch.empty_like(hbuf)
    last2 = torch.empty_like(last)
    ops.gru_fwd(gi, w_hh, b_hh, batch, length, dim, hbuf2, last2)
    torch.testing.assert_close(hbuf, hbuf2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(last, last2, rtol=1e-5, atol=1e-5)
    assert float(hbuf.std()) > 0.05                      # not a degenerate recurrence
    dgi = torch.empty(batch * length, 3 * dim, device=DEV)
    dgh = torch.empty(batch * (length + 1), 3 * dim, device=DEV)
    ops.gru_bwd(gi, w_hh, b_hh, hbuf2, batch, length, dim, glast, dgi, dgh)
    gx2 = torch.empty_like(x)
    rw_ih, rw_hh, rb_ih, rb_hh = (torch.zeros_like(t) for t in (w_ih, w_hh, b_ih, b_hh))
    ops.linear_bwd(x, w_ih, None, dgi, ops.ACT_NONE, gx2, rw_ih, rb_ih)
    rows = batch * (length + 1) - 1
    ops.linear_bwd(hbuf2[:rows], w_hh, None, dgh[1:], ops.ACT_NONE, None, rw_hh, rb_hh)
    torch.testing.assert_close(gx, gx2, rtol=1e-4, atol=1e-5)
    scale = float(rw_ih.abs().max())
    for got, want in ((gw_ih, rw_ih), (gw_hh, rw_hh), (gb_ih, rb_ih), (gb_hh, rb_hh)):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5 * max(1.0, scale))


def test_din_history_gradient_config5_full_size_scatter_from_the_gemm_against_two_passes():
    # DIN cfg5 (1e7 x 64 table, 32768 x 100 positions, a quarter of them the padding id): the layer-1 input gradient
    # added to the table gradient from the dX GEMM's epilogue (ctr_linear_dx_scatter) against the same gradient
    # written to memory by ctr_linear_bwd and scattered by ctr_din_scatter_bwd (model/din.py:35-44 backward)
    from deeplearningrecommendationsystem_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    batch, length, dim, n1, vocab = 32768, 100, 64, 128, 10_000_000
    m = batch * length
    w = torch.randn(n1, dim, device=DEV, generator=g) / n1 ** 0.5
    gz = torch.randn(m, n1, device=DEV, generator=g) * 0.1
    attn = torch.rand(batch, length, device=DEV, generator=g)
    gpool = torch.randn(batch, 2 * dim, device=DEV, generator=g)[:, :dim]
    hist = torch.randint(1, vocab, (batch, length), device=DEV, generator=g)
    hist[torch.rand(batch, length, device=DEV, generator=g) < 0.25] = 0
    hist[0, 0], hist[0, 1] = vocab - 1, 1
    fused = torch.zeros(vocab, dim, device=DEV)
    ops.linear_dx_scatter(w, gz, hist.reshape(-1), attn.reshape(-1), gpool, length, fused)
    gh = torch.empty(m, dim, device=DEV)
    ops.linear_bwd(torch.empty(m, dim, device=DEV), w, None, gz, ops.ACT_NONE, gh, None, None)
    two = torch.zeros(vocab, dim, device=DEV)
    ops.din_scatter_bwd(hist, vocab, dim, gh, attn, gpool, True, two)
    # the padding row sums ~800 k positions in two different orders; every other row a handful
    torch.testing.assert_close(fused[0], two[0], rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(fused[1:], two[1:], rtol=1e-4, atol=1e-5)
    touched = int((two.abs().sum(1) > 0).sum())
    assert touched == int(torch.unique(hist).numel())
    assert float(fused[vocab - 1].abs().sum()) > 0
