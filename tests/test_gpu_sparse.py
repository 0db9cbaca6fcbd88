"""GPU: the opt-in sparse mode of the table gradients / optimizer (sparse.py, csrc/sparse_rows.hip)
against a torch restatement of the same lazy row-wise rule (oracle.adam_update(rows=...))."""
import pytest
import torch

from oracle import ctr_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
HYPER = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)   # a decay large enough to be visible


def test_embedding_stage_sparse_backward_lists_unique_rows_and_sums():
    from deeplearningrecommendationsystem_amd import sparse
    from deeplearningrecommendationsystem_amd.model import EmbeddingStage
    fields, vocab, dim, batch = 5, 3000, 16, 4096
    torch.manual_seed(1)
    stage = EmbeddingStage(fields, vocab, dim).to(DEV).sparse_grads(True, min_rows=1)
    g = torch.Generator().manual_seed(2)
    idx = torch.randint(0, vocab, (batch, fields), generator=g)
    idx[:, 1] = idx[:, 1] % 7                                   # a field with heavy duplication
    gout = torch.randn(batch, fields * dim, generator=g)
    for rep in range(2):                                        # two backward passes accumulate, like dense grads
        out = stage(idx.to(DEV))
        out.backward(gout.to(DEV))
    for f, t in enumerate(stage.tables):
        assert t.grad is None                                   # no dense gradient handed to autograd
        rows, vals = sparse.state_of(t).pending()
        assert torch.equal(rows.cpu(), torch.unique(idx[:, f]))
        ref = torch.zeros(vocab, dim, dtype=torch.float64)
        ref.index_add_(0, idx[:, f], 2.0 * gout[:, f * dim:(f + 1) * dim].double())
        torch.testing.assert_close(vals.cpu().double(), ref[rows.cpu()], rtol=1e-4, atol=1e-4)  # fp32 atomic sums of up to ~1200 rows
        # nothing outside the pending rows
        assert int((sparse.state_of(t).grad != 0).any(1).sum()) <= rows.numel()
    stage.zero_grad()
    for t in stage.tables:
        st = sparse.state_of(t)
        assert int(st.count.item()) == 0 and not bool(st.grad.any()) and not bool(st.flags.any())


def _lazy_reference(name, module, batches, steps, sparse_keys, ids_of):
    """CPU: oracle gradients + Adam restated; tables in ``sparse_keys`` get the lazy row-wise rule"""
    params = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    for step in range(1, steps + 1):
        inputs, y = batches[(step - 1) % len(batches)]
        _, _, grads = orc.step(name, params, inputs, y)
        for k in params:
            rows = torch.unique(ids_of(k, inputs)) if k in sparse_keys else None
            orc.adam_update(params[k], grads[k], m[k], v[k], step, rows=rows, **HYPER)
    return params


def _train(module, batches, steps):
    from deeplearningrecommendationsystem_amd.optim import Adam
    opt = Adam(module.parameters(), **HYPER)
    loss_fn = torch.nn.BCELoss()
    for step in range(steps):
        inputs, y = batches[step % len(batches)]
        opt.zero_grad()
        loss = loss_fn(module(*[t.to(DEV) for t in inputs]), y.to(DEV))
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return {k: v.detach().cpu() for k, v in module.state_dict().items()}


def _compare(got, want):
    # Both sides scatter their embedding gradients with fp32 atomics, whose order is not fixed; Adam turns a last-bit
    # difference in a tiny gradient into a visible difference of the update (its step is lr * m / sqrt(v): the
    # magnitude cancels).  The absolute slack is therefore tied to the learning rate -- 1 % of one step -- on top of
    # the rounding terms; a wrong row, a missed update or a wrong hyper-parameter moves an element by a whole step.
    for k in want:
        scale = float(want[k].abs().max())
        torch.testing.assert_close(got[k], want[k], rtol=2e-4, atol=1e-6 + 2e-5 * scale + 1e-2 * HYPER["lr"],
                                   msg=lambda s, k=k: f"{k}: {s}")


def test_deepfm_n_fields_sparse_training_matches_lazy_adam_restatement():
    from deeplearningrecommendationsystem_amd.model import DeepFM
    fields, vocab, dim = 6, 500, 16
    torch.manual_seed(7)
    module = DeepFM(None, None, [32, 16, 1], dim, num_fields=fields, vocab=vocab)
    g = torch.Generator().manual_seed(8)
    batches = []
    for _ in range(2):   # two different batches: rows touched in step 1 but not in step 2 must keep their state
        ids = torch.randint(0, vocab, (300, fields), generator=g)
        batches.append(([ids], (torch.rand(300, 1, generator=g) < 0.5).float()))
    sparse_keys = {k for k in module.state_dict() if k.startswith(("embeddings.", "first_order."))} - {"first_order_bias"}
    want = _lazy_reference("deepfm_fields", module, batches, 4, sparse_keys,
                           lambda k, inputs: inputs[0][:, int(k.split(".")[1])])
    module = module.to(DEV).sparse_grads(True, min_rows=1)
    _compare(_train(module, batches, 4), want)


def test_din_sparse_training_matches_lazy_adam_restatement():
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import DIN
    torch.manual_seed(9)
    module = DIN(400, 16)
    gen = synth.generator(10)
    batches = []
    for _ in range(2):
        hist, target = synth.hist_batch(200, 12, 400, gen)
        batches.append(([hist, target], synth.labels(200, True, gen)))
    want = _lazy_reference("din", module, batches, 3, {"item_embedding.weight"},
                           lambda k, inputs: torch.cat([inputs[0].reshape(-1), inputs[1]]))
    module = module.to(DEV).sparse_grads(True, min_rows=1)
    _compare(_train(module, batches, 3), want)


def test_reference_deepfm_sparse_mode_float_id_columns():
    """the six-field reference model: ids arrive as float columns of the (B,45) matrix"""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import DeepFM
    torch.manual_seed(11)
    module = DeepFM(300, 400, [32, 16, 1], 16)
    gen = synth.generator(12)
    batches = [([synth.feature_batch(256, 300, 400, gen)], synth.labels(256, True, gen)) for _ in range(2)]
    keys = {"user_embedding.weight": 0, "item_embedding.weight": 1, "user.weight": 0, "item.weight": 1}
    want = _lazy_reference("deepfm", module, batches, 3, set(keys), lambda k, inputs: inputs[0][:, keys[k]].long())
    module = module.to(DEV).sparse_grads(True, min_rows=1)
    _compare(_train(module, batches, 3), want)


def test_sparse_mode_under_hipgraph_replay():
    """a captured step in sparse mode (scatter + row marking inside the graph, row-wise Adam outside)"""
    from deeplearningrecommendationsystem_amd.graph import GraphedStep
    from deeplearningrecommendationsystem_amd.loss import BCELoss
    from deeplearningrecommendationsystem_amd.model import DeepFM
    from deeplearningrecommendationsystem_amd.optim import Adam
    fields, vocab, dim = 4, 200, 16
    g = torch.Generator().manual_seed(13)
    ids = torch.randint(0, vocab, (2048, fields), generator=g)
    y = (torch.rand(2048, 1, generator=g) < 0.5).float()
    results = []
    for graphed in (False, True):
        torch.manual_seed(14)
        module = DeepFM(None, None, [32, 1], dim, num_fields=fields, vocab=vocab).to(DEV).sparse_grads(True, min_rows=1)
        opt = Adam(module.parameters(), **HYPER)
        ids_d, y_d = ids.to(DEV), y.to(DEV)
        if graphed:
            step = GraphedStep(module, BCELoss(), [ids_d], y_d)
            from deeplearningrecommendationsystem_amd import sparse
            sparse.discard(module.parameters())   # pending rows of the warm-up passes (NOT zero_grad: the dense
            #                                       parameters' .grad are the graph's static tensors now)
        for _ in range(3):
            if graphed:
                step()
            else:
                opt.zero_grad()
                BCELoss()(module(ids_d), y_d).backward()
            opt.step()
        torch.cuda.synchronize()
        results.append({k: v.detach().cpu() for k, v in module.state_dict().items()})
    _compare(results[1], results[0])
