"""GPU: model-level parity.  Every golden fixture (outputs of the reference's own
classes, tests/golden/*.npz) is replayed through the HIP modules: state_dict
loaded unchanged, forward + BCELoss + backward on cuda:0, compared with the
recorded prob / loss / parameter gradients.  Then the BASELINE.json shapes are
checked against the CPU oracle on seeded synthetic batches.

Tolerance (north_star): logits/loss within 1e-5 relative; gradients are sums of
up to B terms accumulated with fp32 atomics in no fixed order, so they get
rtol 1e-4 with an absolute floor scaled to the fixture's gradient magnitude."""
import os

import numpy as np

import pytest
import torch

import golden_util as gu
from oracle import ctr_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models():
    from deeplearningrecommendationsystem_amd import model
    return {name: getattr(model, cls) for name, cls in
            dict(mf="MatrixFactorization", neuralcf="NeuralCF", ffm="FFM", pnn="PNN",
                 deepcrossing="DeepCrossing", deepfm="DeepFM", din="DIN", dien="DIEN", deepcross="DeepCross",
                 widedeep="WideDeep", lr="LogisticRegression", nfm="NFM", afm="AFM", autorec="AutoRec").items()
            if hasattr(model, cls)}


def _implemented(name):
    return gu.load(name)["meta"]["model"] in _models()


def _run(module, inputs, y):
    module.train()
    module.zero_grad()
    prob = module(*[t.to(DEV) for t in inputs])
    loss = torch.nn.BCELoss()(prob, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return prob.detach().cpu(), loss.detach().cpu(), {k: p.grad.detach().cpu() for k, p in module.named_parameters()}


def _check_grads(got, want):
    assert set(got) == set(want)
    for k in want:
        floor = 1e-6 + 1e-5 * float(want[k].abs().max())
        torch.testing.assert_close(got[k], want[k], rtol=1e-4, atol=floor, msg=lambda m, k=k: f"grad {k}: {m}")


@pytest.mark.parametrize("name", gu.names())
def test_hip_module_reproduces_reference_fixture(name):
    g = gu.load(name)
    meta = g["meta"]
    if meta["model"] not in _models():
        pytest.skip(f"{meta['model']} not built yet")
    module = _models()[meta["model"]](*meta["args"], **meta["kwargs"])
    module.load_state_dict(g["params"], strict=True)
    module = module.to(DEV)
    prob, loss, grads = _run(module, g["inputs"], g["y"])
    assert prob.shape == g["prob"].shape
    torch.testing.assert_close(prob, g["prob"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-5, atol=1e-6)
    _check_grads(grads, g["grads"])


def _vs_oracle(key, module, inputs, y, **kw):
    params = {k: v.detach().clone() for k, v in module.state_dict().items()}
    prob_ref, loss_ref, grads_ref = orc.step(key, params, inputs, y, **kw)
    prob, loss, grads = _run(module.to(DEV), inputs, y)
    torch.testing.assert_close(prob, prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, loss_ref, rtol=1e-5, atol=1e-6)
    _check_grads(grads, grads_ref)


def test_mf_config1_against_oracle():
    # BASELINE configs[0]: model/mf.py on ml-100k ids, batch 1024
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import MatrixFactorization
    torch.manual_seed(1)
    gen = synth.generator(11)
    u, i = synth.id_batch(1024, gen=gen)
    _vs_oracle("mf", MatrixFactorization(943, 1682, 64), [u, i], synth.labels(1024, False, gen))


def test_neuralcf_config2_full_batch_against_oracle():
    # BASELINE configs[1]: emb_dim 64, batch 65536, ml-100k ids
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    torch.manual_seed(2)
    gen = synth.generator(12)
    u, i = synth.id_batch(65536, gen=gen)
    _vs_oracle("neuralcf", NeuralCF(943, 1682, 64, [128, 64, 32, 16, 8]), [u, i], synth.labels(65536, True, gen))


@pytest.mark.parametrize("layers", [[16], [16, 8], [32, 24, 12]])
def test_neuralcf_other_towers_against_oracle(layers):
    # the folded head (linear + linear2 as one dot product) with no tower, one hidden layer, and a stack
    # whose widths are not the pinned BASELINE shape (generic fused kernel)
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    torch.manual_seed(4)
    gen = synth.generator(14)
    u, i = synth.id_batch(3000, 50, 60, gen=gen)
    _vs_oracle("neuralcf", NeuralCF(50, 60, 12, layers), [u, i], synth.labels(3000, True, gen))


def test_index_out_of_range_raises_like_nn_embedding():
    from deeplearningrecommendationsystem_amd.model import MatrixFactorization
    m = MatrixFactorization(5, 6, 4).to(DEV)
    m.check_index = True
    with pytest.raises(IndexError):
        m(torch.tensor([5], device=DEV), torch.tensor([0], device=DEV))
    # and the next valid call works
    assert m(torch.tensor([4], device=DEV), torch.tensor([0], device=DEV)).shape == (1,)


def test_eval_under_no_grad_matches_train_forward():
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    torch.manual_seed(3)
    m = NeuralCF(50, 60, 8, [16, 8]).to(DEV)
    u, i = synth.id_batch(100, 50, 60)
    a = m(u.to(DEV), i.to(DEV))
    m.eval()
    with torch.no_grad():
        b = m(u.to(DEV), i.to(DEV))
    assert torch.equal(a.detach(), b)


def _feature_case(batch, nu, ni, seed):
    from deeplearningrecommendationsystem_amd import synth
    gen = synth.generator(seed)
    return [synth.feature_batch(batch, nu, ni, gen)], synth.labels(batch, True, gen)


def test_deepfm_config3_shape_against_oracle():
    # BASELINE configs[2] (parity shape): 1e6-row id tables, emb 16, batch 65536
    from deeplearningrecommendationsystem_amd.model import DeepFM
    torch.manual_seed(4)
    inputs, y = _feature_case(65536, 1_000_000, 1_000_000, 14)
    _vs_oracle("deepfm", DeepFM(1_000_000, 1_000_000, [512, 256, 128, 1], 16), inputs, y)


def test_pnn_config3_shape_against_oracle():
    from deeplearningrecommendationsystem_amd.model import PNN
    torch.manual_seed(5)
    inputs, y = _feature_case(65536, 943, 1682, 15)
    _vs_oracle("pnn", PNN(16, [256, 128, 64, 32]), inputs, y)


def test_ffm_script_shape_against_oracle():
    # scripts/ffm.py:56 FFM(43, 32); batch of BASELINE configs[3] per GPU
    from deeplearningrecommendationsystem_amd.model import FFM
    torch.manual_seed(6)
    inputs, y = _feature_case(16384, 943, 1682, 16)
    _vs_oracle("ffm", FFM(43, 32), inputs, y)


def test_deepcrossing_script_shape_against_oracle():
    # scripts/deepcrossing.py:52-53
    from deeplearningrecommendationsystem_amd.model import DeepCrossing
    torch.manual_seed(7)
    inputs, y = _feature_case(8192, 943, 1682, 17)
    _vs_oracle("deepcrossing", DeepCrossing(943, 1682, 32, [256, 128, 64, 32]), inputs, y)


def _sequence_case(batch, length, vocab, seed):
    from deeplearningrecommendationsystem_amd import synth
    gen = synth.generator(seed)
    hist, target = synth.hist_batch(batch, length, vocab, gen)
    return [hist, target], synth.labels(batch, True, gen)


def test_din_config5_shape_against_oracle():
    # BASELINE configs[4] shape (emb 64, L = 100) at a vocabulary / batch the CPU oracle finishes in seconds
    from deeplearningrecommendationsystem_amd.model import DIN
    torch.manual_seed(8)
    inputs, y = _sequence_case(2048, 100, 100_000, 18)
    _vs_oracle("din", DIN(100_000, 64), inputs, y)


def test_dien_config5_shape_against_oracle():
    from deeplearningrecommendationsystem_amd.model import DIEN
    torch.manual_seed(9)
    inputs, y = _sequence_case(2048, 100, 100_000, 19)
    _vs_oracle("dien", DIEN(100_000, 16), inputs, y)


@pytest.mark.parametrize("fields,vocab,dim,batch", [(26, 5000, 16, 4096), (3, 50, 8, 1000), (5, 100000, 64, 777)])
def test_embedding_stage_gathers_bit_exact_and_scatters_dense_grads(fields, vocab, dim, batch):
    # generalised F-field stage (SURVEY 8d cfg3b shape at test size): forward == torch.cat of
    # nn.Embedding lookups bit for bit, backward == embedding_dense_backward per table
    from deeplearningrecommendationsystem_amd.model import EmbeddingStage
    torch.manual_seed(fields)
    stage = EmbeddingStage(fields, vocab, dim).to(DEV)
    g = torch.Generator().manual_seed(batch)
    idx = torch.randint(0, vocab, (batch, fields), generator=g)
    gout = torch.randn(batch, fields * dim, generator=g)
    out = stage(idx.to(DEV))
    out.backward(gout.to(DEV))
    tabs = [t.detach().cpu() for t in stage.tables]
    assert torch.equal(out.detach().cpu(), torch.cat([tabs[f][idx[:, f]] for f in range(fields)], 1))
    for f in range(fields):
        ref = torch.zeros(vocab, dim, dtype=torch.float64)
        ref.index_add_(0, idx[:, f], gout[:, f * dim:(f + 1) * dim].double())
        torch.testing.assert_close(stage.tables[f].grad.cpu(), ref.float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["neuralcf", "pnn", "din"])
def test_gradients_of_a_model_share_one_storage_for_the_data_parallel_all_reduce(name):
    # dist.GradBucket reduces the storage behind the gradients in place when there are few
    # of them: every model's backward must hand autograd views of its one flat buffer
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.dist import GradBucket
    from deeplearningrecommendationsystem_amd.loss import BCELoss
    from deeplearningrecommendationsystem_amd.model import DIN, PNN, NeuralCF
    torch.manual_seed(11)
    gen = synth.generator(21)
    if name == "neuralcf":
        model, inputs = NeuralCF(943, 1682, 64, [128, 64, 32, 16, 8]), list(synth.id_batch(4096, gen=gen))
    elif name == "pnn":
        model, inputs = PNN(16, [256, 128, 64, 32]), [synth.feature_batch(4096, gen=gen)]
    else:
        model, inputs = DIN(5000, 16), list(synth.hist_batch(512, 20, 5000, gen))
    model = model.to(DEV)
    y = synth.labels(inputs[0].shape[0], True, gen).to(DEV)
    BCELoss()(model(*[t.to(DEV) for t in inputs]), y).backward()
    shared = GradBucket(model.parameters())._shared_storages()
    assert shared is not None and len(shared) <= GradBucket.MAX_STORAGES
    covered = sum(f.numel() for f in shared)
    assert covered >= sum(p.numel() for p in model.parameters())


def _sharded_sequence_worker(rank, world, port, kind, capacity, out):
    # two ranks share the one GPU of the test box; gloo carries the exchange (staged through
    # the host, dist._host_staged) -- the protocol and every HIP step are the production ones
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeplearningrecommendationsystem_amd import synth
        from deeplearningrecommendationsystem_amd.dist import GradBucket
        from deeplearningrecommendationsystem_amd.loss import BCELoss
        from deeplearningrecommendationsystem_amd.model import DIEN, DIN
        cls = DIN if kind == "din" else DIEN
        vocab, dim, length, per_rank = 997, 16, 7, 96
        if capacity:
            os.environ["CTR_SHARD_CAPACITY"] = capacity      # dist.ShardedEmbedding: capacity-bounded exchange layout
        torch.manual_seed(3)
        full = cls(vocab, dim).to(DEV)                      # the unsharded model on the GLOBAL batch
        torch.manual_seed(4)
        shard = cls(vocab, dim, sharded=True).to(DEV)       # this rank's rows + replicated dense layers
        sd = full.state_dict()
        for name, p in shard.named_parameters():
            if name.endswith("item_embedding.weight"):
                table = shard.item_embedding if kind == "din" else shard.din.item_embedding
                table.load_full_table(sd[name])
            else:
                p.data.copy_(sd[name])
        gen = synth.generator(5)
        table = shard.item_embedding if kind == "din" else shard.din.item_embedding
        ref = dict(full.named_parameters())
        for step in range(3):                                # a NEW (hist, target) every step, zero_grad in between
            hist, target = synth.hist_batch(world * per_rank, length, vocab, gen)
            y = synth.labels(world * per_rank, True, gen)
            hist, target, y = hist.to(DEV), target.to(DEV), y.to(DEV)
            full.zero_grad(set_to_none=True)
            shard.zero_grad(set_to_none=True)
            prob_full = full(hist, target)
            BCELoss()(prob_full, y).backward()
            mine = slice(rank * per_rank, (rank + 1) * per_rank)
            prob = shard(hist[mine], target[mine])
            BCELoss()(prob, y[mine]).backward()
            GradBucket(shard.parameters()).all_reduce_mean()
            torch.testing.assert_close(prob, prob_full[mine], rtol=1e-5, atol=1e-6)
            for name, p in shard.named_parameters():
                want = ref[name].grad
                if name.endswith("item_embedding.weight"):
                    want = want[rank::world]
                    got = p.grad[:want.shape[0]]
                else:
                    got = p.grad
                torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"step {step} {n}: {m}")
        assert table.fallbacks == 0
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def _sharded_ffm_worker(rank, world, port, capacity, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeplearningrecommendationsystem_amd import synth
        from deeplearningrecommendationsystem_amd.dist import GradBucket
        from deeplearningrecommendationsystem_amd.loss import BCELoss
        from deeplearningrecommendationsystem_amd.model import FFM
        from deeplearningrecommendationsystem_amd.model.ffm import SHARDED
        nu, ni, k, per_rank = 301, 407, 8, 160
        if capacity:
            os.environ["CTR_SHARD_CAPACITY"] = capacity
        torch.manual_seed(3)
        full = FFM(43, k, num_users=nu, num_items=ni).to(DEV)
        torch.manual_seed(4)
        shard = FFM(43, k, num_users=nu, num_items=ni, sharded=True).to(DEV)
        sd = full.state_dict()
        for name, p in shard.named_parameters():
            base = name.split(".")[0]
            if base in SHARDED:
                getattr(shard, base).load_full_table(sd[name])
            else:
                p.data.copy_(sd[name])
        gen = synth.generator(9)
        ref = dict(full.named_parameters())
        for step in range(3):                                # a NEW feature matrix every step, zero_grad in between
            x = synth.feature_batch(world * per_rank, nu, ni, gen).to(DEV)
            y = synth.labels(world * per_rank, True, gen).to(DEV)
            full.zero_grad(set_to_none=True)
            shard.zero_grad(set_to_none=True)
            prob_full = full(x)
            BCELoss()(prob_full, y).backward()
            mine = slice(rank * per_rank, (rank + 1) * per_rank)
            prob = shard(x[mine])
            BCELoss()(prob, y[mine]).backward()
            GradBucket(shard.parameters()).all_reduce_mean()
            torch.testing.assert_close(prob, prob_full[mine], rtol=1e-5, atol=1e-6)
            for name, p in shard.named_parameters():
                want = ref[name].grad
                if name.split(".")[0] in SHARDED:
                    want = want[rank::world]
                    got = p.grad[:want.shape[0]]
                else:
                    got = p.grad
                torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"step {step} {n}: {m}")
        assert all(getattr(shard, n).fallbacks == 0 for n in SHARDED)
        if capacity:
            # every user id owned by rank 0: the capacity is exceeded, every rank falls back to the exact layout
            big = 4000
            x = synth.feature_batch(world * big, nu, ni, gen)
            x[:, 0] = torch.floor(x[:, 0] / world) * world
            x = x.to(DEV)
            mine = slice(rank * big, (rank + 1) * big)
            torch.testing.assert_close(shard(x[mine]), full(x)[mine], rtol=1e-5, atol=1e-6)
            assert shard.userid_user.fallbacks == 1 and shard.itemid_user.fallbacks == 0
        out.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def _sharded_ffm_1e6_worker(rank, world, port, out):
    """BASELINE configs[3] shape on two ranks: k = 32, 1e6-row id tables (HBM-resident), 16384 samples per rank,
    against the CPU oracle of the UNSHARDED model on the global batch (trainer/trainer.py:30-38 semantics: the global
    loss is the mean over both ranks' samples)"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeplearningrecommendationsystem_amd import synth
        from deeplearningrecommendationsystem_amd.dist import GradBucket
        from deeplearningrecommendationsystem_amd.loss import BCELoss
        from deeplearningrecommendationsystem_amd.model import FFM
        from deeplearningrecommendationsystem_amd.model.ffm import SHARDED
        nu = ni = 1_000_000
        k, per_rank = 32, 16384
        torch.manual_seed(3)
        full = FFM(43, k, num_users=nu, num_items=ni)       # CPU: parameters for the oracle
        with torch.no_grad():                                # xavier rows of a 1e6-row table are ~1e-3: scale them up
            for name, p in full.named_parameters():
                if name.split(".")[0] in SHARDED + ("user", "item"):
                    p.mul_(300.0)
        sd = {n: v.detach().clone() for n, v in full.state_dict().items()}
        torch.manual_seed(4)
        with torch.device(DEV):
            shard = FFM(43, k, num_users=nu, num_items=ni, sharded=True)
        for name, p in shard.named_parameters():
            base = name.split(".")[0]
            if base in SHARDED:
                getattr(shard, base).load_full_table(sd[name].to(DEV))
            else:
                p.data.copy_(sd[name])
        gen = synth.generator(9)
        x = synth.feature_batch(world * per_rank, nu, ni, gen)
        x[0, 0], x[1, 1], x[2, 0] = nu - 1, ni - 1, 0       # table edges
        x[3:7, 0] = x[7, 0]                                  # duplicates inside a rank's batch
        y = synth.labels(world * per_rank, True, gen)
        prob_ref, loss_ref, grads_ref = orc.step("ffm", sd, [x], y)
        assert float(prob_ref.std()) > 0.01, "degenerate case: the output must depend on the gathered rows"
        mine = slice(rank * per_rank, (rank + 1) * per_rank)
        prob = shard(x[mine].to(DEV))
        BCELoss()(prob, y[mine].to(DEV)).backward()
        GradBucket(shard.parameters()).all_reduce_mean()
        torch.testing.assert_close(prob.detach().cpu(), prob_ref[mine], rtol=1e-5, atol=1e-6)
        for name, p in shard.named_parameters():
            want = grads_ref[name]
            floor = 1e-6 + 1e-5 * float(want.abs().max())
            if name.split(".")[0] in SHARDED:
                want = want[rank::world]
                got = p.grad[:want.shape[0]].cpu()
            else:
                got = p.grad.cpu()
            torch.testing.assert_close(got, want, rtol=1e-4, atol=floor, msg=lambda m, n=name: f"{n}: {m}")
        out.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_row_sharded_ffm_config3_shape_two_ranks_against_the_oracle():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_ffm_1e6_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
    results = dict(out.get(timeout=5) for _ in procs)
    assert results == {0: "ok", 1: "ok"}, results


@pytest.mark.timeout(300)
@pytest.mark.parametrize("capacity", [None, "1.25"])
def test_row_sharded_ffm_two_ranks_match_the_unsharded_model(capacity):
    # BASELINE configs[3]: the field-aware id tables row-sharded, lookups by all-to-all
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_ffm_worker, args=(r, 2, port, capacity, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    results = dict(out.get(timeout=5) for _ in procs)
    assert results == {0: "ok", 1: "ok"}, results


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,capacity", [("din", None), ("dien", None), ("din", "1.25")])
def test_row_sharded_item_table_two_ranks_match_the_unsharded_model(kind, capacity):
    # rows dealt round-robin to the ranks, ids / rows / row gradients exchanged
    # with all-to-all, dense layers replicated and their gradients averaged
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_sequence_worker, args=(r, 2, port, kind, capacity, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    results = dict(out.get(timeout=5) for _ in procs)
    assert results == {0: "ok", 1: "ok"}, results


# ---------------------------------------------------------------------------
# NeuralCF with the first tower layer on the table rows (csrc/ncf_proj.hip): taken when the vocabularies are much
# smaller than the batch (BASELINE configs[1]); the per-sample kernels are its independent cross-check
# ---------------------------------------------------------------------------
def _ncf(nu, ni, seed=0):
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    torch.manual_seed(seed)
    return NeuralCF(nu, ni, 64, [128, 64, 32, 16, 8])


def _ncf_step(module, u, i, y, project):
    from deeplearningrecommendationsystem_amd.model import neuralcf as ncf_mod
    from deeplearningrecommendationsystem_amd import ops
    old = ncf_mod.PROJECT_TABLES
    ncf_mod.PROJECT_TABLES = project
    calls = []
    real = ops.NcfProj.forward
    ops.NcfProj.forward = lambda self: (calls.append(1), real(self))[1]
    try:
        out = _run(module, [u, i], y)
    finally:
        ncf_mod.PROJECT_TABLES = old
        ops.NcfProj.forward = real
    assert bool(calls) == bool(project), "the path under test was not the one that ran"
    return out


@pytest.mark.parametrize("nu,ni,batch,dist", [(943, 1682, 16384, "uniform"), (943, 1682, 65536 + 37, "uniform"),
                                              (943, 1682, 20000, "zipf"), (5, 7, 4099, "uniform"),
                                              (3000, 5000, 40000, "uniform")])
def test_neuralcf_table_row_path_against_oracle_and_per_sample_path(nu, ni, batch, dist):
    """forward, BCELoss, backward through ctr_ncf_proj_fwd / _bwd against the CPU oracle (prob / loss 1e-5, gradients at
    the repo's tolerance) and against the per-sample kernels: ragged batches, a few huge rows (5 x 7 vocabulary),
    Zipf ids (hot rows shared by many waves of the segment sum), vocabularies that are not multiples of 16"""
    from deeplearningrecommendationsystem_amd import synth
    module = _ncf(nu, ni, 11)
    gen = synth.generator(batch)
    if dist == "zipf":
        u = (float(nu) ** torch.rand(batch, generator=gen, dtype=torch.float64) - 1.0).long().clamp_(0, nu - 1)
        i = (float(ni) ** torch.rand(batch, generator=gen, dtype=torch.float64) - 1.0).long().clamp_(0, ni - 1)
    else:
        u, i = synth.id_batch(batch, nu, ni, gen)
    u[0], i[0], u[1], i[1] = 0, 0, nu - 1, ni - 1
    y = synth.labels(batch, True, gen)
    params = {k: v.detach().clone() for k, v in module.state_dict().items()}
    prob_ref, loss_ref, grads_ref = orc.step("neuralcf", params, [u, i], y)
    module = module.to(DEV)
    prob, loss, grads = _ncf_step(module, u, i, y, True)
    torch.testing.assert_close(prob, prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, loss_ref, rtol=1e-5, atol=1e-6)
    _check_grads(grads, grads_ref)
    prob2, loss2, grads2 = _ncf_step(module, u, i, y, False)
    torch.testing.assert_close(prob, prob2, rtol=1e-5, atol=1e-6)
    _check_grads(grads, grads2)


@pytest.mark.parametrize("shape,batch", [((943, 1682, 256, [512, 256, 128, 64, 32]), 65536),
                                         ((943, 1682, 32, [64, 32, 16]), 12000), ((50, 70, 16, [96, 24, 8]), 5003)])
def test_neuralcf_any_tower_table_row_path_against_oracle_and_per_sample_path(shape, batch):
    """the reference script's own shape (scripts/neuralcf.py:60: NeuralCF(943, 1682, 256, [512, 256, 128, 64, 32])) at batch
    65536, and two other towers: the composed table-row path (_NeuralCFRowsFunction: projected tables, row sums, products
    over the table rows) against the CPU oracle and against the per-sample kernels"""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    from deeplearningrecommendationsystem_amd.model import neuralcf as ncf_mod
    nu, ni, mf, layers = shape
    torch.manual_seed(17)
    module = NeuralCF(nu, ni, mf, layers)
    gen = synth.generator(batch)
    u, i = synth.id_batch(batch, nu, ni, gen)
    u[0], i[0], u[1], i[1] = 0, 0, nu - 1, ni - 1
    y = synth.labels(batch, True, gen)
    params = {k: v.detach().clone() for k, v in module.state_dict().items()}
    prob_ref, loss_ref, grads_ref = orc.step("neuralcf", params, [u, i], y)
    module = module.to(DEV)
    calls = []
    real = ncf_mod._NeuralCFRowsFunction.forward
    try:
        ncf_mod._NeuralCFRowsFunction.forward = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        prob, loss, grads = _run(module, [u, i], y)
    finally:
        ncf_mod._NeuralCFRowsFunction.forward = staticmethod(real)
    assert calls, "the composed table-row path did not run"
    torch.testing.assert_close(prob, prob_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, loss_ref, rtol=1e-5, atol=1e-6)
    _check_grads(grads, grads_ref)
    prob2, loss2, grads2 = _ncf_step(module, u, i, y, False)
    torch.testing.assert_close(prob, prob2, rtol=1e-5, atol=1e-6)
    _check_grads(grads, grads2)


def test_neuralcf_table_row_path_counters_survive_unusual_call_orders():
    """the sample counters behind the sort-free bucketing (ops.NcfCounts) are cleared by the backward for the next
    forward: training forwards without a backward, two forwards before their backwards, a second backward"""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.loss import BCELoss
    model = _ncf(301, 407, seed=11).to(DEV)
    gen = synth.generator(12)

    def batch():
        u, i = synth.id_batch(8192, 301, 407, gen=gen)
        return u.to(DEV), i.to(DEV), synth.labels(8192, True, gen).to(DEV)

    def grads_of(u, i, y):
        model.zero_grad(set_to_none=True)
        BCELoss()(model(u, i), y).backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    a, b = batch(), batch()
    want_a, want_b = grads_of(*a), grads_of(*b)
    # a training forward whose graph is dropped, then a normal step
    model(*a[:2])
    got = grads_of(*a)
    for n in want_a:
        torch.testing.assert_close(got[n], want_a[n], rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"after a dropped forward, {n}: {m}")
    # two forwards alive at once, backwards in the other order
    model.zero_grad(set_to_none=True)
    pa, pb = model(*a[:2]), model(*b[:2])
    BCELoss()(pb, b[2]).backward()
    gb = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    BCELoss()(pa, a[2]).backward()
    ga = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    for n in want_a:
        torch.testing.assert_close(ga[n], want_a[n], rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"interleaved a, {n}: {m}")
        torch.testing.assert_close(gb[n], want_b[n], rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"interleaved b, {n}: {m}")
    # a second backward over one forward is refused, not silently wrong
    loss = BCELoss()(model(*a[:2]), a[2])
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="backward"):     # (the autograd function drops its saved state after a backward)
        loss.backward()
    got = grads_of(*b)                      # and the model is still usable
    for n in want_b:
        torch.testing.assert_close(got[n], want_b[n], rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"after the refusal, {n}: {m}")


def test_neuralcf_table_row_path_bad_ids_and_inference():
    """an id outside its table reads row 0 and raises the flag in the forward (nn.Embedding would raise IndexError:
    the gradients of such a step are not defined, they only have to stay finite and leave every row of the bad id's
    table alone that no good sample touched); under no_grad the forward keeps no counters and gives the same values"""
    from deeplearningrecommendationsystem_amd import synth
    nu, ni, batch = 50, 70, 8192
    module = _ncf(nu, ni, 3).to(DEV)
    gen = synth.generator(4)
    u, i = synth.id_batch(batch, nu, ni, gen)
    u[u == 7] = 8                      # no good sample touches user row 7
    y = synth.labels(batch, True, gen)
    ub, ib = u.clone(), i.clone()
    ub[(torch.arange(batch) % 97) == 5] = nu + 3
    ib[11] = -1
    module.check_index = False
    prob, loss, grads = _ncf_step(module, ub, ib, y, True)
    with pytest.raises(IndexError):
        module.check_bad_index()
    assert all(bool(torch.isfinite(g).all()) for g in grads.values()) and bool(torch.isfinite(prob).all())
    assert float(grads["MLP_Embedding_User.weight"][7].abs().sum()) == 0.0
    assert float(grads["GMF_Embedding_User.weight"][7].abs().sum()) == 0.0
    with torch.no_grad():
        p_eval = module(u.to(DEV), i.to(DEV))
    p_train = module(u.to(DEV), i.to(DEV))
    torch.testing.assert_close(p_eval, p_train.detach(), rtol=0, atol=0)


def test_neuralcf_recommendation_over_the_whole_grid_matches_the_per_user_loop():
    # the reference ranks user by user (model/neuralcf.py:61-72); the mirror scores
    # the user x item grid in a few launches -- same ranking
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    torch.manual_seed(8)
    nu, ni = 23, 31
    model = NeuralCF(nu, ni, 8, [16, 8, 4]).to(DEV)
    got = model.recommendation(nu, ni, chunk=200)            # several chunks, last one ragged
    items = torch.arange(ni, device=DEV)
    with torch.no_grad():
        for u in range(nu):
            scores = model(torch.full((ni,), u, device=DEV), items).view(-1)
            want = torch.topk(scores, ni).indices.cpu().numpy()
            assert (got[u] == want).all() or torch.equal(scores[got[u]], scores[want])   # ties may swap


def test_trainer_mirror_trains_neuralcf_eager_and_graphed():
    # Trainer call convention of trainer/trainer.py:23-78 on the HIP module; the graphed
    # step must produce the same parameter updates as the eager one
    from torch import optim
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.model import NeuralCF
    from deeplearningrecommendationsystem_amd.trainer import Trainer
    gen = synth.generator(5)
    u, i = synth.id_batch(4096, 50, 60, gen)
    y = ((u + i) % 2 == 0).float().view(-1, 1)
    u, i, y = u.to(DEV), i.to(DEV), y.to(DEV)
    finals = []
    for graph in (False, True):
        torch.manual_seed(0)
        m = NeuralCF(50, 60, 8, [16, 8]).to(DEV)
        t = Trainer(m, torch.nn.BCELoss(), optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-5), graph=graph)
        first = None
        for _ in range(30):
            t.train_loop(u, i, train_rating=y)
            first = first if first is not None else t.train_loss.item()
        t.valid_loop(u, i, valid_rating=y)
        t.test_loop(u, i, test_rating=y)
        tr, va, te = t.model_eval(0)
        assert t.train_loss.item() < first            # it learns
        assert 0.0 <= va[4] <= 1.0
        finals.append(t.valid_loss.item())
    assert abs(finals[0] - finals[1]) < 1e-3 * max(1.0, abs(finals[0]))
    with pytest.raises(ValueError):
        t.train_loop(u, i, y, train_rating=y)


def test_evaluator_matches_definitions():
    from deeplearningrecommendationsystem_amd.evaluator import Evaluator
    y = torch.tensor([1., 0., 1., 1., 0., 0.])
    p = torch.tensor([0.9, 0.8, 0.4, 0.6, 0.2, 0.6])
    acc, prec, rec, f1, auc = Evaluator.eval(y, p)
    assert abs(acc - 3 / 6) < 1e-6 and abs(prec - 2 / 4) < 1e-6 and abs(rec - 2 / 3) < 1e-6
    # the reference feeds the THRESHOLDED predictions to roc_auc_score (evaluator/evaluator.py:17-19):
    # hard = [1,1,0,1,0,1] -> TPR 2/3, TNR 1/3 -> 0.5
    assert abs(auc - 0.5) < 1e-6
    # the ranking AUC of the raw scores is the separately named extra.  9 (pos, neg) pairs;
    # pos > neg: (.9: 3) + (.4: 1) + (.6: 1 + tie .5) = 5.5
    assert abs(Evaluator.score_auc(y, p) - 5.5 / 9) < 1e-6


def test_graphed_step_survives_pool_reallocation_between_replays():
    """DESIGN "hipGraph" note (r1: a captured hipMemsetAsync of the BCE loss scalar faulted on replay).  The
    buffers a captured step zeroes and writes -- the loss scalar, the flat gradient buffer, the reduction
    workspaces -- live in the graph's private pool for the graph's lifetime.  This test replays a captured step,
    then churns the allocator (big allocations, empty_cache, a SECOND captured step of another model whose
    pool is carved afterwards), replays the first graph again and requires the same loss and gradients as an
    eager step: the kernels that took over the zeroing write the same addresses a memset node would, so a
    lifetime problem of those buffers would show here."""
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd.graph import GraphedStep
    from deeplearningrecommendationsystem_amd.loss import BCELoss
    from deeplearningrecommendationsystem_amd.model import NeuralCF, MatrixFactorization
    torch.manual_seed(21)
    gen = synth.generator(21)
    u, i = synth.id_batch(8192, gen=gen)
    y = synth.labels(8192, True, gen)
    m = NeuralCF(943, 1682, 64, [128, 64, 32, 16, 8]).to(DEV)
    ud, idv, yd = u.to(DEV), i.to(DEV), y.to(DEV)
    # eager reference on the same weights
    m.zero_grad(set_to_none=True)
    loss_e = BCELoss()(m(ud, idv), yd)
    loss_e.backward()
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    loss_e = float(loss_e)
    g1 = GraphedStep(m, BCELoss(), [ud, idv], yd)

    def check():
        loss = float(g1())
        assert abs(loss - loss_e) <= 1e-6 * max(1.0, abs(loss_e))
        for k, p in m.named_parameters():
            floor = 1e-6 + 1e-5 * float(ref[k].abs().max())
            torch.testing.assert_close(p.grad, ref[k], rtol=1e-4, atol=floor, msg=lambda s, k=k: f"{k}: {s}")

    check()
    check()
    # churn: grow and release the ordinary pool, then capture another graph (its own private pool)
    junk = [torch.zeros(64 << 20, device=DEV) for _ in range(4)]
    del junk
    torch.cuda.empty_cache()
    m2 = MatrixFactorization(943, 1682, 64).to(DEV)
    g2 = GraphedStep(m2, BCELoss(), [ud, idv], y.view(-1).to(DEV))
    g2()
    torch.cuda.empty_cache()
    check()
    g2()
    check()


# ---------------------------------------------------------------------------
# N-id-field generalisation of DeepFM / PNN (BASELINE configs[2]: "26 fields x 1e6 vocab")
# ---------------------------------------------------------------------------
def _fields_case(batch, vocabs, seed, as_float=False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.stack([torch.randint(0, v, (batch,), generator=g) for v in vocabs], 1)
    ids[0, 0], ids[1 % batch, 0] = 0, vocabs[0] - 1           # first / last row
    if batch > 4:
        ids[2:4] = ids[4]                                      # duplicate samples -> accumulated gradient rows
    y = (torch.rand(batch, 1, generator=g) < 0.5).float()
    return [ids.float() if as_float else ids], y


@pytest.mark.parametrize("fields,vocab,dim,hidden,batch,as_float", [
    (26, 5000, 16, [64, 32, 1], 4096, False),      # configs[2] field count at test size
    (3, [7, 50, 1000], 8, [16, 1], 333, False),    # a vocabulary per field, ragged batch
    (6, 40, 32, [32, 16, 1], 1, False),            # single sample
    (26, 300, 16, [32, 1], 2500, True),            # ids carried as float32 like the reference's id columns
    (5, 64, 64, [16, 8, 1], 700, False),
])
def test_deepfm_n_fields_against_oracle(fields, vocab, dim, hidden, batch, as_float):
    from deeplearningrecommendationsystem_amd.model import DeepFM
    torch.manual_seed(fields * 10 + dim)
    m = DeepFM(None, None, hidden, dim, num_fields=fields, vocab=vocab)
    with torch.no_grad():
        m.first_order_bias.fill_(0.3)
    vocabs = [vocab] * fields if isinstance(vocab, int) else vocab
    inputs, y = _fields_case(batch, vocabs, 31 + fields, as_float)
    _vs_oracle("deepfm_fields", m, inputs, y)


@pytest.mark.parametrize("fields,vocab,dim,hidden,batch", [
    (26, 5000, 16, [64, 32], 4096),                # 325 inner products
    (3, [7, 50, 1000], 8, [16, 8], 333),
    (7, 100, 16, [32, 16], 1),
])
def test_pnn_n_fields_against_oracle(fields, vocab, dim, hidden, batch):
    from deeplearningrecommendationsystem_amd.model import PNN
    torch.manual_seed(fields * 10 + dim + 1)
    m = PNN(dim, hidden, num_fields=fields, vocab=vocab)
    vocabs = [vocab] * fields if isinstance(vocab, int) else vocab
    inputs, y = _fields_case(batch, vocabs, 41 + fields)
    _vs_oracle("pnn_fields", m, inputs, y)


def test_n_field_ctor_defaults_leave_the_reference_models_unchanged():
    from deeplearningrecommendationsystem_amd.model import DeepFM, PNN
    assert "user_embedding.weight" in DeepFM(943, 1682, [32, 1], 8).state_dict()
    assert "user_embed.weight" in PNN(8, [16, 8]).state_dict()
    with pytest.raises(ValueError):
        DeepFM(None, None, [32, 1], 8, num_fields=26)           # vocab missing
    with pytest.raises(ValueError):
        PNN(8, [16, 8], "out", num_fields=4, vocab=10)           # inner mode only


@pytest.mark.parametrize("script", ["mf", "neuralcf", "ffm", "pnn", "deepcrossing", "deepfm", "din", "dien"])
def test_script_counterparts_run_with_the_reference_import_lines(script, capsys, monkeypatch):
    """scripts/<m>.py: the reference's ``from model.<m> import ...`` / ``from trainer.trainer import Trainer``
    lines resolved through compat/, the script's model construction, two epochs on synthetic data"""
    import runpy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setattr(sys, "argv", [script + ".py", "--epochs", "2", "--train", "6000"])
    monkeypatch.syspath_prepend(os.path.join(root, "scripts"))
    runpy.run_path(os.path.join(root, "scripts", script + ".py"), run_name="__main__")
    out = capsys.readouterr().out
    assert "Epoch 2:" in out and "Training Loss" in out and "ROC AUC Score" in out


# ---------------------------------------------------------------------------
# batched recommendation(): same ranking as the reference's per-user loop
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", gu.rec_names())
def test_recommendation_reproduces_the_reference_fixture(name):
    """the ids the REFERENCE's recommendation() returned (tests/golden/rec_*.npz, oracle/make_golden.py: model/mf.py:28-35,
    neuralcf.py:61-72, pnn.py:133-143, deepfm.py:85-95, din.py:55-66, dien.py:70-81) against the batched HIP
    recommendation() on the same state_dict and arguments; a differing position must be a tie in the reference's scores"""
    import pandas as pd
    g = gu.load_rec(name)
    meta = g["meta"]
    module = _models()[meta["model"]](*meta["args"])
    module.load_state_dict(g["params"], strict=True)
    module = module.to(DEV).eval()
    nu, ni = g["num_users"], g["num_items"]
    if meta["kind"] == "ids":
        got = module.recommendation(nu, ni)
    elif meta["kind"] == "frame":
        cols = ["user_id", "item_id"] + [f"f{c}" for c in range(43)]
        got = module.recommendation(nu, pd.DataFrame(g["frame"], columns=cols), g["k"])
    else:
        got = module.recommendation(nu, ni, g["hist_list"], g["k"])
    # fp32 scores differ from the reference's CPU sums in the last bits: near-ties (1e-5 relative) may swap
    gu.assert_same_ranking(got, g["topk"], g["scores"], tol=1e-5)


def _per_user_loop_feature(model, num_users, user_item, k):
    """the reference's recommendation() body (model/pnn.py:133-143), one forward per user"""
    rows = []
    with torch.no_grad():
        for u in range(num_users):
            feats = torch.tensor(user_item[user_item['user_id'] == u].values, dtype=torch.float32, device=DEV)
            scores = model(feats)
            rows.append(torch.topk(scores, k, dim=0).indices.view(1, -1).tolist()[0])
    return np.array(rows)


@pytest.mark.parametrize("name", ["pnn", "deepfm", "ffm", "deepcrossing"])
def test_batched_recommendation_matches_the_per_user_loop_feature_models(name):
    import pandas as pd
    from deeplearningrecommendationsystem_amd import synth
    from deeplearningrecommendationsystem_amd import model as zoo
    nu, ni, k = 23, 40, 10
    torch.manual_seed(31)
    m = dict(pnn=lambda: zoo.PNN(8, [16, 8], num_users=nu, num_items=ni), deepfm=lambda: zoo.DeepFM(nu, ni, [16, 1], 8),
             ffm=lambda: zoo.FFM(43, 8, num_users=nu, num_items=ni),
             deepcrossing=lambda: zoo.DeepCrossing(nu, ni, 8, [16, 8]))[name]().to(DEV)
    gen = synth.generator(5)
    x = synth.feature_batch(nu * ni, nu, ni, gen)
    # every (user, item) pair once, rows shuffled: the frame need not be grouped by user
    grid = torch.cartesian_prod(torch.arange(nu), torch.arange(ni))[torch.randperm(nu * ni, generator=gen)]
    x[:, 0], x[:, 1] = grid[:, 0].float(), grid[:, 1].float()
    cols = ['user_id', 'item_id'] + [f"f{c}" for c in range(43)]
    frame = pd.DataFrame(x.numpy(), columns=cols)
    want = _per_user_loop_feature(m, nu, frame, k)
    got = m.recommendation(nu, frame, k)
    assert got.shape == (nu, k) and np.array_equal(got, want)


@pytest.mark.parametrize("cls", ["DIN", "DIEN"])
def test_batched_recommendation_matches_the_per_user_loop_sequence_models(cls):
    from deeplearningrecommendationsystem_amd import model as zoo
    nu, ni, k = 17, 60, 12
    torch.manual_seed(33)
    m = getattr(zoo, cls)(ni, 16).to(DEV)
    rng = np.random.default_rng(3)
    hist_list = np.array([rng.integers(0, ni, size=rng.choice([3, 3, 7, 40, 40, 1])) for _ in range(nu)], dtype=object)
    want = []
    with torch.no_grad():   # the reference's loop (model/din.py:55-66)
        for u in range(nu):
            target = torch.arange(0, ni).to(DEV)
            hist = torch.tensor(hist_list[u]).repeat(ni, 1).to(DEV)
            want.append(torch.topk(m(hist, target), k, dim=0).indices.view(1, -1).tolist()[0])
    got = m.recommendation(nu, ni, hist_list, k)
    assert np.array_equal(got, np.array(want))
