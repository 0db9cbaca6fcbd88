#!/usr/bin/env python3
"""Headline benchmark: CTR samples/s, forward+backward, batch 65536 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload neuralcf]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one Trainer.train_loop body without the optimizer (reference
trainer/trainer.py:24-38): zero_grad, model forward, BCELoss, backward -- on
synthetic ml-100k-shaped ids already resident in HBM.  With N > 1 every rank
runs its own batch of 65536 (weak scaling) and the replicated parameters'
gradients are averaged with one RCCL all-reduce inside the step.

Prints ONE JSON line (rank 0) with the contract keys plus
  roofline      dominant kernel of the step (by measured time), HIP events on the launch stream
  cpu_baseline  the CPU oracle (oracle/ctr_oracle.py) timed on this box's host cores
  kernels       per-kernel breakdown of one step; gather_roofline = the embedding-stage kernel
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TF = 157.3    # dense fp32 matrix peak
BATCH = 65536


WORKLOADS = {
    # name: (oracle key, description)   -- shapes from BASELINE.json configs / SURVEY.md 8(d)
    "neuralcf": "neuralcf mf_dim=64 layers=[128,64,32,16,8] users=943 items=1682 batch=65536/gpu (BASELINE configs[1])",
    "neuralcf_script": "neuralcf mf_dim=256 layers=[512,256,128,64,32] users=943 items=1682 batch=65536/gpu (the reference "
                       "script's own shape, scripts/neuralcf.py:60)",
    "mf": "mf emb=64 users=943 items=1682 batch=65536/gpu",
    "deepfm": "deepfm users=items=1e6 emb=16 hidden=[512,256,128,1] batch=65536/gpu (BASELINE configs[2])",
    "pnn": "pnn inner emb=16 hidden=[256,128,64,32] ml-100k vocab batch=65536/gpu (BASELINE configs[2])",
    "ffm": "ffm k=32 ml-100k vocab batch=65536/gpu",
    "deepcrossing": "deepcrossing emb=32 hidden=[256,128,64,32] ml-100k vocab batch=65536/gpu",
    "din": "din items=1e7 emb=64 L=100 batch=32768/gpu (BASELINE configs[4], single GPU)",
    "dien": "dien items=1e7 emb=16 L=100 batch=32768/gpu (BASELINE configs[4], single GPU)",
    # further zoo models built from the same kernels (script shapes)
    "deepcross": "deepcross (DCN) 3 cross layers, deep=[512,256,128,1] emb=128 (d=641) ml-100k vocab batch=65536/gpu",
    "widedeep": "widedeep hidden=[512,256,128,1] emb=128 ml-100k vocab batch=65536/gpu",
    "lr": "lr 43 features ml-100k vocab batch=65536/gpu",
    "nfm": "nfm hidden=[512,256,128,1] emb=128 ml-100k vocab batch=65536/gpu",
    "afm": "afm emb=128 attention=64 ml-100k vocab batch=65536/gpu",
    # the embedding stage alone at the roofline shape of SURVEY.md 8(d) cfg3b (metric ii: gather GB/s)
    # BASELINE configs[2] as worded: the N-id-field generalisation of the two models
    "deepfm26": "deepfm 26 id fields x 1e6 rows emb=16 hidden=[512,256,128,1] batch=65536/gpu (BASELINE configs[2] as worded)",
    "pnn26": "pnn inner 26 id fields x 1e6 rows emb=16 hidden=[256,128,64,32] batch=65536/gpu (BASELINE configs[2] as worded)",
    "gather26": "embedding stage 26 fields x 1e6 rows x emb=16 batch=65536/gpu, uniform ids (gather fwd + dense-grad scatter bwd)",
    "gather26zipf": "embedding stage 26 fields x 1e6 rows x emb=16 batch=65536/gpu, Zipf(1.05) ids",
}


def make_inputs(name: str, rank: int, batch: int):
    from deeplearningrecommendationsystem_amd import synth
    gen = synth.generator(1234 + rank)
    if name in ("neuralcf", "mf", "neuralcf_script"):
        u, i = synth.id_batch(batch, gen=gen)
        return [u, i], synth.labels(batch, name != "mf", gen)
    if name in ("deepfm",):
        return [synth.feature_batch(batch, 1_000_000, 1_000_000, gen)], synth.labels(batch, True, gen)
    if name in ("pnn", "ffm", "deepcrossing", "deepcross", "widedeep", "lr", "nfm", "afm"):
        return [synth.feature_batch(batch, gen=gen)], synth.labels(batch, True, gen)
    if name in ("din", "dien"):
        hist, target = synth.hist_batch(batch, 100, 10_000_000, gen)
        return [hist, target], synth.labels(batch, True, gen)
    if name in ("deepfm26", "pnn26"):
        return [torch.randint(0, 1_000_000, (batch, 26), generator=gen)], synth.labels(batch, True, gen)
    if name in ("gather26", "gather26zipf"):
        if name == "gather26":
            idx = torch.randint(0, 1_000_000, (batch, 26), generator=gen)
        else:
            idx = zipf_ids(batch, 26, 1_000_000, gen)
        # the "target" is the gradient fed back into the stage (fixed, N(0,1))
        return [idx], torch.randn(batch, 26 * 16, generator=gen)
    raise SystemExit(f"unknown workload {name}")


def make_model(name: str, shard: bool = False):
    from deeplearningrecommendationsystem_amd import model as zoo
    torch.manual_seed(1234)  # identical replicas on every rank
    if shard:
        # the 1e7-row item table dealt round-robin to the ranks, lookups by all-to-all
        if name == "din":
            return zoo.DIN(10_000_000, 64, sharded=True)
        if name == "dien":
            return zoo.DIEN(10_000_000, 16, sharded=True)
        if name == "ffm":  # BASELINE configs[3]: id vocabularies raised to 1e6, field-aware id tables sharded
            return zoo.FFM(43, 32, num_users=1_000_000, num_items=1_000_000, sharded=True)
        raise SystemExit("--shard applies to the din / dien / ffm workloads (their 1e6..1e7-row tables)")
    if name == "neuralcf":
        return zoo.NeuralCF(943, 1682, 64, [128, 64, 32, 16, 8])
    if name == "neuralcf_script":
        return zoo.NeuralCF(943, 1682, 256, [512, 256, 128, 64, 32])
    if name == "mf":
        return zoo.MatrixFactorization(943, 1682, 64)
    if name == "deepfm":
        return zoo.DeepFM(1_000_000, 1_000_000, [512, 256, 128, 1], 16)
    if name == "pnn":
        return zoo.PNN(16, [256, 128, 64, 32])
    if name == "ffm":
        return zoo.FFM(43, 32)
    if name == "deepcrossing":
        return zoo.DeepCrossing(943, 1682, 32, [256, 128, 64, 32])
    if name == "din":
        return zoo.DIN(10_000_000, 64)
    if name == "dien":
        return zoo.DIEN(10_000_000, 16)
    if name == "deepcross":
        return zoo.DeepCross(943, 1682, 3, [512, 256, 128, 1], 128)   # scripts/deepcross.py:52-53
    if name == "widedeep":
        return zoo.WideDeep(943, 1682, [512, 256, 128, 1], 128)       # scripts/widedeep.py
    if name == "lr":
        return zoo.LogisticRegression(943, 1682, 43)                  # scripts/lr.py
    if name == "afm":
        return zoo.AFM(943, 1682, 128, 64)                            # scripts/afm.py:52
    if name == "nfm":
        return zoo.NFM(943, 1682, [512, 256, 128, 1], 128)            # scripts/nfm.py:53
    if name == "deepfm26":
        return zoo.DeepFM(None, None, [512, 256, 128, 1], 16, num_fields=26, vocab=1_000_000)
    if name == "pnn26":
        return zoo.PNN(16, [256, 128, 64, 32], num_fields=26, vocab=1_000_000)
    if name in ("gather26", "gather26zipf"):
        return zoo.EmbeddingStage(26, 1_000_000, 16)
    raise SystemExit(f"unknown workload {name}")


def batch_of(name: str) -> int:
    return 32768 if name in ("din", "dien") else BATCH


def build_workload(name: str, device, rank: int, shard: bool = False, world: int = 1):
    if name not in WORKLOADS:
        raise SystemExit(f"unknown workload {name}; choose from {sorted(WORKLOADS)}")
    if shard:
        # strong scaling, as BASELINE configs[4] words it: the global batch of 32768 split over the ranks
        with torch.device(device):
            m = make_model(name, True)
        if name == "ffm":  # configs[3]: global batch 131072, 1e6-row id columns
            from deeplearningrecommendationsystem_amd import synth
            gen = synth.generator(1234 + rank)
            per = 131072 // world
            return (m.to(device), [synth.feature_batch(per, 1_000_000, 1_000_000, gen).to(device)],
                    synth.labels(per, True, gen).to(device),
                    f"ffm k=32 users=items=1e6 global batch 131072 (BASELINE configs[3]), id tables row-sharded over {world} rank(s)")
        inputs, y = make_inputs(name, rank, batch_of(name) // world)
        return m.to(device), [t.to(device) for t in inputs], y.to(device), WORKLOADS[name].replace(
            "single GPU", f"item table row-sharded over {world} rank(s), all-to-all lookup")
    with torch.device(device):  # parameters are created on the GPU (the 1e6..1e7-row tables take seconds on the host)
        m = make_model(name)
    inputs, y = make_inputs(name, rank, batch_of(name))
    return m.to(device), [t.to(device) for t in inputs], y.to(device), WORKLOADS[name]


class _FeedGradient(torch.autograd.Function):
    """"loss" of the gather workloads: a zero scalar whose backward hands `grad` to the stage output"""

    @staticmethod
    def forward(ctx, out, grad):
        ctx.save_for_backward(grad)
        return out.new_zeros(())

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0], None


ORACLE_KEY = {"deepfm26": "deepfm_fields", "pnn26": "pnn_fields", "neuralcf_script": "neuralcf"}
# workloads whose full batch takes the CPU minutes per step: the baseline runs a slice of the same batch
CPU_SAMPLE_BATCH = {"din": 2048, "dien": 2048, "deepfm26": 16384, "pnn26": 16384, "deepfm": 16384}


def cpu_baseline(name: str, model, budget_s: float = 15.0):
    """time the CPU oracle on the same workload (bounded sample), rank 0 only"""
    from oracle import ctr_oracle as orc  # checker/baseline only, never on the product path
    threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU
    torch.set_num_threads(threads)
    full = batch_of(name)
    batch = CPU_SAMPLE_BATCH.get(name, full)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    inputs, y = make_inputs(name, 0, full)
    inputs, y = [t[:batch] for t in inputs], y[:batch]
    name = ORACLE_KEY.get(name, name)
    orc.step(name, params, inputs, y)  # warm-up
    t0 = time.perf_counter()
    orc.step(name, params, inputs, y)
    one = time.perf_counter() - t0
    reps = max(2, min(200, int(budget_s / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.step(name, params, inputs, y)
    dt = (time.perf_counter() - t0) / reps
    return {"value": batch / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{reps} fwd+bwd steps of batch {batch}" + ("" if batch == full else f" (the first {batch} samples of "
                      f"the {full}-sample workload batch; the dense (V,E) gradients are built in full)") +
                      " through oracle/ctr_oracle.py (torch CPU, fp32)",
            "ms_per_step": dt * 1e3}


def torch_gpu_baseline(name: str, model, device, steps: int = 20):
    """the same step through plain PyTorch ops on the SAME GPU: oracle/ctr_oracle.py is an op-by-op
    restatement of the reference's modules (nn.Embedding / matmul / cat / Linear ...), so run on
    cuda it is what the reference itself executes on this card (eager, hipBLASLt GEMMs, torch's
    embedding kernels).  Reported next to the CPU baseline; never part of `value`."""
    from oracle import ctr_oracle as orc  # checker/baseline only, never on the product path
    batch = batch_of(name)
    params = {k: v.detach().to(device).clone() for k, v in model.state_dict().items()}
    inputs, y = make_inputs(name, 0, batch)
    inputs, y = [t.to(device) for t in inputs], y.to(device)
    name = ORACLE_KEY.get(name, name)
    for _ in range(3):
        orc.step(name, params, inputs, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(name, params, inputs, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": batch / dt, "unit": "samples/s", "ms_per_step": dt * 1e3, "kind": "port",
            "what": f"{steps} eager fwd+bwd steps of batch {batch} through oracle/ctr_oracle.py on cuda (plain torch ops)"}


def zipf_ids(batch: int, fields: int, vocab: int, gen) -> torch.Tensor:
    """rank r of a Zipf(~1) law over the rows by inverse CDF on a log grid: P(rank <= r) = log r / log V"""
    u = torch.rand(batch, fields, generator=gen, dtype=torch.float64)
    return (float(vocab) ** u - 1.0).long().clamp_(0, vocab - 1)


# dev/gather_ceiling.hip (profiles/r02_gather_ceiling_microbench.txt): random 64-B and 128-B rows of an
# HBM-resident table are read at the SAME row rate (46.7 G rows/s = 6.0 TB/s of 128-B sectors), i.e. DRAM
# serves 128 B per random request and a 64-B row uses half of it; reads and writes share the HBM bus
# (copy time = read time + write time).  So one launch moves, on the DRAM side:
DRAM_SECTOR = 128


def gather_stage_leg(device, reps: int = 30):
    """metric (ii): the fused embedding-stage kernels at the SURVEY 8(d) cfg3b roofline shape -- 26 id fields x
    1e6 rows x E=16 (1.66 GB of tables, HBM-resident), batch 65536 -- timed in isolation: `reps` back-to-back
    launches between two HIP events on the launch stream (torch's current stream), after 5 warm-up launches."""
    from deeplearningrecommendationsystem_amd import _lib, ops
    from deeplearningrecommendationsystem_amd._lib import FIELD_ID_I64
    F, V, E, B = 26, 1_000_000, 16, BATCH
    gen = torch.Generator().manual_seed(1234)
    std = (2.0 / (V + E)) ** 0.5  # xavier_normal_ of a (V, E) table
    tables = [torch.randn(V, E, device=device) * std for _ in range(F)]
    grads = {id(t): torch.zeros_like(t) for t in tables}
    out = torch.empty(B, F * E, device=device)
    gout = torch.randn(B, F * E, device=device)
    ws = torch.empty(ops.SCRATCH_FLOATS, dtype=torch.float32, device=device)
    lib, st = _lib.load(), _lib.stream_ptr()
    fwd_bytes = B * F * (E * 4 + 8 + E * 4)          # rows + int64 ids + output (3536 B/sample)
    bwd_bytes = B * F * (E * 4 + 8 + 2 * E * 4)      # gout + ids + read-modify-write of each touched grad row
    dram_fwd = B * F * (DRAM_SECTOR + 8 + E * 4)     # what DRAM moves for it at 128 B per random row

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    res = {}
    for dist_name in ("uniform", "zipf"):
        idx = (torch.randint(0, V, (B, F), generator=gen) if dist_name == "uniform" else zipf_ids(B, F, V, gen)).to(device)
        specs = [ops.FieldSpec(FIELD_ID_I64, E, f * E, table=tables[f], idx=idx[:, f], idx_stride=F) for f in range(F)]
        arr_f = ops._field_array(specs)
        arr_b = ops._field_array(specs, grads)

        def fwd():
            _lib.check(lib.ctr_embed_fwd(arr_f, F, None, 0, B, out.data_ptr(), F * E, None, st), "ctr_embed_fwd")

        def bwd():
            _lib.check(lib.ctr_embed_bwd(arr_b, F, None, 0, B, gout.data_ptr(), F * E, ws.data_ptr(), ws.numel(), st),
                       "ctr_embed_bwd")

        t_f, t_b = timed(fwd), timed(bwd)
        # parity on the spot: bit-exact rows (the last forward's output is still in `out`)
        ref = torch.stack([tables[f][idx[:, f]] for f in range(F)], 1).view(B, F * E)
        assert torch.equal(out, ref), "gather26 forward is not bit-exact"
        res[dist_name] = {"fwd_us": t_f, "bwd_us": t_b, "fwd_gbs": fwd_bytes / t_f / 1e3, "bwd_gbs": bwd_bytes / t_b / 1e3}
    # the same forward launch where a training step has it: right behind the zero fill of the step's dense gradients
    # (1.66 GB of dirty lines still draining) -- event pair around the launch alone, the fill outside
    in_step = []
    gl = list(grads.values())
    for rep in range(8):
        torch._foreach_zero_(gl)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fwd()
        b.record()
        bwd()
        torch.cuda.synchronize()
        if rep >= 2:
            in_step.append(a.elapsed_time(b) * 1e3)
    in_step_us = sum(in_step) / len(in_step)
    u, z = res["uniform"], res["zipf"]
    return {
        "kernel": "embed_ids_fast_kernel", "bound": "hbm", "achieved": u["fwd_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": u["fwd_gbs"] / HBM_PEAK_GBS, "traffic": pmc_traffic("gather26", "embed_fwd_fast"), "avg_us": u["fwd_us"],
        "algorithmic_bytes": fwd_bytes,
        "workload": "SURVEY 8(d) cfg3b: 26 id fields x 1e6 rows x emb=16 (1.66 GB of tables), batch 65536, uniform ids, "
                    f"{reps} back-to-back launches",
        "dram_side": {"bytes": dram_fwd, "gbs": dram_fwd / u["fwd_us"] / 1e3, "frac_of_peak": dram_fwd / u["fwd_us"] / 1e3 / HBM_PEAK_GBS,
                      "note": "random rows cost one 128-B DRAM sector each, whether 64 or 128 B are used "
                              "(profiles/r02_gather_ceiling_microbench.txt): this is the rate the memory system sees"},
        "zipf": {"avg_us": z["fwd_us"], "achieved": z["fwd_gbs"], "frac": z["fwd_gbs"] / HBM_PEAK_GBS,
                 "what": "same launch, ids ~ Zipf(1): hot rows are served by L2 / Infinity Cache"},
        "in_step": {"avg_us": in_step_us, "achieved": fwd_bytes / in_step_us / 1e3, "frac": fwd_bytes / in_step_us / 1e3 / HBM_PEAK_GBS,
                    "what": "the same launch (Zipf ids) inside a training step: issued right behind the zero fill of the 1.66 GB of "
                            "dense gradients, whose dirty lines are still draining; one event pair per launch, 6 steps"},
        "scatter_bwd": {"kernel": "embed_ids_fast_bwd_kernel", "bound": "atomic", "algorithmic_bytes": bwd_bytes,
                        "uniform": {"avg_us": u["bwd_us"], "achieved": u["bwd_gbs"], "frac_of_hbm": u["bwd_gbs"] / HBM_PEAK_GBS,
                                    "added_gbs": B * F * E * 4 / u["bwd_us"] / 1e3},
                        "zipf": {"avg_us": z["bwd_us"], "achieved": z["bwd_gbs"], "added_gbs": B * F * E * 4 / z["bwd_us"] / 1e3},
                        "atomic_peak_gbs": 1300.0,
                        "note": "dense-grad scatter: fp32 atomics, ceiling ~1.3 TB/s of added bytes (MI355X_MICROARCH.md)"},
    }


# kernel label (ops.py) -> kernel name in the rocprofv3 / PMC summaries under profiles/
# label of an event-bracketed call -> the kernel name prefixes it may appear under in the PMC table (first match wins)
KERNEL_NAMES = {"ncfp_prep": ("ncfp_prep_kernel",), "ncfp_fwd": ("ncfp_fwd_kernel",), "ncfp_bwd": ("ncfp_bwd_kernel",),
                "ncfp_segsum": ("ncfp_segsum_kernel",), "ncfp_finish": ("ncfp_finish_kernel",),
                "embed_fwd_fast": ("embed_ids_fast_kernel",), "mlp_fused_bwd": ("ncf16_bwd_kernel", "mlp_bwd_kernel"),
                "mlp_fused_fwd": ("ncf16_fwd_kernel", "mlp_fwd_direct_kernel", "mlp_fwd_kernel"),
                "embed_mlp_fused_fwd": ("ncf16_fwd_kernel",),
                "embed_fwd": ("embed_rows_fast_kernel", "embed_fwd_kernel"),
                "embed_bwd": ("seg_reduce_kernel", "embed_bwd_kernel"), "mf_fwd": ("mf_fwd_kernel",),
                "mf_bwd": ("mf_bwd_kernel",)}


# what one event-bracketed C-ABI call covers when it is more than one kernel
LABEL_NOTES = {
    "ncfp_fwd": "ncfp_fwd_kernel: ids -> rank atomics -> four cache-resident rows per sample by LDS-DMA -> 64-32-16-8 tower + "
                "folded head (csrc/ncf_proj.hip); the first tower layer is a sum of two PROJECTED table rows (ncfp_prep)",
    "ncfp_bwd": "ncfp_bwd_kernel: head + tower backward per sample; the one 64-float row a sample owes the first layer is "
                "stored into its user's and its item's bucket (csrc/ncf_proj.hip)",
    "ncfp_segsum": "ncfp_segsum_kernel: streaming sums over the buckets (+ the tower's slab reduction and the head fold's "
                   "chain rule on workgroups of the same launch)",
    "embed_mlp_fused_fwd": "one ctr_embed_mlp_head_fwd call = ncf16_fwd_kernel<true>: ids -> table rows -> tower -> folded head",
    "mlp_fused_bwd": "one ctr_embed_mlp_head_bwd / ctr_mlp_head_bwd call = ncf16_bwd_kernel (mlp_bwd_kernel for other stacks) + "
                     "reduce_segments(_fold)_kernel (with the head fold's backward) + the gap between them; "
                     "rocprofv3 lists them separately in profiles/*_kernel_stats.csv",
    "embed_bwd": "one ctr_embed_bwd call = sort_count/colscan/scatter + seg_reduce (small tables) and/or "
                 "bag_bwd + reduce_segments and/or embed_bwd_kernel",
}


def pmc_traffic(workload, label):
    """HBM bytes per launch of `label` from the newest committed PMC pass (dev/pmc_traffic.sh, FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 passes).  FETCH_SIZE is doubled for the gfx950 wide-read
    under-count only when that keeps the total <= 2x the raw reading; None when no pass is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_traffic.json")))
    names = KERNEL_NAMES.get(label)
    if not files or names is None:
        return None
    try:
        table = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    for name in names:
        for k, v in table.items():
            if k.startswith(name):
                return {"hbm_bytes_raw": v["hbm_bytes_raw"], "hbm_bytes_fetch_x2": v["hbm_bytes_fetchx2"],
                        "kernel": k, "source": os.path.relpath(files[-1], ROOT)}
    return None


def rocprof_kernel(workload, label, flops, nbytes):
    """the dominant KERNEL of a multi-kernel call alone, from the newest committed rocprofv3 summary of this command
    (profiles/*_<workload>_kernel_stats.csv): the event pair above brackets the whole C call (kernel + its reduction
    launch), this is the cross-check the contract asks for, per kernel.  None when no summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_kernel_stats.csv")))
    names = KERNEL_NAMES.get(label)
    if not files or names is None:
        return None
    try:
        rows = list(csv.DictReader(open(files[-1])))
    except OSError:
        return None
    for name in names:
        for r in rows:
            short = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            if short.startswith(name):
                us = float(r["AverageNs"]) / 1e3
                out = {"kernel": short.split("(")[0], "avg_us": round(us, 2), "calls": int(r["Calls"]),
                       "source": os.path.relpath(files[-1], ROOT)}
                if flops:
                    out["tflops"] = round(flops / us / 1e6, 2)
                    out["frac_of_f32_mfma_peak"] = round(flops / us / 1e6 / F32_MFMA_PEAK_TF, 4)
                if nbytes:
                    out["algorithmic_gbs"] = round(nbytes / us / 1e3, 1)
                return out
    return None


L2_PEAK_GBS = 17800.0       # MI355X_MICROARCH.md, "Indexed rows: gather into LDS": rows of a table every workgroup shares,
#                             served by the XCDs' L2s: 66-73 GB/s per CU = 16.8-18.8 TB/s chip-wide (measured; the L2's
#                             streaming peak is 34.5 TB/s).  What this zoo reads from L2 are gathered table rows.


def roofline_entry(label, rec, traffic=None):
    """the roof a launch is priced against:
      mfma     its flop fraction exceeds its byte fraction
      hbm      byte-dominant and the bytes do come from DRAM
      l2       byte-dominant, but the committed PMC pass of this kernel (profiles/*_traffic.json: FETCH_SIZE x 2 +
               WRITE_SIZE) shows less than half of the algorithmic bytes on the memory side -- the operands are served by
               L2 / Infinity Cache, so the figure is priced against the guide's measured rate of row gathers out of L2
               (L2_PEAK_GBS), not called an HBM fraction; ``frac_vs_hbm`` = the same bytes against 8 TB/s, for comparison
               with earlier rounds' lines
      latency  a short launch (< 10 us) far below both roofs: launch + a few dependent round trips, neither roof binds
    ``frac`` is always achieved / peak of the named roof."""
    secs = rec["avg_us"] * 1e-6
    gbs = rec["bytes"] / secs / 1e9
    tfs = rec["flops"] / secs / 1e12
    if rec["flops"] and tfs / F32_MFMA_PEAK_TF > gbs / HBM_PEAK_GBS:
        return {"kernel": label, "bound": "mfma", "achieved": tfs, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tfs / F32_MFMA_PEAK_TF, "traffic": traffic, "avg_us": rec["avg_us"],
                "algorithmic_flops": rec["flops"]}
    dram = traffic["hbm_bytes_fetch_x2"] if traffic else None
    if dram is None and gbs > HBM_PEAK_GBS:
        # more than the HBM peak can only have come out of the caches (no PMC pass committed for this workload)
        return {"kernel": label, "bound": "l2", "achieved": gbs, "peak": L2_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / L2_PEAK_GBS, "frac_vs_hbm": gbs / HBM_PEAK_GBS, "traffic": None, "avg_us": rec["avg_us"],
                "algorithmic_bytes": rec["bytes"],
                "note": "cache-resident operands: the algorithmic rate exceeds the HBM peak"}
    if dram is not None and rec["bytes"] and dram < 0.5 * rec["bytes"]:
        return {"kernel": label, "bound": "l2", "achieved": gbs, "peak": L2_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / L2_PEAK_GBS, "frac_vs_hbm": gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_us": rec["avg_us"],
                "algorithmic_bytes": rec["bytes"], "dram_gbs": dram / secs / 1e9,
                "note": "cache-resident operands: DRAM-side traffic (PMC) is under half of the algorithmic bytes"}
    if rec["avg_us"] < 10.0 and gbs / HBM_PEAK_GBS < 0.15:
        return {"kernel": label, "bound": "latency", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_us": rec["avg_us"], "algorithmic_bytes": rec["bytes"],
                "note": "short launch: launch + dependent round trips, neither roof binds"}
    return {"kernel": label, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_us": rec["avg_us"],
            "algorithmic_bytes": rec["bytes"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="neuralcf")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-leg", action="store_true", help="skip metric (ii): the cfg3b embedding-stage timing")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from Python instead of replaying a hipGraph")
    ap.add_argument("--sparse", action="store_true", help="opt-in sparse mode of the big tables' gradients (SURVEY 8f-3): "
                    "no dense zero-fill in backward, row-wise lazy Adam in full_step; NOT the reference's dense semantics")
    ap.add_argument("--fresh-ids", action="store_true", help="with --shard: clone the input tensors every step (a new "
                    "mini-batch object per step: the exchange plan is rebuilt every step)")
    ap.add_argument("--capacity", type=float, default=0.0, help="with --shard: capacity factor of the capacity-bounded "
                    "exchange layout (e.g. 1.25; 0 = exact layout with its one host read per fresh id tensor)")
    ap.add_argument("--shard", action="store_true", help="din / dien / ffm: row-shard the big id tables over the ranks "
                    "(all-to-all lookup, eager launches, global batch split over the ranks = strong scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path with several ranks sharing one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    device = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)

    from deeplearningrecommendationsystem_amd import ops
    from deeplearningrecommendationsystem_amd.dist import GradBucket

    from deeplearningrecommendationsystem_amd.loss import BCELoss
    from deeplearningrecommendationsystem_amd.optim import Adam

    if args.capacity:
        os.environ["CTR_SHARD_CAPACITY"] = str(args.capacity)   # dist.ShardedEmbedding: capacity-bounded exchange layout
    if args.shard:
        args.no_graph = True  # the row exchanges are RCCL collectives between launches: eager (the plan of an id tensor
        #                       -- bucketing, id exchange, the one host read -- is built once and reused every step)
        if world == 1:  # a one-rank group: the same code path with an identity exchange
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(args.backend, rank=0, world_size=1)
    model, inputs, y, desc = build_workload(args.workload, device, rank, args.shard, world)
    if args.sparse:
        model.sparse_grads(True)
        desc += " [sparse-mode table gradients: opt-in, lazy row-wise Adam]"
    per_rank = inputs[0].shape[0]
    loss_fn = BCELoss()  # drop-in for torch.nn.BCELoss(), parity-tested against it
    if args.workload.startswith("gather26"):
        loss_fn = _FeedGradient.apply  # no head on the bare embedding stage: backward starts from a fixed gradient
    bucket = GradBucket(model.parameters()) if world > 1 else None
    model.train()

    def eager_step():
        model.zero_grad(set_to_none=True)
        # --fresh-ids: a data loader hands over NEW tensors every step -> no exchange plan can be reused
        ins = [t.clone() for t in inputs] if args.fresh_ids else inputs
        prob = model(*ins)
        loss = loss_fn(prob, y)
        loss.backward()
        if bucket is not None:
            bucket.all_reduce_mean()
        return loss

    step = eager_step
    if not args.no_graph:
        # the same fwd + loss + bwd launches, captured once and replayed as one hipGraph
        from deeplearningrecommendationsystem_amd.graph import GraphedStep
        graphed = GraphedStep(model, loss_fn, inputs, y)

        def step():
            loss = graphed()
            if bucket is not None:
                bucket.all_reduce_mean()
            return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * per_rank * args.steps / elapsed

    # the whole train_loop body: the same step + optimizer.step() (Adam lr 1e-3, weight_decay 1e-5 as
    # the scripts).  Reported next to the headline, not part of `value`.
    full_ms = None
    if world == 1:
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        for _ in range(3):
            step()
            opt.step()
        barrier()
        t1 = time.perf_counter()
        nfull = max(3, args.steps // 3)
        for _ in range(nfull):
            step()
            opt.step()
        barrier()
        full_ms = (time.perf_counter() - t1) / nfull * 1e3

    # per-kernel durations: the same steps again with a HIP event pair around every launch
    prof = ops.KernelProfiler(spacer_us=60.0)  # GPU kept busy ahead of each bracketed launch: no host gap in the pair
    ops.set_profiler(prof)
    for _ in range(args.steps):
        eager_step()
    ops.set_profiler(None)
    kernels = prof.summary()
    if world > 1:
        dist.barrier()

    gather_leg = None
    if rank == 0 and world == 1 and not args.no_gather_leg and not args.shard:
        gather_leg = gather_stage_leg(device)
    if rank == 0:
        entries = {k: roofline_entry(k, v, pmc_traffic(args.workload, k)) for k, v in kernels.items()}
        dominant = max(kernels, key=lambda k: kernels[k]["total_us"])
        kernel_us = sum(v["total_us"] for v in kernels.values()) / args.steps
        out = {
            "metric": "embedding-stage samples/sec (26-field gather fwd + scatter bwd) at batch 65536"
                      if args.workload.startswith("gather26") else
                      f"CTR samples/sec fwd+bwd at global batch {world * per_rank}" if args.shard else
                      "CTR samples/sec fwd+bwd at batch 65536" if batch_of(args.workload) == BATCH else
                      f"CTR samples/sec fwd+bwd at batch {batch_of(args.workload)}",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.shard else "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "global_batch": world * per_rank,
                       "parallelism": (f"dp{world}+rowshard{world}" if args.shard else f"dp{world}") if world > 1 or args.shard
                       else "single"},
            "loss": float(loss.item()), "launch": "eager" if args.no_graph else "hipGraph replay",
            **({"exchange": {"layout": f"capacity-bounded x{args.capacity}" if args.capacity else "exact",
                             "fresh_id_tensors_every_step": bool(args.fresh_ids)}} if args.shard else {}),
            "roofline": dict(entries[dominant], note=LABEL_NOTES.get(dominant) or entries[dominant].get("note"),
                             kernel_alone_rocprofv3=rocprof_kernel(args.workload, dominant, kernels[dominant]["flops"],
                                                                   kernels[dominant]["bytes"])),
            # metric (ii) is defined on the HBM-resident cfg3b shape (gather_stage_leg); the step's own embedding
            # kernel is reported under its own key -- for NeuralCF / ml-100k vocabularies its tables are cache-resident
            "gather_roofline": gather_leg,
            "step_embed_fwd": None if "embed_fwd" not in entries else
            dict(entries["embed_fwd"],
                 note="the timed step's embedding-stage launch; ml-100k-sized tables sit in L2 (cache-resident, "
                      "not an HBM figure)" if args.workload in ("neuralcf", "mf", "pnn", "ffm", "deepcrossing") else None),
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "calls_per_step": v["calls"] / args.steps,
                            "bound": entries[k]["bound"], "frac": round(entries[k]["frac"], 4),
                            **({"shapes": v["shapes"]} if v["shapes"] > 1 else {})}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_us"])},
            "gpu_kernel_us_per_step": round(kernel_us, 1),
            "full_step": None if full_ms is None else {
                "ms_per_step": full_ms, "samples_per_s": per_rank / full_ms * 1e3,
                "what": "zero_grad + forward + BCELoss + backward + Adam(lr=1e-3, weight_decay=1e-5).step()" +
                        (" -- row-wise (lazy) on the sparse-mode tables" if args.sparse else "")},
        }
        if world == 1 and not args.no_cpu_baseline and not args.shard and not args.workload.startswith("gather26"):
            try:
                out["torch_gpu_baseline"] = torch_gpu_baseline(args.workload, model, device)
            except Exception as exc:  # the oracle is CPU test infrastructure first: report, do not fail the bench
                out["torch_gpu_baseline"] = {"error": repr(exc)[:200]}
            out["cpu_baseline"] = cpu_baseline(args.workload, model)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
