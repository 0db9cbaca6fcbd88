"""Generate tests/golden/*.npz by running the REFERENCE's own classes.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Each fixture holds plain arrays: the reference module's state_dict
(``param/<name>``), the inputs (``in/<k>``), labels ``y``, the forward output
``prob``, ``torch.nn.BCELoss`` value ``loss`` and the gradient of every
parameter after ``loss.backward()`` (``grad/<name>``) -- i.e. one
``Trainer.train_loop`` body (trainer/trainer.py:30-38) without the optimizer.
Nothing of the reference's source is stored.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("CTR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REFERENCE)

from deeplearningrecommendationsystem_amd import synth  # noqa: E402


def _ref_classes():
    from model.mf import MatrixFactorization
    from model.neuralcf import NeuralCF
    from model.ffm import FFM
    from model.pnn import PNN
    from model.deepcrossing import DeepCrossing
    from model.deepfm import DeepFM
    from model.din import DIN
    from model.dien import DIEN
    from model.deepcross import DeepCross
    from model.widedeep import WideDeep
    from model.lr import LogisticRegression
    from model.nfm import NFM
    from model.afm import AFM
    from model.autorec import AutoRec
    return dict(mf=MatrixFactorization, neuralcf=NeuralCF, ffm=FFM, pnn=PNN,
                deepcrossing=DeepCrossing, deepfm=DeepFM, din=DIN, dien=DIEN,
                deepcross=DeepCross, widedeep=WideDeep, lr=LogisticRegression, nfm=NFM, afm=AFM, autorec=AutoRec)


# name -> (model key, ctor args, ctor kwargs, input builder(gen) -> (inputs, y))
def _cases():
    c = {}

    def ids(b, nu, ni, two_d):
        def f(g):
            u, i = synth.id_batch(b, nu, ni, g)
            return [u, i], synth.labels(b, two_d, g)
        return f

    def feats(b, nu=943, ni=1682, zero_genre=0):
        def f(g):
            return [synth.feature_batch(b, nu, ni, g, zero_genre_rows=zero_genre)], synth.labels(b, True, g)
        return f

    def seq(b, length, ni, pad=0.25):
        def f(g):
            h, t = synth.hist_batch(b, length, ni, g, pad_fraction=pad)
            return [h, t], synth.labels(b, True, g)
        return f

    for s in (0, 1):
        c[f"mf_s{s}"] = ("mf", (30, 40, 8), {}, ids(64, 30, 40, False), s)
        c[f"neuralcf_s{s}"] = ("neuralcf", (30, 40, 8, [16, 8, 4]), {}, ids(64, 30, 40, True), s)
        c[f"ffm_s{s}"] = ("ffm", (43, 8), {}, feats(64, zero_genre=4), s)
        c[f"pnn_s{s}"] = ("pnn", (8, [32, 16, 8]), {}, feats(64, zero_genre=4), s)
        c[f"deepcrossing_s{s}"] = ("deepcrossing", (30, 40, 8, [16, 8]), {}, feats(64, 30, 40, 4), s)
        c[f"deepfm_s{s}"] = ("deepfm", (30, 40, [32, 16, 1], 8), {}, feats(64, 30, 40, 4), s)
        c[f"din_s{s}"] = ("din", (50, 8), {}, seq(64, 10, 50), s)
        c[f"dien_s{s}"] = ("dien", (50, 8), {}, seq(64, 10, 50), s)
    # edge cases: ragged batch (not a multiple of the wave size), a single
    # sample, heavy id duplication, L = 1, a long un-truncated history
    c["mf_b37"] = ("mf", (5, 7, 12), {}, ids(37, 5, 7, False), 2)
    c["mf_b1"] = ("mf", (5, 7, 4), {}, ids(1, 5, 7, False), 3)
    c["neuralcf_b37"] = ("neuralcf", (5, 7, 4, [8, 4]), {}, ids(37, 5, 7, True), 2)
    c["ffm_b37"] = ("ffm", (43, 4), {}, feats(37), 2)
    c["pnn_b37"] = ("pnn", (4, [16, 8]), {}, feats(37), 2)
    c["pnn_outer_b8"] = ("pnn", (8, [16, 8]), {"model": "out"}, feats(8), 2)
    c["deepcrossing_b37"] = ("deepcrossing", (5, 7, 4, [8]), {}, feats(37, 5, 7), 2)
    c["deepfm_b37"] = ("deepfm", (5, 7, [16, 1], 4), {}, feats(37, 5, 7), 2)
    c["din_l1"] = ("din", (20, 4), {}, seq(37, 1, 20), 2)
    c["din_l130"] = ("din", (20, 4), {}, seq(5, 130, 20, pad=0.0), 3)
    c["din_allpad"] = ("din", (20, 8), {}, seq(16, 6, 20, pad=1.0), 4)
    c["dien_l1"] = ("dien", (20, 4), {}, seq(37, 1, 20), 2)
    c["dien_l33"] = ("dien", (20, 4), {}, seq(9, 33, 20), 3)
    # models built from the same primitives (DCN cross layers, Wide&Deep, LR)
    for s in (0, 1):
        c[f"deepcross_s{s}"] = ("deepcross", (30, 40, 3, [32, 16, 1], 8), {}, feats(64, 30, 40, 4), s)
        c[f"widedeep_s{s}"] = ("widedeep", (30, 40, [32, 16, 1], 8), {}, feats(64, 30, 40, 4), s)
        c[f"lr_s{s}"] = ("lr", (30, 40, 43), {}, feats(64, 30, 40, 4), s)
    c["deepcross_b37"] = ("deepcross", (5, 7, 2, [16, 8], 4), {}, feats(37, 5, 7), 2)
    c["widedeep_b37"] = ("widedeep", (5, 7, [16, 1], 4), {}, feats(37, 5, 7), 2)
    c["lr_b37"] = ("lr", (5, 7, 43), {}, feats(37, 5, 7), 2)
    for s in (0, 1):
        c[f"nfm_s{s}"] = ("nfm", (30, 40, [32, 16, 1], 8), {}, feats(64, 30, 40, 4), s)
    c["nfm_b37"] = ("nfm", (5, 7, [16, 1], 4), {}, feats(37, 5, 7), 2)
    for s in (0, 1):
        c[f"afm_s{s}"] = ("afm", (30, 40, 8, 4), {}, feats(64, 30, 40, 4), s)
    c["afm_b37"] = ("afm", (5, 7, 4, 8), {}, feats(37, 5, 7), 2)

    def ratings(rows, cols):
        # rows of a rating matrix as scripts/autorec.py builds it: 1 liked, 0 disliked, 0.5 unknown
        def f(g):
            x = torch.randint(0, 3, (rows, cols), generator=g).float() * 0.5
            return [x], (torch.rand(rows, cols, generator=g) < 0.5).float()
        return f

    c["autorec_s0"] = ("autorec", (50, 16), {}, ratings(64, 50), 0)
    c["autorec_b37"] = ("autorec", (21, 8), {}, ratings(37, 21), 2)
    return c


# ---------------------------------------------------------------------------------------------------------------
# recommendation(): the reference's own ranking loops (model/mf.py:28-35, model/neuralcf.py:61-72,
# model/pnn.py:133-143, model/deepfm.py:85-95, model/din.py:55-66, model/dien.py:70-81) on small seeded models.
# Stored: the state_dict, the call's arguments as arrays, the returned ids (``topk``) and the reference's scores of
# every candidate (``scores``: what its topk ranked), so that a test can tell a wrong ranking from a tie.
# ---------------------------------------------------------------------------------------------------------------
def _user_item_frame(num_users, num_items, gen):
    """the frame data/reader.py:104-112 builds: every (user, item) pair, user-major, with the 45 feature columns
    (user features constant per user, item features constant per item)"""
    import pandas as pd
    u = synth.feature_batch(num_users, 943, 1682, gen)      # row r: the features of user r (cols 2:26)
    it = synth.feature_batch(num_items, 943, 1682, gen)     # row r: the genre flags of item r (cols 26:45)
    rows = np.zeros((num_users * num_items, 45), dtype=np.float32)
    rows[:, 0] = np.repeat(np.arange(num_users), num_items)
    rows[:, 1] = np.tile(np.arange(num_items), num_users)
    rows[:, 2:26] = np.repeat(u[:, 2:26].numpy(), num_items, axis=0)
    rows[:, 26:45] = np.tile(it[:, 26:45].numpy(), (num_users, 1))
    cols = ["user_id", "item_id", "age", "F", "M"] + [f"occ{k}" for k in range(21)] + [f"genre{k}" for k in range(19)]
    return pd.DataFrame(rows, columns=cols)


def _rec_cases():
    c = {}
    c["rec_mf"] = ("mf", (12, 20, 8), "ids", dict(num_users=12, num_items=20), 0)
    c["rec_neuralcf"] = ("neuralcf", (12, 20, 8, [16, 8, 4]), "ids", dict(num_users=12, num_items=20), 1)
    c["rec_pnn"] = ("pnn", (8, [32, 16, 8]), "frame", dict(num_users=6, num_items=30, k=10), 2)
    c["rec_deepfm"] = ("deepfm", (6, 30, [32, 16, 1], 8), "frame", dict(num_users=6, num_items=30, k=10), 3)
    c["rec_din"] = ("din", (30, 8), "hist", dict(num_users=7, num_items=30, k=10), 4)
    c["rec_dien"] = ("dien", (30, 8), "hist", dict(num_users=7, num_items=30, k=10), 5)
    return c


def make_rec_fixtures(out_dir, classes):
    total = 0
    for name, (key, args, kind, kw, seed) in sorted(_rec_cases().items()):
        torch.manual_seed(seed)
        model = classes[key](*args)
        model.eval()
        gen = synth.generator(2000 + seed)
        arrays = {}
        nu, ni = kw["num_users"], kw["num_items"]
        with torch.no_grad():
            if kind == "ids":
                topk = model.recommendation(nu, ni)
                if key == "mf":   # model/mf.py:31-33 ranks the raw dot products
                    scores = torch.matmul(model.user_embeddings.weight[:nu], model.item_embeddings.weight[:ni].T)
                else:
                    scores = torch.stack([model(torch.full((ni,), u), torch.arange(ni)).view(-1) for u in range(nu)])
            elif kind == "frame":
                frame = _user_item_frame(nu, ni, gen)
                topk = model.recommendation(nu, frame, kw["k"])
                scores = torch.stack([model(torch.Tensor(frame[frame["user_id"] == u].values)).view(-1)
                                      for u in range(nu)])
                arrays["frame"] = frame.values.astype(np.float32)
                arrays["k"] = np.int64(kw["k"])
            else:
                lengths = [3, 7, 12, 5, 7, 1, 12][:nu]          # ragged, un-truncated histories (scripts/din.py:99-101)
                hist_list = [torch.randint(0, ni, (n,), generator=gen).tolist() for n in lengths]
                hist_list[1][0] = 0                              # the padding id inside a history
                topk = model.recommendation(nu, ni, hist_list, kw["k"])
                scores = torch.stack([model(torch.tensor(hist_list[u]).repeat(ni, 1), torch.arange(ni)).view(-1)
                                      for u in range(nu)])
                pad = np.full((nu, max(lengths)), -1, dtype=np.int64)
                for u, h in enumerate(hist_list):
                    pad[u, :len(h)] = h
                arrays["hist"] = pad
                arrays["hist_len"] = np.asarray(lengths, dtype=np.int64)
                arrays["k"] = np.int64(kw["k"])
        arrays["topk"] = np.asarray(topk, dtype=np.int64)
        arrays["scores"] = scores.numpy()
        arrays["num_users"], arrays["num_items"] = np.int64(nu), np.int64(ni)
        for pname, pt in model.state_dict().items():
            arrays[f"param/{pname}"] = pt.detach().numpy()
        meta = {"model": key, "args": list(args), "kind": kind, "seed": seed, "torch": torch.__version__}
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **arrays)
        total += os.path.getsize(path)
        print(f"{name:24s} topk {arrays['topk'].shape} {os.path.getsize(path) / 1024:.1f} KiB")
    return total


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    classes = _ref_classes()
    torch.set_num_threads(1)  # single-thread accumulation order for the index-add grads
    total = 0
    for name, (key, args, kwargs, build, seed) in sorted(_cases().items()):
        torch.manual_seed(seed)
        model = classes[key](*args, **kwargs)
        inputs, y = build(synth.generator(1000 + seed))
        model.train()
        prob = model(*inputs)
        loss = torch.nn.BCELoss()(prob, y)
        loss.backward()
        arrays = {"y": y.numpy(), "prob": prob.detach().numpy(), "loss": loss.detach().numpy()}
        for k, t in enumerate(inputs):
            arrays[f"in/{k}"] = t.numpy()
        for pname, pt in model.state_dict().items():
            arrays[f"param/{pname}"] = pt.detach().numpy()
        for pname, pt in model.named_parameters():
            g = pt.grad if pt.grad is not None else torch.zeros_like(pt)
            arrays[f"grad/{pname}"] = g.numpy()
        meta = {"model": key, "args": list(args), "kwargs": kwargs, "seed": seed,
                "torch": torch.__version__}
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **arrays)
        total += os.path.getsize(path)
        print(f"{name:24s} loss={loss.item():.6f} {os.path.getsize(path) / 1024:.1f} KiB")
    total += make_rec_fixtures(out_dir, classes)
    print(f"total {total / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
