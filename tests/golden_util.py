"""helpers to read tests/golden/*.npz (plain arrays written by oracle/make_golden.py)"""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=""):
    """train-step fixtures; the ``rec_*`` files (recommendation() outputs) are listed by ``rec_names``"""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
                  if not n.startswith("rec_"))


def rec_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "rec_*.npz")))


def load_rec(name):
    """a recommendation() fixture: the reference's returned ids, its scores of every candidate, the call's arguments"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    out = dict(meta=json.loads(bytes(z["meta"]).decode()),
               params={k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")},
               topk=z["topk"], scores=z["scores"], num_users=int(z["num_users"]), num_items=int(z["num_items"]))
    if "k" in z.files:
        out["k"] = int(z["k"])
    if "frame" in z.files:
        out["frame"] = z["frame"]
    if "hist" in z.files:
        out["hist_list"] = [z["hist"][u, :n].tolist() for u, n in enumerate(z["hist_len"])]
    return out


def assert_same_ranking(got, want, scores, tol=1e-6):
    """``got`` / ``want``: (users, k) candidate positions, best first.  Equal, or where they differ the reference's
    scores of the two candidates are tied within ``tol`` (relative): a tie may be broken either way"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    for u in range(want.shape[0]):
        assert len(set(got[u].tolist())) == got.shape[1], f"user {u}: a candidate is ranked twice"
        for r in np.flatnonzero(got[u] != want[u]):
            a, b = float(scores[u, got[u, r]]), float(scores[u, want[u, r]])
            assert abs(a - b) <= tol * max(1.0, abs(a), abs(b)), \
                f"user {u} rank {r}: candidate {got[u, r]} (score {a}) instead of {want[u, r]} (score {b})"


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    grads = {k[len("grad/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    inputs = [torch.from_numpy(z[f"in/{k}"]) for k in range(sum(1 for f in z.files if f.startswith("in/")))]
    return dict(meta=meta, params=params, grads=grads, inputs=inputs, y=torch.from_numpy(z["y"]),
                prob=torch.from_numpy(z["prob"]), loss=torch.from_numpy(z["loss"]))
