"""helpers to read tests/golden/*.npz (plain arrays written by oracle/make_golden.py)"""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    grads = {k[len("grad/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    inputs = [torch.from_numpy(z[f"in/{k}"]) for k in range(sum(1 for f in z.files if f.startswith("in/")))]
    return dict(meta=meta, params=params, grads=grads, inputs=inputs, y=torch.from_numpy(z["y"]),
                prob=torch.from_numpy(z["prob"]), loss=torch.from_numpy(z["loss"]))
