"""Shared by the script counterparts: synthetic ml-100k-shaped splits (the dataset's licence forbids shipping
it) and the reference scripts' epoch loop (e.g. scripts/pnn.py:57-63), driven through the Trainer mirror.

The scripts import the models exactly as the reference's do (``from model.pnn import PNN``,
``from trainer.trainer import Trainer``): ``compat/`` is put first on ``sys.path``."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "compat"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from deeplearningrecommendationsystem_amd import synth  # noqa: E402

device = 'cuda' if torch.cuda.is_available() else 'cpu'   # as the reference; the HIP modules need 'cuda'
NUM_USERS, NUM_ITEMS = synth.NUM_USERS_ML100K, synth.NUM_ITEMS_ML100K


def args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the training step as one hipGraph")
    ap.add_argument("--train", type=int, default=229_000, help="training samples (positives + sampled negatives)")
    return ap.parse_args()


def _learnable(u, i, n, gen):
    # a target with structure: users and items of matching parity interact
    return (((u + i) % 2 == 0).float() * 0.8 + 0.1 > torch.rand(n, generator=gen)).float()


def id_splits(n_train, shape_2d=True):
    """(user, item, rating) x train/valid/test, as scripts/mf.py:38-52 / scripts/neuralcf.py build them"""
    out = []
    for n, seed in ((n_train, 1), (20_000, 2), (20_000, 3)):
        g = synth.generator(seed)
        u, i = synth.id_batch(n, gen=g)
        y = _learnable(u, i, n, g)
        out.append((u.to(device), i.to(device), (y.view(-1, 1) if shape_2d else y).to(device)))
    return out


def feature_splits(n_train):
    """((B,45) feature matrix, rating) x train/valid/test -- the layout of data/reader.py:98-112"""
    out = []
    for n, seed in ((n_train, 1), (20_000, 2), (20_000, 3)):
        g = synth.generator(seed)
        x = synth.feature_batch(n, gen=g)
        y = _learnable(x[:, 0].long(), x[:, 1].long(), n, g).view(-1, 1)
        out.append((x.to(device), y.to(device)))
    return out


def sequence_splits(n_train, hist_len):
    """(hist (B,L), target (B,), rating) x train/valid/test, histories left-padded with id 0 (scripts/din.py:23-31)"""
    out = []
    for n, seed in ((n_train, 1), (10_000, 2), (10_000, 3)):
        g = synth.generator(seed)
        hist, target = synth.hist_batch(n, hist_len, NUM_ITEMS, g)
        y = _learnable(hist[:, -1], target, n, g).view(-1, 1)
        out.append((hist.to(device), target.to(device), y.to(device)))
    return out


def run(trainer, splits, epochs):
    """the reference scripts' training loop: train / valid / test loops every epoch, the report every 5th"""
    train, valid, test = splits
    for epoch in range(epochs):
        trainer.train_loop(*train[:-1], train_rating=train[-1])
        trainer.valid_loop(*valid[:-1], valid_rating=valid[-1])
        trainer.test_loop(*test[:-1], test_rating=test[-1])
        if epoch % 5 == 4 or epoch == epochs - 1:
            trainer.model_eval(epoch)
