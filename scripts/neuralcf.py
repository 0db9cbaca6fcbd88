#!/usr/bin/env python3
"""Counterpart of the reference's scripts/neuralcf.py on synthetic ml-100k-shaped data
(the dataset's licence forbids shipping it): same model construction, loss, optimizer
and epoch loop (reference scripts/neuralcf.py:60-71), driven through the Trainer mirror.

    python scripts/neuralcf.py [--epochs 20] [--graph]
"""
import argparse
import os
import sys

import torch
from torch import optim

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplearningrecommendationsystem_amd import synth  # noqa: E402
from deeplearningrecommendationsystem_amd.model.neuralcf import NeuralCF  # noqa: E402
from deeplearningrecommendationsystem_amd.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    device = 'cuda' if torch.cuda.is_available() else 'cpu'   # as the reference; the HIP modules need 'cuda'
    num_users, num_items = synth.NUM_USERS_ML100K, synth.NUM_ITEMS_ML100K

    def split(n, seed):
        g = synth.generator(seed)
        u, i = synth.id_batch(n, gen=g)
        # a learnable synthetic target: users and items with matching parity interact
        y = (((u + i) % 2 == 0).float() * 0.8 + 0.1 > torch.rand(n, generator=g)).float().view(-1, 1)
        return u.to(device), i.to(device), y.to(device)

    train, valid, test = split(229_000, 1), split(20_000, 2), split(20_000, 3)
    model = NeuralCF(num_users, num_items, 256, [512, 256, 128, 64, 32]).to(device)   # reference scripts/neuralcf.py:60
    trainer = Trainer(model, torch.nn.BCELoss(), optim.Adam(model.parameters(), lr=0.001, weight_decay=1e-5),
                      graph=args.graph)
    for epoch in range(args.epochs):
        trainer.train_loop(train[0], train[1], train_rating=train[2])
        trainer.valid_loop(valid[0], valid[1], valid_rating=valid[2])
        trainer.test_loop(test[0], test[1], test_rating=test[2])
        if epoch % 5 == 4 or epoch == args.epochs - 1:
            trainer.model_eval(epoch)


if __name__ == "__main__":
    main()
