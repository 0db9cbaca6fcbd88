#!/usr/bin/env python3
"""Counterpart of the reference's scripts/mf.py on synthetic ml-100k-shaped data: the reference's import
lines, model construction, loss and optimizer (scripts/mf.py), the same epoch loop through the Trainer mirror.

    python scripts/mf.py [--epochs 20] [--graph]
"""
import _common as c
import torch.nn
from torch import optim

from model.mf import MatrixFactorization
from trainer.trainer import Trainer

a = c.args()
device = c.device
splits = c.id_splits(a.train, shape_2d=False)
model = MatrixFactorization(c.NUM_USERS, c.NUM_ITEMS, 64).to(device)
loss_fn = torch.nn.BCELoss()
optimizer = optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-5)
trainer = Trainer(model, loss_fn, optimizer, graph=a.graph)
c.run(trainer, splits, a.epochs)
