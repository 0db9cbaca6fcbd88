#!/usr/bin/env python3
"""Counterpart of the reference's scripts/dien.py on synthetic ml-100k-shaped data: the reference's import
lines, model construction, loss and optimizer (scripts/dien.py), the same epoch loop through the Trainer mirror.

    python scripts/dien.py [--epochs 20] [--graph]
"""
import _common as c
import torch.nn
from torch import optim

from model.dien import DIEN
from trainer.trainer import Trainer

a = c.args()
device = c.device
splits = c.sequence_splits(min(a.train, 60_000), 10)
model = DIEN(c.NUM_ITEMS, 16).to(device)
loss_fn = torch.nn.BCELoss()
optimizer = optim.Adam(model.parameters(), lr=0.001, weight_decay=1e-5)
trainer = Trainer(model, loss_fn, optimizer, graph=a.graph)
c.run(trainer, splits, a.epochs)
