#!/usr/bin/env python3
"""Counterpart of the reference's scripts/pnn.py on synthetic ml-100k-shaped data: the reference's import
lines, model construction, loss and optimizer (scripts/pnn.py), the same epoch loop through the Trainer mirror.

    python scripts/pnn.py [--epochs 20] [--graph]
"""
import _common as c
import torch.nn
from torch import optim

from model.pnn import PNN
from trainer.trainer import Trainer

a = c.args()
device = c.device
splits = c.feature_splits(a.train)
model = PNN(256, [256, 128, 64, 32]).to(device)
loss_fn = torch.nn.BCELoss()
optimizer = optim.Adam(model.parameters(), lr=0.001, weight_decay=1e-5)
trainer = Trainer(model, loss_fn, optimizer, graph=a.graph)
c.run(trainer, splits, a.epochs)
