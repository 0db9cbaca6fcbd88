#!/usr/bin/env python3
"""Counterpart of the reference's scripts/neuralcf.py on synthetic ml-100k-shaped data: the reference's import
lines, model construction, loss and optimizer (scripts/neuralcf.py:60-66), the same epoch loop through the
Trainer mirror.

    python scripts/neuralcf.py [--epochs 20] [--graph]
"""
import _common as c
import torch.nn
from torch import optim

from model.neuralcf import NeuralCF
from trainer.trainer import Trainer

a = c.args()
device = c.device
splits = c.id_splits(a.train)
model = NeuralCF(c.NUM_USERS, c.NUM_ITEMS, 256, [512, 256, 128, 64, 32]).to(device)
loss_fn = torch.nn.BCELoss()
optimizer = optim.Adam(model.parameters(), lr=0.001, weight_decay=1e-5)
trainer = Trainer(model, loss_fn, optimizer, graph=a.graph)
c.run(trainer, splits, a.epochs)
