"""Counterpart of the reference's trainer/trainer.py:8-146: same constructor, same loop
methods and call convention, so a script written against the reference drives the HIP
modules unchanged.  ``graph=True`` additionally replays the training step as one hipGraph
(the reference's scripts pass the same full-batch tensors every epoch)."""
from __future__ import annotations

import torch

from ..evaluator import Evaluator


class Trainer:
    def __init__(self, model, loss_fn, optimizer, graph: bool = False):
        self.model = model
        self.loss_fn = loss_fn
        self.optimizer = optimizer
        self.train_loss = self.valid_loss = self.test_loss = None
        self.predictions_train = self.predictions_valid = self.predictions_test = None
        self.train_rating = self.valid_rating = self.test_rating = None
        self._graph = graph
        self._graphed = None
        self._graph_key = None

    def _forward(self, args):
        # trainer/trainer.py:28-35: two tensors -> model(a, b); one -> model(x)
        if len(args) not in (1, 2):
            raise ValueError("Invalid number of arguments provided to train_loop")
        return self.model(*args)

    def train_loop(self, *args, train_rating):
        """zero_grad, forward, loss, backward, optimizer.step (trainer/trainer.py:23-40)"""
        if len(args) not in (1, 2):
            raise ValueError("Invalid number of arguments provided to train_loop")
        self.model.train()
        if self._graph:
            key = tuple(t.data_ptr() for t in args) + (train_rating.data_ptr(),)
            if self._graphed is None or key != self._graph_key:
                from ..graph import GraphedStep
                self._graphed = GraphedStep(self.model, self.loss_fn, args, train_rating)
                self._graph_key = key
            self.train_loss = self._graphed()
            self.predictions_train = self._graphed.prob
        else:
            self.optimizer.zero_grad()
            self.predictions_train = self._forward(args)
            self.train_loss = self.loss_fn(self.predictions_train, train_rating)
            self.train_loss.backward()
        self.optimizer.step()
        self.train_rating = train_rating

    def _eval_loop(self, args, rating):
        self.model.eval()
        with torch.no_grad():
            pred = self._forward(args)
            loss = self.loss_fn(pred, rating)
        return pred, loss

    def valid_loop(self, *args, valid_rating):
        self.predictions_valid, self.valid_loss = self._eval_loop(args, valid_rating)
        self.valid_rating = valid_rating

    def test_loop(self, *args, test_rating):
        self.predictions_test, self.test_loss = self._eval_loop(args, test_rating)
        self.test_rating = test_rating

    # masked variants (trainer/trainer.py:81-113, used by the AutoRec scripts)
    def train_loop2(self, train_matrix, mask):
        self.model.train()
        self.optimizer.zero_grad()
        self.predictions_train = self.model(train_matrix)[mask]
        train_rating = train_matrix[mask]
        self.train_loss = self.loss_fn(self.predictions_train, train_rating)
        self.train_loss.backward()
        self.optimizer.step()
        self.train_rating = train_rating

    def valid_loop2(self, valid_matrix, mask):
        self.model.eval()
        with torch.no_grad():
            self.predictions_valid = self.model(valid_matrix)[mask]
            self.valid_rating = valid_matrix[mask]
            self.valid_loss = self.loss_fn(self.predictions_valid, self.valid_rating)

    def test_loop2(self, test_matrix, mask):
        self.model.eval()
        with torch.no_grad():
            self.predictions_test = self.model(test_matrix)[mask]
            self.test_rating = test_matrix[mask]
            self.test_loss = self.loss_fn(self.predictions_test, self.test_rating)

    def model_eval(self, epoch):
        """prints the reference's per-epoch report (trainer/trainer.py:116-146)"""
        # the report syncs on loss.item() anyway: the place to surface a bad id seen by any step since the
        # last report (the HIP modules flag it on the device instead of asserting like nn.Embedding)
        check = getattr(self.model, "check_bad_index", None)
        if check is not None:
            check()
        ev = Evaluator()
        tr = ev.eval(self.train_rating, self.predictions_train)
        va = ev.eval(self.valid_rating, self.predictions_valid)
        te = ev.eval(self.test_rating, self.predictions_test)
        names = ["Accuracy", "Precision", "Recall", "F1 Score", "ROC AUC Score"]
        lines = [f"        Epoch {epoch + 1}:",
                 f"          - Training Loss: {self.train_loss.item()}",
                 f"          - Valid Loss: {self.valid_loss.item()}",
                 f"          - Test Loss: {self.test_loss.item()}", ""]
        for k, n in enumerate(names):
            lines += [f"          - Training {n}: {tr[k]}", f"          - Valid {n}: {va[k]}",
                      f"          - Test {n}: {te[k]}", ""]
        print("\n" + "\n".join(lines))
        return tr, va, te
