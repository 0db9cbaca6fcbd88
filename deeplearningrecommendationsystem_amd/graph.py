"""hipGraph replay of a training step.

The zoo's steps are short chains of small kernels (a NeuralCF step is ~45 launches
of 5-70 us each): enqueueing them one by one from Python costs more host time than
the GPU needs to run them.  The reference's scripts train full-batch -- the same
input tensors every epoch (e.g. scripts/neuralcf.py:67-71) -- which is exactly the
static-shape, static-address case a captured graph needs.  ``GraphedStep`` captures
``zero_grad -> model(*inputs) -> loss_fn -> backward`` once on torch's capture stream
(libctrhip launches are ordinary stream work, so they are captured like torch's own
kernels) and replays it with one ``hipGraphLaunch`` per step.  Parameters are updated
in place by the optimizer between replays, so the graph always sees current weights;
``.grad`` tensors live in the graph's memory pool and are rewritten by every replay.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable, inputs: Sequence[torch.Tensor],
                 target: torch.Tensor, warmup: int = 3):
        self.model, self.loss_fn = model, loss_fn
        self.inputs, self.target = list(inputs), target
        model.train()
        # warm-up runs on a side stream, capture on the graph's stream: the AccumulateGrad nodes
        # are recreated per iteration, the stream-mismatch warning does not apply
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        model.zero_grad(set_to_none=True)
        # the root gradient of loss.backward() is a constant 1: made once here instead of by a fill
        # kernel inside every replay
        from .loss import unit_grad  # BCELoss recognises this tensor and skips its scaling launch
        self._one = unit_grad(target.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.prob = model(*self.inputs)
            self.loss = loss_fn(self.prob, self.target)
            self.loss.backward(self._one)

    def _eager(self):
        self.model.zero_grad(set_to_none=True)
        prob = self.model(*self.inputs)
        loss = self.loss_fn(prob, self.target)
        loss.backward()
        return loss

    def load(self, inputs: Sequence[torch.Tensor], target: torch.Tensor) -> None:
        """copy a new batch of the captured shape into the static input buffers"""
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src)
        self.target.copy_(target)

    def __call__(self) -> torch.Tensor:
        """replay; returns the (static) loss tensor, ``.grad`` of every parameter is fresh"""
        self.graph.replay()
        return self.loss
