"""AutoRec -- counterpart of the reference's model/autorec.py:5-24."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..ops import ACT_SIGMOID, Layer
from ._base import CtrModule, _ModelFunction


class AutoRec(CtrModule):
    """``AutoRec(num_input, hidden_units)``; ``forward(x: (B, num_input)) -> (B, num_input)``:
    ``sigmoid(decoder(sigmoid(encoder(x))))`` -- two GEMMs with the sigmoid in their epilogues.
    Rows whose length is not a multiple of 4 floats are staged through 16-byte aligned copies."""

    def __init__(self, num_input, hidden_units):
        super().__init__()
        self.encoder = nn.Linear(num_input, hidden_units)
        self.decoder = nn.Linear(hidden_units, num_input)

    def forward(self, x):
        params = [self.encoder.weight, self.encoder.bias, self.decoder.weight, self.decoder.bias]
        self._need_device(x, params[0])
        if x.dim() != 2 or x.shape[1] != self.encoder.in_features or x.dtype != torch.float32:
            raise ValueError(f"expected a (B,{self.encoder.in_features}) float32 matrix, got {tuple(x.shape)} {x.dtype}")
        return _ModelFunction.apply(self, 1, x, *params)

    @staticmethod
    def _rows4(t):
        """the same matrix with a row stride that is a multiple of 4 floats"""
        if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
            return t
        buf = torch.empty((t.shape[0], (t.shape[1] + 3) // 4 * 4), dtype=t.dtype, device=t.device)[:, :t.shape[1]]
        buf.copy_(t)
        return buf

    def run_forward(self, inputs, params):
        (x,) = inputs
        w1, b1, w2, b2 = params
        xa = self._rows4(x)
        hidden = ops.linear_fwd(xa, self._rows4(w1.detach()), b1, ACT_SIGMOID)
        out = torch.empty((x.shape[0], (w2.shape[0] + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :w2.shape[0]]
        ops.linear_fwd(hidden, w2, b2, ACT_SIGMOID, out=out)
        return out, (xa, hidden, out)

    def run_backward(self, state, inputs, params, gout):
        xa, hidden, out = state
        w1, b1, w2, b2 = params
        zeros = ops.zero_grads(params)
        ghidden = torch.empty_like(hidden)
        ops.linear_bwd(hidden, w2, out, self._rows4(gout), ACT_SIGMOID, ghidden, zeros[id(w2)], zeros[id(b2)])
        ops.linear_bwd(xa, self._rows4(w1.detach()), hidden, ghidden, ACT_SIGMOID, None, zeros[id(w1)], zeros[id(b1)])
        return [zeros[id(p)] for p in params]

    def recommendation(self, rating_matrix, k):
        with torch.no_grad():
            return ops.topk_rows(self.forward(rating_matrix), k, dim=1).cpu().numpy()

    def i_recommendation(self, rating_matrix, k):
        with torch.no_grad():
            return ops.topk_rows(self.forward(rating_matrix), k, dim=0).cpu().numpy()
