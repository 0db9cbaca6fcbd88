"""Logistic regression -- counterpart of the reference's model/lr.py:11-37."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import xavier_normal_

from .. import ops
from ..ops import ACT_SIGMOID
from ._base import FeatureModel


class LogisticRegression(FeatureModel):
    """``LogisticRegression(num_users, num_items, num_feature)``; ``forward(x: (B,45)) -> (B,1)``:
    ``sigmoid(user(u) + item(i) + linear(x[:,2:]))`` (lr.py:24-25) -- the wide kernel of DeepFM
    (run on one all-zero vector, so that its FM term vanishes) followed by the sigmoid."""

    def __init__(self, num_users, num_items, num_feature: int):
        super().__init__()
        self.user = nn.Embedding(num_users, 1)
        self.item = nn.Embedding(num_items, 1)
        self.linear = nn.Linear(num_feature, 1, True)
        xavier_normal_(self.user.weight.data)
        xavier_normal_(self.item.weight.data)

    def _params(self):
        return [self.user.weight, self.item.weight, self.linear.weight, self.linear.bias]

    def forward(self, feature_vector):
        return self._run_model(feature_vector, self._params())

    @staticmethod
    def _unit(device):
        return torch.ones((1, 1), dtype=torch.float32, device=device)

    def run_forward(self, inputs, params):
        (x,) = inputs
        user1, item1, w, b = params
        batch, dev = x.shape[0], x.device
        none = torch.zeros((batch, 4), dtype=torch.float32, device=dev)   # the "FM vectors": one zero vector
        logit = torch.empty((batch, 1), dtype=torch.float32, device=dev)
        ops.fm_wide_fwd(none, 1, 4, x, user1, item1, w, b, logit, self._flag)
        prob = ops.linear_fwd(logit, self._unit(dev), None, ACT_SIGMOID)   # sigmoid(1 * logit)
        return prob, (none, logit, prob)

    def run_backward(self, state, inputs, params, gprob):
        (x,) = inputs
        none, logit, prob = state
        user1, item1, w, b = params
        zeros = ops.zero_grads(params)
        glogit = torch.empty_like(logit)
        ops.linear_bwd(logit, self._unit(x.device), prob, gprob, ACT_SIGMOID, glogit, None, None)
        ops.fm_wide_bwd(none, 1, 4, x, user1, item1, w, b, glogit, zeros[id(user1)], zeros[id(item1)],
                        zeros[id(w)], zeros[id(b)], None, accumulate=False)
        return [zeros[id(p)] for p in params]

    def recommendation(self, num_users, user_item, k):
        return self._rank_users(num_users, user_item, k)
