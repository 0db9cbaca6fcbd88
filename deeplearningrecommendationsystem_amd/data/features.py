"""Device-side feature assembly: what ``MovieLens100K.feature(df)`` (reference data/reader.py:98-101) builds with
two pandas merges -- [user_id, item_id] joined with the user's row (age, gender one-hot, occupation one-hot) and
the item's row (19 genre flags) -- as ONE gather on the device, straight into the (B, 45) float32 matrix the
feature models read (the scripts drop the rating column before use, scripts/pnn.py:41-43)."""
from __future__ import annotations

import torch

from .. import _lib


class FeatureAssembler:
    """``FeatureAssembler(user_features (U, 24), item_features (I, 19))``: row u / i = the columns that
    ``user_data`` / ``item_data`` carry besides the id (reader.py:36-41: min-max scaled age, get_dummies of gender
    and occupation; reader.py:21-27: the genre flags).  ``from_frames`` takes the reader's two frames."""

    def __init__(self, user_features: torch.Tensor, item_features: torch.Tensor):
        _lib.require_device(user_features, item_features)
        self.user_features = user_features.float().contiguous()
        self.item_features = item_features.float().contiguous()
        self._err = torch.zeros(1, dtype=torch.int32, device=user_features.device)

    @classmethod
    def from_frames(cls, user_data, item_data, device="cuda"):
        """``user_data`` / ``item_data``: the frames of reader.py (first column the id, rows in any order)"""
        import numpy as np
        u = user_data.sort_values(user_data.columns[0]).iloc[:, 1:].to_numpy(dtype=np.float32)
        i = item_data.sort_values(item_data.columns[0]).iloc[:, 1:].to_numpy(dtype=np.float32)
        return cls(torch.from_numpy(u).to(device), torch.from_numpy(i).to(device))

    @property
    def width(self) -> int:
        return 2 + self.user_features.shape[1] + self.item_features.shape[1]

    def feature(self, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
        """(B,) int64 ids -> (B, 2 + 24 + 19) float32: [user_id, item_id, user columns, item columns]"""
        _lib.require_device(users, items)
        if users.dtype != torch.int64 or items.dtype != torch.int64 or users.shape != items.shape or users.dim() != 1:
            raise ValueError("feature(): users and items must be 1-D int64 tensors of one length")
        users, items = users.contiguous(), items.contiguous()
        out = torch.empty((users.numel(), self.width), dtype=torch.float32, device=users.device)
        rc = _lib.load().ctr_assemble_features(users.data_ptr(), items.data_ptr(), users.numel(),
                                               self.user_features.data_ptr(), self.user_features.shape[1],
                                               self.user_features.shape[0], self.item_features.data_ptr(),
                                               self.item_features.shape[1], self.item_features.shape[0], out.data_ptr(),
                                               out.stride(0), self._err.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "ctr_assemble_features")
        return out

    def check_bad_index(self):
        if int(self._err.item()):
            self._err.zero_()
            raise IndexError("index out of range in self")
