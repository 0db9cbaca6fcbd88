"""shared host-side pieces of the model mirrors"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import _lib


class CtrModule(nn.Module):
    """nn.Module whose forward/backward are libctrhip kernels.

    Index validation: kernels never fault on a bad id (the row is read as row 0)
    and raise a device flag; with ``CTRHIP_CHECK_INDEX=1`` (or
    ``self.check_index = True``) the flag is read back after the forward and an
    ``IndexError`` is raised like ``nn.Embedding`` does on CPU (costs a sync).
    """

    check_index = os.environ.get("CTRHIP_CHECK_INDEX", "0") == "1"

    def _err_flag(self, device):
        flag = getattr(self, "_err", None)
        if flag is None or flag.device != device:
            flag = torch.zeros(1, dtype=torch.int32, device=device)
            object.__setattr__(self, "_err", flag)
        return flag

    def _raise_if_bad_index(self):
        if self.check_index and getattr(self, "_err", None) is not None:
            if int(self._err.item()) != 0:
                self._err.zero_()
                raise IndexError("index out of range in self")

    @staticmethod
    def _need_device(*tensors):
        _lib.require_device(*tensors)


def topk_rows(scores: torch.Tensor, k: int) -> np.ndarray:
    return torch.topk(scores, k, dim=-1).indices.cpu().numpy()
