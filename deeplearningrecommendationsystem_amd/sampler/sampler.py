"""Counterpart of the reference's sampler/sampler.py:11-48 with the rejection loop on the device.

Same class, same two methods, same return values: ``negative_sampling`` -> (users, items, zeros) tensors on
``device``; ``negative_sampling2`` -> a pandas frame with columns user_id / item_id / rating.  Like the
reference, a ``Sampler`` instance ACCUMULATES: a second call appends to the negatives of the first
(sampler.py:13-14, 26-27).  The draws come from a counter-based generator seeded per call, independent of launch
geometry; the distribution is the reference's -- uniform over the items of a user that are not in
``excluded_pairs``.  ``Sampler(seed=s)`` is reproducible.  ``Sampler()`` -- what the reference's scripts build, three
of them for train / valid / test (scripts/neuralcf.py:27-47) -- takes a fresh stream per INSTANCE, as the reference's
draws from Python's global ``random`` do: two default-constructed samplers never share their negatives (equal
streams would leak the evaluation negatives into training).  The streams follow ``torch.initial_seed()``, so a
script that calls ``torch.manual_seed`` first is still reproducible as a whole."""
from __future__ import annotations

import itertools

import numpy as np
import torch

from .. import _lib

_INSTANCES = itertools.count(1)   # process-wide: one stream per default-constructed Sampler


def excluded_bitmap(num_user: int, num_item: int, excluded_pairs, device) -> torch.Tensor:
    """(num_user, ceil(num_item/32)) int32 bitmap of the observed pairs -- ``excluded_pairs`` is the reference's
    set of (user, item) tuples, or an (n, 2) array / tensor"""
    if isinstance(excluded_pairs, torch.Tensor):
        pairs = excluded_pairs.detach().cpu().numpy().astype(np.int64).reshape(-1, 2)
    elif isinstance(excluded_pairs, np.ndarray):
        pairs = excluded_pairs.astype(np.int64).reshape(-1, 2)
    else:
        pairs = np.fromiter((v for p in excluded_pairs for v in p), dtype=np.int64, count=2 * len(excluded_pairs)).reshape(-1, 2)
    words = (num_item + 31) // 32
    bits = np.zeros((num_user, words), dtype=np.uint32)
    if pairs.size:
        ok = (pairs[:, 0] >= 0) & (pairs[:, 0] < num_user) & (pairs[:, 1] >= 0) & (pairs[:, 1] < num_item)
        u, i = pairs[ok, 0], pairs[ok, 1]
        np.bitwise_or.at(bits, (u, i >> 5), (np.uint32(1) << (i & 31).astype(np.uint32)))
    return torch.from_numpy(bits.view(np.int32)).to(device)


class Sampler:
    def __init__(self, seed=None):
        self.negative_users = []
        self.negative_items = []
        if seed is None:
            # (initial_seed, instance number) -> a 64-bit stream id; the odd multipliers keep distinct pairs distinct
            seed = (torch.initial_seed() * 0xD1342543DE82EF95 + next(_INSTANCES) * 0xA0761D6478BD642F) & 0xFFFFFFFFFFFFFFFF
        self._seed, self._calls = int(seed), 0

    def _draw(self, num_user, num_item, excluded_pairs, num_negatives, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.CtrHipError("the device-side sampler needs a HIP device (no CPU fallback)")
        bitmap = excluded_pairs if isinstance(excluded_pairs, torch.Tensor) and excluded_pairs.dtype == torch.int32 \
            and excluded_pairs.dim() == 2 and excluded_pairs.shape[0] == num_user else \
            excluded_bitmap(num_user, num_item, excluded_pairs, dev)
        n = num_user * num_negatives
        users = torch.empty(n, dtype=torch.int64, device=dev)
        items = torch.empty(n, dtype=torch.int64, device=dev)
        fail = torch.zeros(1, dtype=torch.int32, device=dev)
        seed = (self._seed * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF
        self._calls += 1
        rc = _lib.load().ctr_negative_sample(bitmap.data_ptr(), bitmap.shape[1], num_user, num_item, num_negatives, seed,
                                             users.data_ptr(), items.data_ptr(), fail.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "ctr_negative_sample")
        if int(fail.item()):
            raise RuntimeError("negative sampling: a user has (practically) no item outside excluded_pairs")
        return users, items

    def negative_sampling(self, num_user: int, num_item: int, excluded_pairs, num_negatives: int, device: str = 'cuda'):
        """sampler/sampler.py:16-30 -> (negative users, negative items, zero ratings) on ``device``"""
        users, items = self._draw(num_user, num_item, excluded_pairs, num_negatives, device)
        self.negative_users.append(users)
        self.negative_items.append(items)
        all_u, all_i = torch.cat(self.negative_users), torch.cat(self.negative_items)
        return all_u, all_i, torch.zeros(all_u.numel(), device=all_u.device)

    def negative_sampling2(self, num_user: int, num_item: int, excluded_pairs, num_negatives: int, device: str = 'cuda'):
        """sampler/sampler.py:32-48 -> DataFrame(user_id, item_id, rating = 0)"""
        import pandas as pd
        all_u, all_i, _ = self.negative_sampling(num_user, num_item, excluded_pairs, num_negatives, device)
        return pd.DataFrame({'user_id': all_u.cpu().numpy(), 'item_id': all_i.cpu().numpy(),
                             'rating': np.zeros(all_u.numel(), dtype=np.int64)})
