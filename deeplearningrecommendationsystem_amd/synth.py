"""Synthetic MovieLens-100k-shaped batches (host side, seeded).

The reference builds its inputs with pandas from ml-100k (data/reader.py:14-112)
whose licence forbids redistribution, and nothing in it is seeded, so benchmarks
and tests use generated batches with the same layout:

* feature models: ``(B,45)`` float32 -- col 0 user id, col 1 item id (as floats,
  scripts/pnn.py:41-43), col 2 age in [0,1) (reader.py:38-41), 3:5 gender
  one-hot, 5:26 occupation one-hot, 26:45 genre multi-hot (mean ~1.7 ones);
* id models: two int64 vectors (scripts/mf.py:24-62);
* sequence models: ``hist (B,L)`` int64 left-padded with id 0 and a target
  ``(B,)`` (scripts/din.py:23-31).
"""
from __future__ import annotations

import torch

NUM_USERS_ML100K = 943
NUM_ITEMS_ML100K = 1682
NUM_COLS = 45


def generator(seed: int = 1234) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def feature_batch(batch: int, num_users: int = NUM_USERS_ML100K, num_items: int = NUM_ITEMS_ML100K,
                  gen: torch.Generator | None = None, zero_genre_rows: int = 0) -> torch.Tensor:
    gen = gen or generator()
    x = torch.zeros(batch, NUM_COLS, dtype=torch.float32)
    x[:, 0] = torch.randint(0, num_users, (batch,), generator=gen).float()
    x[:, 1] = torch.randint(0, num_items, (batch,), generator=gen).float()
    x[:, 2] = torch.rand(batch, generator=gen)
    gender = torch.randint(0, 2, (batch,), generator=gen)
    occ = torch.randint(0, 21, (batch,), generator=gen)
    rows = torch.arange(batch)
    x[rows, 3 + gender] = 1.0
    x[rows, 5 + occ] = 1.0
    x[:, 26:45] = (torch.rand(batch, 19, generator=gen) < 0.09).float()
    if zero_genre_rows:
        x[:zero_genre_rows, 26:45] = 0.0
    return x


def id_batch(batch: int, num_users: int = NUM_USERS_ML100K, num_items: int = NUM_ITEMS_ML100K,
             gen: torch.Generator | None = None):
    gen = gen or generator()
    u = torch.randint(0, num_users, (batch,), generator=gen)
    i = torch.randint(0, num_items, (batch,), generator=gen)
    return u, i


def hist_batch(batch: int, hist_len: int, num_items: int, gen: torch.Generator | None = None,
               pad_fraction: float = 0.25):
    """histories are left-padded with id 0 for ``pad_fraction`` of the rows
    (random pad length), like scripts/din.py:23-31."""
    gen = gen or generator()
    hist = torch.randint(0, num_items, (batch, hist_len), generator=gen)
    npad = torch.randint(0, hist_len + 1, (batch,), generator=gen)
    padded = torch.rand(batch, generator=gen) < pad_fraction
    pos = torch.arange(hist_len).unsqueeze(0)
    hist = torch.where(padded.unsqueeze(1) & (pos < npad.unsqueeze(1)), torch.zeros_like(hist), hist)
    target = torch.randint(0, num_items, (batch,), generator=gen)
    return hist, target


def labels(batch: int, shape_2d: bool = True, gen: torch.Generator | None = None) -> torch.Tensor:
    gen = gen or generator()
    y = (torch.rand(batch, generator=gen) < 0.5).float()
    return y.view(-1, 1) if shape_2d else y
