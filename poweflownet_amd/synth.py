"""Synthetic power-grid batches with the tensor layout of `datasets/PowerFlowData.py:171-217`.

No dataset files exist on the target image (and pandapower is absent), so tests and bench.py use
topologies generated as SURVEY.md 8(d) specifies: a random spanning tree (node i attaches to a
uniform j < i, stored parent -> child) plus uniform random chords (self-loops rejected, parallel
edges allowed), one stored edge per branch; bus types / masks follow `datasets/PowerFlowData.py:71-74`.
"""
from __future__ import annotations

import numpy as np
import torch

from .data import Batch, Data

CASES = {"14": (14, 20), "118": (118, 186), "118v2": (118, 186), "6470rte": (6470, 9005)}

# bus_type -> which of (Vm, Va, P, Q) must be predicted; datasets/PowerFlowData.py:71-74
_MASK_TABLE = torch.tensor([[0, 0, 1, 1],      # slack
                            [0, 1, 0, 1],      # PV
                            [1, 1, 0, 0]],     # PQ
                           dtype=torch.long)


def make_topology(n: int, e: int, seed: int = 0, hub_frac: float = 0.0) -> torch.Tensor:
    """(2, e) int64 stored-once edge list, connected.  `hub_frac` > 0 routes that fraction of the
    chords into n/1000 hub nodes (high-degree buses, max degree >= 64 on 6470rte, for the LDS tile sizing case)."""
    if e < n - 1:
        raise ValueError("need at least n-1 edges for a connected grid")
    rng = np.random.default_rng(seed)
    src = np.empty(e, dtype=np.int64)
    dst = np.empty(e, dtype=np.int64)
    for i in range(1, n):
        src[i - 1] = rng.integers(0, i)
        dst[i - 1] = i
    hubs = rng.choice(n, size=max(1, n // 1000), replace=False) if hub_frac > 0 else None
    k = n - 1
    while k < e:
        a = int(rng.integers(0, n))
        if hubs is not None and rng.random() < hub_frac:
            b = int(hubs[rng.integers(0, len(hubs))])
        else:
            b = int(rng.integers(0, n))
        if a == b:
            continue
        src[k], dst[k] = a, b
        k += 1
    return torch.from_numpy(np.stack([src, dst]))


def make_graph(n: int, e: int, seed: int = 0, topo_seed: int = 0, hub_frac: float = 0.0,
               edge_index: torch.Tensor | None = None) -> Data:
    """One normalised sample: y ~ N(0,1); x = y * (1 - mask); edge_attr ~ N(0,1)."""
    if edge_index is None:
        edge_index = make_topology(n, e, topo_seed, hub_frac)
    g = torch.Generator().manual_seed(seed)
    bus_type = torch.full((n,), 2, dtype=torch.long)
    bus_type[::3] = 1
    bus_type[0] = 0
    pred_mask = _MASK_TABLE[bus_type]
    y = torch.randn(n, 4, generator=g)
    x = y * (1 - pred_mask).float()
    edge_attr = torch.randn(edge_index.shape[1], 2, generator=g)
    return Data(x=x, y=y, bus_type=bus_type, pred_mask=pred_mask, edge_index=edge_index, edge_attr=edge_attr)


def make_batch(case: str, batch_size: int, seed: int = 0, hub_frac: float = 0.0, first: int = 0) -> Batch:
    """`batch_size` samples of one case sharing a topology (as every sample of a reference case does),
    collated exactly like the loader would."""
    n, e = CASES[str(case)]
    topo = make_topology(n, e, 0, hub_frac)
    return Batch.from_data_list([make_graph(n, e, seed=seed * 1_000_003 + first + b, edge_index=topo)
                                 for b in range(batch_size)])


def make_dataset(case: str, num_samples: int, seed: int = 0):
    n, e = CASES[str(case)]
    topo = make_topology(n, e, 0)
    return [make_graph(n, e, seed=seed * 1_000_003 + s, edge_index=topo) for s in range(num_samples)]
