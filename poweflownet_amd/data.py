"""Minimal PyG-shaped `Data` / `Batch` / `DataLoader` surface for the hot path's callers.

torch_geometric is not available on the target image, so the boundary consumers
(`utils/training.py:55-77`, `train.py:90-92` of the reference) get a build-authored mini surface with
the collate rule PyG documents: concatenate node/edge tensors along dim 0, offset `edge_index`
by the cumulative node count, add `batch` (N,) and `ptr` (B+1,).  Real PyG objects duck-type through
`MaskEmbdMultiMPN.forward` unchanged because the model only reads attributes.

Parity note: pinned by the analytic fixture tests/golden/g7_collate.npz (PyG itself is absent).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch

_NODE_KEYS = ("x", "y", "bus_type", "pred_mask")
_EDGE_KEYS = ("edge_attr",)


class Data:
    """Attribute bag: x (n,4) f32, y (n,4) f32, bus_type (n,) i64, pred_mask (n,4) i64,
    edge_index (2,e) i64, edge_attr (e,2) f32 -- the layout `datasets/PowerFlowData.py:196-205` emits."""

    def __init__(self, **kwargs):
        self._keys: List[str] = []
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if not key.startswith("_") and key not in self.__dict__.get("_keys", []):
            self.__dict__.setdefault("_keys", []).append(key)
        object.__setattr__(self, key, value)

    def keys(self):
        return list(self._keys)

    def __len__(self):
        # PyG's Data.__len__ is the number of stored attributes; train_epoch weights the running
        # loss by it (utils/training.py:76-77), which cancels in the mean.
        return len(self._keys)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device, non_blocking: bool = False):
        out = self.__class__.__new__(self.__class__)
        out.__dict__["_keys"] = []
        for k in self._keys:
            v = getattr(self, k)
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        out.__dict__["_keys"] = []
        for k in self._keys:
            v = getattr(self, k)
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    def __repr__(self):
        parts = [f"{k}={list(getattr(self, k).shape)}" if torch.is_tensor(getattr(self, k)) else f"{k}={getattr(self, k)!r}"
                 for k in self._keys]
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Block-diagonal concatenation of `Data` objects (PyG `Batch.from_data_list` rule)."""

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data]) -> "Batch":
        if len(data_list) == 0:
            raise ValueError("empty data list")
        keys = data_list[0].keys()
        out = cls()
        counts = [d.num_nodes for d in data_list]
        offsets = [0]
        for c in counts:
            offsets.append(offsets[-1] + c)
        for k in keys:
            vals = [getattr(d, k) for d in data_list]
            if not torch.is_tensor(vals[0]):
                setattr(out, k, vals)
            elif k == "edge_index":
                setattr(out, k, torch.cat([v + off for v, off in zip(vals, offsets)], dim=1))
            else:
                setattr(out, k, torch.cat(vals, dim=0))
        dev = data_list[0].x.device
        out.batch = torch.repeat_interleave(torch.arange(len(data_list), device=dev),
                                            torch.tensor(counts, device=dev))
        out.ptr = torch.tensor(offsets, dtype=torch.long, device=dev)
        return out

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel() - 1)


class DataLoader:
    """Single-process loader (the reference uses num_workers=0, train.py:90): iterates a sequence of
    `Data`, collating `batch_size` of them per step.

    `shard=(rank, world)` (SURVEY 8e partitioning; new -- the reference is single-process): `batch_size` is then the
    GLOBAL batch and rank r takes graphs r::world of it.  Every rank must run the same number of steps with equally sized
    shards -- each step ends in one collective, and the mean of the rank means equals the global mean only for equal
    shards -- so a global batch is truncated to a multiple of `world` and a tail batch with fewer than `world` samples
    is dropped ON EVERY RANK (`len(loader)` counts exactly the batches that are yielded)."""

    def __init__(self, dataset: Sequence[Data], batch_size: int = 1, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None, shard: Optional[tuple] = None,
                 drop_last: bool = False):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.generator, self.shard, self.drop_last = generator, shard, drop_last
        if shard is not None:
            r, w = shard
            if not (0 <= r < w):
                raise ValueError(f"shard=(rank, world) needs 0 <= rank < world, got {shard}")
            if self.batch_size < w:
                raise ValueError(f"global batch_size {self.batch_size} < world size {w}")

    def _num_batches(self) -> int:
        n = len(self.dataset)
        full, tail = divmod(n, self.batch_size)
        if self.drop_last or tail == 0:
            return full
        if self.shard is not None and tail < self.shard[1]:
            return full                                        # a tail no rank could share: dropped everywhere
        return full + 1

    def __len__(self):
        return self._num_batches()

    def _index_lists(self):
        """The epoch's batches as lists of sample indices (one draw of the permutation, the shard rule applied)."""
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for b in range(self._num_batches()):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            if self.shard is not None:
                r, w = self.shard
                idx = idx[:len(idx) // w * w][r::w]            # equal shards: len(idx) // w graphs on every rank
            yield idx

    def index_batches(self, device):
        """The same batches as `__iter__` would collate -- same permutation, same shard rule -- as DEVICE int64 index tensors:
        ONE host->device copy per epoch (the whole permutation), every batch a view of it.  For a consumer that gathers the
        samples itself (utils.training.GraphedTrainStep.step_indexed with a device-resident dataset)."""
        lists = list(self._index_lists())
        flat = torch.tensor([i for l in lists for i in l], dtype=torch.long).to(device, non_blocking=True)
        off = 0
        for l in lists:
            yield flat[off:off + len(l)]
            off += len(l)

    def __iter__(self) -> Iterable[Batch]:
        for idx in self._index_lists():
            if hasattr(self.dataset, "collate_indices"):       # device-resident dataset: one gather per field, no host loop
                yield self.dataset.collate_indices(idx)
            else:
                yield Batch.from_data_list([self.dataset[i] for i in idx])
