"""`PowerFlowData` -- the dataset in front of the hot path (reference datasets/PowerFlowData.py:44-217),
kept RESIDENT on the device as dense per-split tensors instead of a pickled list of PyG `Data`.

Same constructor, constants, statistics and per-sample fields as the reference class:
  raw files   root/raw/case{case}_{edge,node}_features.npy       (S, e, 4) [from, to, r, x] / (S, n, 6) [index, type, Vm, Va, P, Q]
  split       `int(S * f)` samples per fraction, in file order (:183-187); the fractions must cover S exactly, as torch.split demands
  per sample  y = node[:, 2:], bus_type = node[:, 1], pred_mask = bus_type_mask[bus_type] (1 = predict),
              x = y * (1 - pred_mask), edge_index = edge[:, :2].T (stored once per branch), edge_attr = edge[:, 2:]   (:189-205)
  normalise   x and y with the per-feature mean / unbiased std of THIS split's y (or the ones handed in), edge_attr with its own;
              divisor std + 1e-7 (:126-139) -- masked entries of x become -mean/std, not 0.
What differs is the storage: the reference collates a Python list into one pickled (data, slices) pair and slices samples
back out one by one for PyG's DataLoader to re-collate on the host for every batch.  Here a split is a handful of dense
tensors [S, n, .] on `device`; a batch is one `index_select` per field plus a cached block-diagonal `edge_index` (the topology of
a case is the same for every sample, dataset_generator.py:250-253) -- no per-batch host work, nothing to copy host -> device.
`poweflownet_amd.data.DataLoader` takes this path automatically (`collate_indices`).

Parity: tests/golden/g9_powerflowdata.npz was produced by the reference class itself (oracle/make_goldens.py g9).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from ..data import Batch, Data


def random_bus_type(data: Data) -> Data:
    """Train-time transform of the reference (:36-40): random bus types in {0, 1}.  The model never reads `bus_type`."""
    data.bus_type = torch.randint_like(data.bus_type, low=0, high=2)
    return data


def denormalize(input, mean, std):
    """Inverse of the dataset's z-scoring (:42-43)."""
    return input * (std.to(input.device) + 1e-7) + mean.to(input.device)


class _Block:
    """All samples of one grid case in one split: dense tensors with a leading sample dimension."""
    __slots__ = ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr", "static_topology", "_ei_cache")

    def __init__(self, node: torch.Tensor, edge: torch.Tensor, table: torch.Tensor):
        self.y = node[:, :, 2:].contiguous()
        self.bus_type = node[:, :, 1].to(torch.long)
        self.pred_mask = table[self.bus_type]
        self.x = self.y * (1.0 - self.pred_mask)
        self.edge_index = edge[:, :, 0:2].transpose(1, 2).to(torch.long).contiguous()   # (S, 2, e)
        self.edge_attr = edge[:, :, 2:].contiguous()
        self.static_topology = bool(self.edge_index.shape[0] > 0 and (self.edge_index == self.edge_index[:1]).all())
        self._ei_cache = {}

    def __len__(self):
        return int(self.y.shape[0])

    def to(self, device):
        for k in ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr"):
            setattr(self, k, getattr(self, k).to(device))
        self._ei_cache = {}
        return self


class PowerFlowData:
    partial_file_names = ["edge_features.npy", "node_features.npy"]
    split_order = {"train": 0, "val": 1, "test": 2}
    mixed_cases = ["118v2", "14v2"]
    slack_mask = (0, 0, 1, 1)   # 1 = need to predict, 0 = given
    gen_mask = (0, 1, 0, 1)
    load_mask = (1, 1, 0, 0)
    bus_type_mask = (slack_mask, gen_mask, load_mask)

    def __init__(self, root: str, case: str = "14", split: Optional[List[float]] = None, task: str = "train",
                 transform: Optional[Callable] = None, pre_transform: Optional[Callable] = None,
                 pre_filter: Optional[Callable] = None, normalize=True, xymean=None, xystd=None, edgemean=None, edgestd=None,
                 device=None):
        assert split is not None and len(split) == 3
        assert task in ["train", "val", "test"]
        self.root, self.case, self.split, self.task = root, case, split, task
        self.normalize, self.transform = normalize, transform
        self.xymean, self.xystd = (xymean, xystd) if xymean is not None and xystd is not None else (None, None)
        self.edgemean, self.edgestd = (edgemean, edgestd) if edgemean is not None and edgestd is not None else (None, None)
        table = torch.tensor(self.bus_type_mask)
        self._blocks: List[_Block] = []
        self._list: Optional[List[Data]] = None
        part = self.split_order[task]
        for edge_path, node_path in self._raw_pairs():
            edge = torch.from_numpy(np.load(edge_path)).float()
            node = torch.from_numpy(np.load(node_path)).float()
            split_len = [int(len(node) * f) for f in split]
            edge_t = torch.split(edge, split_len, dim=0)[part]      # raises, like the reference, when the fractions do not cover S
            node_t = torch.split(node, split_len, dim=0)[part]
            self._blocks.append(_Block(node_t, edge_t, table))
        if pre_filter is not None or pre_transform is not None:      # arbitrary per-sample callables: the slow, list-backed path
            items = [self._sample(i) for i in range(self._dense_len())]
            if pre_filter is not None:
                items = [d for d in items if pre_filter(d)]
            if pre_transform is not None:
                items = [pre_transform(d) for d in items]
            self._list = items
        self._normalize_dataset()
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------------------------------------ files
    @property
    def raw_file_names(self) -> List[str]:
        cases = [self.case] if self.case != "mixed" else self.mixed_cases
        return [f"case{c}_{name}" for c in cases for name in self.partial_file_names]

    def _raw_pairs(self):
        paths = [os.path.join(self.root, "raw", f) for f in self.raw_file_names]
        assert len(paths) % 2 == 0
        return [(paths[i], paths[i + 1]) for i in range(0, len(paths), 2)]

    # ------------------------------------------------------------------------------------------- statistics
    def _all(self, key):
        if self._list is not None:
            return torch.cat([getattr(d, key) for d in self._list], dim=0)
        return torch.cat([getattr(b, key).reshape(-1, getattr(b, key).shape[-1]) for b in self._blocks], dim=0)

    def _normalize_dataset(self):
        if not self.normalize:
            return
        if self.xymean is None or self.xystd is None:
            xy = self._all("y")
            self.xymean, self.xystd = torch.mean(xy, dim=0, keepdim=True), torch.std(xy, dim=0, keepdim=True)
        if self.edgemean is None or self.edgestd is None:
            ea = self._all("edge_attr")
            self.edgemean, self.edgestd = torch.mean(ea, dim=0, keepdim=True), torch.std(ea, dim=0, keepdim=True)
        holders = self._list if self._list is not None else self._blocks
        for h in holders:
            dev = h.x.device
            xm, xs = self.xymean.to(dev), self.xystd.to(dev)
            em, es = self.edgemean.to(dev), self.edgestd.to(dev)
            h.x = (h.x - xm) / (xs + 0.0000001)
            h.y = (h.y - xm) / (xs + 0.0000001)
            h.edge_attr = (h.edge_attr - em) / (es + 0.0000001)

    def get_data_dimensions(self):
        d = self[0]
        return d.x.shape[1], d.y.shape[1], d.edge_attr.shape[1]

    def get_data_means_stds(self):
        assert self.normalize == True  # noqa: E712  (the reference's own guard)
        return self.xymean[:1, :], self.xystd[:1, :], self.edgemean[:1, :], self.edgestd[:1, :]

    # ---------------------------------------------------------------------------------------------- samples
    def _dense_len(self):
        return sum(len(b) for b in self._blocks)

    def len(self):
        return len(self._list) if self._list is not None else self._dense_len()

    def __len__(self):
        return self.len()

    def _locate(self, idx):
        for b in self._blocks:
            if idx < len(b):
                return b, idx
            idx -= len(b)
        raise IndexError("sample index out of range")

    def _sample(self, idx) -> Data:
        b, i = self._locate(idx)
        return Data(x=b.x[i], y=b.y[i], bus_type=b.bus_type[i], pred_mask=b.pred_mask[i], edge_index=b.edge_index[i],
                    edge_attr=b.edge_attr[i])

    def __getitem__(self, idx) -> Data:
        if idx < 0:
            idx += len(self)
        d = self._list[idx].clone() if self._list is not None else self._sample(idx)
        return d if self.transform is None else self.transform(d)

    @property
    def device(self):
        return (self._list[0].x if self._list is not None else self._blocks[0].x).device

    def to(self, device):
        """Move the whole split to `device` (in place): from then on batches are assembled there."""
        if self._list is not None:
            self._list = [d.to(device) for d in self._list]
        for b in self._blocks:
            b.to(device)
        return self

    # ---------------------------------------------------------------------------------------------- batches
    def can_gather(self) -> bool:
        """One dense device-resident block, one topology for every sample, no per-sample transform: a batch is then five row
        gathers INTO tensors that already exist (`gather_into`) -- what lets a captured training step pull its own batch."""
        return (self._list is None and len(self._blocks) == 1 and self.transform is None and self._blocks[0].static_topology
                and self._blocks[0].x.is_cuda)

    def gather_into(self, batch: Batch, idx: torch.Tensor) -> None:
        """Overwrite the sample-dependent fields of `batch` (built earlier by `collate_indices` for the same number of samples)
        with the samples `idx` (a device int64 tensor): index_select(out=...) per field, no allocation, no host work -- hipGraph-
        capturable, so `GraphedTrainStep` replays "gather the batch + train on it" as ONE graph launch (zero per-batch
        host work).  edge_index / batch / ptr do not depend on the samples (static topology) and stay as they are."""
        b = self._blocks[0]
        B, n, e = int(idx.numel()), int(b.x.shape[1]), int(b.edge_index.shape[2])
        torch.index_select(b.x, 0, idx, out=batch.x.view(B, n, -1))
        torch.index_select(b.y, 0, idx, out=batch.y.view(B, n, -1))
        torch.index_select(b.bus_type, 0, idx, out=batch.bus_type.view(B, n))
        torch.index_select(b.pred_mask, 0, idx, out=batch.pred_mask.view(B, n, -1))
        torch.index_select(b.edge_attr, 0, idx, out=batch.edge_attr.view(B, e, -1))

    def collate_indices(self, indices: Sequence[int]) -> Batch:
        """The batch PyG's collate would build from samples `indices` (cat along dim 0, edge_index offset by the cumulative
        node count, `batch`, `ptr`) -- assembled on the dataset's device with one gather per field.  Falls back to the
        per-sample rule for list-backed datasets, mixed grid cases in one batch, and sample-dependent topologies."""
        if self._list is None and len(self._blocks) == 1 and self.transform is None:
            b = self._blocks[0]
            dev = b.x.device
            idx = torch.as_tensor(list(indices), dtype=torch.long, device=dev)
            B, n, e = int(idx.numel()), int(b.x.shape[1]), int(b.edge_index.shape[2])
            out = Batch()
            out.x = b.x.index_select(0, idx).reshape(B * n, -1)
            out.y = b.y.index_select(0, idx).reshape(B * n, -1)
            out.bus_type = b.bus_type.index_select(0, idx).reshape(B * n)
            out.pred_mask = b.pred_mask.index_select(0, idx).reshape(B * n, -1)
            if b.static_topology:
                ei = b._ei_cache.get(B)
                if ei is None:
                    off = (torch.arange(B, device=dev) * n).view(B, 1, 1)
                    ei = (b.edge_index[:1] + off).permute(1, 0, 2).reshape(2, B * e).contiguous()
                    b._ei_cache[B] = ei
                out.edge_index = ei      # the same tensor object for every batch of this size: the model's topology cache hits
            else:
                off = (torch.arange(B, device=dev) * n).view(B, 1, 1)
                out.edge_index = (b.edge_index.index_select(0, idx) + off).permute(1, 0, 2).reshape(2, B * e).contiguous()
            out.edge_attr = b.edge_attr.index_select(0, idx).reshape(B * e, -1)
            out.batch = torch.arange(B, device=dev).repeat_interleave(n)
            out.ptr = torch.arange(B + 1, device=dev) * n
            return out
        return Batch.from_data_list([self[i] for i in indices])
