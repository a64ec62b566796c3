"""TEST INFRASTRUCTURE ONLY -- stand-in for torch_geometric.data.{Data, InMemoryDataset}, just enough of PyG's
published behaviour for /root/reference/datasets/PowerFlowData.py to be imported UNMODIFIED by oracle/make_goldens.py:

  * Data: an attribute bag;
  * InMemoryDataset: root/raw + root/processed directories, `process()` when a processed file is missing,
    `collate(list) -> (data, slices)` = concatenation of every attribute (edge_index along its last dimension, everything
    else along dim 0, NO index offsetting -- that is the loader's job) plus cumulative slice boundaries, and
    `dataset[i]` = the i-th slice of every attribute, then `transform`.

PyG is absent from the image and unpinned by the reference: parity at this boundary is unpinned against PyG itself.
Never imported by the product.
"""
import copy
import os

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)


def _cat_dim(key):
    return -1 if "index" in key else 0


class InMemoryDataset:
    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        self.root = root
        self.transform, self.pre_transform, self.pre_filter = transform, pre_transform, pre_filter
        self.data, self.slices = None, None
        if not all(os.path.exists(p) for p in self.processed_paths):
            os.makedirs(self.processed_dir, exist_ok=True)
            self.process()

    @property
    def raw_dir(self):
        return os.path.join(self.root, "raw")

    @property
    def processed_dir(self):
        return os.path.join(self.root, "processed")

    @property
    def raw_paths(self):
        return [os.path.join(self.raw_dir, f) for f in self.raw_file_names]

    @property
    def processed_paths(self):
        return [os.path.join(self.processed_dir, f) for f in self.processed_file_names]

    @staticmethod
    def collate(data_list):
        keys = data_list[0].keys()
        data, slices = Data(), {}
        for k in keys:
            items = [d[k] for d in data_list]
            dim = _cat_dim(k)
            sizes = torch.tensor([0] + [it.shape[dim] for it in items])
            slices[k] = torch.cumsum(sizes, 0)
            setattr(data, k, torch.cat(items, dim=dim))
        return data, slices

    def len(self):
        return next(iter(self.slices.values())).shape[0] - 1

    def __len__(self):
        return self.len()

    def get(self, idx):
        out = Data()
        for k in self.data.keys():
            lo, hi = int(self.slices[k][idx]), int(self.slices[k][idx + 1])
            v = self.data[k]
            setattr(out, k, v[..., lo:hi] if _cat_dim(k) == -1 else v[lo:hi])
        return out

    def __getitem__(self, idx):
        d = copy.copy(self.get(idx))
        return d if self.transform is None else self.transform(d)
