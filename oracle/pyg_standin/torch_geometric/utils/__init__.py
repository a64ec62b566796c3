"""Stand-in for torch_geometric.utils (only `degree`, used at networks/MPN.py:44)."""
import torch


def degree(index, num_nodes=None, dtype=None):
    """Number of occurrences of each node id in `index` (PyG: zeros(N).scatter_add_(0, index, ones))."""
    if num_nodes is None:
        num_nodes = int(index.max()) + 1 if index.numel() > 0 else 0
    out = torch.zeros(num_nodes, dtype=dtype if dtype is not None else torch.get_default_dtype(),
                      device=index.device)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype, device=index.device))


def from_scipy_sparse_matrix(*a, **k):   # importable names only (datasets/PowerFlowData.py:12), never called on the path
    raise NotImplementedError


def dense_to_sparse(*a, **k):
    raise NotImplementedError
