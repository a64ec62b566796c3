"""`torch.ops.pfn.*`: the hot path registered as PyTorch operators (TORCH_LIBRARY in csrc/torch_ops.cpp) above the C ABI of
include/pfn_hip.h -- what a C++ / TorchScript / `torch.ops` caller of utils/training.py:58's `model(data)` binds.

    from poweflownet_amd import torch_ops
    ops = torch_ops.load()                       # == torch.ops.pfn
    g = ops.graph_build(edge_index, num_nodes, -1)
    out = ops.mpn(g, E, 0, [4, 2, 4, 129, 4, 3], 0.2, False, params, x, pred_mask, edge_attr, None)      # differentiable
    out, ws = ops.mpn_forward(g, E, 0, [4, 2, 4, 129, 4, 3], 0.2, False, False, params, x, pred_mask, edge_attr, None)

The operators validate with TORCH_CHECK (RuntimeError), allocate through ATen, run on torch's current stream and call exactly
one C-ABI entry point each; the nn.Module classes of this package bind the SAME C ABI with ctypes (`_lib.py`), and
tests/test_torch_ops.py holds the two bindings bit-identical.  HIP tensors only: there is no CPU kernel behind any of them.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpfn_torch.so")

OPS = ("abi_version", "graph_build", "graph_check", "graph_segments", "mpn", "mpn_forward", "mpn_backward", "mpn_mse_tail_ok", "mpn_backward_mse", "edge_aggr", "tag_conv", "edge_aggr_forward", "edge_aggr_backward",
       "tag_conv_forward", "tag_conv_backward", "scatter_add", "mse_loss", "adamw_step_")

_loaded = False


def load():
    """Register the operators (once) and return `torch.ops.pfn`.  Fails loudly when the extension has not been built."""
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it first (make -C poweflownet_amd/csrc, or __graft_entry__.build())")
        L.load()                                  # libpfn_hip.so (also found through libpfn_torch.so's $ORIGIN rpath)
        torch.ops.load_library(LIB_PATH)
        if int(torch.ops.pfn.abi_version()) != L.ABI_VERSION:
            raise RuntimeError("libpfn_torch.so was built against another libpfn_hip.so ABI")
        _loaded = True
    return torch.ops.pfn


def model_dims(model) -> list:
    """The `dims` argument of mpn_forward / mpn_backward for a MaskEmbdMultiMPN instance (its constructor arguments in order)."""
    return [model.nfeature_dim, model.efeature_dim, model.output_dim, model.hidden_dim, model.n_gnn_layers, model.K]
