"""Shared helpers for the parity tests (fixture loading, tolerances)."""
import os

import numpy as np
import torch

from poweflownet_amd.data import Data

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: 1e-5 relative, fp32.  "Relative" is taken against the tensor's scale
# (max |ref|), plus the same factor elementwise, i.e. |a-b| <= RTOL * (|b| + max|b|).
RTOL = 1e-5


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind in "fiub" and z[k].dtype != np.bool_ else z[k]
            for k in z.files}


def data_from(fx, prefix="", device=None):
    d = Data(**{k: fx[prefix + k] for k in ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr")})
    if prefix + "batch" in fx:
        d.batch = fx[prefix + "batch"]
    else:
        d.batch = torch.zeros(d.x.shape[0], dtype=torch.long)
    return d.to(device) if device is not None else d


def params_from(fx, prefix="param."):
    return {k[len(prefix):]: v for k, v in fx.items() if k.startswith(prefix)}


def assert_close(a, b, rtol=RTOL, what=""):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if b.numel() == 0:
        return
    scale = b.abs().max().item()
    err = (a - b).abs()
    bound = rtol * (b.abs() + scale) + 1e-30
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"{what}: max err {err.max().item():.3e} (scale {scale:.3e}) exceeds rtol {rtol:g} by x{worst:.2f}"
