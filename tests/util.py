"""Shared helpers for the parity tests (fixture loading, tolerances)."""
import os

import numpy as np
import torch

from poweflownet_amd.data import Data

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: 1e-5 relative, fp32.  "Relative" is normwise and FLAT: |a - b| <= RTOL * max|b| for every element
# (no extra elementwise slack).  Every comparison also records the error it achieved (`REPORT`), which conftest.py dumps
# to gpurun_out/parity_report.json at the end of a GPU session -- the bounds in the tests are set from those numbers.
RTOL = 1e-5
REPORT = {}


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


def rel_err(a, b):
    """(max |a - b|, max |b|) in float64."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if b.numel() == 0:
        return 0.0, 0.0
    return (a - b).abs().max().item(), b.abs().max().item()


def record(what, err, scale, bound=None):
    import os
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    REPORT.setdefault(test, []).append({"what": what, "max_abs_err": err, "scale": scale,
                                        "rel": (err / scale if scale > 0 else 0.0), "bound_rel": bound})


def assert_close(a, b, rtol=RTOL, what=""):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if b.numel() == 0:
        return
    err, scale = rel_err(a, b)
    record(what, err, scale, rtol)
    assert err <= rtol * scale + 1e-30, \
        f"{what}: max err {err:.3e} = {err / max(scale, 1e-300):.2e} of the largest entry {scale:.3e}, bound {rtol:g}"
