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


def record_elementwise(a, b, what, rtol=1e-5, atol_of_max=1e-6):
    """The ELEMENTWISE reading of "1e-5 relative" next to the normwise one of assert_close: how many entries violate
    |a - b| <= rtol * |b| + atol_of_max * max|b|, and by how much at worst.  Recorded (gpurun_out/parity_report.json), returned as
    (violations, entries, worst ratio err / bound)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if b.numel() == 0:
        return 0, 0, 0.0
    bound = rtol * b.abs() + atol_of_max * b.abs().max()
    ratio = ((a - b).abs() / bound.clamp_min(1e-300))
    bad, worst = int((ratio > 1.0).sum()), float(ratio.max())
    record(f"{what}: elementwise |a-b| <= {rtol:g}|b| + {atol_of_max:g} max|b|: {bad} of {b.numel()} entries exceed it, worst {worst:.3f} x the bound",
           float((a - b).abs().max()), float(b.abs().max()), None)
    return bad, b.numel(), worst
