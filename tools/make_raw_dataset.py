#!/usr/bin/env python3
"""Write a synthetic dataset in the reference's RAW file format (datasets/PowerFlowData.py:61-64,171-205):
<root>/raw/case<case>_node_features.npy (S, n, 6) [index, type, Vm, Va, P, Q] and case<case>_edge_features.npy (S, e, 4)
[from, to, r, x] -- so that train.py / PowerFlowData can be exercised without the real files (no pandapower here).

    python tools/make_raw_dataset.py --root /tmp/pfdata --case 118v2 --samples 2000
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poweflownet_amd.synth import CASES, make_topology  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--root", default="data")
ap.add_argument("--case", default="118v2")
ap.add_argument("--samples", type=int, default=1000)
a = ap.parse_args()
n, e = CASES[a.case]
rng = np.random.default_rng(0)
ei = make_topology(n, e).numpy()
node = np.zeros((a.samples, n, 6))
node[:, :, 0] = np.arange(n)
node[:, :, 1] = np.where(np.arange(n) == 0, 0, np.where(np.arange(n) % 3 == 0, 1, 2))      # slack, every 3rd PV, rest PQ
node[:, :, 2:] = rng.normal(size=(a.samples, n, 4)) * np.array([0.05, 10.0, 50.0, 20.0]) + np.array([1.0, 0.0, 30.0, 10.0])
edge = np.zeros((a.samples, e, 4))
edge[:, :, :2] = ei.T
edge[:, :, 2:] = np.abs(rng.normal(size=(a.samples, e, 2))) * 0.1 + 0.01
os.makedirs(os.path.join(a.root, "raw"), exist_ok=True)
np.save(os.path.join(a.root, "raw", f"case{a.case}_edge_features.npy"), edge)
np.save(os.path.join(a.root, "raw", f"case{a.case}_node_features.npy"), node)
print(f"wrote {a.samples} samples of case{a.case} ({n} buses, {e} branches) under {a.root}/raw")
