#!/usr/bin/env python3
"""Batch-assembly throughput of the device-resident PowerFlowData vs the per-sample host collate + copy (what PyG's
DataLoader does for the reference), same samples, batch 128.   python tools/loader_bench.py /tmp/pfdata 118v2"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poweflownet_amd.data import Batch, DataLoader
from poweflownet_amd.datasets import PowerFlowData
root, case = sys.argv[1], sys.argv[2]
host = PowerFlowData(root=root, case=case, split=[.5, .2, .3], task="train")
dev = PowerFlowData(root=root, case=case, split=[.5, .2, .3], task="train", device="cuda")
B = 128
def run_dev():
    n = 0
    for b in DataLoader(dev, batch_size=B, shuffle=True, generator=torch.Generator().manual_seed(0)):
        n += b.num_graphs
    torch.cuda.synchronize(); return n
def run_host():
    n = 0
    order = torch.randperm(len(host), generator=torch.Generator().manual_seed(0)).tolist()
    for i in range(0, len(host), B):
        b = Batch.from_data_list([host[j] for j in order[i:i + B]]).to("cuda"); n += b.num_graphs
    torch.cuda.synchronize(); return n
for name, fn in (("device-resident", run_dev), ("host collate + H2D", run_host)):
    fn(); t0 = time.perf_counter(); n = sum(fn() for _ in range(3)); dt = time.perf_counter() - t0
    print(f"{name:20s}: {n / dt:12.0f} graphs/s  ({1e3 * dt / (3 * ((len(host) + B - 1) // B)):.3f} ms per batch of {B})")
