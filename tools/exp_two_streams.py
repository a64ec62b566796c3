#!/usr/bin/env python3
"""Experiment: does running two independent half-batch chains on two streams (captured into one hipGraph) beat one
full-batch chain at case118v2 x 128?  Upper bound for a native two-chain split of the latency-bound small-batch step."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.loss import MSELoss
from poweflownet_amd.synth import make_batch

torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nchain = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(1234)
model = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to(dev).train()
model.seed_dropout(1)
full = make_batch("118v2", B, seed=0).to(dev)
parts = [make_batch("118v2", B // nchain, seed=i).to(dev) for i in range(nchain)]
losses = [MSELoss() for _ in range(nchain + 1)]


def fb(d, lf):
    loss = lf(model(d), d.y)
    loss.backward(lf.unit_grad(loss))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


# warm-up eager
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        fb(full, losses[-1])
        for p, lf in zip(parts, losses):
            fb(p, lf)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()

model.zero_grad(set_to_none=True)
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    fb(full, losses[-1])
t1 = timeit(g1.replay)

model.zero_grad(set_to_none=True)
streams = [torch.cuda.Stream() for _ in range(nchain - 1)]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for i, (p, lf) in enumerate(zip(parts, losses)):
        if i == 0:
            fb(p, lf)
        else:
            with torch.cuda.stream(streams[i - 1]):
                fb(p, lf)
    for s in streams:
        cur.wait_stream(s)
t2 = timeit(g2.replay)

model.zero_grad(set_to_none=True)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    for p, lf in zip(parts, losses):
        fb(p, lf)
t3 = timeit(g3.replay)
print(f"B={B} chains={nchain}: one chain of {B}: {t1:.4f} ms | {nchain} concurrent chains of {B // nchain}: {t2:.4f} ms | "
      f"{nchain} chains of {B // nchain} back to back: {t3:.4f} ms")
