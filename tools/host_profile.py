"""Host-side profile of the EAGER training step (no hipGraph): where the Python / ctypes / launch time goes.
   gpurun -- 'PYTHONPATH=$GRAFT_REPO_ROOT python tools/host_profile.py > gpurun_out/host_profile.txt'"""
import cProfile, pstats, io, time, torch
from poweflownet_amd.synth import make_batch
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.optim import FlatAdamW
from poweflownet_amd.loss import MSELoss

dev = torch.device("cuda:0")
b = make_batch("118v2", 128).to(dev)
m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to(dev).train()
opt = FlatAdamW(m, lr=1e-3)
loss_fn = MSELoss()

def step():
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(m(b), b.y)
    loss.backward(loss_fn.unit_grad(loss))
    opt.step()

for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 200:.4f} ms/step, with final sync {1e3 * (t2 - t0) / 200:.4f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue())
