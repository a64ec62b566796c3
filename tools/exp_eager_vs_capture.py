import sys, time, torch
sys.path.insert(0, '/root/repo')
from poweflownet_amd.synth import make_batch
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.optim import FlatAdamW
from poweflownet_amd.loss import MSELoss
from poweflownet_amd import dp
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
dev = torch.device("cuda:0")
b = make_batch("118v2", 128).to(dev)
m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to(dev).train()
opt = FlatAdamW(m, lr=1e-3)
lf = MSELoss()
box=[None]
def fb():
    opt.zero_grad(set_to_none=True)
    loss = lf(m(b), b.y)
    loss.backward(lf.unit_grad(loss))
    box[0]=loss
    return loss
def step():
    fb(); opt.step()
def timed(fn, n):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return 1e3*(t1-t0)/n, 1e3*(t2-t0)/n
import os
if os.environ.get("FIRST_ON_CURRENT"):
    step(); torch.cuda.synchronize()
side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("eager before capture (host, total): %.4f %.4f" % timed(step, 100))
print("eager again 200: %.4f %.4f" % timed(step, 200))
opt.zero_grad(set_to_none=True)
gs = dp.GraphedStep(fb, opt.step, m).capture()
print("replay: %.4f %.4f" % timed(gs.replay, 100))
print("eager after capture: %.4f %.4f" % timed(step, 100))
print("eager after capture 200: %.4f %.4f" % timed(step, 200))
del gs
print("eager after del graph: %.4f %.4f" % timed(step, 100))
