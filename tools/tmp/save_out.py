import sys, torch
sys.path.insert(0, "/root/repo")
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.synth import make_batch
res = {}
for B, hub, train in ((8, 0.0, 1), (5, 0.2, 0), (64, 0.0, 1), (13, 0.2, 1)):
    torch.manual_seed(3)
    m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to("cuda:0")
    m.train() if train else m.eval()
    if train: m.seed_dropout(9)
    d = make_batch("6470rte", B, seed=4, hub_frac=hub).to("cuda:0")
    out = m(d)
    torch.nn.MSELoss()(out, d.y).backward()
    res[(B, hub, train)] = (out.detach().cpu(), m.flat_grad().cpu())
torch.save(res, sys.argv[1])
