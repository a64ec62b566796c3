"""Whole model at config 2: fused C entry point vs our single layers chained by torch autograd vs fp32/fp64 oracle."""
import sys, copy, torch
sys.path.insert(0, '.')
from oracle import ref_cpu
from poweflownet_amd.networks import MPN
from poweflownet_amd.synth import make_batch
torch.manual_seed(1234)
torch.set_num_threads(8)
ref = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).eval()
m = MPN.MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0); m.load_state_dict(ref.state_dict()); m = m.cuda().eval()
data = make_batch("118v2", 128, seed=0)
ref64 = copy.deepcopy(ref).double(); d64 = data.clone(); d64.x, d64.y, d64.edge_attr = data.x.double(), data.y.double(), data.edge_attr.double()
out64, int64_ = ref64(d64, return_intermediates=True); torch.nn.MSELoss()(out64, d64.y).backward()
out32, int32_ = ref(data, return_intermediates=True); torch.nn.MSELoss()(out32, data.y).backward()
dd = data.to('cuda'); outo = m(dd); torch.nn.MSELoss()(outo, dd.y).backward()

# chained single layers (weights shared through a second module set)
layers = []
for l in ref.layers:
    if isinstance(l, ref_cpu.EdgeAggregation):
        o = MPN.EdgeAggregation(l.nfeature_dim, l.efeature_dim, 129, l.output_dim)
    else:
        o = MPN.TAGConv(l.in_channels, l.out_channels, l.K)
    o.load_state_dict(l.state_dict()); layers.append(o.cuda())
me = copy.deepcopy(ref.mask_embd).cuda()
ei, ea = ref_cpu.undirect_graph(data.edge_index, data.edge_attr); ei, ea = ei.cuda(), ea.cuda()
x = me(dd.pred_mask.float()) + dd.x
inter = [x]
for l in layers[:-1]:
    x = l(x, ei, ea) if isinstance(l, MPN.EdgeAggregation) else l(x, ei)
    inter.append(x); x = x.relu()
x = layers[-1](x, ei, ea); inter.append(x)
torch.nn.MSELoss()(x, dd.y).backward()
print("forward intermediates (chained vs fp64 | cpu32 vs fp64), relative to max")
for a, b, c in zip(inter, int32_, int64_):
    sc = c.abs().max().item()
    print(f"  {tuple(c.shape)} chained {(a.detach().cpu().double()-c).abs().max().item()/sc:.2e} cpu32 {(b.double()-c).abs().max().item()/sc:.2e}  max {sc:.3g} median {c.abs().median().item():.3g}")
print(f"fused out {(outo.detach().cpu().double()-out64).abs().max().item()/out64.abs().max().item():.2e}")
chained = {}
for i, l in enumerate(layers):
    for k, p in l.named_parameters():
        chained[f"layers.{i}.{k}"] = p.grad
for k, p in me.named_parameters():
    chained[f"mask_embd.{k}"] = p.grad
for (k, p), q, t in zip(m.named_parameters(), ref.parameters(), ref64.parameters()):
    g64 = t.grad; sc = g64.abs().max().item()
    e = lambda g: (g.cpu().double() - g64).abs().max().item() / sc
    print(f"{k:34s} fused {e(p.grad):.2e} chained {e(chained[k]):.2e} cpu32 {e(q.grad):.2e}")

# ---- flip census at the last EdgeAggregation: do ReLU masks of the hidden pre-activation differ from float64's?
l6 = ref64.layers[-1].edge_aggr
src, dst = ei.cpu()[0], ei.cpu()[1]
def pre(xin):
    xin = xin.relu()
    z = torch.cat([xin[dst], xin[src], ea.cpu().double()], -1)
    return torch.nn.functional.linear(z, l6[0].weight.detach(), l6[0].bias.detach())
p64 = pre(int64_[-2].detach())
for nm, xv in (("ours", inter[-2].detach().cpu().double()), ("cpu32", int32_[-2].detach().double())):
    pv = pre(xv)
    fl = ((pv > 0) != (p64 > 0))
    print(f"flips at layer 6 hidden ({nm} inputs, float64 math): {int(fl.sum())} of {fl.numel()}  max|pre-pre64| {(pv-p64).abs().max().item():.2e}  pre std {p64.std().item():.3g}")
# layer-output ReLU gates (y > 0) of every hidden layer
for li, (a, b, c) in enumerate(zip(inter[1:-1], int32_[1:-1], int64_[1:-1])):
    fo = int(((a.detach().cpu() > 0) != (c > 0)).sum()); fc = int(((b > 0) != (c > 0)).sum())
    print(f"layer {li} output gate flips: ours {fo} cpu32 {fc} of {c.numel()}; exact zeros in fp64 {int((c == 0).sum())}")
