"""Single-layer accuracy at full size (15,104 nodes): HIP layer vs float64 oracle vs fp32 oracle."""
import sys, copy, torch
sys.path.insert(0, '.')
from oracle import ref_cpu
from poweflownet_amd.networks import MPN
from poweflownet_amd.synth import make_batch
torch.manual_seed(7)
torch.set_num_threads(8)
data = make_batch("118v2", 128, seed=0)
ei, ea = ref_cpu.undirect_graph(data.edge_index, data.edge_attr)
n = data.x.shape[0]


def run(kind, fin, fout, relu_in):
    x = torch.randn(n, fin); x = x.relu() if relu_in else x
    g = torch.randn(n, fout)
    if kind == "ea":
        ref = ref_cpu.EdgeAggregation(fin, 2, 129, fout); ours = MPN.EdgeAggregation(fin, 2, 129, fout)
    else:
        ref = ref_cpu.TAGConv(fin, fout, 3); ours = MPN.TAGConv(fin, fout, 3)
    ours.load_state_dict(ref.state_dict()); ours = ours.cuda()
    ref64 = copy.deepcopy(ref).double()

    def go(mod, x, ei, ea, g):
        x = x.clone().requires_grad_(True); ea = ea.clone().requires_grad_(True)
        y = mod(x, ei, ea) if kind == "ea" else mod(x, ei)
        y.backward(g)
        return dict(y=y.detach(), dx=x.grad, **({"dea": ea.grad} if kind == "ea" else {}),
                    **{k: p.grad for k, p in mod.named_parameters()})
    r64 = go(ref64, x.double(), ei, ea.double(), g.double())
    r32 = go(ref, x, ei, ea, g)
    ro = go(ours, x.cuda(), ei.cuda(), ea.cuda(), g.cuda())
    print(f"--- {kind} {fin}->{fout}")
    for k in r64:
        sc = r64[k].abs().max().item()
        print(f"{k:22s} ours {(ro[k].cpu().double()-r64[k]).abs().max().item()/sc:.2e}  cpu32 {(r32[k].double()-r64[k]).abs().max().item()/sc:.2e}")


run("ea", 129, 4, True)
run("ea", 129, 129, True)
run("ea", 4, 129, False)
run("tag", 129, 129, True)
