import sys, copy, torch
sys.path.insert(0, '.')
from oracle import ref_cpu
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.synth import make_batch
torch.manual_seed(1234)
ref = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).eval()
m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0); m.load_state_dict(ref.state_dict()); m = m.to('cuda').eval()
data = make_batch("118v2", 128, seed=0)
torch.set_num_threads(8)
ref64 = copy.deepcopy(ref).double(); d64 = data.clone(); d64.x, d64.y, d64.edge_attr = data.x.double(), data.y.double(), data.edge_attr.double()
torch.nn.MSELoss()(ref64(d64), d64.y).backward()
torch.nn.MSELoss()(ref(data), data.y).backward()
dd = data.to('cuda'); torch.nn.MSELoss()(m(dd), dd.y).backward()
for (k, p), q, t in zip(m.named_parameters(), ref.parameters(), ref64.parameters()):
    g, g32, g64 = p.grad.cpu().double(), q.grad.double(), t.grad
    sc = g64.abs().max().item()
    line = f"{k:34s} ours {(g-g64).abs().max().item()/sc:.2e}  cpu32 {(g32-g64).abs().max().item()/sc:.2e}"
    if k.endswith("edge_aggr.0.weight"):
        fi = (g.shape[1]-2)//2
        for nm, sl in (("xi", slice(0,fi)), ("xj", slice(fi,2*fi)), ("ea", slice(2*fi,2*fi+2))):
            line += f" | {nm} {(g[:,sl]-g64[:,sl]).abs().max().item()/sc:.1e}"
    print(line)
