import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
script = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.synth import make_batch
B, hub, train = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(3)
m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to("cuda:0")
m.train() if train else m.eval()
d = make_batch("6470rte", B, seed=4, hub_frac=hub).to("cuda:0")
out = m(d)
torch.nn.MSELoss()(out, d.y).backward()
torch.save({{"out": out.detach().cpu(), "g": m.flat_grad().cpu()}}, sys.argv[1])
"""
ok = True
for B, hub, train in ((8, 0.0, 1), (5, 0.2, 0), (64, 0.0, 1), (13, 0.2, 1)):
    res = {}
    for tag, env in (("v2", {}), ("v1", {"PFN_BIG_HOPS_V1": "1"}), ("v2_wpg3", {"PFN_BIG_HOPS_WPG": "3"})):
        path = f"/tmp/ab_{tag}.pt"
        subprocess.run([sys.executable, "-c", script, path, str(B), str(hub), str(train)], check=True, env=dict(os.environ, **env), timeout=900)
        res[tag] = torch.load(path)
    for tag in ("v2", "v2_wpg3"):
        same = torch.equal(res[tag]["out"], res["v1"]["out"]) and torch.equal(res[tag]["g"], res["v1"]["g"])
        print(f"B={B} hub={hub} train={train} {tag} vs v1: bit-identical={same} finite={bool(torch.isfinite(res[tag]['g']).all())}", flush=True)
        ok = ok and same
print("ALL IDENTICAL" if ok else "MISMATCH")
