#!/usr/bin/env python3
"""Evaluation entry counterpart of the reference's test.py: load the checkpoint and the normalising parameters of a training
run, rebuild the test split with them (test.py:44-53) and report MaskedL2V2 / MaskedL1 terms (normalised and de-normalised)
plus PowerImbalance, Masked_L2_loss and MSE on it (test.py:113-130).

    python test.py --cfg_json configs/standard.json --case 118v2 --data-dir DATA --run-id 20260928-120000

`--run-id` replaces the run id the reference hard-codes at test.py:26.  Needs the raw dataset files; the model runs on the HIP
device (no CPU path)."""
import os
import sys
from functools import partial

import torch

from poweflownet_amd.data import DataLoader
from poweflownet_amd.datasets import PowerFlowData, denormalize
from poweflownet_amd.loss import MSELoss
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN, MPN_simplenet
from poweflownet_amd.utils.argument_parser import argument_parser
from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss, MaskedL1, MaskedL2V2, PowerImbalance
from poweflownet_amd.utils.evaluation import evaluate_epoch_v2, load_model


@torch.no_grad()
def main():
    argv = sys.argv[1:]
    run_id = None
    if "--run-id" in argv:
        i = argv.index("--run-id")
        run_id = argv[i + 1]
        del argv[i:i + 2]
    if run_id is None:
        raise SystemExit("test.py needs --run-id <id of a train.py run> (models/model_<id>.pt, <data-dir>/params/data_params_<id>.pt)")
    args = argument_parser(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test.py needs a HIP device: poweflownet_amd has no CPU fallback")
    device = torch.device("cuda")
    p = torch.load(os.path.join(args.data_dir, "params", f"data_params_{run_id}.pt"), map_location="cpu")
    testset = PowerFlowData(root=args.data_dir, case=args.case, split=[.5, .2, .3], task="test", xymean=p["xymean"],
                            xystd=p["xystd"], edgemean=p["edgemean"], edgestd=p["edgestd"], device=device)
    loader = DataLoader(testset, batch_size=args.batch_size, shuffle=False)
    sample = testset[0]
    print(f"#slack:{int((sample.bus_type == 0).sum())},\t#pv:{int((sample.bus_type == 1).sum())},\t#pq:{int((sample.bus_type == 2).sum())}")
    models = {"MaskEmbdMultiMPN": MaskEmbdMultiMPN, "MPN_simplenet": MPN_simplenet}
    nin, nout, ne = testset.get_data_dimensions()
    model = models[args.model](nfeature_dim=nin, efeature_dim=ne, output_dim=nout, hidden_dim=args.hidden_dim,
                               n_gnn_layers=args.n_gnn_layers, K=args.K, dropout_rate=args.dropout_rate).to(device)
    model.eval()
    model, _ = load_model(model, run_id, device)
    print(f"Model: {args.model}\nCase: {args.case}")
    de = partial(denormalize, mean=p["xymean"], std=p["xystd"])
    for title, loss, pre in (("MaskedL2", MaskedL2V2(), None), ("MaskedL2(denorm)", MaskedL2V2(), de), ("MaskedL1(denorm)", MaskedL1(), de)):
        for key, value in evaluate_epoch_v2(model, loader, loss, device, pre_loss_fn=pre).items():
            print(f"{title} {key}:\t{value:.6f}")
    stats = [t.cpu() for t in testset.get_data_means_stds()]
    for name, loss in (("PowerImbalance", PowerImbalance(*stats)), ("Masked_L2_loss", Masked_L2_loss(regularize=False)), ("MSE", MSELoss())):
        terms = evaluate_epoch_v2(model, loader, loss, device)
        print(f"{name}:\t{terms['total']:.6f}")
        if "ref" in terms:
            print(f"{name}(ref):\t{terms['ref']:.6f}")


if __name__ == "__main__":
    main()
