#!/usr/bin/env python3
"""Training entry counterpart of the reference's train.py (same CLI: `python3 train.py --cfg_json configs/standard.json
--case 118v2 --model MaskEmbdMultiMPN --train_loss_fn mse_loss --batch-size 128 --lr 0.001 --num-epochs N`).

Data: when `<data-dir>/raw/case<case>_{node,edge}_features.npy` exist they are loaded by the device-resident
`PowerFlowData` (same splits, masks and normalisation as datasets/PowerFlowData.py; batches are assembled on the GPU);
otherwise (this image ships no dataset files and no pandapower) samples are synthetic grids with the same tensor layout
(poweflownet_amd/synth.py).  No wandb.  Like the reference, the three model
dims come from the data (4/2/4), not from the JSON (train.py:106-117).  Under torchrun every rank trains on its shard of
each global batch and gradients are averaged with one flat all-reduce (poweflownet_amd/dp.py)."""
import os
import time

import numpy as np
import torch

from poweflownet_amd import dp
from poweflownet_amd.data import DataLoader
from poweflownet_amd.datasets import PowerFlowData, random_bus_type
from poweflownet_amd.networks.MPN import (MPN, MaskEmbdMPN, MaskEmbdMultiMPN, MaskEmbdMultiMPN_NoMP, MPN_simplenet, MultiMPN,
                                          SkipMPN)
from poweflownet_amd.optim import FlatAdamW
from poweflownet_amd.synth import make_dataset
from poweflownet_amd.utils.argument_parser import argument_parser
from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss, MixedMSEPoweImbalance, PowerImbalance
from poweflownet_amd.utils.evaluation import evaluate_epoch
from poweflownet_amd.utils.training import GraphedTrainStep, append_to_json, train_epoch


def main():
    args = argument_parser()
    # train.py:30-38.  As in the reference, every entry but MaskEmbdMultiMPN / MPN_simplenet asserts a 12-wide node layout
    # the dataset does not produce (networks/MPN.py:194,...) and stops at its first forward.  `MultiConvNet` (ChebConv over
    # 5 edge features, :671-750) is not on the EdgeAggregation/TAGConv path and is not built; `MaskEmbdMultiMPN_NoMP` is
    # defined by the reference but absent from its table -- accepted here for completeness.
    models = {"MPN": MPN, "MPN_simplenet": MPN_simplenet, "SkipMPN": SkipMPN, "MaskEmbdMPN": MaskEmbdMPN,
              "MultiMPN": MultiMPN, "MaskEmbdMultiMPN": MaskEmbdMultiMPN, "MaskEmbdMultiMPN_NoMP": MaskEmbdMultiMPN_NoMP}
    if args.model not in models:
        raise SystemExit(f"--model {args.model}: not one of {sorted(models)}")
    rank, local_rank, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a HIP device: poweflownet_amd has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.manual_seed(1234)
    np.random.seed(1234)
    raw = os.path.join(args.data_dir, "raw", f"case{args.case}_node_features.npy")
    if os.path.exists(raw):                                       # the reference's own files (train.py:76-79)
        # (the reference applies transform=random_bus_type to the train split; the model never reads bus_type, and
        #  without the per-sample transform the batches are assembled on the device)
        trainset = PowerFlowData(root=args.data_dir, case=args.case, split=[.5, .2, .3], task="train",
                                 normalize=not args.disable_normalize, device=device)
        valset = PowerFlowData(root=args.data_dir, case=args.case, split=[.5, .2, .3], task="val",
                               normalize=not args.disable_normalize, device=device)
    else:
        n = args.synthetic_samples
        full = make_dataset(args.case, n, seed=0)
        n_tr, n_va = int(0.5 * n), int(0.2 * n)                  # split [.5, .2, .3] (train.py:76)
        trainset, valset = full[:n_tr], full[n_tr:n_tr + n_va]
    shard = (rank, world) if world > 1 else None
    train_loader = DataLoader(trainset, batch_size=args.batch_size * world, shuffle=True,
                              generator=torch.Generator().manual_seed(1234), shard=shard)
    val_loader = DataLoader(valset, batch_size=args.batch_size, shuffle=False)
    if args.train_loss_fn == "masked_l2":
        loss_fn = Masked_L2_loss(regularize=args.regularize, regcoeff=args.regularization_coeff)
    elif args.train_loss_fn == "mse_loss":
        from poweflownet_amd.loss import MSELoss
        loss_fn = MSELoss()                                       # torch.nn.MSELoss semantics (train.py:103), one kernel
    elif args.train_loss_fn in ("power_imbalance", "mixed_mse_power_imbalance"):
        if not hasattr(trainset, "get_data_means_stds"):
            raise SystemExit("the physics losses need the dataset statistics: pass --data-dir with the raw files")
        cls = PowerImbalance if args.train_loss_fn == "power_imbalance" else MixedMSEPoweImbalance
        kw = {} if args.train_loss_fn == "power_imbalance" else {"alpha": 0.9}       # train.py:97,101
        loss_fn = cls(*[t.cpu() for t in trainset.get_data_means_stds()], **kw)
    else:
        raise SystemExit(f"unknown --train_loss_fn {args.train_loss_fn}")
    eval_loss_fn = Masked_L2_loss(regularize=False)
    model = models[args.model](nfeature_dim=4, efeature_dim=2, output_dim=4, hidden_dim=args.hidden_dim,
                             n_gnn_layers=args.n_gnn_layers, K=args.K, dropout_rate=args.dropout_rate).to(device)
    dp.broadcast_parameters(model)
    if rank == 0:
        print("Total number of parameters: ", sum(p.numel() for p in model.parameters()))
    optimizer = FlatAdamW(model, lr=args.lr)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=args.lr, steps_per_epoch=len(train_loader),
                                                    epochs=args.num_epochs)
    run_id = time.strftime("%Y%m%d-%H%M%S")
    if rank == 0 and hasattr(trainset, "xymean"):     # normalising params (train.py:81-88); None entries under --disable_normalize
        os.makedirs(os.path.join(args.data_dir, "params"), exist_ok=True)
        torch.save({k: (None if getattr(trainset, k) is None else getattr(trainset, k).cpu())
                    for k in ("xymean", "xystd", "edgemean", "edgestd")},
                   os.path.join(args.data_dir, "params", f"data_params_{run_id}.pt"))
    best_val = float("inf")
    graphed = GraphedTrainStep(model, loss_fn, optimizer, dp_mode=getattr(args, "dp_mode", None))   # one hipGraph launch per batch; under DP it contains the all-reduce
    for epoch in range(args.num_epochs):
        t0 = time.time()
        train_loss = train_epoch(model, train_loader, loss_fn, optimizer, device, graph=graphed)
        t_train = time.time() - t0
        val_loss = evaluate_epoch(model, val_loader, eval_loss_fn, device)
        scheduler.step()                                          # once per epoch, like train.py:145
        if rank == 0:
            print(f"Epoch {epoch + 1} / {args.num_epochs}, train={train_loss:.4f}, val={val_loss:.4f}, "
                  f"{len(trainset) / max(t_train, 1e-9):.0f} train graphs/s")
            if args.save and val_loss < best_val:
                best_val = val_loss
                os.makedirs("models", exist_ok=True)
                torch.save({"epoch": epoch, "args": vars(args), "val_loss": best_val,
                            "model_state_dict": model.state_dict()}, os.path.join("models", f"model_{run_id}.pt"))
                print(f"saved models/model_{run_id}.pt")
                append_to_json(os.path.join("logs", "save_logs.json"), run_id,
                               {"val_loss": f"{best_val:.4f}", "train_loss": f"{train_loss:.4f}", "epoch": epoch})


if __name__ == "__main__":
    main()
