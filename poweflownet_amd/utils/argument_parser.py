"""Two-stage argument parsing of the reference (utils/argument_parser.py:5-65): defaults < JSON config < command
line.  Same flag names so `train.py --cfg_json ... --case ... --model ...` invocations carry over.  (`--regularize`
keeps the reference's `type=bool` quirk: any non-empty string is True.)

Like the reference (:10), `--cfg_json` defaults to `configs/standard.json` (hidden_dim 129), so a bare
`train.py --case 118v2` builds the same 129-wide model there and here; the path is tried relative to the working
directory first (the reference's behaviour) and then relative to this repository.  One deliberate deviation: `--model`
defaults to `MaskEmbdMultiMPN` (the reference's default `MPN` asserts a stale 12-wide node layout, networks/MPN.py:194,
and cannot run on its own dataset)."""
import argparse
import json
import os

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def argument_parser(argv=None):
    cfg = argparse.ArgumentParser(prog="PowerFlowNet", add_help=False)
    cfg.add_argument("--cfg_json", "--config", "--configs", default="configs/standard.json", type=str)
    p = argparse.ArgumentParser(prog="PowerFlowNet", description="train the MI355X PowerFlowNet hot path")
    p.add_argument("--nfeature_dim", type=int, default=6)
    p.add_argument("--efeature_dim", type=int, default=2)
    p.add_argument("--hidden_dim", type=int, default=128)
    p.add_argument("--output_dim", type=int, default=6)
    p.add_argument("--n_gnn_layers", type=int, default=4)
    p.add_argument("--K", type=int, default=3)
    p.add_argument("--dropout_rate", type=float, default=0.2)
    p.add_argument("--model", type=str, default="MaskEmbdMultiMPN")
    p.add_argument("--regularize", type=bool, default=True)
    p.add_argument("--regularization_coeff", type=float, default=1.0)
    p.add_argument("--data-dir", type=str, default="data")
    p.add_argument("--disable_normalize", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--train_loss_fn", type=str, default="masked_l2",
                   choices=["masked_l2", "power_imbalance", "mse_loss", "mixed_mse_power_imbalance"])
    p.add_argument("--num-epochs", type=int, default=100)
    p.add_argument("--batch-size", type=int, default=128)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--case", type=str, default="14")
    p.add_argument("--wandb", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--wandb-entity", type=str, default="PowerFlowNet")
    p.add_argument("--save", default=True, action=argparse.BooleanOptionalAction)
    # additions of this build (not in the reference): synthetic data size
    p.add_argument("--synthetic-samples", type=int, default=512)
    # ... and the launch form of the training step (poweflownet_amd.dp.GraphedStep): one hipGraph incl. the gradient all-reduce,
    # graph / eager all-reduce / graph, or eager launches.  Default: PFN_DP_MODE or "graph"; ranks always agree on the form.
    p.add_argument("--dp-mode", type=str, default=None, choices=["graph", "split", "eager"])
    args, left = cfg.parse_known_args(argv)
    if args.cfg_json is not None:
        path = args.cfg_json
        if not os.path.exists(path) and not os.path.isabs(path) and os.path.exists(os.path.join(_REPO, path)):
            path = os.path.join(_REPO, path)
        with open(path) as f:
            d = json.load(f)
        jargv = []
        for k, v in d.items():
            jargv += ["--" + k, str(v)]
        p.parse_known_args(jargv, args)
    p.parse_args(left, args)
    return args
