"""Two-stage argument parsing of the reference (utils/argument_parser.py:5-65): defaults < JSON config < command
line.  Same flag names so `train.py --cfg_json ... --case ... --model ...` invocations carry over.  (`--regularize`
keeps the reference's `type=bool` quirk: any non-empty string is True.)"""
import argparse
import json


def argument_parser(argv=None):
    cfg = argparse.ArgumentParser(prog="PowerFlowNet", add_help=False)
    cfg.add_argument("--cfg_json", "--config", "--configs", default=None, type=str)
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
    args, left = cfg.parse_known_args(argv)
    if args.cfg_json is not None:
        with open(args.cfg_json) as f:
            d = json.load(f)
        jargv = []
        for k, v in d.items():
            jargv += ["--" + k, str(v)]
        p.parse_known_args(jargv, args)
    p.parse_args(left, args)
    return args
