"""One rank of a data-parallel TRAINING run through `train_epoch` + `GraphedTrainStep` (launched by tests/test_gpu_dp.py; not a
test module itself): the replayed step contains the gradient all-reduce (dp.GraphedStep) -- one hipGraph with the RCCL
collective captured inside, or graph / eager all-reduce / graph under gloo -- and must leave the parameters exactly where the
eager data-parallel loop leaves them.  Rank 0 writes {"mode", "max_abs_diff", "losses_graph", "losses_eager"} as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    if os.environ.get("PFN_HANG_DUMP"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PFN_HANG_DUMP"]), exit=False)
    from poweflownet_amd import dp
    from poweflownet_amd.data import DataLoader
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
    from poweflownet_amd.optim import FlatAdamW
    from poweflownet_amd.datasets import PowerFlowData
    from poweflownet_amd.utils.training import GraphedTrainStep, train_epoch

    out_path, root, case, gb = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    rank, local_rank, world = dp.init_from_env()
    assert dp.active(), "no process group (world 1 needs PFN_FORCE_DIST=1)"
    dev = torch.device("cuda", 0 if os.environ.get("PFN_SINGLE_DEVICE") else local_rank)
    # the device-resident dataset (raw files written by the parent test): one cached edge_index tensor per batch size, so the
    # captured step replays for every full batch
    ds = PowerFlowData(root=root, case=case, split=[.5, .25, .25], task="train", device=dev)
    shard = (rank, world) if world > 1 else None

    def run(graphed, dp_mode=None):
        torch.manual_seed(1234)
        model = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).to(dev).train()
        opt = FlatAdamW(model, lr=1e-3)
        loader = DataLoader(ds, batch_size=gb * world, shard=shard)      # one topology tensor per batch size -> replays
        g = GraphedTrainStep(model, MSELoss(), opt, allreduce=True, dp_mode=dp_mode) if graphed else None
        losses = [train_epoch(model, loader, MSELoss(), opt, dev, allreduce=True, graph=g) for _ in range(2)]
        torch.cuda.synchronize()
        mode = None
        if g is not None:
            assert g.captured() is not None and not g.any_disabled(), "the data-parallel step was not captured"
            assert g._children, "the device-resident dataset did not take the indexed path (the step gathers its own batch)"
            mode = g.captured().mode
            if dp_mode is not None:
                mode = g.captured().form
        return opt.flat_param.detach().clone(), losses, mode

    p_graph, l_graph, mode = run(True)
    p_eager, l_eager, _ = run(False)
    # the three launch forms of dp.GraphedStep, asked for by name: the same kernels in the same order -> the same bits
    forms = {}
    for want in ("graph", "split", "eager"):
        p_m, l_m, form = run(True, want)
        forms[want] = {"form": form, "max_abs_diff": float((p_m - p_eager).abs().max()), "losses_equal": l_m == l_eager}
    gathered = [torch.zeros_like(p_graph) for _ in range(world)]
    torch.distributed.all_gather(gathered, p_graph)
    assert all(torch.equal(t, gathered[0]) for t in gathered), "replicas diverged"
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"mode": mode, "max_abs_diff": float((p_graph - p_eager).abs().max()), "scale": float(p_eager.abs().max()),
                       "losses_graph": l_graph, "losses_eager": l_eager, "world": world, "forms": forms,
                       "backend": torch.distributed.get_backend()}, f)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
