"""One rank of a data-parallel run with per-batch topologies where ONE rank meets a poisoned batch (launched by
tests/test_gpu_dp.py; not a test module itself).  The captured step's guarded AdamW update must be decided on a value every
rank sees (the rank-summed loss, dp.GraphedStep `extra`): all ranks skip that update together, nobody's parameters turn NaN,
replicas stay identical, and the skip is counted where the host loop can see it.  Rank 0 writes a JSON report."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from poweflownet_amd import dp
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
    from poweflownet_amd.optim import FlatAdamW
    from poweflownet_amd.synth import make_batch
    from poweflownet_amd.utils.training import GraphedTrainStep

    out_path, bad_step, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, local_rank, world = dp.init_from_env()
    assert dp.active()
    dev = torch.device("cuda", 0 if os.environ.get("PFN_SINGLE_DEVICE") else local_rank)
    torch.manual_seed(1234)
    model = MaskEmbdMultiMPN(4, 2, 4, 32, 2, 2, 0.0).to(dev).train()
    opt = FlatAdamW(model, lr=1e-2)
    g = GraphedTrainStep(model, MSELoss(), opt, allreduce=True)
    losses = []
    for step in range(nsteps):
        d = make_batch("14", 6, seed=100 * rank + step).to(dev)
        d.edge_index = d.edge_index.clone()                 # a NEW edge_index tensor per batch: the step goes `dynamic`
        if step == bad_step and rank == world - 1:
            d.edge_index[0, 3] = d.x.shape[0] + 5            # out of range: flagged on the device, the loss arrives as NaN
        losses.append(float(g(d).item()))
    torch.cuda.synchronize()
    assert g.dynamic and g.graph is not None and not g.disabled
    flat = opt.flat_param.detach().clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    counts = opt.step_count.clone()
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    torch.distributed.all_gather(all_counts, counts)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"replicas_equal": all(torch.equal(t, gathered[0]) for t in gathered),
                       "finite": bool(torch.isfinite(flat).all()), "form": g.graph.form,
                       "step_counts": [c.tolist() for c in all_counts], "losses_rank0": losses, "world": world}, f)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
