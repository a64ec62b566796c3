"""One rank of the data-parallel HIP path (launched by tests/test_gpu_dp.py; not a test module itself).

Runs MaskEmbdMultiMPN on the HIP kernels on its shard of a global batch, checks that the in-place branch of
`dp.allreduce_gradients` is the one taken (the `.grad`s are views of the flat buffer `pfn_mpn_backward` wrote), averages the
gradients with ONE collective, applies one FlatAdamW step, and (rank 0) saves the averaged flat gradient and the updated
flat parameters for the parent to compare with a single-process run on the whole batch."""
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
    from poweflownet_amd.synth import make_dataset

    out_path, case, gb, nsamples = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rank, local_rank, world = dp.init_from_env()
    dev = torch.device("cuda", 0 if os.environ.get("PFN_SINGLE_DEVICE") else local_rank)
    ds = make_dataset(case, nsamples, seed=0)
    torch.manual_seed(1234 + rank)                      # deliberately different replicas: the broadcast must fix it
    model = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).to(dev).train()
    dp.broadcast_parameters(model)
    opt = FlatAdamW(model, lr=1e-3)
    loader = DataLoader(ds, batch_size=gb, shard=(rank, world))
    assert len(loader) == len(DataLoader(ds, batch_size=gb, shard=((rank + 1) % world, world)))
    loss_fn = MSELoss()
    steps = 0
    for batch in loader:                                 # includes a short tail batch when nsamples % gb != 0
        batch = batch.to(dev)
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(batch), batch.y)
        loss.backward()
        params = model._ordered_params()
        assert dp._grads_are_views_of(model.flat_grad(), params), "in-place branch not taken: .grad is not a view of the flat buffer"
        before = model.flat_grad().clone()
        dp.allreduce_gradients(model)
        assert model.flat_grad().data_ptr() == params[0].grad.data_ptr()
        if steps == 0:
            first_grad = model.flat_grad().clone()
            changed = not torch.equal(before, first_grad)
        opt.step()
        steps += 1
    torch.cuda.synchronize()
    # every rank holds identical parameters after identical averaged updates
    flat = opt.flat_param.detach().clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered), "replicas diverged"
    if rank == 0:
        torch.save({"grad": first_grad.cpu(), "params": flat.cpu(), "steps": steps, "changed": changed}, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
