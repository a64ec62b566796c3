"""Host logic without a GPU: argument parser precedence, loss dispatch helpers, train/eval loop semantics (driven with
the CPU oracle model, which is allowed in tests), and the world-size-2 gloo data-parallel step."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu
from poweflownet_amd import dp
from poweflownet_amd.data import DataLoader
from poweflownet_amd.synth import make_batch, make_dataset
from poweflownet_amd.utils.argument_parser import argument_parser
from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss, PowerImbalance
from poweflownet_amd.utils.evaluation import evaluate_epoch, num_params
from poweflownet_amd.utils.training import train_epoch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_argument_parser_precedence():
    a = argument_parser(["--cfg_json", os.path.join(ROOT, "configs", "wide.json"), "--case", "6470rte", "--K", "5"])
    assert (a.hidden_dim, a.n_gnn_layers, a.K, a.case) == (129, 6, 5, "6470rte")      # defaults < JSON < CLI
    b = argument_parser([])                  # like the reference, no flag = configs/standard.json (hidden_dim 129)
    assert (b.hidden_dim, b.train_loss_fn, b.batch_size, b.cfg_json) == (129, "masked_l2", 128, "configs/standard.json")


class OracleMaskedL2(Masked_L2_loss):
    """The product's Masked_L2_loss has no CPU path; the host-logic tests below run the loops on the CPU oracle model, so
    they use the oracle's restatement of the loss behind the SAME class (the loops dispatch on isinstance)."""

    def forward(self, output, target, mask):
        return ref_cpu.masked_l2_loss(output, target, mask, self.regularize, self.regcoeff)


def test_losses_have_no_cpu_path():
    torch.manual_seed(0)
    out, y = torch.randn(10, 4), torch.randn(10, 4)
    mask = torch.randint(0, 2, (10, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        Masked_L2_loss(regularize=True, regcoeff=0.5)(out, y, mask)
    from poweflownet_amd.loss import MSELoss
    with pytest.raises(RuntimeError, match="HIP device"):
        MSELoss()(out, y)
    pi = PowerImbalance(torch.zeros(1, 4), torch.ones(1, 4), torch.zeros(1, 2), torch.ones(1, 2))
    with pytest.raises(RuntimeError):
        pi(out, torch.zeros(2, 3, dtype=torch.long), torch.randn(3, 2))


def test_train_and_eval_epoch_semantics_on_oracle_model():
    torch.manual_seed(0)
    ds = make_dataset("14", 12)
    loader = DataLoader(ds, batch_size=4)
    model = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 16, 2, 2, 0.0)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    before = evaluate_epoch(model, loader, OracleMaskedL2(regularize=False), "cpu")
    l1 = train_epoch(model, loader, torch.nn.MSELoss(), opt, "cpu")
    l2 = train_epoch(model, loader, OracleMaskedL2(), opt, "cpu")
    for _ in range(20):
        l2 = train_epoch(model, loader, OracleMaskedL2(), opt, "cpu")
    after = evaluate_epoch(model, loader, OracleMaskedL2(regularize=False), "cpu")
    assert l1 > 0 and l2 > 0 and after < before
    assert num_params(model) == sum(p.numel() for p in model.parameters())


def _dp_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, _, w = dp.init_from_env(backend="gloo")
    ds = make_dataset("14", 8)
    torch.manual_seed(1234 + rank)                       # deliberately different init: broadcast must fix it
    model = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 8, 2, 3, 0.0)
    dp.broadcast_parameters(model)
    loader = DataLoader(ds, batch_size=8, shard=(r, w))
    batch = next(iter(loader))
    loss = torch.nn.MSELoss()(model(batch), batch.y)
    loss.backward()
    dp.allreduce_gradients(model, ordered_params=list(model.parameters()))
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    if rank == 0:
        ret["grad"], ret["params"] = flat.clone(), params.clone()
    gathered = [torch.zeros_like(flat) for _ in range(w)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    dist.destroy_process_group()


def test_dp_world2_gloo_equals_single_process_global_batch():
    """Mean of the two ranks' gradients on their shards == gradient on the global batch (MSELoss mean, equal node counts)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    torch.set_num_threads(1)
    ds = make_dataset("14", 8)
    torch.manual_seed(1234)
    model = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 8, 2, 3, 0.0)
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), ret["params"])
    batch = next(iter(DataLoader(ds, batch_size=8)))
    torch.nn.MSELoss()(model(batch), batch.y).backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    got = ret["grad"]
    assert (got - want).abs().max().item() <= 1e-6 * want.abs().max().item() + 1e-9


def _agree_worker(rank, w, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=w)
    from poweflownet_amd import dp
    calls = []
    gs = dp.GraphedStep(lambda: calls.append("fb") or torch.zeros(()), lambda: calls.append("opt"), model=None, allreduce=True, mode="graph")
    gs._reduce = lambda: calls.append("reduce")
    gs._reduce_captured = gs._reduce
    # a capture attempt that fails on rank 1 ONLY: every rank must report failure (and then agree on the next form)
    def build():
        if rank == 1:
            raise RuntimeError("capture failed here")
        gs.graphs = ["a graph"]
    ok_mixed = gs._try(build)
    ok_all = gs._try(lambda: None)
    # no GPU in this process: both graph forms fail on every rank -> everybody lands in "eager", whose replay is the three calls
    gs.capture()
    gs.replay()
    ret[rank] = (ok_mixed, ok_all, gs.form, list(gs.graphs), calls[-3:])
    dist.destroy_process_group()


def test_dp_ranks_agree_on_the_launch_form():
    """dp.GraphedStep: a capture that fails on ONE rank demotes EVERY rank (the success flag is MIN-all-reduced before a form is
    chosen), so ranks never disagree on how a step is launched; world 2 over gloo, no GPU: both graph forms fail everywhere and
    the step runs as eager launches (fwd_bwd, all-reduce, optimizer)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_agree_worker, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        ok_mixed, ok_all, form, graphs, calls = ret[rank]
        assert ok_mixed is False and ok_all is True and form == "eager" and graphs == [] and calls == ["fb", "reduce", "opt"], (rank, ret[rank])


def test_evaluation_metrics_and_evaluate_epoch_v2():
    """MaskedL2V2 / MaskedL1 (reference utils/custom_loss_functions.py:48-97) by their definitions, and evaluate_epoch_v2's
    accumulation rule (utils/evaluation.py:158-165: first batch unweighted, the rest weighted by len(data))."""
    from poweflownet_amd.utils.custom_loss_functions import MaskedL1, MaskedL2V2
    from poweflownet_amd.utils.evaluation import evaluate_epoch_v2
    torch.manual_seed(1)
    out, y = torch.randn(30, 4), torch.randn(30, 4)
    mask = torch.randint(0, 2, (30, 4))
    for cls, err in ((MaskedL2V2, (out - y) ** 2), (MaskedL1, (out - y).abs())):
        t = cls()(out, y, mask)
        per = torch.stack([err[:, f][mask[:, f].bool()].mean() for f in range(4)])
        assert torch.allclose(torch.stack([t["vm"], t["va"], t["p"], t["q"]]), per, atol=1e-6)
        assert torch.allclose(t["balanced total"], per.mean(), atol=1e-6)
        assert torch.allclose(t["total"], err[mask.bool()].mean(), atol=1e-6)

    class Echo(torch.nn.Module):             # "model" whose prediction is its input
        def forward(self, data):
            return data.x

    batches = [make_batch("14", b, seed=b) for b in (2, 3, 1)]
    got = evaluate_epoch_v2(Echo(), batches, MaskedL2V2(), "cpu")
    terms = [MaskedL2V2()(b.x, b.y, b.pred_mask) for b in batches]
    lens = [len(b) for b in batches]
    want = (terms[0]["total"].item() + sum(t["total"].item() * n for t, n in zip(terms[1:], lens[1:]))) / sum(lens)
    assert abs(got["total"] - want) < 1e-6 and set(got) == {"total", "balanced total", "vm", "va", "p", "q"}


def test_gemm_nt_block_order_covers_every_row_group_and_slice_once():
    """gemm_nt_kernel reads its launch-order id as [chunk of 8 row groups][slice][row group in chunk] so that the column slices of a
    row group share an XCD (csrc/gemm_nt.hip, round 6; a tail of gx % 8 row groups keeps the plain order).  The mapping restated here
    line by line must be a bijection onto (row group, slice) for every grid the launcher can produce, and must put the slices of a
    row group of a full chunk on launch ids that are equal modulo 8 (= the same XCD under round-robin dispatch)."""
    import re
    src = open(os.path.join(ROOT, "poweflownet_amd", "csrc", "gemm_nt.hip")).read()
    # (the test follows the source: if these lines change, restate the mapping below)
    for line in ("const int chunk = lin / (8 * ns), r = lin - chunk * 8 * ns;", "slice = r >> 3;", "bx = chunk * 8 + (r & 7);",
                 "const int t = lin - full * ns, tail = gx - full;", "slice = t / tail;", "bx = full + (t - slice * tail);"):
        assert line in src, line
    assert re.search(r"full = gx & ~7", src)

    def block_of(lin, gx, ns):
        full = gx & ~7
        if ns > 1 and lin < full * ns:
            chunk, r = divmod(lin, 8 * ns)
            return chunk * 8 + (r & 7), r >> 3
        if ns > 1:
            t, tail = lin - full * ns, gx - full
            s = t // tail
            return full + (t - s * tail), s
        return lin % gx, lin // gx

    for ns in (1, 2, 3, 4):
        for gx in list(range(1, 40)) + [59, 118, 120, 127, 128, 256]:
            seen = {}
            for lin in range(gx * ns):
                seen.setdefault(block_of(lin, gx, ns), []).append(lin)
            assert sorted(seen) == [(b, s) for b in range(gx) for s in range(ns)], (gx, ns)
            assert all(len(v) == 1 for v in seen.values())
            if ns > 1:
                for b in range(gx & ~7):
                    assert len({seen[(b, s)][0] % 8 for s in range(ns)}) == 1, (gx, ns, b)
