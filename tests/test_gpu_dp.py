"""Data parallelism on the HIP path with more than one rank (BASELINE configs[4]'s code path).  A gpurun box has
ONE GPU, so the ranks share device 0 and exchange gradients over gloo (PFN_SINGLE_DEVICE / PFN_DIST_BACKEND test aids of
poweflownet_amd/dp.py -- RCCL refuses two ranks on one device); the RCCL collective itself runs in the world-size-1 test.
Every child runs in its own process group under a hard timeout and is killed as a group when it overruns."""
import json
import os
import signal
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:          # a fresh port per launch: a fixed one can still be in TIME_WAIT from the last run
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env, timeout):
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)        # the launcher AND its ranks: nothing may be left holding the GPU
        out, err = proc.communicate()
        pytest.fail(f"{' '.join(cmd[-8:])} did not finish within {timeout} s\n--- stderr tail ---\n{err[-3000:]}")
    assert proc.returncode == 0, err[-3000:]
    return out, err


def _torchrun(nproc, script_and_args, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_HANG_DUMP="150", **(extra_env or {}))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
            "127.0.0.1", "--master-port", str(_free_port())] + script_and_args, env


@pytest.mark.parametrize("case,gb,nsamples", [("118v2", 8, 8), ("14", 6, 15)])
def test_two_ranks_hip_inplace_allreduce_equals_single_process(tmp_path, case, gb, nsamples):
    """Two ranks on the HIP kernels: the in-place flat-gradient branch is taken, the averaged gradient of the first global
    batch equals the single-process gradient on that whole batch, the replicas stay identical
    through FlatAdamW steps, and a short tail batch is truncated to a multiple of the world size on every rank
    (15 = 2 x 6 + 3 -> 3 steps of 3, 3 and 1 graphs per rank; a tail of fewer samples than ranks is dropped everywhere,
    tests/test_data.py)."""
    out_path = str(tmp_path / "dp.pt")
    cmd, env = _torchrun(2, [os.path.join(ROOT, "tests", "dp_hip_worker.py"), out_path, case, str(gb), str(nsamples)],
                         {"PFN_SINGLE_DEVICE": "1", "PFN_DIST_BACKEND": "gloo"})
    _run(cmd, env, 420)
    got = torch.load(out_path)
    from poweflownet_amd.data import DataLoader
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
    from poweflownet_amd.optim import FlatAdamW
    from poweflownet_amd.synth import make_dataset
    from tests.util import assert_close
    assert got["changed"], "the all-reduce left the gradient untouched"
    ds = make_dataset(case, nsamples, seed=0)
    torch.manual_seed(1234)                               # rank 0's replica = what was broadcast
    model = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).to("cuda:0").train()
    opt = FlatAdamW(model, lr=1e-3)
    loader = DataLoader(ds, batch_size=gb)
    expect_steps = nsamples // gb + (1 if nsamples % gb >= 2 else 0)
    assert got["steps"] == expect_steps
    steps = 0
    for batch in loader:
        if steps == expect_steps:
            break
        batch = batch.to("cuda:0")
        nb = batch.num_graphs // 2 * 2                    # the sharded loader truncates a global batch to a multiple of world
        if nb != batch.num_graphs:
            from poweflownet_amd.data import Batch
            idx = list(range(steps * gb, steps * gb + nb))
            batch = Batch.from_data_list([ds[i] for i in idx]).to("cuda:0")
        opt.zero_grad(set_to_none=True)
        loss = MSELoss()(model(batch), batch.y)
        loss.backward()
        if steps == 0:
            assert_close(got["grad"], model.flat_grad(), 1e-5, "averaged DP gradient vs single-process global batch")
        opt.step()
        steps += 1
    # (AdamW's first steps move every weight by ~lr * g / |g|: an update is far less smooth in the gradient than the
    #  gradient itself, so the parameters are held to 1e-4 of their largest entry -- 0.05 lr -- not to 1e-5)
    assert_close(got["params"], opt.flat_param, 1e-4, f"parameters after {steps} DP steps")


def test_bench_two_ranks_on_one_gpu_gloo():
    """bench.py's multi-process path (what the driver launches for N > 1): hipGraph(fwd+bwd) -> eager all-reduce ->
    hipGraph(AdamW), every rank entering every collective including the ones inside rank 0's profiling pass."""
    cmd, env = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                             "--profile-steps", "2", "--no-cpu-baseline", "--case", "14", "--batch", "8"],
                         {"PFN_SINGLE_DEVICE": "1", "PFN_DIST_BACKEND": "gloo"})
    out, _ = _run(cmd, env, 420)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                             # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"] is not None


def test_bench_rccl_collective_path_world1():
    """The RCCL leg of the same sequence on the one GPU there is: PFN_FORCE_DIST=1 initialises a world-size-1 NCCL(=RCCL)
    process group, so bench.py takes its DP branch -- graph replay, `all_reduce(AVG)` on the flat gradient buffer through
    RCCL on the compute stream, graph replay -- next to a live process group and its watchdog."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PFN_HANG_DUMP="150")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--profile-steps", "2",
           "--no-cpu-baseline", "--case", "14", "--batch", "16"]
    out, err = _run(cmd, env, 420)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:] + err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["launch"] == "hipGraph replay", d["config"]
