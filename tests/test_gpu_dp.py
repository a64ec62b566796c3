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
    """bench.py's multi-process path as the driver launches it for N > 1 (torchrun).  Under gloo (the collective goes through
    the host and cannot be captured) the step is hipGraph(fwd+bwd) -> eager all-reduce -> hipGraph(AdamW); every rank enters
    every collective including the ones inside rank 0's profiling pass."""
    cmd, env = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                             "--profile-steps", "2", "--no-cpu-baseline", "--case", "14", "--batch", "8"],
                         {"PFN_SINGLE_DEVICE": "1", "PFN_DIST_BACKEND": "gloo"})
    out, _ = _run(cmd, env, 420)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                             # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"] is not None
    assert d["config"]["launch"] == "hipGraph replay: graph(fwd+bwd) -> eager all-reduce -> graph(optimizer)"
    # the line proves who took part: one entry per rank with the device it ran on and its own step time (here both ranks share
    # the box's one GPU -- the roster says so instead of pretending to two devices)
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and d["distinct_devices"] == 1
    assert all("pci" in r["device"] and r["ms_per_step"] > 0 for r in d["ranks"])
    assert max(r["ms_per_step"] for r in d["ranks"]) <= d["ms_per_step"] * 1.0001


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment starts its own two ranks (torch.distributed.run on a free port) and
    rank 0 prints the one JSON line, labelled n_gpus = 2; asking for more GPUs than the box has (without the one-device test aid)
    is an error, not a mislabelled 1-GPU number."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_SINGLE_DEVICE="1", PFN_DIST_BACKEND="gloo", PFN_HANG_DUMP="150")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--profile-steps", "0",
           "--no-cpu-baseline", "--case", "14", "--batch", "8"]
    out, err = _run(cmd, env, 420)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:] + err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["value"] > 0
    if torch.cuda.device_count() < 2:
        env.pop("PFN_SINGLE_DEVICE")
        proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert proc.returncode != 0 and "device(s) are visible" in proc.stderr and "{" not in proc.stdout


def test_config5_workload_two_ranks_one_device():
    """BASELINE configs[4]'s per-rank workload (case6470rte x 64 per rank, data parallel) with two ranks -- sharing the one GPU
    of this box, gradients over gloo: the step bench.py times on the 8-GPU node, at its real size (2 x 11 GB of workspaces)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_SINGLE_DEVICE="1", PFN_DIST_BACKEND="gloo", PFN_HANG_DUMP="300")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--profile-steps", "0",
           "--no-cpu-baseline", "--case", "6470rte", "--batch", "64"]
    out, err = _run(cmd, env, 600)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:] + err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["nodes_per_gpu"] == 64 * 6470
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]            # finite loss


@pytest.mark.parametrize("world", [1, 2])
def test_graphed_train_step_under_dp_equals_eager_dp(tmp_path, world):
    """train_epoch + GraphedTrainStep under data parallelism: the replayed step contains the all-reduce.  World 1 runs it over a
    real RCCL process group (PFN_FORCE_DIST=1): ONE hipGraph with ncclAllReduce captured between backward and AdamW.  World 2
    (two ranks on the one GPU, gloo): graph / eager all-reduce / graph.  Either way two epochs end on the parameters of the
    eager data-parallel loop, bit for bit, and the replicas agree."""
    out_path = str(tmp_path / "dp_train.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_raw_dataset.py"), "--root", str(tmp_path), "--case", "14",
                    "--samples", "60"], check=True, capture_output=True)        # train split: 30 samples = 5 global batches of 6
    script = [os.path.join(ROOT, "tests", "dp_train_worker.py"), out_path, str(tmp_path), "14", "6"]
    if world == 1:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PFN_HANG_DUMP="150")
        cmd = [sys.executable] + script
    else:
        cmd, env = _torchrun(2, script, {"PFN_SINGLE_DEVICE": "1", "PFN_DIST_BACKEND": "gloo"})
    _run(cmd, env, 420)
    got = json.load(open(out_path))
    want_mode = "one graph incl. RCCL all-reduce" if world == 1 else "graph(fwd+bwd) -> eager all-reduce -> graph(optimizer)"
    assert got["mode"] == want_mode and got["world"] == world, got
    assert got["max_abs_diff"] == 0.0, got
    assert got["losses_graph"] == got["losses_eager"], got
    # --dp-mode / PFN_DP_MODE: every form asked for by name lands where it should (a backend that cannot be captured turns
    # "graph" into "split" on every rank alike) and all of them give the eager loop's parameters bit for bit
    want_forms = {"graph": "graph" if world == 1 else "split", "split": "split", "eager": "eager"}
    for name, rec in got["forms"].items():
        assert rec["form"] == want_forms[name] and rec["max_abs_diff"] == 0.0 and rec["losses_equal"], (name, rec)


def test_guarded_update_is_rank_consistent_under_dp(tmp_path):
    """with per-batch topologies the captured step guards its AdamW update on the loss; under data parallelism a bad
    batch on ONE rank reaches the others only as NaN gradients through the all-reduce.  The guard is therefore the rank-SUMMED
    loss (all-reduced next to the gradients): every rank skips that update, replicas stay bit-identical and finite, and the skip
    is counted in step_count[2] on every rank.  Two ranks on the one GPU (gloo), the last rank poisons its batch at step 2."""
    out_path = str(tmp_path / "dp_guard.json")
    script = [os.path.join(ROOT, "tests", "dp_guard_worker.py"), out_path, "2", "5"]
    cmd, env = _torchrun(2, script, {"PFN_SINGLE_DEVICE": "1", "PFN_DIST_BACKEND": "gloo"})
    _run(cmd, env, 420)
    got = json.load(open(out_path))
    assert got["replicas_equal"] and got["finite"], got
    assert got["step_counts"] == [[4, 0, 1], [4, 0, 1]], got          # four updates applied, one skipped -- on BOTH ranks
    assert all(l == l for l in got["losses_rank0"]), got               # rank 0's own losses were finite all along


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
    # the collective is INSIDE the replayed graph (dp.GraphedStep): one hipGraph per data-parallel step
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["launch"] == "hipGraph replay: one graph incl. RCCL all-reduce", d["config"]


def test_bench_under_torchrun_world1_with_rccl_overhead_probe():
    """The driver's launch form at ONE rank (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`): no process
    group is needed, but the `dp_overhead_ms` probe creates a private world-size-1 RCCL group.  Under torchrun
    (TORCHELASTIC_USE_AGENT_STORE) a `tcp://` init made the rank a CLIENT of a store nobody serves and the bench sat in
    init_process_group for half an hour; the probe now brings its own TCPStore.  One JSON line, the probe's numbers in it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PFN_HANG_DUMP="150")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PFN_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--profile-steps", "2", "--no-cpu-baseline", "--no-live-traffic", "--case", "14", "--batch", "16"]
    out, err = _run(cmd, env, 300)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:] + err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0, d
    assert d.get("dp_graph_mode") == "one graph incl. RCCL all-reduce" and "dp_overhead_error" not in d, d
