"""The multi-process path of bench.py on a ONE-GPU box: two ranks share device 0 and exchange gradients over gloo
(PFN_SINGLE_DEVICE / PFN_DIST_BACKEND test aids of poweflownet_amd/dp.py).  Guards the collective sequence -- every rank
must enter every all-reduce, including the ones inside rank 0's profiling pass -- which RCCL would turn into a hang."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.environ.get("PFN_TEST_DP_ONE_GPU"),
                    reason="opt-in (PFN_TEST_DP_ONE_GPU=1): two processes time-sharing one GPU over gloo hung once in five runs "
                           "on a fresh box (rendezvous / cold start, not reproduced); not worth a 10-minute stall in a routine run")
def test_bench_two_ranks_on_one_gpu_gloo():
    env = dict(os.environ, PFN_SINGLE_DEVICE="1", PFN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--profile-steps", "2", "--no-cpu-baseline", "--case", "14", "--batch", "8"]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)        # the launcher AND its ranks: nothing may be left holding the GPU
        proc.communicate()
        pytest.fail("two-rank bench did not finish within 240 s")
    assert proc.returncode == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                             # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"] is not None
