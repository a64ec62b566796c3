#!/bin/bash
# round 6, GPU session 2: the new tests, bench with the new fields, gemm_nt phase stamps / harness at config-2 size, per-launch
# gemm_nt durations inside a replayed config-3 forward
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_torch_ops.py tests/test_gpu_parity.py -m gpu -q -x -k "bad_graph or rebuilt_on_the_device or mse_tail_ops or wide_json_on_case6470 or graphed_train_step or dynamic_topology or guarded" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python bench.py --no-cpu-baseline --no-live-traffic > $O/b2.json 2> $O/b2.err
bash tools/ubench/run_gemm_nt_small.sh > $O/nt_small.txt 2>&1
bash tools/ubench/run_gemm_nt_ts.sh > $O/nt_ts.txt 2>&1
bash tools/ubench/run_gemm_nt_ts2_small.sh > $O/nt_ts2_small.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof3; mkdir -p /tmp/prof3
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof3 -o t -- python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 2048 --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-live-traffic --no-other-configs > /tmp/prof3/bench.out 2>&1
python $GRAFT_REPO_ROOT/tools/ktimeline.py /tmp/prof3/t_results.db 40 > $GRAFT_REPO_ROOT/$O/timeline_c3.txt 2>&1
rm -rf /tmp/prof4; mkdir -p /tmp/prof4
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof4 -o t -- python $GRAFT_REPO_ROOT/bench.py --case 6470rte --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-live-traffic --no-other-configs > /tmp/prof4/bench.out 2>&1
python $GRAFT_REPO_ROOT/tools/ktimeline.py /tmp/prof4/t_results.db 70 > $GRAFT_REPO_ROOT/$O/timeline_c4.txt 2>&1
