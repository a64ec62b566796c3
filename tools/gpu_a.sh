#!/bin/bash
# GPU session A (round 2): full -m gpu suite, DP hang hunt, baseline benches with and without the side stream.
cd $GRAFT_REPO_ROOT
O=gpurun_out/a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/pytest.txt
echo "pytest rc=$?" >> $O/pytest.txt
# hang hunt: the two-rank bench, 6 times, each under its own timeout with a stack dump at 100 s
for i in 1 2 3 4 5 6; do
  P=$((29600 + i))
  PFN_HANG_DUMP=100 PFN_SINGLE_DEVICE=1 PFN_DIST_BACKEND=gloo timeout -k 5 160 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
     --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 4 --warmup 2 --profile-steps 2 --no-cpu-baseline --case 14 --batch 8 \
     > $O/hunt_$i.out 2> $O/hunt_$i.err
  echo "hunt $i rc=$? $(date +%s)" >> $O/hunt.txt
done
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
PFN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline > $O/b2_noside.json 2> $O/b2_noside.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
PFN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_noside.json 2> $O/b4_noside.err
python bench.py --no-cpu-baseline --mode infer --batch 2048 --steps 20 --warmup 5 > $O/b3.json 2> $O/b3.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2 > $O/b4_hub.json 2> $O/b4_hub.err
ls -la $O
