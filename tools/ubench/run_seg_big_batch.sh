#!/bin/bash
R=$GRAFT_REPO_ROOT
export ASYNC_CHECK=$R/tools/check_async_fragments.py
d=/tmp/exp_segbig; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $d/ 2>/dev/null
sed -i 's/const long per_cu = 4L;/const long per_cu = 64L;/' $d/poweflownet_amd/csrc/seg_lin_hops.hip $d/poweflownet_amd/csrc/ea_seg.hip
grep -n "per_cu = " $d/poweflownet_amd/csrc/*.hip
( cd $d/poweflownet_amd/csrc && rm -f seg_lin_hops.o ea_seg.o libpfn_hip.so && make -j16 libpfn_hip.so > /tmp/sb_make.log 2>&1 ) || tail -5 /tmp/sb_make.log
cd $d && python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step']); [print(k, v.get('launches_per_step'), v.get('avg_us')) for k,v in d['kernels'].items()]"
