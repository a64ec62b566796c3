#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# train.py end to end (the reference's loop shape, its file format, the device-resident dataset, one hipGraph launch per batch):
# graphs/s per epoch for the three loss functions at case118v2 x 128, 8,000 samples (4,000 in the training split)
cd $GRAFT_REPO_ROOT
D=/tmp/pfdata; rm -rf $D
python tools/make_raw_dataset.py --root $D --case 118v2 --samples 8000 > /dev/null
for lossfn in mse_loss masked_l2 mixed_mse_power_imbalance; do
  echo "== --train_loss_fn $lossfn"
  timeout 600 python train.py --cfg_json configs/standard.json --case 118v2 --data-dir $D --num-epochs 6 --batch-size 128 --train_loss_fn $lossfn --no-save 2>&1 | grep -i "epoch\|error\|Traceback" | tail -7
done
