#!/bin/bash
# end-to-end check of the callers after the round-2 kernel work: train.py on raw-format data (graphed step), both losses,
# a sibling model's refusal of 4-wide data, and the full GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest.txt
python tools/make_raw_dataset.py --root /tmp/pfdata --case 118v2 --samples 8000 > $O/train.txt 2>&1
python train.py --cfg_json configs/standard.json --case 118v2 --data-dir /tmp/pfdata --train_loss_fn masked_l2 --num-epochs 4 --no-save >> $O/train.txt 2>&1
python train.py --cfg_json configs/standard.json --case 118v2 --data-dir /tmp/pfdata --train_loss_fn mse_loss --num-epochs 3 --no-save >> $O/train.txt 2>&1
python train.py --cfg_json configs/standard.json --case 118v2 --data-dir /tmp/pfdata --train_loss_fn mixed_mse_power_imbalance --num-epochs 2 --no-save >> $O/train.txt 2>&1
python train.py --cfg_json configs/standard.json --case 118v2 --data-dir /tmp/pfdata --model MultiMPN --num-epochs 1 --no-save >> $O/train.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/train.txt 2>&1
