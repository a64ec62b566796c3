"""`train_epoch` counterpart of the reference's utils/training.py:30-80: same signature, same per-batch order
(to(device) -> zero_grad -> forward -> loss dispatch by isinstance -> backward -> step -> loss.item()), same return
value (sum(loss * len(data)) / sum(len(data))).  `allreduce` plugs in the one-collective DP step (dp.py)."""
import json
import os
from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import dp
from .custom_loss_functions import Masked_L2_loss, MixedMSEPoweImbalance, PowerImbalance


def append_to_json(log_path, run_id, result):
    """utils/training.py:15-27: merge {run_id: result} into a JSON log file."""
    os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
    try:
        with open(log_path, "r") as f:
            log = json.load(f)
    except FileNotFoundError:
        log = {}
    log.update({str(run_id): result})
    with open(log_path, "w") as f:
        json.dump(log, f, indent=4)


def train_epoch(model: nn.Module, loader, loss_fn: Callable, optimizer, device, progress: bool = False,
                allreduce: Optional[bool] = None) -> float:
    model = model.to(device)
    total_loss, num_samples = 0.0, 0
    model.train()
    if allreduce is None:
        allreduce = dp.world_size() > 1
    it = loader
    if progress:
        from tqdm import tqdm
        it = tqdm(loader, total=len(loader), desc="Training")
    for data in it:
        data = data.to(device)
        optimizer.zero_grad()
        out = model(data)
        if isinstance(loss_fn, Masked_L2_loss):
            loss = loss_fn(out, data.y, data.pred_mask)
        elif isinstance(loss_fn, PowerImbalance):
            masked_out = out * data.pred_mask + data.x * (1 - data.pred_mask)
            loss = loss_fn(masked_out, data.edge_index, data.edge_attr)
        elif isinstance(loss_fn, MixedMSEPoweImbalance):
            loss = loss_fn(out, data.edge_index, data.edge_attr, data.y)
        else:
            loss = loss_fn(out, data.y)
        if hasattr(loss_fn, "unit_grad"):
            loss.backward(loss_fn.unit_grad(loss))   # same as loss.backward() (utils/training.py:74), two tiny kernels fewer
        else:
            loss.backward()
        if allreduce:
            dp.allreduce_gradients(model)
        optimizer.step()
        num_samples += len(data)
        total_loss += loss.item() * len(data)
    return total_loss / max(num_samples, 1)
