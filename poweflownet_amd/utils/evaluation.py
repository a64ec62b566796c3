"""`evaluate_epoch` / `load_model` / `num_params` counterparts of the reference's utils/evaluation.py:20-104."""
from typing import Callable, Optional

import torch
import torch.nn as nn

from .custom_loss_functions import Masked_L2_loss, MixedMSEPoweImbalance, PowerImbalance


def load_model(model: nn.Module, run_id: str, device, models_dir: str = "models"):
    """utils/evaluation.py:20-36: load `model_state_dict` of the best-validation checkpoint of a run."""
    import os
    path = os.path.join(models_dir, f"model_{run_id}.pt")
    # a reference checkpoint stores `args` as an argparse.Namespace (reference train.py:160-165), which torch's default
    # weights_only unpickler rejects; these are the user's own local training artefacts, exactly what the reference loads
    saved = torch.load(path, map_location=device, weights_only=False)
    model.load_state_dict(saved["model_state_dict"])
    return model, saved


def num_params(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


@torch.no_grad()
def evaluate_epoch(model: nn.Module, loader, loss_fn: Callable, device="cpu", pre_loss_fn: Optional[Callable] = None) -> float:
    pre = pre_loss_fn or (lambda x: x)
    model.eval()
    total_loss, num_samples = 0.0, 0
    for data in loader:
        data = data.to(device)
        out = model(data)
        if isinstance(loss_fn, Masked_L2_loss):
            loss = loss_fn(pre(out), pre(data.y), data.pred_mask)
        elif isinstance(loss_fn, PowerImbalance):
            # (sic) the reference adds pred_mask * (1 - pred_mask), which is zero for 0/1 masks (:88-89); evaluate_epoch_v2 uses x
            masked_out = out * data.pred_mask + data.pred_mask * (1 - data.pred_mask)
            loss = loss_fn(pre(masked_out), data.edge_index, data.edge_attr)
        elif isinstance(loss_fn, MixedMSEPoweImbalance):
            loss = loss_fn(pre(out), data.edge_index, data.edge_attr, data.y)
        else:
            loss = loss_fn(pre(out), pre(data.y))
        num_samples += len(data)
        total_loss += loss.item() * len(data)
    return total_loss / max(num_samples, 1)


@torch.no_grad()
def evaluate_epoch_v2(model: nn.Module, loader, loss_fn: Callable, device="cpu", pre_loss_fn: Optional[Callable] = None) -> dict:
    """utils/evaluation.py:106-165: like `evaluate_epoch` but returns a dict of loss terms (`MaskedL2V2` / `MaskedL1` produce
    several; every other loss one, under 'total'; `PowerImbalance` adds 'ref' = the loss of the ground truth).

    Kept quirk (:158-163): the FIRST batch enters the running sums unweighted, later batches weighted by len(data); the sums
    are divided by the total of len(data)."""
    from .custom_loss_functions import MaskedL1, MaskedL2V2
    pre = pre_loss_fn or (lambda x: x)
    model.eval()
    totals, num_samples = None, 0
    for data in loader:
        data = data.to(device)
        out = model(data)
        if isinstance(loss_fn, Masked_L2_loss):
            terms = {"total": loss_fn(pre(out), pre(data.y), data.pred_mask)}
        elif isinstance(loss_fn, (MaskedL2V2, MaskedL1)):
            terms = loss_fn(pre(out), pre(data.y), data.pred_mask)
        elif isinstance(loss_fn, PowerImbalance):
            masked_out = pre(out * data.pred_mask + data.x * (1 - data.pred_mask))
            terms = {"total": loss_fn(masked_out, data.edge_index, data.edge_attr),
                     "ref": loss_fn(data.y, data.edge_index, data.edge_attr)}
        elif isinstance(loss_fn, MixedMSEPoweImbalance):
            terms = {"total": loss_fn(pre(out), data.edge_index, data.edge_attr, data.y)}
        else:
            terms = {"total": loss_fn(pre(out), pre(data.y))}
        num_samples += len(data)
        if totals is None:
            totals = {k: v.item() for k, v in terms.items()}
        else:
            totals = {k: v + terms[k].item() * len(data) for k, v in totals.items()}
    return {k: v / num_samples for k, v in (totals or {}).items()}
