"""`MSELoss` with forward and backward fused into one pass over (out, y) (`pfn_mse_loss`).

Counterpart of `torch.nn.MSELoss()` at train.py:103 as used by the else-branch of train_epoch
(utils/training.py:72): loss = mean((out - y)^2); the gradient 2 (out - y) / numel is produced by the same
kernel that accumulates the loss, so `loss.backward()` costs one scale instead of torch's four small kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, y):
        L.require_device(out, y, what="MSELoss input")
        out, y = L.f32c(out, "out"), L.f32c(y, "y")
        if out.shape != y.shape:
            raise RuntimeError(f"MSELoss: shape mismatch {tuple(out.shape)} vs {tuple(y.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ws = torch.empty(256, dtype=torch.float32, device=out.device)
        with torch.cuda.device(out.device):
            L.check(L.load().pfn_mse_loss(out.data_ptr(), y.data_ptr(), out.numel(), loss.data_ptr(), L.ptr(grad),
                                          ws.data_ptr(), ws.numel() * 4, L.stream_ptr()), "pfn_mse_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        return (ctx.grad * gloss if ctx.grad is not None else None), None


class MSELoss(nn.Module):
    """Drop-in for `torch.nn.MSELoss()` (reduction='mean') on HIP tensors."""

    def forward(self, input, target):
        return _MseFn.apply(input, target)
