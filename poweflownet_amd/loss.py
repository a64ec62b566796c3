"""`MSELoss` and `Masked_L2_loss` with forward and backward fused into one pass over (out, y) (`pfn_mse_loss`,
`pfn_masked_l2_loss`).

Counterpart of `torch.nn.MSELoss()` at train.py:103 as used by the else-branch of train_epoch
(utils/training.py:72): loss = mean((out - y)^2); the gradient 2 (out - y) / numel is produced by the same
kernel that accumulates the loss, so `loss.backward()` costs one scale instead of torch's four small kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L


_WS = {}      # per device: (workspace with the zeroed arrival counter, the constant 1 used as the unit loss gradient)


def _state(device):
    key = (device.type, device.index)
    if key not in _WS:
        _WS[key] = (torch.zeros(264, dtype=torch.float32, device=device), torch.ones((), dtype=torch.float32, device=device),
                    torch.zeros(1032, dtype=torch.float32, device=device), torch.zeros(320, dtype=torch.float32, device=device))
    return _WS[key]


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, y):
        L.require_device(out, y, what="MSELoss input")
        out, y = L.f32c(out, "out"), L.f32c(y, "y")
        if out.shape != y.shape:
            raise RuntimeError(f"MSELoss: shape mismatch {tuple(out.shape)} vs {tuple(y.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ws = _state(out.device)[0]
        with torch.cuda.device(out.device):
            L.check(L.load().pfn_mse_loss(out.data_ptr(), y.data_ptr(), out.numel(), loss.data_ptr(), L.ptr(grad),
                                          ws.data_ptr(), ws.numel() * 4, L.stream_ptr()), "pfn_mse_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grad is None:
            return None, None
        if gloss.data_ptr() == _state(ctx.grad.device)[1].data_ptr():   # MSELoss.unit_grad(): the constant 1, never written
            return ctx.grad, None
        return ctx.grad * gloss, None


class MSELoss(nn.Module):
    """Drop-in for `torch.nn.MSELoss()` (reduction='mean') on HIP tensors."""

    def forward(self, input, target):
        return _MseFn.apply(input, target)

    @staticmethod
    def unit_grad(loss):
        """The constant 1 on `loss`'s device.  `loss.backward(MSELoss.unit_grad(loss))` is `loss.backward()` without the
        two tiny kernels autograd spends on creating that 1 and multiplying the gradient by it."""
        return _state(loss.device)[1]


class _MaskedL2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, y, mask, regularize, regcoeff):
        L.require_device(out, y, mask, what="Masked_L2_loss input")
        out, y = L.f32c(out, "output"), L.f32c(y, "target")
        if out.shape != y.shape or mask.shape != out.shape:
            raise RuntimeError(f"Masked_L2_loss: shape mismatch {tuple(out.shape)} / {tuple(y.shape)} / {tuple(mask.shape)}")
        if mask.dtype == torch.int64:
            code = 0
        else:
            mask, code = mask.to(torch.float32), 1
        mask = mask.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ws = _state(out.device)[2]
        with torch.cuda.device(out.device):
            L.check(L.load().pfn_masked_l2_loss(out.data_ptr(), y.data_ptr(), mask.data_ptr(), code, out.numel(), int(bool(regularize)),
                                                float(regcoeff), loss.data_ptr(), L.ptr(grad), ws.data_ptr(), ws.numel() * 4,
                                                L.stream_ptr()), "pfn_masked_l2_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grad is None:
            return None, None, None, None, None
        if gloss.data_ptr() == _state(ctx.grad.device)[1].data_ptr():
            return ctx.grad, None, None, None, None
        return ctx.grad * gloss, None, None, None, None


def masked_l2_loss(output, target, mask, regularize=True, regcoeff=1):
    """Masked_L2_loss.forward (utils/custom_loss_functions.py:30-46) on HIP tensors: loss and its gradient in two launches
    instead of four `masked_select` compactions, two means and their autograd graph."""
    return _MaskedL2Fn.apply(output, target, mask, regularize, regcoeff)


def unit_grad(loss):
    """The constant 1 on `loss`'s device: `loss.backward(unit_grad(loss))` == `loss.backward()` minus two tiny kernels."""
    return _state(loss.device)[1]


class _PowerImbalanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph, edge_attr, stats):
        L.require_device(x, edge_attr, what="PowerImbalance input")
        x, edge_attr = L.f32c(x, "x"), L.f32c(edge_attr, "edge_attr")
        n = x.shape[0]
        if x.dim() != 2 or x.shape[1] != 4 or edge_attr.shape != (graph.e_stored, 2) or graph.num_nodes != n:
            raise RuntimeError(f"PowerImbalance: x must be (N, 4) and edge_attr (E, 2); got {tuple(x.shape)}, {tuple(edge_attr.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dpq = torch.empty(max(n, 1), 2, dtype=torch.float32, device=x.device)
        ws = _state(x.device)[3]
        import ctypes as C
        st = (C.c_float * 12)(*stats)
        with torch.cuda.device(x.device):
            L.check(L.load().pfn_power_imbalance(graph.ws.data_ptr(), n, graph.e_stored, x.data_ptr(), edge_attr.data_ptr(), st,
                                                 loss.data_ptr(), L.ptr(grad), dpq.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                 L.stream_ptr()), "pfn_power_imbalance")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grad is None:
            return None, None, None, None
        if gloss.data_ptr() == _state(ctx.grad.device)[1].data_ptr():
            return ctx.grad, None, None, None
        return ctx.grad * gloss, None, None, None


def power_imbalance(x, graph, edge_attr, stats):
    """PowerImbalance.forward on HIP tensors; `graph` = the GraphCSR of the stored-once edge_index (mode -1), `stats` = 12
    floats {xymean[4], xystd[4], edgemean[2], edgestd[2]}."""
    return _PowerImbalanceFn.apply(x, graph, edge_attr, stats)
