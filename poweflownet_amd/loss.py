"""`MSELoss` and `Masked_L2_loss` with forward and backward fused into one pass over (out, y) (`pfn_mse_loss`,
`pfn_masked_l2_loss`).

Counterpart of `torch.nn.MSELoss()` at train.py:103 as used by the else-branch of train_epoch
(utils/training.py:72): loss = mean((out - y)^2); the gradient 2 (out - y) / numel is produced by the same
kernel that accumulates the loss, so `loss.backward()` costs one scale instead of torch's four small kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L


_ONE = {}     # per device: the constant 1 used as the unit loss gradient (immutable after creation)


def _one(device):
    key = (device.type, device.index)
    if key not in _ONE:
        _ONE[key] = torch.ones((), dtype=torch.float32, device=device)
    return _ONE[key]


class _Workspace:
    """Reduction scratch of ONE loss object (partials + an arrival counter that is zero between calls), per device.  Owned
    by the loss module that uses it -- two models / host threads / streams bring two loss objects and share nothing; one
    object serves one call at a time (the rule of pfn_context, include/pfn_hip.h).  Allocated on the first call, i.e. in
    the warm-up steps that precede a hipGraph capture, never inside one."""

    def __init__(self, floats: int):
        self.floats, self._buf = floats, {}

    def on(self, device) -> torch.Tensor:
        key = (device.type, device.index)
        if key not in self._buf:
            self._buf[key] = torch.zeros(self.floats, dtype=torch.float32, device=device)
        return self._buf[key]

    def __reduce__(self):
        return (_Workspace, (self.floats,))


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, y, wsp):
        L.require_device(out, y, what="MSELoss input")
        out, y = L.f32c(out, "out"), L.f32c(y, "y")
        if out.shape != y.shape:
            raise RuntimeError(f"MSELoss: shape mismatch {tuple(out.shape)} vs {tuple(y.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ws = wsp.on(out.device)
        with torch.cuda.device(out.device):
            L.check(L.load().pfn_mse_loss(out.data_ptr(), y.data_ptr(), out.numel(), loss.data_ptr(), L.ptr(grad),
                                          ws.data_ptr(), ws.numel() * 4, L.stream_ptr()), "pfn_mse_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grad is None:
            return None, None, None
        if gloss.data_ptr() == _one(ctx.grad.device).data_ptr():   # MSELoss.unit_grad(): the constant 1, never written
            return ctx.grad, None, None
        return ctx.grad * gloss, None, None


class MseTail:
    """What `MSELoss.attach` arranged for ONE forward/backward pair of a model (pfn_mpn_backward_mse): the model left its output
    rows unwritten; its backward pass writes `out`, `loss` and `grad_out` in its first launch."""
    __slots__ = ("target", "target_version", "loss", "grad_out", "ws", "masked", "mask")

    def __init__(self, target, loss, grad_out, ws, masked=None, mask=None):
        self.target, self.target_version, self.loss, self.grad_out, self.ws = target, target._version, loss, grad_out, ws
        self.masked, self.mask = masked, mask      # Masked_L2_loss: (regularize, regcoeff) and the mask tensor the model was given


class _MseTailFn(torch.autograd.Function):
    """The loss node of an attached pair: nothing is launched here.  forward hands out the tensor the model's backward pass will
    write the loss into; backward hands the model the (still unwritten) grad_out buffer as the token that says "form it yourself"."""

    @staticmethod
    def forward(ctx, out, tail):
        ctx.tail = tail
        return tail.loss.detach()      # (an alias: returning tail.loss itself would tie loss -> grad_fn -> tail -> loss into a cycle)

    @staticmethod
    def backward(ctx, gloss):
        tail = ctx.tail
        if gloss.data_ptr() != _one(tail.grad_out.device).data_ptr():
            raise RuntimeError("MSELoss.attach(): the attached loss must be differentiated with loss.backward(MSELoss.unit_grad(loss)) "
                               "(a scaled loss needs the plain path: do not call attach)")
        return tail.grad_out, None


class MSELoss(nn.Module):
    """Drop-in for `torch.nn.MSELoss()` (reduction='mean') on HIP tensors."""

    def __init__(self):
        super().__init__()
        self._ws = _Workspace(264)
        self._tail_ws = _Workspace(1028)     # pfn_mpn_backward_mse: 1024 partials + the arrival counter

    def attach(self, model, target):
        """Promise of the training loop, made right before `out = model(data)`: the next three statements are
        `loss = self(out, target)`, `loss.backward(self.unit_grad(loss))`, and nobody reads `out` or `loss` before that backward
        has run.  A model that can (`MaskEmbdMultiMPN` on a batch of small graphs, output_dim 4: pfn_mpn_mse_tail_ok) then leaves
        its output Linear to its backward pass, whose first launch forms `out`, the loss and its gradient -- two launches fewer
        per step (train_epoch's per-batch body, utils/training.py:55-77, is exactly this sequence).  One-shot: consumed by the
        next forward whether it could use it or not; where it could not, nothing changes.  Results: `out` and every gradient bit
        for bit those of the plain path, the loss to the rounding of another summation order."""
        if hasattr(model, "_mse_attach"):
            model._mse_attach = (target, self._tail_ws, None, None)

    def forward(self, input, target):
        tail = getattr(input, "_pfn_mse_tail", None)
        if tail is not None:
            if not (torch.is_tensor(target) and tail.target is target and tail.target_version == target._version
                    and input.shape == target.shape and tail.masked is None):
                raise RuntimeError("MSELoss.attach(): the loss was called with another target (or a modified one) than the one "
                                   "attached -- the model's output rows are not written on this path")
            return _MseTailFn.apply(input, tail)
        return _MseFn.apply(input, target, self._ws)

    @staticmethod
    def unit_grad(loss):
        """The constant 1 on `loss`'s device.  `loss.backward(MSELoss.unit_grad(loss))` is `loss.backward()` without the
        two tiny kernels autograd spends on creating that 1 and multiplying the gradient by it."""
        return _one(loss.device)


class _MaskedL2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, y, mask, regularize, regcoeff, wsp):
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
        ws = wsp.on(out.device)
        with torch.cuda.device(out.device):
            L.check(L.load().pfn_masked_l2_loss(out.data_ptr(), y.data_ptr(), mask.data_ptr(), code, out.numel(), int(bool(regularize)),
                                                float(regcoeff), loss.data_ptr(), L.ptr(grad), ws.data_ptr(), ws.numel() * 4,
                                                L.stream_ptr()), "pfn_masked_l2_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.grad is None:
            return None, None, None, None, None, None
        if gloss.data_ptr() == _one(ctx.grad.device).data_ptr():
            return ctx.grad, None, None, None, None, None
        return ctx.grad * gloss, None, None, None, None, None


MASKED_L2_WS_FLOATS = 1032
POWER_IMBALANCE_WS_FLOATS = 320


def masked_l2_attach(model, target, mask, regularize, regcoeff, tail_ws):
    """`Masked_L2_loss.attach`: the promise of `MSELoss.attach` for the reference's default training loss
    (utils/custom_loss_functions.py:10-46; dispatch utils/training.py:61-62).  `mask` must be the very tensor the model is about to
    read as `data.pred_mask` (its first launch counts the two index sets of the loss while it converts the mask)."""
    if hasattr(model, "_mse_attach"):
        model._mse_attach = (target, tail_ws, (bool(regularize), float(regcoeff)), mask)


def masked_l2_loss(output, target, mask, regularize=True, regcoeff=1, workspace=None):
    """Masked_L2_loss.forward (utils/custom_loss_functions.py:30-46) on HIP tensors: loss and its gradient in two launches
    instead of four `masked_select` compactions, two means and their autograd graph.  `workspace`: the calling loss
    object's `_Workspace(MASKED_L2_WS_FLOATS)` (a throw-away one is made when omitted)."""
    tail = getattr(output, "_pfn_mse_tail", None)
    if tail is not None:
        if not (tail.masked == (bool(regularize), float(regcoeff)) and tail.target is target and tail.target_version == target._version
                and tail.mask is mask and output.shape == target.shape):
            raise RuntimeError("Masked_L2_loss.attach(): the loss was called with another target / mask / setting than the one "
                               "attached -- the model's output rows are not written on this path")
        return _MseTailFn.apply(output, tail)
    return _MaskedL2Fn.apply(output, target, mask, regularize, regcoeff, workspace or _Workspace(MASKED_L2_WS_FLOATS))


def unit_grad(loss):
    """The constant 1 on `loss`'s device: `loss.backward(unit_grad(loss))` == `loss.backward()` minus two tiny kernels."""
    return _one(loss.device)


class _PowerImbalanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph, edge_attr, stats, wsp):
        L.require_device(x, edge_attr, what="PowerImbalance input")
        x, edge_attr = L.f32c(x, "x"), L.f32c(edge_attr, "edge_attr")
        n = x.shape[0]
        if x.dim() != 2 or x.shape[1] != 4 or edge_attr.shape != (graph.e_stored, 2) or graph.num_nodes != n:
            raise RuntimeError(f"PowerImbalance: x must be (N, 4) and edge_attr (E, 2); got {tuple(x.shape)}, {tuple(edge_attr.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dpq = torch.empty(max(n, 1), 2, dtype=torch.float32, device=x.device)
        ws = wsp.on(x.device)
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
            return None, None, None, None, None
        if gloss.data_ptr() == _one(ctx.grad.device).data_ptr():
            return ctx.grad, None, None, None, None
        return ctx.grad * gloss, None, None, None, None


def power_imbalance(x, graph, edge_attr, stats, workspace=None):
    """PowerImbalance.forward on HIP tensors; `graph` = the GraphCSR of the stored-once edge_index (mode -1), `stats` = 12
    floats {xymean[4], xystd[4], edgemean[2], edgestd[2]}; `workspace` as in masked_l2_loss."""
    return _PowerImbalanceFn.apply(x, graph, edge_attr, stats, workspace or _Workspace(POWER_IMBALANCE_WS_FLOATS))
