"""AdamW over ONE flat fp32 buffer (a single HIP kernel launch per step).

Counterpart of `torch.optim.AdamW(model.parameters(), lr=lr)` at train.py:123 of the reference (torch defaults:
betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, decoupled decay).  The model's parameters are re-pointed at
views of one flat buffer (same order as the C ABI's parameter table), the HIP backward already writes all
gradients into one flat buffer in that order, so the update is `pfn_adamw_step(flat_param, flat_grad, m, v)`:
hipGraph-capturable (the step counter lives on the device), no per-tensor kernels.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .dp import _grads_are_views_of


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = model._ordered_params() if hasattr(model, "_ordered_params") else list(model.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("FlatAdamW: move the model to its HIP device first (there is no CPU path)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model, self._params = model, params
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
            off = 0
            for p in params:
                p.data = flat[off:off + p.numel()].view(p.shape)     # parameters become views of the flat buffer
                off += p.numel()
        self.flat_param = flat
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = torch.zeros(3, dtype=torch.int64, device=flat.device)   # {steps done, arrival scratch, skipped (guard)}
        self._gather = None
        # optional device scalar (a loss): the update is SKIPPED on the device when it is not finite -- set by a caller whose
        # step may be fed a batch it could not validate on the host (GraphedTrainStep with per-batch topologies)
        self.guard = None
        # {lr, beta1, beta2, eps, weight_decay} live on the device: the update kernel reads them there, so a captured step
        # follows a scheduler (OneCycleLR moves lr AND beta1) through a 20-byte copy instead of a re-capture
        self.hyper = torch.zeros(5, dtype=torch.float32, device=flat.device)
        self._hyper_host = None
        self.sync_hyper()

    def _hyper_now(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def sync_hyper(self) -> None:
        """Push the param group's scalars to the device when they changed (call it outside a stream capture: the captured
        step itself contains no copy)."""
        now = self._hyper_now()
        if now != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamW: hyper-parameters changed inside a stream capture")
            self.hyper.copy_(torch.tensor(now, dtype=torch.float32))
            self._hyper_host = now

    def _flat_grad(self) -> torch.Tensor:
        fg = self._model.flat_grad() if hasattr(self._model, "flat_grad") else None
        if fg is not None and _grads_are_views_of(fg, self._params):   # (every parameter, every call: ~15 us)
            return fg
        # generic path (gradients accumulated elsewhere): gather into a scratch buffer
        if self._gather is None:
            self._gather = torch.empty_like(self.flat_param)
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is None:
                self._gather[off:off + n].zero_()
            else:
                self._gather[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        return self._gather

    def skipped_steps(self) -> int:
        """Updates the device-side guard has skipped so far (a host read: call it once per epoch, not per step)."""
        return int(self.step_count[2].item())

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        grad = self._flat_grad()
        with torch.cuda.device(self.flat_param.device):
            self.sync_hyper()
            if self.guard is not None:
                L.check(L.load().pfn_adamw_step_guarded(self.flat_param.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                                                        self.exp_avg_sq.data_ptr(), self.flat_param.numel(), self.hyper.data_ptr(),
                                                        self.step_count.data_ptr(), self.guard.data_ptr(), L.stream_ptr()),
                        "pfn_adamw_step_guarded")
            else:
                L.check(L.load().pfn_adamw_step_dev(self.flat_param.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                                                    self.exp_avg_sq.data_ptr(), self.flat_param.numel(), self.hyper.data_ptr(),
                                                    self.step_count.data_ptr(), L.stream_ptr()), "pfn_adamw_step_dev")
        return loss
