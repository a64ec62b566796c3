"""Batched-graph data parallelism: one process per GPU, one collective per step.

The reference is single-process (train.py:69); this layer is new.  Graphs of a batch are independent
components, so rank r simply takes graphs r::world of every global batch (DataLoader(shard=...)), parameters are
replicated (identical seed / broadcast), and after backward ONE all-reduce averages the flat fp32 gradient buffer
that `MaskEmbdMultiMPN`'s backward wrote (354,500 floats = 1.418 MB for standard.json): RCCL over xGMI on the
GPUs (`backend="nccl"` is RCCL on ROCm), gloo in the CPU plumbing tests.  With MSELoss(mean) and equal node counts
per graph the averaged gradient equals the single-device gradient on the global batch.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sys
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test aids for a ONE-GPU box (the multi-process code path of bench.py / train.py cannot be exercised there otherwise):
    # PFN_SINGLE_DEVICE=1 puts every rank on device 0, PFN_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks per GPU)
    if os.environ.get("PFN_SINGLE_DEVICE"):
        local_rank = 0
    backend = backend or os.environ.get("PFN_DIST_BACKEND")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    # PFN_FORCE_DIST=1: initialise the process group even at world size 1, so the collective code path (RCCL all-reduce of
    # the flat gradient buffer between two hipGraph replays) can be exercised on a one-GPU box
    if (world > 1 or os.environ.get("PFN_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        with stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on the C-level stdout when a communicator is created.  A caller whose stdout is a protocol
    (bench.py: ONE JSON line) wraps the process-group / communicator creation in this: file descriptor 1 points at stderr for
    the duration, and the C stdio buffer is flushed before it is restored."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                          # noqa: BLE001
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active() -> bool:
    """A process group exists (normally world > 1): steps end in the gradient all-reduce."""
    return dist.is_available() and dist.is_initialized()


def broadcast_parameters(model: torch.nn.Module, src: int = 0) -> None:
    """Replicate rank `src`'s parameters (one flat broadcast)."""
    if not active():
        return
    params = [p.data for p in model.parameters()]
    flat = torch.cat([p.reshape(-1) for p in params])
    dist.broadcast(flat, src)
    off = 0
    for p in params:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()


def _grads_are_views_of(flat: torch.Tensor, params: Iterable[torch.nn.Parameter]) -> bool:
    """Is EVERY parameter's `.grad` the view of `flat` at its place in the parameter order?  Checked for all of them on every
    call (a frozen parameter, a hook that cloned one gradient in the middle of the list must not go unnoticed -- ADVICE r03):
    address + contiguity, ~15 us for 35 tensors, eager steps only -- a replayed step never comes here)."""
    off, base = 0, flat.data_ptr()
    for p in params:
        g = p.grad
        if g is None or g.data_ptr() != base + 4 * off or not g.is_contiguous():
            return False
        off += p.numel()
    return off == flat.numel()


def allreduce_flat(flat: torch.Tensor) -> None:
    """Average one flat buffer across ranks in place (RCCL: ncclAvg; gloo: sum, then scale)."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world_size())


def allreduce_gradients(model: torch.nn.Module, ordered_params=None) -> None:
    """Average gradients across ranks with ONE collective.  Fast path: the `.grad`s are views of the flat buffer
    the HIP backward produced (model.flat_grad()) -> all-reduce it in place, nothing is copied.  Generic path
    (any nn.Module, used by the gloo CPU tests): flatten -> all-reduce -> scatter back."""
    if not active():
        return
    flat = model.flat_grad() if hasattr(model, "flat_grad") else None
    params = ordered_params if ordered_params is not None else (
        model._ordered_params() if hasattr(model, "_ordered_params") else list(model.parameters()))
    in_place = flat is not None and _grads_are_views_of(flat, params)
    if not in_place:
        grads = [p.grad for p in params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_flat(flat)
    if not in_place:
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


DP_MODES = ("graph", "split", "eager")


def dp_mode_from_env(default: str = "graph") -> str:
    """PFN_DP_MODE = graph | split | eager: the launch form of a data-parallel step (see GraphedStep)."""
    mode = os.environ.get("PFN_DP_MODE", default)
    if mode not in DP_MODES:
        raise ValueError(f"PFN_DP_MODE={mode!r}: expected one of {DP_MODES}")
    return mode


def all_ranks_agree(ok: bool, device=None) -> bool:
    """True iff EVERY rank passes True (one tiny MIN all-reduce, outside any stream capture): the ranks of a data-parallel job
    must never end up in different launch forms -- one would wait in a collective the other never enters."""
    if not active():
        return bool(ok)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class GraphedStep:
    """`fwd_bwd()` -> gradient all-reduce -> `opt_step()`, replayed.  Three launch forms (`mode`, default PFN_DP_MODE or "graph"):

    * "graph": ONE hipGraph.  Without a process group that is all there is.  With RCCL (`backend == "nccl"`) `ncclAllReduce` is
      captured between the backward pass and the optimizer kernel like any other node (RCCL supports stream capture; the process
      group's watchdog thread queries events, hence thread-local capture mode): one graph launch per step, the collective starts
      the moment the last gradient kernel retires.  A backend that cannot be captured (gloo in the one-GPU plumbing tests: its
      all-reduce goes through the host) takes "split".
    * "split": graph(fwd+bwd) -> an eager all-reduce -> graph(optimizer).
    * "eager": no graph at all -- the three calls, every step (the fallback of last resort, and the form to bring up a new
      multi-GPU box with).

    A capture that fails on ANY rank demotes EVERY rank to the next form (the success flag is MIN-all-reduced before a form is
    chosen; `form` says which one was built): ranks never disagree on how a step is launched.  `extra` are device tensors
    SUM-all-reduced next to the gradients (the optimizer's guard scalar: a NaN loss on one rank must skip the update on all).

    `fwd_bwd` must write its loss into tensors it returns (kept as `self.out`); capture happens on the current stream after the
    caller's own warm-up (the process group's communicator must already exist: run at least one eager all-reduce first)."""

    def __init__(self, fwd_bwd, opt_step, model=None, allreduce: Optional[bool] = None, mode: Optional[str] = None, extra=()):
        self.fwd_bwd, self.opt_step, self.model = fwd_bwd, opt_step, model
        self.allreduce = active() if allreduce is None else (allreduce and active())
        self.requested = dp_mode_from_env() if mode is None else mode
        if self.requested not in DP_MODES:
            raise ValueError(f"GraphedStep mode {self.requested!r}: expected one of {DP_MODES}")
        self.extra = list(extra)
        self.graphs, self.mode, self.form, self.out = [], None, None, None
        self._flat, self._grads = None, None       # split form: the gradient tensors the CAPTURED backward writes

    def _reduce_extra(self):
        for t in self.extra:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _reduce(self):
        allreduce_gradients(self.model)
        self._reduce_extra()

    def _remember_captured_grads(self):
        """Split form only: the eager all-reduce between the two graphs must reduce the buffers the captured backward writes and
        the captured optimizer reads -- NOT whatever `model.flat_grad()` / `.grad` point at by then (an eager step in between,
        e.g. the short last batch of an epoch, re-points both at fresh tensors)."""
        m = self.model
        params = m._ordered_params() if hasattr(m, "_ordered_params") else list(m.parameters())
        flat = m.flat_grad() if hasattr(m, "flat_grad") else None
        if flat is not None and _grads_are_views_of(flat, params):
            self._flat = flat
        else:
            self._grads = [p.grad for p in params if p.grad is not None]

    def _reduce_captured(self):
        if self._flat is not None:
            allreduce_flat(self._flat)
        else:
            flat = torch.cat([g.reshape(-1) for g in self._grads])
            allreduce_flat(flat)
            off = 0
            for g in self._grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        self._reduce_extra()

    def _try(self, build) -> bool:
        """Run one capture attempt; True iff it succeeded on EVERY rank."""
        err = None
        try:
            build()
        except Exception as exc:                   # noqa: BLE001
            err = exc
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        if not self.allreduce:
            if err is not None:
                raise err
            return True
        if all_ranks_agree(err is None):
            return True
        self.graphs = []
        self._last_error = err
        return False

    def capture(self):
        kw = {"capture_error_mode": "thread_local"} if self.allreduce else {}
        want = self.requested
        if want == "graph" and self.allreduce and dist.get_backend() != "nccl":
            want = "split"                         # (decided from the backend: the same on every rank)
        if want == "graph":
            def one():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **kw):
                    self.out = self.fwd_bwd()
                    if self.allreduce:
                        self._reduce()
                    self.opt_step()
                self.graphs = [g]
            if self._try(one):
                self.form = "graph"
                self.mode = "one graph incl. RCCL all-reduce" if self.allreduce else "one graph"
                return self
            want = "split"
        if want == "split":
            def two():
                g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb, **kw):
                    self.out = self.fwd_bwd()
                self._remember_captured_grads()
                with torch.cuda.graph(g_opt, **kw):
                    self.opt_step()
                self.graphs = [g_fb, g_opt]
            if self._try(two):
                self.form = "split"
                self.mode = "graph(fwd+bwd) -> eager all-reduce -> graph(optimizer)"
                return self
        self.graphs, self.form, self.mode = [], "eager", "eager launches (no graph)"
        return self

    def replay(self):
        if self.form == "eager":
            self.out = self.fwd_bwd()
            if self.allreduce:
                self._reduce()
            self.opt_step()
        elif len(self.graphs) == 1:
            self.graphs[0].replay()
        else:
            self.graphs[0].replay()
            if self.allreduce:
                self._reduce_captured()
            self.graphs[1].replay()
        return self.out
