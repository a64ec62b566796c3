"""`train_epoch` counterpart of the reference's utils/training.py:30-80: same signature, same per-batch order
(to(device) -> zero_grad -> forward -> loss dispatch by isinstance -> backward -> step -> loss.item()), same return
value (sum(loss * len(data)) / sum(len(data))).  `allreduce` plugs in the one-collective DP step (dp.py)."""
import json
import os
from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import dp
from ..loss import MSELoss
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


def _announce_loss(loss_fn, model, data):
    """Right before `model(data)` in a loop body whose next statements are the loss and its backward (utils/training.py:59-74):
    an MSELoss says so, and a model that can leaves its output rows, the loss and its gradient to the first launch of its
    backward pass (loss.MSELoss.attach -> pfn_mpn_backward_mse)."""
    if isinstance(loss_fn, MSELoss):
        loss_fn.attach(model, data.y)
    elif isinstance(loss_fn, Masked_L2_loss):
        loss_fn.attach(model, data.y, data.pred_mask)


def _dispatch_loss(loss_fn, out, data):
    """The isinstance dispatch of utils/training.py:61-72."""
    if isinstance(loss_fn, Masked_L2_loss):
        return loss_fn(out, data.y, data.pred_mask)
    if isinstance(loss_fn, PowerImbalance):
        masked_out = out * data.pred_mask + data.x * (1 - data.pred_mask)
        return loss_fn(masked_out, data.edge_index, data.edge_attr)
    if isinstance(loss_fn, MixedMSEPoweImbalance):
        return loss_fn(out, data.edge_index, data.edge_attr, data.y)
    return loss_fn(out, data.y)


def _backward(loss_fn, loss):
    if hasattr(loss_fn, "unit_grad"):
        loss.backward(loss_fn.unit_grad(loss))   # same as loss.backward() (utils/training.py:74), two tiny kernels fewer
    else:
        loss.backward()


def _unverified_batch(model) -> bool:
    """Did the model's last forward run on an adjacency whose id-range / segment checks stayed on the device (a NEW edge_index
    tensor per batch, `dynamic_topology`, a build inside a capture)?  Then a bad batch arrives as a NaN loss instead of an exception."""
    g = getattr(getattr(model, "_graphs", None), "_graph", None)
    return bool(g is not None and getattr(g, "unverified", False))


def _guarded_opt_step(optimizer, model, loss, allreduce: bool):
    """`optimizer.step()` of an EAGER loop body.  When the batch could not be validated on the host (`_unverified_batch`) and the
    optimizer can skip on the device (FlatAdamW.guard), the update is decided on THIS step's own loss -- summed across ranks under
    data parallelism, so every replica skips or none does -- and the guard is cleared again: it never outlives the step it was
    set for (a guard left bound to an earlier loss made a later eager step decide on the wrong batch)."""
    if not hasattr(optimizer, "guard") or optimizer.guard is not None or not _unverified_batch(model):
        optimizer.step()
        return
    guard = loss.detach()
    if allreduce and dp.active():
        guard = guard.clone()
        torch.distributed.all_reduce(guard, op=torch.distributed.ReduceOp.SUM)
    optimizer.guard = guard
    try:
        optimizer.step()
    finally:
        optimizer.guard = None


class GraphedTrainStep:
    """The per-batch body of `train_epoch` (zero_grad -> forward -> loss -> backward -> step) captured ONCE into a hipGraph
    and replayed for every following batch of the same shape and topology: the ~34 kernel launches of a step cost one
    graph launch (what bench.py measures).  A batch is copied into the captured input tensors (4 small device copies).
    Under data parallelism the gradient all-reduce is inside the replayed step (dp.GraphedStep).

    Replays only when it is safe, otherwise runs the eager body: the batch must have the captured shapes.  While the loader
    hands in the SAME `edge_index` tensor (the device-resident `PowerFlowData` does: one cached tensor per batch size) the
    adjacency is built once, outside the graph.  The first batch that brings another `edge_index` of the same shape switches
    the step to `dynamic` mode: re-captured with the adjacency build INSIDE the graph (pfn_graph_build from the captured
    edge_index buffer, checks left on the device -- a bad batch gives a NaN loss, see GraphCSR.unverified), so per-batch
    topologies replay too, with no host sync per step (the reference syncs in every forward, networks/MPN.py:498-504)."""

    def __init__(self, model, loss_fn, optimizer, allreduce: Optional[bool] = None, dp_mode: Optional[str] = None):
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.dp_mode = dp_mode     # graph | split | eager (dp.GraphedStep; None: PFN_DP_MODE or "graph")
        # indexed mode (step_indexed): one child step per batch SIZE (the epoch's short last batch gets a graph of its own), each
        # gathering its samples from the device-resident dataset INSIDE its captured graph
        self._children = {}
        self._source = None        # (dataset, captured index buffer) of a child
        self._guard_buf = None     # data parallel + guarded update: the loss, SUM-all-reduced, so that every rank skips together
        # data parallel (dp.py): the gradient all-reduce is part of the replayed step -- ONE hipGraph with the RCCL collective
        # captured between backward and optimizer, or graph / eager all-reduce / graph for a backend that cannot be captured
        self.allreduce = (dp.world_size() > 1) if allreduce is None else bool(allreduce)
        self.graph = self.static = self.loss = None
        self.side = None           # the warm-up's stream; eager fallbacks run on it too (see _eager)
        self.key = None            # optimiser hyper-parameters baked into the captured launches
        self.disabled = False      # a failed capture: stay eager
        # a dataset that hands in a NEW edge_index per batch (topologies that differ per sample -- the reference's `perturbed`
        # sets; or a list-backed loader that re-collates): the step is captured once more WITH the adjacency build inside
        # (model.dynamic_topology: pfn_graph_build + the on-device checks, no host sync), and every batch's edge_index is copied
        # into the captured buffer like x / y / edge_attr -- still one graph launch per batch
        self.dynamic = False
        self._held = []            # GraphCSRs whose workspaces the captured launches read (kept alive as long as the graph is)

    def _topology_owners(self):
        """Everything that keeps an adjacency per `edge_index`: the model and the loss modules that walk the grid themselves
        (PowerImbalance, also inside MixedMSEPoweImbalance)."""
        owners = [self.model] if hasattr(self.model, "dynamic_topology") else []
        if isinstance(self.loss_fn, nn.Module):
            owners += [m for m in self.loss_fn.modules() if hasattr(m, "dynamic_topology")]
        return owners

    def _set_dynamic_topology(self, on: bool):
        for o in self._topology_owners():
            o.dynamic_topology = bool(on)

    def _drop_graph(self):
        self.graph = self.static = None
        self._held = []

    def _hyper_key(self):
        """Every optimiser scalar a captured update would hold BY VALUE (as a kernel argument): lr, betas (OneCycleLR
        cycles beta1 together with lr), eps, weight_decay; any change re-captures.  `FlatAdamW` keeps them on the device
        (`sync_hyper`), so its captured step follows a scheduler without a new capture: constant key."""
        if hasattr(self.opt, "sync_hyper"):
            return ("device-resident",)
        g = self.opt.param_groups[0]
        betas = tuple(float(b) for b in g.get("betas", ()))
        return (float(g["lr"]), betas, float(g.get("eps", 0.0)), float(g.get("weight_decay", 0.0)))

    def _fwd_bwd(self, data):
        if self._source is not None:                               # indexed mode: pull the batch named by the index buffer
            self._source[0].gather_into(data, self._source[1])
        self.opt.zero_grad()
        _announce_loss(self.loss_fn, self.model, data)
        loss = _dispatch_loss(self.loss_fn, self.model(data), data)
        _backward(self.loss_fn, loss)
        return loss.detach()

    def _eager_body(self, data):
        loss = self._fwd_bwd(data)
        if self.allreduce:
            dp.allreduce_gradients(self.model)
        _guarded_opt_step(self.opt, self.model, loss, self.allreduce)
        return loss

    def _replay_eager_form(self):
        """dp.GraphedStep demoted to (or asked for) its "eager" form calls the captured closure on every step: it needs the
        state `_capture` raised around the capture -- per-batch topologies rebuilt on the device instead of the cached,
        host-synchronising build -- and must not leave the optimizer's guard bound to this step's loss afterwards."""
        prev = [(o, o.dynamic_topology) for o in self._topology_owners()]
        if self.dynamic:
            self._set_dynamic_topology(True)
        try:
            return self.graph.replay()
        finally:
            if hasattr(self.opt, "guard"):
                self.opt.guard = None
            for o, was in prev:
                o.dynamic_topology = was

    def _eager(self, data):
        """The eager body -- on the side stream of the capture's warm-up once there is one: autograd binds a parameter's
        gradient accumulator to the stream of its first backward, and a backward on any OTHER stream pays an event record + wait
        per parameter on the host (35 parameters: 1.06 instead of 0.69 ms per step at case118v2 x 128)."""
        if self.side is None:
            return self._eager_body(data)
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            loss = self._eager_body(data)
        cur.wait_stream(self.side)
        return loss

    def _snapshot(self):
        """Everything a training step mutates, so that the warm-up steps torch asks for before a capture leave no trace."""
        opt, snap = self.opt, {}
        if hasattr(opt, "flat_param"):                             # FlatAdamW: four flat device tensors
            snap["flat"] = [t.clone() for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_count)]
        else:
            import copy
            snap["params"] = [p.detach().clone() for p in self.model.parameters()]
            snap["opt"] = copy.deepcopy(opt.state_dict())
        rng = getattr(self.model, "_rng_state", None)
        snap["rng"] = None if rng is None else rng.clone()
        return snap

    def _restore(self, snap):
        opt = self.opt
        with torch.no_grad():
            if "flat" in snap:
                for t, s0 in zip((opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_count), snap["flat"]):
                    t.copy_(s0)
            else:
                for p, s0 in zip(self.model.parameters(), snap["params"]):
                    p.copy_(s0)
                opt.load_state_dict(snap["opt"])
            if snap["rng"] is not None:
                self.model._rng_state.copy_(snap["rng"])

    def _capture(self, data):
        self.static = data.clone()
        if not self.dynamic:
            self.static.edge_index = data.edge_index               # identity matters: the model's adjacency cache keys on it
        # dynamic: the clone IS the captured edge_index buffer.  `dynamic_topology` is raised on the model AND on every loss
        # module that keeps an adjacency of its own (PowerImbalance: with its cache the replayed loss kept walking the
        # capture-time topology) for the warm-up and the capture only: the captured launches rebuild from the
        # buffer; forwards outside the graph (evaluation, the short last batch) keep the validated, cached build.
        snap = self._snapshot()
        if self.side is None:
            self.side = torch.cuda.Stream()
        side = self.side
        prev = [(o, o.dynamic_topology) for o in self._topology_owners()]
        if self.dynamic:
            self._set_dynamic_topology(True)
        guarded = self.dynamic and hasattr(self.opt, "guard")
        # a guarded update under data parallelism must be decided on a value EVERY rank sees: one rank's bad batch reaches the
        # others only as NaN gradients through the all-reduce -- their own losses are finite -- so the losses are summed across
        # ranks (NaN / inf survive a sum) next to the gradients and every rank skips, or none does
        shared_guard = guarded and self.allreduce and dp.active()
        if shared_guard and self._guard_buf is None:
            self._guard_buf = torch.zeros((), dtype=torch.float32, device=data.x.device)
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                          # warm-up off the capture stream (allocator, adjacency cache)
                for _ in range(2):
                    self._eager_body(self.static)
            torch.cuda.current_stream().wait_stream(side)
            self._restore(snap)
            self.opt.zero_grad(set_to_none=True)
            box = {}

            def fwd_bwd():
                box["loss"] = self._fwd_bwd(self.static)
                if guarded:
                    # unverified batches: a bad one (ids out of range, edges across the claimed graph boundaries) reaches the
                    # step as a NaN loss (pfn_graph_poison_if_bad); the captured update then SKIPS instead of turning every
                    # parameter NaN for good (the reference would have raised before the step)
                    if shared_guard:
                        self._guard_buf.copy_(box["loss"])
                        self.opt.guard = self._guard_buf
                    else:
                        self.opt.guard = box["loss"]
                return box["loss"]
            self.graph = dp.GraphedStep(fwd_bwd, self.opt.step, self.model, self.allreduce, mode=self.dp_mode,
                                        extra=[self._guard_buf] if shared_guard else ()).capture()
        finally:
            if guarded:
                self.opt.guard = None
            for o, was in prev:
                o.dynamic_topology = was
        self.loss = self.graph.out
        self.key = self._hyper_key()
        # static mode: the captured launches read the workspaces of the adjacencies built during the warm-up, whose only other
        # owner is a one-entry cache -- a batch with another edge_index (evaluation, a short last batch) would evict and free them
        self._held = [o._graphs._graph for o in self._topology_owners() if getattr(o, "_graphs", None) is not None]
        return self.loss                                           # the capture pass does not execute: caller replays

    def _same_shapes(self, data):
        s = self.static
        if s is None or data.x.shape != s.x.shape or data.edge_attr.shape != s.edge_attr.shape or data.x.device != s.x.device:
            return False
        if data.edge_index.shape != s.edge_index.shape:
            return False
        ps, pd = getattr(s, "ptr", None), getattr(data, "ptr", None)
        return (ps is None) == (pd is None) and (ps is None or ps.shape == pd.shape)

    def _compatible(self, data):
        return self._same_shapes(data) and (self.dynamic or data.edge_index is self.static.edge_index)

    def step_indexed(self, dataset, idx):
        """One training step on the samples `idx` (a device int64 tensor) of a device-resident dataset (`dataset.can_gather()`).
        The captured graph contains the five row gathers that assemble the batch, so a step costs the host ONE 1-KiB device copy
        (the indices) and ONE graph launch -- no collate, no per-field copies; every batch size met gets its own
        captured step (the short last batch of an epoch no longer runs eager).  Returns (loss, len(batch)) like the loop needs."""
        B = int(idx.numel())
        child = self._children.get(B)
        if child is None:
            child = GraphedTrainStep(self.model, self.loss_fn, self.opt, self.allreduce, self.dp_mode)
            child._source = (dataset, idx.clone())
            child._template = dataset.collate_indices(idx.tolist())   # shapes, edge_index / batch / ptr of this batch size
            self._children[B] = child
        child.allreduce = self.allreduce
        child._source[1].copy_(idx)
        return child(child._template), len(child._template)

    def captured(self):
        """The dp.GraphedStep this step (or, in indexed mode, one of its per-size children) replays; None before the first capture."""
        if self.graph is not None:
            return self.graph
        for ch in self._children.values():
            if ch.graph is not None:
                return ch.graph
        return None

    def any_disabled(self) -> bool:
        return self.disabled or any(ch.disabled for ch in self._children.values())

    def _drop_all(self):
        self._drop_graph()
        for ch in self._children.values():
            ch._drop_graph()

    def __call__(self, data):
        if self.disabled or not data.x.is_cuda:
            return self._eager(data)
        if self.graph is not None and self.key != self._hyper_key():
            self._drop_graph()                                     # a scheduler moved lr / betas / ...: capture again
        if self.graph is None:
            snap = self._snapshot()
            try:
                self._capture(data)
            except Exception as exc:                               # noqa: BLE001  (the eager body is always available)
                if self.allreduce and dp.world_size() > 1:
                    # ranks must not disagree on how a step is launched (one would wait in a collective the other never
                    # enters): under data parallelism a failed capture is an error, not a silent change of mode
                    raise
                import warnings
                warnings.warn(f"GraphedTrainStep: hipGraph capture failed ({exc}); running eager launches from here on")
                torch.cuda.synchronize()
                self._restore(snap)
                self._drop_graph()
                self.disabled = True
                return self._eager(data)
        if not self._compatible(data):
            if not self.dynamic and self._same_shapes(data) and hasattr(self.model, "dynamic_topology"):
                # same shapes, another edge_index tensor: this loader changes (or re-collates) the topology per batch.  Capture
                # once more with the adjacency build inside the graph; from here on every batch of this shape replays.
                self.dynamic = True
                self._drop_graph()
                return self(data)
            return self._eager(data)                               # e.g. the short last batch of an epoch
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()                                  # a scheduler moved lr / betas: 20 bytes to the device
        if self._source is None:
            for k in ("x", "y", "pred_mask", "edge_attr") + (("edge_index",) if self.dynamic else ()):
                getattr(self.static, k).copy_(getattr(data, k))
        # (graph forms: the captured loss tensor; the eager form: this step's)
        self.loss = self._replay_eager_form() if self.graph.form == "eager" else self.graph.replay()
        return self.loss


def train_epoch(model: nn.Module, loader, loss_fn: Callable, optimizer, device, progress: bool = False,
                allreduce: Optional[bool] = None, graph: Optional[GraphedTrainStep] = None) -> float:
    """`graph`: a `GraphedTrainStep(model, loss_fn, optimizer)` kept by the caller across epochs -> batches are replayed from
    one hipGraph where that is safe.  Either way the running loss is accumulated on the device and read back ONCE per epoch
    (the reference's per-batch `loss.item()`, :77, is a host sync per step; the returned value is the same sum)."""
    model = model.to(device)
    num_samples = 0
    total = None
    model.train()
    if allreduce is None:
        allreduce = dp.world_size() > 1
    it = loader
    if progress:
        from tqdm import tqdm
        it = tqdm(loader, total=len(loader), desc="Training")
    if graph is not None and graph.allreduce != bool(allreduce):
        graph._drop_all()                       # the replayed step contains (or not) the collective: capture again
        graph.allreduce = bool(allreduce)
    ds = getattr(loader, "dataset", None)
    if (graph is not None and not progress and hasattr(loader, "index_batches") and hasattr(ds, "can_gather") and ds.can_gather()
            and ds.device == torch.device(device)):
        # device-resident dataset: the captured step gathers its own batch -- per batch one index copy and one graph launch
        for idx in loader.index_batches(device):
            loss, n_keys = graph.step_indexed(ds, idx)
            num_samples += n_keys
            term = loss.detach().double() * n_keys
            total = term if total is None else total + term
        it = ()
    for data in it:
        data = data.to(device)
        if graph is not None:
            loss = graph(data)
        else:
            optimizer.zero_grad()
            _announce_loss(loss_fn, model, data)
            loss = _dispatch_loss(loss_fn, model(data), data)
            _backward(loss_fn, loss)
            if allreduce:
                dp.allreduce_gradients(model)
            _guarded_opt_step(optimizer, model, loss, allreduce)
        num_samples += len(data)
        term = loss.detach().double() * len(data)
        total = term if total is None else total + term
    skipped = getattr(optimizer, "skipped_steps", None)
    if skipped is not None and graph is not None and graph.dynamic:
        # the guarded update (FlatAdamW.guard) skips on a non-finite loss without telling the host: say so once per epoch, so a
        # poisoned batch -- or a run that has diverged for good -- does not replay silently (the reference would have raised)
        n_skipped = skipped()
        if n_skipped > getattr(graph, "_skipped_seen", 0):
            import warnings
            warnings.warn(f"train_epoch: {n_skipped - getattr(graph, '_skipped_seen', 0)} optimizer update(s) skipped this epoch "
                          f"(non-finite loss: a batch flagged bad on the device, or a diverged run)")
            graph._skipped_seen = n_skipped
    return float(total.item()) / max(num_samples, 1) if total is not None else 0.0
