"""Host-side mirror of the reference's `networks/MPN.py` hot-path classes on the MI355X HIP kernels.

Same constructor signatures, attribute names, `state_dict` keys and `forward` semantics as

  * `EdgeAggregation`   -- networks/MPN.py:6-56
  * `TAGConv`           -- torch_geometric.nn.TAGConv as used at networks/MPN.py:477-484,:545
  * `MaskEmbdMultiMPN`  -- networks/MPN.py:456-559

so `train.py` / `utils/training.py` / `utils/evaluation.py`-shaped callers are drop-in.  All arithmetic runs
in libpfn_hip.so through the C ABI of include/pfn_hip.h; torch supplies device memory, streams and the
autograd graph edges only.  There is no CPU implementation here: CPU tensors raise RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib as L


def _padded(f: int) -> int:
    return (f + 3) // 4 * 4


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


# ================================================================================================ graph
class GraphCSR:
    """Device adjacency built by `pfn_graph_build` from a PyG-style `edge_index` (2, E) int64.

    mode -1: the reference's first-edge `is_directed` heuristic decides on device whether the reversed
    copies are appended (networks/MPN.py:498-523); 0: use the list as given; 1: always undirect.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, mode: int = -1, validate: bool = True,
                 seg_hint: int = 0, async_checks: bool = False):
        lib = L.load()
        L.require_device(edge_index, what="edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError(f"edge_index must be int64 of shape (2, E), got {edge_index.dtype} {tuple(edge_index.shape)}")
        edge_index = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
        self.num_nodes, self.e_stored, self.mode = int(num_nodes), int(edge_index.shape[1]), int(mode)
        self.device = edge_index.device
        nbytes = lib.pfn_graph_workspace_bytes(self.num_nodes, self.e_stored)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._keepalive = edge_index
        with torch.cuda.device(self.device):
            L.check(lib.pfn_graph_build(edge_index.data_ptr(), self.e_stored, self.num_nodes, self.mode,
                                        self.ws.data_ptr(), nbytes, L.stream_ptr()), "pfn_graph_build")
        self._keepalive = None
        self.seg_nodes = 0       # > 0: the batch is a union of index-contiguous graphs of this many nodes (checked on device)
        # `unverified`: the id-range and segment checks ran on the device but were NOT read back (no host sync: a topology that
        # changes per batch, or a build inside a stream capture).  The segment hint is then taken on trust and the model ends its
        # forward with pfn_graph_poison_if_bad: a violated check turns the output into NaN instead of going unnoticed.
        self.unverified = bool(async_checks or _capturing())
        if self.unverified:
            if seg_hint > 0 and self.num_nodes > 0 and self.num_nodes % seg_hint == 0:
                with torch.cuda.device(self.device):
                    L.check(lib.pfn_graph_segments_async(self.ws.data_ptr(), self.num_nodes, self.e_stored, int(seg_hint),
                                                         L.stream_ptr()), "pfn_graph_segments_async")
                self.seg_nodes = int(seg_hint)
        elif validate:
            self.info()          # raises on out-of-range ids (one sync per NEW topology only)
            if seg_hint > 0 and self.num_nodes % seg_hint == 0:
                ok = C.c_int32(0)
                with torch.cuda.device(self.device):
                    L.check(lib.pfn_graph_segments(self.ws.data_ptr(), self.num_nodes, self.e_stored, int(seg_hint),
                                                   C.byref(ok), L.stream_ptr()), "pfn_graph_segments")
                self.seg_nodes = int(seg_hint) if ok.value else 0

    def info(self):
        """(directed, effective_edge_count); synchronises.  Raises RuntimeError on an out-of-range node id."""
        lib = L.load()
        d, e = C.c_int32(0), C.c_int64(0)
        with torch.cuda.device(self.device):
            L.check(lib.pfn_graph_info(self.ws.data_ptr(), self.num_nodes, self.e_stored, C.byref(d), C.byref(e),
                                       L.stream_ptr()), "pfn_graph_info")
        return bool(d.value), int(e.value)

    def export_edges(self) -> torch.Tensor:
        """The effective edge list (2, E_eff) int64, in edge-id order (originals first, reverses second)."""
        lib = L.load()
        _, e_eff = self.info()
        out = torch.empty(2, max(2 * self.e_stored, 1), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(lib.pfn_graph_export_edges(self.ws.data_ptr(), self.num_nodes, self.e_stored, out.data_ptr(),
                                               L.stream_ptr()), "pfn_graph_export_edges")
        return out[:, :e_eff].clone()


class _GraphCache:
    """Reuses the adjacency while the caller hands in the very same, unmodified edge_index tensor (identity + torch's in-place
    version counter): legitimate because every sample of a reference case shares one topology, and safe because a new or mutated
    tensor misses.  A NEW tensor of the cached shape (what a PyG-style loader hands out per batch, train.py:90-92) is NOT compared
    with the cached list -- the verdict of a device-side compare has to be read back, a host sync per batch that doubled the step
    (1.14 vs 0.58 ms at case118v2 x 128, round 5) -- but rebuilt on the device with the checks left there (`GraphCSR.unverified`:
    no host sync, hipGraph-capturable; a node id out of range or an edge between two graphs of the batch turns the output into
    NaN, pfn_graph_poison_if_bad, instead of raising).  Only the FIRST build of a shape is validated with a read-back (and raises).
    The last VALIDATED build is kept next to the current one: a caller that comes back to that very tensor (evaluation between
    training batches, two loaders taking turns) gets the validated adjacency -- and with it the loss tail -- back."""

    def __init__(self):
        self._ref, self._key, self._graph = None, None, None
        self._validated = None        # (weakref of its edge_index, key, graph) of the last build whose checks were read back
        self.device_rebuilds = 0      # new tensors of the cached shape that took the sync-free path

    def get(self, edge_index: torch.Tensor, num_nodes: int, mode: int, seg_hint: int = 0, rebuild: bool = False) -> GraphCSR:
        """`rebuild`: the caller's topology changes per batch -- build anew from `edge_index` every time, checks left on the
        device (no host sync, hipGraph-capturable: a captured step then re-derives the adjacency from whatever the captured
        edge_index buffer holds at replay time)."""
        key = (edge_index._version, edge_index.data_ptr(), tuple(edge_index.shape), num_nodes, mode, seg_hint)
        if not rebuild and self._ref is not None and self._ref() is edge_index and self._key == key:
            return self._graph
        if not rebuild and self._validated is not None and self._validated[0]() is edge_index and self._validated[1] == key:
            self._ref, self._key, self._graph = self._validated
            return self._graph
        same_shape = self._key is not None and key[2:] == self._key[2:] and self._graph is not None and edge_index.device == self._graph.device
        if same_shape and not rebuild:
            self.device_rebuilds += 1
        g = GraphCSR(edge_index, num_nodes, mode, seg_hint=seg_hint, async_checks=rebuild or same_shape)
        self._ref, self._key, self._graph = weakref.ref(edge_index), key, g
        if not g.unverified:
            self._validated = (self._ref, key, g)
        return g


def _pad_rows(x: torch.Tensor, f: int) -> torch.Tensor:
    """(N, f) -> (N, roundup(f, 4)) with zero pad columns (no copy when f % 4 == 0)."""
    ld = _padded(f)
    if ld == f:
        return x
    out = torch.empty(x.shape[0], ld, dtype=torch.float32, device=x.device)
    L.check(L.load().pfn_pad_rows(x.data_ptr(), f, out.data_ptr(), ld, x.shape[0], f, L.stream_ptr()), "pfn_pad_rows")
    return out


def _unpad_rows(x: torch.Tensor, f: int) -> torch.Tensor:
    return x if x.shape[1] == f else x[:, :f].contiguous()


# ================================================================================= EdgeAggregation layer
class _EdgeAggrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph: GraphCSR, dims, x, edge_attr, w1, b1, w2, b2):
        lib = L.load()
        fi, fe, h, fo = dims
        n = x.shape[0]
        xp = _pad_rows(x, fi)
        out = torch.empty(n, _padded(fo), dtype=torch.float32, device=x.device)
        nbytes = lib.pfn_edge_aggr_workspace_bytes(n, graph.e_stored, fi, fe, h, fo)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(lib.pfn_edge_aggr_forward(graph.ws.data_ptr(), n, graph.e_stored, fi, fe, h, fo, xp.data_ptr(), _padded(fi),
                                          edge_attr.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                          out.data_ptr(), _padded(fo), ws.data_ptr(), nbytes, L.stream_ptr()),
                "pfn_edge_aggr_forward")
        ctx.graph, ctx.dims, ctx.ws = graph, dims, ws
        ctx.save_for_backward(xp, edge_attr, w1, b1, w2, b2)
        return _unpad_rows(out, fo)

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        fi, fe, h, fo = ctx.dims
        xp, edge_attr, w1, b1, w2, b2 = ctx.saved_tensors
        graph, n = ctx.graph, xp.shape[0]
        gp = _pad_rows(L.f32c(gout, "grad_out"), fo)
        gx = torch.empty_like(xp)
        gea = torch.empty_like(edge_attr) if ctx.needs_input_grad[3] else None
        gw1, gb1, gw2, gb2 = (torch.empty_like(t) for t in (w1, b1, w2, b2))
        L.check(lib.pfn_edge_aggr_backward(graph.ws.data_ptr(), n, graph.e_stored, fi, fe, h, fo, xp.data_ptr(), _padded(fi),
                                           edge_attr.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                           gp.data_ptr(), _padded(fo), gx.data_ptr(), _padded(fi), L.ptr(gea),
                                           gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(),
                                           ctx.ws.data_ptr(), ctx.ws.numel(), L.stream_ptr()), "pfn_edge_aggr_backward")
        return None, None, _unpad_rows(gx, fi), gea, gw1, gb1, gw2, gb2


class EdgeAggregation(nn.Module):
    """networks/MPN.py:6-56.  out[i] = sum_{e: dst(e)=i} MLP(cat[x_i, x_src(e), edge_attr_e]), aggr='add';
    the degree `norm` the reference computes at :43-47 never reaches `message` and is not reproduced."""

    def __init__(self, nfeature_dim, efeature_dim, hidden_dim, output_dim):
        super().__init__()
        self.nfeature_dim = nfeature_dim
        self.efeature_dim = efeature_dim
        self.output_dim = output_dim
        self.edge_aggr = nn.Sequential(
            nn.Linear(nfeature_dim * 2 + efeature_dim, hidden_dim),
            nn.ReLU(),
            nn.Linear(hidden_dim, output_dim),
        )
        self._graphs = _GraphCache()

    @property
    def hidden_dim(self):
        return self.edge_aggr[0].out_features

    def forward(self, x, edge_index, edge_attr):
        L.require_device(x, edge_index, edge_attr, *self.parameters(), what="EdgeAggregation input")
        with torch.cuda.device(x.device):
            graph = self._graphs.get(edge_index, x.shape[0], 0)     # the layer takes the list as given
            return self.on_graph(graph, x, edge_attr)

    def on_graph(self, graph: GraphCSR, x, edge_attr):
        """The layer over an adjacency that is already built (model compositions build ONE per batch, with the model's
        undirect rule; `edge_attr` stays the stored list -- the kernels read reversed copies through eid mod E)."""
        L.require_device(x, edge_attr, *self.parameters(), what="EdgeAggregation input")
        x, edge_attr = L.f32c(x, "x"), L.f32c(edge_attr, "edge_attr")
        if x.dim() != 2 or x.shape[1] != self.nfeature_dim:
            raise RuntimeError(f"x must be (N, {self.nfeature_dim}), got {tuple(x.shape)}")
        if edge_attr.shape != (graph.e_stored, self.efeature_dim):
            raise RuntimeError(f"edge_attr must be ({graph.e_stored}, {self.efeature_dim}), got {tuple(edge_attr.shape)}")
        with torch.cuda.device(x.device):
            l1, l2 = self.edge_aggr[0], self.edge_aggr[2]
            dims = (self.nfeature_dim, self.efeature_dim, l1.out_features, self.output_dim)
            return _EdgeAggrFn.apply(graph, dims, x, edge_attr, l1.weight, l1.bias, l2.weight, l2.bias)


# ========================================================================================== TAGConv layer
class _TagConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph: GraphCSR, dims, x, bias, *weights):
        lib = L.load()
        cin, cout, K = dims
        n = x.shape[0]
        xp = _pad_rows(x, cin)
        out = torch.empty(n, _padded(cout), dtype=torch.float32, device=x.device)
        nbytes = lib.pfn_tag_conv_workspace_bytes(n, graph.e_stored, cin, cout, K)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(lib.pfn_tag_conv_forward(graph.ws.data_ptr(), n, graph.e_stored, cin, cout, K, xp.data_ptr(), _padded(cin),
                                         L.ptr_table(weights), L.ptr(bias), out.data_ptr(), _padded(cout), ws.data_ptr(),
                                         nbytes, graph.seg_nodes, L.stream_ptr()), "pfn_tag_conv_forward")
        ctx.graph, ctx.dims, ctx.ws, ctx.has_bias = graph, dims, ws, bias is not None
        ctx.save_for_backward(xp, *weights)
        return _unpad_rows(out, cout)

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        cin, cout, K = ctx.dims
        xp, *weights = ctx.saved_tensors
        graph, n = ctx.graph, xp.shape[0]
        gp = _pad_rows(L.f32c(gout, "grad_out"), cout)
        gx = torch.empty_like(xp)
        gws = [torch.empty_like(w) for w in weights]
        gb = torch.empty(cout, dtype=torch.float32, device=xp.device) if ctx.has_bias else None
        L.check(lib.pfn_tag_conv_backward(graph.ws.data_ptr(), n, graph.e_stored, cin, cout, K, xp.data_ptr(), _padded(cin),
                                          L.ptr_table(weights), gp.data_ptr(), _padded(cout), gx.data_ptr(), _padded(cin),
                                          L.ptr_table(gws), L.ptr(gb), ctx.ws.data_ptr(), ctx.ws.numel(), graph.seg_nodes,
                                          L.stream_ptr()), "pfn_tag_conv_backward")
        return (None, None, _unpad_rows(gx, cin), gb, *gws)


class TAGConv(nn.Module):
    """PyG `TAGConv(in_channels, out_channels, K)` with its defaults (normalize=True, bias=True, no self loops):
    out = sum_k (A_hat^k x) W_k^T + b.  state_dict keys `lins.{k}.weight`, `bias` (zero-initialised)."""

    def __init__(self, in_channels, out_channels, K=3, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self._graphs = _GraphCache()

    def forward(self, x, edge_index):
        L.require_device(x, edge_index, *self.parameters(), what="TAGConv input")
        with torch.cuda.device(x.device):
            graph = self._graphs.get(edge_index, x.shape[0], 0)
            return self.on_graph(graph, x)

    def on_graph(self, graph: GraphCSR, x):
        """The layer over an adjacency that is already built (see EdgeAggregation.on_graph)."""
        L.require_device(x, *self.parameters(), what="TAGConv input")
        x = L.f32c(x, "x")
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"x must be (N, {self.in_channels}), got {tuple(x.shape)}")
        with torch.cuda.device(x.device):
            return _TagConvFn.apply(graph, (self.in_channels, self.out_channels, self.K), x, self.bias,
                                    *[l.weight for l in self.lins])


def _hip_linear(graph: GraphCSR, x, lin: nn.Linear):
    """`nn.Linear` on the HIP GEMM: a TAGConv with K = 0 IS x W^T + b (no hop runs; the adjacency is only a handle)."""
    return _TagConvFn.apply(graph, (lin.in_features, lin.out_features, 0), L.f32c(x, "x"), lin.bias, lin.weight)


# ============================================================================================ whole model
class _MpnFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = pfn_mpn_forward, backward = pfn_mpn_backward writing
    every parameter gradient into ONE flat buffer (the data-parallel all-reduce unit)."""

    @staticmethod
    def forward(ctx, model, graph, x, pred_mask, edge_attr, *params):
        lib = L.load()
        cfg = model._config()
        # (autograd records this call only when grad mode was on at `apply`: inside forward it is always off, and
        #  ctx.needs_input_grad ignores torch.no_grad())
        cfg.need_backward = 1 if (model._grad_mode_at_apply and any(ctx.needs_input_grad)) else 0
        n = x.shape[0]
        fo = model.output_dim
        out = torch.empty(n, _padded(fo), dtype=torch.float32, device=x.device)
        nbytes = lib.pfn_mpn_workspace_bytes(C.byref(cfg), n, graph.e_stored)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        mask_dtype = 0 if pred_mask.dtype == torch.int64 else 1
        # MSELoss announced for this very forward (loss.MSELoss.attach, the training loop's promise): the output rows, the loss and
        # its gradient are left to the backward pass's first launch (pfn_mpn_backward_mse).  Not with checks left on the device
        # (the poison below must reach the loss through `out`).
        attach, model._mse_attach = getattr(model, "_mse_attach", None), None
        tail = None
        if (attach is not None and cfg.need_backward and fo == 4 and not graph.unverified and graph.seg_nodes > 0
                and not ctx.needs_input_grad[4] and torch.is_tensor(attach[0]) and attach[0].is_cuda
                and attach[0].device == x.device and attach[0].dtype == torch.float32 and tuple(attach[0].shape) == (n, 4)
                and attach[0].is_contiguous()
                and (attach[2] is None or attach[3] is model._mask_seen)      # Masked_L2_loss: its mask IS the model's pred_mask
                and lib.pfn_mpn_mse_tail_ok(C.byref(cfg), n, graph.e_stored, graph.seg_nodes) == 1):
            from ..loss import MseTail
            tail = MseTail(attach[0], torch.empty((), dtype=torch.float32, device=x.device),
                           torch.empty(n, 4, dtype=torch.float32, device=x.device), attach[1].on(x.device), attach[2], attach[3])
        L.check(lib.pfn_mpn_forward(C.byref(cfg), graph.ws.data_ptr(), n, graph.e_stored, L.ptr_table(params), x.data_ptr(),
                                    pred_mask.data_ptr(), mask_dtype, edge_attr.data_ptr(), None if tail is not None else out.data_ptr(),
                                    ws.data_ptr(), nbytes, L.ptr(model._rng_state_on(x.device)), graph.seg_nodes, L.stream_ptr()),
                "pfn_mpn_forward")
        # (an alias, not `out` itself: the returned tensor gets this node as grad_fn -- keeping IT here would be a reference cycle
        #  that holds the whole workspace until the garbage collector runs)
        ctx.tail, ctx.out = tail, (out.detach() if tail is not None else None)
        model._mse_tail = tail        # picked up by MaskEmbdMultiMPN.forward, which hangs it on the tensor it returns
        if graph.unverified:     # checks not read back (changing topology / capture): a bad batch must not pass silently
            L.check(lib.pfn_graph_poison_if_bad(graph.ws.data_ptr(), n, graph.e_stored, out.data_ptr(), out.numel(),
                                                L.stream_ptr()), "pfn_graph_poison_if_bad")
        ctx.model, ctx.graph, ctx.cfg, ctx.ws, ctx.mask_dtype = model, graph, cfg, ws, mask_dtype
        ctx.save_for_backward(x, pred_mask, edge_attr, *params)
        # (verification aid, `export_gates`: what the last recorded forward left behind -- weak, nothing is kept alive)
        model._last_forward = (weakref.ref(ws), weakref.ref(graph), cfg, weakref.ref(edge_attr)) if cfg.need_backward else None
        return _unpad_rows(out, fo)

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        x, pred_mask, edge_attr, *params = ctx.saved_tensors
        model, graph, n = ctx.model, ctx.graph, x.shape[0]
        gp = _pad_rows(L.f32c(gout, "grad_out"), model.output_dim)
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=x.device)
        # (one split call + a view per 2-D weight: slicing the flat buffer tensor by tensor was ~100 us of host time per step)
        grads = [g if p.dim() == 1 else g.view(p.shape) for g, p in zip(flat.split_with_sizes(sizes), params)]
        gx = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        gea = torch.empty_like(edge_attr) if ctx.needs_input_grad[4] else None
        tail = ctx.tail
        if tail is not None:         # the attached MSELoss: out, loss and grad_out are formed by the first backward launch
            if gout.data_ptr() != tail.grad_out.data_ptr():
                raise RuntimeError("MSELoss.attach(): the model output was used by something else than the attached loss "
                                   "(its rows are only written by this backward pass)")
            if tail.masked is not None:
                L.check(lib.pfn_mpn_backward_masked_l2(C.byref(ctx.cfg), graph.ws.data_ptr(), n, graph.e_stored, L.ptr_table(params),
                                                       L.ptr_table(grads), x.data_ptr(), edge_attr.data_ptr(), tail.target.data_ptr(),
                                                       int(tail.masked[0]), float(tail.masked[1]), ctx.out.data_ptr(),
                                                       tail.loss.data_ptr(), tail.grad_out.data_ptr(), L.ptr(gx), ctx.ws.data_ptr(),
                                                       ctx.ws.numel(), tail.ws.data_ptr(), tail.ws.numel() * 4, graph.seg_nodes,
                                                       L.stream_ptr()), "pfn_mpn_backward_masked_l2")
            else:
                L.check(lib.pfn_mpn_backward_mse(C.byref(ctx.cfg), graph.ws.data_ptr(), n, graph.e_stored, L.ptr_table(params),
                                                 L.ptr_table(grads), x.data_ptr(), edge_attr.data_ptr(), tail.target.data_ptr(),
                                                 ctx.out.data_ptr(), tail.loss.data_ptr(), tail.grad_out.data_ptr(), L.ptr(gx),
                                                 ctx.ws.data_ptr(), ctx.ws.numel(), tail.ws.data_ptr(), tail.ws.numel() * 4,
                                                 graph.seg_nodes, L.stream_ptr()), "pfn_mpn_backward_mse")
            model._last_flat_grad = flat
            return (None, None, gx, None, gea, *grads)
        L.check(lib.pfn_mpn_backward(C.byref(ctx.cfg), graph.ws.data_ptr(), n, graph.e_stored, L.ptr_table(params),
                                     L.ptr_table(grads), x.data_ptr(), pred_mask.data_ptr(), ctx.mask_dtype,
                                     edge_attr.data_ptr(), gp.data_ptr(), L.ptr(gx), L.ptr(gea), ctx.ws.data_ptr(),
                                     ctx.ws.numel(), graph.seg_nodes, L.stream_ptr()), "pfn_mpn_backward")
        model._last_flat_grad = flat
        return (None, None, gx, None, gea, *grads)


class _UndirectHelpers:
    """`is_directed` / `undirect_graph` as every model class of the reference exposes them (networks/MPN.py:172-193,
    :245-266, ..., :498-523), evaluated by the device kernel the forward pass uses."""

    def is_directed(self, edge_index):
        """First-edge heuristic (networks/MPN.py:498-504): with (u0, v0) the first stored edge, 'directed' iff no stored edge
        (v0 -> u0) exists."""
        if edge_index.shape[1] == 0:
            return False
        L.require_device(edge_index, what="edge_index")
        with torch.cuda.device(edge_index.device):
            n = int(edge_index.max().item()) + 1
            return GraphCSR(edge_index, n, mode=-1).info()[0]

    def undirect_graph(self, edge_index, edge_attr):
        """networks/MPN.py:506-523 (originals first, reversed copies second, attributes duplicated)."""
        if edge_index.shape[1] == 0:
            return edge_index, edge_attr
        L.require_device(edge_index, edge_attr, what="edge_index/edge_attr")
        with torch.cuda.device(edge_index.device):
            n = int(edge_index.max().item()) + 1
            g = GraphCSR(edge_index, n, mode=-1)
            if not g.info()[0]:
                return edge_index, edge_attr
            return g.export_edges(), torch.cat([edge_attr, edge_attr], dim=0)


class MaskEmbdMultiMPN(_UndirectHelpers, nn.Module):
    """networks/MPN.py:456-559: mask embedding + (EdgeAggregation, TAGConv) x (L-1) + EdgeAggregation."""

    _mse_attach = None   # (target, workspace) announced by loss.MSELoss.attach for the NEXT forward, consumed by it (one-shot)
    _mse_tail = None
    _mask_seen = None

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self.nfeature_dim = nfeature_dim
        self.efeature_dim = efeature_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.n_gnn_layers = n_gnn_layers
        self.K = K
        self.dropout_rate = dropout_rate
        if n_gnn_layers < 2:
            # the reference's n_gnn_layers == 1 branch (networks/MPN.py:475-477) builds TAGConv(H, output_dim)
            # followed by EdgeAggregation(H, ...): shape-broken unless H == output_dim.
            raise ValueError("MaskEmbdMultiMPN needs n_gnn_layers >= 2 (the reference's L == 1 model cannot run)")
        self.layers = nn.ModuleList()
        self.layers.append(EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, hidden_dim))
        self.layers.append(TAGConv(hidden_dim, hidden_dim, K=K))
        for _ in range(n_gnn_layers - 2):
            self.layers.append(EdgeAggregation(hidden_dim, efeature_dim, hidden_dim, hidden_dim))
            self.layers.append(TAGConv(hidden_dim, hidden_dim, K=K))
        self.layers.append(EdgeAggregation(hidden_dim, efeature_dim, hidden_dim, output_dim))
        self.mask_embd = nn.Sequential(
            nn.Linear(nfeature_dim, hidden_dim),
            nn.ReLU(),
            nn.Linear(hidden_dim, nfeature_dim),
        )
        self.dropout = nn.Dropout(self.dropout_rate, inplace=False)
        self._graphs = _GraphCache()
        self._rng_state: Optional[torch.Tensor] = None
        self._last_flat_grad: Optional[torch.Tensor] = None
        # True: every forward rebuilds the adjacency from the batch's edge_index on the device without a host sync (datasets
        # whose topology differs per sample: the reference's `perturbed` sets, dataset_generator.py:250-253) -- hipGraph-
        # capturable; the id-range / segment checks then surface as a NaN output (GraphCSR.unverified)
        self.dynamic_topology = False

    # ------------------------------------------------------------------------------------- plumbing
    def _config(self) -> L.MpnConfig:
        return L.MpnConfig(self.nfeature_dim, self.efeature_dim, self.output_dim, self.hidden_dim, self.n_gnn_layers,
                           self.K, float(self.dropout_rate), 1 if self.training else 0, 0)

    def _ordered_params(self):
        """The C ABI's parameter table order (include/pfn_hip.h).  The list is built once (walking the module tree costs ~60 us,
        several times per eager step) and rebuilt when a parameter OBJECT of the model was replaced; in-place updates,
        load_state_dict, .to() and FlatAdamW's re-pointing of `.data` keep the objects."""
        cached, slots = self.__dict__.get("_param_list"), self.__dict__.get("_param_slots")
        if cached is not None and self.__dict__.get("_param_list_n") == len(self.layers):
            # EVERY parameter object is confirmed in its owner's `_parameters` dict (35 lookups, ~3 us): replacing one in the
            # middle of the list -- a re-assigned weight, a parametrization -- must not leave a stale tensor in the table
            for (d, name), p in zip(slots, cached):
                if d.get(name) is not p:
                    break
            else:
                return cached
        out, slots = self._walk_params()
        self.__dict__["_param_list"], self.__dict__["_param_slots"], self.__dict__["_param_list_n"] = out, slots, len(self.layers)
        return out

    def _walk_params(self):
        """(parameters in the C ABI's order, their (owner._parameters, name) slots)"""
        mods = []
        for layer in self.layers:
            if isinstance(layer, EdgeAggregation):
                l1, l2 = layer.edge_aggr[0], layer.edge_aggr[2]
                mods += [(l1, "weight"), (l1, "bias"), (l2, "weight"), (l2, "bias")]
            else:
                mods += [(lin, "weight") for lin in layer.lins] + [(layer, "bias")]
        a, b = self.mask_embd[0], self.mask_embd[2]
        mods += [(a, "weight"), (a, "bias"), (b, "weight"), (b, "bias")]
        return [getattr(m, n) for m, n in mods], [(m._parameters, n) for m, n in mods]

    def _rng_state_on(self, device):
        if not (self.training and self.dropout_rate > 0):
            return None
        if self._rng_state is None or self._rng_state.device != device:
            seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
            self._rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        return self._rng_state

    def seed_dropout(self, seed: int):
        """Re-seed the counter-based dropout stream (device state {seed, offset}; offset advances per forward)."""
        dev = next(self.parameters()).device
        self._rng_state = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)

    def flat_grad(self) -> Optional[torch.Tensor]:
        """The flat fp32 buffer the last backward wrote all parameter gradients into (views of it are the
        `.grad`s autograd handed out) -- the unit poweflownet_amd.dp all-reduces."""
        return self._last_flat_grad

    def export_gates(self):
        """Verification aid (`pfn_mpn_export_gates`): the ReLU decisions of the last forward pass that autograd recorded, while
        its workspace is still alive (i.e. before the output / loss tensor is dropped): {"edge": {layer index: bool (E_eff, H)},
        "out": {layer index: bool (N, H)}, "mask_embd": bool (N, H)}, edges in the order `undirect_graph` produces.  Tests feed
        them to a float64 run of the CPU oracle so that gradients are compared on the SAME piecewise-linear branch."""
        lf = getattr(self, "_last_forward", None)
        ws, graph, edge_attr = (lf[0](), lf[1](), lf[3]()) if lf is not None else (None, None, None)
        if ws is None or graph is None or edge_attr is None:
            raise RuntimeError("export_gates: no recorded forward pass is alive (call it before dropping the output / loss)")
        lib, cfg = L.load(), lf[2]
        n, h, dev = graph.num_nodes, self.hidden_dim, ws.device
        params = self._ordered_params()
        _, e_eff = graph.info()
        nlayers = 2 * self.n_gnn_layers - 1

        def run(kind, layer, rows):
            out = torch.empty(max(rows, 1), h, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                L.check(lib.pfn_mpn_export_gates(C.byref(cfg), graph.ws.data_ptr(), n, graph.e_stored, L.ptr_table(params),
                                                 edge_attr.data_ptr(), ws.data_ptr(), ws.numel(), graph.seg_nodes, kind, layer,
                                                 out.data_ptr(), L.stream_ptr()), "pfn_mpn_export_gates")
            return out
        gates = {"edge": {}, "out": {}}
        for i in range(nlayers):
            if i % 2 == 0:
                gates["edge"][i] = run(0, i, 2 * graph.e_stored)[:e_eff].bool()
            if i + 1 < nlayers:
                gates["out"][i] = run(1, i, n).bool()
        gates["mask_embd"] = run(2, 0, n).bool()
        return gates

    # -------------------------------------------------------------------------------------- forward
    def forward(self, data):
        assert data.x.shape[-1] == 4                       # networks/MPN.py:528
        x = data.x
        bus_type = data.bus_type                           # read like the reference (:531-532), unused
        batch = data.batch
        mask = data.pred_mask
        edge_index = data.edge_index
        edge_features = data.edge_attr
        del bus_type, batch
        if self.nfeature_dim != 4:
            raise RuntimeError("MaskEmbdMultiMPN.forward asserts 4 node features (networks/MPN.py:528); "
                               f"this model was built with nfeature_dim={self.nfeature_dim}")
        self._mask_seen = mask                             # (identity of the tensor as the caller holds it: Masked_L2_loss.attach)
        try:
            return self._forward(data, x, mask, edge_index, edge_features)
        finally:
            # the loss announced for THIS forward (loss.MSELoss.attach) is consumed by it or dropped with it: a forward that raises
            # (a shape check, a bad first topology) or never reaches the autograd node must not leave it to an unrelated later one
            self._mse_attach = None
            self._mask_seen = None

    def _forward(self, data, x, mask, edge_index, edge_features):
        params = self._ordered_params()
        L.require_device(x, mask, edge_index, edge_features, params[0], params[-1], what="MaskEmbdMultiMPN input")
        x, edge_features = L.f32c(x, "data.x"), L.f32c(edge_features, "data.edge_attr")
        if mask.dtype != torch.int64:
            mask = mask.float()                            # `.float()` of the reference (:533)
        mask = mask if mask.is_contiguous() else mask.contiguous()
        if mask.shape != x.shape:
            raise RuntimeError(f"pred_mask shape {tuple(mask.shape)} != x shape {tuple(x.shape)}")
        if edge_features.shape != (edge_index.shape[1], self.efeature_dim):
            raise RuntimeError(f"edge_attr must be ({edge_index.shape[1]}, {self.efeature_dim}), got {tuple(edge_features.shape)}")
        with torch.cuda.device(x.device):
            # PyG-style batches carry `ptr` (B+1,): equal-sized graphs let the TAGConv hops stay in LDS per graph; the
            # hint is verified on device against the actual edge list (pfn_graph_segments) before it is trusted
            ptr = getattr(data, "ptr", None)
            nseg = int(ptr.numel()) - 1 if torch.is_tensor(ptr) else 0
            seg_hint = x.shape[0] // nseg if nseg > 0 and x.shape[0] % nseg == 0 else 0
            graph = self._graphs.get(edge_index, x.shape[0], -1, seg_hint, rebuild=self.dynamic_topology)   # is_directed + undirect_graph (:539)
            self._grad_mode_at_apply = torch.is_grad_enabled()
            self._mse_tail = None
            out = _MpnFn.apply(self, graph, x, mask, edge_features, *params)
            if self._mse_tail is not None:     # loss.MSELoss.forward finds the arrangement on the tensor it is handed
                out._pfn_mse_tail, self._mse_tail = self._mse_tail, None
            return out


# ============================================================================================ MPN_simplenet
class MPN_simplenet(nn.Module):
    """networks/MPN.py:753-792 -- the one sibling of `MaskEmbdMultiMPN` in train.py's `models` table (:30-38) that accepts
    the 4-wide node features the dataset produces (the others assert a stale 12-wide layout, :194,:267,:349,:430,:625,
    :728): ONE EdgeAggregation on the edge list AS GIVEN (no undirecting), then `n_gnn_layers` TAGConvs with
    dropout -> ReLU between them.  A composition of this package's two HIP layers under torch autograd.

    Kept quirk (:788): the reference builds a fresh `nn.Dropout` inside forward, which is always in training mode, so
    dropout stays active under `model.eval()`."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self.nfeature_dim, self.efeature_dim, self.output_dim = nfeature_dim, efeature_dim, output_dim
        self.hidden_dim, self.n_gnn_layers, self.K, self.dropout_rate = hidden_dim, n_gnn_layers, K, dropout_rate
        self.edge_aggr = EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, hidden_dim)
        self.convs = nn.ModuleList()
        self.convs.append(TAGConv(hidden_dim, output_dim if n_gnn_layers == 1 else hidden_dim, K=K))
        for _ in range(n_gnn_layers - 2):
            self.convs.append(TAGConv(hidden_dim, hidden_dim, K=K))
        self.convs.append(TAGConv(hidden_dim, output_dim, K=K))

    def forward(self, data):
        x = self.edge_aggr(data.x, data.edge_index, data.edge_attr)
        for conv in list(self.convs)[:-1]:
            x = conv(x, data.edge_index)
            x = torch.relu(torch.nn.functional.dropout(x, self.dropout_rate, training=True))
        return self.convs[-1](x, data.edge_index)


# ===================================================================== sibling models
class _Stale12Wide(_UndirectHelpers, nn.Module):
    """Shared front end of the reference's older model classes (`MPN`, `SkipMPN`, `MaskEmbdMPN`, `MultiMPN`,
    `MaskEmbdMultiMPN_NoMP`; train.py:30-38 lists them): they read a node tensor of width 2 * nfeature_dim + 4 -- four
    one-hot node-type columns, the features, then their mask -- which the reference's CURRENT dataset no longer produces
    (it emits 4-wide `x` and a separate `pred_mask`), so on that dataset their first line raises, there and here alike
    (the assert is kept verbatim).  On a tensor of the width they ask for they run: re-compositions of the two HIP layers
    over ONE adjacency per batch built with the model's undirect rule.

    Kept quirk: like `MPN_simplenet` they build a fresh `nn.Dropout` inside forward (:208, :281, ...), which is always in
    training mode, so dropout stays active under `model.eval()`."""

    def _split(self, data):
        assert data.x.shape[-1] == self.nfeature_dim * 2 + 4      # networks/MPN.py:194,:267,:349,:430,:625
        L.require_device(data.x, data.edge_index, data.edge_attr, *self.parameters(), what=f"{type(self).__name__} input")
        x = data.x[:, 4:4 + self.nfeature_dim]
        mask = data.x[:, -self.nfeature_dim:]
        graph = self._graphs.get(data.edge_index, data.x.shape[0], -1)     # is_directed + undirect_graph
        return x.contiguous(), mask.contiguous(), graph, L.f32c(data.edge_attr, "data.edge_attr")

    def _act(self, x):
        return torch.relu(torch.nn.functional.dropout(x, self.dropout_rate, training=True))

    def _store_dims(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        self.nfeature_dim, self.efeature_dim, self.output_dim = nfeature_dim, efeature_dim, output_dim
        self.hidden_dim, self.n_gnn_layers, self.K, self.dropout_rate = hidden_dim, n_gnn_layers, K, dropout_rate
        self._graphs = _GraphCache()

    def _conv_stack(self):
        """`convs` of MPN / SkipMPN / MaskEmbdMPN (:158-170): TAGConv(H, out if L == 1 else H), (L-2) x TAGConv(H, H),
        TAGConv(H, out)."""
        h, o, L_, K = self.hidden_dim, self.output_dim, self.n_gnn_layers, self.K
        convs = nn.ModuleList([TAGConv(h, o if L_ == 1 else h, K=K)])
        for _ in range(L_ - 2):
            convs.append(TAGConv(h, h, K=K))
        convs.append(TAGConv(h, o, K=K))
        return convs

    def _mask_mlp(self):
        return nn.Sequential(nn.Linear(self.nfeature_dim, self.hidden_dim), nn.ReLU(),
                             nn.Linear(self.hidden_dim, self.nfeature_dim))

    def _embed(self, graph, mask, x):
        """x + mask_embd(mask) (:352, :634) on the HIP GEMM."""
        a, b = self.mask_embd[0], self.mask_embd[2]
        return _hip_linear(graph, torch.relu(_hip_linear(graph, mask, a)), b) + x

    def _run_convs(self, graph, x):
        for conv in list(self.convs)[:-1]:
            x = self._act(conv.on_graph(graph, x))
        return self.convs[-1].on_graph(graph, x)

    def _run_layers(self, graph, x, edge_attr):
        """The mixed EdgeAggregation / TAGConv stack of MultiMPN and MaskEmbdMultiMPN_NoMP (:437-451, :636-648)."""
        layers = list(self.layers)
        for layer in layers[:-1]:
            x = layer.on_graph(graph, x, edge_attr) if isinstance(layer, EdgeAggregation) else layer.on_graph(graph, x)
            x = self._act(x)
        last = layers[-1]
        return last.on_graph(graph, x, edge_attr) if isinstance(last, EdgeAggregation) else last.on_graph(graph, x)


class MPN(_Stale12Wide):
    """networks/MPN.py:143-213: one EdgeAggregation, then the TAGConv stack."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self._store_dims(nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate)
        self.edge_aggr = EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, hidden_dim)
        self.convs = self._conv_stack()

    def forward(self, data):
        x, _, graph, ea = self._split(data)
        return self._run_convs(graph, self.edge_aggr.on_graph(graph, x, ea))


class SkipMPN(MPN):
    """networks/MPN.py:215-289: `MPN` plus the skip connection input_x + x (:286-287; needs output_dim == nfeature_dim)."""

    def forward(self, data):
        x, _, graph, ea = self._split(data)
        return x + self._run_convs(graph, self.edge_aggr.on_graph(graph, x, ea))


class MaskEmbdMPN(_Stale12Wide):
    """networks/MPN.py:291-371: mask embedding, one EdgeAggregation, then the TAGConv stack."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self._store_dims(nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate)
        self.edge_aggr = EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, hidden_dim)
        self.convs = self._conv_stack()
        self.mask_embd = self._mask_mlp()

    def forward(self, data):
        x, mask, graph, ea = self._split(data)
        x = self._embed(graph, mask, x)
        return self._run_convs(graph, self.edge_aggr.on_graph(graph, x, ea))


class MultiMPN(_Stale12Wide):
    """networks/MPN.py:374-453: `MaskEmbdMultiMPN` without the mask embedding (E T E T ... E)."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self._store_dims(nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate)
        h = hidden_dim
        self.layers = nn.ModuleList([EdgeAggregation(nfeature_dim, efeature_dim, h, h),
                                     TAGConv(h, output_dim if n_gnn_layers == 1 else h, K=K)])
        for _ in range(n_gnn_layers - 2):
            self.layers.append(EdgeAggregation(h, efeature_dim, h, h))
            self.layers.append(TAGConv(h, h, K=K))
        self.layers.append(EdgeAggregation(h, efeature_dim, h, output_dim))

    def forward(self, data):
        x, _, graph, ea = self._split(data)
        return self._run_layers(graph, x, ea)


class MaskEmbdMultiMPN_NoMP(_Stale12Wide):
    """networks/MPN.py:562-650: mask embedding, TAGConvs only, one closing EdgeAggregation.  Its first TAGConv takes
    hidden_dim inputs but is fed the nfeature_dim-wide embedding (:579-585, :634-641), so -- as in the reference -- it runs
    only when nfeature_dim == hidden_dim."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        self._store_dims(nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate)
        h = hidden_dim
        self.layers = nn.ModuleList([TAGConv(h, output_dim if n_gnn_layers == 1 else h, K=K)])
        for _ in range(n_gnn_layers - 2):
            self.layers.append(TAGConv(h, h, K=K))
        self.layers.append(EdgeAggregation(h, efeature_dim, h, output_dim))
        self.mask_embd = self._mask_mlp()

    def forward(self, data):
        x, mask, graph, ea = self._split(data)
        return self._run_layers(graph, self._embed(graph, mask, x), ea)
