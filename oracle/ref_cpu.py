"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MaskEmbdMultiMPN hot path.

A build-authored, pure-torch (no torch_geometric) restatement of the *reference dataflow* of
/root/reference/networks/MPN.py: materialised cat[x_i, x_j, e], per-edge Linear-ReLU-Linear,
index_add aggregation, TAGConv as K gather-scale-scatter hops + (K+1) GEMMs.  It is

  * the parity checker for the HIP path in tests/ (-m gpu) and __graft_entry__.smoke(),
  * the timed `cpu_baseline` leg of bench.py (kind "port"),

and NOTHING else: the product (poweflownet_amd/) never imports it and has no CPU fallback.

How it is pinned: tests/test_oracle.py checks every function here against tests/golden/*.npz,
which oracle/make_goldens.py produced IN THE BUILD CONTAINER by importing the reference's own
networks/MPN.py unmodified (through oracle/pyg_standin for the five PyG symbols it needs).
The PyG primitives underneath (propagate/TAGConv/degree) are third-party, unpinned by the
reference and absent from the image -> parity at *that* boundary is UNPINNED against PyG itself;
it is cross-checked analytically (dense sum_k A_hat^k X W_k^T + b, explicit per-edge loops).

Reference lines each function follows are cited in its docstring.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- graph helpers
def is_directed(edge_index: torch.Tensor) -> bool:
    """networks/MPN.py:498-504 -- one-edge heuristic: with (u0, v0) the first stored edge, the
    list is 'directed' iff no edge (v0 -> u0) exists.  E == 0 -> False."""
    if edge_index.shape[1] == 0:
        return False
    u0, v0 = edge_index[0, 0], edge_index[1, 0]
    back = edge_index[1][edge_index[0] == v0]
    return not bool((back == u0).any())


def undirect_graph(edge_index: torch.Tensor, edge_attr: torch.Tensor):
    """networks/MPN.py:506-523 -- originals first, reversed copies second; attrs duplicated."""
    if not is_directed(edge_index):
        return edge_index, edge_attr
    rev = edge_index.flip(0)
    return torch.cat([edge_index, rev], dim=1), torch.cat([edge_attr, edge_attr], dim=0)


def in_degree(edge_index: torch.Tensor, num_nodes: int, dtype=torch.float32) -> torch.Tensor:
    """PyG utils.degree / gcn_norm degree: occurrences of each node in edge_index[1] (multiplicity kept)."""
    deg = torch.zeros(num_nodes, dtype=dtype, device=edge_index.device)
    return deg.index_add_(0, edge_index[1], torch.ones(edge_index.shape[1], dtype=dtype, device=edge_index.device))


# --------------------------------------------------------------------------- the two layers
def gated_relu(z, gate=None, record=None):
    """relu(z) -- or, test hook, z * gate with the {0,1} decisions supplied from outside (the HIP path exports the ones its
    forward took, pfn_mpn_export_gates): the same function on the branch those decisions select, so that a float64 run and an
    fp32 run are differentiated on the SAME piecewise-linear piece.  `record`: list that receives this call's own z > 0."""
    if record is not None:
        record.append((z > 0).detach())
    return F.relu(z) if gate is None else z * gate.to(z.dtype)


def edge_aggregation(x, edge_index, edge_attr, w1, b1, w2, b2, gate=None, record=None):
    """EdgeAggregation.forward / .message (networks/MPN.py:23-56) under PyG propagate(aggr='add'):
    out[i] = sum_{e: dst(e)=i} ( W2 relu(W1 [x_i ; x_src(e) ; a_e] + b1) + b2 ).
    Concat order target, source, attr (:28).  The degree `norm` computed at :43-47 is dead."""
    src, dst = edge_index[0], edge_index[1]
    z = torch.cat([x.index_select(0, dst), x.index_select(0, src), edge_attr], dim=-1)
    msg = F.linear(gated_relu(F.linear(z, w1, b1), gate, record), w2, b2)
    out = torch.zeros(x.shape[0], msg.shape[1], dtype=msg.dtype, device=msg.device)
    return out.index_add(0, dst, msg)


def tag_conv(x, edge_index, lin_weights, bias):
    """PyG TAGConv.forward (call sites networks/MPN.py:477-484, :545), normalize=True,
    add_self_loops=False: out = sum_k (A_hat^k x) W_k^T + b."""
    src, dst = edge_index[0], edge_index[1]
    n = x.shape[0]
    dis = in_degree(edge_index, n, x.dtype).pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    w = dis.index_select(0, src) * dis.index_select(0, dst)
    out = F.linear(x, lin_weights[0])
    for wk in lin_weights[1:]:
        x = torch.zeros_like(x).index_add(0, dst, w.unsqueeze(1) * x.index_select(0, src))
        out = out + F.linear(x, wk)
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------- modules (same state_dict keys)
class EdgeAggregation(nn.Module):
    """Mirror of networks/MPN.py:6-56: ctor (nfeature_dim, efeature_dim, hidden_dim, output_dim),
    parameters under edge_aggr.0 / edge_aggr.2."""

    def __init__(self, nfeature_dim, efeature_dim, hidden_dim, output_dim):
        super().__init__()
        self.nfeature_dim, self.efeature_dim, self.output_dim = nfeature_dim, efeature_dim, output_dim
        self.edge_aggr = nn.Sequential(nn.Linear(2 * nfeature_dim + efeature_dim, hidden_dim), nn.ReLU(),
                                       nn.Linear(hidden_dim, output_dim))

    def forward(self, x, edge_index, edge_attr, gate=None, record=None):
        l1, l2 = self.edge_aggr[0], self.edge_aggr[2]
        return edge_aggregation(x, edge_index, edge_attr, l1.weight, l1.bias, l2.weight, l2.bias, gate, record)


class TAGConv(nn.Module):
    """Mirror of PyG TAGConv(in_channels, out_channels, K): lins.{k}.weight, bias (zero init)."""

    def __init__(self, in_channels, out_channels, K=3):
        super().__init__()
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        return tag_conv(x, edge_index, [l.weight for l in self.lins], self.bias)


class MaskEmbdMultiMPN(nn.Module):
    """Mirror of networks/MPN.py:456-559 (ctor :462-496, forward :525-559)."""

    def __init__(self, nfeature_dim, efeature_dim, output_dim, hidden_dim, n_gnn_layers, K, dropout_rate):
        super().__init__()
        if n_gnn_layers < 2:
            raise ValueError("n_gnn_layers == 1 is shape-broken in the reference (networks/MPN.py:475-477)")
        self.nfeature_dim, self.efeature_dim, self.output_dim = nfeature_dim, efeature_dim, output_dim
        self.hidden_dim, self.n_gnn_layers, self.K, self.dropout_rate = hidden_dim, n_gnn_layers, K, dropout_rate
        layers = [EdgeAggregation(nfeature_dim, efeature_dim, hidden_dim, hidden_dim), TAGConv(hidden_dim, hidden_dim, K)]
        for _ in range(n_gnn_layers - 2):
            layers += [EdgeAggregation(hidden_dim, efeature_dim, hidden_dim, hidden_dim), TAGConv(hidden_dim, hidden_dim, K)]
        layers.append(EdgeAggregation(hidden_dim, efeature_dim, hidden_dim, output_dim))
        self.layers = nn.ModuleList(layers)
        self.mask_embd = nn.Sequential(nn.Linear(nfeature_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, nfeature_dim))
        self.dropout = nn.Dropout(dropout_rate)
        # test hook: per hidden layer a {0,1} keep mask (N, H).  When set, dropout(x) is x * keep / (1 - p) -- nn.Dropout's
        # definition with the Bernoulli draw supplied from outside (the HIP path exports its masks: pfn_dropout_mask)
        self.dropout_masks = None
        # test hook: externally supplied ReLU decisions {"edge": {layer: (E, H)}, "out": {layer: (N, H)}, "mask_embd": (N, H)}
        # (what MaskEmbdMultiMPN.export_gates of the HIP mirror returns); every relu(z) becomes z * gate (see gated_relu)
        self.gates = None
        self.recorded_gates = None     # set to {} before a forward: receives this run's own decisions in the same layout

    is_directed = staticmethod(is_directed)
    undirect_graph = staticmethod(undirect_graph)

    def forward(self, data, return_intermediates: bool = False):
        assert data.x.shape[-1] == 4                                       # :528
        # `.float()` in the reference (:533); cast to x's dtype so the same oracle also runs in float64 as a yardstick
        gt, rec = self.gates, self.recorded_gates
        if rec is not None:
            rec.update({"edge": {}, "out": {}})

        def hook(kind, li):
            """(gate, record list) of one ReLU site; the record list's single entry is moved into `rec` by `keep`."""
            g = None if gt is None else (gt[kind] if li is None else gt[kind][li])
            return g, ([] if rec is not None else None)

        def keep(kind, li, lst):
            if rec is not None:
                if li is None:
                    rec[kind] = lst[0]
                else:
                    rec[kind][li] = lst[0]
        g, r = hook("mask_embd", None)
        a, b = self.mask_embd[0], self.mask_embd[2]
        x = b(gated_relu(a(data.pred_mask.to(data.x.dtype)), g, r)) + data.x   # :533,:537 (mask_embd = Linear, ReLU, Linear :491-495)
        keep("mask_embd", None, r)
        edge_index, edge_attr = undirect_graph(data.edge_index, data.edge_attr)   # :539
        inter = [x]

        def run_layer(li, layer, x):
            if not isinstance(layer, EdgeAggregation):
                return layer(x, edge_index)
            g, r = hook("edge", li)
            y = layer(x, edge_index, edge_attr, g, r)
            keep("edge", li, r)
            return y
        for li, layer in enumerate(self.layers[:-1]):                      # :541-547
            x = run_layer(li, layer, x)
            inter.append(x)                                                # pre-activation layer output
            if self.dropout_masks is not None:
                x = x * self.dropout_masks[li].to(x.dtype) / (1.0 - self.dropout_rate)
            else:
                x = self.dropout(x)
            g, r = hook("out", li)
            x = gated_relu(x, g, r)
            keep("out", li, r)
        x = run_layer(len(self.layers) - 1, self.layers[-1], x)            # :554-555
        inter.append(x)
        return (x, inter) if return_intermediates else x


def masked_l2_loss(output, target, mask, regularize=True, regcoeff=1):
    """Masked_L2_loss.forward (utils/custom_loss_functions.py:30-46), the reference's default loss
    (utils/argument_parser.py:36): mean squared error over the entries with mask != 0, plus regcoeff x the mean over the
    entries with (1 - mask) != 0.  A mask value outside {0, 1} is in both sets; an empty set gives NaN (mean of nothing)."""
    d2 = (output - target) ** 2
    sel = mask != 0
    loss = d2[sel].sum() / sel.sum()
    if regularize:
        rest = (1 - mask) != 0
        loss = loss + regcoeff * (d2[rest].sum() / rest.sum())
    return loss


def power_imbalance(x, edge_index, edge_attr, xymean, xystd, edgemean, edgestd):
    """PowerImbalance.forward (utils/custom_loss_functions.py:248-281) with its message (:159-228) and update (:229-246):
    undirect a stored-once list, de-normalise (x * std + mean, no epsilon, :127-132), then with flow='target_to_source'
    (i = edge_index[0], j = edge_index[1], aggregation onto i):
        e = Vm cos(Va pi/180), f = Vm sin(Va pi/180), g = r / (r^2 + x^2), b = -x / (r^2 + x^2)
        Pji = g (e_i e_j - e_i^2 + f_i f_j - f_i^2) + b (f_i e_j - e_i f_j)
        Qji = g (f_i e_j - e_i f_j) + b (-e_i e_j + e_i^2 - f_i f_j + f_i^2)
        dP_i = P_i - sum_j Pji,  dQ_i = Q_i - sum_j Qji,  loss = mean_i (dP_i^2 + dQ_i^2)."""
    edge_index, edge_attr = undirect_graph(edge_index, edge_attr)
    xd = x * xystd.to(x.dtype) + xymean.to(x.dtype)
    ed = edge_attr * edgestd.to(x.dtype) + edgemean.to(x.dtype)
    i, j = edge_index[0], edge_index[1]
    r, xx = ed[:, 0], ed[:, 1]
    g, b = r / (r ** 2 + xx ** 2), -xx / (r ** 2 + xx ** 2)
    vm, va = xd[:, 0], xd[:, 1] * (torch.pi / 180.0)
    e, f = vm * torch.cos(va), vm * torch.sin(va)
    ei_, fi_, ej_, fj_ = e[i], f[i], e[j], f[j]
    pji = g * (ei_ * ej_ - ei_ ** 2 + fi_ * fj_ - fi_ ** 2) + b * (fi_ * ej_ - ei_ * fj_)
    qji = g * (fi_ * ej_ - ei_ * fj_) + b * (-ei_ * ej_ + ei_ ** 2 - fi_ * fj_ + fi_ ** 2)
    agg = torch.zeros(x.shape[0], 2, dtype=x.dtype).index_add(0, i, torch.stack([pji, qji], dim=1))
    dpq = xd[:, 2:4] - agg
    return dpq.square().sum(dim=-1).mean()


def mixed_mse_power_imbalance(x, edge_index, edge_attr, y, xymean, xystd, edgemean, edgestd, alpha=0.5):
    """MixedMSEPoweImbalance.forward (:300-306): alpha * MSE(x, y) + (1 - alpha) * 0.020 * power_imbalance."""
    return alpha * F.mse_loss(x, y) + (1 - alpha) * 0.020 * power_imbalance(x, edge_index, edge_attr, xymean, xystd, edgemean,
                                                                           edgestd)


def train_step(model, data, optimizer, loss_fn=None):
    """The per-batch body of train_epoch (utils/training.py:55-77) for the default-else loss branch
    (:72): zero_grad -> forward -> loss(out, y) -> backward -> step.  Returns the loss tensor."""
    loss_fn = loss_fn if loss_fn is not None else nn.MSELoss()
    optimizer.zero_grad()
    out = model(data)
    loss = loss_fn(out, data.y)
    loss.backward()
    optimizer.step()
    return loss.detach()
