"""Stand-in for torch_geometric.nn: MessagePassing, TAGConv (+ importable GCNConv/ChebConv names).

Restates PyG's published semantics (see package docstring).  Call sites in the reference:
MessagePassing.__init__(aggr='add') networks/MPN.py:11; self.propagate(...) :53;
TAGConv(hidden, hidden, K=K) :477,480,484; TAGConv.forward(x=, edge_index=) :545.
"""
import inspect
import math

import torch
import torch.nn as nn


class MessagePassing(nn.Module):
    """flow='source_to_target': *_j <- edge_index[0] (source), *_i <- edge_index[1] (target), messages are summed onto
    edge_index[1]; flow='target_to_source' swaps the roles (*_i <- edge_index[0], summed onto edge_index[0]).  kwargs that
    message() does not name are dropped; update() gets the aggregate plus the kwargs it names (default: identity)."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        if aggr not in ('add', 'sum'):
            raise NotImplementedError("stand-in supports aggr='add' only")
        if flow not in ('source_to_target', 'target_to_source'):
            raise ValueError(flow)
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = (0, 1) if self.flow == 'source_to_target' else (1, 0)
        num_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v) and v.dim() >= 2:
                num_nodes = v.size(0)
                break
        if size is not None:
            num_nodes = size[1] if isinstance(size, (tuple, list)) else size
        wanted = [p for p in inspect.signature(self.message).parameters]
        margs = {}
        for name in wanted:
            if name.endswith('_i') or name.endswith('_j'):
                src = kwargs[name[:-2]]
                if num_nodes is None:
                    num_nodes = src.size(0)
                margs[name] = src.index_select(0, edge_index[i if name.endswith('_i') else j])
            else:
                margs[name] = kwargs[name]
        msg = self.message(**margs)
        out = torch.zeros((num_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out = out.index_add(0, edge_index[i], msg)
        # PyG hands update() the propagate kwargs its signature names (PowerImbalance.update(aggregated, x),
        # utils/custom_loss_functions.py:229)
        uargs = {name: kwargs[name] for name in list(inspect.signature(self.update).parameters)[1:] if name in kwargs}
        return self.update(out, **uargs)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def gcn_norm(edge_index, num_nodes, dtype):
    """PyG gcn_norm(edge_weight=None, improved=False, add_self_loops=False, flow='source_to_target'):
    w_e = d(src)^-1/2 * d(dst)^-1/2 with d = in-degree on edge_index[1] counted with multiplicity."""
    row, col = edge_index[0], edge_index[1]
    w = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    deg = torch.zeros(num_nodes, dtype=dtype, device=edge_index.device).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0.)
    return dis[row] * w * dis[col]


class TAGConv(MessagePassing):
    """out = sum_{k=0..K} (A_hat^k x) W_k^T + b, A_hat = D^-1/2 A D^-1/2, no self loops.
    state_dict keys: lins.{k}.weight (out,in), bias (out) -- zero-initialised."""

    def __init__(self, in_channels, out_channels, K=3, bias=True, normalize=True, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        self.in_channels, self.out_channels, self.K, self.normalize = in_channels, out_channels, K, normalize
        self.lins = nn.ModuleList([nn.Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)

    def forward(self, x, edge_index, edge_weight=None):
        if self.normalize:
            edge_weight = gcn_norm(edge_index, x.size(0), x.dtype)
        out = self.lins[0](x)
        for lin in self.lins[1:]:
            x = self.propagate(edge_index, x=x, edge_weight=edge_weight)
            out = out + lin(x)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_weight):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j


class GCNConv(MessagePassing):  # importable name only (networks/MPN.py:3); out-of-scope siblings use it
    def __init__(self, *a, **k):
        raise NotImplementedError("GCNConv is outside the hot path; the stand-in only provides the name")


class ChebConv(MessagePassing):  # importable name only
    def __init__(self, *a, **k):
        raise NotImplementedError("ChebConv is outside the hot path; the stand-in only provides the name")
