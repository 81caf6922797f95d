"""Oracle (test infrastructure): DBGNN forward on CPU with torch autograd.

PARITY UNPINNED by the reference: its only DBGNN test asserts ``out is not None``
(reference tests/nn/test_dbgnn.py:33-43).  This file restates
  * ``DBGNN.__init__/forward``          reference src/pathpyG/nn/dbgnn.py:86-151
  * ``BipartiteGraphOperator``          reference src/pathpyG/nn/dbgnn.py:39-69
and, from the published GCN formula / PyG 2.7.0 documented behaviour (not in
the reference tree): ``gcn_norm`` with
``add_remaining_self_loops(fill=1)``, ``GCNConv`` (bias-free linear map,
normalised sum over incoming edges, then bias), ``MessagePassing("add")``.
Message passing uses ``index_add_`` — what PyG dispatches to on CPU.
``dense_gcn`` is an independent dense-matrix evaluation used to cross-check.

Parameter names/shapes follow the reference module's ``state_dict`` so
checkpoints line up: ``first_order_layers.{i}.lin.weight`` [out,in],
``first_order_layers.{i}.bias`` [out], same for ``higher_order_layers``,
``bipartite_layer.lin1/lin2.{weight,bias}``, ``lin.{weight,bias}``.

Not product code: see oracle/__init__.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def gcn_norm(edge_index: torch.Tensor, edge_weight: torch.Tensor, num_nodes: int):
    """Self-loop completion + symmetric normalisation.  Returns (edge_index', norm)."""
    row, col = edge_index[0], edge_index[1]
    not_loop = row != col
    loop_w = edge_weight.new_ones(num_nodes)
    loop_w[row[~not_loop]] = edge_weight[~not_loop]       # existing loop keeps its weight (last wins)
    ids = torch.arange(num_nodes)
    row = torch.cat((row[not_loop], ids))
    col = torch.cat((col[not_loop], ids))
    w = torch.cat((edge_weight[not_loop], loop_w))
    deg = torch.zeros(num_nodes, dtype=w.dtype).index_add_(0, col, w)
    dinv = deg.pow(-0.5)
    dinv[dinv == float("inf")] = 0
    return torch.stack((row, col)), dinv[row] * w * dinv[col]


def gcn_conv(x, edge_index, edge_weight, weight, bias):
    n = x.size(0)
    ei, norm = gcn_norm(edge_index, edge_weight, n)
    h = x @ weight.t()
    out = torch.zeros(n, h.size(1), dtype=h.dtype).index_add_(0, ei[1], norm.unsqueeze(1) * h[ei[0]])
    return out + bias


def bipartite_op(x_h, x, bip_index, n_fo, w1, b1, w2, b2):
    """out[i] = sum over (j -> i) of (lin2(x)[i] + lin1(x_h)[j])  (dbgnn.py:64-69)."""
    h_ho = x_h @ w1.t() + b1
    h_fo = x @ w2.t() + b2
    msg = h_fo[bip_index[1]] + h_ho[bip_index[0]]
    return torch.zeros(n_fo, msg.size(1), dtype=msg.dtype).index_add_(0, bip_index[1], msg)


def init_params(num_classes: int, num_features, hidden_dims, seed: int = 0) -> dict:
    """Glorot weights / zero GCN biases / torch-Linear-style init for the dense layers."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def glorot(o, i):
        a = math.sqrt(6.0 / (i + o))
        return (torch.rand(o, i, generator=g) * 2 - 1) * a

    def linear(o, i):
        a = 1.0 / math.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * a, (torch.rand(o, generator=g) * 2 - 1) * a

    n_gcn = len(hidden_dims) - 1
    for stack, f_in in (("first_order_layers", num_features[0]), ("higher_order_layers", num_features[1])):
        dims = [f_in] + list(hidden_dims[:n_gcn])
        for i in range(n_gcn):
            p[f"{stack}.{i}.lin.weight"] = glorot(dims[i + 1], dims[i])
            p[f"{stack}.{i}.bias"] = torch.zeros(dims[i + 1])
    for name in ("lin1", "lin2"):
        w, b = linear(hidden_dims[-1], hidden_dims[-2])
        p[f"bipartite_layer.{name}.weight"], p[f"bipartite_layer.{name}.bias"] = w, b
    p["lin.weight"], p["lin.bias"] = linear(num_classes, hidden_dims[-1])
    return p


def forward(params: dict, data: dict) -> torch.Tensor:
    """DBGNN.forward with p_dropout = 0 (dbgnn.py:127-151)."""
    x, x_h = data["x"], data["x_h"]
    n_gcn = sum(1 for k in params if k.startswith("first_order_layers.") and k.endswith(".bias"))
    for i in range(n_gcn):
        x = F.elu(gcn_conv(x, data["edge_index"], data["edge_weights"],
                           params[f"first_order_layers.{i}.lin.weight"], params[f"first_order_layers.{i}.bias"]))
    for i in range(n_gcn):
        x_h = F.elu(gcn_conv(x_h, data["edge_index_higher_order"], data["edge_weights_higher_order"],
                             params[f"higher_order_layers.{i}.lin.weight"], params[f"higher_order_layers.{i}.bias"]))
    x = F.elu(bipartite_op(x_h, x, data["bipartite_edge_index"], data["num_nodes"],
                           params["bipartite_layer.lin1.weight"], params["bipartite_layer.lin1.bias"],
                           params["bipartite_layer.lin2.weight"], params["bipartite_layer.lin2.bias"]))
    return x @ params["lin.weight"].t() + params["lin.bias"]


def loss_and_grads(params: dict, data: dict, y: torch.Tensor, mask: torch.Tensor | None = None):
    """Cross-entropy training objective of the bench's train step + gradients of every parameter."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = forward(leaves, data)
    loss = F.cross_entropy(out if mask is None else out[mask], y if mask is None else y[mask])
    loss.backward()
    return out.detach(), loss.detach(), {k: v.grad for k, v in leaves.items()}


def dense_gcn(x, edge_index, edge_weight, weight, bias):
    """diag(s) A^T diag(s) (X W^T) + b with A[j,i] = total weight of edges j->i."""
    n = x.size(0)
    a = torch.zeros(n, n, dtype=torch.float64)
    row, col = edge_index[0], edge_index[1]
    loops = row == col
    a.index_put_((row[~loops], col[~loops]), edge_weight[~loops].double(), accumulate=True)
    diag = torch.ones(n, dtype=torch.float64)
    diag[row[loops]] = edge_weight[loops].double()
    a += torch.diag(diag)
    d = a.sum(0)
    s = d.pow(-0.5)
    s[torch.isinf(s)] = 0
    h = x.double() @ weight.double().t()
    return (s.unsqueeze(1) * (a.t() @ (s.unsqueeze(1) * h))) + bias.double()
