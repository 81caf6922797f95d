"""DBGNN oracle self-consistency (CPU).  The reference does not pin DBGNN numerics
(tests/nn/test_dbgnn.py:43 only asserts ``out is not None``), so the oracle's sparse GCN is
cross-checked against an independent dense-matrix evaluation of the published formula."""
import torch

from oracle import dbgnn as od
from oracle import model as om


def _random_graph(seed, n, e, loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if loops:
        ei[:, : max(1, e // 10)] = torch.randint(0, n, (1, max(1, e // 10)), generator=g).repeat(2, 1)
    w = torch.rand(e, generator=g) + 0.25
    return ei, w, g


def test_gcn_matches_dense_formula():
    for seed, n, e in [(0, 12, 40), (1, 50, 400), (2, 7, 5)]:
        ei, w, g = _random_graph(seed, n, e)
        x = torch.randn(n, 6, generator=g)
        wt = torch.randn(5, 6, generator=g)
        b = torch.randn(5, generator=g)
        sparse = od.gcn_conv(x, ei, w, wt, b)
        dense = od.dense_gcn(x, ei, w, wt, b)
        torch.testing.assert_close(sparse.double(), dense, rtol=1e-5, atol=1e-6)


def test_gcn_norm_self_loop_rules():
    # node 0 has a self loop of weight 3 (kept), node 1 none (gets weight 1), node 2 is isolated
    ei = torch.tensor([[0, 0, 1], [0, 1, 0]])
    w = torch.tensor([3.0, 2.0, 5.0])
    idx, norm = od.gcn_norm(ei, w, 3)
    assert idx.tolist() == [[0, 1, 0, 1, 2], [1, 0, 0, 1, 2]]
    deg = torch.tensor([5.0 + 3.0, 2.0 + 1.0, 1.0])
    want = torch.tensor([2.0, 5.0, 3.0, 1.0, 1.0]) / torch.sqrt(deg[idx[0]] * deg[idx[1]])
    torch.testing.assert_close(norm, want)


def test_forward_shapes_and_state_dict_names():
    paths = om.walks_to_path_tensors([[0, 2, 3], [0, 2, 3], [1, 2, 4], [1, 2, 4]], [1.0] * 4)
    layers = om.layers_from_paths(paths, max_order=2)
    data = om.dbgnn_inputs(layers)
    params = od.init_params(2, (data["num_nodes"], data["num_ho_nodes"]), [16, 32, 8], seed=1)
    assert sorted(params) == sorted([
        "first_order_layers.0.lin.weight", "first_order_layers.0.bias", "first_order_layers.1.lin.weight", "first_order_layers.1.bias",
        "higher_order_layers.0.lin.weight", "higher_order_layers.0.bias", "higher_order_layers.1.lin.weight", "higher_order_layers.1.bias",
        "bipartite_layer.lin1.weight", "bipartite_layer.lin1.bias", "bipartite_layer.lin2.weight", "bipartite_layer.lin2.bias",
        "lin.weight", "lin.bias"])
    out = od.forward(params, data)
    assert out.shape == (5, 2) and torch.isfinite(out).all()
    y = torch.tensor([0, 0, 1, 1, 1])
    _, loss, grads = od.loss_and_grads(params, data, y)
    assert torch.isfinite(loss) and all(torch.isfinite(g).all() for g in grads.values())
