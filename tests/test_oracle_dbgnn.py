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


def _dense_forward(params, data):
    """The whole DBGNN forward in float64 with DENSE matrices only (no index_add_, no gcn_norm): normalised adjacency of both graphs from
    the published GCN formula, bipartite operator as an incidence-matrix product, ELU, head — independent of oracle.dbgnn.forward's sparse
    message passing except for the parameter layout."""
    import torch.nn.functional as F
    d = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in data.items()}
    p = {k: v.double() for k, v in params.items()}

    def a_hat(ei, w, n):
        a = torch.zeros(n, n, dtype=torch.float64)
        loops = ei[0] == ei[1]
        a.index_put_((ei[0][~loops], ei[1][~loops]), w[~loops], accumulate=True)
        diag = torch.ones(n, dtype=torch.float64)
        diag[ei[0][loops]] = w[loops]
        a += torch.diag(diag)
        s = a.sum(0).pow(-0.5)
        s[torch.isinf(s)] = 0
        return s.unsqueeze(1) * a.t() * s.unsqueeze(0)            # [dst, src]

    n, n_ho = d["num_nodes"], d["num_ho_nodes"]
    a1 = a_hat(d["edge_index"], d["edge_weights"], n)
    a2 = a_hat(d["edge_index_higher_order"], d["edge_weights_higher_order"], n_ho)
    x, x_h = d["x"], d["x_h"]
    n_gcn = sum(1 for k in p if k.startswith("first_order_layers.") and k.endswith(".bias"))
    for i in range(n_gcn):
        x = F.elu(a1 @ (x @ p[f"first_order_layers.{i}.lin.weight"].t()) + p[f"first_order_layers.{i}.bias"])
        x_h = F.elu(a2 @ (x_h @ p[f"higher_order_layers.{i}.lin.weight"].t()) + p[f"higher_order_layers.{i}.bias"])
    bip = d["bipartite_edge_index"]
    inc = torch.zeros(n, n_ho, dtype=torch.float64)                # inc[i, j] = number of pairs (j -> i)
    inc.index_put_((bip[1], bip[0]), torch.ones(bip.size(1), dtype=torch.float64), accumulate=True)
    h_ho = x_h @ p["bipartite_layer.lin1.weight"].t() + p["bipartite_layer.lin1.bias"]
    h_fo = x @ p["bipartite_layer.lin2.weight"].t() + p["bipartite_layer.lin2.bias"]
    x = F.elu(inc @ h_ho + inc.sum(1, keepdim=True) * h_fo)
    return x @ p["lin.weight"].t() + p["lin.bias"]


def _small_bundle(seed, mapping):
    g = torch.Generator().manual_seed(seed)
    n, n_ho, f = 9, 17, 5
    ei, w, _ = _random_graph(seed, n, 30)
    ei_h, w_h, _ = _random_graph(seed + 1, n_ho, 45)
    ns = torch.randint(0, n, (n_ho, 2), generator=g)
    return {"num_nodes": n, "num_ho_nodes": n_ho, "x": torch.randn(n, f, generator=g), "x_h": torch.randn(n_ho, f, generator=g),
            "edge_index": ei, "edge_weights": w, "edge_index_higher_order": ei_h, "edge_weights_higher_order": w_h,
            "bipartite_edge_index": om.bipartite_edge_index(ns, mapping)}, f


def test_whole_forward_matches_dense_float64_for_every_bipartite_mapping():
    """the dense cross-check used to cover gcn_conv only.  Here the WHOLE oracle forward (both GCN stacks, the bipartite
    operator with the "last", "first" and "both" mappings, ELUs, head) is compared with a dense float64 evaluation."""
    for seed, mapping in ((0, "last"), (1, "first"), (2, "both")):
        data, f = _small_bundle(seed, mapping)
        for dims in ([7, 6, 4], [7, 5, 6, 4]):
            params = od.init_params(3, (f, f), dims, seed=seed)
            params = {k: (v + 0.1 if k.endswith(".bias") else v) for k, v in params.items()}      # non-zero GCN biases
            got = od.forward(params, data)
            want = _dense_forward(params, data)
            torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-6)


def test_oracle_gradients_pass_gradcheck_in_float64():
    """torch.autograd.gradcheck of the oracle's loss w.r.t. every parameter (float64, finite differences): the gradients the HIP kernels
    are compared with are the gradients of the function the oracle computes."""
    import torch.nn.functional as F
    data, f = _small_bundle(3, "both")
    data64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in data.items()}
    params = od.init_params(3, (f, f), [6, 5, 4], seed=7)
    names = sorted(params)
    leaves = [params[k].double().requires_grad_(True) for k in names]
    y = torch.randint(0, 3, (data["num_nodes"],), generator=torch.Generator().manual_seed(5))

    def loss_of(*tensors):
        return F.cross_entropy(od.forward(dict(zip(names, tensors)), data64), y)

    assert torch.autograd.gradcheck(loss_of, leaves, eps=1e-6, atol=1e-6, rtol=1e-4)
    # and loss_and_grads returns exactly these autograd gradients
    _, _, grads = od.loss_and_grads({k: v.detach() for k, v in zip(names, leaves)}, data64, y)
    loss_of(*leaves).backward()
    for k, leaf in zip(names, leaves):
        torch.testing.assert_close(grads[k], leaf.grad)
