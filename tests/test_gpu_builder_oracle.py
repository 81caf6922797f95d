"""The fused order-2 builder (pp_debruijn2_count / _fill) held DIRECTLY to the CPU oracle (oracle/model.py, oracle/dbgnn.py — the
restatement of reference multi_order_model.py:124-192, algorithms/temporal.py:17-54, lift_order.py:109-152 and PyG's gcn_norm), not
through the generic HIP kernels: for every stream both layers' ``edge_index``, the MERGED WEIGHTS of both layers, ``node_sequence``,
``inverse_idx``, E2, and the normalised coefficients / self-loop coefficients of both GCN plans.  Also the reference API on
top of it: ``MultiOrderModel.from_temporal_graph(g, delta, max_order=2)`` -> ``to_dbgnn_data`` -> ``DBGNN.forward`` reach the builder, the
layer tensors are deferred views of its CSR plans, and the model's output is the one the generic kernels give."""
import numpy as np
import pytest
import torch

from test_gpu_builder import CASES, HUB_KINDS, _hub_stream, _stream

pytestmark = pytest.mark.gpu

DELTAS = [7, 7.5, torch.tensor(7.5, dtype=torch.float64), torch.tensor(6, dtype=torch.int32), 0, 10 ** 9]
DELTA_IDS = ["int", "py-float=f32", "f64-tensor", "i32-tensor", "zero", "everything"]


def _weights(seed, m, integer=False):
    rng = np.random.default_rng(seed + 100)
    if integer:
        return torch.from_numpy(rng.integers(1, 5, m).astype(np.float32))
    return torch.from_numpy(rng.random(m).astype(np.float32) + 0.25)


def _oracle_layers(ei, t, n, delta, w):
    from oracle import model as om
    sei, st, perm = om.stable_time_sort(ei, t)
    return sei, st, (None if w is None else w[perm]), om.layers_from_temporal(sei, st, n, delta=delta, max_order=2,
                                                                              edge_weight=None if w is None else w[perm])


def _plan_edges(plan, transposed=False):
    """(source, destination, coefficient) triples of one CSR direction, on the host."""
    ptr = (plan.fwd_ptr if transposed else plan.bwd_ptr).long().cpu()
    idx = (plan.fwd_idx if transposed else plan.bwd_idx).long().cpu()
    val = (plan.fwd_val if transposed else plan.bwd_val).cpu()
    rows = torch.repeat_interleave(torch.arange(ptr.numel() - 1), ptr[1:] - ptr[:-1])
    return (idx, rows, val) if transposed else (rows, idx, val)


def _check_against_oracle(ei, t, n, delta, w, exact_weights=True):
    """One stream through ``_hip.debruijn2``; returns the number of arrays compared with the oracle."""
    from oracle import dbgnn as od
    from pathpyg_amd import _hip
    dev = torch.device("cuda:0")
    sei, st, sw, want = _oracle_layers(ei, t, n, delta, w)
    built = _hip.debruijn2(sei.to(dev), st.to(dev), n, delta, None if sw is None else sw.to(dev), want_weights=True)
    assert built is not None, "the fused builder did not take this stream"
    checked = 0
    u2, a2 = want[2]["num_nodes"], want[2]["edge_index"].size(1)
    assert built.sizes["U2"] == u2 and built.sizes["A2"] == a2 and built.sizes["A1"] == want[1]["edge_index"].size(1)
    assert built.sizes["E2"] == _e2(sei, st, n, delta), "number of lifted instance pairs"
    for k, plan in ((1, built.fo), (2, built.ho)):
        src, dst, val = _plan_edges(plan)
        assert torch.equal(torch.stack((src, dst)), want[k]["edge_index"]), f"layer {k}: edge_index (source-major rows)"
        tsrc, tdst, tval = _plan_edges(plan, transposed=True)
        # the destination-major rows hold the same edges, grouped by destination, sources ascending inside a row
        order = torch.argsort(want[k]["edge_index"][1] * plan.n_src + want[k]["edge_index"][0])
        assert torch.equal(torch.stack((tsrc, tdst)), want[k]["edge_index"][:, order]), f"layer {k}: edge_index (destination-major rows)"
        checked += 2
        # merged weights
        w_want = want[k]["edge_weight"].to(torch.float32)
        if k == 1:
            w_got = built.fo_weight.cpu()
        else:
            w_got = torch.empty(a2)
            w_got[order] = built.ho_fwd_weight.cpu()                      # destination-major -> the layer's (source-major) edge order
        if exact_weights:
            assert torch.equal(w_got, w_want), f"layer {k}: merged weights"
        else:
            torch.testing.assert_close(w_got, w_want, rtol=2e-6, atol=0)
        checked += 1
        # gcn_norm of the oracle (PyG defaults: self loops completed, symmetric normalisation by weighted in-degree)
        e = want[k]["edge_index"]
        off = e[0] != e[1]
        n_off = int(off.sum())
        _, norm = od.gcn_norm(e, w_want, want[k]["num_nodes"])           # [non-loop edges in order | one loop per node]
        torch.testing.assert_close(plan.self_coef.cpu(), norm[n_off:], rtol=2e-6, atol=1e-12)
        # off-diagonal coefficients; an existing self loop stays in the plan's rows as an entry of value 0 (its weight lives in self_coef)
        coef_want = torch.zeros(e.size(1))
        coef_want[off] = norm[:n_off]
        torch.testing.assert_close(val, coef_want, rtol=2e-6, atol=1e-12)
        torch.testing.assert_close(tval, coef_want[order], rtol=2e-6, atol=1e-12)
        checked += 3
    # bipartite "last" grouping: the order-2 nodes (., b) per first-order node b, ascending
    last = want[2]["node_sequence"][:, 1]
    ptr = built.bip.fwd_ptr.long().cpu()
    rows = torch.repeat_interleave(torch.arange(n), ptr[1:] - ptr[:-1])
    got_nodes = built.bip.fwd_idx.long().cpu()
    assert torch.equal(last[got_nodes], rows), "bipartite grouping: node (a, b) listed under b"
    assert torch.equal(torch.sort(got_nodes).values, torch.arange(u2)), "bipartite grouping: every order-2 node once"
    assert torch.equal(built.bip.self_coef.cpu(), torch.bincount(last, minlength=n).float())
    checked += 3
    return checked


def _e2(sei, st, n, delta):
    from oracle.lift import temporal_lift_sorted
    return int(temporal_lift_sorted(sei, st, delta, n).size(1))


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_fused_builder_against_the_oracle_directly(case):
    seed, m, n, span, delta, tdt, weighted = case
    ei, t = _stream(seed, m, n, span, tdt)
    w = _weights(seed, m) if weighted else None
    # non-integer weights: the oracle's index_add_ and the builder's left-to-right run sums agree to the last bit on short runs only
    assert _check_against_oracle(ei, t, n, delta, w, exact_weights=not weighted) >= 15


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4]], ids=["seed1", "seed3", "seed5"])
def test_fused_builder_integer_weights_against_the_oracle(case):
    """Integer-valued event weights: every merged weight is exact whatever the summation order."""
    seed, m, n, span, delta, tdt, _ = case
    ei, t = _stream(seed, m, n, span, tdt)
    _check_against_oracle(ei, t, n, delta, _weights(seed, m, integer=True))


@pytest.mark.parametrize("delta", DELTAS, ids=DELTA_IDS)
def test_fused_builder_delta_promotion_modes_against_the_oracle(delta):
    """torch.tensor(delta) promotion of reference temporal.py:30,43 (int64 / float32 / float64 thresholds): the oracle applies torch's own
    promotion rules, the builder's window test reproduces them in-kernel (pp_window.h)."""
    ei, t = _stream(11, 3000, 250, 900)
    _check_against_oracle(ei, t, 250, delta, None)


@pytest.mark.parametrize("kind", HUB_KINDS)
def test_fused_builder_hub_streams_against_the_oracle(kind):
    """Hub nodes (more than 64 in- / out-events) straight against the oracle: both layers' edges, merged weights, coefficients, E2."""
    ei, t, n = _hub_stream(kind)
    assert _check_against_oracle(ei, t, n, 150, None) >= 15
    w = torch.from_numpy(np.random.default_rng(23).integers(1, 4, ei.size(1)).astype(np.float32))
    _check_against_oracle(ei, t, n, 150, w)


def _graph(ei, t, n, w=None):
    import pathpyg_amd as pp
    dev = torch.device("cuda:0")
    data = pp.Data(edge_index=ei.to(dev), time=t.to(dev), num_nodes=n)
    if w is not None:
        data["edge_weight"] = w.to(dev)
    return pp.TemporalGraph(data)


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_reference_api_reaches_the_fused_builder(case):
    """from_temporal_graph(max_order=2) runs pp_debruijn2_*; every layer tensor the reference exposes equals the oracle's."""
    import pathpyg_amd as pp
    from pathpyg_amd.data import Lazy
    seed, m, n, span, delta, tdt, weighted = case
    ei, t = _stream(seed, m, n, span, tdt)
    w = _weights(seed, m, integer=True) if weighted else None
    _, _, _, want = _oracle_layers(ei, t, n, delta, w)
    mom = pp.MultiOrderModel.from_temporal_graph(_graph(ei, t, n, w), delta=delta, max_order=2)
    assert getattr(mom, "_pp_fused", None) is not None, "from_temporal_graph did not take the fused builder"
    assert isinstance(mom.layers[2].data.peek("edge_index"), Lazy), "layer tensors are deferred until read"
    assert (mom.layers[1].n, mom.layers[1].m, mom.layers[2].n, mom.layers[2].m) == (
        n, want[1]["edge_index"].size(1), want[2]["num_nodes"], want[2]["edge_index"].size(1))
    assert mom.layers[1].order == 1 and mom.layers[2].order == 2
    for k in (1, 2):
        d = mom.layers[k].data
        assert torch.equal(d.edge_index.cpu(), want[k]["edge_index"]), f"layer {k} edge_index"
        assert d.edge_index.dtype == torch.int64 and d.edge_index.is_contiguous()
        assert torch.equal(d.edge_weight.cpu(), want[k]["edge_weight"].float()), f"layer {k} edge_weight"
        assert torch.equal(d.node_sequence.cpu(), want[k]["node_sequence"]), f"layer {k} node_sequence"
        assert torch.equal(d.inverse_idx.cpu(), want[k]["inverse_idx"]), f"layer {k} inverse_idx"
        assert int(d.num_nodes) == want[k]["num_nodes"]
    only_top = pp.MultiOrderModel.from_temporal_graph(_graph(ei, t, n, w), delta=delta, max_order=2, cached=False)
    assert sorted(only_top.layers) == [2]
    assert torch.equal(only_top.layers[2].data.edge_index.cpu(), want[2]["edge_index"])
    assert torch.equal(only_top.layers[2].data.node_sequence.cpu(), want[2]["node_sequence"])


def test_fused_api_model_equals_generic_kernels_and_oracle():
    """to_dbgnn_data hands the builder's plans to DBGNN.forward: logits and gradients are those of the generic kernels (same plan arrays,
    same layer kernels) and agree with the oracle at 1e-5; the bundle's reference tensors resolve to the oracle's on demand."""
    import pathpyg_amd as pp
    import pathpyg_amd.core.multi_order_model as mm
    from oracle import dbgnn as od
    from oracle import model as om
    from tolerance import assert_embeddings_close
    dev = torch.device("cuda:0")
    ei, t = _stream(41, 6000, 300, 4000)
    n, delta = 300, 120
    _, _, _, want = _oracle_layers(ei, t, n, delta, None)
    gen = torch.Generator().manual_seed(3)
    x, x_h = torch.randn(n, 16, generator=gen), torch.randn(want[2]["num_nodes"], 16, generator=gen)
    y = torch.randint(0, 4, (n,), generator=gen)
    params = od.init_params(4, (16, 16), [32, 32, 16], seed=1)

    def run(fused):
        old = mm.FUSED_BUILDER
        mm.FUSED_BUILDER = fused
        try:
            mom = pp.MultiOrderModel.from_temporal_graph(_graph(ei, t, n), delta=delta, max_order=2)
            data = mom.to_dbgnn_data(max_order=2, mapping="last", x=x.to(dev), x_h=x_h.to(dev))
        finally:
            mm.FUSED_BUILDER = old
        net = pp.nn.DBGNN(num_classes=4, num_features=(16, 16), hidden_dims=[32, 32, 16]).to(dev)
        net.load_state_dict(params)
        out = net(data)
        loss = pp.nn.cross_entropy(out, y.to(dev))
        loss.backward()
        return mom, data, net, out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}

    mom_f, data_f, net_f, out_f, grads_f = run(True)
    mom_g, data_g, _, out_g, grads_g = run(False)
    assert getattr(data_f, "_pp_plans", None) is not None and getattr(data_g, "_pp_plans", None) is None
    assert torch.equal(out_f, out_g), "fused-builder plans and generic plans give different logits"
    from tolerance import assert_gradients_close
    ref_out, _, ref_grads = od.loss_and_grads(params, om.dbgnn_inputs(want, 2, "last", x=x, x_h=x_h), y)
    assert_embeddings_close(out_f, ref_out, what="logits")
    for k in grads_f:
        # (same plans, same kernels; the bias gradients' column sums are folded by float atomics, so two runs agree to rounding, not to the bit)
        assert_gradients_close(grads_f[k], grads_g[k], f"gradient of {k}: fused plans vs generic plans", rtol=1e-6)
        assert_gradients_close(grads_f[k], ref_grads[k], f"gradient of {k} vs the oracle")
    # the reference's bundle tensors, made on demand, are the oracle's
    ref = om.dbgnn_inputs(want, 2, "last", x=x, x_h=x_h)
    for name in ("edge_index", "edge_index_higher_order", "edge_weights", "edge_weights_higher_order", "bipartite_edge_index"):
        assert torch.equal(getattr(data_f, name).cpu(), ref[name]), name
    # resolved, unedited tensors keep the handed plans valid; a replaced tensor sends forward() to plans built from the tensors
    from pathpyg_amd.nn.dbgnn import _valid_plans
    assert _valid_plans(data_f) is not None
    assert torch.equal(net_f(data_f).detach().cpu(), out_f)
    data_f.edge_index_higher_order = data_f.edge_index_higher_order.clone()
    assert _valid_plans(data_f) is None
    assert torch.equal(net_f(data_f).detach().cpu(), out_f)
    # other mappings build their bundle from the (deferred) layer tensors
    both = mom_f.to_dbgnn_data(max_order=2, mapping="both", x=x.to(dev), x_h=x_h.to(dev))
    assert getattr(both, "_pp_plans", None) is None
    assert torch.equal(both.bipartite_edge_index.cpu(), om.bipartite_edge_index(want[2]["node_sequence"], "both"))


def test_fused_api_unsorted_or_host_streams_take_the_generic_path():
    import pathpyg_amd as pp
    ei, t = _stream(43, 800, 60, 500)
    n = 60
    _, _, _, want = _oracle_layers(ei, t, n, 20, None)
    # host-resident stream: staged by the generic shims, result on the host
    host = pp.MultiOrderModel.from_temporal_graph(pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n)), delta=20, max_order=2)
    assert getattr(host, "_pp_fused", None) is None
    assert torch.equal(host.layers[2].data.edge_index.cpu(), want[2]["edge_index"])
    # a stream whose time was shuffled after construction: the builder reports it and steps aside; the generic path raises as before
    # (the reference lifts g.data as it is, multi_order_model.py:167 — there the result would silently be wrong)
    g = _graph(ei, t, n)
    perm = torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(0)).to(g.data.edge_index.device)
    g.data.edge_index = g.data.edge_index[:, perm].contiguous()
    g.data.time = g.data.time[perm].contiguous()
    with pytest.raises(ValueError):
        pp.MultiOrderModel.from_temporal_graph(g, delta=20, max_order=2)


def test_fused_api_objects_are_freed_without_the_cyclic_collector():
    """A model / bundle from the fused path holds gigabytes of plans at the bench sizes: dropping the last reference must free them at once
    (a bag -> Lazy -> closure -> bag cycle kept them until the cyclic collector ran: 14.0 -> 20.8 ms per API step from the 11th step on)."""
    import gc
    import weakref

    import pathpyg_amd as pp
    ei, t = _stream(45, 3000, 200, 2000)
    g = _graph(ei, t, 200)
    dev = torch.device("cuda:0")
    gc.collect()
    gc.disable()
    try:
        mom = pp.MultiOrderModel.from_temporal_graph(g, delta=50, max_order=2)
        data = mom.to_dbgnn_data(max_order=2, x=torch.randn(200, 8, device=dev), x_h=torch.randn(mom.layers[2].n, 8, device=dev))
        _ = mom.layers[2].data.edge_weight, data.edge_index_higher_order                 # resolved and unresolved tensors alike
        refs = [weakref.ref(mom), weakref.ref(mom.layers[1].data), weakref.ref(mom.layers[2].data), weakref.ref(mom.layers[2]), weakref.ref(data)]
        del mom, data, _
        assert [r() is None for r in refs] == [True] * len(refs)
    finally:
        gc.enable()
