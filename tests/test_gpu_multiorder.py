"""GPU parity tests of the level-by-level multi-order builder (``pp_multiorder_prepare`` / ``pp_multiorder_step``: what
``MultiOrderModel.from_temporal_graph(max_order >= 3)`` runs on device-resident streams) against the CPU oracle
(reference src/pathpyG/core/multi_order_model.py:83-192, algorithms/lift_order.py:48-152, algorithms/temporal.py:17-54) and, at sizes the
oracle does not finish, against the generic kernels (``pp_temporal_*`` / ``pp_linegraph_*`` / ``pp_coalesce_*``), tensor by tensor, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import pathpyg_amd
    return pathpyg_amd


def _level_by_level(model) -> bool:
    return "layers" in getattr(model, "sizes", {})


def _stream(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "sparse":            # every node a handful of events: one instance per type, 0..3 children
        m, n, span, delta = 20_000, 2_000, 100_000, 8_000
        ei = rng.integers(0, n, (2, m))
    elif kind == "hubs":            # Zipf targets: node pairs with several events, types with tens of children
        m, n, span, delta = 20_000, 1_500, 60_000, 3_000
        ei = np.stack((rng.integers(0, n, m), (n * rng.random(m) ** 5).astype(np.int64)))
    elif kind == "contact":         # few nodes, every node a hub on both sides: hundreds of children per type (workgroup kernel)
        m, n, span, delta = 6_000, 12, 60_000, 150
        ei = rng.integers(0, n, (2, m))
    elif kind == "ties":            # many events per timestamp: windows that start behind a run of ties
        m, n, span, delta = 8_000, 60, 300, 6
        ei = rng.integers(0, n, (2, m))
    elif kind == "loops":           # self loops and repeated pairs
        m, n, span, delta = 5_000, 25, 4_000, 40
        ei = rng.integers(0, n, (2, m))
        ei[1, ::3] = ei[0, ::3]
    else:
        raise ValueError(kind)
    t = rng.integers(0, span, m)
    w = rng.integers(1, 4, m).astype(np.float32)
    return torch.from_numpy(ei), torch.from_numpy(t), torch.from_numpy(w), n, delta


def _check_against_oracle(pp, ei, t, w, n, delta, K, cached, weighted, float_time=False):
    from oracle import model as om
    tt = t.double() / 4 if float_time else t
    dd = delta / 4 if float_time else delta
    data = pp.Data(edge_index=ei.to(DEV), time=tt.to(DEV), num_nodes=n)
    if weighted:
        data["edge_weight"] = w.to(DEV)
    g = pp.TemporalGraph(data)
    sei, st, perm = om.stable_time_sort(ei, tt)
    want = om.layers_from_temporal(sei, st, n, delta=dd, max_order=K, edge_weight=w[perm] if weighted else None, cached=cached)
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=dd, max_order=K, cached=cached)
    assert sorted(model.layers) == sorted(want)
    for k in want:
        d = model.layers[k].data
        for key in ("edge_index", "edge_weight", "node_sequence", "inverse_idx"):
            assert torch.equal(d[key].cpu(), want[k][key]), (k, key)
        assert d.num_nodes == want[k]["num_nodes"] and model.layers[k].order == k
    return model


@pytest.mark.parametrize("kind", ["sparse", "hubs", "contact", "ties", "loops"])
@pytest.mark.parametrize("weighted", [False, True])
def test_levels_equal_the_oracle(pp, kind, weighted):
    ei, t, w, n, delta = _stream(kind, 11)
    model = _check_against_oracle(pp, ei, t, w, n, delta, 4, True, weighted)
    assert _level_by_level(model), "from_temporal_graph(max_order=4) did not take the level-by-level builder"
    # every instance the reference would have lifted is accounted for: instances of level k+1 = sum of the children of level k
    sizes = model.sizes["layers"]
    assert sizes[0][2] == ei.size(1) and all(sizes[k][0] == sizes[k - 1][1] for k in range(1, len(sizes)))


@pytest.mark.parametrize("K,cached,float_time", [(3, False, False), (5, True, True), (5, False, False), (3, True, True)])
def test_top_layer_only_and_float_time(pp, K, cached, float_time):
    ei, t, w, n, delta = _stream("hubs", 5)
    model = _check_against_oracle(pp, ei, t, w, n, delta, K, cached, True, float_time)
    assert _level_by_level(model)
    assert sorted(model.layers) == (list(range(1, K + 1)) if cached else [K])


def test_more_children_than_a_workgroup_sorts_falls_back(pp):
    # two nodes, every event continues into hundreds of others: a type with more than 4096 children — the builder reports it, the
    # generic kernels take the stream, the layers are the oracle's
    from oracle import model as om
    rng = np.random.default_rng(3)
    m, n = 1_500, 2
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    t = torch.from_numpy(rng.integers(0, 600, m))
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=60, max_order=3)
    assert not _level_by_level(model)
    sei, st, _ = om.stable_time_sort(ei, t)
    want = om.layers_from_temporal(sei, st, n, delta=60, max_order=3)
    for k in want:
        d = model.layers[k].data
        for key in ("edge_index", "edge_weight", "node_sequence", "inverse_idx"):
            assert torch.equal(d[key].cpu(), want[k][key]), (k, key)


def test_empty_layers_and_host_streams_take_the_generic_kernels(pp):
    # no event continues another one: layer 2 has nodes and no edges, layer 3 no nodes (the reference's empty tensors)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    t = torch.tensor([10, 5, 1])
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=4))
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=2, max_order=3)
    assert not _level_by_level(model)
    assert [(model.layers[k].n, model.layers[k].m) for k in (1, 2, 3)] == [(4, 3), (3, 0), (0, 0)]
    g_cpu = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=4))
    assert not _level_by_level(pp.MultiOrderModel.from_temporal_graph(g_cpu, delta=20, max_order=3))


def _generic(pp, g, delta, K):
    from pathpyg_amd.core import multi_order_model as mm
    mm.FUSED_BUILDER = False
    try:
        return pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K)
    finally:
        mm.FUSED_BUILDER = True


@pytest.mark.parametrize("shape", ["er", "scale_free", "contact"])
def test_large_streams_equal_the_generic_kernels(pp, shape):
    gen = torch.Generator(device=DEV).manual_seed(9)
    if shape == "er":               # the headline generator at a tenth of its size
        n, m, span, delta, K = 50_000, 1_000_000, 1_000_000, 100_000, 5
        ei = torch.randint(0, n, (2, m), generator=gen, device=DEV)
    elif shape == "scale_free":     # BASELINE configs[2]'s generator at a tenth of its size: a node with a tenth of all in-events
        n, m, span, delta, K = 100_000, 2_000_000, 1_000_000, 150_000, 3
        src = torch.randint(0, n, (m,), generator=gen, device=DEV)
        dst = (n * torch.rand(m, generator=gen, device=DEV, dtype=torch.float64).pow(6.0)).long().clamp_(max=n - 1)
        ei = torch.stack((src, dst))
    else:                           # contact network: 96 nodes, 2 * 10^5 events
        n, m, span, delta, K = 96, 200_000, 200_000, 60, 3
        ei = torch.randint(0, n, (2, m), generator=gen, device=DEV)
    t = torch.randint(0, span, (m,), generator=gen, device=DEV)
    w = torch.randint(1, 5, (m,), generator=gen, device=DEV).float()
    for weighted in (False, True):
        data = pp.Data(edge_index=ei, time=t, num_nodes=n)
        if weighted:
            data["edge_weight"] = w
        g = pp.TemporalGraph(data)
        fast = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K)
        assert _level_by_level(fast)
        slow = _generic(pp, g, delta, K)
        assert not _level_by_level(slow)
        for k in range(1, K + 1):
            a, b = fast.layers[k].data, slow.layers[k].data
            assert a.num_nodes == b.num_nodes
            for key in ("edge_index", "edge_weight", "node_sequence"):
                assert torch.equal(a[key], b[key]), (shape, weighted, k, key)
        assert torch.equal(fast.layers[2].data.inverse_idx, slow.layers[2].data.inverse_idx)
        assert torch.equal(fast.layers[3].data.inverse_idx, slow.layers[3].data.inverse_idx)      # (made by the generic kernels on demand)


def test_dbgnn_bundle_of_a_higher_layer(pp):
    # to_dbgnn_data(max_order=3) on layers that are CSR views: the bundle's tensors are the oracle's
    from oracle import model as om
    ei, t, w, n, delta = _stream("hubs", 2)
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=3)
    assert _level_by_level(model)
    sei, st, _ = om.stable_time_sort(ei, t)
    want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=3)
    x = torch.randn(n, 4)
    x_h = torch.randn(want[3]["num_nodes"], 4)
    ref = om.dbgnn_inputs(want, 3, "last", x=x, x_h=x_h)
    got = model.to_dbgnn_data(max_order=3, mapping="last", x=x.to(DEV), x_h=x_h.to(DEV))
    for key in ("edge_index", "edge_index_higher_order", "edge_weights", "edge_weights_higher_order", "bipartite_edge_index"):
        assert torch.equal(got[key].cpu(), ref[key]), key


def fuzz_against_generic(pp, cases, seed, max_events=60_000, verbose=False):
    """Streams of random shape (sparse / few nodes / Zipf sources or targets / timestamp ties / float64 time / weights / cached or not), K = 3..5:
    every tensor of every layer equal to the generic kernels'.  Returns (builds the level-by-level builder took, builds it handed back)."""
    from pathpyg_amd.core import multi_order_model as mm
    rng = np.random.default_rng(seed)
    taken = back = 0
    for case in range(cases):
        shape = rng.integers(0, 5)
        m = int(rng.integers(1, max_events))
        if shape == 0:        # sparse
            n = int(rng.integers(max(m // 40, 1), max(m // 4, 2)))
        elif shape == 1:      # dense, few nodes
            n = int(rng.integers(1, 40))
            m = min(m, 8_000)
        else:
            n = int(rng.integers(1, 3_000))
        span = int(rng.integers(1, 4 * m + 2))
        K = int(rng.integers(3, 6))
        src = rng.integers(0, n, m)
        dst = (n * rng.random(m) ** rng.choice([1.0, 3.0, 6.0])).astype(np.int64) if shape >= 3 else rng.integers(0, n, m)
        if rng.integers(0, 4) == 0:
            src, dst = dst, src                                            # out-hubs instead of in-hubs
        float_time = bool(rng.integers(0, 3) == 0)
        t = np.round(rng.random(m) * span, 1) if float_time else rng.integers(0, span, m)
        per_node = max(m / n, 1e-9)                                        # a window that yields ~0.3 .. 3 continuations per event
        delta = max(span * rng.uniform(0.3, 3.0) / per_node, 0.1 if float_time else 1)
        delta = float(np.round(delta, 1)) if float_time else int(delta)
        weighted = bool(rng.integers(0, 2))
        data = pp.Data(edge_index=torch.from_numpy(np.stack((src, dst))).to(DEV), time=torch.from_numpy(t).to(DEV), num_nodes=n)
        if weighted:
            data["edge_weight"] = torch.from_numpy(rng.integers(1, 5, m).astype(np.float32)).to(DEV)
        cached = bool(rng.integers(0, 4))
        g = pp.TemporalGraph(data)
        mm.FUSED_BUILDER = False
        try:
            slow = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K, cached=cached)
        except RuntimeError as err:          # (a dense draw whose instance graph passes 2^31 edges at some order: beyond the generic kernels)
            if "2^31" not in str(err):
                raise
            continue
        finally:
            mm.FUSED_BUILDER = True
        fast = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K, cached=cached)
        lbl = _level_by_level(fast)
        taken += lbl
        back += not lbl
        what = f"case {case}: m={m} n={n} span={span} delta={delta} K={K} weighted={weighted} float_time={float_time} shape={shape} level-by-level={lbl}"
        assert sorted(fast.layers) == sorted(slow.layers), what
        for k in fast.layers:
            a, b = fast.layers[k].data, slow.layers[k].data
            assert a.num_nodes == b.num_nodes, (what, k)
            for key in ("edge_index", "edge_weight", "node_sequence") + (("inverse_idx",) if case % 8 == 0 else ()):
                assert torch.equal(a[key], b[key]), (what, k, key)
        if verbose and case % 25 == 0:
            print(what, [(l.n, l.m) for l in fast.layers.values()], flush=True)
    return taken, back


def test_fuzz_against_the_generic_kernels(pp):
    taken, back = fuzz_against_generic(pp, 120, 7)
    assert taken > 90          # (a handful of dense draws exceed 4096 continuations of one node sequence and go back to the generic kernels)


@pytest.mark.parametrize("kind", ["sparse", "hubs", "ties"])
def test_fractional_event_weights_are_summed_in_the_reference_order(pp, kind):
    # merged weights = left-to-right fp32 sums over the instance edges in the reference's order (PyG coalesce on CPU): with non-integer
    # weights any other association would change low bits — the layers must still equal the oracle's bit for bit
    from oracle import model as om
    ei, t, _, n, delta = _stream(kind, 23)
    rng = np.random.default_rng(5)
    w = torch.from_numpy((rng.random(ei.size(1)) + 0.25).astype(np.float32))
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n, edge_weight=w.to(DEV)))
    sei, st, perm = om.stable_time_sort(ei, t)
    want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=4, edge_weight=w[perm])
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=4)
    assert _level_by_level(model)
    for k in want:
        assert torch.equal(model.layers[k].data.edge_index.cpu(), want[k]["edge_index"]), k
        assert torch.equal(model.layers[k].data.edge_weight.cpu(), want[k]["edge_weight"]), k


@pytest.mark.parametrize("kind", ["sparse", "hubs", "contact"])
def test_given_event_graph_takes_the_level_by_level_builder_too(pp, kind):
    # from_temporal_graph(..., event_graph=lift_order_temporal(g, delta)) (reference multi_order_model.py:124-192 with `event_graph` set): the
    # event graph's edges are the continuation windows (pp_multiorder_prepare_graph); also a SUBSET of the event graph (every other edge), which
    # no delta produces: the layers are those of the given graph
    from oracle import model as om
    ei, t, w, n, delta = _stream(kind, 17)
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n, edge_weight=w.to(DEV)))
    sei, st, perm = om.stable_time_sort(ei, t)
    eg = pp.algorithms.lift_order_temporal(g, delta)
    for graph in (eg, eg[:, ::2].contiguous()):
        model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=4, event_graph=graph)
        assert _level_by_level(model)
        want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=4, edge_weight=w[perm], event_graph=graph.cpu())
        for k in want:
            d = model.layers[k].data
            for key in ("edge_index", "edge_weight", "node_sequence", "inverse_idx"):
                assert torch.equal(d[key].cpu(), want[k][key]), (k, key)
    # an event graph on the host goes to the generic kernels
    assert not _level_by_level(pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=3, event_graph=eg.cpu()))


def test_a_node_with_more_out_events_than_a_workgroup_sorts(pp):
    # the first level sorts every node's out-list by target in LDS (up to 4096 events per list); one node with 6000 out-events in an otherwise
    # sparse stream: the list kernel reports it and the radix sort of the (source, target) keys takes the stream — same layers either way
    from oracle import model as om
    rng = np.random.default_rng(9)
    n, m = 3000, 30_000
    src = rng.integers(0, n, m)
    src[:6000] = 7
    ei = torch.from_numpy(np.stack((src, rng.integers(0, n, m))))
    t = torch.from_numpy(rng.integers(0, 200_000, m))
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
    sei, st, _ = om.stable_time_sort(ei, t)
    want = om.layers_from_temporal(sei, st, n, delta=700, max_order=3)
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=700, max_order=3)
    assert _level_by_level(model)
    for k in want:
        d = model.layers[k].data
        for key in ("edge_index", "edge_weight", "node_sequence"):
            assert torch.equal(d[key].cpu(), want[k][key]), (k, key)


def test_order_two_of_a_stream_with_a_very_large_hub(pp):
    # multi_order_model.LIFT_ONLY_ORDER2: from_temporal_graph(max_order=2) on a stream with a node of 10^5 in-events through the level-by-level builder
    # (the order-2 builder's hub kernels are the slower way to the LAYERS there; the default keeps the order-2 builder because it also makes the GCN
    # plans); the layers are the generic kernels', and the DBGNN bundle / forward / backward work from them (plans made on the way)
    from pathpyg_amd.core import multi_order_model as mm
    gen = torch.Generator(device=DEV).manual_seed(4)
    n, m, span, delta = 5_000, 400_000, 400_000, 3_000
    src = torch.randint(0, n, (m,), generator=gen, device=DEV)
    dst = (n * torch.rand(m, generator=gen, device=DEV, dtype=torch.float64).pow(6.0)).long().clamp_(max=n - 1)
    t = torch.randint(0, span, (m,), generator=gen, device=DEV)
    g = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
    assert int(torch.bincount(dst).max()) >= 65536
    assert getattr(pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=2), "_pp_fused", None) is not None          # the default
    mm.LIFT_ONLY_ORDER2 = True
    try:
        model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=2)
    finally:
        mm.LIFT_ONLY_ORDER2 = False
    assert _level_by_level(model) and getattr(model, "_pp_fused", None) is None
    slow = _generic(pp, g, delta, 2)
    for k in (1, 2):
        a, b = model.layers[k].data, slow.layers[k].data
        for key in ("edge_index", "edge_weight", "node_sequence", "inverse_idx"):
            assert torch.equal(a[key], b[key]), (k, key)
    data = model.to_dbgnn_data(max_order=2, x=torch.randn(n, 16, device=DEV), x_h=torch.randn(model.layers[2].n, 16, device=DEV))
    net = pp.nn.DBGNN(num_classes=3, num_features=(16, 16), hidden_dims=[16, 16, 16]).to(DEV)
    out = net(data)
    out.sum().backward()
    assert bool(torch.isfinite(out).all())
    # a stream without such a hub stays on the order-2 builder even then
    ei, tt, _, nn_, dd = _stream("sparse", 3)
    mm.LIFT_ONLY_ORDER2 = True
    try:
        small = pp.MultiOrderModel.from_temporal_graph(pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=tt.to(DEV), num_nodes=nn_)), delta=dd, max_order=2)
    finally:
        mm.LIFT_ONLY_ORDER2 = False
    assert getattr(small, "_pp_fused", None) is not None
