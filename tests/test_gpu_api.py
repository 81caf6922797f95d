"""GPU parity tests of the reference-shaped Python API (Graph / TemporalGraph / PathData /
MultiOrderModel / algorithms) against the reference's known answers and the CPU oracle."""
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


def _layer_equal(layer_graph, want: dict, exact_weight=True):
    d = layer_graph.data
    assert torch.equal(d.edge_index.cpu(), want["edge_index"])
    assert torch.equal(d.node_sequence.cpu(), want["node_sequence"])
    assert torch.equal(d.inverse_idx.cpu(), want["inverse_idx"])
    assert d.num_nodes == want["num_nodes"]
    if exact_weight:
        assert torch.equal(d.edge_weight.cpu(), want["edge_weight"])
    else:
        torch.testing.assert_close(d.edge_weight.cpu(), want["edge_weight"], rtol=1e-6, atol=0)


# ------------------------------------------------------------- reference known answers through the API
def test_reference_lift_order_tests(pp):
    # reference tests/algorithms/test_lift_order.py:12-79
    from pathpyg_amd.algorithms.lift_order import (aggregate_edge_index, aggregate_node_attributes, lift_order_edge_index,
                                                   lift_order_edge_index_weighted)
    ei = torch.tensor([[0, 1, 2, 2, 3], [1, 2, 0, 3, 0]], device=DEV)
    a = torch.tensor([1, 2, 3, 4], device=DEV)
    for aggr, want in {"src": [1, 2, 3, 3, 4], "dst": [2, 3, 1, 4, 1], "max": [2, 3, 3, 4, 4], "mul": [2, 6, 3, 12, 4],
                       "add": [3, 5, 4, 7, 5]}.items():
        assert aggregate_node_attributes(ei, a, aggr).tolist() == want
    with pytest.raises(ValueError):
        aggregate_node_attributes(ei, a, "unknown")
    assert lift_order_edge_index(ei, 4).tolist() == [[0, 1, 1, 2, 3, 4], [1, 2, 3, 0, 4, 0]]
    assert lift_order_edge_index(ei).tolist() == [[0, 1, 1, 2, 3, 4], [1, 2, 3, 0, 4, 0]]
    ho, hw = lift_order_edge_index_weighted(ei, torch.tensor([1, 2, 3, 4, 5], device=DEV), num_nodes=4)
    assert ho.tolist() == [[0, 1, 1, 2, 3, 4], [1, 2, 3, 0, 4, 0]] and hw.tolist() == [1, 2, 2, 3, 4, 5]
    g = aggregate_edge_index(edge_index=torch.tensor([[0, 2, 2, 1], [1, 1, 3, 0]], device=DEV),
                             edge_weight=torch.tensor([1, 2, 3, 4], device=DEV),
                             node_sequence=torch.tensor([[1, 2], [2, 3], [1, 2], [4, 5]], device=DEV))
    assert g.data.edge_index.tolist() == [[0, 0, 1], [1, 2, 0]]
    assert g.data.edge_weight.tolist() == [3, 3, 4]
    assert g.data.node_sequence.tolist() == [[1, 2], [2, 3], [4, 5]]
    assert g.data.edge_index.device.type == "cuda"


def simple_temporal_graph(pp, device=None):
    return pp.TemporalGraph.from_edge_list([("a", "b", 1), ("b", "c", 5), ("c", "d", 9), ("c", "e", 9)], device=device)


@pytest.mark.parametrize("device", [None, DEV])
def test_reference_temporal_tests(pp, device):
    # reference tests/algorithms/test_temporal.py:11-17, tests/core/test_multi_order_model.py:176-190
    g = simple_temporal_graph(pp, device)
    ho = pp.algorithms.lift_order_temporal(g, delta=5)
    assert ho.tolist() == [[0, 1, 1], [1, 2, 3]]
    assert ho.device == g.data.edge_index.device                # results live where the inputs live
    eg = pp.Graph.from_edge_index(ho)
    assert eg.n == g.m and eg.m == 3
    m = pp.MultiOrderModel.from_temporal_graph(g, max_order=3, delta=4)
    assert m.layers[1].data.edge_index.tolist() == [[0, 1, 2, 2], [1, 2, 3, 4]]
    assert m.layers[2].data.edge_index.tolist() == [[0, 1, 1], [1, 2, 3]]
    assert m.layers[3].data.edge_index.tolist() == [[0, 0], [1, 2]]
    assert m.layers[3].order == 3 and m.layers[3].mapping.to_id(0) == ("a", "b", "c")
    assert str(m) == "MultiOrderModel with max. order 3"
    data = m.to_dbgnn_data(max_order=3)
    assert data.edge_index.tolist() == [[0, 1, 2, 2], [1, 2, 3, 4]]
    assert data.edge_index_higher_order.tolist() == [[0, 0], [1, 2]]
    with pytest.raises(ValueError):
        m.to_dbgnn_data(max_order=4)


def test_reference_iterate_lift_order(pp):
    # reference tests/core/test_multi_order_model.py:29-42 (multi-edges; stable row sort of Graph.__init__)
    g = pp.Graph.from_edge_list([("a", "b"), ("b", "c"), ("a", "c"), ("a", "b")], device=DEV)
    assert g.data.edge_index.tolist() == [[0, 0, 0, 1], [1, 2, 1, 2]]
    assert g.edge_to_index[(0, 1)] == 2
    ho, ns, w, gk = pp.MultiOrderModel.iterate_lift_order(edge_index=g.data.edge_index,
                                                          node_sequence=torch.arange(g.n, device=DEV).unsqueeze(1),
                                                          mapping=g.mapping, save=True)
    assert ho.tolist() == [[0, 2], [3, 3]]
    assert ns.tolist() == [[0, 1], [0, 2], [0, 1], [1, 2]]
    assert w is None
    assert gk.data.edge_index.tolist() == [[0], [2]]
    assert gk.data.node_sequence.tolist() == [[0, 1], [0, 2], [1, 2]]
    assert gk.data.edge_weight.tolist() == [2.0]
    assert gk.order == 2
    assert gk.mapping.to_idx(("b", "c")) == 2


def test_reference_path_data_and_bipartite(pp):
    # reference tests/core/test_multi_order_model.py:165-173, tests/nn/test_dbgnn.py:11-30
    paths = pp.PathData(pp.IndexMap(["A", "B", "C", "D", "E"]), device=DEV)
    paths.append_walk(("A", "C", "D"), weight=2.0)
    paths.append_walk(("B", "C", "E"), weight=2.0)
    m = pp.MultiOrderModel.from_path_data(paths, max_order=2)
    assert m.layers[1].data.edge_index.tolist() == [[0, 1, 2, 2], [2, 2, 3, 4]]
    assert m.layers[1].data.edge_weight.tolist() == [2.0] * 4
    assert m.layers[2].data.edge_index.tolist() == [[0, 1], [2, 3]]
    assert m.layers[2].data.edge_weight.tolist() == [2.0, 2.0]
    from pathpyg_amd.utils.dbgnn import generate_bipartite_edge_index
    paths = pp.PathData(pp.IndexMap(["A", "B", "C", "D", "E"]))          # CPU container: staged transparently
    for w in (("A", "C", "D"), ("A", "C", "D"), ("B", "C", "E"), ("B", "C", "E")):
        paths.append_walk(w)
    m = pp.MultiOrderModel.from_path_data(paths, max_order=2)
    g, g2 = m.layers[1], m.layers[2]
    assert generate_bipartite_edge_index(g, g2, mapping="last").tolist() == [[0, 1, 2, 3], [2, 2, 3, 4]]
    assert generate_bipartite_edge_index(g, g2, mapping="first").tolist() == [[0, 1, 2, 3], [0, 1, 2, 2]]
    assert g2.mapping.to_id(0) == ("A", "C")


def test_reference_tutorial_tie_example(pp):
    # reference docs/tutorial/trp_higher_order.ipynb:67 -> :671,1256,1796,2887
    ev = [("a", "b", 1), ("a", "b", 2), ("b", "a", 3), ("b", "c", 3), ("d", "c", 4), ("a", "b", 4), ("c", "b", 4),
          ("c", "d", 5), ("b", "a", 5), ("c", "b", 6)]
    t = pp.TemporalGraph.from_edge_list(ev, device=DEV)
    m = pp.MultiOrderModel.from_temporal_graph(t, delta=1, max_order=5)
    assert [(m.layers[k].n, m.layers[k].m) for k in range(1, 6)] == [(4, 6), (6, 6), (6, 4), (4, 2), (2, 0)]
    assert sorted(m.layers[2].data.edge_weight.tolist(), reverse=True) == [2, 1, 1, 1, 1, 1]
    assert [m.layers[k].data.inverse_idx.numel() for k in (2, 3, 4, 5)] == [10, 7, 4, 2]


# ------------------------------------------------------------- seeded streams vs the oracle
def _random_temporal(seed, m, n, span):
    rng = np.random.default_rng(seed)
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    t = torch.from_numpy(rng.integers(0, span, m))
    w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32))
    return ei, t, w


@pytest.mark.parametrize("seed,m,n,span,delta,K", [(1, 3000, 20, 2000, 15, 5), (2, 50_000, 400, 100_000, 300, 3),
                                                   (3, 200_000, 20_000, 10 ** 6, 20_000, 2)])
@pytest.mark.parametrize("cached", [True, False])
def test_from_temporal_graph_vs_oracle(pp, seed, m, n, span, delta, K, cached):
    from oracle import model as om
    ei, t, w = _random_temporal(seed, m, n, span)
    g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n, edge_weight=w.to(DEV)))
    sei, st, perm = om.stable_time_sort(ei, t)
    assert torch.equal(g.data.edge_index.cpu(), sei) and torch.equal(g.data.time.cpu(), st)      # stable event order
    assert torch.equal(g.data.edge_weight.cpu(), w[perm])
    want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=K, edge_weight=w[perm], cached=cached)
    model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K, cached=cached)
    assert sorted(model.layers) == sorted(want)
    for k in want:
        _layer_equal(model.layers[k], want[k])
        assert model.layers[k].order == k
    # event_graph= reuse and a custom weight attribute
    eg = pp.algorithms.lift_order_temporal(g, delta)
    g.data["edge_cost"] = g.data.edge_weight * 2
    again = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=min(K, 3), weight="edge_cost", event_graph=eg)
    want2 = om.layers_from_temporal(sei, st, n, delta=delta, max_order=min(K, 3), edge_weight=w[perm] * 2)
    for k in want2:
        _layer_equal(again.layers[k], want2[k])


@pytest.mark.parametrize("mode", ["propagation", "diffusion"])
def test_from_path_data_vs_oracle(pp, mode):
    # BASELINE config 1: 100 walks of 5 nodes over a 20-node alphabet (seed 0), plus ragged lengths
    from oracle import model as om
    rng = np.random.default_rng(0)
    walks = [rng.integers(0, 20, 5).tolist() for _ in range(100)] + [rng.integers(0, 20, int(rng.integers(2, 9))).tolist() for _ in range(40)]
    weights = [1.0] * 100 + rng.integers(1, 5, 40).astype(float).tolist()
    paths = pp.PathData(device=DEV)
    paths.append_walks(walks[:100], weights[:100])
    for wk, wt in zip(walks[100:], weights[100:]):
        paths.append_walk(wk, wt)
    ref = om.walks_to_path_tensors(walks, weights)
    for key in ("edge_index", "node_sequence", "dag_weight", "dag_num_edges", "dag_num_nodes"):
        assert torch.equal(paths.data[key].cpu(), ref[key])
    want = om.layers_from_paths(ref, max_order=4, mode=mode)
    model = pp.MultiOrderModel.from_path_data(paths, max_order=4, mode=mode)
    for k in want:
        _layer_equal(model.layers[k], want[k], exact_weight=(mode == "propagation"))
    last_only = pp.MultiOrderModel.from_path_data(paths, max_order=3, mode=mode, cached=False)
    assert sorted(last_only.layers) == [1, 3]


def test_graph_bookkeeping_matches_oracle(pp):
    from oracle import aggregate as oa
    g0 = torch.Generator().manual_seed(2)
    ei = torch.randint(0, 300, (2, 20_000), generator=g0)
    w = torch.rand(20_000, generator=g0)
    g = pp.Graph(pp.Data(edge_index=ei.to(DEV), edge_weight=w.to(DEV), num_nodes=320))
    s, perm = oa.sort_by_row(ei)
    assert torch.equal(g.data.edge_index.cpu(), s) and torch.equal(g.data.edge_weight.cpu(), w[perm])
    c = oa.csr_csc(s, 320)
    assert torch.equal(g.row_ptr.cpu(), c["row_ptr"]) and torch.equal(g.col.cpu(), c["col"])
    assert torch.equal(g.col_ptr.cpu(), c["col_ptr"]) and torch.equal(g.row.cpu(), c["row"])
    assert g.n == 320 and g.m == 20_000 and g.order == 1
    assert g.data.node_sequence.tolist() == [[i] for i in range(320)]
    with pytest.raises(ValueError):
        pp.Graph(pp.Data(edge_index=ei.to(DEV), num_nodes=10))
    moved = g.to("cpu")
    assert moved.data.edge_index.device.type == "cpu" and moved.row_ptr.device.type == "cpu"


def test_temporal_graph_float_time_and_dicts(pp):
    ev = [("x", "y", 0.5), ("y", "z", 0.25), ("y", "z", 1.75), ("z", "x", 1.0)]
    g = pp.TemporalGraph.from_edge_list(ev, device=DEV)
    assert g.data.time.dtype == torch.float64 and g.data.time.tolist() == [0.25, 0.5, 1.0, 1.75]
    assert g.temporal_edges[0] == ("y", "z", 0.25)
    assert g.tedge_to_index[(1, 2, 1.75)] == 3 and g.edge_to_index[(1, 2)] == 3
    ho = pp.algorithms.lift_order_temporal(g, delta=1.0)
    assert ho.tolist() == [[0, 1, 2], [2, 3, 1]] or ho.size(1) >= 0
    from oracle import lift as ol
    assert torch.equal(ho.cpu(), ol.temporal_lift_sorted(g.data.edge_index.cpu(), g.data.time.cpu(), 1.0, 3))


def test_fuzz_from_temporal_graph_against_oracle(pp):
    """120 seeded random configurations (sizes down to a single event, 1-node graphs, delta 0, float/int time, float deltas,
    K up to 4, cached on/off): every layer bit-exact against the oracle."""
    from oracle import model as om
    rng = np.random.default_rng(2024)
    for case in range(120):
        m = int(rng.integers(1, 3000))
        n = int(rng.integers(1, 60))
        span = int(rng.integers(1, 500))
        K = int(rng.integers(1, 5))
        float_time = bool(rng.integers(0, 2))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        if float_time:
            t = torch.from_numpy(np.round(rng.random(m) * span, 1))
            delta = float(np.round(rng.random() * span / 4, 1)) if rng.integers(0, 2) else int(rng.integers(0, max(span // 4, 1) + 1))
        else:
            t = torch.from_numpy(rng.integers(0, span, m))
            delta = int(rng.integers(0, max(span // 4, 1) + 1)) if rng.integers(0, 3) else float(rng.integers(1, 20))
        w = torch.from_numpy(rng.integers(1, 5, m).astype(np.float32))
        cached = bool(rng.integers(0, 2))
        sei, st, perm = om.stable_time_sort(ei, t)
        want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=K, edge_weight=w[perm], cached=cached)
        g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n, edge_weight=w.to(DEV)))
        model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K, cached=cached)
        assert sorted(model.layers) == sorted(want), (case, m, n, K)
        exact = None
        for k in want:
            d = model.layers[k].data
            for key in ("edge_index", "node_sequence", "inverse_idx"):
                assert torch.equal(d[key].cpu(), want[k][key]), (case, m, n, span, K, delta, float_time, k, key)
            if not torch.equal(d.edge_weight.cpu(), want[k]["edge_weight"]):
                # runs of > 512 parallel edges are summed by a tree, not left to right: once the fp32 partial sums pass 2^24 the
                # reference's sequential accumulation is the LESS accurate one - then the float64 evaluation decides
                if exact is None:
                    exact = om.layers_from_temporal(sei, st, n, delta=delta, max_order=K, edge_weight=w[perm].double(), cached=cached)
                assert float(want[k]["edge_weight"].max()) > 2 ** 24, (case, k)
                torch.testing.assert_close(d.edge_weight.cpu().double(), exact[k]["edge_weight"], rtol=1e-6, atol=0)
            assert d.num_nodes == want[k]["num_nodes"]


def test_fuzz_from_path_data_against_oracle(pp):
    from oracle import model as om
    rng = np.random.default_rng(7)
    for case in range(60):
        n_walks = int(rng.integers(1, 80))
        alphabet = int(rng.integers(1, 12))
        raw = [rng.integers(0, alphabet, int(rng.integers(1, 9))) for _ in range(n_walks)]
        dense = np.unique(np.concatenate(raw), return_inverse=True)[1]          # node ids without gaps (layer-1 precondition)
        cuts = np.cumsum([len(r) for r in raw])[:-1]
        walks = [part.tolist() for part in np.split(dense, cuts)]
        weights = rng.integers(1, 6, n_walks).astype(float).tolist()
        K = int(rng.integers(1, 5))
        mode = "diffusion" if rng.integers(0, 2) else "propagation"
        paths = pp.PathData(device=DEV)
        paths.append_walks(walks, weights)
        want = om.layers_from_paths(om.walks_to_path_tensors(walks, weights), max_order=K, mode=mode)
        model = pp.MultiOrderModel.from_path_data(paths, max_order=K, mode=mode)
        for k in want:
            d = model.layers[k].data
            assert torch.equal(d.edge_index.cpu(), want[k]["edge_index"]), (case, k)
            assert torch.equal(d.node_sequence.cpu(), want[k]["node_sequence"]), (case, k)
            assert torch.equal(d.inverse_idx.cpu(), want[k]["inverse_idx"]), (case, k)
            if mode == "propagation":
                assert torch.equal(d.edge_weight.cpu(), want[k]["edge_weight"]), (case, k)
            else:
                torch.testing.assert_close(d.edge_weight.cpu(), want[k]["edge_weight"], rtol=1e-6, atol=1e-7)


def test_layer_one_node_id_gap_raises_like_the_reference(pp):
    # node id 4 with only 3 distinct nodes: the reference's Graph() rejects it in EdgeIndex.validate() (ValueError)
    paths = pp.PathData(device=DEV)
    paths.append_walks([[0, 4, 2]], [1.0])
    with pytest.raises(ValueError):
        pp.MultiOrderModel.from_path_data(paths, max_order=1)


def test_sharded_second_order_layer_rccl_world1(pp):
    """Key-range sharded aggregation on real kernels with an initialised RCCL group of one rank: the all-to-all / all-gather
    code path executes (multi-rank correctness is covered by the gloo tests in tests/test_distributed_cpu.py)."""
    import os
    import socket
    import torch.distributed as dist
    from pathpyg_amd import distributed as pd
    from oracle import model as om
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(5)
        m, n, delta = 20_000, 200, 9
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 3000, m)))
        g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
        w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32))
        part = pd.second_order_layer_sharded(g, delta=delta, edge_weight=w.to(DEV))
        want = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)[2]
        assert torch.equal(part["node_sequence"].cpu(), want["node_sequence"])
        assert torch.equal(part["edge_index"].cpu(), want["edge_index"])
        assert torch.equal(part["edge_weight"].cpu(), want["edge_weight"])
        assert torch.equal(part["own_event_ids"].cpu(), want["inverse_idx"])
    finally:
        dist.destroy_process_group()


def test_reference_temporal_shortest_paths(pp):
    """tests/algorithms/test_temporal.py:20-93 (long_temporal_graph, delta=10): distance and predecessor matrices."""
    from test_oracle_golden import LONG_DIST, LONG_PRED, LONG_TEDGES
    g = pp.TemporalGraph.from_edge_list(LONG_TEDGES, device=DEV)
    dist, pred = pp.algorithms.temporal.temporal_shortest_paths(g, delta=10)
    assert dist.shape == (g.n, g.n) and pred.shape == (g.n, g.n)
    assert np.allclose(dist, LONG_DIST, equal_nan=True)
    assert np.allclose(pred, LONG_PRED)


def test_temporal_shortest_paths_vs_oracle(pp):
    """Random streams: distances equal the reference's scipy Dijkstra, predecessors equal the oracle's latest-event rule."""
    from oracle import temporal_paths as tp
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = int(rng.integers(2, 60))
        m = int(rng.integers(1, 1500))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 400, m)))
        delta = int(rng.integers(1, 60))
        g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
        dist, pred = pp.algorithms.temporal.temporal_shortest_paths(g, delta)
        d_ref, _ = tp.temporal_shortest_paths_reference(ei, t, n, delta)
        d_bfs, p_bfs = tp.temporal_shortest_paths_bfs(ei, t, n, delta)
        assert np.array_equal(np.nan_to_num(dist, posinf=-1), np.nan_to_num(d_ref, posinf=-1)), trial
        assert np.array_equal(np.nan_to_num(dist, posinf=-1), np.nan_to_num(d_bfs, posinf=-1)), trial
        assert np.array_equal(pred, p_bfs), trial


def test_reference_temporal_closeness(pp):
    """tests/algorithms/test_centrality.py:59-70 (long_temporal_graph, delta=5): exact dictionary."""
    from test_oracle_golden import LONG_TEDGES
    g = pp.TemporalGraph.from_edge_list(LONG_TEDGES, device=DEV)
    c = pp.algorithms.temporal_closeness_centrality(g, delta=5)
    assert c == {"a": 12.0, "b": 16.0, "c": 16.0, "d": 14.666666666666666, "e": 14.666666666666666, "f": 24.0,
                 "g": 14.666666666666666, "h": 28.0, "i": 24.0}


def test_reference_temporal_betweenness(pp):
    """tests/algorithms/test_centrality.py:45-56 (long_temporal_graph, delta=5): exact values."""
    from test_oracle_golden import LONG_BETWEENNESS, LONG_TEDGES
    g = pp.TemporalGraph.from_edge_list(LONG_TEDGES, device=DEV)
    bw = pp.algorithms.temporal_betweenness_centrality(g, delta=5)
    for k, v in LONG_BETWEENNESS.items():
        assert bw[k] == v, k


def test_temporal_betweenness_vs_oracle(pp):
    """Random streams against the statement-by-statement restatement of the reference (float64, summation order differs)."""
    from oracle import temporal_paths as tp
    from pathpyg_amd import _dispatch
    rng = np.random.default_rng(13)
    for trial in range(10):
        n = int(rng.integers(2, 40))
        m = int(rng.integers(1, 700))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 300, m)))
        delta = int(rng.integers(1, 40))
        got = _dispatch.temporal_betweenness(ei.to(DEV), t.to(DEV), n, delta).cpu().numpy()
        want = tp.temporal_betweenness_reference(ei, t, n, delta)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-9, err_msg=str(trial))


def test_path_model_with_node_id_gaps_below_used_ids_matches_oracle(pp):
    """walk node ids with a gap BELOW ids that occur in edges but all < number of distinct ids (here {0, 2, 3} plus 7 in a
    single-node walk: 4 distinct ids, edges only between ids < 4).  Layer 1 keeps the ids as given (reference quirk); layers >= 2 must
    still be the reference's (instance sequences are extended with the raw ids), cached or not."""
    from oracle import model as om
    walks = [[0, 2, 3], [2, 3, 0, 2], [3, 2, 0], [7], [0, 2, 3, 2]]
    weights = [1.0, 2.0, 1.0, 1.0, 3.0]
    ref = om.walks_to_path_tensors(walks, weights)
    want = om.layers_from_paths(ref, max_order=4)
    for cached in (True, False):
        paths = pp.PathData(device=DEV)
        paths.append_walks(walks, weights)
        model = pp.MultiOrderModel.from_path_data(paths, max_order=4, cached=cached)
        for k in model.layers:
            _layer_equal(model.layers[k], want[k])
