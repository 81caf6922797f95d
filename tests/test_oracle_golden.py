"""The oracle against (1) vectors produced by the reference's own source and (2) every
known-answer value of the reference's unit tests for the hot path."""
import numpy as np
import pytest
import torch

from conftest import golden_delta
from oracle import aggregate as oa
from oracle import lift as ol
from oracle import model as om

TEMPORAL = ["int_ties", "int_unique", "int_wide", "int_delta0", "int_fdelta", "int_f64delta",
            "f64_ties", "f64_npdelta", "f64_intdelta", "int_big"]
LINEGRAPH = ["small", "multi", "isolated", "hub", "wide"]


@pytest.mark.parametrize("name", TEMPORAL)
def test_temporal_lift_matches_reference_source(golden, name):
    ei = torch.from_numpy(golden[f"temporal/{name}/edge_index"])
    t = torch.from_numpy(golden[f"temporal/{name}/time"])
    delta = golden_delta(golden, name)
    want = torch.from_numpy(golden[f"temporal/{name}/out"])
    fast = ol.temporal_lift_sorted(ei, t, delta, int(golden[f"temporal/{name}/num_nodes"]))
    assert torch.equal(fast, want)
    if str(golden[f"temporal/{name}/raised"]):
        with pytest.raises((RuntimeError, ValueError)):
            ol.temporal_lift_per_timestamp(ei, t, delta)
    else:
        assert torch.equal(ol.temporal_lift_per_timestamp(ei, t, delta), want)


@pytest.mark.parametrize("name", LINEGRAPH)
def test_line_graph_lift_matches_reference_source(golden, name):
    ei = torch.from_numpy(golden[f"linegraph/{name}/edge_index"])
    n = int(golden[f"linegraph/{name}/num_nodes"])
    w = torch.from_numpy(golden[f"linegraph/{name}/edge_weight"])
    want = torch.from_numpy(golden[f"linegraph/{name}/out"])
    assert torch.equal(ol.line_graph_lift(ei, n), want)
    for aggr in ("src", "dst", "max", "mul", "add"):
        ho, hw = ol.line_graph_lift_weighted(ei, w, n, aggr)
        assert torch.equal(ho, want)
        assert torch.equal(hw, torch.from_numpy(golden[f"linegraph/{name}/w_{aggr}"]))


def test_line_graph_lift_infers_num_nodes(golden):
    ei = torch.from_numpy(golden["linegraph/infer/edge_index"])
    assert torch.equal(ol.line_graph_lift(ei), torch.from_numpy(golden["linegraph/infer/out"]))


def test_chained_lifts_match_reference_source(golden):
    ei = torch.from_numpy(golden["chain/edge_index"])
    t = torch.from_numpy(golden["chain/time"])
    ho = ol.temporal_lift_sorted(ei, t, int(golden["chain/delta"]), int(golden["chain/num_nodes"]))
    assert torch.equal(ho, torch.from_numpy(golden["chain/k2"]))
    n_inst = ei.size(1)
    for k in (3, 4, 5):
        nxt = ol.line_graph_lift(ho, n_inst)
        n_inst, ho = ho.size(1), nxt
        assert torch.equal(ho, torch.from_numpy(golden[f"chain/k{k}"]))


# ---- known answers of the reference's own tests -------------------------------------------------

def test_known_answer_node_attributes():
    # reference tests/algorithms/test_lift_order.py:12-31
    ei = torch.tensor([[0, 1, 2, 2, 3], [1, 2, 0, 3, 0]])
    a = torch.tensor([1, 2, 3, 4])
    want = {"src": [1, 2, 3, 3, 4], "dst": [2, 3, 1, 4, 1], "max": [2, 3, 3, 4, 4],
            "mul": [2, 6, 3, 12, 4], "add": [3, 5, 4, 7, 5]}
    for aggr, vals in want.items():
        assert ol.edge_attribute_from_nodes(ei, a, aggr).tolist() == vals
    with pytest.raises(ValueError):
        ol.edge_attribute_from_nodes(ei, a, "unknown")


def test_known_answer_line_graph():
    # reference tests/algorithms/test_lift_order.py:34-57
    ei = torch.tensor([[0, 1, 2, 2, 3], [1, 2, 0, 3, 0]])
    assert ol.line_graph_lift(ei, 4).tolist() == [[0, 1, 1, 2, 3, 4], [1, 2, 3, 0, 4, 0]]
    ho, hw = ol.line_graph_lift_weighted(ei, torch.tensor([1, 2, 3, 4, 5]), 4)
    assert ho.tolist() == [[0, 1, 1, 2, 3, 4], [1, 2, 3, 0, 4, 0]]
    assert hw.tolist() == [1, 2, 2, 3, 4, 5]


def test_known_answer_aggregate_edge_index():
    # reference tests/algorithms/test_lift_order.py:60-79
    g = oa.aggregate_edge_index(torch.tensor([[0, 2, 2, 1], [1, 1, 3, 0]]),
                                torch.tensor([[1, 2], [2, 3], [1, 2], [4, 5]]),
                                torch.tensor([1, 2, 3, 4]))
    assert g["edge_index"].tolist() == [[0, 0, 1], [1, 2, 0]]
    assert g["edge_weight"].tolist() == [3, 3, 4]
    assert g["node_sequence"].tolist() == [[1, 2], [2, 3], [4, 5]]


SIMPLE_TEMPORAL = (torch.tensor([[0, 1, 2, 2], [1, 2, 3, 4]]), torch.tensor([1, 5, 9, 9]), 5)


def test_known_answer_temporal_lift():
    # reference tests/algorithms/test_temporal.py:11-17 ((a,b,1),(b,c,5),(c,d,9),(c,e,9), delta=5)
    ei, t, n = SIMPLE_TEMPORAL
    assert ol.temporal_lift_per_timestamp(ei, t, 5).tolist() == [[0, 1, 1], [1, 2, 3]]
    assert ol.temporal_lift_sorted(ei, t, 5, n).tolist() == [[0, 1, 1], [1, 2, 3]]


def test_known_answer_iterate_lift_order():
    # reference tests/core/test_multi_order_model.py:29-42; Graph.from_edge_list of
    # (a,b),(b,c),(a,c),(a,b) is stably row-sorted to [[0,0,0,1],[1,2,1,2]]
    ei, _ = oa.sort_by_row(torch.tensor([[0, 1, 0, 0], [1, 2, 2, 1]]))
    assert ei.tolist() == [[0, 0, 0, 1], [1, 2, 1, 2]]
    ho, ns, w, gk = om.lift_step(ei, torch.arange(3).unsqueeze(1), None, "src", True)
    assert ho.tolist() == [[0, 2], [3, 3]]
    assert ns.tolist() == [[0, 1], [0, 2], [0, 1], [1, 2]]
    assert w is None
    assert gk["edge_index"].tolist() == [[0], [2]]
    assert gk["node_sequence"].tolist() == [[0, 1], [0, 2], [1, 2]]
    assert gk["edge_weight"].tolist() == [2.0]


def test_known_answer_from_path_data():
    # reference tests/core/test_multi_order_model.py:165-173 (walks A-C-D and B-C-E, weight 2 each)
    paths = om.walks_to_path_tensors([[0, 2, 3], [1, 2, 4]], [2.0, 2.0])
    layers = om.layers_from_paths(paths, max_order=2)
    assert layers[1]["edge_index"].tolist() == [[0, 1, 2, 2], [2, 2, 3, 4]]
    assert layers[1]["edge_weight"].tolist() == [2.0, 2.0, 2.0, 2.0]
    assert layers[2]["edge_index"].tolist() == [[0, 1], [2, 3]]
    assert layers[2]["edge_weight"].tolist() == [2.0, 2.0]


def test_known_answer_from_temporal_graph_and_dbgnn_data():
    # reference tests/core/test_multi_order_model.py:176-190
    ei, t, n = SIMPLE_TEMPORAL
    for loop in (True, False):
        layers = om.layers_from_temporal(ei, t, n, delta=4, max_order=3, loop_lift=loop)
        assert layers[1]["edge_index"].tolist() == [[0, 1, 2, 2], [1, 2, 3, 4]]
        assert layers[2]["edge_index"].tolist() == [[0, 1, 1], [1, 2, 3]]
        assert layers[3]["edge_index"].tolist() == [[0, 0], [1, 2]]
        assert layers[3]["node_sequence"].tolist() == [[0, 1, 2], [1, 2, 3], [1, 2, 4]]
    data = om.dbgnn_inputs(layers, max_order=3)
    assert data["edge_index"].tolist() == [[0, 1, 2, 2], [1, 2, 3, 4]]
    assert data["edge_index_higher_order"].tolist() == [[0, 0], [1, 2]]
    with pytest.raises(ValueError):
        om.dbgnn_inputs(layers, max_order=4)


def test_known_answer_bipartite_index():
    # reference tests/nn/test_dbgnn.py:11-30 (walks ACD, ACD, BCE, BCE)
    paths = om.walks_to_path_tensors([[0, 2, 3], [0, 2, 3], [1, 2, 4], [1, 2, 4]], [1.0] * 4)
    layers = om.layers_from_paths(paths, max_order=2)
    ns = layers[2]["node_sequence"]
    assert om.bipartite_edge_index(ns, "last").tolist() == [[0, 1, 2, 3], [2, 2, 3, 4]]
    assert om.bipartite_edge_index(ns, "first").tolist() == [[0, 1, 2, 3], [0, 1, 2, 2]]
    assert om.bipartite_edge_index(ns, "both").tolist() == [[0, 1, 2, 3, 0, 1, 2, 3], [0, 1, 2, 2, 2, 2, 3, 4]]


def test_known_answer_tutorial_tie_example():
    # reference docs/tutorial/trp_higher_order.ipynb:67 (input) and :671,1256,1796,2887 (sizes):
    # delta=1, K=5 -> (nodes, edges) per layer (4,6),(6,6),(6,4),(4,2),(2,0); L2 weights [2,1,1,1,1,1]
    ids = {c: i for i, c in enumerate("abcd")}
    ev = [("a", "b", 1), ("a", "b", 2), ("b", "a", 3), ("b", "c", 3), ("d", "c", 4), ("a", "b", 4), ("c", "b", 4),
          ("c", "d", 5), ("b", "a", 5), ("c", "b", 6)]
    ei = torch.tensor([[ids[u] for u, _, _ in ev], [ids[v] for _, v, _ in ev]])
    t = torch.tensor([x for _, _, x in ev])
    ei, t, _ = om.stable_time_sort(ei, t)
    layers = om.layers_from_temporal(ei, t, 4, delta=1, max_order=5, loop_lift=True)
    sizes = [(layers[k]["num_nodes"], layers[k]["edge_index"].size(1)) for k in range(1, 6)]
    assert sizes == [(4, 6), (6, 6), (6, 4), (4, 2), (2, 0)]
    assert sorted(layers[2]["edge_weight"].tolist(), reverse=True) == [2, 1, 1, 1, 1, 1]
    assert [layers[k]["inverse_idx"].numel() for k in (2, 3, 4, 5)] == [10, 7, 4, 2]


def test_graph_bookkeeping_oracle():
    ei = torch.tensor([[2, 0, 1, 0, 2], [0, 1, 2, 2, 1]])
    s, perm = oa.sort_by_row(ei)
    assert s.tolist() == [[0, 0, 1, 2, 2], [1, 2, 2, 0, 1]]
    c = oa.csr_csc(s, 3)
    assert c["row_ptr"].tolist() == [0, 2, 3, 5]
    assert c["col_ptr"].tolist() == [0, 1, 3, 5]
    assert c["row"].tolist() == [2, 0, 2, 0, 1]


def test_sorted_lift_equals_loop_on_random_streams():
    rng = np.random.default_rng(3)
    for trial in range(6):
        m, n = int(rng.integers(50, 400)), int(rng.integers(3, 30))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 60, m)))
        delta = int(rng.integers(1, 10))
        a = ol.temporal_lift_sorted(ei, t, delta, n)
        if a.size(1):
            assert torch.equal(a, ol.temporal_lift_per_timestamp(ei, t, delta))


# ------------------------------------------------------------------------------------------------------------------
# f2: temporal shortest paths (reference known answer: tests/algorithms/test_temporal.py:20-93, fixture conftest.py:58-83)
LONG_TEDGES = [("a", "b", 1), ("b", "c", 5), ("c", "d", 9), ("c", "e", 9), ("c", "f", 11), ("f", "a", 13), ("a", "g", 18), ("b", "f", 21),
               ("a", "g", 26), ("c", "f", 27), ("h", "f", 27), ("g", "h", 28), ("a", "c", 30), ("a", "b", 31), ("c", "h", 32), ("f", "h", 33),
               ("b", "i", 42), ("i", "b", 42), ("c", "i", 47), ("h", "i", 50)]
INF = float("inf")
LONG_DIST = np.array([[0, 1, 1, 3, 3, 3, 1, 2, INF], [3, 0, 1, 2, 2, 1, 4, 5, 1], [2, INF, 0, 1, 1, 1, 3, 1, 1],
                      [INF, INF, INF, 0, INF, INF, INF, INF, INF], [INF, INF, INF, INF, 0, INF, INF, INF, INF],
                      [1, INF, INF, INF, INF, 0, 2, 1, INF], [INF, INF, INF, INF, INF, INF, 0, 1, INF],
                      [INF, INF, INF, INF, INF, 1, INF, 0, 1], [INF, 1, INF, INF, INF, INF, INF, INF, 0]])
LONG_PRED = np.array([[0, 0, 0, 2, 2, 2, 0, 2, -1], [5, 1, 1, 2, 2, 1, 0, 6, 1], [5, -1, 2, 2, 2, 2, 0, 2, 2], [-1, -1, -1, 3, -1, -1, -1, -1, -1],
                      [-1, -1, -1, -1, 4, -1, -1, -1, -1], [5, -1, -1, -1, -1, 5, 0, 5, -1], [-1, -1, -1, -1, -1, -1, 6, 6, -1],
                      [-1, -1, -1, -1, -1, 7, -1, 7, 7], [-1, 8, -1, -1, -1, -1, -1, -1, 8]])


def long_temporal_arrays():
    names = sorted({x for e in LONG_TEDGES for x in e[:2]})
    ix = {k: i for i, k in enumerate(names)}
    ei = torch.tensor([[ix[a] for a, _, _ in LONG_TEDGES], [ix[b] for _, b, _ in LONG_TEDGES]])
    return ei, torch.tensor([t for _, _, t in LONG_TEDGES]), len(names)


def test_temporal_shortest_paths_reference_known_answer():
    from oracle import temporal_paths as tp
    ei, t, n = long_temporal_arrays()
    for fn in (tp.temporal_shortest_paths_reference, tp.temporal_shortest_paths_bfs):
        dist, pred = fn(ei, t, n, 10)
        assert dist.shape == (n, n) and pred.shape == (n, n)
        assert np.allclose(dist, LONG_DIST, equal_nan=True), fn.__name__
        assert np.array_equal(pred, LONG_PRED), fn.__name__


def test_temporal_shortest_paths_bfs_distances_equal_scipy_and_trees_are_valid():
    from oracle import temporal_paths as tp
    rng = np.random.default_rng(3)
    for _ in range(25):
        n, m = int(rng.integers(3, 14)), int(rng.integers(5, 80))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 50, m)))
        delta = int(rng.integers(1, 20))
        d_ref, p_ref = tp.temporal_shortest_paths_reference(ei, t, n, delta)
        d_bfs, p_bfs = tp.temporal_shortest_paths_bfs(ei, t, n, delta)
        assert np.array_equal(np.nan_to_num(d_ref, posinf=-1), np.nan_to_num(d_bfs, posinf=-1))
        assert np.array_equal(p_ref < 0, p_bfs < 0)                      # same reachability
        # both predecessor matrices name a node with an event into v: a valid last hop
        has_event = np.zeros((n, n), dtype=bool)
        has_event[ei[0].numpy(), ei[1].numpy()] = True
        for p in (p_ref, p_bfs):
            s_idx, v_idx = np.nonzero((p >= 0) & ~np.eye(n, dtype=bool))
            assert has_event[p[s_idx, v_idx], v_idx].all()


LONG_BETWEENNESS = {"a": 2.0, "b": 2.0, "c": 4.5, "d": 0, "e": 0, "f": 2.0, "g": 0.5, "h": 0, "i": 0}     # test_centrality.py:45-56, delta=5


def test_temporal_betweenness_reference_known_answer():
    from oracle import temporal_paths as tp
    ei, t, n = long_temporal_arrays()
    names = sorted({x for e in LONG_TEDGES for x in e[:2]})
    for fn in (tp.temporal_betweenness_reference, tp.temporal_betweenness_levels):
        bw = fn(ei, t, n, 5)
        assert {k: float(bw[i]) for i, k in enumerate(names)} == {k: float(v) for k, v in LONG_BETWEENNESS.items()}, fn.__name__


def test_temporal_betweenness_level_form_equals_reference_form():
    from oracle import temporal_paths as tp
    rng = np.random.default_rng(5)
    for _ in range(20):
        n, m = int(rng.integers(3, 14)), int(rng.integers(5, 90))
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 50, m)))
        delta = int(rng.integers(1, 20))
        a = tp.temporal_betweenness_reference(ei, t, n, delta)
        b = tp.temporal_betweenness_levels(ei, t, n, delta)
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-12)
