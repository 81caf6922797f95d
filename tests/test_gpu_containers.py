"""Graph / TemporalGraph container API against the reference's own unit tests (tests/core/test_graph.py:113-480,
tests/core/test_temporal_graph.py:57-125): derived graphs, attribute access, adjacency / Laplacian export, graph union."""
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


@pytest.fixture
def simple_graph(pp):
    return pp.Graph.from_edge_list([("a", "b"), ("b", "c"), ("a", "c")])


@pytest.fixture
def long_temporal_graph(pp):
    from test_oracle_golden import LONG_TEDGES
    return pp.TemporalGraph.from_edge_list(LONG_TEDGES)


def test_to_undirected_and_weighted_graph(pp, simple_graph):
    g_u = simple_graph.to_undirected()
    assert g_u.is_undirected() and g_u.data.edge_index.is_undirected and g_u.m == 3
    assert g_u.data.edge_index.tolist() == [[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]]
    simple_graph.data["edge_weight"] = torch.tensor([[1], [5], [2]])           # edges (a,b), (a,c), (b,c) after the row sort
    g_u = simple_graph.to_undirected()
    assert g_u.data.edge_weight.reshape(-1).tolist() == [1, 5, 1, 2, 5, 2]
    multi = pp.Graph.from_edge_list([("a", "b"), ("b", "c"), ("a", "c"), ("a", "b")])
    assert multi.m == 4
    weighted = multi.to_weighted_graph()
    assert weighted.data.num_edges == 3 and weighted.data.num_nodes == 3
    assert weighted["edge_weight", "a", "b"] == 2


def test_neighbours_edges_and_degrees(simple_graph):
    assert simple_graph.successors("a") == ["b", "c"] and len(simple_graph.successors("c")) == 0
    assert simple_graph.predecessors("b") == ["a"] and len(simple_graph.predecessors("a")) == 0
    assert simple_graph.get_successors(0).tolist() == [1, 2] and simple_graph.get_predecessors(2).tolist() == [0, 1]
    assert simple_graph.get_successors(7).numel() == 0
    for v, w, there in (("a", "b", True), ("b", "a", False), ("a", "c", True), ("c", "a", False), ("b", "c", True), ("c", "b", False)):
        assert simple_graph.is_edge(v, w) is there
    assert simple_graph.in_degrees == {"a": 0, "b": 1, "c": 2}
    assert simple_graph.out_degrees == {"a": 2, "b": 1, "c": 0}
    assert not simple_graph.has_self_loops() and simple_graph.is_directed() and not simple_graph.is_undirected()


def test_sparse_adj_matrix_and_laplacian(pp, simple_graph):
    import scipy.sparse as s
    adj = simple_graph.sparse_adj_matrix()
    assert adj.shape == (3, 3) and adj.nnz == 3
    simple_graph.data["edge_weight"] = torch.tensor([[1], [1], [2]])
    weighted = simple_graph.sparse_adj_matrix("edge_weight")
    assert isinstance(weighted, s.coo_matrix) and weighted.shape == (3, 3) and weighted.nnz == 3
    assert weighted.data.tolist() == [1, 1, 2]
    g = pp.Graph.from_edge_index(torch.tensor([[0], [1]]), num_nodes=5)
    assert g.sparse_adj_matrix().shape == (5, 5) and g.sparse_adj_matrix().nnz == 1
    g.data.edge_attr = torch.tensor([[1]])
    assert g.sparse_adj_matrix("edge_attr").nnz == 1
    lap = pp.Graph.from_edge_list([("a", "b"), ("b", "c"), ("a", "c")]).laplacian()
    assert isinstance(lap, s.coo_matrix) and lap.shape == (3, 3) and lap.nnz == 6
    assert lap.data.tolist() == [-1, -1, -1, 2, 1, 0]
    dense = pp.Graph.from_edge_list([("a", "b"), ("b", "a"), ("b", "c"), ("c", "b")])
    np.testing.assert_allclose(dense.laplacian("sym").toarray(), [[1, -2 ** -0.5, 0], [-2 ** -0.5, 1, -2 ** -0.5], [0, -2 ** -0.5, 1]], atol=1e-6)
    np.testing.assert_allclose(dense.laplacian("rw").toarray(), [[1, -1, 0], [-0.5, 1, -0.5], [0, -1, 1]], atol=1e-6)


def test_attribute_access(simple_graph):
    simple_graph["node_class"] = torch.tensor([[1], [2], [3]])
    assert simple_graph["node_class"].shape == (3, 1) and simple_graph["node_class", "b"].item() == 2
    simple_graph["node_class", "a"] = 42
    assert simple_graph["node_class", "a"].item() == 42
    with pytest.raises(KeyError):
        simple_graph["node_class", "d"].item()
    with pytest.raises(KeyError):
        simple_graph["node_class_1", "a"].item()
    with pytest.raises(KeyError):
        simple_graph["node_class_1", "a"] = 42
    simple_graph["edge_weight"] = torch.tensor([[1], [1], [2]])
    assert simple_graph["edge_weight", "a", "b"].item() == 1 and simple_graph["edge_weight", "b", "c"].item() == 2
    simple_graph["edge_weight", "a", "b"] = 42
    assert simple_graph["edge_weight", "a", "b"].item() == 42
    with pytest.raises(KeyError):
        simple_graph["edge_weight", "a", "d"].item()
    with pytest.raises(KeyError):
        simple_graph["edge_weight_1", "a", "b"] = 42
    simple_graph["graph_feature"] = torch.tensor([42])
    assert simple_graph["graph_feature"].item() == 42
    with pytest.raises(KeyError):
        simple_graph["graph_feature", "a"] = 42
    with pytest.raises(KeyError):
        simple_graph["nothing"]
    with pytest.raises(ValueError):
        simple_graph["node_x"] = torch.zeros(5)
    assert isinstance(str(simple_graph), str)


def test_add_operator(pp):
    idx = torch.IntTensor([[0, 1, 1], [1, 2, 3]])
    g1, g2 = pp.Graph.from_edge_index(idx, num_nodes=4), pp.Graph.from_edge_index(idx, num_nodes=4)
    g = g1 + g2
    assert g.n == 4 and g.m == 6 and g.data.edge_index.tolist() == [[0, 0, 1, 1, 1, 1], [1, 1, 2, 3, 2, 3]]
    g3 = pp.Graph.from_edge_index(torch.IntTensor([[0, 2, 3], [2, 3, 4]]), num_nodes=5)
    g = g1 + g2 + g3
    assert g.n == 5 and g.m == 9 and g.data.edge_index.tolist() == [[0, 0, 0, 1, 1, 1, 1, 2, 3], [1, 1, 2, 2, 3, 2, 3, 3, 4]]
    abcd, efgh, abgh = (pp.IndexMap(list(x)) for x in ("abcd", "efgh", "abgh"))
    g = pp.Graph.from_edge_index(idx, mapping=abcd) + pp.Graph.from_edge_index(idx, mapping=pp.IndexMap(list("abcd")))
    assert g.n == 4 and g.data.edge_index.tolist() == [[0, 0, 1, 1, 1, 1], [1, 1, 2, 3, 2, 3]]
    g = pp.Graph.from_edge_index(idx, mapping=abcd) + pp.Graph.from_edge_index(idx, mapping=efgh)
    assert g.n == 8 and g.data.edge_index.tolist() == [[0, 1, 1, 4, 5, 5], [1, 2, 3, 5, 6, 7]]
    a, b = pp.Graph.from_edge_index(idx, mapping=abcd), pp.Graph.from_edge_index(idx, mapping=abgh)
    a["node_class"], b["node_class"] = torch.tensor([[1], [2], [3], [4]]), torch.tensor([[5], [6], [7], [8]])
    a["edge_weight"], b["edge_weight"] = torch.tensor([[1], [2], [3]]), torch.tensor([[4], [5], [6]])
    g = a + b
    assert g.n == 6 and g.m == 6 and g.data.edge_index.tolist() == [[0, 0, 1, 1, 1, 1], [1, 1, 2, 3, 4, 5]]
    assert g["node_class"].tolist() == [[6], [8], [3], [4], [7], [8]]
    assert g["edge_weight"].tolist() == [[1], [4], [2], [3], [5], [6]]


def test_add_higher_order_graphs(pp):
    def walks(weight, count):
        p = pp.PathData(mapping=pp.IndexMap(["A", "B", "C", "D", "E"]))
        for _ in range(count):
            p.append_walk(("A", "C", "D"), weight=weight)
        for _ in range(count):
            p.append_walk(("B", "C", "E"), weight=weight)
        return p
    ho1 = pp.MultiOrderModel.from_path_data(walks(1.0, 2), max_order=2).layers[2]
    ho2 = pp.MultiOrderModel.from_path_data(walks(2.0, 1), max_order=2).layers[2]
    g = ho1 + ho2
    assert g.n == 4 and g.m == 4
    k = ho1.data.inverse_idx.size(0)
    assert (np.asarray(g.mapping.to_ids(g.data.inverse_idx[:k].cpu())) == np.asarray(ho1.mapping.to_ids(ho1.data.inverse_idx.cpu()))).all()
    assert (np.asarray(g.mapping.to_ids(g.data.inverse_idx[k:].cpu())) == np.asarray(ho2.mapping.to_ids(ho2.data.inverse_idx.cpu()))).all()


def test_temporal_graph_derived_graphs(long_temporal_graph):
    tg = long_temporal_graph
    g = tg.to_static_graph()
    assert g.n == tg.n and g.m == tg.m
    g = tg.to_static_graph(weighted=True)
    assert g.n == tg.n
    # a->b twice, a->c once (the reference reads positions 2 and 0: its unstable row sort reorders the edges of a row; here the
    # coalesced (row, col) order is kept, so the lookup goes by edge)
    assert g["edge_weight", "a", "b"].item() == 2.0 and g["edge_weight", "a", "c"].item() == 1.0
    assert g.data.edge_weight.sum().item() == tg.m
    assert tg.to_static_graph(time_window=(1, 10)).m == 4
    u = tg.to_undirected()
    assert u.n == tg.n and u.m == 2 * tg.m
    t1, t2 = tg.get_batch(1, 9), tg.get_batch(9, 13)
    assert (t1.n, t1.m, t2.n, t2.m) == (9, 8, 9, 4)
    tg.data.edge_tensor = torch.arange(tg.m, device=tg.data.edge_index.device)
    tg.data.edge_array = np.arange(tg.m)
    t3 = tg.get_batch(1, 9)
    assert t3.data.edge_tensor.tolist() == [1, 2, 3, 4, 5, 6, 7, 8] and t3.data.edge_array.tolist() == [1, 2, 3, 4, 5, 6, 7, 8]
    assert tg.get_window(1, 10).m == 4 and tg.get_window(10, 14).m == 2
    t4 = tg.get_window(2, 10)
    assert t4.data.edge_tensor.tolist() == [1, 2, 3] and t4.data.edge_array.tolist() == [1, 2, 3]
    assert tg["edge_tensor", "a", "b", 31].item() == 13 and tg["edge_tensor", "c", "e"].item() == 3
    before = sorted(tg.data.time.tolist())
    tg.shuffle_time()
    assert sorted(tg.data.time.tolist()) == before and tg.m == 20
    assert (tg.data.time[1:] >= tg.data.time[:-1]).all()
    assert isinstance(str(tg), str)


def test_static_graph_dataframe_io(pp, simple_graph, tmp_path):
    """tests/io/test_pandas.py:174-298,365-420: df_to_graph, add_node/edge_attributes, graph_to_df, csv round trip."""
    import pandas as pd
    io = pp.io
    g = io.df_to_graph(pd.DataFrame({"v": ["a", "b", "c"], "w": ["b", "c", "a"], "edge_weight": [2.0, 1.0, 42.0]}))
    assert (g.n, g.m) == (3, 3) and "edge_weight" in g.edge_attrs()
    assert g.data.edge_weight.tolist() == [2.0, 1.0, 42.0]
    g = io.df_to_graph(pd.DataFrame([["a", "b", 2.0], ["b", "c", 1.0], ["c", "a", 42.0]]))
    assert g.data.edge_attr_0.tolist() == [2.0, 1.0, 42.0]
    multi = pd.DataFrame({"v": ["a", "b", "c", "a"], "w": ["b", "c", "a", "b"], "edge_weight": [2.0, 1.0, 42.0, 3.0]})
    assert io.df_to_graph(multi, multiedges=False).m == 3 and io.df_to_graph(multi, multiedges=True).m == 4
    g = io.df_to_graph(pd.DataFrame({"v": ["a", "b", "c"], "w": ["b", "c", "a"], "edge_weight": ["a", "b", "c"]}), is_undirected=True)
    assert (g.n, g.m) == (3, 3) and g.is_undirected()

    io.add_node_attributes(pd.DataFrame({"v": ["b", "a", "c"], "x": [2, 1, 3], "node_y": [0.2, 0.1, 0.3]}), simple_graph)
    assert simple_graph.data["node_x"].tolist() == [1, 2, 3]
    assert torch.allclose(simple_graph.data["node_y"].cpu(), torch.tensor([0.1, 0.2, 0.3], dtype=torch.double))
    io.add_node_attributes(pd.DataFrame({"index": [1, 0, 2], "z": [20, 10, 30]}), simple_graph)
    assert simple_graph.data["node_z"].tolist() == [10, 20, 30]
    with pytest.raises(ValueError, match="multiple attribute values for single node"):
        io.add_node_attributes(pd.DataFrame({"v": ["a", "a", "b", "c"], "x": [1, 2, 3, 4]}), simple_graph)
    with pytest.raises(ValueError, match="Mismatch between nodes"):
        io.add_node_attributes(pd.DataFrame({"v": ["a", "b", "d"], "x": [1, 2, 3]}), simple_graph)
    with pytest.raises(ValueError, match="must either have `index` or `v` column"):
        io.add_node_attributes(pd.DataFrame({"foo": [1, 2, 3]}), simple_graph)

    io.add_edge_attributes(pd.DataFrame({"v": ["a", "b", "a"], "w": ["b", "c", "c"], "weight": [1, 3, 2]}), simple_graph)
    assert simple_graph.data["edge_weight"].tolist() == [1, 2, 3]
    io.add_edge_attributes(pd.DataFrame({"v": ["a", "b", "a"], "w": ["b", "c", "c"], "edge_score": [5, 6, 7]}), simple_graph)
    assert simple_graph.data["edge_score"].tolist() == [5, 7, 6]
    with pytest.raises(ValueError, match="Please ensure all nodes in the DataFrame are present in the graph."):
        io.add_edge_attributes(pd.DataFrame({"v": ["a", "x", "a"], "w": ["b", "c", "c"], "weight": [1.0, 2.0, 3.0]}), simple_graph)
    with pytest.raises(ValueError, match="does not exist in the graph"):
        io.add_edge_attributes(pd.DataFrame({"v": ["a", "b", "a"], "w": ["a", "c", "c"], "weight": [1.0, 2.0, 3.0]}), simple_graph)
    tg = pp.TemporalGraph.from_edge_list([("a", "b", 1), ("b", "c", 5), ("c", "d", 9), ("c", "e", 9)])
    io.add_edge_attributes(pd.DataFrame({"v": ["a", "b", "c", "c"], "w": ["b", "c", "e", "d"], "t": [1, 5, 9, 9], "weight": [1, 2, 4, 3]}),
                           tg, time_attr="t")
    assert tg.data["edge_weight"].tolist() == [1, 2, 3, 4]
    with pytest.raises(ValueError, match="Please ensure the DataFrame matches the number of edges in the graph"):
        io.add_edge_attributes(pd.DataFrame({"v": ["a"], "w": ["b"], "t": [99], "weight": [1.0]}), tg, time_attr="t")
    with pytest.raises(ValueError, match="does not exist at time"):
        io.add_edge_attributes(pd.DataFrame({"v": ["a", "b", "c", "c"], "w": ["b", "c", "d", "e"], "t": [1, 5, 9, 10],
                                             "weight": [1.0, 2.0, 3.0, 4.0]}), tg, time_attr="t")

    plain = pp.Graph.from_edge_list([("a", "b"), ("b", "c"), ("a", "c")])
    df = io.graph_to_df(plain)
    assert set(df.columns) == {"v", "w"} and len(df) == 3 and set(df["v"]) == {"a", "b"} and set(df["w"]) == {"b", "c"}
    plain.data.edge_weight = torch.tensor([1.0, 2.0, 3.0])
    plain.data.edge_label = torch.tensor([0, 1, 2])
    df = io.graph_to_df(plain)
    assert list(df["edge_weight"]) == [1.0, 2.0, 3.0] and list(df["edge_label"]) == [0, 1, 2]
    assert set(io.graph_to_df(plain, node_indices=True)["v"]) == {0, 1}
    path = tmp_path / "graph.csv"
    io.write_csv(plain, path_or_buf=str(path))
    back = io.read_csv_graph(str(path))
    assert (back.n, back.m) == (3, 3) and back.data.edge_weight.tolist() == [1.0, 2.0, 3.0]
