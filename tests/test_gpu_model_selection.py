"""Model-selection statistics of MultiOrderModel against the reference's own known answers
(reference tests/core/test_multi_order_model.py:45-162,193-224)."""
import numpy as np
import pytest
import torch
from scipy.stats import chi2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import pathpyg_amd
    return pathpyg_amd


def _paths(pp, walks, weights=None, ids="abcde", device=DEV):
    p = pp.PathData(pp.IndexMap(list(ids)), device=device)
    for i, w in enumerate(walks):
        p.append_walk(w, weight=1.0 if weights is None else weights[i])
    return p


def test_dof(pp):
    line = _paths(pp, [("a", "b", "c", "d")], ids="abcd")
    m = pp.MultiOrderModel.from_path_data(line, max_order=4)
    for order in range(5):
        assert m.get_mon_dof(assumption="paths", max_order=order) == 3
    toy = _paths(pp, [("a", "c", "d"), ("b", "c", "e")])
    m = pp.MultiOrderModel.from_path_data(toy, max_order=2, mode="propagation")
    assert [m.get_mon_dof(assumption="paths", max_order=k) for k in (0, 1, 2)] == [4, 5, 7]
    assert m.get_mon_dof(assumption="ngrams", max_order=2) == 4 + 5 * 4 + 25 * 4
    with pytest.raises(ValueError):
        m.get_mon_dof(max_order=3)
    with pytest.raises(ValueError):
        m.get_mon_dof(assumption="nonsense")


@pytest.mark.parametrize("device", [DEV, None])
def test_log_likelihood(pp, device):
    toy = _paths(pp, [("a", "c", "d"), ("b", "c", "e")], device=device)
    m = pp.MultiOrderModel.from_path_data(toy, max_order=2, mode="propagation")
    d = toy.data
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=0), np.log(1 / 6) * 4 + np.log(2 / 6) * 2)
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=1), np.log(1 / 6) * 2 + 0 + 2 * np.log(1 / 2))
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=2), np.log(1 / 6) * 2 + 0 + 0)
    toy = _paths(pp, [("a", "c", "d"), ("b", "c", "e"), ("a", "c", "e"), ("b", "c", "d")], device=device)
    m = pp.MultiOrderModel.from_path_data(toy, max_order=2, mode="propagation")
    d = toy.data
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=0), np.log(2 / 12) * 8 + np.log(4 / 12) * 4)
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=1), np.log(2 / 12) * 4 + 0 + 4 * np.log(1 / 2))
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=2), np.log(1 / 6) * 4 + 0 + 4 * np.log(1 / 2))
    toy = _paths(pp, [("a",), ("a", "b"), ("a", "b", "c")], device=device)
    m = pp.MultiOrderModel.from_path_data(toy, max_order=2, mode="propagation")
    d = toy.data
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=0), np.log(3 / 6) * 3 + np.log(2 / 6) * 2 + np.log(1 / 6) * 1)
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=1), np.log(3 / 6) * 3)
    assert np.isclose(m.get_mon_log_likelihood(d, max_order=2), np.log(3 / 6) * 3)


def test_likelihood_ratio_test_and_estimate_order(pp):
    thr = 0.1
    llh0 = np.log(1 / 6) * 4 + np.log(2 / 6) * 2
    llh1 = np.log(1 / 6) * 2 + 0 + 2 * np.log(1 / 2)
    llh2 = np.log(1 / 6) * 2
    p01 = 1 - chi2.cdf(-2 * (llh0 - llh1), 5 - 4)
    p12 = 1 - chi2.cdf(-2 * (llh1 - llh2), 7 - 5)
    toy = _paths(pp, [("a", "c", "d"), ("b", "c", "e")])
    m = pp.MultiOrderModel.from_path_data(toy, max_order=2)
    r01, q01 = m.likelihood_ratio_test(toy.data, max_order_null=0, max_order=1, assumption="paths", significance_threshold=thr)
    r12, q12 = m.likelihood_ratio_test(toy.data, max_order_null=1, max_order=2, assumption="paths", significance_threshold=thr)
    assert r01 == (p01 < thr) and np.isclose(q01, p01) and r12 == (p12 < thr) and np.isclose(q12, p12)
    with pytest.raises(ValueError):
        m.likelihood_ratio_test(toy.data, max_order_null=2, max_order=1)
    weak = _paths(pp, [("a", "c", "d"), ("b", "c", "e")], [3, 3])
    assert pp.MultiOrderModel.from_path_data(weak, max_order=2).estimate_order(weak, max_order=2, significance_threshold=0.01) == 1
    strong = _paths(pp, [("a", "c", "d"), ("b", "c", "e")], [4, 4])
    assert pp.MultiOrderModel.from_path_data(strong, max_order=2).estimate_order(strong, max_order=2, significance_threshold=0.01) == 2


def test_paths_indexing(pp):
    # reference tests/core/test_multi_order_model.py:193-224: start indices of shrunken walks
    walks = [("d", "b", "c"), ("a", "b", "c"), ("a", "b", "e"), ("d", "b", "e"), ("a",)]
    data = pp.PathData(pp.IndexMap(sorted({v for w in walks for v in w})), device=DEV)
    data.append_walks(node_seqs=walks, weights=[1, 20, 1, 20, 1])
    mon = pp.MultiOrderModel.from_path_data(data, max_order=3)
    assert mon.estimate_order(data, max_order=3) == 2


def test_degrees_and_transition_probabilities(pp):
    g0 = torch.Generator().manual_seed(0)
    ei = torch.randint(0, 50, (2, 900), generator=g0)
    w = torch.rand(900, generator=g0) + 0.1
    g = pp.Graph(pp.Data(edge_index=ei.to(DEV), edge_weight=w.to(DEV), num_nodes=55))
    s, wt = g.data.edge_index.cpu(), g.data.edge_weight.cpu()
    assert torch.equal(g.degrees("out", return_tensor=True).cpu().long(), torch.bincount(s[0], minlength=55))
    assert torch.equal(g.degrees("in", return_tensor=True).cpu().long(), torch.bincount(s[1], minlength=55))
    wout = torch.zeros(55).index_add_(0, s[0], wt)
    win = torch.zeros(55).index_add_(0, s[1], wt)
    torch.testing.assert_close(g.degrees("out", "edge_weight", True).cpu(), wout, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g.degrees("in", "edge_weight", True).cpu(), win, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g.transition_probabilities("edge_weight").cpu(), wt / wout[s[0]], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(g.transition_probabilities().cpu(), 1.0 / torch.bincount(s[0], minlength=55)[s[0]].float())
    assert g.degrees("out")[3] == int(torch.bincount(s[0], minlength=55)[3])
