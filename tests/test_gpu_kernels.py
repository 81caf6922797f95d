"""GPU parity tests of the HIP kernels, called through the C ABI (ctypes), against the CPU oracle
and the golden vectors produced by the reference's own source.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import golden_delta

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from pathpyg_amd import _hip
    return _hip


def cu(x):
    return x.to(DEV)


# ---------------------------------------------------------------- primitives
@pytest.mark.parametrize("n", [0, 1, 63, 64, 2047, 2048, 2049, 100_003, 3_000_001])
def test_exclusive_scan(hip, n):
    g = torch.Generator().manual_seed(n)
    for dtype in (torch.int32, torch.int64):
        v = torch.randint(0, 1000, (n,), generator=g, dtype=dtype)
        want = torch.zeros(n + 1, dtype=torch.int64)
        want[1:] = torch.cumsum(v.long(), 0)
        assert torch.equal(hip.exclusive_scan(cu(v)).cpu(), want)


@pytest.mark.parametrize("sorted_input", [True, False])
def test_degree(hip, sorted_input):
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 5000, (1_000_003,), generator=g)
    idx[:70000] = 17                       # a hub: long runs inside waves
    if sorted_input:
        idx = idx.sort().values
    want = torch.bincount(idx, minlength=6000).int()
    assert torch.equal(hip.degree(cu(idx), 6000).cpu(), want)
    assert hip.minmax(cu(idx)) == (int(idx.min()), int(idx.max()))


@pytest.mark.parametrize("n", [1, 5, 64, 4095, 4096, 4097, 10_000, 1_000_003])
@pytest.mark.parametrize("bits", [1, 7, 8, 9, 19, 32])
def test_sort_pairs_u32(hip, n, bits):
    g = torch.Generator().manual_seed(n * 31 + bits)
    hi = 2 ** min(bits, 31)
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64).int()
    ks, vs = hip.sort_pairs(cu(keys), None, 0, bits)
    order = torch.sort(keys.long(), stable=True)
    assert torch.equal(ks.cpu().long(), order.values)
    assert torch.equal(vs.cpu().long(), order.indices)
    # explicit values + a bit window: stable w.r.t. the masked key
    vals = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).int()
    ks2, vs2 = hip.sort_pairs(cu(keys), cu(vals), 2, max(bits, 3))
    masked = (keys.long() >> 2) & ((1 << (max(bits, 3) - 2)) - 1)
    perm = torch.sort(masked, stable=True).indices
    assert torch.equal(ks2.cpu(), keys[perm])
    assert torch.equal(vs2.cpu(), vals[perm])


@pytest.mark.parametrize("n", [3, 4097, 300_001])
@pytest.mark.parametrize("bits", [5, 33, 48, 64])
def test_sort_pairs_u64(hip, n, bits):
    g = torch.Generator().manual_seed(n + bits)
    keys = torch.randint(0, 2 ** min(bits, 62), (n,), generator=g, dtype=torch.int64)
    ks, vs = hip.sort_pairs(cu(keys), None, 0, bits)
    order = torch.sort(keys, stable=True)
    assert torch.equal(ks.cpu(), order.values)
    assert torch.equal(vs.cpu().long(), order.indices)


# ---------------------------------------------------------------- lifts vs golden (reference source) vectors
TEMPORAL = ["int_ties", "int_unique", "int_wide", "int_delta0", "int_fdelta", "int_f64delta",
            "f64_ties", "f64_npdelta", "f64_intdelta", "int_big"]


@pytest.mark.parametrize("name", TEMPORAL)
def test_temporal_lift_golden(hip, golden, name):
    ei = torch.from_numpy(golden[f"temporal/{name}/edge_index"])
    t = torch.from_numpy(golden[f"temporal/{name}/time"])
    out = hip.temporal_lift(cu(ei), cu(t), int(golden[f"temporal/{name}/num_nodes"]), golden_delta(golden, name))
    want = torch.from_numpy(golden[f"temporal/{name}/out"])
    assert out.dtype == torch.int64 and out.is_contiguous()
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("name", ["small", "multi", "isolated", "hub", "wide"])
def test_linegraph_lift_golden(hip, golden, name):
    ei = torch.from_numpy(golden[f"linegraph/{name}/edge_index"])
    n = int(golden[f"linegraph/{name}/num_nodes"])
    out = hip.linegraph_lift(cu(ei), n)
    assert torch.equal(out.cpu(), torch.from_numpy(golden[f"linegraph/{name}/out"]))
    w = torch.from_numpy(golden[f"linegraph/{name}/edge_weight"])
    for aggr in ("src", "dst", "max", "mul", "add"):
        got = hip.edge_attr(out, cu(w), aggr)
        assert torch.equal(got.cpu(), torch.from_numpy(golden[f"linegraph/{name}/w_{aggr}"]))


def test_chained_lifts_golden(hip, golden):
    ei = cu(torch.from_numpy(golden["chain/edge_index"]))
    t = cu(torch.from_numpy(golden["chain/time"]))
    ho = hip.temporal_lift(ei, t, int(golden["chain/num_nodes"]), int(golden["chain/delta"]))
    assert torch.equal(ho.cpu(), torch.from_numpy(golden["chain/k2"]))
    n_inst = ei.size(1)
    for k in (3, 4, 5):
        nxt = hip.linegraph_lift(ho, n_inst)
        n_inst, ho = ho.size(1), nxt
        assert torch.equal(ho.cpu(), torch.from_numpy(golden[f"chain/k{k}"]))


# ---------------------------------------------------------------- lifts vs the oracle on seeded streams
def _stream(seed, m, n, span, float_time=False):
    rng = np.random.default_rng(seed)
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    t = np.sort(np.round(rng.random(m) * span, 1) if float_time else rng.integers(0, span, m))
    return ei, torch.from_numpy(t)


@pytest.mark.parametrize("seed,m,n,span,delta,ft", [
    (1, 20_000, 50, 400, 7, False),          # heavy ties, hubs
    (2, 200_000, 10_000, 10 ** 6, 30_000, False),
    (3, 100_000, 300, 5_000, 2.5, True),
    (4, 100_000, 300, 10 ** 9, 1.0e6, False),  # python float delta on int64 time: float32 path
    (5, 50_000, 3, 50_000, 40, False),        # 3 nodes: very long per-node lists
    (6, 1, 4, 10, 3, False),
    (7, 4097, 1, 4097, 5, False),             # single node, every event continues every close one
])
def test_temporal_lift_oracle(hip, seed, m, n, span, delta, ft):
    from oracle import lift as ol
    ei, t = _stream(seed, m, n, span, ft)
    want = ol.temporal_lift_sorted(ei, t, delta, n)
    got = hip.temporal_lift(cu(ei), cu(t), n, delta)
    assert got.shape == want.shape
    assert torch.equal(got.cpu(), want)


def test_temporal_lift_empty_and_errors(hip):
    ei = torch.empty((2, 0), dtype=torch.long)
    out = hip.temporal_lift(cu(ei), cu(torch.empty(0, dtype=torch.long)), 5, 1)
    assert out.shape == (2, 0)
    ei = torch.tensor([[0, 9], [1, 2]])
    with pytest.raises(IndexError):
        hip.temporal_lift(cu(ei), cu(torch.tensor([1, 2])), 3, 1)
    with pytest.raises(TypeError):
        hip.temporal_lift(cu(torch.tensor([[0], [1]])), cu(torch.tensor([1.0])), 3, 1)


@pytest.mark.parametrize("seed,e,n", [(1, 100_000, 1000), (2, 300_000, 50_000), (3, 20_000, 7)])
def test_linegraph_lift_oracle(hip, seed, e, n):
    from oracle import lift as ol
    rng = np.random.default_rng(seed)
    src = np.sort(rng.integers(0, n, e))
    dst = rng.integers(0, n, e)
    ei = torch.from_numpy(np.stack([src, dst]))
    want = ol.line_graph_lift(ei, n)
    got = hip.linegraph_lift(cu(ei), n)
    assert torch.equal(got.cpu(), want)


def test_edge_attr_dtypes_and_rows(hip):
    from oracle import lift as ol
    g = torch.Generator().manual_seed(0)
    ei = torch.randint(0, 100, (2, 5000), generator=g)
    for dtype in (torch.int32, torch.int64, torch.float32, torch.float64):
        a = (torch.rand(100, generator=g) * 50).to(dtype)
        for aggr in ("src", "dst", "max", "mul", "add"):
            assert torch.equal(hip.edge_attr(cu(ei), cu(a), aggr).cpu(), ol.edge_attribute_from_nodes(ei, a, aggr))
    a2 = torch.rand(100, 3, generator=g)
    assert torch.equal(hip.edge_attr(cu(ei), cu(a2), "add").cpu(), ol.edge_attribute_from_nodes(ei, a2, "add"))
    with pytest.raises(ValueError):
        hip.edge_attr(cu(ei), cu(a2), "unknown")


def test_extend_node_sequence(hip):
    from oracle import model as om
    g = torch.Generator().manual_seed(0)
    for k in (1, 2, 4):
        ns = torch.randint(0, 50, (300, k), generator=g)
        ei = torch.randint(0, 300, (2, 2000), generator=g)
        assert torch.equal(hip.extend_node_sequence(cu(ei), cu(ns)).cpu(), om.extend_node_sequence(ns, ei))


# ---------------------------------------------------------------- aggregation
@pytest.mark.parametrize("m,k,hi", [(1, 1, 5), (1000, 1, 50), (5000, 2, 30), (200_000, 2, 3000), (100_000, 3, 40),
                                     (50_000, 5, 6), (10_000, 2, 2 ** 40)])
def test_unique_rows(hip, m, k, hi):
    g = torch.Generator().manual_seed(m + k)
    rows = torch.randint(0, hi, (m, k), generator=g)
    if hi > 2 ** 32:
        rows[::2] = rows[1::2] if m % 2 == 0 else rows[::2]
    uniq, inv = hip.unique_rows(cu(rows))
    wu, wi = torch.unique(rows, dim=0, return_inverse=True)
    assert torch.equal(uniq.cpu(), wu)
    assert torch.equal(inv.cpu(), wi)


def test_unique_rows_negative_values(hip):
    rows = torch.tensor([[3, -1], [-5, 2], [3, -1], [0, 0], [-5, 1]])
    uniq, inv = hip.unique_rows(cu(rows))
    wu, wi = torch.unique(rows, dim=0, return_inverse=True)
    assert torch.equal(uniq.cpu(), wu) and torch.equal(inv.cpu(), wi)


@pytest.mark.parametrize("e,n", [(1, 1), (1000, 10), (100_000, 300), (300_000, 70_000)])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_coalesce(hip, e, n, reduce):
    from oracle import aggregate as oa
    g = torch.Generator().manual_seed(e)
    ei = torch.randint(0, n, (2, e), generator=g)
    for w in (torch.randint(1, 5, (e,), generator=g).float(), torch.rand(e, generator=g),
              torch.randint(-9, 9, (e,), generator=g), torch.rand(e, generator=g, dtype=torch.float64)):
        wi, ww = oa.coalesce(ei, w, n, reduce)
        gi, gw = hip.coalesce(cu(ei), cu(w), n, reduce)
        assert torch.equal(gi.cpu(), wi)
        if w.is_floating_point() and reduce == "mean":
            torch.testing.assert_close(gw.cpu(), ww, rtol=1e-6, atol=0)
        else:
            assert torch.equal(gw.cpu(), ww)          # left-to-right accumulation order matches the CPU scatter


@pytest.mark.parametrize("e,n", [(1, 1), (5000, 7), (200_000, 300), (300_000, 70_000)])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_coalesce_unit_weights_equal_a_ones_vector(hip, e, n, reduce):
    """`weight=UNIT` (the reference's default torch.ones, lift_order.py:130-131, without the vector): merged weight = run length for "sum",
    1 otherwise — bit-identical to coalescing an explicit float32 ones vector, also with a remap and with the inverse map, and through
    aggregate_edge_index's default."""
    import pathpyg_amd as pp
    from oracle import aggregate as oa
    g = torch.Generator().manual_seed(e + 17)
    ei = torch.randint(0, n, (2, e), generator=g)
    ones = torch.ones(e)
    wi, ww = oa.coalesce(ei, ones, n, reduce)
    gi, gw = hip.coalesce(cu(ei), hip.UNIT, n, reduce)
    assert gw.dtype == torch.float32 and torch.equal(gi.cpu(), wi) and torch.equal(gw.cpu(), ww)
    hi, hw, hinv = hip.coalesce(cu(ei), cu(ones), n, reduce, None, True)
    ui, uw, uinv = hip.coalesce(cu(ei), hip.UNIT, n, reduce, None, True)
    assert torch.equal(ui, hi) and torch.equal(uw, hw) and torch.equal(uinv, hinv)
    remap = torch.randperm(n, generator=g)
    ri, rw = hip.coalesce(cu(ei), hip.UNIT, n, reduce, cu(remap))
    si, sw = hip.coalesce(cu(ei), cu(ones), n, reduce, cu(remap))
    assert torch.equal(ri, si) and torch.equal(rw, sw)
    if reduce == "sum" and e > 1:
        seq = torch.arange(n).unsqueeze(1)
        a = pp.algorithms.lift_order.aggregate_edge_index(cu(ei), cu(seq))                      # default weights
        b = pp.algorithms.lift_order.aggregate_edge_index(cu(ei), cu(seq), cu(ones))
        assert torch.equal(pp._dispatch.plain(a.data.edge_index), pp._dispatch.plain(b.data.edge_index))
        assert torch.equal(a.data.edge_weight, b.data.edge_weight)
    with pytest.raises(ValueError):
        hip.coalesce(cu(ei), "ones", n, reduce)


def test_coalesce_remap_and_bad_index(hip):
    from oracle import aggregate as oa
    g = torch.Generator().manual_seed(3)
    remap = torch.randint(0, 40, (500,), generator=g)
    ei = torch.randint(0, 500, (2, 20_000), generator=g)
    w = torch.ones(20_000)
    wi, ww = oa.coalesce(remap[ei], w, 40, "sum")
    gi, gw = hip.coalesce(cu(ei), cu(w), 40, "sum", remap=cu(remap))
    assert torch.equal(gi.cpu(), wi) and torch.equal(gw.cpu(), ww)
    with pytest.raises(ValueError):
        hip.coalesce(cu(ei), cu(w), 10, "sum", remap=cu(remap))


def test_graph_bookkeeping(hip):
    from oracle import aggregate as oa
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, 1000, (2, 50_000), generator=g)
    assert not hip.is_sorted(cu(ei[0]))
    perm = hip.argsort(cu(ei[0]))
    s, wperm = oa.sort_by_row(ei)
    assert torch.equal(perm.cpu(), wperm)
    assert hip.is_sorted(cu(s[0]))
    c = oa.csr_csc(s, 1200)
    assert torch.equal(hip.ptr_from_sorted(cu(s[0]), 1200).cpu(), c["row_ptr"])
    by_col = hip.argsort(cu(s[1]))
    assert torch.equal(by_col.cpu(), c["csc_perm"])
    assert torch.equal(hip.ptr_from_sorted(cu(s[1])[by_col], 1200).cpu(), c["col_ptr"])


def test_temporal_lift_edge_range_shards_reassemble(hip):
    """The multi-GPU decomposition on one GPU: lifting every edge-range shard (own events + forward halo,
    ``n_own`` sources, ``id_offset``) and concatenating in shard order gives the whole result."""
    from pathpyg_amd.distributed import event_ranges, halo_end
    ei, t = _stream(11, 150_000, 700, 30_000)
    full = hip.temporal_lift(cu(ei), cu(t), 700, 250)
    for world in (2, 5):
        parts = []
        for lo, hi in event_ranges(ei.size(1), world):
            end = halo_end(t, hi, 250)
            parts.append(hip.temporal_lift(cu(ei[:, lo:end]), cu(t[lo:end]), 700, 250, n_own=hi - lo, id_offset=lo))
        assert torch.equal(torch.cat(parts, dim=1), full)


def test_linegraph_lift_edge_range_shards_reassemble(hip):
    """Same for the line-graph lift (k -> k+1): every shard expands its own edge range against the degrees / row pointers of the
    whole list; the blocks concatenate to the whole result."""
    from pathpyg_amd.distributed import event_ranges
    ei, t = _stream(12, 120_000, 500, 20_000)
    ho = hip.temporal_lift(cu(ei), cu(t), 500, 300)                  # a source-sorted edge list over 120 000 "nodes" (events)
    full = hip.linegraph_lift(ho, ei.size(1))
    for world in (1, 3, 8):
        parts = [hip.linegraph_lift(ho, ei.size(1), (lo, hi)) for lo, hi in event_ranges(ho.size(1), world)]
        assert torch.equal(torch.cat(parts, dim=1), full)
    assert hip.linegraph_lift(ho, ei.size(1), (5, 5)).shape == (2, 0)
    with pytest.raises(ValueError):
        hip.linegraph_lift(ho, ei.size(1), (10, ho.size(1) + 1))


@pytest.mark.parametrize("case", ["all_same_time", "negative_delta", "huge_delta", "one_hub", "self_loops_only", "float_ties_exact_boundary"])
def test_temporal_lift_adversarial_streams(hip, case):
    """Edge cases of the window logic against the oracle (which is pinned to the reference source on the golden vectors)."""
    from oracle import lift as ol
    rng = np.random.default_rng(sum(map(ord, case)))
    m, n, delta = 20_000, 40, 5
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    t = torch.from_numpy(np.sort(rng.integers(0, 3000, m)))
    if case == "all_same_time":
        t = torch.full((m,), 7, dtype=torch.long)                     # no pair at all: t_j > t_i never holds
    elif case == "negative_delta":
        delta = -3
    elif case == "huge_delta":
        m = 3000
        ei, t, delta = ei[:, :m], t[:m], 10 ** 15                     # every later event of the head node continues
    elif case == "one_hub":
        ei[1, :] = 3                                                  # every event points at node 3
        ei[0, ::2] = 3                                                # and half of them leave it: long per-node list, big counts
    elif case == "self_loops_only":
        ei[1] = ei[0]
    elif case == "float_ties_exact_boundary":
        t = torch.from_numpy(np.sort(rng.integers(0, 400, m)) * 0.25)  # exact binary fractions: t_j == t_i + delta happens often
        delta = 0.75
    want = ol.temporal_lift_sorted(ei, t, delta, n)
    got = hip.temporal_lift(cu(ei), cu(t), n, delta)
    assert got.shape == want.shape
    assert torch.equal(got.cpu(), want)
    if case in ("all_same_time", "negative_delta"):
        assert got.size(1) == 0


def test_linegraph_lift_hub_and_empty(hip):
    from oracle import lift as ol
    # star: every edge points into the hub 0, which has 3000 out-edges -> 3000 lifted edges per in-edge
    src = torch.cat((torch.zeros(3000, dtype=torch.long), torch.arange(1, 501)))
    dst = torch.cat((torch.arange(1, 3001) % 900 + 1, torch.zeros(500, dtype=torch.long)))
    ei = torch.stack((src, dst))
    assert torch.equal(hip.linegraph_lift(cu(ei), 3001).cpu(), ol.line_graph_lift(ei, 3001))
    empty = torch.empty((2, 0), dtype=torch.long)
    assert hip.linegraph_lift(cu(empty), 5).shape == (2, 0)
    no_continuation = torch.tensor([[0, 1], [2, 3]])                 # nobody leaves nodes 2 and 3
    assert hip.linegraph_lift(cu(no_continuation), 4).shape == (2, 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int64, torch.int32])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_coalesce_with_long_duplicate_runs(hip, dtype, reduce):
    """One node pair carries 40 % of all edges (20 000 parallel edges) and another 900: the wave-cooperative reduction of long runs
    must give the oracle's result (integer-valued weights: exact also for float sums)."""
    from oracle import aggregate as oa
    _hip = hip
    g = torch.Generator().manual_seed(3)
    n, e = 500, 50_000
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[:, :20_000] = torch.tensor([[7], [3]])
    ei[:, 20_000:20_900] = torch.tensor([[400], [400]])
    ei = ei[:, torch.randperm(e, generator=g)]
    w = torch.randint(-50, 50, (e,), generator=g).to(dtype)
    want_index, want_w = oa.coalesce(ei, w, n, reduce)
    got_index, got_w = _hip.coalesce(ei.to(DEV), w.to(DEV), n, reduce)
    assert torch.equal(got_index.cpu(), want_index)
    assert torch.equal(got_w.cpu(), want_w), (dtype, reduce)


def test_temporal_lift_rejects_unsorted_time(hip):
    """every search of the kernel assumes ascending timestamps; a stream modified after TemporalGraph sorted it must raise."""
    g = torch.Generator().manual_seed(2)
    ei = torch.randint(0, 20, (2, 500), generator=g).to(DEV)
    t = torch.sort(torch.randint(0, 300, (500,), generator=g)).values.to(DEV)
    assert hip.temporal_lift(ei, t, 20, 5).size(0) == 2
    t[250] = 0                                                     # a descent in the middle of the stream
    with pytest.raises(ValueError, match="not sorted by time"):
        hip.temporal_lift(ei, t, 20, 5)


@pytest.mark.parametrize("n,f", [(1, 1), (37, 5), (1000, 16), (4097, 64), (300, 250)])
def test_dropout_kernels_match_the_integer_hash(hip, n, f):
    """pp_dropout_f32 / pp_dropout_act_backward_f32 against the torch statement of the same counter-based mask (nn.sharded.dropout_mask):
    bit-identical keep decisions for consecutive rows (row0), explicit global row ids (rows), in place, and the fused ELU backward +
    column sums."""
    from pathpyg_amd.nn.sharded import dropout_mask
    g = torch.Generator().manual_seed(n * 131 + f)
    x = torch.randn(n, f, generator=g)
    p, seed, tag = 0.4, 123456789, 65
    row0 = 2 ** 33 + 11                                               # the row * width product crosses 32 bits
    want = x * dropout_mask(torch.arange(row0, row0 + n), f, p, seed, tag)
    got = hip.dropout(x.to(DEV), p, seed, tag, row0)
    assert torch.equal(got.cpu(), want)
    rows = torch.randint(0, 2 ** 40, (n,), generator=g)
    want_rows = x * dropout_mask(rows, f, p, seed, tag)
    buf = x.to(DEV)
    assert hip.dropout(buf, p, seed, tag, 0, rows.to(DEV), buf) is buf
    assert torch.equal(buf.cpu(), want_rows)
    assert 0.5 < float((want != 0).float().mean()) < 0.7 or n * f < 500   # about 1 - p of the elements survive
    # backward: dy * mask * ELU'(y) with y recovered from the dropped activation
    y = torch.nn.functional.elu(torch.randn(n, f, generator=g))
    mask = dropout_mask(torch.arange(row0, row0 + n), f, p, seed, tag)
    dy = torch.randn(n, f, generator=g)
    for act in (True, False):
        want_g = dy * mask * (torch.where(y > 0, torch.ones_like(y), y + 1) if act else 1.0)
        got_g, got_b = hip.dropout_act_backward(dy.to(DEV), (y * mask).to(DEV) if act else None, p, seed, tag, row0, None, act, True)
        torch.testing.assert_close(got_g.cpu(), want_g, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(got_b.cpu(), want_g.sum(0), rtol=1e-4, atol=1e-4 * max(1.0, n ** 0.5))
    # p = 0 keeps everything; bad p is refused
    assert torch.equal(hip.dropout(x.to(DEV), 0.0, seed, tag).cpu(), x)
    with pytest.raises(Exception):
        hip.dropout(x.to(DEV), 1.0, seed, tag)


@pytest.mark.parametrize("dtype", [torch.int64, torch.float64])
def test_time_stats_and_fused_event_gather(hip, dtype):
    """pp_time_stats (one read-back: descents, min, max) and pp_gather_events (edge_index and time permuted in one pass) against torch."""
    g = torch.Generator().manual_seed(3)
    m = 100_003
    t = torch.randint(-50, 10_000, (m,), generator=g).to(dtype)
    ei = torch.randint(0, 5000, (2, m), generator=g)
    descents, lo, hi = hip.time_stats(t.to(DEV))
    assert descents == int((t[1:] < t[:-1]).sum())
    if dtype == torch.int64:
        assert (lo, hi) == (int(t.min()), int(t.max()))
    perm = hip.argsort(t.to(DEV), (lo, hi) if dtype == torch.int64 else None)
    want = torch.sort(t, stable=True)
    assert torch.equal(perm.cpu(), want.indices)
    got_ei, got_t = hip.gather_events(ei.to(DEV), t.to(DEV), perm)
    assert torch.equal(got_t.cpu(), want.values) and torch.equal(got_ei.cpu(), ei[:, want.indices])
    assert hip.time_stats(want.values.to(DEV))[0] == 0
    assert hip.time_stats(torch.empty(0, dtype=dtype, device=DEV))[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,f,k", [(0, 64, 0), (1000, 64, 300), (70_000, 256, 70_000), (4097, 8, 1)])
def test_halo_fold_matches_the_separate_passes(hip, n, f, k):
    """pp_halo_fold_f32 (partitioned backward: own partial sums + returned halo rows + CSR sums + self-loop term in one pass) against
    index_add_ / addcmul; every combination of the optional addends, in place and out of place."""
    g = torch.Generator().manual_seed(n + f)
    own = torch.randn(n, f, generator=g)
    recv = torch.randn(k, f, generator=g)
    send_idx = torch.randperm(n, generator=g)[:k]
    slot = torch.full((n,), -1, dtype=torch.int32)
    slot[send_idx] = torch.arange(k, dtype=torch.int32)
    extra = torch.randn(n, f, generator=g)
    coef = torch.rand(n, generator=g)
    dpre = torch.randn(n, f, generator=g)
    dev = "cuda"
    for use_recv in (False, True):
        for use_extra in (False, True):
            for use_self in (False, True):
                want = own.clone()
                if use_recv and k:
                    want.index_add_(0, send_idx, recv)
                if use_extra:
                    want += extra
                if use_self:
                    want += coef.unsqueeze(1) * dpre
                got = hip.halo_fold(own.to(dev), recv.to(dev) if use_recv else None, slot.to(dev) if use_recv else None,
                                    extra.to(dev) if use_extra else None, coef.to(dev) if use_self else None, dpre.to(dev) if use_self else None,
                                    inplace=False)
                torch.testing.assert_close(got.cpu(), want, rtol=1e-6, atol=1e-6)
    buf = own.to(dev)
    out = hip.halo_fold(buf, recv.to(dev), slot.to(dev), None, coef.to(dev), dpre.to(dev))
    assert out.data_ptr() == buf.data_ptr()
    torch.testing.assert_close(out.cpu(), own.clone().index_add_(0, send_idx, recv) + coef.unsqueeze(1) * dpre, rtol=1e-6, atol=1e-6)
