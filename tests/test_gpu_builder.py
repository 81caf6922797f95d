"""The fused order-2 De Bruijn builder (pp_debruijn2_count / _fill, csrc/pp_debruijn.hip) against the generic kernels it replaces
(pp_coalesce_* -> pp_temporal_* -> pp_coalesce_* -> pp_gcn_plan x 2, themselves pinned against the oracle and the reference's golden
vectors in test_gpu_kernels.py / test_gpu_api.py): every array of both GCN plans and of the bipartite plan must be IDENTICAL, bit for bit
(int32 indices, fp32 coefficients), and the layer sizes equal — on streams with timestamp ties, self loops, isolated nodes, float64
timestamps, every delta promotion mode of lift_order_temporal (reference algorithms/temporal.py:30,43) and non-integer event weights.
Also: hub nodes (more than 64 in- / out-events: chunk-wise inside the same builder, round 5), the oracle directly (see also
test_gpu_builder_oracle.py), and the 2*10^6-event BASELINE configs[1] size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PLAN_FIELDS = ("fwd_ptr", "fwd_idx", "fwd_val", "bwd_ptr", "bwd_idx", "bwd_val", "self_coef")


def _stream(seed, m, n, span, time_dtype=torch.int64):
    rng = np.random.default_rng(seed)
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    if time_dtype == torch.float64:
        t = torch.from_numpy(np.sort(rng.random(m) * span))
    else:
        t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
    return ei, t


def _build(ei, t, n, delta, weight, fused):
    import pathpyg_amd as pp
    from pathpyg_amd import distributed as ppd
    dev = torch.device("cuda:0")
    data = pp.Data(edge_index=ei.to(dev), time=t.to(dev), num_nodes=n)
    if weight is not None:
        data["edge_weight"] = weight.to(dev)
    g = pp.TemporalGraph(data)
    old = ppd.FUSED_BUILDER
    ppd.FUSED_BUILDER = "always" if fused else False          # ("always": also on contact-shaped streams, where the callers would choose the generic kernels)
    try:
        x = torch.zeros(n, 4, device=dev)
        shard = ppd.build_dbgnn_shard(g, delta, x, lambda num_ho_nodes: torch.zeros(num_ho_nodes, 4, device=dev), None, ppd.Comm()).resolve()
    finally:
        ppd.FUSED_BUILDER = old
    return shard


def _assert_same(a, b, what):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} != {tuple(b.shape)}"
    if not torch.equal(a, b):
        bad = torch.nonzero(a.reshape(-1) != b.reshape(-1)).flatten()
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.numel()} of {a.numel()} entries differ, first at {i}: {a.reshape(-1)[i].item()!r} != {b.reshape(-1)[i].item()!r}")


def _assert_close(a, b, what, rtol):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} != {tuple(b.shape)}"
    torch.testing.assert_close(a, b, rtol=rtol, atol=0, msg=lambda m: f"{what}: {m}")


def _compare(fused, generic, hubs=False, float_rtol=None):
    """``float_rtol``: non-integer event weights on hub nodes — the hub tasks' partial sums are combined in task order, a different
    association of the same fp32 terms than the generic path's left-to-right sums (integer arrays stay bit-identical)."""
    assert fused.sizes.get("builder") == "fused", "the fused builder did not run"
    assert generic.sizes.get("builder") is None
    for k in ("m", "N", "E2", "U2", "A1", "A2"):
        assert fused.sizes[k] == generic.sizes[k], f"sizes[{k}]: {fused.sizes[k]} != {generic.sizes[k]}"
    for name in ("fo", "ho"):
        pf, pg = getattr(fused, name).plan, getattr(generic, name).plan
        assert (pf.n_dst, pf.n_src) == (pg.n_dst, pg.n_src)
        for fld in PLAN_FIELDS:
            if float_rtol is not None and getattr(pf, fld).dtype == torch.float32:
                _assert_close(getattr(pf, fld), getattr(pg, fld), f"{name}.{fld}", float_rtol)
            else:
                _assert_same(getattr(pf, fld), getattr(pg, fld), f"{name}.{fld}")
        if hubs:            # rows beyond 512 entries: both paths must have found the same ones
            for side in ("fwd_heavy", "bwd_heavy"):
                hf, hg = getattr(pf, side), getattr(pg, side)
                assert (hf is None) == (hg is None), f"{name}.{side}"
                if hf is not None:
                    _assert_same(hf.slot, hg.slot, f"{name}.{side}.slot")
        else:
            assert pf.fwd_heavy is None and pf.bwd_heavy is None
    _assert_same(fused.fo.plan.dst_order, generic.fo.plan.dst_order, "fo.dst_order")
    for fld in ("fwd_ptr", "fwd_idx", "bwd_ptr", "bwd_idx", "self_coef"):
        _assert_same(getattr(fused.bip, fld), getattr(generic.bip, fld), f"bip.{fld}")
    _assert_same(fused.indeg, generic.indeg, "indeg")


CASES = [
    # seed, m, n, span, delta, time dtype, weighted
    (1, 4000, 400, 3000, 40, torch.int64, False),            # timestamp ties (span < m), ~10 events per node
    (2, 6000, 300, 100000, 3000, torch.int64, False),        # few ties, 20 events per node
    (3, 2500, 90, 700, 15, torch.int64, True),               # small node set: self loops, parallel events, non-integer weights
    (4, 3000, 5000, 20000, 2500, torch.int64, False),        # most nodes isolated
    (5, 5000, 350, 1000.0, 30.5, torch.float64, False),      # float64 timestamps
    (6, 5000, 350, 1000.0, 17.25, torch.float64, True),
    (7, 1, 3, 10, 5, torch.int64, False),                    # one event
    (8, 40, 30, 5, 1, torch.int64, False),                   # almost everything ties
]


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_fused_builder_equals_generic_path(case):
    seed, m, n, span, delta, tdt, weighted = case
    ei, t = _stream(seed, m, n, span, tdt)
    w = None
    if weighted:
        w = torch.from_numpy(np.random.default_rng(seed + 100).random(m).astype(np.float32) + 0.25)
    _compare(_build(ei, t, n, delta, w, True), _build(ei, t, n, delta, w, False))


@pytest.mark.parametrize("delta", [7, 7.5, torch.tensor(7.5, dtype=torch.float64), torch.tensor(6, dtype=torch.int32), 0, 10 ** 9],
                         ids=["int", "py-float=f32", "f64-tensor", "i32-tensor", "zero", "everything"])
def test_fused_builder_delta_promotion_modes(delta):
    ei, t = _stream(11, 3000, 250, 900)
    _compare(_build(ei, t, 250, delta, None, True), _build(ei, t, 250, delta, None, False))


@pytest.mark.parametrize("shift,scale,delta", [(0, 1, 40), (-(2 ** 62), 1, 40), (2 ** 62, 1, 2 ** 40), (0, 2 ** 33, 40 * 2 ** 33), (-(2 ** 40), 2 ** 22, 2 ** 27),
                                               (5, 715828, 2 ** 31 + 5), (0, 1, -3)],
                         ids=["plain", "at-int64-min", "huge-delta", "wide", "wide-negative", "span-2^31", "negative-delta"])
def test_fused_builder_timestamps_anywhere_on_the_int64_axis(shift, scale, delta):
    """The window test is done on the full 64-bit timestamps: streams near the ends of the int64 axis, wider than 2^31, with delta beyond
    2^31 or negative give the plans of the generic kernels."""
    ei, t = _stream(31, 4000, 300, 3000)
    t = t * scale + shift
    _compare(_build(ei, t, 300, delta, None, True), _build(ei, t, 300, delta, None, False))


def test_fused_builder_against_the_oracle():
    """Edge lists + merged weights of both layers straight against the CPU oracle (not only against the generic kernels)."""
    from oracle import model as om
    ei, t = _stream(21, 5000, 400, 5000)
    n, delta = 400, 300
    shard = _build(ei, t, n, delta, None, True)
    assert shard.sizes.get("builder") == "fused"
    sei, st, _ = om.stable_time_sort(ei, t)
    want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=2)
    for k, gs in ((1, shard.fo), (2, shard.ho)):
        plan = gs.plan
        ptr = plan.bwd_ptr.long().cpu()
        src = torch.repeat_interleave(torch.arange(plan.n_src), ptr[1:] - ptr[:-1])
        got = torch.stack((src, plan.bwd_idx.long().cpu()))
        assert torch.equal(got, want[k]["edge_index"]), f"layer {k} edge_index"
    # merged first-order weights = run lengths; order-2 weights through the normalisation: w = val / (d_src^-1/2 d_dst^-1/2)
    from pathpyg_amd import _hip
    dev = torch.device("cuda:0")
    tg_ei, tg_t = sei.to(dev), st.to(dev)
    built = _hip.debruijn2(tg_ei, tg_t, n, delta, None)
    assert torch.equal(built.fo_weight.cpu(), want[1]["edge_weight"].to(torch.float32))
    e2 = _hip.temporal_lift(tg_ei, tg_t, n, delta).size(1)
    assert built.sizes["E2"] == e2


def _hub_stream(kind, seed=5):
    """Streams with nodes of more than 64 in- / out-events (the builder's hub path, csrc/pp_debruijn.hip "hub nodes")."""
    rng = np.random.default_rng(seed)
    if kind == "in-hub":                 # one node with 100 in-events
        m, n, span = 3000, 400, 5000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[1, :100] = 7
    elif kind == "out-hub":              # one node with 100 out-events
        m, n, span = 3000, 400, 5000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, 200:300] = 9
    elif kind == "both":                 # a node that is both, several hubs, hub -> hub events, self loops on a hub
        m, n, span = 6000, 300, 4000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[1, :400] = 3
        ei[0, 300:900] = 3
        ei[0, 1000:1200] = 11
        ei[1, 1100:1400] = 12
        ei[1, 2000:2100] = 11
    elif kind == "dense":                # few nodes, many events: every node a hub on both sides, long (source, node) and (node, successor) runs
        m, n, span = 8000, 12, 3000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    elif kind == "many-successors":      # a hub with more than 64 distinct successors AND sources: several rounds of 64 runs, chunks of 256 in-events
        m, n, span = 9000, 700, 6000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, :1500] = 5
        ei[1, 1500:3500] = 5
    elif kind == "zipf":                 # scale-free destinations (BASELINE configs[2] generator): many in-hubs of every size
        m, n, span = 40000, 3000, 30000
        ranks = np.arange(1, n + 1, dtype=np.float64) ** -1.2
        dst = rng.choice(n, size=m, p=ranks / ranks.sum())
        ei = torch.from_numpy(np.stack((rng.integers(0, n, m), dst)))
    elif kind == "source-sink":          # a hub with out-events only (no in-event at all) and one with in-events only
        m, n, span = 3000, 400, 5000
        ei = torch.from_numpy(rng.integers(2, n, (2, m)))
        ei[0, :150] = 0                      # node 0: 150 out-events, never a destination
        ei[1, 200:350] = 1                   # node 1: 150 in-events, never a source
    elif kind == "long-run-few-successors":   # an in-run of 1500 instances at a node with ~15 out-events: counted from the out-events' side
        m, n, span = 6000, 400, 4000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, ::4] = 7
        ei[1, ::4] = 9
    elif kind == "few-in-long-out":      # ~70 in-events, 67 000 out-events: the node's tasks get chunks of 8 in-events (hub_chunk)
        m, n, span = 200_000, 3000, 150_000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, ::3] = 7
    elif kind == "some-in-long-out":     # ~700 in-events, 67 000 out-events: chunks of 16 in-events
        m, n, span = 200_000, 300, 150_000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, ::3] = 7
    elif kind == "long-run":             # one node pair carrying a sixth of the stream: an in-run far longer than a chunk
        m, n, span = 6000, 50, 4000
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        ei[0, ::6] = 4
        ei[1, ::6] = 8
    else:
        raise ValueError(kind)
    t = torch.from_numpy(np.sort(rng.integers(0, span, ei.size(1))))
    return ei, t, n


HUB_KINDS = ["in-hub", "out-hub", "both", "dense", "many-successors", "zipf", "long-run", "long-run-few-successors", "few-in-long-out", "some-in-long-out",
             "source-sink"]


@pytest.mark.parametrize("kind", HUB_KINDS)
@pytest.mark.parametrize("delta", [60, 900], ids=["narrow", "wide"])
def test_fused_builder_hub_nodes_equal_generic_path(kind, delta):
    """Nodes with more than 64 in- or out-events stay inside the fused builder (no whole-stream fallback) and every plan array equals the
    generic kernels' — reference algorithms/temporal.py:33-53 and lift_order.py:133-144 know no such limit."""
    from pathpyg_amd import _hip
    ei, t, n = _hub_stream(kind)
    dev = torch.device("cuda:0")
    built = _hip.debruijn2(ei.to(dev), t.to(dev), n, delta, None)
    assert built is not None and built.sizes["hub_nodes"] > 0, "the stream has no hub node / the builder gave up"
    _compare(_build(ei, t, n, delta, None, True), _build(ei, t, n, delta, None, False), hubs=True)


@pytest.mark.parametrize("kind", ["both", "dense", "many-successors", "zipf", "long-run", "long-run-few-successors", "few-in-long-out"])
@pytest.mark.parametrize("integer", [True, False], ids=["integer-weights", "fractional-weights"])
def test_fused_builder_hub_nodes_weighted(kind, integer):
    ei, t, n = _hub_stream(kind, seed=9)
    rng = np.random.default_rng(19)
    m = ei.size(1)
    w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32) if integer else (rng.random(m).astype(np.float32) + 0.25))
    _compare(_build(ei, t, n, 200, w, True), _build(ei, t, n, 200, w, False), hubs=True, float_rtol=None if integer else 1e-5)


@pytest.mark.parametrize("delta", [7, 7.5, torch.tensor(7.5, dtype=torch.float64), torch.tensor(6, dtype=torch.int32), 0, 10 ** 9],
                         ids=["int", "py-float=f32", "f64-tensor", "i32-tensor", "zero", "everything"])
def test_fused_builder_hub_nodes_delta_promotion_modes(delta):
    ei, t, n = _hub_stream("both")
    t = torch.sort(t % 900).values
    _compare(_build(ei, t, n, delta, None, True), _build(ei, t, n, delta, None, False), hubs=True)


@pytest.mark.parametrize("kind", ["both", "dense"])
@pytest.mark.parametrize("shift,scale,delta", [(-(2 ** 62), 1, 40), (2 ** 62, 1, 2 ** 40), (0, 2 ** 33, 40 * 2 ** 33), (0, 1, -3), (0, 1, 0)],
                         ids=["at-int64-min", "huge-delta", "wide", "negative-delta", "zero-delta"])
def test_fused_builder_hub_nodes_timestamps_anywhere_on_the_int64_axis(kind, shift, scale, delta):
    """The hub kernels' window tests (ballot in registers, forward-only windows, the run-at-a-time scan) on the full 64-bit timestamps."""
    ei, t, n = _hub_stream(kind)
    t = t * scale + shift
    _compare(_build(ei, t, n, delta, None, True), _build(ei, t, n, delta, None, False), hubs=True)
    w = torch.from_numpy(np.random.default_rng(29).integers(1, 4, ei.size(1)).astype(np.float32))
    _compare(_build(ei, t, n, delta, w, True), _build(ei, t, n, delta, w, False), hubs=True)


def test_fused_builder_hub_nodes_float64_time_and_rows_beyond_512():
    """float64 timestamps through the bisection window test of the out-hubs; a delta that connects everything makes rows of several
    thousand entries (the chunk tables of the DBGNN row kernels must appear on both paths)."""
    ei, t, n = _hub_stream("dense")
    tf = torch.sort(torch.from_numpy(np.random.default_rng(3).random(ei.size(1)) * 3000.0)).values
    _compare(_build(ei, tf, n, 45.5, None, True), _build(ei, tf, n, 45.5, None, False), hubs=True)
    ei2, t2, n2 = _hub_stream("many-successors")
    fused = _build(ei2, t2, n2, 10 ** 9, None, True)
    assert fused.ho.plan.fwd_heavy is not None or fused.ho.plan.bwd_heavy is not None
    _compare(fused, _build(ei2, t2, n2, 10 ** 9, None, False), hubs=True)


def test_fused_builder_rejects_bad_input():
    from pathpyg_amd import _hip
    dev = torch.device("cuda:0")
    ei = torch.tensor([[0, 1, 5], [1, 2, 0]], device=dev)
    t = torch.tensor([1, 2, 3], device=dev)
    with pytest.raises(IndexError):
        _hip.debruijn2(ei, t, 3, 1, None)
    with pytest.raises(ValueError):
        _hip.debruijn2(torch.tensor([[0, 1, 2], [1, 2, 0]], device=dev), torch.tensor([3, 2, 1], device=dev), 3, 1, None)


def test_fused_builder_config1_size_equals_generic_path():
    """BASELINE configs[1]: 10^5 nodes / 2*10^6 events, every plan array identical to the generic path."""
    gen = torch.Generator().manual_seed(1)
    m, n = 2_000_000, 100_000
    ei = torch.randint(0, n, (2, m), generator=gen)
    t = torch.sort(torch.randint(0, 1_000_000, (m,), generator=gen)).values
    _compare(_build(ei, t, n, 100_000, None, True), _build(ei, t, n, 100_000, None, False))


@pytest.mark.parametrize("kind", ["dense", "many-successors", "zipf"])
def test_fused_builder_hub_nodes_fractional_weights_are_bit_reproducible(kind):
    """Fractional event weights on hub nodes: the task partials are combined in task order and the run-at-a-time kernel's fp32 histogram is fed by
    ONE wave in scan order — two builds of the same stream give the same bits."""
    from pathpyg_amd import _hip
    ei, t, n = _hub_stream(kind, seed=13)
    w = torch.from_numpy(np.random.default_rng(31).random(ei.size(1)).astype(np.float32) + 0.25)
    dev = torch.device("cuda:0")
    builds = [_hip.debruijn2(ei.to(dev), t.to(dev), n, 250, w.to(dev), want_weights=True) for _ in range(3)]
    for other in builds[1:]:
        for name in ("fo", "ho"):
            for fld in PLAN_FIELDS:
                _assert_same(getattr(getattr(builds[0], name), fld), getattr(getattr(other, name), fld), f"{name}.{fld}")
        _assert_same(builds[0].ho_fwd_weight, other.ho_fwd_weight, "ho_fwd_weight")
