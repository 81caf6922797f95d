"""world-size-2 ``gloo`` tests (CPU) of the multi-GPU layer: range planner, halo, sharded lift, gradient all-reduce.
The local lift runs on the GPU in production; here the test substitutes the CPU oracle for it (tests may use the
oracle, the product may not) so that the sharding logic and the collectives are what is being exercised."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lift as ol


def test_event_ranges_cover_and_balance():
    from pathpyg_amd.distributed import event_ranges
    assert event_ranges(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert event_ranges(0, 2) == [(0, 0), (0, 0)]
    assert event_ranges(5, 1) == [(0, 5)]
    w = torch.tensor([0, 0, 10, 0, 1, 1, 1, 1, 1, 5], dtype=torch.float32)
    ranges = event_ranges(10, 2, w)
    assert ranges[0][0] == 0 and ranges[-1][1] == 10 and ranges[0][1] == ranges[1][0]
    loads = [float(w[a:b].sum()) for a, b in ranges]
    assert abs(loads[0] - loads[1]) <= 10
    for ws in (2, 3, 7):
        r = event_ranges(101, ws, torch.rand(101))
        assert r[0][0] == 0 and r[-1][1] == 101 and all(r[i][1] == r[i + 1][0] for i in range(ws - 1))


def test_halo_end_matches_definition():
    from pathpyg_amd.distributed import halo_end
    t = torch.tensor([0, 1, 1, 3, 4, 4, 7, 9, 12])
    assert halo_end(t, 3, 2) == 4              # t[2]=1, delta 2 -> events with t <= 3 -> ids < 4
    assert halo_end(t, 3, 3) == 6
    assert halo_end(t, 9, 1) == 9 and halo_end(t, 0, 1) == 0
    assert halo_end(t.double(), 4, 0.5) == 4
    assert halo_end(t, 1, 100) == 9


def _oracle_local_lift(edge_index, time, num_nodes, delta, n_own=None, id_offset=0):
    out = ol.temporal_lift_sorted(edge_index, time, delta, num_nodes)
    if n_own is not None:
        out = out[:, out[0] < n_own]
    return out + id_offset


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seed, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathpyg_amd as pp
        from pathpyg_amd import _dispatch, distributed as pd
        _dispatch.temporal_lift = _oracle_local_lift          # test-only substitution of the GPU kernel
        rng = np.random.default_rng(seed)
        m, n = 3000, 25
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 400, m)))
        g = type("G", (), {})()
        g.data = pp.Data(edge_index=ei, time=t, num_nodes=n)
        want = ol.temporal_lift_sorted(ei, t, 9, n)
        for weights in (None, torch.from_numpy(rng.random(m))):
            local, offset, total = pd.lift_order_temporal_sharded(g, delta=9, weights=weights)
            assert total == want.size(1)
            assert torch.equal(local, want[:, offset: offset + local.size(1)])      # my block of the global result
            full = pd.gather_lifted(local)
            assert torch.equal(full, want)
        # edge-range sharded LINE-GRAPH lift of the (source-sorted) event graph: blocks concatenate to the single-process result
        def oracle_range_lift(edge_index, num_nodes, edge_range=None):
            full_lift = ol.line_graph_lift(edge_index, num_nodes)
            lo_, hi_ = edge_range
            return full_lift[:, (full_lift[0] >= lo_) & (full_lift[0] < hi_)]
        _dispatch.linegraph_lift = oracle_range_lift
        want3 = ol.line_graph_lift(want, m)
        local3, ranges3, total3 = pd.lift_order_edge_index_sharded(want, m)
        assert total3 == want3.size(1)
        lo3, hi3 = ranges3[rank]
        assert torch.equal(local3, want3[:, (want3[0] >= lo3) & (want3[0] < hi3)])
        assert torch.equal(pd.gather_lifted(local3), want3)
        # gradient all-reduce: mean over ranks, one flattened collective
        net = pp.nn.DBGNN(num_classes=2, num_features=(3, 3), hidden_dims=[4, 4, 2])
        for i, p in enumerate(net.parameters()):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        pd.all_reduce_gradients(net)
        for i, p in enumerate(net.parameters()):
            assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * (1 + world) / 2))
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lift_and_gradient_allreduce_gloo(world):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 5, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(results) == {r: "ok" for r in range(world)}


# ------------------------------------------------------------------------------------------------------------------
# Destination-partitioned DBGNN: the sharding, the halo exchange, the rectangular plans and the collectives run for real (gloo);
# the device operations are the torch-CPU stand-ins of tests/cpu_ops.py (injected through the `ops` argument).
def _spawn(target, world, *args, timeout=300):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, results) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    assert dict(results) == {r: "ok" for r in range(world)}


def _check_against_oracle(sharded, shard, net, want_out, want_loss, want_grads, lo, hi):
    import pathpyg_amd.distributed as pd
    out_local = sharded(shard)
    torch.testing.assert_close(out_local.detach(), want_out[lo:hi], rtol=1e-4, atol=1e-5)
    loss = sharded.loss(shard)
    loss.backward()
    pd.all_reduce_gradients(net, average=False)
    total = loss.detach().clone()
    dist.all_reduce(total)
    torch.testing.assert_close(total, want_loss, rtol=1e-5, atol=1e-6)
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad, want_grads[name], rtol=1e-3, atol=1e-5, msg=lambda m: f"{name}: {m}")


def _dbgnn_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathpyg_amd as pp
        from pathpyg_amd import distributed as pd
        from oracle import dbgnn as od
        from tests.cpu_ops import CpuOps
        g = torch.Generator().manual_seed(0)
        n, n_ho, f = 37, 90, 8

        def graph(nn, ee):
            key = torch.unique(torch.randint(0, nn * nn, (ee,), generator=g))
            ei = torch.stack((key // nn, key % nn))
            return ei, torch.randint(1, 4, (ei.size(1),), generator=g).float()
        ei, w = graph(n, 150)
        ei_h, w_h = graph(n_ho, 260)
        ns = torch.randint(0, n, (n_ho, 2), generator=g)
        for mapping in ("last", "both"):
            bip = torch.stack((torch.arange(n_ho), ns[:, 1]))
            if mapping == "both":
                bip = torch.cat((bip, torch.stack((torch.arange(n_ho), ns[:, 0]))), dim=1)
            bundle = dict(num_nodes=n, num_ho_nodes=n_ho, x=torch.randn(n, f, generator=g), x_h=torch.randn(n_ho, f, generator=g),
                          edge_index=ei, edge_weights=w, edge_index_higher_order=ei_h, edge_weights_higher_order=w_h,
                          bipartite_edge_index=bip, y=torch.randint(0, 3, (n,), generator=g))
            for dims in ([12, 10, 6], [12, 9, 10, 6]):                        # two and three GCN layers per stack
                params = od.init_params(3, (f, f), dims, seed=3)
                want = od.loss_and_grads(params, bundle, bundle["y"])
                net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=dims)
                net.load_state_dict(params)
                sharded = pd.ShardedDBGNN(net, ops=CpuOps())
                shard = sharded.prepare(pp.Data(**bundle))
                assert shard.fo.n_own + 0 == shard.x.size(0) - shard.fo.n_halo
                _check_against_oracle(sharded, shard, net, *want, shard.fo.lo, shard.fo.hi)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_destination_partitioned_dbgnn_matches_single_process_oracle(world):
    _spawn(_dbgnn_worker, world)


class _ReducedScatter:
    """gloo has no reduce_scatter_tensor: the same result from an all-reduce, behind the Work interface the native branch waits on."""

    def __init__(self, out, src, group):
        full = src.clone()
        dist.all_reduce(full, group=group)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        rows = full.size(0) // world
        out.copy_(full[rank * rows: (rank + 1) * rows])

    def wait(self):
        return True


def _stream_worker(rank, world, port, results, native=False):
    """The whole north-star split from the event stream: sharded lift -> destination-owner aggregation -> graph shards -> DBGNN."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathpyg_amd as pp
        from pathpyg_amd import distributed as pd
        from oracle import dbgnn as od
        from oracle import model as om
        from tests.cpu_ops import CpuOps
        rng = np.random.default_rng(23)
        for m, n, delta, span in ((2500, 40, 9, 700), (600, 12, 30, 300), (40, 30, 2, 50)):
            ei = torch.from_numpy(rng.integers(0, n, (2, m)))
            t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
            w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32))
            layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)
            n_ho = layers[2]["num_nodes"]
            gen = torch.Generator().manual_seed(4)
            f = 8
            x, x_h = torch.randn(n, f, generator=gen), torch.randn(n_ho, f, generator=gen)
            y = torch.randint(0, 3, (n,), generator=gen)
            params = od.init_params(3, (f, f), [12, 10, 6], seed=5)
            want = od.loss_and_grads(params, om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h), y)
            tg = type("G", (), {})()
            tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n, edge_weight=w)
            comm = pd.Comm()
            if native:                    # the branches an RCCL run takes, over gloo's CPU tensors (see _node_partition_worker)
                comm.native = True
                dist.reduce_scatter_tensor = lambda out, src, group=None, async_op=False: _ReducedScatter(out, src, group)
            shard = pd.build_dbgnn_shard(tg, delta, x, x_h, y, comm, CpuOps())
            sz = pd.global_sizes(shard, comm)
            assert sz["U2"] == n_ho and sz["A2"] == layers[2]["edge_index"].size(1) and \
                sz["E2"] == om.temporal_lift_sorted(ei, t, delta, n).size(1)
            # De Bruijn property of the aligned cuts: every higher-order row is requested by at most one peer
            if shard.ho.n_send:
                assert int(torch.bincount(shard.ho.send_idx).max()) == 1
            assert sum(comm.all_gather_ints([shard.ho.n_own], ei.device)[r][0] for r in range(world)) == n_ho
            net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=[12, 10, 6])
            net.load_state_dict(params)
            sharded = pd.ShardedDBGNN(net, comm, ops=CpuOps())
            _check_against_oracle(sharded, shard, net, *want, shard.fo.lo, shard.fo.hi)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_stream_to_sharded_dbgnn_matches_single_process_oracle(world):
    _spawn(_stream_worker, world)


def zipf_stream(rng, m, n, span, alpha=1.2):
    """Scale-free temporal stream (SURVEY §8d C3 generator): destinations ~ Zipf(alpha) truncated to n nodes, sources uniform."""
    p = 1.0 / np.arange(1, n + 1) ** alpha
    dst = rng.choice(n, size=m, p=p / p.sum())
    ei = torch.from_numpy(np.stack((rng.integers(0, n, m), rng.permutation(n)[dst])))
    return ei, torch.from_numpy(np.sort(rng.integers(0, span, m)))


def _world8_worker(rank, world, port, results):
    """Eight ranks on an ER and on a Zipf stream: the fully sharded build (layer 1 by start-node range, edge-range lift, destination-owner
    aggregation, structural higher-order halo) + the partitioned DBGNN against the single-process oracle; features and labels through ROW
    LOADERS (no rank may ask for the whole matrix); the plan's cuts must balance the estimated work."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import pathpyg_amd as pp
        from pathpyg_amd import distributed as pd
        from oracle import dbgnn as od
        from oracle import model as om
        from tests.cpu_ops import CpuOps
        rng = np.random.default_rng(41)
        for kind, m, n, delta, span in (("er", 4000, 64, 12, 900), ("zipf", 4000, 64, 12, 900), ("er", 30, 9, 3, 40)):
            if kind == "er":
                ei = torch.from_numpy(rng.integers(0, n, (2, m)))
                t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
            else:
                ei, t = zipf_stream(rng, m, n, span)
            w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32)) if kind == "zipf" else None
            layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)
            n_ho = layers[2]["num_nodes"]
            gen = torch.Generator().manual_seed(4)
            f = 8
            x, x_h = torch.randn(n, f, generator=gen), torch.randn(n_ho, f, generator=gen)
            y = torch.randint(0, 3, (n,), generator=gen)
            params = od.init_params(3, (f, f), [12, 10, 6], seed=5)
            want = od.loss_and_grads(params, om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h), y)
            tg = type("G", (), {})()
            tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n, **({} if w is None else {"edge_weight": w}))
            comm = pd.Comm()
            asked = {"x": 0, "x_h": 0}

            def load_x(rows):
                asked["x"] += int(rows.numel())
                return x.index_select(0, rows)

            def load_xh(rows):
                asked["x_h"] += int(rows.numel())
                return x_h.index_select(0, rows)
            shard = pd.build_dbgnn_shard(tg, delta, load_x, load_xh, lambda rows: y.index_select(0, rows), comm, CpuOps())
            sz = pd.global_sizes(shard, comm)
            assert sz["U2"] == n_ho and sz["A2"] == layers[2]["edge_index"].size(1) and sz["E2"] == om.temporal_lift_sorted(ei, t, delta, n).size(1)
            assert asked["x"] == shard.fo.n_src and asked["x_h"] == shard.ho.n_src            # owned + halo rows, nothing else
            assert shard.ho.send_unique and (shard.ho.n_send == 0 or int(torch.bincount(shard.ho.send_idx).max()) == 1)
            # every stage is sharded: the ranks' layer-1 events partition the stream, their lift ranges tile it
            gathered = comm.all_gather_ints([sz["layer1_events_local"], shard.ho.n_own, shard.fo.n_own, sz["E2_local"]], ei.device)
            assert sum(r[0] for r in gathered) == m and sum(r[1] for r in gathered) == n_ho and sum(r[2] for r in gathered) == n
            assert sum(r[3] for r in gathered) == sz["E2"]
            assert sz["ev_cuts"][0] == 0 and sz["ev_cuts"][-1] == m and sz["fo_cuts"][0] == 0 and sz["fo_cuts"][-1] == n
            if m >= 1000:
                # balance: no rank carries more than twice the mean of the quantity its cut is meant to equalise
                assert max(r[3] for r in gathered) <= 2.0 * sz["E2"] / world + 50, [r[3] for r in gathered]
                work = comm.all_gather_ints([ROW_COST_ * shard.ho.n_own + sz["A2_local"]], ei.device)
                assert max(r[0] for r in work) <= 2.0 * sum(r[0] for r in work) / world + 50, work
            net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=[12, 10, 6])
            net.load_state_dict(params)
            sharded = pd.ShardedDBGNN(net, comm, ops=CpuOps())
            _check_against_oracle(sharded, shard, net, *want, shard.fo.lo, shard.fo.hi)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


ROW_COST_ = 2


@pytest.mark.parametrize("dense_fo", [True, False])
@pytest.mark.parametrize("world", [2, 8])
def test_thread_world_ranks_match_oracle(world, dense_fo, monkeypatch):
    """ThreadWorld — R ranks as threads of ONE process (what `bench.py --emulate-ranks` runs on the one GPU): collectives are copies between
    the ranks' tensors, the ranks take turns.  Same numbers as the oracle, a compute time and a collective log per rank.  ``dense_fo``: the
    first-order shard with the dense halo (every foreign node, no discovery round) and with the discovered one."""
    import pathpyg_amd as pp
    from pathpyg_amd import distributed as pd
    monkeypatch.setattr(pd, "FO_DENSE_HALO", dense_fo)
    from oracle import dbgnn as od
    from oracle import model as om
    from tests.cpu_ops import CpuOps
    rng = np.random.default_rng(37)
    m, n, delta, span, f = 3000, 48, 10, 800, 8
    ei, t = zipf_stream(rng, m, n, span)
    w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32))
    layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)
    gen = torch.Generator().manual_seed(4)
    x, x_h = torch.randn(n, f, generator=gen), torch.randn(layers[2]["num_nodes"], f, generator=gen)
    y = torch.randint(0, 3, (n,), generator=gen)
    params = od.init_params(3, (f, f), [12, 10, 6], seed=5)
    want_out, want_loss, want_grads = od.loss_and_grads(params, om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h), y)
    tg = type("G", (), {})()
    tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n, edge_weight=w)

    def body(comm):
        net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=[12, 10, 6])
        net.load_state_dict(params)
        sharded = pd.ShardedDBGNN(net, comm, ops=CpuOps())
        for _ in range(2):
            comm.barrier()
            net.zero_grad()
            shard = pd.build_dbgnn_shard(tg, delta, lambda r: x.index_select(0, r), lambda r: x_h.index_select(0, r), lambda r: y.index_select(0, r),
                                         comm, CpuOps())
            out = sharded(shard)
            loss = sharded.loss(shard)
            loss.backward()
            pd.all_reduce_gradients(net, average=False, comm=comm)
        total = loss.detach().clone().reshape(1)
        comm.all_reduce_(total)
        return {"out": out.detach(), "lo": shard.fo.lo, "hi": shard.fo.hi, "loss": total[0], "grads": {k: p.grad.clone() for k, p in net.named_parameters()},
                "compute_s": comm.compute_s, "events": list(comm.events), "sizes": shard.sizes}

    results = pd.run_thread_world(world, body)
    assert sum(r["hi"] - r["lo"] for r in results) == n
    for r in results:
        torch.testing.assert_close(r["out"], want_out[r["lo"]: r["hi"]], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r["loss"], want_loss, rtol=1e-5, atol=1e-6)
        for name, gr in r["grads"].items():
            torch.testing.assert_close(gr, want_grads[name], rtol=1e-3, atol=1e-5, msg=lambda s_: f"{name}: {s_}")
        assert r["compute_s"] > 0 and any(o for _, _, o in r["events"]) and r["sizes"]["U2"] == layers[2]["num_nodes"]
        if dense_fo:
            assert r["sizes"]["fo_halo"] == n - (r["hi"] - r["lo"])
        else:
            assert r["sizes"]["fo_halo"] <= n - (r["hi"] - r["lo"])


def test_world8_er_and_zipf_streams_match_oracle_and_balance():
    _spawn(_world8_worker, 8, timeout=600)


def _dropout_worker(rank, world, port, results, node_ops=False):
    """Training-mode dropout on the partitioned path: the masks are functions of (seed, tag, GLOBAL row, column), so every world size must
    reproduce the single-process evaluation of the reference forward (dbgnn.py:131-150) with those masks — logits, loss, every gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        import pathpyg_amd as pp
        from pathpyg_amd import distributed as pd
        from pathpyg_amd.nn.sharded import dropout_mask
        from oracle import dbgnn as od
        from oracle import model as om
        from tests.cpu_ops import CpuOps, CpuOpsNode
        rng = np.random.default_rng(31)
        m, n, delta, span, f, p = 1800, 35, 10, 600, 8, 0.4
        if node_ops:          # ~10 events per node: the node-range partition on the node-by-node builder (its stand-in), masks by explicit row ids
            CpuOps = CpuOpsNode
            m, n, delta, span = 700, 70, 25, 900
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
        layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2)
        n_ho = layers[2]["num_nodes"]
        gen = torch.Generator().manual_seed(4)
        x, x_h = torch.randn(n, f, generator=gen), torch.randn(n_ho, f, generator=gen)
        y = torch.randint(0, 3, (n,), generator=gen)
        for dims in ([12, 10, 6], [12, 9, 10, 6]):
            params = od.init_params(3, (f, f), dims, seed=5)
            params = {k: (v + 0.05 if k.endswith(".bias") else v) for k, v in params.items()}
            # ---- reference: the whole graph in one process, masks by global row id
            torch.manual_seed(77)
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).item())
            data = om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h)
            leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            n_gcn = len(dims) - 1

            def run_stack(h, prefix, eidx, ew, tag):
                rows = torch.arange(h.size(0))
                for i in range(n_gcn):
                    h = h * dropout_mask(rows, h.size(1), p, seed, tag + i)
                    h = F.elu(od.gcn_conv(h, eidx, ew, leaves[f"{prefix}.{i}.lin.weight"], leaves[f"{prefix}.{i}.bias"]))
                return h
            hx = run_stack(x, "first_order_layers", data["edge_index"], data["edge_weights"], 0)
            hh = run_stack(x_h, "higher_order_layers", data["edge_index_higher_order"], data["edge_weights_higher_order"], 64)
            hx = hx * dropout_mask(torch.arange(n), hx.size(1), p, seed, 32)
            hh = hh * dropout_mask(torch.arange(n_ho), hh.size(1), p, seed, 96)
            z = F.elu(od.bipartite_op(hh, hx, data["bipartite_edge_index"], n, leaves["bipartite_layer.lin1.weight"], leaves["bipartite_layer.lin1.bias"],
                                      leaves["bipartite_layer.lin2.weight"], leaves["bipartite_layer.lin2.bias"]))
            want_out = (z * dropout_mask(torch.arange(n), z.size(1), p, seed, 128)) @ leaves["lin.weight"].t() + leaves["lin.bias"]
            want_loss = F.cross_entropy(want_out, y)
            want_loss.backward()
            # ---- partitioned
            tg = type("G", (), {})()
            tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n)
            comm = pd.Comm()
            shard = pd.build_dbgnn_shard(tg, delta, x, x_h, y, comm, CpuOps())
            assert not (node_ops and world > 1) or shard.sizes.get("builder") == "fused"
            net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=dims, p_dropout=p)
            net.load_state_dict(params)
            net.train()
            sharded = pd.ShardedDBGNN(net, comm, ops=CpuOps())
            torch.manual_seed(77)
            _check_against_oracle_train(sharded, shard, net, want_out.detach(), want_loss.detach(), {k: v.grad for k, v in leaves.items()})
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _check_against_oracle_train(sharded, shard, net, want_out, want_loss, want_grads):
    import pathpyg_amd.distributed as pd
    loss = sharded.loss(shard)
    loss.backward()
    pd.all_reduce_gradients(net, average=False)
    total = loss.detach().clone()
    dist.all_reduce(total)
    torch.testing.assert_close(total, want_loss, rtol=1e-5, atol=1e-6)
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad, want_grads[name], rtol=1e-3, atol=1e-5, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("world", [1, 2, 3])
def test_partitioned_dropout_is_reproducible_across_world_sizes(world):
    _spawn(_dropout_worker, world)


@pytest.mark.parametrize("world", [2, 3])
def test_node_range_partition_dropout_is_reproducible_across_world_sizes(world):
    _spawn(_dropout_worker, world, True)


def _node_partition_worker(rank, world, port, results, native=False):
    """The NODE-RANGE partition on the node-by-node builder (round 4: pathpyg_amd.distributed._build_partitioned_by_node) under gloo, with the
    torch-CPU stand-in of pp_debruijn2_part_* (tests/cpu_ops.py::CpuOpsNode): rows numbered in send order, halo rows without an id exchange,
    first-order shard with a dense halo — against the single-process oracle; row loaders see exactly the owned rows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import pathpyg_amd as pp
        from pathpyg_amd import distributed as pd
        from oracle import dbgnn as od
        from oracle import model as om
        from tests.cpu_ops import CpuOpsNode
        rng = np.random.default_rng(43)
        for kind, m, n, delta, span, weighted in (("er", 900, 80, 30, 1200, False), ("er-weighted", 700, 60, 25, 900, True), ("tiny", 25, 12, 4, 40, False)):
            ei = torch.from_numpy(rng.integers(0, n, (2, m)))
            t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
            w = torch.from_numpy((rng.random(m) + 0.5).astype(np.float32)) if weighted else None
            layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)
            n_ho = layers[2]["num_nodes"]
            gen = torch.Generator().manual_seed(4)
            f = 8
            x, x_h = torch.randn(n, f, generator=gen), torch.randn(n_ho, f, generator=gen)
            y = torch.randint(0, 3, (n,), generator=gen)
            params = od.init_params(3, (f, f), [12, 10, 6], seed=5)
            want = od.loss_and_grads(params, om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h), y)
            tg = type("G", (), {})()
            tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n, **({} if w is None else {"edge_weight": w}))
            comm = pd.Comm()
            if native:
                # the branches an RCCL run takes — tensors handed to the collectives where they live, all_to_all_single with split
                # lists and out= views, all_gather_into_tensor, the asynchronous forms behind Comm._Pending — on gloo's CPU tensors
                comm.native = True
                dist.reduce_scatter_tensor = lambda out, src, group=None, async_op=False: _ReducedScatter(out, src, group)
            asked = {"x_h": 0}

            def load_xh(rows):
                asked["x_h"] += int(rows.numel())
                return x_h.index_select(0, rows)
            shard = pd.build_dbgnn_shard(tg, delta, x, load_xh, lambda rows: y.index_select(0, rows), comm, CpuOpsNode())
            sz = pd.global_sizes(shard, comm)
            assert shard.sizes.get("builder") == "fused", kind
            assert sz["U2"] == n_ho and sz["A2"] == layers[2]["edge_index"].size(1) and sz["E2"] == om.temporal_lift_sorted(ei, t, delta, n).size(1)
            assert asked["x_h"] == shard.ho.n_own                                          # the loader is asked for the owned rows only: halo rows travel
            assert shard.ho.send_unique and shard.ho.send_idx is None and shard.ho.send_prefix == shard.ho.n_send
            gathered = comm.all_gather_ints([shard.ho.n_own, shard.fo.n_own, sz["E2_local"], shard.ho.n_send, shard.ho.n_halo], ei.device)
            assert sum(r[0] for r in gathered) == n_ho and sum(r[1] for r in gathered) == n and sum(r[2] for r in gathered) == sz["E2"]
            assert sum(r[3] for r in gathered) == sum(r[4] for r in gathered)              # every row sent is somebody's halo row
            ids = shard.ho.local_rows()                                                    # (collective: the halo ids are fetched on first use)
            assert torch.equal(torch.sort(ids[: shard.ho.n_own]).values, torch.arange(shard.ho.lo, shard.ho.hi))
            torch.testing.assert_close(shard.x_h, x_h.index_select(0, ids))               # owned rows in send order + halo rows = the rows the ids name
            net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=[12, 10, 6])
            net.load_state_dict(params)
            sharded = pd.ShardedDBGNN(net, comm, ops=CpuOpsNode())
            _check_against_oracle(sharded, shard, net, *want, shard.fo.lo, shard.fo.hi)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_node_range_partition_matches_single_process_oracle_gloo(world):
    _spawn(_node_partition_worker, world)


def test_edge_range_split_through_the_native_collective_branches():
    """The round-3 split (streams with hub nodes take it at every world size) through ``Comm.native``'s branches, world 3."""
    _spawn(_stream_worker, 3, True)


def test_node_range_partition_through_the_native_collective_branches():
    """The code an RCCL run executes (Comm.native: direct all_to_all_single / all_gather_into_tensor / reduce_scatter_tensor, asynchronous
    handles waited where the _ShardedTrunk schedule needs the rows) — driven over gloo's CPU tensors, world 3, against the oracle."""
    _spawn(_node_partition_worker, 3, True)


def test_node_range_partition_world8_threads_match_oracle():
    import pathpyg_amd as pp
    from pathpyg_amd import distributed as pd
    from oracle import dbgnn as od
    from oracle import model as om
    from tests.cpu_ops import CpuOpsNode
    rng = np.random.default_rng(47)
    m, n, delta, span, f = 2400, 200, 40, 2400, 8
    ei = torch.from_numpy(rng.integers(0, n, (2, m)))
    t = torch.from_numpy(np.sort(rng.integers(0, span, m)))
    layers = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2)
    gen = torch.Generator().manual_seed(4)
    x, x_h = torch.randn(n, f, generator=gen), torch.randn(layers[2]["num_nodes"], f, generator=gen)
    y = torch.randint(0, 3, (n,), generator=gen)
    params = od.init_params(3, (f, f), [12, 10, 6], seed=5)
    want_out, want_loss, want_grads = od.loss_and_grads(params, om.dbgnn_inputs(layers, 2, "last", x=x, x_h=x_h), y)
    tg = type("G", (), {})()
    tg.data = pp.Data(edge_index=ei, time=t, num_nodes=n)

    def body(comm):
        net = pp.nn.DBGNN(num_classes=3, num_features=(f, f), hidden_dims=[12, 10, 6])
        net.load_state_dict(params)
        sharded = pd.ShardedDBGNN(net, comm, ops=CpuOpsNode())
        shard = pd.build_dbgnn_shard(tg, delta, x, x_h, y, comm, CpuOpsNode())
        out = sharded(shard)
        loss = sharded.loss(shard)
        loss.backward()
        pd.all_reduce_gradients(net, average=False, comm=comm)
        total = loss.detach().clone().reshape(1)
        comm.all_reduce_(total)
        return {"out": out.detach(), "lo": shard.fo.lo, "hi": shard.fo.hi, "loss": total[0], "grads": {k: p.grad.clone() for k, p in net.named_parameters()},
                "builder": shard.sizes.get("builder"), "events": list(comm.events)}

    results = pd.run_thread_world(8, body)
    assert sum(r["hi"] - r["lo"] for r in results) == n and all(r["builder"] == "fused" for r in results)
    for r in results:
        torch.testing.assert_close(r["out"], want_out[r["lo"]: r["hi"]], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r["loss"], want_loss, rtol=1e-5, atol=1e-6)
        for name, gr in r["grads"].items():
            torch.testing.assert_close(gr, want_grads[name], rtol=1e-3, atol=1e-5, msg=lambda s_: f"{name}: {s_}")
        assert any(kind == "exchange" and over for kind, _, over in r["events"])           # the halo feature rows travel as a logged (async) exchange


# ------------------------------------------------------------------------------------------------------------------
# Distributed aggregation: range-partitioned keys, one exchange per layer (gloo; the local coalesce is the oracle's).
def _oracle_coalesce(edge_index, weight, num_nodes, reduce="sum", remap=None, want_inverse=False, col_block=None):
    from oracle import aggregate as oa
    assert remap is None and not want_inverse
    return oa.coalesce(edge_index, weight, num_nodes, reduce)


def _agg_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathpyg_amd as pp
        from pathpyg_amd import _dispatch, distributed as pd
        from oracle import model as om
        _dispatch.temporal_lift = _oracle_local_lift
        _dispatch.coalesce = _oracle_coalesce
        rng = np.random.default_rng(17)
        m, n, delta = 4000, 30, 11
        ei = torch.from_numpy(rng.integers(0, n, (2, m)))
        t = torch.from_numpy(np.sort(rng.integers(0, 500, m)))
        w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32))
        g = type("G", (), {})()
        g.data = pp.Data(edge_index=ei, time=t, num_nodes=n)
        want = om.layers_from_temporal(ei, t, n, delta=delta, max_order=2, edge_weight=w)[2]
        part = pd.second_order_layer_sharded(g, delta=delta, edge_weight=w)
        assert torch.equal(part["node_sequence"], want["node_sequence"]) and part["num_nodes"] == want["num_nodes"]
        lo, hi = pd.event_ranges(m, world)[rank]
        assert torch.equal(part["own_event_ids"], want["inverse_idx"][lo:hi])          # global node ids of my events
        # my slice = the reference layer's edges whose row falls into my row range; slices concatenate to the whole layer
        cuts = part["row_cuts"]
        rows = want["edge_index"][0]
        mine = (rows >= cuts[rank]) & (rows < cuts[rank + 1])
        assert torch.equal(part["edge_index"], want["edge_index"][:, mine])
        assert torch.equal(part["edge_weight"], want["edge_weight"][mine])
        sizes = [None] * world
        dist.all_gather_object(sizes, int(part["edge_index"].size(1)))
        assert sum(sizes) == want["edge_index"].size(1)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_second_order_layer_matches_oracle(world):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_agg_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert dict(results) == {r: "ok" for r in range(world)}
