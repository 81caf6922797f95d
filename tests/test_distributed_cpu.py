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
