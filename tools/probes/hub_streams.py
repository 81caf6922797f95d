"""Streams with hub nodes through the fused order-2 builder (round 5) against the generic kernels: BASELINE configs[2]'s scale-free generator
(10^6 nodes / 2*10^7 events), a contact-network shape (96 nodes / 2*10^6 events: every node a hub on both sides) and the headline ER stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402
from pathpyg_amd import distributed as ppd  # noqa: E402

dev = torch.device("cuda:0")


def streams():
    g = torch.Generator(device=dev).manual_seed(3)
    n, m, span = 1_000_000, 20_000_000, 10_000_000
    src = torch.randint(0, n, (m,), generator=g, device=dev)
    u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
    dst = (n * u.pow(6.0)).long().clamp_(max=n - 1)
    t = torch.randint(0, span, (m,), generator=g, device=dev)
    yield "configs[2] scale-free 1e6 nodes / 2e7 events", torch.stack((src, dst)), t, n, (150_000, 1_500_000)
    n, m, span = 96, 2_000_000, 2_000_000
    ei = torch.randint(0, n, (2, m), generator=g, device=dev)
    t = torch.randint(0, span, (m,), generator=g, device=dev)
    yield "contact network 96 nodes / 2e6 events", ei, t, n, (30, 300)
    n, m, span = 500_000, 10_000_000, 10_000_000
    ei = torch.randint(0, n, (2, m), generator=g, device=dev)
    t = torch.randint(0, span, (m,), generator=g, device=dev)
    yield "headline ER 5e5 nodes / 1e7 events", ei, t, n, (1_000_000,)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for name, ei, t, n, deltas in streams():
    tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
    sei, st = tg.data.edge_index, tg.data.time
    for delta in deltas:
        torch.cuda.reset_peak_memory_stats()
        ms_f, built = timed(lambda: _hip.debruijn2(sei, st, n, delta, None))
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        tag = "None (generic fallback)" if built is None else f"fused, sizes {built.sizes}"
        ppd.FUSED_BUILDER = False
        x = torch.zeros(n, 4, device=dev)
        ms_g, shard = timed(lambda: ppd.build_dbgnn_shard(tg, delta, x, lambda num_ho_nodes: torch.zeros(num_ho_nodes, 4, device=dev), None, ppd.Comm()).resolve(), reps=2)
        ppd.FUSED_BUILDER = True
        same = None
        if built is not None:
            same = all(torch.equal(getattr(built.ho, f), getattr(shard.ho.plan, f)) and torch.equal(getattr(built.fo, f), getattr(shard.fo.plan, f))
                       for f in ("fwd_ptr", "fwd_idx", "fwd_val", "bwd_ptr", "bwd_idx", "bwd_val", "self_coef"))
        print(f"{name}, delta={delta}: fused builder {ms_f:8.2f} ms (peak {peak:.1f} GiB), generic kernels {ms_g:8.2f} ms, plans identical: {same}; {tag}", flush=True)
        del built, shard
