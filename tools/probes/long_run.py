"""One node pair carrying a third of a 2*10^6-event stream (an in-run of ~6.7*10^5 instances: ONE hub task walks it): fused builder against the
generic kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402
from pathpyg_amd import distributed as ppd  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
n, m, span, delta = 100_000, 2_000_000, 2_000_000, 200_000
for share in (3, 30, 300):
    ei = torch.randint(0, n, (2, m), generator=g, device=dev)
    ei[0, ::share] = 7
    ei[1, ::share] = 9
    t = torch.randint(0, span, (m,), generator=g, device=dev)
    tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    ms_f, built = timed(lambda: _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, None))
    ppd.FUSED_BUILDER = False
    x = torch.zeros(n, 4, device=dev)
    ms_g, shard = timed(lambda: ppd.build_dbgnn_shard(tg, delta, x, lambda num_ho_nodes: torch.zeros(num_ho_nodes, 4, device=dev), None, ppd.Comm()).resolve(), 2)
    ppd.FUSED_BUILDER = True
    same = all(torch.equal(getattr(built.ho, f), getattr(shard.ho.plan, f)) for f in ("fwd_ptr", "fwd_idx", "fwd_val", "bwd_ptr", "bwd_idx", "bwd_val"))
    print(f"pair (7, 9) carries 1/{share} of {m} events: fused builder {ms_f:.2f} ms, generic kernels {ms_g:.2f} ms, plans identical {same}; {built.sizes}", flush=True)
