"""BASELINE configs[2]: scale-free temporal stream, 10^6 nodes / 2*10^7 events, MultiOrderModel K = 1..3 (lift + aggregation only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402

dev = torch.device("cuda:0")
n, m, span = 1_000_000, 20_000_000, 10_000_000
g = torch.Generator(device=dev).manual_seed(3)
src = torch.randint(0, n, (m,), generator=g, device=dev)
u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
dst = (n * u.pow(6.0)).long().clamp_(max=n - 1)                     # heavy hubs
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
for delta in (150_000, 1_500_000):
    for k in (1, 2, 3):
        torch.cuda.reset_peak_memory_stats()
        mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        sizes = {o: (l.n, l.m) for o, l in mom.layers.items()}
        print(f"delta={delta} max_order={k}: {ms:8.2f} ms  layers (nodes, edges): {sizes}  level-by-level={'layers' in getattr(mom, 'sizes', {})}  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
        del mom
