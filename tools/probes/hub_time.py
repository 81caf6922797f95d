"""Time of the fused builder on the 96-node / 2*10^6-event contact stream (same-box A/B of library variants)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
n, m, span = 96, 2_000_000, 2_000_000
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
w = torch.rand(m, generator=g, device=dev) + 0.25
for weight in (None, w):
    for delta in (30, 300):
        for _ in range(2):
            _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, weight)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b = _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, weight)
        torch.cuda.synchronize()
        print(f"{'unit weights' if weight is None else 'float32 weights'}, delta={delta}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  E2={b.sizes['E2']}")
