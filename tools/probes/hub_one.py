"""One hub stream through the fused builder only (for rocprofv3): argv[1] = zipf | contact."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
if sys.argv[1] == "zipf":
    n, m, span, delta = 1_000_000, 20_000_000, 10_000_000, 1_500_000
    src = torch.randint(0, n, (m,), generator=g, device=dev)
    u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
    ei = torch.stack((src, (n * u.pow(6.0)).long().clamp_(max=n - 1)))
else:
    n, m, span, delta = 96, 2_000_000, 2_000_000, 300
    ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
for _ in range(3):
    built = _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, None)
torch.cuda.synchronize()
print(built.sizes)
