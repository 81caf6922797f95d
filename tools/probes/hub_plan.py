import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
from pathpyg_amd import _hip
dev = "cuda:0"
m, n, span, delta = 2_000_000, 100_000, 2_000_000, 20_000
g = torch.Generator(device=dev).manual_seed(1)
src = torch.randint(0, n, (m,), generator=g, device=dev)
u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
dst = (n * u.pow(6.0)).long().clamp_(max=n - 1)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
gl = mom.layers[1]
for _ in range(5):
    _hip.gcn_plan(gl.data.edge_index, gl.data.edge_weight, gl.n)
torch.cuda.synchronize()
