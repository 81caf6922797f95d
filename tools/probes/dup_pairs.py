"""A stream in which ONE node pair carries 30 % of all events (high-frequency contact): long duplicate runs in the coalesce."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
dev = "cuda:0"
m, n, span, delta = 2_000_000, 100_000, 2_000_000, 50
g = torch.Generator(device=dev).manual_seed(1)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
hot = torch.rand(m, generator=g, device=dev) < 0.3
ei[0, hot], ei[1, hot] = 0, 1
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"from_temporal_graph with a 30% hot pair: {dt*1e3:.1f} ms; E1={mom.layers[1].data.edge_index.size(1)} max weight {float(mom.layers[1].data.edge_weight.max())}")
