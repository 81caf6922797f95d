"""BASELINE configs[2] generator, from_temporal_graph(max_order=3) at one delta (kernel table).  usage: config2_k3.py [delta] [K]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
dev = torch.device("cuda:0")
delta = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, m, span = 1_000_000, 20_000_000, 10_000_000
g = torch.Generator(device=dev).manual_seed(3)
src = torch.randint(0, n, (m,), generator=g, device=dev)
u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
dst = (n * u.pow(6.0)).long().clamp_(max=n - 1)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"delta={delta} K={K}: {dt*1e3:.2f} ms {getattr(mom, 'sizes', None)}", flush=True)
