"""BASELINE configs[2]'s generator, max_order = 2: the two routes of from_temporal_graph (fused order-2 builder with its GCN plans / level-by-level builder,
plans made by DBGNN.forward) — time of the model alone and of model + to_dbgnn_data + one DBGNN forward/backward (F = 16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
from pathpyg_amd.core import multi_order_model as mm
dev = torch.device("cuda:0")
n, m, span, delta = 1_000_000, 20_000_000, 10_000_000, 1_500_000
g = torch.Generator(device=dev).manual_seed(3)
src = torch.randint(0, n, (m,), generator=g, device=dev)
dst = (n * torch.rand(m, generator=g, device=dev, dtype=torch.float64).pow(6.0)).long().clamp_(max=n - 1)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
x = torch.randn(n, 16, device=dev)
net = pp.nn.DBGNN(num_classes=3, num_features=(16, 16), hidden_dims=[16, 16, 16]).to(dev)
for name, limit in (("level-by-level builder (multi_order_model.LIFT_ONLY_ORDER2)", True), ("fused order-2 builder (the default)", False)):
    mm.LIFT_ONLY_ORDER2 = limit
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        data = mom.to_dbgnn_data(max_order=2, x=x, x_h=torch.zeros(mom.layers[2].n, 16, device=dev))
        out = net(data)
        out.sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: model {1e3 * (t1 - t0):.2f} ms, + to_dbgnn_data + DBGNN forward/backward {1e3 * (t2 - t1):.2f} ms, total {1e3 * (t2 - t0):.2f} ms", flush=True)
