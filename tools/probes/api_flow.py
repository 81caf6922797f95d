"""Wall time of every step a user of the reference API takes on the headline stream (one GPU, device-resident inputs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402

dev = torch.device("cuda:0")
m, n, span, delta, f = 10_000_000, 500_000, 10_000_000, 1_000_000, 64
g = torch.Generator(device=dev).manual_seed(1)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)


def timed(name, fn, reps=5):
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:<58} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms", flush=True)
    return out


tg = timed("TemporalGraph(Data(unsorted stream))", lambda: pp.TemporalGraph(pp.Data(edge_index=ei.clone(), time=t.clone(), num_nodes=n)))
timed("TemporalGraph(Data(time-sorted stream))", lambda: pp.TemporalGraph(pp.Data(edge_index=tg.data.edge_index.clone(), time=tg.data.time.clone(), num_nodes=n)))
ho = timed("lift_order_temporal", lambda: pp.algorithms.lift_order_temporal(tg, delta))
timed("lift_order_edge_index (k=2 -> 3)", lambda: pp.algorithms.lift_order_edge_index(ho, num_nodes=m))
for k in (1, 2, 3):
    mom = timed(f"MultiOrderModel.from_temporal_graph(max_order={k})", lambda: pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=k), 3)
mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
timed("to_static_graph", lambda: tg.to_static_graph(), 3)
x = torch.randn(n, f, device=dev)
x_h = torch.randn(mom.layers[2].n, f, device=dev)
data = timed("to_dbgnn_data(x, x_h)", lambda: mom.to_dbgnn_data(max_order=2, mapping="last", x=x, x_h=x_h))
y = torch.randint(0, 8, (n,), device=dev)
net = pp.nn.DBGNN(num_classes=8, num_features=(f, f), hidden_dims=[f, f, f]).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def train_step(d):
    opt.zero_grad(set_to_none=True)
    loss = pp.nn.cross_entropy(net(d), y)
    loss.backward()
    opt.step()
    return loss


timed("train step, plans rebuilt (fresh bundle each step)", lambda: train_step(mom.to_dbgnn_data(max_order=2, mapping="last", x=x, x_h=x_h)))
timed("train step, cached plans (same bundle)", lambda: train_step(data))
net.eval()
with torch.no_grad():
    timed("inference forward, cached plans", lambda: net(data))
g1 = mom.layers[1]
timed("Graph: layer-1 edge_to_index (lazy dict) first lookup", lambda: pp.Graph(g1.data.clone()).is_edge(0, 1) if hasattr(g1, "is_edge") else None, 1)
