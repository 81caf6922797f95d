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


timed("train step, fresh bundle each step (plans handed over)", lambda: train_step(mom.to_dbgnn_data(max_order=2, mapping="last", x=x, x_h=x_h)))
timed("train step, same bundle", lambda: train_step(data))


def api_step():
    model = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
    return train_step(model.to_dbgnn_data(max_order=2, mapping="last", x=x, x_h=x_h))


timed("WHOLE API STEP: from_temporal_graph(2) + to_dbgnn_data + train step", api_step)
import pathpyg_amd.core.multi_order_model as mm  # noqa: E402
mm.FUSED_BUILDER = False
timed("  the same with the generic kernels (FUSED_BUILDER = False)", api_step, 3)
mm.FUSED_BUILDER = True
# the reference's layer tensors are deferred views of the builder's plans: what reading each of them costs (fresh model per read)
for layer, key in ((1, "edge_index"), (2, "edge_index"), (2, "edge_weight"), (2, "node_sequence"), (2, "inverse_idx")):
    def read(layer=layer, key=key):
        model = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
        return model.layers[layer].data[key]
    timed(f"from_temporal_graph(2) + read layers[{layer}].data.{key}", read, 3)
net.eval()
with torch.no_grad():
    timed("inference forward, cached plans", lambda: net(data))
g1 = mom.layers[1]
timed("Graph: layer-1 edge_to_index (lazy dict) first lookup", lambda: pp.Graph(g1.data.clone()).is_edge(0, 1) if hasattr(g1, "is_edge") else None, 1)
