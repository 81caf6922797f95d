"""Upper bound for a head-major processing order of the fused backward kernel: the higher-order graph of the headline stream RELABELLED so that
node (a, b) sits at its rank in (b, a) order — sequential self rows AND local gathers in the transposed pass — against the API's (a, b) order."""
import sys
import torch
sys.path.insert(0, ".")
import pathpyg_amd as pp
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
n, m, span, delta, f = 500_000, 10_000_000, 10_000_000, 1_000_000, 64
g = torch.Generator(device=dev).manual_seed(1)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
l2 = mom.layers[2].data
ho = pp._dispatch.plain(l2.edge_index)
w = l2.edge_weight.float()
u2 = int(l2.num_nodes)
seq = l2.node_sequence                      # [U, 2]
key = seq[:, 1] * n + seq[:, 0]              # (b, a)
order = torch.argsort(key)                   # head-major order of the nodes
rank = torch.empty_like(order)
rank[order] = torch.arange(u2, device=dev)


def bench(edge_index, weight, label):
    plan = _hip.gcn_plan(edge_index, weight, u2, row_sorted=True)
    gen = torch.Generator(device=dev).manual_seed(3)
    dpre, x = torch.randn(u2, f, generator=gen, device=dev), torch.randn(u2, f, generator=gen, device=dev)
    wt = torch.randn(f, f, generator=gen, device=dev) * 0.1
    for name, fn in (("backward", lambda: _hip.gcn_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, u2, dpre, plan.self_coef, x, wt, True, True)),
                     ("forward ", lambda: _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, u2, x, plan.self_coef, wt, None, True))):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:28s} {name}: {e0.elapsed_time(e1) / 5:7.3f} ms")


bench(ho, w, "tail-major (API order)")
rel = rank[ho]
merged, mw = _hip.coalesce(rel, w, u2, "sum")          # (row, col)-sorted again
bench(merged, mw, "head-major relabelling")
