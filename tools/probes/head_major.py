"""The order-2 layer of the headline stream under two node numberings: the reference's (lexicographic (source, target): rows (b, .) of one
block consecutive) and HEAD-MAJOR ((target, source): rows (., b) consecutive).  Same graph, same kernels (k_gcn_forward / k_gcn_backward 64 x 64)
— what a head-major order would buy the backward aggregation (all rows (., b) gather from the same ~20 rows (b, .)) and cost the forward one."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
from pathpyg_amd import _hip
dev = torch.device("cuda:0")
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
g2 = mom.layers[2]
e2, w2, ns = g2.data.edge_index, g2.data.edge_weight.float(), g2.data.node_sequence
u2 = g2.n
order = torch.argsort(ns[:, 1] * n + ns[:, 0])               # head-major: new position -> old id
new_id = torch.empty_like(order)
new_id[order] = torch.arange(u2, device=dev)
e2h = new_id[e2]
srt = torch.argsort(e2h[0] * u2 + e2h[1])
e2h, w2h = e2h[:, srt].contiguous(), w2[srt].contiguous()
x = torch.randn(u2, 64, generator=g, device=dev)
dpre = torch.randn(u2, 64, generator=g, device=dev)
wq = torch.randn(64, 64, generator=g, device=dev) / 8
bias = torch.randn(64, generator=g, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, idx, wt in (("reference numbering (source, target)", e2, w2), ("head-major numbering (target, source)", e2h, w2h)):
    plan = _hip.gcn_plan(idx, wt, u2)
    f = timed(lambda: _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, u2, x, plan.self_coef, wq, bias, True))
    b = timed(lambda: _hip.gcn_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, u2, dpre, plan.self_coef, x, wq, True, True))
    print(f"{name}: k_gcn_forward {f:.3f} ms, k_gcn_backward {b:.3f} ms  ({u2} rows, {idx.size(1)} entries)", flush=True)
