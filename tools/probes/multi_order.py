"""MultiOrderModel.from_temporal_graph up to order 5 on the headline stream (BASELINE configs[4]: k=2..5 lift)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
dev = "cuda:0"
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
for K in (2, 3, 4, 5):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sizes = {k: (v.n, v.m) for k, v in mom.layers.items()}
    how = "level by level (pp_multiorder_*)" if "layers" in getattr(mom, "sizes", {}) else ("fused order-2 builder" if getattr(mom, "_pp_fused", None) is not None else "generic kernels")
    print(f"max_order={K}: {dt*1e3:.1f} ms  {how}  (nodes, edges) per layer: {sizes}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del mom
