"""from_temporal_graph(max_order = K) on stream shapes beside the headline: a contact network (every node a hub), a stream with timestamp ties,
weighted events.  Prints the time, the layer sizes and whether the level-by-level builder (pp_multiorder_*) took the stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def run(name, ei, t, n, delta, K, w=None):
    d = pp.Data(edge_index=ei, time=t, num_nodes=n)
    if w is not None:
        d["edge_weight"] = w
    tg = pp.TemporalGraph(d)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sizes = {k: (v.n, v.m) for k, v in mom.layers.items()}
    print(f"{name} delta={delta} K={K}: {dt*1e3:.2f} ms {sizes} level-by-level={'layers' in getattr(mom, 'sizes', {})}", flush=True)


n, m, span = 96, 2_000_000, 2_000_000
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
for delta in (30, 300):
    run("contact 96 nodes / 2e6 events", ei, t, n, delta, 3)
n, m, span = 5_000, 2_000_000, 200_000
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
run("ties 5e3 nodes / 2e6 events / 2e5 stamps", ei, t, n, 300, 4)
run("ties, weighted", ei, t, n, 300, 4, torch.randint(1, 4, (m,), generator=g, device=dev).float())
