"""MultiOrderModel.from_temporal_graph at ONE max_order on the headline stream (kernel table of the multi-order lift), or on another ER stream.
usage: multi_order_k.py K [iters [events nodes span delta]]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
if len(sys.argv) > 6:
    m, n, span, delta = (int(float(a)) for a in sys.argv[3:7])
g = torch.Generator(device=dev).manual_seed(0)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sizes = {k: (v.n, v.data.peek("edge_index").shape[1]) for k, v in mom.layers.items()}
    print(f"m={m} max_order={K}: {dt*1e3:.1f} ms  {sizes}  level-by-level={hasattr(mom, 'sizes') and 'layers' in mom.sizes}  "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del mom
