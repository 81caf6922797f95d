import os, sys, time, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
dev = "cuda:0"
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
for it in range(12):
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=5)
    if it % 3 == 0:
        _ = mom.layers[5].data.node_sequence.shape, mom.layers[3].data.edge_index.shape
    del mom
    torch.cuda.synchronize()
    print(it, round(torch.cuda.memory_allocated() / 2**30, 3), "GiB allocated", gc.get_count(), flush=True)
