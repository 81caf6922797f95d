"""from_path_data at scale: 10^6 walks of length 5 over 10^5 nodes (host-side assembly vs device work)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pathpyg_amd as pp
n, walks, length = 5_000, 1_000_000, 5
rng = np.random.default_rng(0)
seqs = rng.integers(0, n, (walks, length))
t0 = time.perf_counter()
paths = pp.PathData(pp.IndexMap(list(range(n))))
paths.append_walks([tuple(r) for r in seqs[:50_000].tolist()], weights=[1.0] * 50_000)
t1 = time.perf_counter()
print(f"append_walks of 50k python tuples: {t1 - t0:.2f} s")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = pp.MultiOrderModel.from_path_data(paths.to("cuda:0") if hasattr(paths, "to") else paths, max_order=3)
    torch.cuda.synchronize(); print(f"from_path_data(max_order=3) on 50k walks: {(time.perf_counter() - t0)*1e3:.1f} ms", {k: v.data.edge_index.size(1) for k, v in m.layers.items()})
