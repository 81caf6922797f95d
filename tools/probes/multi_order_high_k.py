import sys, torch, numpy as np
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pathpyg_amd as pp
from pathpyg_amd.core import multi_order_model as mm
DEV="cuda:0"
rng = np.random.default_rng(77)
for case in range(40):
    m = int(rng.integers(100, 30000)); n = int(rng.integers(2, 2000)); span = int(rng.integers(10, 4*m))
    K = int(rng.integers(6, 9))
    ei = torch.from_numpy(rng.integers(0, n, (2, m))).to(DEV); t = torch.from_numpy(rng.integers(0, span, m)).to(DEV)
    delta = max(int(span * rng.uniform(0.2, 1.2) / max(m / n, 1)), 1)
    w = torch.from_numpy(rng.integers(1, 4, m).astype(np.float32)).to(DEV)
    g = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n, edge_weight=w))
    mm.FUSED_BUILDER = False
    try:
        slow = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K)
    except RuntimeError as e:
        if "2^31" in str(e): continue
        raise
    finally:
        mm.FUSED_BUILDER = True
    fast = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K)
    for k in range(1, K + 1):
        for key in ("edge_index", "edge_weight", "node_sequence"):
            assert torch.equal(fast.layers[k].data[key], slow.layers[k].data[key]), (case, k, key)
    print(case, m, n, K, delta, [l.m for l in fast.layers.values()], "layers" in getattr(fast, "sizes", {}), flush=True)
print("ok")
