"""Streams with node sequences realised by more than 4096 paths (dense contact streams; bursts in a sparse stream): the level-by-level
builder hands them back (status bit 2) and the generic kernels take the whole build.  Times of both routes + equality of the layers.
Round 6 tried a second attempt instead (the children of such types through one global radix sort of (type, last node) keys, a flat write
kernel, one workgroup per type for the types pass): identical layers, but 14.8 against 5.9 ms (sparse stream of 2 * 10^6 events + three bursts,
K = 4) and 7.9 against 1.1 ms (2 nodes, 1500 events, K = 3) — the wasted first attempt, the sort over the level's child index space and a
few workgroups walking 10^5 .. 10^6 children each cost more than the generic kernels' flat passes; not kept (DESIGN, round 6).
    gpurun -- python tools/probes/multi_order_dense.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import numpy as np
import torch
import pathpyg_amd as pp
from pathpyg_amd.core import multi_order_model as mm

DEV = "cuda"
rng = np.random.default_rng(1)
bad = 0


def mixed(scale):
    # ~3 continuations per event + bursts a -> b -> c -> d of 40 events per hop: a few node sequences with 64000 paths in a sparse stream
    gen = torch.Generator().manual_seed(5)
    n, m = 20_000 * scale, 400_000 * scale
    ei = torch.randint(0, n, (2, m), generator=gen)
    t = torch.randint(0, 4_000_000, (m,), generator=gen)
    bursts_ei, bursts_t = [], []
    for hop, (a, b) in enumerate(((7, 11), (11, 13), (13, 17))):
        bursts_ei.append(torch.tensor([[a] * 40, [b] * 40]))
        bursts_t.append(2_000_000 + 200 * hop + torch.arange(40))
    for with_bursts in (False, True):
        e = torch.cat([ei] + bursts_ei, dim=1) if with_bursts else ei
        tt = torch.cat([t] + bursts_t) if with_bursts else t
        g = pp.TemporalGraph(pp.Data(edge_index=e.to(DEV), time=tt.to(DEV), num_nodes=n))
        ms = {}
        for which in ("generic", "level"):
            mm.FUSED_BUILDER = which == "level"
            try:
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    model = pp.MultiOrderModel.from_temporal_graph(g, delta=600_000, max_order=4)
                    for lay in model.layers.values():
                        lay.data.edge_index, lay.data.edge_weight
                    torch.cuda.synchronize(); ms[which] = round((time.perf_counter() - t0) * 1e3, 2)
            finally:
                mm.FUSED_BUILDER = True
        print(f"mixed x{scale} bursts={with_bursts}: level-by-level={'layers' in getattr(model, 'sizes', {})} edges={[l.m for l in model.layers.values()]} ms={ms}", flush=True)


mixed(1)
mixed(5)
for case, (n, m, span, delta, K) in enumerate([(2, 1500, 600, 60, 3), (3, 3000, 1000, 50, 4), (8, 20000, 5000, 40, 3), (30, 100000, 20000, 30, 3),
                                               (96, 200000, 20000, 40, 3), (5, 600, 300, 30, 5), (1, 300, 300, 20, 4), (500, 400000, 100000, 150, 3)]):
    ei = torch.from_numpy(rng.integers(0, n, (2, m))).to(DEV)
    t = torch.from_numpy(rng.integers(0, span, m)).to(DEV)
    for weighted in (False, True):
        data = pp.Data(edge_index=ei, time=t, num_nodes=n)
        if weighted:
            data["edge_weight"] = torch.from_numpy(rng.integers(1, 5, m).astype(np.float32)).to(DEV)
        g = pp.TemporalGraph(data)
        times = {}
        for which in ("generic", "level"):
            mm.FUSED_BUILDER = which == "level"
            try:
                for _ in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=K)
                    for lay in model.layers.values():            # (the layers' tensors are deferred: make them inside the timed region)
                        lay.data.edge_index, lay.data.edge_weight
                    torch.cuda.synchronize(); times[which] = (time.perf_counter() - t0) * 1e3
            except RuntimeError as err:
                model = None
                times[which] = str(err)[:60]
            finally:
                mm.FUSED_BUILDER = True
            if which == "generic":
                slow = model
            else:
                fast = model
        lbl = fast is not None and "layers" in getattr(fast, "sizes", {})
        ok = None
        if slow is not None and fast is not None:
            ok = all(torch.equal(fast.layers[k].data[key], slow.layers[k].data[key]) for k in fast.layers for key in ("edge_index", "edge_weight", "node_sequence"))
            bad += not ok
        print(f"n={n} m={m} delta={delta} K={K} weighted={weighted}: level-by-level={lbl} equal={ok} sizes={[(l.n, l.m) for l in fast.layers.values()] if fast else None} ms={times}", flush=True)
print("MISMATCHES", bad)
