#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own function source.

Run in the build container only (needs /root/reference; the GPU box never runs
this).  ``import pathpyG`` is impossible there (torch_geometric is absent), so
the four tensor kernels of the hot path are loaded from the reference files by
name with ``ast`` and executed against torch 2.10 CPU, with two stand-ins for
the PyG helpers they call (``degree`` = histogram, ``cumsum`` = prefix sum with a
leading zero, both restated from PyG's documentation).
Nothing of the reference's text is written to the repo: only seeded inputs and
the outputs the reference code produced for them.

    python tests/golden/make_golden.py            # rewrites the fixtures
"""
from __future__ import annotations

import ast
import pathlib
import sys

import numpy as np
import torch

REF = pathlib.Path("/root/reference/src/pathpyG")
OUT = pathlib.Path(__file__).resolve().parent


def _pyg_degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.get_default_dtype())
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))


def _pyg_cumsum(x, dim=0):
    out = x.new_zeros(x.size(0) + 1)
    out[1:] = torch.cumsum(x, dim)
    return out


def load_reference_functions() -> dict:
    ns = {"torch": torch, "degree": _pyg_degree, "cumsum": _pyg_cumsum, "tqdm": lambda it: it,
          "TemporalGraph": object, "Graph": object, "Data": object}
    wanted = {
        "algorithms/lift_order.py": ["aggregate_node_attributes", "lift_order_edge_index",
                                     "lift_order_edge_index_weighted"],
        "algorithms/temporal.py": ["lift_order_temporal"],
    }
    for rel, names in wanted.items():
        tree = ast.parse((REF / rel).read_text())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                mod = ast.Module(body=[node], type_ignores=[])
                exec(compile(mod, str(REF / rel), "exec"), ns)  # noqa: S102 - reference code, build container only
    return ns


class _Bag:
    pass


def duck_temporal_graph(edge_index, time):
    g = _Bag()
    g.data = _Bag()
    g.data.edge_index, g.data.time = edge_index, time
    return g


def synth_events(seed, m, n, t_span, float_time=False):
    """Seeded event stream, STABLY time-sorted (event order = position)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, m)
    dst = rng.integers(0, n, m)
    if float_time:
        t = np.round(rng.random(m) * t_span, 1)          # one decimal => many exact boundary hits
    else:
        t = rng.integers(0, t_span, m)
    order = np.argsort(t, kind="stable")
    ei = torch.from_numpy(np.stack([src[order], dst[order]])).long()
    tt = torch.from_numpy(t[order])
    return ei, (tt.double() if float_time else tt.long())


def main() -> int:
    ref = load_reference_functions()
    store = {}

    # --- temporal event-graph lift (temporal.py:17-54) ------------------------------------
    temporal_cases = [
        # name,        seed, m,    n,   span, float_t, delta
        ("int_ties",     11, 400,  12,   40, False, 3),
        ("int_unique",   12, 300,  10, 100000, False, 9000),
        ("int_wide",     13, 600,  40,  200, False, 50),
        ("int_delta0",   14, 200,   8,   30, False, 0),
        ("int_fdelta",   15, 300,  10, 3000000000, False, 150000000.0),   # python float delta -> float32 path
        ("int_f64delta", 16, 300,  10, 3000000000, False, np.float64(150000000.0)),
        ("f64_ties",     17, 400,  12,   20, True, 1.5),
        ("f64_npdelta",  18, 400,  12,   20, True, np.float64(0.3)),
        ("f64_intdelta", 19, 300,  10,   50, True, 2),
        ("int_big",      20, 2500, 60, 1500, False, 40),
    ]
    for name, seed, m, n, span, ft, delta in temporal_cases:
        ei, t = synth_events(seed, m, n, span, ft)
        try:
            out = ref["lift_order_temporal"](duck_temporal_graph(ei, t), delta)
            raised = ""
        except (RuntimeError, ValueError) as exc:      # torch.cat([]) when no pair exists (temporal.py:53)
            out, raised = torch.empty((2, 0), dtype=torch.long), type(exc).__name__
        store[f"temporal/{name}/raised"] = np.asarray(raised)
        store[f"temporal/{name}/edge_index"] = ei.numpy()
        store[f"temporal/{name}/time"] = t.numpy()
        store[f"temporal/{name}/delta"] = np.asarray(delta)
        store[f"temporal/{name}/delta_kind"] = np.asarray(type(delta).__name__)
        store[f"temporal/{name}/num_nodes"] = np.asarray(n)
        store[f"temporal/{name}/out"] = out.numpy()
        print(f"temporal/{name}: m={m} E2={out.shape[1]}")

    # --- line-graph lift (lift_order.py:48-79) + weighted (:82-106) --------------------------
    rng = np.random.default_rng(5)
    for name, e, n, extra in [("small", 50, 9, 0), ("multi", 400, 25, 0), ("isolated", 300, 40, 15),
                              ("hub", 600, 12, 0), ("wide", 3000, 500, 0)]:
        src = np.sort(rng.integers(0, n, e))
        dst = rng.integers(0, n, e)
        if name == "hub":
            dst[rng.random(e) < 0.6] = src[0]
        ei = torch.from_numpy(np.stack([src, dst])).long()
        w = torch.from_numpy(rng.random(e).astype(np.float32) + 0.5)
        nn = n + extra
        out = ref["lift_order_edge_index"](ei, nn)
        store[f"linegraph/{name}/edge_index"] = ei.numpy()
        store[f"linegraph/{name}/num_nodes"] = np.asarray(nn)
        store[f"linegraph/{name}/edge_weight"] = w.numpy()
        store[f"linegraph/{name}/out"] = out.numpy()
        for aggr in ("src", "dst", "max", "mul", "add"):
            ho, hw = ref["lift_order_edge_index_weighted"](ei, w, nn, aggr)
            assert torch.equal(ho, out)
            store[f"linegraph/{name}/w_{aggr}"] = hw.numpy()
        print(f"linegraph/{name}: E={e} E'={out.shape[1]}")
    # num_nodes=None branch
    ei = torch.tensor([[0, 0, 1, 3, 3], [1, 3, 3, 0, 1]])
    store["linegraph/infer/edge_index"] = ei.numpy()
    store["linegraph/infer/out"] = ref["lift_order_edge_index"](ei).numpy()

    # --- chained lifts k=2..5 of an event graph (input of iterate_lift_order) ----------------
    ei, t = synth_events(31, 500, 15, 300, False)
    ho = ref["lift_order_temporal"](duck_temporal_graph(ei, t), 12)
    store["chain/edge_index"] = ei.numpy()
    store["chain/time"] = t.numpy()
    store["chain/delta"] = np.asarray(12)
    store["chain/num_nodes"] = np.asarray(15)
    store["chain/k2"] = ho.numpy()
    n_inst = ei.shape[1]
    for k in (3, 4, 5):
        nxt = ref["lift_order_edge_index"](ho, n_inst)
        n_inst = ho.shape[1]
        ho = nxt
        store[f"chain/k{k}"] = ho.numpy()
        print(f"chain k={k}: E={ho.shape[1]}")

    np.savez_compressed(OUT / "reference_vectors.npz", **store)
    print("wrote", OUT / "reference_vectors.npz", f"{(OUT / 'reference_vectors.npz').stat().st_size / 1024:.0f} KiB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
