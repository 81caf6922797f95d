"""How many DISTINCT source rows a 16-row tile (one wave) / a 64-row group (one workgroup) of the order-2 layer kernels gathers on the headline
stream, against the gathers it issues: the fetch a kernel would need if equal rows of a tile were fetched once (compare with the FETCH_SIZE
counter of k_gcn_forward / k_gcn_backward in profiles/*_pmc_traffic.json)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synth_stream  # noqa: E402

m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
if len(sys.argv) > 1:
    m, n = int(sys.argv[1]), int(sys.argv[2])
ei, t = synth_stream(m, n, span, seed=1, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
b = _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, None)
print(b.sizes)
for name, ptr, idx in (("forward (rows gather their predecessors)", b.ho.fwd_ptr, b.ho.fwd_idx), ("backward (rows gather their successors)", b.ho.bwd_ptr, b.ho.bwd_idx)):
    rows = ptr.numel() - 1
    cnt = (ptr[1:] - ptr[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(rows, device=dev), cnt)
    nnz = idx.numel()
    line = f"{name}: rows {rows}, gathers {nnz} ({nnz * 256 / 1e9:.2f} GB of 256-byte rows)"
    for span_rows in (16, 64, 256, 1024):
        key = (row_of // span_rows) * rows + idx.long()
        distinct = torch.unique(key).numel()
        line += f"; distinct per {span_rows} rows {distinct} ({distinct * 256 / 1e9:.2f} GB)"
    # a tile also reads its own 16 rows: how many gathered rows ARE rows of the same tile
    same = int(((idx.long() // 16) == (row_of // 16)).sum())
    print(line + f"; gathers that land in the tile's own rows {same}", flush=True)
