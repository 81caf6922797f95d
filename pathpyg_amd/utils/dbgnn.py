"""DBGNN helpers (reference ``pathpyG.utils.dbgnn``)."""
from __future__ import annotations

from typing import Optional

import torch


def generate_bipartite_edge_index(g, g2, mapping: str = "last", device: Optional[torch.device] = None) -> torch.Tensor:
    """``[2, .]`` index linking every second-order node to first-order node(s)
    (reference src/pathpyG/utils/dbgnn.py:10-46).

    "last" -> column 1 of the node sequence (literally column 1, as in the reference, so only meaningful
    for second-order graphs), "first" -> column 0, anything else -> both (first block, then last block).
    Built from two tensor slices instead of the reference's per-node Python lists.
    """
    ns = g2.data.node_sequence
    if device is not None:
        ns = ns.to(device)
    ids = torch.arange(g2.n, device=ns.device)
    if mapping == "last":
        return torch.stack((ids, ns[:, 1]))
    if mapping == "first":
        return torch.stack((ids, ns[:, 0]))
    return torch.stack((torch.cat((ids, ids)), torch.cat((ns[:, 0], ns[:, 1]))))
