"""Concatenated walk store, API-compatible with ``pathpyG.core.path_data.PathData``
(reference src/pathpyG/core/path_data.py:10-204).

All walks live in ONE graph: ``edge_index`` chains consecutive path-node positions of each walk,
``node_sequence[p, 0]`` is the first-order node of position ``p``, and ``dag_weight`` / ``dag_num_edges`` /
``dag_num_nodes`` hold one entry per walk.  This is host-side bookkeeping (concatenations); the lifts
that consume it run on the GPU.
"""
from __future__ import annotations

import torch

from ..data import Data
from .index_map import IndexMap


class PathData:
    def __init__(self, mapping: IndexMap | None = None, device: torch.device | None = None) -> None:
        self.mapping = mapping if mapping else IndexMap()
        self.data = Data(
            edge_index=torch.empty((2, 0), dtype=torch.long, device=device),
            node_sequence=torch.empty((0, 1), dtype=torch.long, device=device),
            dag_weight=torch.empty(0, dtype=torch.float, device=device),
            dag_num_edges=torch.empty(0, dtype=torch.long, device=device),
            dag_num_nodes=torch.empty(0, dtype=torch.long, device=device),
        )
        self.data.num_nodes = 0

    @property
    def num_paths(self) -> int:
        return int(self.data.dag_num_edges.numel())

    @property
    def _device(self):
        return self.data.edge_index.device

    def _append(self, chain: torch.Tensor, nodes: torch.Tensor, weights: torch.Tensor, lengths: torch.Tensor) -> None:
        """``chain``: [2, e] positions relative to the appended block; ``lengths``: nodes per new walk."""
        d = self.data
        d.edge_index = torch.cat((d.edge_index, chain + d.num_nodes), dim=1)
        d.node_sequence = torch.cat((d.node_sequence, nodes))
        d.dag_weight = torch.cat((d.dag_weight, weights.to(d.dag_weight.dtype)))
        d.dag_num_edges = torch.cat((d.dag_num_edges, lengths - 1))
        d.dag_num_nodes = torch.cat((d.dag_num_nodes, lengths))
        d.num_nodes = d.num_nodes + int(lengths.sum())

    def to(self, device) -> "PathData":
        self.data = self.data.to(device)
        return self

    def append_walk(self, node_seq: list | tuple, weight: float = 1.0) -> None:
        """Add one observed walk given as node IDs (reference path_data.py:100-124)."""
        self.append_walks([node_seq], [weight])

    def append_walks(self, node_seqs: list | tuple, weights: list | tuple) -> None:
        """Add several walks at once (reference path_data.py:126-159)."""
        dev = self._device
        nodes = torch.cat([self.mapping.to_idxs(seq, device=dev).reshape(-1) for seq in node_seqs]).unsqueeze(1)
        lengths = torch.tensor([len(seq) for seq in node_seqs], device=dev, dtype=torch.long)
        pos = torch.arange(int(lengths.sum()), device=dev)
        last_of_walk = torch.zeros(pos.numel(), dtype=torch.bool, device=dev)
        last_of_walk[torch.cumsum(lengths, 0) - 1] = True
        tails = pos[~last_of_walk]                          # every position except a walk's last one starts an edge
        self._append(torch.stack((tails, tails + 1)), nodes, torch.tensor(weights, device=dev, dtype=torch.float), lengths)

    def get_walk(self, i: int) -> tuple:
        start = int(self.data.dag_num_nodes[:i].sum())
        end = start + int(self.data.dag_num_nodes[i])
        return tuple(self.mapping.to_ids(self.data.node_sequence[start:end, 0].cpu()).tolist())

    def map_node_seq(self, node_seq: list | tuple) -> list:
        return self.mapping.to_ids(node_seq).tolist()

    def __str__(self) -> str:
        return f"PathData with {self.num_paths} paths with total weight {self.data.dag_weight.sum().item()}"
