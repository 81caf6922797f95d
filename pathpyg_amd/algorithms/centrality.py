"""Centralities that belong to the temporal / path hot path, under the reference's module name
(``pathpyG.algorithms.centrality``): the temporal ones run on the GPU event graph (see :mod:`.temporal`), the path
traversal statistics are single ``torch.unique`` calls.  The networkx-wrapped static-graph centralities of the reference
(centrality.py:327-356) are out of scope."""
from __future__ import annotations

import torch

from .temporal import temporal_betweenness_centrality, temporal_closeness_centrality  # noqa: F401


def path_node_traversals(paths) -> dict:
    """Number of times any path traverses each node (reference centrality.py:52-59): instance counts of the node-sequence
    entries, unweighted exactly like the reference."""
    nodes, counts = torch.unique(paths.data.node_sequence, return_counts=True)
    return {paths.mapping.to_id(int(node)): count.item() for node, count in zip(nodes.cpu(), counts.cpu())}


def path_visitation_probabilities(paths) -> dict:
    """Probability that a randomly chosen node visit falls on each node (reference centrality.py:136-161)."""
    visits = path_node_traversals(paths)
    total = 0.0
    for v in visits:
        total += visits[v]
    return {v: count / total for v, count in visits.items()}


def map_to_nodes(graph, centralities: dict) -> dict:
    """Node index -> node ID for a dictionary of node-level values (reference centrality.py:62-80)."""
    return {graph.mapping.to_id(i): centralities[i] for i in centralities}
