"""CPU oracle for the consumers of the temporal event graph — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import this package; the product never does.

``temporal_shortest_paths_reference`` restates ``pathpyG.algorithms.temporal.temporal_shortest_paths``
(src/pathpyG/algorithms/temporal.py:57-107) line by line, including its use of ``scipy.sparse.csgraph.dijkstra`` — so
distances AND scipy's heap-order tie-breaking of predecessors are the reference's.  Pinned by the reference's own known
answer (tests/algorithms/test_temporal.py:20-93; checked in tests/test_oracle_golden.py).

``temporal_shortest_paths_bfs`` is the level-synchronous restatement the HIP kernel follows: identical distances; among the
events that reach a node on a shortest path the LATEST one (largest event id) names the predecessor.  That rule reproduces the
reference's known answer exactly; on random inputs it differs from scipy's Fibonacci-heap pop order in <1 % of the entries
(both are valid shortest-path trees: dist[s, pred[s, v]] + 1 == dist[s, v] up to the reference's source convention).
"""
from __future__ import annotations

import numpy as np
import torch

from . import lift as _lift


def temporal_shortest_paths_reference(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta):
    """(dist [n,n] float64 with inf, pred [n,n] int64) exactly as temporal.py:57-107 computes them (time-sorted events)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra

    m, n = edge_index.size(1), num_nodes
    event_graph = _lift.temporal_lift_sorted(edge_index, time, delta, n)                     # temporal.py:70
    src_edges = torch.stack([edge_index[0] + m, torch.arange(m)])                              # :74-75
    dst_edges = torch.stack([torch.arange(m), edge_index[1] + m + n])                          # :77-78
    full = torch.cat([event_graph, src_edges, dst_edges], dim=1)                               # :81-83
    size = m + 2 * n
    adj = coo_matrix((np.ones(full.size(1)), (full[0].numpy(), full[1].numpy())), shape=(size, size)).tocsr()   # :86-87
    dist, pred = dijkstra(adj, directed=True, indices=np.arange(m, m + n), return_predecessors=True, unweighted=True)  # :92-94
    dist_fo = dist[:, m + n:] - 1                                                              # :97
    np.fill_diagonal(dist_fo, 0)
    pred_fo = pred[:, n + m:]                                                                  # :101
    pred_fo[pred_fo == -9999] = -1
    idx_map = np.concatenate([edge_index[0].numpy(), [-1]])
    pred_fo = idx_map[pred_fo]
    np.fill_diagonal(pred_fo, np.arange(n))
    return dist_fo, pred_fo


def temporal_shortest_paths_bfs(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta):
    """Frontier BFS over the event DAG per source node; predecessor = source node of the latest tight event."""
    m, n = edge_index.size(1), num_nodes
    ho = _lift.temporal_lift_sorted(edge_index, time, delta, n)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(ptr, ho[0].numpy() + 1, 1)
    ptr = np.cumsum(ptr)
    succ = ho[1].numpy()
    src, dst = edge_index[0].numpy(), edge_index[1].numpy()
    dist = np.full((n, n), np.inf)
    pred = np.full((n, n), -1, dtype=np.int64)
    by_src = [np.flatnonzero(src == s) for s in range(n)]
    for s in range(n):
        level = np.full(m, -1, dtype=np.int64)
        frontier = by_src[s]
        level[frontier] = 1
        depth = 1
        best_event = np.full(n, -1, dtype=np.int64)
        while frontier.size:
            heads = dst[frontier]
            fresh = ~np.isfinite(dist[s, heads])
            dist[s, heads[fresh]] = depth
            tight = dist[s, heads] == depth
            np.maximum.at(best_event, heads[tight], frontier[tight])
            nxt = []
            for e in frontier:
                cand = succ[ptr[e]:ptr[e + 1]]
                cand = cand[level[cand] < 0]
                level[cand] = depth + 1
                nxt.append(cand)
            frontier = np.unique(np.concatenate(nxt)) if nxt else np.empty(0, dtype=np.int64)
            depth += 1
        reached = best_event >= 0
        pred[s, reached] = src[best_event[reached]]
        dist[s, s] = 0
        pred[s, s] = s
    return dist, pred
