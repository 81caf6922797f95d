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


def temporal_betweenness_reference(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta) -> np.ndarray:
    """Temporal betweenness per node index, following centrality.py:164-297 statement by statement (Brandes on the event DAG with
    a virtual source per first-order node; queue / stack / predecessor sets as in the reference)."""
    from collections import defaultdict, deque
    from math import isnan

    m, n = edge_index.size(1), num_nodes
    ho = _lift.temporal_lift_sorted(edge_index, time, delta, n)
    src_edges = torch.stack([edge_index[0] + m, torch.arange(m)])                              # :204-209
    full = torch.cat([ho, src_edges], dim=1)
    order = torch.argsort(full[0], stable=True)                                                # Graph.from_edge_index row sort
    full = full[:, order]
    ptr = np.zeros(m + n + 1, dtype=np.int64)
    np.add.at(ptr, full[0].numpy() + 1, 1)
    ptr = np.cumsum(ptr)
    col = full[1].numpy()
    src_indices = torch.unique(edge_index[0] + m).tolist()                                     # :210
    e_dst = edge_index[1].numpy()
    fo = lambda v: int(e_dst[v]) if v < m else v - m                                           # :216-221
    bw = defaultdict(float)
    for s in src_indices:                                                                      # :226
        delta_, sigma, sigma_fo = defaultdict(float), defaultdict(float), defaultdict(float)
        sigma[s] = 1.0
        sigma_fo[fo(s)] = 1.0
        dist, dist_fo = defaultdict(lambda: -1), defaultdict(lambda: -1)
        dist[s] = 0
        dist_fo[fo(s)] = 0
        P = defaultdict(set)
        Q = deque([s])
        S = []
        while Q:                                                                               # :254
            v = Q.popleft()
            for w in col[ptr[v]:ptr[v + 1]].tolist():
                if dist[w] == -1:
                    dist[w] = dist[v] + 1
                    if dist_fo[fo(w)] == -1:
                        dist_fo[fo(w)] = dist[v] + 1
                    S.append(w)
                    Q.append(w)
                if dist[w] == dist[v] + 1:
                    sigma[w] += sigma[v]
                    P[w].add(v)
                    if dist[w] == dist_fo[fo(w)]:
                        sigma_fo[fo(w)] += sigma[v]
        c = 0.0
        for i in dist_fo:                                                                      # :274-278
            if dist_fo[i] >= 0:
                c += 1.0
        bw[fo(s)] = bw[fo(s)] - c + 1.0
        while S:                                                                               # :280
            w = S.pop()
            if dist[w] == dist_fo[fo(w)]:
                x = sigma[w] / sigma_fo[fo(w)]
                if isnan(x):
                    x = 0.0
                delta_[w] += x
            for v in P[w]:
                x = sigma[v] / sigma[w]
                if isnan(x):
                    x = 0.0
                delta_[v] += x * delta_[w]
                bw[fo(v)] += delta_[w] * x
    out = np.zeros(n)
    for k, val in bw.items():
        out[k] = val
    return out


def temporal_betweenness_levels(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta) -> np.ndarray:
    """The same quantity in the level-synchronous form the HIP kernel uses: path counts level by level, dependencies pulled from
    the next level, per-node sums over the in-events in event order.  Equal to the reference up to float64 summation order."""
    m, n = edge_index.size(1), num_nodes
    ho = _lift.temporal_lift_sorted(edge_index, time, delta, n)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(ptr, ho[0].numpy() + 1, 1)
    ptr = np.cumsum(ptr)
    succ = ho[1].numpy()
    src, dst = edge_index[0].numpy(), edge_index[1].numpy()
    bw = np.zeros(n)
    for s in np.unique(src):
        level = np.full(m, -1, dtype=np.int64)
        sigma = np.zeros(m)
        dist_fo = np.full(n, -1, dtype=np.int64)
        sigma_fo = np.zeros(n)
        dist_fo[s], sigma_fo[s] = 0, 1.0
        frontier = np.flatnonzero(src == s)
        level[frontier] = 1
        sigma[frontier] = 1.0
        levels = []
        depth = 1
        while frontier.size:
            levels.append(frontier)
            heads = dst[frontier]
            fresh = dist_fo[heads] < 0
            dist_fo[heads[fresh]] = depth
            tight = dist_fo[heads] == depth
            np.add.at(sigma_fo, heads[tight], sigma[frontier[tight]])
            nxt = []
            for v in frontier:
                cand = succ[ptr[v]:ptr[v + 1]]
                new = cand[level[cand] < 0]
                level[new] = depth + 1
                nxt.append(new)
                hit = cand[level[cand] == depth + 1]
                np.add.at(sigma, hit, sigma[v])
            frontier = np.unique(np.concatenate(nxt)) if nxt else np.empty(0, dtype=np.int64)
            depth += 1
        dep = np.zeros(m)
        credit = np.zeros(m)
        for d in range(len(levels), 0, -1):
            for v in levels[d - 1]:
                acc = 0.0
                cr = 0.0
                for w in succ[ptr[v]:ptr[v + 1]]:
                    if level[w] == d + 1:
                        x = sigma[v] / sigma[w]
                        acc += x * dep[w]
                        cr += dep[w] * x
                if level[v] == dist_fo[dst[v]]:
                    acc += sigma[v] / sigma_fo[dst[v]]
                dep[v] = acc
                credit[v] = cr
        row = np.zeros(n)
        np.add.at(row, dst, credit)                                          # event order within every node
        row[s] += dep[levels[0]].sum() - float((dist_fo >= 0).sum()) + 1.0 if levels else 0.0
        bw += row
    return bw
