"""Oracle (test infrastructure): order lifts on CPU tensors.

Restates, with torch-CPU ops,
  * ``aggregate_node_attributes``      reference src/pathpyG/algorithms/lift_order.py:10-45
  * ``lift_order_edge_index``          reference src/pathpyG/algorithms/lift_order.py:48-79
  * ``lift_order_edge_index_weighted`` reference src/pathpyG/algorithms/lift_order.py:82-106
  * ``lift_order_temporal``            reference src/pathpyG/algorithms/temporal.py:17-54
The PyG helpers the reference calls are restated from their documented
behaviour (torch_geometric 2.7.0, not vendored in the reference tree):
``degree`` = histogram of an index vector, ``cumsum`` = inclusive prefix sum with
a leading zero.

Not product code: see oracle/__init__.py.
"""
from __future__ import annotations

import torch

_EDGE_ATTR_MODES = ("src", "dst", "max", "mul", "add")


def histogram(index: torch.Tensor, size: int) -> torch.Tensor:
    """PyG ``degree(index, num_nodes=size, dtype=long)``: occurrences of each value."""
    out = torch.zeros(size, dtype=torch.long)
    out.scatter_add_(0, index.to(torch.long), torch.ones_like(index, dtype=torch.long))
    return out


def zero_led_cumsum(x: torch.Tensor) -> torch.Tensor:
    """PyG ``cumsum(x)``: ``[0, x0, x0+x1, ...]`` (one longer than ``x``)."""
    out = x.new_zeros(x.numel() + 1)
    torch.cumsum(x, 0, out=out[1:])
    return out


def edge_attribute_from_nodes(edge_index: torch.Tensor, node_attribute: torch.Tensor, aggr: str = "src") -> torch.Tensor:
    """Per-edge attribute built from the attributes of the two endpoints (lift_order.py:33-44)."""
    if aggr not in _EDGE_ATTR_MODES:
        raise ValueError(f"Unknown aggregation method {aggr}")
    at_src = node_attribute[edge_index[0]]
    if aggr == "src":
        return at_src
    at_dst = node_attribute[edge_index[1]]
    if aggr == "dst":
        return at_dst
    if aggr == "max":
        return torch.maximum(at_src, at_dst)
    if aggr == "mul":
        return at_src * at_dst
    return at_src + at_dst


def line_graph_lift(edge_index: torch.Tensor, num_nodes: int | None = None) -> torch.Tensor:
    """Line-graph transformation of a source-sorted ``[2,E]`` edge index (lift_order.py:62-79).

    Edge ``e = (u -> v)`` gets one lifted edge ``(e, f)`` for every edge ``f``
    leaving ``v``; because the input is grouped by source, those ``f`` are the
    positions ``first_out[v] .. first_out[v] + outdeg[v] - 1``.
    """
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1
    tail, head = edge_index[0], edge_index[1]
    n_edges = edge_index.size(1)
    outdeg = histogram(tail, num_nodes)
    first_out = zero_led_cumsum(outdeg)[:-1]          # where v's out-edges start
    fanout = outdeg[head]                             # lifted edges emitted per edge
    block_start = zero_led_cumsum(fanout)             # output offset of each edge's block
    total = int(block_start[-1])
    lifted_src = torch.repeat_interleave(torch.arange(n_edges, dtype=torch.long), fanout)
    rank_in_block = torch.arange(total, dtype=torch.long) - block_start[lifted_src]
    lifted_dst = first_out[head][lifted_src] + rank_in_block
    return torch.stack((lifted_src, lifted_dst))


def line_graph_lift_weighted(edge_index, edge_weight, num_nodes=None, aggr="src"):
    """lift_order.py:99-106: lift, then derive lifted weights from the (k-1)-order edge weights."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1
    lifted = line_graph_lift(edge_index, num_nodes)
    return lifted, edge_attribute_from_nodes(lifted, edge_weight, aggr)


def temporal_lift_per_timestamp(edge_index: torch.Tensor, time: torch.Tensor, delta=1) -> torch.Tensor:
    """Event-graph lift, one unique timestamp at a time (temporal.py:30-53).

    For every distinct timestamp ``t``: the events at ``t`` are sources, the events
    with ``t < t_j <= t + delta`` are candidates, and a pair survives when the
    source's head equals the candidate's tail.  ``delta`` goes through
    ``torch.tensor(delta)`` exactly like the reference, so the threshold and the
    ``<=`` comparison inherit torch's dtype promotion.
    Raises ``RuntimeError`` when no pair exists, like the reference's ``torch.cat([])``.
    """
    delta_t = torch.tensor(delta)
    event_id = torch.arange(edge_index.size(1))
    tails, heads = edge_index[0], edge_index[1]
    blocks = []
    for t in torch.unique(time, sorted=True):
        now = event_id[time == t]
        later = event_id[(time > t) & (time <= t + delta_t)]
        if now.numel() == 0 or later.numel() == 0:
            continue
        pairs = torch.cartesian_prod(now, later)
        if pairs.dim() == 1:            # cartesian_prod of two 1-element vectors is 1-d
            pairs = pairs.view(1, 2)
        keep = heads[pairs[:, 0]] == tails[pairs[:, 1]]
        blocks.append(pairs[keep])
    return torch.cat(blocks, dim=0).t().contiguous()


def temporal_window_bounds(time: torch.Tensor, delta) -> tuple[torch.Tensor, torch.Tensor]:
    """For every event i of a time-sorted stream: ``[g_lo, g_hi)`` = the event-id
    range with ``t_j > t_i`` and ``t_j <= t_i + delta`` evaluated exactly as
    temporal.py:43 does (threshold and comparison in torch's promoted dtype)."""
    delta_t = torch.tensor(delta)
    thr = time + delta_t
    cmp_dtype = torch.result_type(time, thr)
    g_lo = torch.searchsorted(time, time, right=True)
    g_hi = torch.searchsorted(time.to(cmp_dtype), thr.to(cmp_dtype), right=True)
    return g_lo, torch.maximum(g_hi, g_lo)


def temporal_lift_sorted(edge_index: torch.Tensor, time: torch.Tensor, delta=1, num_nodes: int | None = None) -> torch.Tensor:
    """Same result as :func:`temporal_lift_per_timestamp`, O((m+E2) log m).

    Used as the oracle at sizes the per-timestamp loop cannot finish and as the
    "vectorised CPU" baseline of bench.py.  Events are grouped by tail node
    (stable, so ids ascend inside a group); each event looks up, inside the group
    of its head node, the ids falling into its admissible id window.
    Returns an empty ``[2,0]`` tensor when there is no pair.
    """
    m = edge_index.size(1)
    tails, heads = edge_index[0], edge_index[1]
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if m else 0
    g_lo, g_hi = temporal_window_bounds(time, delta)
    order = torch.sort(tails, stable=True).indices          # event ids grouped by tail
    composite = tails[order] * m + order                    # strictly increasing
    lo = torch.searchsorted(composite, heads * m + g_lo)
    hi = torch.searchsorted(composite, heads * m + g_hi)
    count = hi - lo
    start = zero_led_cumsum(count)
    total = int(start[-1])
    src = torch.repeat_interleave(torch.arange(m, dtype=torch.long), count)
    rank = torch.arange(total, dtype=torch.long) - start[src]
    dst = order[lo[src] + rank]
    return torch.stack((src, dst))
