"""Order lifts (line-graph transformation) and De Bruijn aggregation, HIP-backed.

Same names, arguments, defaults and return types as the reference module
``pathpyG.algorithms.lift_order`` (src/pathpyG/algorithms/lift_order.py:10-152).
"""
from __future__ import annotations

import torch

from .. import _dispatch, _hip
from ..core.graph import Graph
from ..data import Data

_NODE_ATTR_AGGR = ("src", "dst", "max", "mul", "add")


def aggregate_node_attributes(edge_index: torch.Tensor, node_attribute: torch.Tensor, aggr: str = "src") -> torch.Tensor:
    """One attribute per edge from the attributes of its two end nodes (reference lift_order.py:10-45).

    ``aggr``: "src" | "dst" | "max" | "mul" | "add"; anything else raises ``ValueError``.
    """
    if aggr not in _NODE_ATTR_AGGR:
        raise ValueError(f"Unknown aggregation method {aggr}")
    return _dispatch.edge_attr(edge_index, node_attribute, aggr)


def lift_order_edge_index(edge_index: torch.Tensor, num_nodes: int | None = None) -> torch.Tensor:
    """Line-graph transformation of a source-sorted ``[2, E]`` edge index (reference lift_order.py:48-79).

    Edge ``e = (u, v)`` is connected to every edge leaving ``v``; the result is int64 ``[2, E']``,
    lexicographically sorted (hence again source-sorted).  ``num_nodes`` defaults to ``max index + 1``.
    """
    if num_nodes is None:
        num_nodes = _dispatch.minmax(edge_index)[1] + 1
    return _dispatch.linegraph_lift(edge_index, int(num_nodes))


def lift_order_edge_index_weighted(
    edge_index: torch.Tensor, edge_weight: torch.Tensor, num_nodes: int | None = None, aggr: str = "src"
) -> tuple[torch.Tensor, torch.Tensor]:
    """Line-graph lift plus lifted edge weights (reference lift_order.py:82-106)."""
    if num_nodes is None:
        num_nodes = _dispatch.minmax(edge_index)[1] + 1
    ho_index = lift_order_edge_index(edge_index, num_nodes)
    return ho_index, aggregate_node_attributes(ho_index, edge_weight, aggr)


def aggregate_edge_index(
    edge_index: torch.Tensor, node_sequence: torch.Tensor, edge_weight: torch.Tensor | None = None, aggr: str = "sum"
) -> Graph:
    """De Bruijn graph of an instance-level edge index (reference lift_order.py:109-152).

    Rows of ``node_sequence`` that are equal become one node (nodes are numbered in lexicographic row
    order), parallel edges are merged and their weights reduced with ``aggr`` ("sum", "mean", "min",
    "max").  The returned graph stores ``edge_index``, ``edge_weight``, ``node_sequence`` (unique rows),
    ``inverse_idx`` (node id of every input row) and ``num_nodes``.
    """
    if edge_weight is None:
        edge_weight = _hip.UNIT          # the reference's torch.ones (:130-131) without the vector: a merged weight is its run length
    unique_nodes, inverse_idx = _dispatch.unique_rows(node_sequence)
    return _aggregate_with_known_nodes(edge_index, node_sequence.size(1), node_sequence, unique_nodes, inverse_idx, edge_weight, aggr)


def _aggregate_with_known_nodes(edge_index, order, node_sequence, unique_nodes, inverse_idx, edge_weight, aggr="sum",
                                want_inverse: bool = False, col_block=None):
    """Second half of :func:`aggregate_edge_index` for callers that already know the distinct node rows and the
    row -> node map (``MultiOrderModel`` derives them from the previous layer instead of re-sorting the rows).
    ``col_block``: see :func:`pathpyg_amd._dispatch.successor_blocks` (shorter coalesce keys, same result)."""
    num_nodes = unique_nodes.size(0)
    if order == 1:
        # first order: the entries of the node sequence already are the node ids (reference :135-136)
        remap = _dispatch.plain(node_sequence).reshape(-1)
    else:
        remap = inverse_idx
    merged = _dispatch.coalesce(edge_index, edge_weight, num_nodes, aggr, remap=remap, want_inverse=want_inverse, col_block=col_block)
    merged_index, merged_weight = merged[0], merged[1]
    data = Data(
        edge_index=merged_index,
        num_nodes=num_nodes,
        node_sequence=unique_nodes,
        edge_weight=merged_weight,
        inverse_idx=inverse_idx,
    )
    g = Graph(data, _row_sorted=True)         # coalesce output is (row, col)-sorted: skip graph.py:103's re-sort
    return (g, merged[2]) if want_inverse else g
