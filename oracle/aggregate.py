"""Oracle (test infrastructure): De Bruijn aggregation and Graph bookkeeping on CPU.

Restates
  * ``aggregate_edge_index``  reference src/pathpyG/algorithms/lift_order.py:109-152
  * ``Graph.__init__``        reference src/pathpyG/core/graph.py:79-119 (row sort, CSR/CSC)
and PyG 2.7.0's ``coalesce`` / ``EdgeIndex.sort_by("row")`` / ``get_csr`` /
``get_csc`` from their documented behaviour.

Not product code: see oracle/__init__.py.
"""
from __future__ import annotations

import torch

from .lift import histogram, zero_led_cumsum


def unique_rows(rows: torch.Tensor):
    """``torch.unique(rows, dim=0, return_inverse=True)`` (lift_order.py:133): rows in
    lexicographic order and, per input row, the rank of its value."""
    return torch.unique(rows, dim=0, return_inverse=True)


def coalesce(edge_index: torch.Tensor, edge_attr: torch.Tensor, num_nodes: int, reduce: str = "sum"):
    """PyG ``coalesce``: order edges by (row, col) with a stable sort, keep one copy of
    each distinct pair and reduce the attributes of the copies.  On CPU PyG reduces
    with ``scatter_add_``-style accumulation in sorted order, i.e. left to right."""
    n_edges = edge_index.size(1)
    key = edge_index[0] * num_nodes + edge_index[1]
    key, perm = torch.sort(key, stable=True)
    edge_index = edge_index[:, perm]
    edge_attr = edge_attr[perm]
    head = torch.ones(n_edges, dtype=torch.bool)
    head[1:] = key[1:] != key[:-1]
    if bool(head.all()):
        return edge_index, edge_attr
    group = torch.cumsum(head.to(torch.long), 0) - 1
    n_out = int(group[-1]) + 1 if n_edges else 0
    merged_index = edge_index[:, head]
    if reduce in ("sum", "add"):
        merged = edge_attr.new_zeros(n_out).index_add_(0, group, edge_attr)
    elif reduce == "mean":
        total = edge_attr.new_zeros(n_out).index_add_(0, group, edge_attr)
        cnt = histogram(group, n_out).clamp_(min=1)
        merged = total / cnt if total.is_floating_point() else torch.div(total, cnt, rounding_mode="floor")
    elif reduce in ("min", "max"):
        merged = edge_attr.new_zeros(n_out).scatter_reduce_(
            0, group, edge_attr, "amin" if reduce == "min" else "amax", include_self=False
        )
    else:
        raise ValueError(f"unknown reduce {reduce}")
    return merged_index, merged


def aggregate_edge_index(edge_index, node_sequence, edge_weight=None, aggr="sum") -> dict:
    """De Bruijn layer of an instance-level (higher-order) edge index (lift_order.py:130-152).

    Returns the tensors the reference stores in ``Graph.data``: ``edge_index``,
    ``edge_weight``, ``node_sequence`` (unique rows), ``inverse_idx``, ``num_nodes``."""
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1))
    uniq, inverse = unique_rows(node_sequence)
    if node_sequence.size(1) == 1:
        mapped = node_sequence.squeeze()[edge_index]
    else:
        mapped = inverse[edge_index]
    merged_index, merged_weight = coalesce(mapped, edge_weight, uniq.size(0), aggr)
    # Graph.__init__ then sorts by row (stable); coalesce output is already (row, col) sorted.
    merged_index, perm = sort_by_row(merged_index)
    return {
        "edge_index": merged_index,
        "edge_weight": merged_weight[perm],
        "node_sequence": uniq,
        "inverse_idx": inverse,
        "num_nodes": uniq.size(0),
    }


def sort_by_row(edge_index: torch.Tensor):
    """``EdgeIndex.sort_by("row")`` as pinned by the reference tests: stable sort on the
    source row only (graph.py:103)."""
    perm = torch.sort(edge_index[0], stable=True).indices
    return edge_index[:, perm], perm


def csr_csc(edge_index: torch.Tensor, num_nodes: int) -> dict:
    """``get_csr`` / ``get_csc`` of a row-sorted EdgeIndex (graph.py:114-115)."""
    row_ptr = zero_led_cumsum(histogram(edge_index[0], num_nodes))
    col = edge_index[1].clone()
    by_col = torch.sort(edge_index[1], stable=True).indices
    col_ptr = zero_led_cumsum(histogram(edge_index[1], num_nodes))
    row = edge_index[0][by_col]
    return {"row_ptr": row_ptr, "col": col, "col_ptr": col_ptr, "row": row, "csc_perm": by_col}
