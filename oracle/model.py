"""Oracle (test infrastructure): multi-order model construction on CPU tensors.

Restates the tensor work of
  * ``MultiOrderModel.iterate_lift_order``   reference src/pathpyG/core/multi_order_model.py:83-122
  * ``MultiOrderModel.from_temporal_graph``  reference src/pathpyG/core/multi_order_model.py:124-192
  * ``MultiOrderModel.from_path_data``       reference src/pathpyG/core/multi_order_model.py:194-241
  * ``MultiOrderModel.to_dbgnn_data``        reference src/pathpyG/core/multi_order_model.py:511-554
  * ``generate_bipartite_edge_index``        reference src/pathpyG/utils/dbgnn.py:33-44
  * ``PathData.append_walks`` layout         reference src/pathpyG/core/path_data.py:126-159
  * event time sort of ``TemporalGraph``     reference src/pathpyG/core/temporal_graph.py:58-63
    (the reference's ``argsort`` is unstable; this build defines event order by a
    STABLE sort)
Layers are plain dicts of tensors (see oracle.aggregate.aggregate_edge_index);
the Python-side IndexMap bookkeeping of the reference is not restated here.

Not product code: see oracle/__init__.py.
"""
from __future__ import annotations

import torch

from .aggregate import aggregate_edge_index
from .lift import (
    edge_attribute_from_nodes,
    histogram,
    line_graph_lift,
    line_graph_lift_weighted,
    temporal_lift_per_timestamp,
    temporal_lift_sorted,
)


def stable_time_sort(edge_index: torch.Tensor, time: torch.Tensor):
    perm = torch.sort(time, stable=True).indices
    return edge_index[:, perm], time[perm], perm


def extend_node_sequence(node_sequence: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """Order-(k+1) instance sequences: sequence of the edge's source followed by the
    last node of its destination (multi_order_model.py:114,165)."""
    return torch.cat((node_sequence[edge_index[0]], node_sequence[edge_index[1]][:, -1:]), dim=1)


def lift_step(edge_index, node_sequence, edge_weight=None, aggr="src", save=True):
    """One ``iterate_lift_order`` call, tensors only."""
    if edge_weight is None:
        lifted = line_graph_lift(edge_index, node_sequence.size(0))
    else:
        lifted, edge_weight = line_graph_lift_weighted(edge_index, edge_weight, node_sequence.size(0), aggr)
    node_sequence = extend_node_sequence(node_sequence, edge_index)
    layer = aggregate_edge_index(lifted, node_sequence, edge_weight) if save else None
    return lifted, node_sequence, edge_weight, layer


def layers_from_temporal(edge_index, time, num_nodes, delta=1, max_order=1, edge_weight=None,
                         cached=True, event_graph=None, loop_lift=False) -> dict:
    """Layers ``{k: dict}`` of ``from_temporal_graph`` for an already time-sorted event list."""
    layers = {}
    node_sequence = torch.arange(num_nodes).unsqueeze(1)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1))
    if cached or max_order == 1:
        layers[1] = aggregate_edge_index(edge_index, node_sequence, edge_weight)
    if max_order > 1:
        node_sequence = extend_node_sequence(node_sequence, edge_index)
        if event_graph is not None:
            ho = event_graph
        elif loop_lift:
            ho = temporal_lift_per_timestamp(edge_index, time, delta)
        else:
            ho = temporal_lift_sorted(edge_index, time, delta, num_nodes)
        weight = edge_attribute_from_nodes(ho, edge_weight, "src")
        if cached or max_order == 2:
            layers[2] = aggregate_edge_index(ho, node_sequence, weight)
        for k in range(3, max_order + 1):
            keep = cached or k == max_order
            ho, node_sequence, weight, layer = lift_step(ho, node_sequence, weight, "src", keep)
            if keep:
                layers[k] = layer
    return layers


def walks_to_path_tensors(walks: list[list[int]], weights: list[float]) -> dict:
    """Concatenated walk store as ``PathData.append_walks`` lays it out (path_data.py:139-159)."""
    lengths = torch.tensor([len(w) for w in walks], dtype=torch.long)
    flat = torch.tensor([v for w in walks for v in w], dtype=torch.long)
    total = int(lengths.sum())
    pos = torch.arange(total)
    ends = torch.cumsum(lengths, 0) - 1                  # last path-node of each walk
    is_end = torch.zeros(total, dtype=torch.bool)
    is_end[ends] = True
    tails = pos[~is_end]
    return {
        "edge_index": torch.stack((tails, tails + 1)),
        "node_sequence": flat.unsqueeze(1),
        "dag_weight": torch.tensor(weights, dtype=torch.float),
        "dag_num_edges": lengths - 1,
        "dag_num_nodes": lengths,
    }


def layers_from_paths(paths: dict, max_order=1, mode="propagation", cached=True) -> dict:
    """Layers of ``from_path_data`` (multi_order_model.py:211-241)."""
    edge_index = paths["edge_index"]
    node_sequence = paths["node_sequence"]
    weight = paths["dag_weight"].repeat_interleave(paths["dag_num_edges"])
    aggr = "src"
    if mode == "diffusion":
        weight = weight / histogram(edge_index[0], node_sequence.size(0))[edge_index[0]]
        aggr = "mul"
    layers = {1: aggregate_edge_index(edge_index, node_sequence, weight)}
    for k in range(2, max_order + 1):
        keep = cached or k == max_order
        edge_index, node_sequence, weight, layer = lift_step(edge_index, node_sequence, weight, aggr, keep)
        if keep:
            layers[k] = layer
    return layers


def bipartite_edge_index(ho_node_sequence: torch.Tensor, mapping: str = "last") -> torch.Tensor:
    """utils/dbgnn.py:33-44.  "last" literally reads column 1."""
    ids = torch.arange(ho_node_sequence.size(0))
    if mapping == "last":
        return torch.stack((ids, ho_node_sequence[:, 1]))
    if mapping == "first":
        return torch.stack((ids, ho_node_sequence[:, 0]))
    return torch.stack((torch.cat((ids, ids)), torch.cat((ho_node_sequence[:, 0], ho_node_sequence[:, 1]))))


def dbgnn_inputs(layers: dict, max_order=2, mapping="last", x=None, x_h=None) -> dict:
    """``to_dbgnn_data`` bundle (multi_order_model.py:525-554); one-hot features unless given."""
    if max_order not in layers:
        raise ValueError(f"Higher-order graph of order {max_order} not found.")
    g, gk = layers[1], layers[max_order]
    n, n_ho = g["num_nodes"], gk["num_nodes"]
    return {
        "num_nodes": n,
        "num_ho_nodes": n_ho,
        "x": torch.eye(n) if x is None else x,
        "x_h": torch.eye(n_ho) if x_h is None else x_h,
        "edge_index": g["edge_index"],
        "edge_index_higher_order": gk["edge_index"],
        "edge_weights": g["edge_weight"].float(),
        "edge_weights_higher_order": gk["edge_weight"].float(),
        "bipartite_edge_index": bipartite_edge_index(gk["node_sequence"], mapping),
    }
