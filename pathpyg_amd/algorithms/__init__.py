"""Algorithms on the hot path of pathpyG: order lifts and De Bruijn aggregation."""
from .lift_order import (  # noqa: F401
    aggregate_edge_index,
    aggregate_node_attributes,
    lift_order_edge_index,
    lift_order_edge_index_weighted,
)
from .temporal import (  # noqa: F401
    lift_order_temporal,
    temporal_betweenness_centrality,
    temporal_closeness_centrality,
    temporal_shortest_paths,
)
