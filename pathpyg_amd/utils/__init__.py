from .dbgnn import generate_bipartite_edge_index  # noqa: F401
