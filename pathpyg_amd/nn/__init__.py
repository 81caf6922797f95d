from .dbgnn import DBGNN, BipartiteGraphOperator, GCNConv, cross_entropy  # noqa: F401
