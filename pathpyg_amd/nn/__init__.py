from .dbgnn import DBGNN, BipartiteGraphOperator, GCNConv  # noqa: F401
