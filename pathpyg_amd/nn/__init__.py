from .dbgnn import DBGNN, BipartiteGraphOperator, GCNConv, cross_entropy  # noqa: F401
from . import optim  # noqa: F401
