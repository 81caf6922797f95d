from .graph import Graph  # noqa: F401
from .index_map import IndexMap  # noqa: F401
from .path_data import PathData  # noqa: F401
from .temporal_graph import TemporalGraph  # noqa: F401


def __getattr__(name):          # MultiOrderModel depends on ..algorithms, which depends on .graph
    if name == "MultiOrderModel":
        from .multi_order_model import MultiOrderModel
        return MultiOrderModel
    raise AttributeError(name)
