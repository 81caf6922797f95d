"""pathpyg_amd — MI355X-native (gfx950) engine behind pathpyG's higher-order-graph API.

Import it where the reference is imported (``import pathpyg_amd as pp``): ``pp.TemporalGraph``,
``pp.PathData``, ``pp.IndexMap``, ``pp.Graph``, ``pp.MultiOrderModel``, ``pp.algorithms``, ``pp.nn.DBGNN``.
"""
__version__ = "0.1.0"

from .core import Graph, IndexMap, PathData, TemporalGraph  # noqa: F401
from .core.multi_order_model import MultiOrderModel  # noqa: F401
from .data import Data  # noqa: F401
from . import algorithms, utils  # noqa: F401,E402


def __getattr__(name):          # ``pp.nn`` / ``pp.io`` (pandas) / ``pp.distributed`` are loaded on demand
    if name in ("nn", "io", "distributed"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
