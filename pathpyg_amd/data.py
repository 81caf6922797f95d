"""A small attribute bag standing in for ``torch_geometric.data.Data`` (PyG is not a dependency).

Only the behaviour the hot path of pathpyG touches is provided: keyword construction, attribute and
item access, ``in``, ``keys()``, ``num_nodes`` / ``num_edges``, ``to(device)``, the node/edge attribute
heuristics PyG applies (first dimension equals the node / edge count; ``*index*`` keys use the last
dimension) and the two time helpers ``MultiOrderModel.from_temporal_graph`` calls
(reference src/pathpyG/core/multi_order_model.py:148-151).
"""
from __future__ import annotations

from typing import Any, Iterator

import numpy as np
import torch

_OPTIONAL = ("x", "y", "pos", "edge_index", "edge_attr", "edge_weight", "time")


class Lazy:
    """A tensor of a :class:`Data` bag that is made on first access (``make()``); ``shape`` answers the size questions without it.
    Layers that come out of the fused order-2 builder (``pp_debruijn2_*``) hold CSR plans: their ``[2, E]`` edge indices, merged weights,
    node sequences and inverse maps are derived from those only when somebody reads them.  After the first access the bag holds the
    tensor itself; the ``Lazy`` remembers it (``value`` / ``version``) so that objects derived from the unresolved form can be told valid."""

    __slots__ = ("make", "shape", "value", "version")

    def __init__(self, make, shape):
        self.make, self.shape = make, tuple(int(s) for s in shape)
        self.value, self.version = None, -1

    def resolve(self):
        if self.value is None:
            self.value = self.make()
            self.version = self.value._version
            self.make = None
        return self.value

    def mapped(self, fn) -> "Lazy":
        """The same deferred tensor seen through ``fn`` (a device move, a copy): still deferred while this one is, ``fn`` of the tensor once made."""
        if self.value is not None:
            done = Lazy(None, self.shape)
            done.value = fn(self.value)
            done.version = done.value._version
            return done
        return Lazy(lambda: fn(self.resolve()), self.shape)

    def __reduce__(self):
        # a maker is a closure over device plans and does not pickle: the tensor does (torch.save / pickle of a layer's Data or a DBGNN bundle)
        return (_made, (self.resolve(),))


def _made(value) -> "Lazy":
    out = Lazy(None, tuple(value.shape))
    out.value, out.version = value, value._version
    return out


class Data:
    def __init__(self, **attrs: Any) -> None:
        object.__setattr__(self, "_store", {})
        for key, value in attrs.items():
            self[key] = value

    def _resolved(self, key: str) -> Any:
        value = self._store[key]
        if isinstance(value, Lazy):
            value = value.resolve()
            self._store[key] = value
        return value

    def peek(self, key: str) -> Any:
        """The stored object as it is (a :class:`Lazy` stays unresolved); ``None`` when absent."""
        return self._store.get(key)

    # ------------------------------------------------------------ mapping protocol
    def __getitem__(self, key: str) -> Any:
        return self._resolved(key)

    def __setitem__(self, key: str, value: Any) -> None:
        if value is None and key in self._store:
            del self._store[key]
        elif value is not None:
            self._store[key] = value

    def __delitem__(self, key: str) -> None:
        del self._store[key]

    def __contains__(self, key: str) -> bool:
        return key in self._store

    def __getattr__(self, key: str) -> Any:
        store = object.__getattribute__(self, "_store")
        if key in store:
            value = store[key]
            if isinstance(value, Lazy):
                value = store[key] = value.resolve()
            return value
        if key in _OPTIONAL:
            return None
        raise AttributeError(f"'Data' object has no attribute '{key}'")

    def __setattr__(self, key: str, value: Any) -> None:
        if key in type(self).__dict__ and isinstance(type(self).__dict__[key], property):
            type(self).__dict__[key].fset(self, value)
        else:
            self[key] = value

    def __delattr__(self, key: str) -> None:
        del self._store[key]

    def keys(self) -> list[str]:
        return list(self._store.keys())

    def __iter__(self) -> Iterator[tuple[str, Any]]:
        return iter([(key, self._resolved(key)) for key in list(self._store)])

    def __len__(self) -> int:
        return len(self._store)

    def to_dict(self) -> dict:
        return dict(iter(self))

    # ------------------------------------------------------------ sizes
    @property
    def num_nodes(self) -> int | None:
        if "num_nodes" in self._store:
            return self._store["num_nodes"]
        x = self._store.get("x")
        if isinstance(x, torch.Tensor):
            return x.size(0)
        ei = self._resolved("edge_index") if "edge_index" in self._store else None
        if isinstance(ei, torch.Tensor) and ei.numel() > 0:
            return int(ei.max()) + 1
        return None

    @num_nodes.setter
    def num_nodes(self, value: int | None) -> None:
        self["num_nodes"] = value

    @property
    def num_edges(self) -> int:
        ei = self._store.get("edge_index")
        if isinstance(ei, Lazy):
            return ei.shape[-1]
        return int(ei.size(-1)) if isinstance(ei, torch.Tensor) else 0

    # ------------------------------------------------------------ attribute classification (PyG heuristics)
    @staticmethod
    def _cat_dim(key: str) -> int:
        return -1 if ("index" in key or key == "face") else 0

    def _length_along_cat_dim(self, key: str):
        value = self._store[key]
        if isinstance(value, Lazy):
            return (value.shape[self._cat_dim(key)], False) if value.shape else (None, False)
        if isinstance(value, (list, tuple)):
            return len(value), True
        if not isinstance(value, (torch.Tensor, np.ndarray)) or value.ndim == 0:
            return None, False
        return value.shape[self._cat_dim(key)], False

    def is_node_attr(self, key: str) -> bool:
        length, is_seq = self._length_along_cat_dim(key)
        if length is None or length != self.num_nodes:
            return False
        if is_seq or self.num_nodes != self.num_edges:
            return True
        return "edge" not in key

    def is_edge_attr(self, key: str) -> bool:
        length, is_seq = self._length_along_cat_dim(key)
        if length is None or length != self.num_edges:
            return False
        if is_seq or self.num_nodes != self.num_edges:
            return True
        return "edge" in key

    def node_attrs(self) -> list[str]:
        return [k for k in self._store if self.is_node_attr(k)]

    def edge_attrs(self) -> list[str]:
        return [k for k in self._store if self.is_edge_attr(k)]

    # ------------------------------------------------------------ device / time helpers
    def to(self, device) -> "Data":
        # (deferred tensors stay deferred: `bundle.to(dev)` on a fused model must not materialise every [2, A2] index)
        for key, value in list(self._store.items()):
            if isinstance(value, torch.Tensor):
                self._store[key] = value.to(device)
            elif isinstance(value, Lazy):
                self._store[key] = value.mapped(lambda t, device=device: t.to(device))
        return self

    def clone(self) -> "Data":
        out = Data()
        for key, value in list(self._store.items()):
            if isinstance(value, Lazy):
                out[key] = value.mapped(lambda t: t.clone())
            else:
                out[key] = value.clone() if isinstance(value, torch.Tensor) else value
        return out

    def is_sorted_by_time(self) -> bool:
        t = self._resolved("time") if "time" in self._store else None
        if t is None or t.numel() < 2:
            return True
        from . import _dispatch
        return _dispatch.is_sorted(t)

    def sort_by_time(self) -> "Data":
        from . import _dispatch
        t = self._resolved("time")
        perm = _dispatch.stable_argsort(t)
        out = Data()
        for key, value in iter(self):
            if key == "edge_index":
                out[key] = value[:, perm]
            elif key == "time" or self.is_edge_attr(key):
                out[key] = value[perm]
            else:
                out[key] = value
        return out

    def __repr__(self) -> str:
        parts = []
        for key, value in self._store.items():
            if isinstance(value, (torch.Tensor, Lazy)):
                parts.append(f"{key}={list(value.shape)}")
            else:
                parts.append(f"{key}={value!r}")
        return f"Data({', '.join(parts)})"
