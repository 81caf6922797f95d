"""Node-ID <-> index bookkeeping (host side), API-compatible with ``pathpyG.core.index_map.IndexMap``
(reference src/pathpyG/core/index_map.py:13-394).

Two things differ from the reference, both to keep O(U) Python loops off the lift path:
the ID -> index dictionary is built on first use, and a higher-order map can be created *lazily* from a
``[U, k]`` node-sequence tensor (``IndexMap.from_node_sequence``) — the tuple IDs are produced by one
vectorised NumPy gather when somebody actually asks for them.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch


def to_numpy(values) -> np.ndarray:
    """List / tuple / tensor / array -> NumPy array (reference src/pathpyG/utils/convert.py:18-34)."""
    if isinstance(values, np.ndarray):
        return values
    if isinstance(values, torch.Tensor):
        return values.detach().cpu().numpy()
    if isinstance(values, (list, tuple)) and len(values) and isinstance(values[0], torch.Tensor):
        return np.asarray([v.detach().cpu().numpy() for v in values])
    return np.asarray(values)


class IndexMap:
    """Maps node indices to IDs (strings, ints or — for higher-order nodes — tuples) and back."""

    def __init__(self, node_ids=None) -> None:
        self._ids: Optional[np.ndarray] = None
        self._lookup: Optional[dict] = None
        self._pending: Optional[Callable[[], np.ndarray]] = None
        self.id_shape: tuple = (-1,)
        if node_ids is not None:
            self.add_ids(node_ids)

    # -------------------------------------------------------------- lazy construction
    @classmethod
    def from_node_sequence(cls, base: "IndexMap", node_sequence: torch.Tensor) -> "IndexMap":
        """Higher-order map whose ID of node ``r`` is the tuple of ``base`` IDs of ``node_sequence[r]``
        (what reference multi_order_model.py:119,177-179 builds with a Python loop)."""
        out = cls()
        k = int(node_sequence.shape[1])
        out.id_shape = (-1, k)

        def materialise() -> np.ndarray:
            ns = node_sequence.resolve() if hasattr(node_sequence, "resolve") else node_sequence       # (a data.Lazy of a builder-made layer)
            idx = ns.detach().cpu().numpy()
            return base.node_ids[idx] if base.has_ids else idx

        out._pending = materialise
        return out

    def _materialise(self) -> None:
        if self._pending is not None:
            pending, self._pending = self._pending, None
            self._ids = pending()

    @property
    def node_ids(self) -> Optional[np.ndarray]:
        self._materialise()
        return self._ids

    @node_ids.setter
    def node_ids(self, value) -> None:
        self._pending = None
        self._ids = value
        self._lookup = None

    def _key(self, value):
        if self.id_shape != (-1,):
            return tuple(value.tolist()) if isinstance(value, np.ndarray) else tuple(value)
        return value.item() if isinstance(value, np.generic) else value

    @property
    def id_to_idx(self) -> dict:
        if self._lookup is None:
            ids = self.node_ids
            self._lookup = {} if ids is None else {self._key(v): i for i, v in enumerate(ids)}
        return self._lookup

    # -------------------------------------------------------------- reference API
    @property
    def has_ids(self) -> bool:
        return self._pending is not None or self._ids is not None

    def num_ids(self) -> int:
        ids = self.node_ids
        return 0 if ids is None else len(ids)

    def add_id(self, node_id) -> None:
        if self._key(node_id) in self.id_to_idx:
            raise ValueError("ID already present in the mapping.")
        if isinstance(node_id, (list, tuple)):
            arr = to_numpy(node_id)
            self.id_shape = (-1, *arr.shape)
            arr = arr.reshape(1, *arr.shape)
        else:
            arr = to_numpy([node_id])
        idx = self.num_ids()
        self._ids = arr if self._ids is None else np.concatenate((self._ids, arr))
        self.id_to_idx[self._key(node_id)] = idx

    def add_ids(self, node_ids) -> None:
        start = self.num_ids()
        if isinstance(node_ids, list) and len(node_ids) and isinstance(node_ids[0], (list, tuple)):
            self.id_shape = (-1, *to_numpy(node_ids[0]).shape)
        new = to_numpy(node_ids)
        merged = new if self._ids is None else np.concatenate((self._ids, new))
        distinct = np.unique(merged, axis=0 if self.id_shape != (-1,) else None)
        if len(distinct) != len(merged):
            raise ValueError("IDs are not unique or already present in the mapping.")
        lookup = self.id_to_idx
        self._ids = merged
        for offset, v in enumerate(new):
            lookup[self._key(v)] = start + offset

    def to_id(self, idx: int):
        if not self.has_ids:
            return idx
        ids = self.node_ids
        if self.id_shape != (-1,):
            return tuple(ids[idx].tolist())
        if ids.dtype.type is np.str_:
            return str(ids[idx])
        return ids[idx]

    def to_ids(self, idxs):
        if not self.has_ids:
            return idxs
        return self.node_ids[to_numpy(idxs)]

    def to_idx(self, node):
        if not self.has_ids:
            return node
        return self.id_to_idx[tuple(node) if self.id_shape != (-1,) else node]

    def to_idxs(self, nodes, device: Optional[torch.device] = None) -> torch.Tensor:
        if not self.has_ids:
            return torch.tensor(nodes, device=device)
        arr = to_numpy(nodes)
        lookup = self.id_to_idx
        if self.id_shape == (-1,):
            flat = [lookup[self._key(v)] for v in arr.reshape(-1)]
            return torch.tensor(flat, device=device).reshape(arr.shape)
        rows = arr.reshape(self.id_shape)
        flat = [lookup[tuple(r.tolist())] for r in rows]
        return torch.tensor(flat, device=device).reshape(arr.shape[: -len(self.id_shape) + 1])

    def __str__(self) -> str:
        return "".join(f"{k} -> {i}\n" for k, i in self.id_to_idx.items())
