"""Time-stamped edge stream container, API-compatible with ``pathpyG.core.temporal_graph.TemporalGraph``
on the hot path (reference src/pathpyG/core/temporal_graph.py:17-176).

Event order: the reference sorts with an *unstable* ``torch.argsort`` (temporal_graph.py:58), so the
relative order of events sharing a timestamp is implementation-defined there.  Here the sort is a
STABLE radix sort on the GPU: ties keep their input order, which makes event ids — and therefore the raw
event-graph index and every ``inverse_idx`` — reproducible.  Aggregated layers are tie-order invariant.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import _dispatch
from ..data import Data
from .graph import Graph
from .index_map import IndexMap


class TemporalGraph(Graph):
    def __init__(self, data: Data, mapping: IndexMap | None = None) -> None:
        self.data = data
        self.mapping = IndexMap() if mapping is None else mapping
        self.is_undirected_flag = False
        edge_index = _dispatch.plain(data.edge_index)
        if not isinstance(edge_index, torch.Tensor):
            edge_index = torch.as_tensor(edge_index)
        data.edge_index = edge_index.to(torch.int64).contiguous()
        if not isinstance(data.time, torch.Tensor):
            data.time = torch.as_tensor(data.time, device=data.edge_index.device)
        if "num_nodes" not in data:
            data.num_nodes = (_dispatch.minmax(data.edge_index)[1] + 1) if data.edge_index.numel() else 0

        # time-sort the events and every per-event attribute (temporal_graph.py:58-63), stably.  One read-back ({descents, min, max})
        # decides whether to sort at all and sizes the radix keys; edge_index and time are permuted in one pass.
        if data.time.numel() > 1:
            ordered = data.time if data.time.dtype in (torch.int64, torch.float64) else None
            if ordered is None:
                descents, value_range = (0 if _dispatch.is_sorted(data.time) else 1), None
            else:
                descents, lo, hi = _dispatch.time_stats(ordered)
                value_range = (lo, hi) if ordered.dtype == torch.int64 else None
            if descents:
                perm = _dispatch.stable_argsort(data.time, value_range)
                fused = ordered is not None and data.time.device == data.edge_index.device
                if fused:
                    data.edge_index, data.time = _dispatch.gather_events(data.edge_index, data.time, perm)
                for key in set(data.edge_attrs()) | {"time"}:
                    value = data[key]
                    if key in ("edge_index", "time") and fused:
                        continue
                    if key == "edge_index":
                        data.edge_index = value[:, perm.to(value.device)].contiguous()
                    elif isinstance(value, torch.Tensor):
                        data[key] = value[perm.to(value.device)]
                    elif isinstance(value, np.ndarray):
                        data[key] = value[perm.cpu().numpy()]

        self._edge_to_index = None
        self._tedge_to_index = None
        self._csr = None
        self._csc = None

    @staticmethod
    def from_edge_list(edge_list, num_nodes: Optional[int] = None, device: Optional[torch.device] = None) -> "TemporalGraph":
        """Temporal graph from ``(source, destination, timestamp)`` tuples (reference temporal_graph.py:77-128).
        Integer timestamps become int64, anything else float64; node IDs are indexed in sorted order."""
        if len(edge_list) == 0:
            return TemporalGraph(Data(edge_index=torch.empty((2, 0), dtype=torch.long, device=device),
                                      time=torch.empty((0,), dtype=torch.long, device=device), num_nodes=num_nodes or 0))
        rows = np.array(edge_list)
        if isinstance(edge_list[0][2], (int, np.integer)):
            ts = torch.tensor(rows[:, 2].astype(np.int64), device=device)
        else:
            ts = torch.tensor(rows[:, 2].astype(np.float64), device=device)
        endpoints = rows[:, :2]
        ids, inverse = np.unique(endpoints, return_inverse=True)      # vectorised ID -> index (no per-endpoint dict lookups)
        index_map = IndexMap(ids)
        edge_index = torch.from_numpy(inverse.reshape(endpoints.shape).T.astype(np.int64)).contiguous()
        if device is not None:
            edge_index = edge_index.to(device)
        return TemporalGraph(Data(edge_index=edge_index, time=ts, num_nodes=num_nodes or index_map.num_ids()), mapping=index_map)

    # ------------------------------------------------------------------ lazy host dictionaries (temporal_graph.py:71-75)
    @property
    def edge_to_index(self) -> dict:
        if self._edge_to_index is None:
            src, dst = self.data.edge_index.cpu().tolist()
            self._edge_to_index = {(u, v): i for i, (u, v) in enumerate(zip(src, dst))}
        return self._edge_to_index

    @property
    def tedge_to_index(self) -> dict:
        if self._tedge_to_index is None:
            src, dst = self.data.edge_index.cpu().tolist()
            t = self.data.time.cpu().tolist()
            self._tedge_to_index = {(u, v, x): i for i, (u, v, x) in enumerate(zip(src, dst, t))}
        return self._tedge_to_index

    @property
    def temporal_edges(self) -> list:
        ids = self.mapping.to_ids(self.data.edge_index.cpu())
        ids = ids.tolist() if isinstance(ids, (np.ndarray, torch.Tensor)) else ids
        return list(zip(ids[0], ids[1], self.data.time.cpu().tolist()))

    def to(self, device) -> "TemporalGraph":
        self.data.edge_index = self.data.edge_index.to(device)
        self.data.time = self.data.time.to(device)
        for attr in self.node_attrs() + self.edge_attrs():
            if isinstance(self.data[attr], torch.Tensor):
                self.data[attr] = self.data[attr].to(device)
        return self

    @property
    def order(self) -> int:
        return 1

    @property
    def start_time(self):
        return self.data.time[0].item() if self.data.time.numel() else None

    @property
    def end_time(self):
        return self.data.time[-1].item() if self.data.time.numel() else None

    def shuffle_time(self) -> None:
        """Randomly permute the timestamps (reference temporal_graph.py:176-178); the events are re-sorted by their new times."""
        data = self.data
        data.time = data.time[torch.randperm(len(data.time), device=data.time.device)]
        TemporalGraph.__init__(self, data, self.mapping)

    def to_static_graph(self, weighted: bool = False, time_window=None) -> Graph:
        """Time-aggregated static graph, optionally restricted to ``time_window = (start, end)`` and with multi-edges merged into
        an ``edge_weight`` (reference temporal_graph.py:180-203)."""
        edge_index = self.data.edge_index
        if time_window is not None:
            keep = (self.data.time >= time_window[0]) & (self.data.time < time_window[1])
            edge_index = edge_index[:, keep]
        n = int(edge_index.max().item()) + 1 if edge_index.numel() else 0
        if weighted:
            merged, weight = _dispatch.coalesce(edge_index, torch.ones(edge_index.size(1), device=edge_index.device), n, "sum")
            return Graph(Data(edge_index=merged, edge_weight=weight, num_nodes=n), self.mapping, _row_sorted=True)
        return Graph(Data(edge_index=edge_index.contiguous(), num_nodes=n), self.mapping)

    def to_undirected(self) -> "TemporalGraph":
        """Every event is duplicated in the opposite direction with the same timestamp (reference temporal_graph.py:205-231;
        like there, edge attributes are not carried over)."""
        ei = self.data.edge_index
        return TemporalGraph(Data(edge_index=torch.cat((ei, ei.flip(0)), dim=1), time=torch.cat((self.data.time, self.data.time)),
                                  num_nodes=self.n), mapping=self.mapping)

    def _subset(self, selector) -> "TemporalGraph":
        data = Data(edge_index=self.data.edge_index[:, selector], time=self.data.time[selector], num_nodes=self.n)
        for key in self.node_attrs():
            data[key] = self.data[key]
        for key in self.edge_attrs():
            value = self.data[key]
            if isinstance(value, torch.Tensor) or isinstance(selector, slice):
                data[key] = value[selector]
            else:
                data[key] = value[selector.cpu().numpy()]
        return TemporalGraph(data, mapping=self.mapping)

    def get_batch(self, start_idx: int, end_idx: int) -> "TemporalGraph":
        """Events ``start_idx .. end_idx - 1`` of the time-ordered stream, with their edge attributes (reference :233-264)."""
        return self._subset(slice(start_idx, end_idx))

    def get_window(self, start_time, end_time) -> "TemporalGraph":
        """Events with ``start_time <= t < end_time``, with their edge attributes (reference temporal_graph.py:266-298)."""
        return self._subset((self.data.time >= start_time) & (self.data.time < end_time))

    def __getitem__(self, key):
        """As :meth:`Graph.__getitem__`; edge attributes also accept ``(name, v, w, t)`` for one time-stamped edge (reference
        temporal_graph.py:300-324)."""
        if isinstance(key, tuple) and key[0] in self.edge_attrs() and len(key) == 4:
            return self.data[key[0]][self.tedge_to_index[self.mapping.to_idx(key[1]), self.mapping.to_idx(key[2]), key[3]]]
        return Graph.__getitem__(self, key)

    def __str__(self) -> str:
        return f"Temporal Graph with {self.n} nodes and {self.data.num_edges} events in [{self.start_time}, {self.end_time}]"
