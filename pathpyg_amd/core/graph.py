"""Static / higher-order graph container, API-compatible with ``pathpyG.core.graph.Graph`` on the hot
path (reference src/pathpyG/core/graph.py:29-119, 273-326, 616-650).

What ``Graph.__init__`` does in the reference — infer ``num_nodes``, stable row sort of the edge
index and of every ``edge_*`` attribute, bounds validation, CSR/CSC, default ``node_sequence`` — is
kept, with two changes that matter at 10^7 edges: the sort/validation run as HIP kernels, and the
derived structures the reference builds eagerly in Python (``edge_to_index`` dict, CSR, CSC) are
computed on first access.
"""
from __future__ import annotations

import logging
from typing import Iterable, Optional

import numpy as np
import torch

from .. import _dispatch
from ..data import Data
from .index_map import IndexMap

logger = logging.getLogger("pathpyg_amd")


class Graph:
    """Directed (multi-)graph with node/edge attributes stored in a :class:`~pathpyg_amd.data.Data` bag.

    ``data.edge_index`` is int64 ``[2, m]`` sorted by source row (stable); ``data.node_sequence`` is
    ``[n, order]`` and maps (higher-order) nodes to first-order node indices.
    """

    def __init__(self, data: Data, mapping: Optional[IndexMap] = None, *, _row_sorted: bool = False):
        self.mapping = IndexMap() if mapping is None else mapping
        self.is_undirected_flag = False

        edge_index = _dispatch.plain(data.edge_index)
        if edge_index is None:
            raise ValueError("data must contain an edge_index")
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.to(torch.int64)
        lo, hi = (0, -1)
        if "num_nodes" not in data or not _row_sorted:
            lo, hi = _dispatch.minmax(edge_index) if edge_index.numel() else (0, -1)
        if "num_nodes" not in data:
            data.num_nodes = hi + 1                        # reference graph.py:85-87
            logger.debug("Inferred number of nodes from edge_index, n = %s", data.num_nodes)
        n = int(data.num_nodes)
        if not _row_sorted and edge_index.numel() and (lo < 0 or hi >= n):
            # the reference fails in EdgeIndex(sparse_size=...) / validate() (graph.py:93-98,107)
            logger.error("edge_index holds node indices outside [0, %s)", n)
            raise ValueError("sparse size of EdgeIndex must match number of nodes!")
        data.edge_index = edge_index
        self.data = data

        # stable sort by source row, permuting every edge_* attribute alike (graph.py:103-105)
        if not _row_sorted and edge_index.size(1) > 1 and not _dispatch.is_sorted(edge_index[0]):
            perm = _dispatch.stable_argsort(edge_index[0], (0, max(n - 1, 0)))
            data.edge_index = edge_index[:, perm].contiguous()
            for attr in self.edge_attrs():
                value = data[attr]
                if isinstance(value, torch.Tensor):
                    data[attr] = value[perm.to(value.device)]
                elif isinstance(value, np.ndarray):
                    data[attr] = value[perm.cpu().numpy()]

        self._edge_to_index: Optional[dict] = None
        self._csr: Optional[tuple] = None
        self._csc: Optional[tuple] = None

        if "node_sequence" not in data:
            data.node_sequence = torch.arange(n, device=data.edge_index.device).reshape(-1, 1)

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def from_edge_index(edge_index: torch.Tensor, mapping: Optional[IndexMap] = None, num_nodes: int | None = None) -> "Graph":
        """Graph from a ``[2, m]`` index tensor (reference graph.py:122-161)."""
        if not num_nodes:
            return Graph(Data(edge_index=edge_index), mapping=mapping)
        if mapping is not None and mapping.num_ids() != num_nodes:
            logger.error("Number of node IDs in mapping must match num_nodes")
            raise ValueError("Number of node IDs in mapping must match num_nodes")
        return Graph(Data(edge_index=edge_index, num_nodes=num_nodes), mapping=mapping)

    @staticmethod
    def from_edge_list(edge_list: Iterable, is_undirected: bool = False, mapping: Optional[IndexMap] = None,
                       device: Optional[torch.device] = None) -> "Graph":
        """Graph from (source, destination) tuples; IDs are indexed in lexicographic order unless a
        mapping is given (reference graph.py:163-211)."""
        edge_list = list(edge_list)
        if len(edge_list) == 0:
            return Graph(Data(edge_index=torch.empty((2, 0), dtype=torch.int64, device=device), num_nodes=0), mapping=IndexMap())
        if mapping is None:
            node_ids = np.unique(np.array(edge_list))
            if np.issubdtype(node_ids.dtype, np.str_) and np.char.isnumeric(node_ids).all():
                node_ids = np.sort(node_ids.astype(int)).astype(str)
            mapping = IndexMap(node_ids)
        edge_index = mapping.to_idxs(edge_list, device=device).T.contiguous()
        data = Data(edge_index=edge_index, num_nodes=mapping.num_ids())
        g = Graph(data, mapping=mapping)
        g.is_undirected_flag = bool(is_undirected)
        return g

    # ------------------------------------------------------------------ derived structures (lazy)
    @property
    def edge_to_index(self) -> dict:
        """``{(u, v): edge position}``; the last occurrence wins for multi-edges (graph.py:110-112)."""
        if self._edge_to_index is None:
            src, dst = self.data.edge_index.cpu().tolist()
            self._edge_to_index = {(u, v): i for i, (u, v) in enumerate(zip(src, dst))}
        return self._edge_to_index

    def _build_csr(self) -> tuple:
        if self._csr is None:
            ei = self.data.edge_index
            self._csr = (_dispatch.ptr_from_sorted(ei[0], self.n), ei[1])
        return self._csr

    def _build_csc(self) -> tuple:
        if self._csc is None:
            ei = self.data.edge_index
            if ei.size(1) == 0:
                self._csc = (torch.zeros(self.n + 1, dtype=torch.int64, device=ei.device), ei[0], torch.empty(0, dtype=torch.int64, device=ei.device))
            else:
                perm = _dispatch.stable_argsort(ei[1], (0, max(self.n - 1, 0)))
                self._csc = (_dispatch.ptr_from_sorted(ei[1][perm], self.n), ei[0][perm], perm)
        return self._csc

    @property
    def row_ptr(self) -> torch.Tensor:
        return self._build_csr()[0]

    @property
    def col(self) -> torch.Tensor:
        return self._build_csr()[1]

    @property
    def col_ptr(self) -> torch.Tensor:
        return self._build_csc()[0]

    @property
    def row(self) -> torch.Tensor:
        return self._build_csc()[1]

    @property
    def csc_perm(self) -> torch.Tensor:
        """Permutation that orders the (row-sorted) edges by destination column (stable)."""
        return self._build_csc()[2]

    # ------------------------------------------------------------------ reference API
    @property
    def device(self) -> torch.device:
        return self.data.edge_index.device

    def to(self, device) -> "Graph":
        """Move all tensors to ``device`` in place and return ``self`` (reference graph.py:273-296)."""
        self.data.edge_index = self.data.edge_index.to(device)
        self.data.node_sequence = self.data.node_sequence.to(device)
        for attr in self.node_attrs() + self.edge_attrs():
            if isinstance(self.data[attr], torch.Tensor):
                self.data[attr] = self.data[attr].to(device)
        if "inverse_idx" in self.data and isinstance(self.data.inverse_idx, torch.Tensor):
            self.data.inverse_idx = self.data.inverse_idx.to(device)
        if self._csr is not None:
            self._csr = tuple(t.to(device) for t in self._csr)
        if self._csc is not None:
            self._csc = tuple(t.to(device) for t in self._csc)
        return self

    def node_attrs(self) -> list:
        return [k for k in self.data.keys() if k != "node_sequence" and k.startswith("node_")]

    def edge_attrs(self) -> list:
        return [k for k in self.data.keys() if k != "edge_index" and k.startswith("edge_")]

    @property
    def n(self) -> int:
        return int(self.data.num_nodes)

    @property
    def m(self) -> int:
        if self.is_directed():
            return self.data.num_edges
        loops = int((self.data.edge_index[0] == self.data.edge_index[1]).sum())
        return int((self.data.edge_index.size(1) - loops) / 2 + loops)

    @property
    def order(self) -> int:
        return int(self.data.node_sequence.size(1))

    def is_directed(self) -> bool:
        return not self.is_undirected_flag

    def is_undirected(self) -> bool:
        return self.is_undirected_flag

    @property
    def nodes(self) -> list:
        ids = self.mapping.to_ids(self.data.node_sequence) if self.order > 1 else None
        if self.order > 1:
            return [tuple(x) for x in np.asarray(ids).tolist()]
        node_list = self.mapping.to_ids(np.arange(self.n))
        return node_list.tolist() if isinstance(node_list, np.ndarray) else list(node_list)

    @property
    def edges(self) -> list:
        ei = self.data.edge_index.cpu()
        if self.order > 1:
            ns = self.data.node_sequence.cpu()
            a = np.asarray(self.mapping.to_ids(ns[ei[0]])).tolist()
            b = np.asarray(self.mapping.to_ids(ns[ei[1]])).tolist()
            return [(tuple(u), tuple(v)) for u, v in zip(a, b)]
        ids = self.mapping.to_ids(ei)
        ids = ids.tolist() if isinstance(ids, (np.ndarray, torch.Tensor)) else ids
        return list(zip(ids[0], ids[1]))

    def degrees(self, mode: str = "in", edge_attr: str | None = None, return_tensor: bool = False):
        """(Weighted) in- or out-degrees (reference graph.py:486-516): counts as int32, weighted sums in the dtype of the
        attribute; a tensor with ``return_tensor`` else ``{node: degree}``.  Counting and the segment sums run on the GPU."""
        from .. import _hip
        ei = self.data.edge_index
        dev = _dispatch.compute_device(ei)
        which = ei[1] if mode == "in" else ei[0]
        if not edge_attr:
            d = _hip.degree(which.to(dev).contiguous(), self.n).to(torch.int32).to(ei.device)
        else:
            w = getattr(self.data, edge_attr, None)
            if w is None:
                raise AttributeError(f"graph has no edge attribute {edge_attr}")
            wf = w.to(dev).to(torch.float32).reshape(-1, 1).contiguous()
            if mode == "in":
                ptr, perm = self.col_ptr.to(dev).to(torch.int32), self.csc_perm.to(dev).to(torch.int32)
            else:
                ptr, perm = self.row_ptr.to(dev).to(torch.int32), torch.arange(ei.size(1), device=dev, dtype=torch.int32)
            d = _hip.spmm(ptr, perm, None, self.n, wf).reshape(-1).to(ei.device)       # segment sum over the node's edges
            if not w.is_floating_point():
                d = d.round().to(w.dtype)
        if return_tensor:
            return d
        return {node: deg.item() for node, deg in zip(self.nodes, d)}

    def transition_probabilities(self, edge_attr: str | None = None) -> torch.Tensor:
        """Per-edge transition probability ``w_e / (weighted) out-degree of its source`` (reference graph.py:518-533)."""
        out = self.degrees(mode="out", edge_attr=edge_attr, return_tensor=True)
        ei = self.data.edge_index
        w = torch.ones(ei.size(1), device=ei.device) if edge_attr is None else getattr(self.data, edge_attr)
        return w / _dispatch.edge_attr(ei, out.to(w.dtype if w.is_floating_point() else torch.float32), "src")

    def successors(self, node) -> list:
        i = self.mapping.to_idx(node)
        ptr = self.row_ptr
        return self.mapping.to_ids(self.col[ptr[i]:ptr[i + 1]].cpu()).tolist()

    def predecessors(self, node) -> list:
        i = self.mapping.to_idx(node)
        ptr = self.col_ptr
        return self.mapping.to_ids(self.row[ptr[i]:ptr[i + 1]].cpu()).tolist()

    def __str__(self) -> str:
        kind = "Directed" if self.is_directed() else "Undirected"
        return f"{kind} graph with {self.n} nodes and {self.m} edges (order {self.order})"
