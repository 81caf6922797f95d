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
    @classmethod
    def _from_parts(cls, data: Data, mapping: Optional[IndexMap] = None) -> "Graph":
        """A layer whose ``data`` is known to be valid and (row, col)-sorted — it comes out of the order-2 builder (``pp_debruijn2_*``): nothing is
        checked and nothing is read, so the tensors may still be :class:`~pathpyg_amd.data.Lazy`."""
        g = object.__new__(cls)
        g.mapping = IndexMap() if mapping is None else mapping
        g.is_undirected_flag = False
        g.data = data
        g._edge_to_index = g._csr = g._csc = None
        return g

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
        return int(self.data.peek("node_sequence").shape[1])            # (a Lazy answers from its shape)

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

    def get_successors(self, row_idx: int) -> torch.Tensor:
        """Indices of all successors of the node with index ``row_idx`` (reference graph.py:379-394)."""
        ptr = self.row_ptr
        if row_idx + 1 < ptr.size(0):
            return self.col[ptr[row_idx]:ptr[row_idx + 1]]
        return torch.tensor([], device=self.data.edge_index.device)

    def get_predecessors(self, col_idx: int) -> torch.Tensor:
        """Indices of all predecessors of the node with index ``col_idx`` (reference graph.py:396-411)."""
        ptr = self.col_ptr
        if col_idx + 1 < ptr.size(0):
            return self.row[ptr[col_idx]:ptr[col_idx + 1]]
        return torch.tensor([], device=self.data.edge_index.device)

    def is_edge(self, v, w) -> bool:
        """Whether the edge (v, w) exists (reference graph.py:447-462)."""
        row = self.mapping.to_idx(v)
        ptr = self.row_ptr
        return bool((self.col[ptr[row]:ptr[row + 1]] == self.mapping.to_idx(w)).any())

    def has_self_loops(self) -> bool:
        ei = self.data.edge_index
        return bool((ei[0] == ei[1]).any())

    @property
    def in_degrees(self) -> dict:
        return self.degrees(mode="in")

    @property
    def out_degrees(self) -> dict:
        return self.degrees(mode="out")

    def sparse_adj_matrix(self, edge_attr=None):
        """scipy ``coo_matrix`` adjacency matrix, optionally weighted by an edge attribute (reference graph.py:464-478)."""
        from scipy.sparse import coo_matrix
        ei = self.data.edge_index.cpu().numpy()
        if edge_attr is None:
            values = np.ones(ei.shape[1])
        else:
            values = self.data[edge_attr]
            values = values.detach().cpu().numpy() if isinstance(values, torch.Tensor) else np.asarray(values)
            values = values.reshape(-1)
        return coo_matrix((values, (ei[0], ei[1])), shape=(self.n, self.n))

    def laplacian(self, normalization=None, edge_attr: str | None = None):
        """Graph Laplacian as a scipy ``coo_matrix`` with PyG's ``get_laplacian`` conventions (reference graph.py:535-560):
        self loops removed, degrees = weighted out-degrees; ``None``: D - A, ``"sym"``: I - D^-1/2 A D^-1/2, ``"rw"``: I - D^-1 A."""
        from scipy.sparse import coo_matrix
        if normalization not in (None, "sym", "rw"):
            raise ValueError(f"unknown normalization {normalization}")
        ei = self.data.edge_index.cpu()
        w = torch.ones(ei.size(1)) if edge_attr is None else self.data[edge_attr].detach().cpu().reshape(-1).to(torch.float32)
        keep = ei[0] != ei[1]
        ei, w = ei[:, keep], w[keep]
        n = self.n
        deg = torch.zeros(n, dtype=w.dtype).index_add_(0, ei[0], w)
        loops = torch.arange(n).repeat(2, 1)
        if normalization is None:
            index, weight = torch.cat((ei, loops), dim=1), torch.cat((-w, deg))
        else:
            if normalization == "sym":
                dis = deg.pow(-0.5)
                dis[torch.isinf(dis)] = 0
                a = dis[ei[0]] * w * dis[ei[1]]
            else:
                dinv = 1.0 / deg
                dinv[torch.isinf(dinv)] = 0
                a = dinv[ei[0]] * w
            index, weight = torch.cat((ei, loops), dim=1), torch.cat((-a, torch.ones(n, dtype=a.dtype)))
        return coo_matrix((weight.numpy(), (index[0].numpy(), index[1].numpy())), shape=(n, n))

    # ------------------------------------------------------------------ derived graphs
    def to_undirected(self) -> "Graph":
        """Undirected version: every edge is added in the opposite direction and parallel edges are merged; edge attributes follow
        the first original edge of each merged pair (reference graph.py:211-250, PyG ``to_undirected(reduce="min")``)."""
        ei = self.data.edge_index
        ids = torch.arange(ei.size(1), device=ei.device)
        both = torch.cat((ei, ei.flip(0)), dim=1)
        merged, attr_idx = _dispatch.coalesce(both, torch.cat((ids, ids)), self.n, "min")
        data = Data(edge_index=merged, num_nodes=self.n)
        for key in self.node_attrs():
            data[key] = self.data[key]
        for key in self.edge_attrs():
            value = self.data[key]
            data[key] = value[attr_idx.to(value.device)] if isinstance(value, torch.Tensor) else value[attr_idx.cpu().numpy()]
        g = Graph(data, self.mapping, _row_sorted=True)
        g.is_undirected_flag = True
        g.data.edge_index.is_undirected = True            # what callers of the reference read from the EdgeIndex
        return g

    def to_weighted_graph(self) -> "Graph":
        """Merge parallel edges into single edges with an ``edge_weight`` counting them (reference graph.py:252-271)."""
        ei = self.data.edge_index
        merged, weight = _dispatch.coalesce(ei, torch.ones(ei.size(1), device=ei.device), self.n, "sum")
        return Graph(Data(edge_index=merged, edge_weight=weight, num_nodes=self.n), mapping=self.mapping, _row_sorted=True)

    # ------------------------------------------------------------------ attribute access
    def __getitem__(self, key):
        """Graph attribute ``g["name"]``, node attribute ``g["node_x", node]`` or edge attribute ``g["edge_x", v, w]``
        (reference graph.py:562-578)."""
        if not isinstance(key, tuple):
            if key in self.data.keys():
                return self.data[key]
            raise KeyError(key + " is not a graph attribute")
        if key[0] in self.node_attrs():
            return self.data[key[0]][self.mapping.to_idx(key[1])]
        if key[0] in self.edge_attrs():
            return self.data[key[0]][self.edge_to_index[self.mapping.to_idx(key[1]), self.mapping.to_idx(key[2])]]
        raise KeyError(key[0] + " is not a node or edge attribute")

    def __setitem__(self, key, val) -> None:
        """Store a graph / node / edge attribute or one of its entries (reference graph.py:580-616)."""
        if not isinstance(key, tuple):
            if key.startswith("node_") and val.size(0) != self.n:
                raise ValueError("Attribute must have same length as number of nodes")
            if key.startswith("edge_") and val.size(0) != self.m:
                raise ValueError("Attribute must have same length as number of edges")
            self.data[key] = val
        elif key[0].startswith("node_"):
            if key[0] not in self.data.keys():
                raise KeyError("Attribute does not yet exist. Setting the value of a specific node attribute requires that the attribute already exists.")
            self.data[key[0]][self.mapping.to_idx(key[1])] = val
        elif key[0].startswith("edge_"):
            if key[0] not in self.data.keys():
                raise KeyError("Attribute does not yet exist. Setting the value of a specific edge attribute requires that the attribute already exists.")
            self.data[key[0]][self.edge_to_index[self.mapping.to_idx(key[1]), self.mapping.to_idx(key[2])]] = val
        else:
            raise KeyError("node and edge specific attributes should be prefixed with 'node_' or 'edge_'")

    def __add__(self, other: "Graph", reduce: str = "sum") -> "Graph":
        """Union of two graphs (reference graph.py:673-771): node IDs of both mappings are merged into a new sorted mapping, edges
        and edge attributes are concatenated, node attributes of nodes present in both graphs are reduced with ``reduce``
        (``sum``, ``mean``, ``mul``, ``min``, ``max``)."""
        m1, m2 = self.mapping, other.mapping
        ids1, ids2 = np.asarray(m1.to_ids(np.arange(self.n))), np.asarray(m2.to_ids(np.arange(other.n)))
        nodes = np.concatenate([ids1, ids2])
        mapping = IndexMap(np.unique(nodes, axis=0).tolist())
        dev = self.data.edge_index.device

        def remap_index(g: "Graph", index: torch.Tensor) -> torch.Tensor:
            return mapping.to_idxs(np.asarray(g.mapping.to_ids(index.cpu())), device=dev)

        data = Data(edge_index=torch.cat((remap_index(self, self.data.edge_index), remap_index(other, other.data.edge_index)), dim=1),
                    num_nodes=mapping.num_ids(),
                    node_sequence=torch.cat((self.data.node_sequence, other.data.node_sequence), dim=0))
        if "inverse_idx" in self.data and "inverse_idx" in other.data:      # higher-order layers: instances -> merged node ids
            data.inverse_idx = torch.cat((remap_index(self, self.data.inverse_idx), remap_index(other, other.data.inverse_idx)))
        for key in self.edge_attrs():
            if key in other.data:
                a, b = self.data[key], other.data[key]
                data[key] = torch.cat((a, b), dim=0) if isinstance(a, torch.Tensor) else np.concatenate((a, b))
        target = mapping.to_idxs(nodes, device=dev)
        for key in self.node_attrs():
            if key not in other.data:
                continue
            a, b = self.data[key], other.data[key]
            if not isinstance(a, torch.Tensor):
                raise ValueError("Node attribute " + key + " is not a tensor and cannot be reduced.")
            src = torch.cat((a, b), dim=0)
            index = target.to(src.device).reshape((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
            mode = {"sum": "sum", "mean": "mean", "mul": "prod", "min": "amin", "max": "amax"}[reduce]
            out = torch.zeros((mapping.num_ids(),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            data[key] = out.scatter_reduce_(0, index, src, reduce=mode, include_self=False)
        return Graph(data, mapping=mapping)

    def __str__(self) -> str:
        kind = "Directed" if self.is_directed() else "Undirected"
        return f"{kind} graph with {self.n} nodes and {self.m} edges (order {self.order})"
