"""Multi-order De Bruijn graph model, API-compatible with ``pathpyG.core.multi_order_model.MultiOrderModel``
for the construction path and the DBGNN export (reference src/pathpyG/core/multi_order_model.py:61-241, 511-554).

Each layer ``k`` is a :class:`Graph` whose nodes are the distinct length-``k`` node sequences observed in the
input (walks or time-respecting paths) and whose weighted edges count their continuations.  All tensor work
(event-graph lift, line-graph lifts, sequence extension, unique/coalesce) runs in HIP kernels; the
higher-order ``IndexMap`` of every layer is created lazily instead of by a Python loop over nodes.
The likelihood / order-selection methods (:243-509) reuse the lift kernels for their path counts.
"""
from __future__ import annotations

import logging
from typing import Optional

import torch
from scipy.stats import chi2

from .. import _dispatch
from ..algorithms.lift_order import (
    aggregate_edge_index,
    aggregate_node_attributes,
    lift_order_edge_index,
    lift_order_edge_index_weighted,
)
from ..algorithms.temporal import lift_order_temporal
from ..data import Data, Lazy
from ..utils.dbgnn import generate_bipartite_edge_index
from .graph import Graph
from .index_map import IndexMap
from .path_data import PathData
from .temporal_graph import TemporalGraph

logger = logging.getLogger("pathpyg_amd")

FUSED_BUILDER = True    # from_temporal_graph(max_order=2) on device-resident streams: the node-by-node order-2 builder (pp_debruijn2_*); False = generic kernels


class MultiOrderModel:
    """Dictionary of De Bruijn graphs ``layers[k]`` for k = 1..max_order."""

    def __init__(self) -> None:
        self.layers: dict[int, Graph] = {}

    def __str__(self) -> str:
        return f"MultiOrderModel with max. order {max(self.layers) if self.layers else 0}"

    def to(self, device) -> "MultiOrderModel":
        for g in self.layers.values():
            g.to(device)
        return self

    @staticmethod
    def iterate_lift_order(
        edge_index: torch.Tensor,
        node_sequence: torch.Tensor,
        mapping: IndexMap,
        edge_weight: torch.Tensor | None = None,
        aggr: str = "src",
        save: bool = True,
    ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None, Graph | None]:
        """One order lift (reference multi_order_model.py:83-122): line-graph lift of ``edge_index``, lifted
        weights, extended instance sequences and — when ``save`` — the aggregated De Bruijn graph."""
        num_instances = node_sequence.size(0)
        if edge_weight is None:
            ho_index = lift_order_edge_index(edge_index, num_nodes=num_instances)
        else:
            ho_index, edge_weight = lift_order_edge_index_weighted(edge_index, edge_weight=edge_weight, num_nodes=num_instances, aggr=aggr)
        node_sequence = _dispatch.extend_node_sequence(edge_index, node_sequence)
        gk = None
        if save:
            gk = aggregate_edge_index(ho_index, node_sequence, edge_weight)
            gk.mapping = IndexMap.from_node_sequence(mapping, gk.data.node_sequence)
        return ho_index, node_sequence, edge_weight, gk

    @staticmethod
    def from_temporal_graph(
        g: TemporalGraph,
        delta: float | int = 1,
        max_order: int = 1,
        weight: str = "edge_weight",
        cached: bool = True,
        event_graph: Optional[torch.Tensor] = None,
    ) -> "MultiOrderModel":
        """De Bruijn layers of the time-respecting paths of ``g`` with waiting time ``delta``
        (reference multi_order_model.py:124-192).  ``cached=False`` keeps only the top layer;
        ``event_graph`` reuses a precomputed ``lift_order_temporal(g, delta)``.

        Same layers as the reference, computed without ever materialising the per-instance node sequences
        (``[E_k, k+1]`` tensors): see :class:`_LiftChain`."""
        if max_order == 2 and event_graph is None and FUSED_BUILDER:
            # LIFT_ONLY_ORDER2 (off by default): a stream with a very large hub (BASELINE configs[2]'s generator: a node with 2 * 10^6 in-events) costs
            # the order-2 builder's hub kernels 10-11 ms where the level-by-level builder makes both layers in 5.3-5.7 ms — but it makes no GCN
            # plans, and DBGNN.forward building them from the layers' tensors costs 20 ms more than the order-2 builder's own (39 against 25 ms
            # for model + bundle + one step, tools/probes/hub_k2_routes.py): the default serves the DBGNN workflow, the switch a caller who only lifts
            fused = _multi_order_fused(g, delta, 2, weight, cached) if (LIFT_ONLY_ORDER2 and _has_large_hub(g)) else None
            if fused is None:
                fused = _second_order_fused(g, delta, weight, cached)
            if fused is not None:
                return fused
        if max_order >= 3 and FUSED_BUILDER:
            fused = _multi_order_fused(g, delta, max_order, weight, cached, event_graph)
            if fused is not None:
                return fused
        return MultiOrderModel._from_temporal_graph_generic(g, delta, max_order, weight, cached, event_graph)

    @staticmethod
    def _from_temporal_graph_generic(g: TemporalGraph, delta, max_order: int, weight: str, cached: bool, event_graph) -> "MultiOrderModel":
        """:meth:`from_temporal_graph` on the generic kernels, order by order as the reference goes: event graph (``pp_temporal_*``), line-graph
        lifts (``pp_linegraph_*``), coalesce per layer (``pp_coalesce_*``) — any ``max_order``, ``event_graph=``, host-resident streams, weights of
        any dtype, streams the level-by-level builders hand back."""
        m = MultiOrderModel()
        data = g.data if g.data.is_sorted_by_time() else g.data.sort_by_time()
        edge_index = data.edge_index
        n = int(data.num_nodes)
        dev = edge_index.device
        if weight in data:
            edge_weight = data[weight]
        else:
            edge_weight = _hip_unit()        # the reference's torch.ones(m) (:150-151) as a marker: unit weights stay unit under the "src" lift,
            #                                  and the weight of a merged edge is the length of its run (pp_coalesce_fill, no gather)
        first_order = torch.arange(n, device=dev).unsqueeze(1)
        chain = _LiftChain.first_order(edge_index, first_order, edge_weight, identity_nodes=True, num_first_order=n,
                                       want_pairs=max_order > 1)
        if cached or max_order == 1:
            m.layers[1] = chain.graph
            m.layers[1].mapping = g.mapping
        if max_order > 1:
            ho_index = lift_order_temporal(g, delta) if event_graph is None else event_graph
            chain = chain.to_second_order(edge_index, ho_index,
                                          edge_weight if isinstance(edge_weight, str) else aggregate_node_attributes(ho_index, edge_weight, "src"),
                                          save=cached or max_order == 2, want_edge_ids=max_order > 2)
            if cached or max_order == 2:
                m.layers[2] = chain.graph
                m.layers[2].mapping = IndexMap.from_node_sequence(g.mapping, chain.graph.data.node_sequence)
            for k in range(3, max_order + 1):
                keep = cached or k == max_order
                chain = chain.lift("src", save=keep, want_edge_ids=k < max_order)
                if keep:
                    m.layers[k] = chain.graph
                    m.layers[k].mapping = IndexMap.from_node_sequence(g.mapping, chain.graph.data.node_sequence)
        return m

    @staticmethod
    def from_path_data(path_data: PathData, max_order: int = 1, mode: str = "propagation", cached: bool = True) -> "MultiOrderModel":
        """De Bruijn layers of observed walks (reference multi_order_model.py:194-241).
        ``mode="propagation"`` carries the walk weight along; ``"diffusion"`` splits it by out-degree."""
        m = MultiOrderModel()
        walks = path_data.data
        edge_index = walks.edge_index
        node_sequence = walks.node_sequence
        edge_weight = walks.dag_weight.repeat_interleave(walks.dag_num_edges)
        aggr = "src"
        if mode == "diffusion":
            outdeg = _dispatch_degree(edge_index[0], node_sequence.size(0))
            edge_weight = edge_weight / aggregate_node_attributes(edge_index, outdeg, "src")
            aggr = "mul"
        chain = _LiftChain.first_order(edge_index, node_sequence, edge_weight, identity_nodes=False, num_first_order=None,
                                       want_pairs=max_order > 1)
        m.layers[1] = chain.graph
        m.layers[1].mapping = path_data.mapping
        for k in range(2, max_order + 1):
            keep = cached or k == max_order
            chain = chain.lift(aggr, save=keep, want_edge_ids=k < max_order)
            if keep:
                m.layers[k] = chain.graph
                m.layers[k].mapping = IndexMap.from_node_sequence(m.layers[1].mapping, chain.graph.data.node_sequence)
        return m

    # ------------------------------------------------------------------ model selection
    def get_mon_dof(self, max_order: Optional[int] = None, assumption: str = "paths") -> int:
        """Degrees of freedom of the multi-order model up to ``max_order`` (reference multi_order_model.py:243-309).

        "paths": sum over k of the number of length-k paths of the first-order topology (k-1 line-graph lifts of layer 1)
        minus, per order, the nodes that start at least one length-k path (their transition rows sum to one);
        "ngrams": all ``n^k (n-1)`` combinations.  The reference counts the row constraint with a sparse matrix power; here
        ``starts_k = A starts_{k-1} > 0`` is propagated instead — same count, no sparse-sparse product."""
        if max_order is None:
            max_order = max(self.layers)
        if max_order > max(self.layers):
            logger.error("max_order cannot be larger than maximum order of multi-order network")
            raise ValueError("max_order cannot be larger than maximum order of multi-order network")
        g1 = self.layers[1]
        n = int(g1.data.num_nodes)
        dof = n - 1
        if assumption == "paths":
            from .. import _hip
            edge_index = g1.data.edge_index
            for k in range(1, max_order + 1):
                if k > 1:
                    num_nodes = 0 if edge_index.numel() == 0 else _dispatch.minmax(edge_index)[1] + 1
                    edge_index = lift_order_edge_index(edge_index, num_nodes)
                dof += int(edge_index.shape[1])                      # number of paths of length k
            dev = _dispatch.compute_device(g1.data.edge_index)
            ptr = g1.row_ptr.to(dev).to(torch.int32)
            col = g1.col.to(dev).to(torch.int32).contiguous()
            starts = torch.ones((n, 1), device=dev)
            for k in range(1, max_order + 1):
                starts = (_hip.spmm(ptr, col, None, n, starts) > 0).to(torch.float32)    # has a length-k path leaving it
                dof -= int(starts.sum().item())
        elif assumption == "ngrams":
            for order in range(1, max_order + 1):
                dof += (n ** order) * (n - 1)
        else:
            logger.error("Unknown assumption %s. Only 'path' and 'ngram' are accepted.", assumption)
            raise ValueError(f"Unknown assumption {assumption}. Only 'path' and 'ngram' are accepted.")
        return int(dof)

    def get_zeroth_order_log_likelihood(self, dag_graph: Data) -> float:
        """Log-likelihood of the walks' first nodes under the node-frequency model (reference multi_order_model.py:311-336)."""
        frequencies = dag_graph.dag_weight
        is_start = torch.ones(dag_graph.num_nodes, dtype=torch.bool, device=frequencies.device)
        is_start[dag_graph.edge_index[1]] = False
        start_nodes = dag_graph.node_sequence.squeeze()[is_start]
        _, counts = torch.unique(dag_graph.node_sequence, return_counts=True)
        emission = counts / counts.sum()
        return torch.mul(frequencies, torch.log(emission[start_nodes])).sum().item()

    def get_intermediate_order_log_likelihood(self, dag_graph: Data, order: int) -> float:
        """Contribution of the first order-``order`` transition of every walk (reference multi_order_model.py:338-369)."""
        frequencies = dag_graph.dag_weight
        lengths_ho = dag_graph.dag_num_nodes - order                 # walks shrink by `order` nodes in order-k encoding
        keep = lengths_ho > 0
        frequencies = frequencies[keep]
        kept = lengths_ho[keep]
        first_of_walk = torch.cumsum(kept, 0) - kept                 # start position of each surviving walk
        probs = self.layers[order].transition_probabilities()[self.layers[order + 1].data.inverse_idx[first_of_walk]]
        return torch.mul(frequencies, torch.log(probs)).sum().item()

    def get_mon_log_likelihood(self, dag_graph: Data, max_order: int = 1) -> float:
        """Log-likelihood of the walks under the multi-order model with layers 0..``max_order``
        (reference multi_order_model.py:371-409)."""
        if max_order > 0:
            llh = self.get_zeroth_order_log_likelihood(dag_graph)
            for order in range(1, max_order):
                llh += self.get_intermediate_order_log_likelihood(dag_graph, order)
            top = self.layers[max_order]
            probs = top.transition_probabilities(edge_attr="edge_weight")
            return llh + (torch.log(probs) * top.data.edge_weight).sum().item()
        frequencies = dag_graph.dag_weight
        counts = torch.bincount(dag_graph.node_sequence.squeeze(), frequencies.repeat_interleave(dag_graph.dag_num_nodes))
        emission = counts / counts.sum()
        return torch.mul(torch.log(emission), counts).sum().item()

    def likelihood_ratio_test(self, dag_graph: Data, max_order_null: int = 0, max_order: int = 1, assumption: str = "paths",
                              significance_threshold: float = 0.01) -> tuple:
        """Likelihood-ratio test of order ``max_order`` against ``max_order_null`` (reference multi_order_model.py:411-459):
        ``(null rejected?, p-value)`` with ``x = -2 (log L0 - log L1)`` chi-square distributed in the dof difference."""
        if max_order_null >= max_order:
            logger.error("order of null hypothesis must be smaller than order of alternative hypothesis")
            raise ValueError("order of null hypothesis must be smaller than order of alternative hypothesis")
        if max_order > max(self.layers):
            logger.error("order of hypotheses must be smaller than max. order of MultiOrderModel")
            raise ValueError(f"order of hypotheses ({max_order_null} and {max_order}) must be smaller than max. order of "
                             f"MultiOrderModel {max(self.layers)}")
        x = -2 * (self.get_mon_log_likelihood(dag_graph, max_order=max_order_null)
                  - self.get_mon_log_likelihood(dag_graph, max_order=max_order))
        dof_diff = self.get_mon_dof(max_order, assumption=assumption) - self.get_mon_dof(max_order_null, assumption=assumption)
        p = 1 - chi2.cdf(x, dof_diff)
        return (p < significance_threshold), p

    def estimate_order(self, dag_data: PathData, max_order: Optional[int] = None, significance_threshold: float = 0.01) -> int:
        """Highest order whose layer significantly improves the likelihood of the walks (reference multi_order_model.py:461-509)."""
        if max_order is None:
            max_order = max(self.layers)
        if max_order > max(self.layers):
            logger.error("max_order cannot be larger than maximum order of multi-order network")
            raise ValueError("max_order cannot be larger than maximum order of multi-order network")
        if max_order <= 1:
            logger.error("max_order must be larger than one")
            raise ValueError("max_order must be larger than one")
        ours = set(self.layers[1].mapping.node_ids)
        if not set(dag_data.mapping.node_ids).issubset(ours):
            logger.error("Input paths do not have same set of nodes as multi-order network")
            raise ValueError("Input paths do not have same set of nodes as multi-order network")
        accepted = 1
        for k in range(2, max_order + 1):
            if self.likelihood_ratio_test(dag_data.data, max_order_null=k - 1, max_order=k,
                                          significance_threshold=significance_threshold)[0]:
                accepted = k
        return accepted

    def to_dbgnn_data(self, max_order: int = 2, mapping: str = "last", x: torch.Tensor | None = None,
                      x_h: torch.Tensor | None = None) -> Data:
        """Input bundle of :class:`pathpyg_amd.nn.DBGNN` (reference multi_order_model.py:511-554).

        Like the reference, node features default to one-hot matrices (``torch.eye``) — usable for small
        graphs only; pass ``x`` / ``x_h`` (``[N, F]`` / ``[U_k, F]``) for anything large."""
        if max_order not in self.layers:
            logger.error("Higher-order graph of specified order not found.")
            raise ValueError(f"Higher-order graph of order {max_order} not found.")
        g = self.layers[1]
        g_ho = self.layers[max_order]
        n, n_ho = g.data.num_nodes, g_ho.data.num_nodes
        built = self._fused_plans(max_order, mapping)
        dev = built.fo.fwd_ptr.device if built is not None else g.data.edge_index.device
        x_eye = x_h_eye = False
        if x is None:
            x_eye = g.data.x is None
            x = g.data.x if g.data.x is not None else torch.eye(n, n, device=dev)
        if x_h is None:
            x_h_eye = True
            x_h = torch.eye(n_ho, n_ho, device=dev if built is not None else g_ho.data.edge_index.device)
        if built is not None:
            # The layers came out of the order-2 builder together with the plans DBGNN.forward needs (GCN normalisation of both graphs, the
            # bipartite "last" grouping): the bundle hands the plans over and keeps the reference's tensors as deferred views of the layers —
            # a training step that only runs the model never materialises [2, A2] indices (reference multi_order_model.py:511-554 builds
            # them eagerly; PyG's GCNConv re-normalises on every forward)
            u2, a2 = int(n_ho), int(built.sizes["A2"])
            # (the makers hold the layers' tensors / Lazy objects AS THEY ARE NOW, not the bags: a layer edited after bundling does not leak into a
            #  bundle whose plans were accepted for the tensors of this moment)
            ei1, ei2, w2 = g.data.peek("edge_index"), g_ho.data.peek("edge_index"), g_ho.data.peek("edge_weight")

            def now(v):
                return v.resolve() if isinstance(v, Lazy) else v

            out = Data(
                num_nodes=n,
                num_ho_nodes=n_ho,
                x=x,
                x_h=x_h,
                edge_index=Lazy(lambda: now(ei1), (2, u2)),
                edge_index_higher_order=Lazy(lambda: now(ei2), (2, a2)),
                edge_weights=g.data.edge_weight.float(),
                edge_weights_higher_order=Lazy(lambda: now(w2).float(), (a2,)),
                bipartite_edge_index=Lazy(lambda: generate_bipartite_edge_index(g, g_ho, mapping=mapping, device=dev), (2, u2)),
                y=g.data.y,
            )
            names = ("edge_index", "edge_weights", "edge_index_higher_order", "edge_weights_higher_order", "bipartite_edge_index")
            stamp = {name: out.peek(name) for name in names}
            object.__setattr__(out, "_pp_plans", {"fo": built.fo, "ho": built.ho, "bi": built.bip, "stamp": stamp,
                                                  "versions": {k: v._version for k, v in stamp.items() if isinstance(v, torch.Tensor)}})
            object.__setattr__(out, "_pp_hints", {"stamp": (), "x_eye": (out.x, out.x._version) if x_eye else None,
                                                  "x_h_eye": (out.x_h, out.x_h._version) if x_h_eye else None})
            return out
        out = Data(
            num_nodes=n,
            num_ho_nodes=n_ho,
            x=x,
            x_h=x_h,
            edge_index=g.data.edge_index,
            edge_index_higher_order=g_ho.data.edge_index,
            edge_weights=g.data.edge_weight.float(),
            edge_weights_higher_order=g_ho.data.edge_weight.float(),
            bipartite_edge_index=generate_bipartite_edge_index(g, g_ho, mapping=mapping, device=dev),
            y=g.data.y,
        )
        # facts DBGNN.forward would otherwise have to verify on the device (every Graph's edge index is row-sorted)
        object.__setattr__(out, "_pp_hints", {
            # the hints describe exactly these tensor objects at these in-place versions; DBGNN.forward ignores them otherwise
            "stamp": tuple((t, t._version) for t in (out.edge_index, out.edge_weights, out.edge_index_higher_order,
                                                     out.edge_weights_higher_order, out.bipartite_edge_index) if t is not None),
            # the default one-hot features: DBGNN's first layers then read W^T through the CSR instead of multiplying by an n x n identity
            "x_eye": (out.x, out.x._version) if x_eye else None, "x_h_eye": (out.x_h, out.x_h._version) if x_h_eye else None,
            "rows_sorted": True, "bipartite_sources_sorted": mapping in ("last", "first"),
            # temporal models: the order-2 nodes ARE the first-order graph's edges, in its edge order (_LiftChain.to_second_order) -
            # DBGNN.forward then derives the "last" bipartite plan from the first-order plan's destination grouping, no extra sort
            "bipartite_is_fo_edge_heads": bool(mapping == "last" and max_order == 2 and getattr(g_ho, "_nodes_are_fo_edges", False)
                                               and n_ho == g.data.edge_index.size(1))})
        return out


    def _fused_plans(self, max_order: int, mapping: str):
        """The builder result behind ``layers[1]`` / ``layers[2]`` when both still are what :func:`_second_order_fused` made (same objects,
        edge tensors neither replaced nor edited in place) and the bundle asked for is the one its plans describe; else ``None``."""
        rec = getattr(self, "_pp_fused", None)
        if rec is None or max_order != 2 or mapping != "last":
            return None
        built, layers = rec
        for k, (graph, lazies) in layers.items():
            if self.layers.get(k) is not graph:
                return None
            for name, made in lazies.items():
                cur = graph.data.peek(name)
                if isinstance(made, Lazy):
                    if not (cur is made or (made.value is not None and cur is made.value and cur._version == made.version)):
                        return None
                elif not (cur is made[0] and cur._version == made[1]):
                    return None
        return built


def _hip_unit() -> str:
    from .._hip import UNIT
    return UNIT


def _csr_rows(ptr: torch.Tensor, total: int) -> torch.Tensor:
    """Row id of every entry of a CSR with int32 row pointers ``ptr`` (int64 [total])."""
    n = ptr.numel() - 1
    return torch.repeat_interleave(torch.arange(n, device=ptr.device), (ptr[1:] - ptr[:-1]).long(), output_size=total)


def _second_order_fused(g: TemporalGraph, delta, weight: str, cached: bool):
    """``from_temporal_graph(g, delta, max_order=2)`` on the fused order-2 builder (``_hip.debruijn2`` -> ``pp_debruijn2_count / _fill``): ONE pass
    over the stream yields both layers as CSR plans — no event graph, no ``[E_2, 2]`` instance tensors, one read-back.  The reference's layer
    tensors (reference multi_order_model.py:153-181: ``edge_index``, ``edge_weight``, ``node_sequence``, ``inverse_idx`` of both layers) are
    :class:`~pathpyg_amd.data.Lazy` views of those plans, identical to what the generic kernels produce, made when somebody reads them.
    ``None``: the builder does not apply (host-resident stream, a weight attribute that is not float32, an unsorted stream) or is not the one to use
    (a contact-shaped stream, ``_hip.debruijn2_wanted``)."""
    from .. import _hip
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    time = data.time
    if ei is None or time is None or not ei.is_cuda or not time.is_cuda:
        return None
    w = None
    if weight in data:
        w = data[weight]
        if not isinstance(w, torch.Tensor) or w.dtype != torch.float32 or not w.is_cuda:
            return None
    n, m_events = int(data.num_nodes), int(ei.size(1))
    if n == 0 or m_events == 0 or not _hip.debruijn2_wanted(m_events, n):
        return None
    if not data.is_sorted_by_time():          # (unsorted: one cheap kernel instead of the builder's whole count pass)
        return None
    built = _hip.debruijn2(ei, time, n, delta, w, want_weights=True, unsorted_ok=True)
    if built is None:
        return None
    dev = ei.device
    fo, ho = built.fo, built.ho
    u2, a2 = int(built.sizes["U2"]), int(built.sizes["A2"])

    # (the makers below capture plans and other Lazy objects, never the Data bags that hold them: a bag -> Lazy -> closure -> bag cycle would keep
    #  ~4 GB of plans per model alive until the cyclic collector runs — measured: 14.0 -> 20.8 ms per API step from the 11th step on)
    ei1 = Lazy(lambda: torch.stack((_csr_rows(fo.bwd_ptr, u2), fo.bwd_idx.long())), (2, u2))
    lazy1 = {"edge_index": ei1,
             "node_sequence": Lazy(lambda: torch.arange(n, device=dev).unsqueeze(1), (n, 1)),
             "inverse_idx": Lazy(lambda: torch.arange(n, device=dev), (n,))}
    d1 = Data(edge_index=ei1, num_nodes=n, node_sequence=lazy1["node_sequence"], edge_weight=built.fo_weight, inverse_idx=lazy1["inverse_idx"])
    g1 = Graph._from_parts(d1, g.mapping)
    ei2 = Lazy(lambda: torch.stack((_csr_rows(ho.bwd_ptr, a2), ho.bwd_idx.long())), (2, a2))
    fwd_weight = built.ho_fwd_weight

    def weights2():
        # the builder keeps the merged weights in destination-major order (contiguous stores); the layer lists its edges source-major
        e2 = ei2.resolve()
        key_fwd = _csr_rows(ho.fwd_ptr, a2) * u2 + ho.fwd_idx.long()                       # ascending: rows ascending, sources ascending inside a row
        return fwd_weight[torch.searchsorted(key_fwd, e2[1] * u2 + e2[0])]

    def inverse2():
        # order-2 node of every event = its merged first-order edge (lift_order.py:133 on the [m, 2] instance rows)
        e1 = ei1.resolve()
        return torch.searchsorted(e1[0] * n + e1[1], ei[0] * n + ei[1])

    lazy2 = {"edge_index": ei2,
             "node_sequence": Lazy(lambda: ei1.resolve().t().contiguous(), (u2, 2)),
             "edge_weight": Lazy(weights2, (a2,)),
             "inverse_idx": Lazy(inverse2, (m_events,))}
    d2 = Data(edge_index=ei2, num_nodes=u2, node_sequence=lazy2["node_sequence"], edge_weight=lazy2["edge_weight"],
              inverse_idx=lazy2["inverse_idx"])
    g2 = Graph._from_parts(d2, IndexMap.from_node_sequence(g.mapping, lazy2["node_sequence"]))
    g2._nodes_are_fo_edges = True
    out = MultiOrderModel()
    if cached:
        out.layers[1] = g1
    out.layers[2] = g2
    if cached:
        out._pp_fused = (built, {1: (g1, {"edge_index": lazy1["edge_index"], "edge_weight": (built.fo_weight, built.fo_weight._version)}),
                                 2: (g2, {"edge_index": lazy2["edge_index"], "edge_weight": lazy2["edge_weight"]})})
    out.sizes = dict(built.sizes)
    return out


LARGE_HUB_EVENTS = 65536
LIFT_ONLY_ORDER2 = False     # from_temporal_graph(max_order=2) on a stream with a node of LARGE_HUB_EVENTS events: True = the level-by-level builder (no GCN plans)


def _has_large_hub(g: TemporalGraph) -> bool:
    """Whether some node of a device-resident stream has ``LARGE_HUB_EVENTS`` or more in- or out-events (two histograms and one read-back, kept
    on the graph object for the tensors they were taken from)."""
    from .. import _hip
    ei = _dispatch.plain(g.data.edge_index)
    if ei is None or not ei.is_cuda or ei.numel() == 0 or ei.size(1) < LARGE_HUB_EVENTS:
        return False
    key = (id(ei), ei._version, int(g.data.num_nodes))
    kept = getattr(g, "_pp_max_degree", None)
    if kept is None or kept[0] != key:
        n = int(g.data.num_nodes)
        ei = ei.contiguous()
        most = int(torch.maximum(_hip.degree(ei[0], n).max(), _hip.degree(ei[1], n).max()))
        kept = g._pp_max_degree = (key, most)
    return kept[1] >= LARGE_HUB_EVENTS


def _multi_order_fused(g: TemporalGraph, delta, max_order: int, weight: str, cached: bool, event_graph=None):
    """``from_temporal_graph(g, delta, max_order >= 3)`` level by level (``_hip.multi_order_temporal`` -> ``pp_multiorder_prepare`` / ``_step``): every
    layer comes out as source-major CSR with merged weights; the reference's layer tensors (reference multi_order_model.py:153-191) are
    :class:`~pathpyg_amd.data.Lazy` views — ``edge_index`` = (row of every CSR entry, column), ``node_sequence`` of layer k = the sequence of the
    entry's row in layer k-1 followed by the entry's last node (De Bruijn property: the nodes of layer k ARE the edges of layer k-1),
    ``inverse_idx`` of the layers from 3 on through the generic kernels (it numbers the reference's instance graph, which this builder never makes).
    ``event_graph``: a given ``lift_order_temporal(g, delta)`` — its edges are the continuation windows (``pp_multiorder_prepare_graph``).
    ``None``: the builder does not apply (host-resident or unsorted stream, a weight attribute that is not float32, a layer without edges, a node
    sequence with more than 4096 continuations, 2^31 instances, an event graph that is not sorted by source)."""
    from .. import _hip
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    time = data.time
    if ei is None or time is None or not ei.is_cuda or not time.is_cuda:
        return None
    w = None
    if weight in data:
        w = data[weight]
        if not isinstance(w, torch.Tensor) or w.dtype != torch.float32 or not w.is_cuda:
            return None
    n, m_events = int(data.num_nodes), int(ei.size(1))
    if n == 0 or m_events == 0 or not data.is_sorted_by_time():
        return None
    if event_graph is not None:
        event_graph = _dispatch.plain(event_graph)
        if not isinstance(event_graph, torch.Tensor) or not event_graph.is_cuda or event_graph.dim() != 2 or event_graph.size(0) != 2:
            return None
    built = _hip.multi_order_temporal(ei, time, n, delta, w, max_order, event_graph=event_graph)
    if built is None:
        return None
    dev = ei.device
    out = MultiOrderModel()
    out.sizes = {"m": m_events, "N": n, "layers": [(b.n_nodes, b.n_edges, b.n_instances) for b in built]}
    prev_index = prev_seq = None
    for k, b in enumerate(built, start=1):
        keep = cached or k == max_order

        def index_of(b=b):
            return torch.stack((_csr_rows(b.row_ptr, b.n_edges), b.col.long()))

        index = Lazy(index_of, (2, b.n_edges))
        if k == 1:
            seq = Lazy(lambda: torch.arange(n, device=dev).unsqueeze(1), (n, 1))
            inverse = Lazy(lambda: torch.arange(n, device=dev), (n,))
        else:
            def seq_of(prev_index=prev_index, prev_seq=prev_seq, last=built[k - 2].last):
                # node u of layer k = edge u of layer k-1: the sequence of that edge's source node, then the edge's last node
                return _dispatch.gather_concat(prev_seq.resolve(), prev_index.resolve()[0].contiguous(), last.long())

            seq = Lazy(seq_of, (b.n_nodes, k))
            if k == 2:
                def inverse_of(first=prev_index):
                    # order-2 node of every event = its merged first-order edge (lift_order.py:133 on the [m, 2] instance rows)
                    e1 = first.resolve()
                    return torch.searchsorted(e1[0] * n + e1[1], ei[0] * n + ei[1])
            else:
                def inverse_of(k=k):
                    # the reference numbers the order-k INSTANCES (edges of the order-(k-1) instance graph, lexicographic): only the generic
                    # kernels make that graph
                    return MultiOrderModel._from_temporal_graph_generic(g, delta, k, weight, False, event_graph).layers[k].data.inverse_idx

            inverse = Lazy(inverse_of, (built[k - 2].n_instances,))
        if keep:
            d = Data(edge_index=index, num_nodes=b.n_nodes, node_sequence=seq, edge_weight=b.weight, inverse_idx=inverse)
            mapping = g.mapping if k == 1 else IndexMap.from_node_sequence(g.mapping, seq)
            out.layers[k] = Graph._from_parts(d, mapping)
            if k == 2:
                out.layers[k]._nodes_are_fo_edges = True
        prev_index, prev_seq = index, seq
    return out


class _LiftChain:
    """State of the order-k instance graph while climbing orders, WITHOUT per-instance node sequences.

    The reference extends an ``[instances, k]`` node-sequence tensor at every order and runs ``torch.unique(dim=0)``
    on it (multi_order_model.py:114, lift_order.py:133).  The same node numbering follows from two vectors per
    order: ``inv[i]`` = De Bruijn node of instance i (its lexicographic rank) and ``last[i]`` = its last first-order
    node.  An order-(k+1) instance is an edge (a -> b) of the order-k instance graph; its node sequence is
    ``seq(a) ++ last[b]``, so its lexicographic rank is the rank of the PAIR ``(inv[a], last[b])`` — a 2-column unique
    instead of a (k+1)-column one — and the distinct sequences are ``unique_k[first] ++ second`` for the distinct pairs.
    When layer k has been aggregated even that sort is unnecessary: the pair ``(inv[a], last[b])`` determines and is determined by
    the layer-k edge ``(inv[a], inv[b])``, so the order-(k+1) nodes are layer k's merged edges in coalesce order and the inverse map
    of that coalesce (``edge_ids``) is the next ``inv``; the 2-column unique remains for ``cached=False`` (layer k not built).
    """

    def __init__(self, index, inv, last, unique_nodes, weight, graph, edge_ids=None):
        self.index = index              # [2, E_k] edges between order-k instances (source-sorted)
        self.inv = inv                  # [M_k] instance -> De Bruijn node id
        self.last = last                # [M_k] last first-order node of every instance
        self.unique_nodes = unique_nodes  # [U_k, k]
        self.weight = weight            # [E_k] or None
        self.graph = graph              # aggregated layer k (or None when not saved)
        # [E_k] position of every instance edge's merged edge in layer k (the inverse map of its coalesce) or None.  De Bruijn property:
        # the order-(k+1) nodes ARE the merged edges of layer k, in the same lexicographic order - so this is the next order's `inv`,
        # and the next order needs no unique over its node sequences at all (reference: torch.unique(dim=0), lift_order.py:133)
        self.edge_ids = edge_ids

    @staticmethod
    def first_order(edge_index, node_sequence, edge_weight, identity_nodes: bool, num_first_order, want_pairs: bool):
        from ..algorithms.lift_order import _aggregate_with_known_nodes
        if identity_nodes:       # temporal graphs: node_sequence = arange(N): every node distinct, already ordered
            unique_nodes = node_sequence
            inv = torch.arange(num_first_order, device=edge_index.device)
        else:
            unique_nodes, inv = _dispatch.unique_rows(node_sequence)
        out = _aggregate_with_known_nodes(edge_index, 1, node_sequence, unique_nodes, inv, edge_weight, "sum", want_inverse=want_pairs)
        graph, edge_ids = out if want_pairs else (out, None)
        if not identity_nodes and edge_ids is not None:
            # Layer 1 of a path model uses the walk node ids AS GIVEN (reference quirk, lift_order.py:135-136), while `unique_nodes` is
            # indexed by RANK.  The shortcut of lift() (order-(k+1) nodes = layer k's merged edges) looks layer-1 edges up in
            # `unique_nodes`, which is only right when id == rank, i.e. the ids are exactly 0..U-1; otherwise the next lift takes the
            # 2-column unique over (rank of prefix, last node), which is id-agnostic.
            u = unique_nodes.size(0)
            ids = _dispatch.plain(unique_nodes).reshape(-1)
            if u and not bool((ids == torch.arange(u, device=ids.device)).all()):
                edge_ids = None
        return _LiftChain(edge_index, inv, _dispatch.plain(node_sequence).reshape(-1), unique_nodes, edge_weight, graph, edge_ids)

    def to_second_order(self, event_index, ho_index, ho_weight, save: bool, want_edge_ids: bool = False):
        """Temporal special case: the order-2 instances are the events themselves, their distinct (src, dst) pairs are
        layer 1's merged edges (already sorted), and ``edge_ids`` from layer 1's coalesce is their inverse map."""
        from ..algorithms.lift_order import _aggregate_with_known_nodes
        unique_nodes = self.graph.data.edge_index.t().contiguous()
        inv = self.edge_ids
        # a successor of node (a, b) is a node (b, c): all of them sit in the contiguous id block of the pairs that start with b
        merged = _dispatch.plain(self.graph.data.edge_index)
        blocks = _dispatch.successor_blocks(merged[0], self.unique_nodes.size(0), merged[1]) if save and merged.size(1) else None
        graph = edge_ids = None
        if save:
            out = _aggregate_with_known_nodes(ho_index, 2, None, unique_nodes, inv, ho_weight, "sum", col_block=blocks, want_inverse=want_edge_ids)
            graph, edge_ids = out if want_edge_ids else (out, None)
            graph._nodes_are_fo_edges = True          # node u of this layer = edge u of layer 1 (same lexicographic order)
        return _LiftChain(ho_index, inv, _dispatch.plain(event_index)[1], unique_nodes, ho_weight, graph, edge_ids)

    def lift(self, aggr: str, save: bool, want_edge_ids: bool = False):
        from ..algorithms.lift_order import _aggregate_with_known_nodes
        num_instances = self.inv.numel()
        if self.weight is None or (isinstance(self.weight, str) and aggr in ("src", "dst", "max", "mul")):      # (unit weights stay unit)
            ho_index, weight = lift_order_edge_index(self.index, num_nodes=num_instances), self.weight
        else:
            w_inst = self.weight
            if isinstance(w_inst, str):            # the UNIT marker under an aggregation that does not preserve it ("add": 1 + 1 = 2): the ones vector
                w_inst = torch.ones(num_instances, dtype=torch.float32, device=_dispatch.plain(self.index).device)      # the reference starts from
            ho_index, weight = lift_order_edge_index_weighted(self.index, w_inst, num_nodes=num_instances, aggr=aggr)
        last = aggregate_node_attributes(self.index, self.last, "dst")                 # last node of every new instance
        k = self.unique_nodes.size(1)
        if self.graph is not None and self.edge_ids is not None:
            # the new nodes ARE layer k's merged edges (same lexicographic order): sequence = prefix node's sequence ++ last node of the
            # suffix node; the coalesce inverse of layer k numbers the new instances - no sort of the instances' node sequences
            merged = _dispatch.plain(self.graph.data.edge_index)
            inv = self.edge_ids
            prefix, suffix = merged[0].contiguous(), merged[1].contiguous()
            tail = _dispatch.plain(self.unique_nodes)[:, -1].contiguous()
            unique_nodes = _dispatch.gather_concat(self.unique_nodes, prefix, aggregate_node_attributes(merged, tail, "dst"))
        else:
            pairs = _dispatch.gather_concat(self.inv.unsqueeze(1), _dispatch.plain(self.index)[0], last)     # (inv[a], last[b])
            hi = max(self.unique_nodes.size(0), int(_dispatch.minmax(self.unique_nodes)[1]) + 1 if self.unique_nodes.numel() else 1)
            unique_pairs, inv = _dispatch.unique_rows(pairs, (0, max(hi - 1, 0)))
            unique_nodes = _dispatch.gather_concat(self.unique_nodes, unique_pairs[:, 0], unique_pairs[:, 1])
            prefix = unique_pairs[:, 0].contiguous()
            suffix = None
            if save and inv.numel():
                # every successor of a new node P starts with P's order-k SUFFIX node, i.e. the order-k node of the second instance of
                # any edge a -> b that realises P
                suffix = torch.empty(unique_pairs.size(0), dtype=torch.int64, device=inv.device)
                suffix[inv] = self.inv[_dispatch.plain(self.index)[1]]
        graph = edge_ids = None
        if save:
            # the new nodes are numbered by (order-k prefix node, last node): all successors of P sit in the id block of P's suffix node
            blocks = _dispatch.successor_blocks(prefix, self.unique_nodes.size(0), suffix) if inv.numel() else None
            out = _aggregate_with_known_nodes(ho_index, k + 1, None, unique_nodes, inv, weight, "sum", col_block=blocks, want_inverse=want_edge_ids)
            graph, edge_ids = out if want_edge_ids else (out, None)
        return _LiftChain(ho_index, inv, last, unique_nodes, weight, graph, edge_ids)


def _dispatch_degree(index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Out-degree histogram as a float tensor on the input's device (PyG ``degree``, multi_order_model.py:220)."""
    from .. import _hip
    dev = _dispatch.compute_device(index)
    deg = _hip.degree(_dispatch.plain(index).to(dev).contiguous(), num_nodes)
    return deg.to(torch.float32).to(index.device)
