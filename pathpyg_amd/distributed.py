"""Multi-GPU sharding of the lift: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL on ROCm).

The reference is single-process; this is new design following SURVEY §8(e):

* temporal lift — **edge-range shards with a forward halo, no data exchange**.  Events are time-sorted, so the
  continuations of the source events ``[lo, hi)`` all lie in ``[lo+1, halo_end)`` with
  ``halo_end = first event with t > t[hi-1] + delta``.  Rank ``r`` lifts its slice ``[lo_r, halo_end_r)`` with only
  the first ``hi_r - lo_r`` events acting as sources (``n_own``) and global ids restored by ``id_offset``; the
  concatenation of the per-rank results in rank order IS the global lexicographic result.  The only collective is
  one all-gather of the per-rank pair counts (8 bytes per rank) for global output offsets.
* ranges are balanced on OUTPUT size when a per-event count estimate is supplied, else on event count.

Everything below the collectives runs through ``pathpyg_amd._dispatch`` (HIP kernels); tests exercise the
planner and the collectives on CPU with the ``gloo`` backend by substituting the local lift.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _dispatch


def _world(group=None) -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def event_ranges(num_events: int, world_size: int, weights: torch.Tensor | None = None) -> list[tuple[int, int]]:
    """Contiguous source-event ranges ``[(lo, hi)] * world_size`` covering ``[0, num_events)``.

    Without ``weights`` the events are split evenly; with per-event ``weights`` (e.g. continuation counts from a
    counting pre-pass) the split equalises the weight per rank (each cut is the first prefix-sum crossing)."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if weights is None or num_events == 0:
        cuts = [(num_events * r) // world_size for r in range(world_size + 1)]
    else:
        if weights.numel() != num_events:
            raise ValueError("weights must hold one value per event")
        prefix = torch.cumsum(weights.to(torch.float64).cpu(), 0)
        total = float(prefix[-1])
        targets = torch.tensor([total * r / world_size for r in range(1, world_size)], dtype=torch.float64)
        inner = torch.searchsorted(prefix, targets, right=False).tolist() if world_size > 1 else []
        cuts = [0] + [min(int(c) + 1, num_events) for c in inner] + [num_events]
        for i in range(1, len(cuts)):                       # keep the cuts monotone
            cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def halo_end(time: torch.Tensor, hi: int, delta) -> int:
    """First event id after ``hi - 1`` that can no longer continue any source event < ``hi``:
    events are time-sorted, so it is the first id with ``t > t[hi-1] + delta`` (evaluated like temporal.py:43).
    A conservative (never too small) bound is all correctness needs; the kernel re-checks every pair."""
    m = int(time.numel())
    if hi <= 0 or hi >= m:
        return m if hi > 0 else 0
    last = time[hi - 1].cpu()
    thr = last + torch.tensor(delta)
    cmp_dtype = torch.result_type(time, thr)
    # searchsorted on a tiny CPU copy would need the whole array; a bisection with O(log m) scalar reads is enough
    lo, up = hi, m
    thr_c = thr.to(cmp_dtype)
    while lo < up:
        mid = (lo + up) // 2
        if bool(time[mid].to(cmp_dtype).cpu() <= thr_c):
            lo = mid + 1
        else:
            up = mid
    return lo


def lift_order_temporal_sharded(g, delta=1, group=None, weights: torch.Tensor | None = None):
    """This rank's part of ``lift_order_temporal(g, delta)`` for a stream replicated on every rank.

    Returns ``(local_index [2, E_r] int64 with GLOBAL event ids, global_offset, global_total)``: rank ``r``'s block
    occupies columns ``[global_offset, global_offset + E_r)`` of the full lexicographic result."""
    rank, world = _world(group)
    data = g.data
    m = int(data.edge_index.size(1))
    lo, hi = event_ranges(m, world, weights)[rank]
    end = halo_end(data.time, hi, delta) if hi > lo else lo
    ei = _dispatch.plain(data.edge_index)[:, lo:end].contiguous()
    t = data.time[lo:end].contiguous()
    local = _dispatch.temporal_lift(ei, t, int(data.num_nodes), delta, n_own=hi - lo, id_offset=lo)
    counts = torch.tensor([local.size(1)], dtype=torch.int64, device=local.device)
    if world > 1:
        gathered = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts, group=group)
        sizes = [int(c.item()) for c in gathered]
    else:
        sizes = [int(counts.item())]
    return local, sum(sizes[:rank]), sum(sizes)


def lift_order_edge_index_sharded(edge_index: torch.Tensor, num_nodes: int, group=None, weights: torch.Tensor | None = None):
    """Edge-range sharded line-graph lift of a source-sorted edge list replicated on every rank: rank r expands
    the edges ``[lo_r, hi_r)``; out-degrees and row pointers are derived locally from the replicated list, so the only collective
    is the all-gather of the per-rank pair counts.  Returns ``(local [2, E'_r] with global ids, ranges, total E')``; the rank-order
    concatenation of the blocks is the single-process result."""
    rank, world = _world(group)
    ei = _dispatch.plain(edge_index)
    ranges = event_ranges(ei.size(1), world, weights)
    lo, hi = ranges[rank]
    local = _dispatch.linegraph_lift(ei, num_nodes, (lo, hi))
    total = local.size(1)
    if world > 1:
        counts = torch.tensor([total], dtype=torch.int64, device=local.device)
        dist.all_reduce(counts, group=group)
        total = int(counts.item())
    return local, ranges, total


def gather_lifted(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate the per-rank blocks (rank order) on every rank — for tests and small graphs only."""
    rank, world = _world(group)
    if world == 1:
        return local
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.size(1)], dtype=torch.int64, device=local.device), group=group)
    cap = max(int(s.item()) for s in sizes)
    padded = torch.zeros((2, cap), dtype=torch.int64, device=local.device)
    padded[:, : local.size(1)] = local
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:, : int(s.item())] for p, s in zip(parts, sizes)], dim=1)


def all_reduce_gradients(module: torch.nn.Module, group=None, average: bool = True) -> None:
    """Sum (``average=False``) or average the (small) DBGNN weight gradients across ranks in ONE flattened all-reduce
    (latency-bound: ~20 k floats; per-tensor calls would pay the xGMI launch latency 14 times)."""
    _, world = _world(group)
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    if average:
        flat /= world
    at = 0
    for g in grads:
        g.copy_(flat[at: at + g.numel()].view_as(g))
        at += g.numel()


# =====================================================================================================
# Destination-partitioned DBGNN: rank r owns a contiguous slice of the DESTINATION rows of the
# first-order graph, of the higher-order graph and of the bipartite map, i.e. the rows of every feature matrix.
# Per propagation: all-gather of the transformed features H (every rank needs the source rows its edges point
# to), local atomics-free CSR aggregation of the owned rows, and in the backward pass a reduce-scatter of the
# source-row gradients.  Weight gradients are averaged with one flattened all-reduce.  xGMI is point-to-point:
# the row all-gather moves N*F*4 bytes per layer in total, each rank receiving (R-1)/R of it over its 7 links.
# =====================================================================================================
from . import _hip  # noqa: E402


def node_ranges(num_nodes: int, world_size: int) -> list[tuple[int, int]]:
    return [((num_nodes * r) // world_size, (num_nodes * (r + 1)) // world_size) for r in range(world_size)]


def _gather_rows(x_local: torch.Tensor, ranges, group) -> torch.Tensor:
    """Concatenate the row slices of all ranks (rank order) -> [N, F]."""
    world = len(ranges)
    if world == 1:
        return x_local
    cap = max(hi - lo for lo, hi in ranges)
    padded = x_local.new_zeros((cap, x_local.size(1)))
    padded[: x_local.size(0)] = x_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, ranges)], dim=0)


def _reduce_scatter_rows(x_full: torch.Tensor, ranges, rank: int, group) -> torch.Tensor:
    """Sum the [N, F] partials of all ranks and keep this rank's row slice."""
    world = len(ranges)
    lo, hi = ranges[rank]
    if world == 1:
        return x_full[lo:hi]
    backend = dist.get_backend(group)
    if backend == "nccl":
        cap = max(b - a for a, b in ranges)
        chunks = []
        for a, b in ranges:
            c = x_full.new_zeros((cap, x_full.size(1)))
            c[: b - a] = x_full[a:b]
            chunks.append(c)
        out = torch.empty_like(chunks[0])
        dist.reduce_scatter(out, chunks, group=group)
        return out[: hi - lo].contiguous()
    summed = x_full.clone()                      # gloo has no reduce_scatter: all-reduce and slice (tests only)
    dist.all_reduce(summed, group=group)
    return summed[lo:hi].contiguous()


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, ranges, rank, group):
        ctx.ranges, ctx.rank, ctx.group = ranges, rank, group
        return _gather_rows(x_local, ranges, group)

    @staticmethod
    def backward(ctx, d_full):
        return _reduce_scatter_rows(d_full.contiguous(), ctx.ranges, ctx.rank, ctx.group), None, None, None


class _LocalPropagate(torch.autograd.Function):
    """y_local = act(A_local x_full + self_coef * s_local + bias): rows = this rank's destinations, columns = all sources."""

    @staticmethod
    def forward(ctx, plan, x_full, s_local, bias, act: bool):
        y = _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, x_full, plan.self_coef, s_local, bias, act, heavy=plan.fwd_heavy)
        ctx.plan, ctx.act, ctx.has_bias = plan, act, bias is not None
        ctx.save_for_backward(y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        plan = ctx.plan
        (y,) = ctx.saved_tensors
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.act or need_b:
            dpre, dbias = _hip.act_backward(dy, y, ctx.act, want_dpre=ctx.act, want_dbias=need_b)
            if not ctx.act:
                dpre = dy.contiguous()
        else:
            dpre, dbias = dy.contiguous(), None
        dx_full = _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, heavy=plan.bwd_heavy) if ctx.needs_input_grad[1] else None
        ds = _hip.scale_rows(dpre, plan.self_coef) if ctx.needs_input_grad[2] else None
        return None, dx_full, ds, dbias, None


def _rect_plan(src_global, dst_local, value, n_src: int, n_dst: int, self_coef):
    plan = _hip.bipartite_plan(torch.stack((src_global, dst_local)), n_src, n_dst, pair_value=value)
    plan.self_coef = self_coef
    return plan


def partition_gcn(edge_index: torch.Tensor, edge_weight: torch.Tensor, num_nodes: int, rank: int, world: int, group=None):
    """This rank's share of the GCN propagation of one graph: the edges pointing into its destination rows with their
    symmetric normalisation (PyG gcn_norm semantics: one self loop per node, weighted in-degree, d^-1/2)."""
    ranges = node_ranges(num_nodes, world)
    lo, hi = ranges[rank]
    src, dst = edge_index[0], edge_index[1]
    mine = (dst >= lo) & (dst < hi)
    src, dst, w = src[mine], dst[mine], edge_weight[mine].to(torch.float32)
    loop = src == dst
    loop_w = torch.ones(hi - lo, dtype=torch.float32, device=w.device)
    loop_w[dst[loop] - lo] = w[loop]                                     # an existing self loop keeps its weight
    src, dst, w = src[~loop], dst[~loop], w[~loop]
    deg = torch.zeros(hi - lo, dtype=torch.float32, device=w.device).index_add_(0, dst - lo, w) + loop_w
    dinv_local = deg.pow(-0.5)
    dinv_local[torch.isinf(dinv_local)] = 0
    dinv = _gather_rows(dinv_local.unsqueeze(1), ranges, group).squeeze(1)       # every rank needs d^-1/2 of all sources
    value = dinv[src] * w * dinv[dst]
    plan = _rect_plan(src, dst - lo, value, num_nodes, hi - lo, (dinv_local * loop_w * dinv_local).contiguous())
    return plan, ranges


def partition_bipartite(bipartite_index: torch.Tensor, n_ho: int, n_fo: int, rank: int, world: int):
    fo_ranges = node_ranges(n_fo, world)
    lo, hi = fo_ranges[rank]
    mine = (bipartite_index[1] >= lo) & (bipartite_index[1] < hi)
    plan = _hip.bipartite_plan(torch.stack((bipartite_index[0][mine], bipartite_index[1][mine] - lo)), n_ho, hi - lo)
    return plan


class ShardedDBGNN(torch.nn.Module):
    """Runs a :class:`pathpyg_amd.nn.DBGNN` with every graph partitioned by destination rows across the process group.

    ``prepare(data)`` slices the (replicated) input bundle for this rank and builds the rectangular CSR plans;
    ``forward(shard)`` returns the logits of the first-order nodes this rank owns.  Parameters are replicated; call
    :func:`all_reduce_gradients` after ``backward``."""

    def __init__(self, model, group=None):
        super().__init__()
        self.model = model
        self.group = group
        self.rank, self.world = _world(group)

    def prepare(self, data) -> dict:
        n_fo, n_ho = int(data.num_nodes), int(data.num_ho_nodes)
        plan_fo, fo_ranges = partition_gcn(data.edge_index, data.edge_weights, n_fo, self.rank, self.world, self.group)
        plan_ho, ho_ranges = partition_gcn(data.edge_index_higher_order, data.edge_weights_higher_order, n_ho, self.rank, self.world, self.group)
        plan_bi = partition_bipartite(data.bipartite_edge_index, n_ho, n_fo, self.rank, self.world)
        (flo, fhi), (hlo, hhi) = fo_ranges[self.rank], ho_ranges[self.rank]
        return {"plan_fo": plan_fo, "plan_ho": plan_ho, "plan_bi": plan_bi, "fo_ranges": fo_ranges, "ho_ranges": ho_ranges,
                "x": data.x[flo:fhi].contiguous(), "x_h": data.x_h[hlo:hhi].contiguous(),
                "y": None if data.y is None else data.y[flo:fhi], "n_fo": n_fo}

    def _gcn_stack(self, layers, x_local, plan, ranges):
        from .nn.dbgnn import dense
        for layer in layers:
            h_local = dense(x_local, layer.lin)
            h_full = _AllGatherRows.apply(h_local, ranges, self.rank, self.group)
            x_local = _LocalPropagate.apply(plan, h_full, h_local, layer.bias, True)
        return x_local

    def forward(self, shard: dict) -> torch.Tensor:
        from .nn.dbgnn import dense
        m = self.model
        x = self._gcn_stack(m.first_order_layers, shard["x"], shard["plan_fo"], shard["fo_ranges"])
        x_h = self._gcn_stack(m.higher_order_layers, shard["x_h"], shard["plan_ho"], shard["ho_ranges"])
        h_ho = _AllGatherRows.apply(dense(x_h, m.bipartite_layer.lin1), shard["ho_ranges"], self.rank, self.group)
        h_fo = dense(x, m.bipartite_layer.lin2)
        x = _LocalPropagate.apply(shard["plan_bi"], h_ho, h_fo, None, True)
        return dense(x, m.lin)

    def loss(self, shard: dict) -> torch.Tensor:
        """Cross-entropy over ALL first-order nodes: local sum divided by the global node count, so that summing the
        per-rank gradients (``all_reduce_gradients(..., average=False)``) reproduces the single-process gradient."""
        out = self.forward(shard)
        return torch.nn.functional.cross_entropy(out, shard["y"], reduction="sum") / shard["n_fo"]


# =====================================================================================================
# Distributed De Bruijn aggregation: global lexicographic unique / coalesce = a distributed sort with ONE
# exchange step per layer.  Keys are range-partitioned (rank r owns the keys in [cut_r, cut_{r+1})), every rank sends each key
# (+ weight) to its owner, the owner coalesces its range with the single-GPU kernels, and an all-gather of the per-rank counts
# turns local ranks into global ids.  Concatenating the ranks' outputs in rank order IS the single-process result.
# =====================================================================================================
def _exchange(buckets: list[torch.Tensor], group=None) -> torch.Tensor:
    """Send ``buckets[r]`` to rank r, return the concatenation of what the other ranks sent here (rank order).
    RCCL: one all_to_all of the sizes + one of the payload; gloo (tests) has no all_to_all: all-gather and pick."""
    rank, world = _world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return buckets[0]
    if dist.get_backend(group) == "nccl":
        send_sizes = torch.tensor([b.size(0) for b in buckets], dtype=torch.int64, device=buckets[0].device)
        recv_sizes = torch.empty_like(send_sizes)
        dist.all_to_all_single(recv_sizes, send_sizes, group=group)
        recv = list(torch.empty((int(recv_sizes.sum()),) + tuple(buckets[0].shape[1:]), dtype=buckets[0].dtype,
                                device=buckets[0].device).split(recv_sizes.tolist()))
        dist.all_to_all(recv, [b.contiguous() for b in buckets], group=group)
        return torch.cat(recv, dim=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, [b.cpu() for b in buckets], group=group)
    return torch.cat([gathered[src][rank] for src in range(world)], dim=0).to(buckets[0].device)


def _owner_cuts(num_keys_space: int, world: int) -> torch.Tensor:
    """Equal-width key ranges over ``[0, num_keys_space)``: ``cuts[r] .. cuts[r+1]`` is owned by rank r."""
    return torch.tensor([(num_keys_space * r) // world for r in range(world + 1)], dtype=torch.int64)


def coalesce_sharded(rows: torch.Tensor, cols: torch.Tensor, weight: torch.Tensor, num_nodes: int, group=None):
    """Global ``coalesce`` of edges scattered over the ranks.  Rank r ends up with the distinct edges whose ROW lies in
    ``[cuts[r], cuts[r+1])`` (sorted by (row, col), weights summed); returns ``(edge_index [2, A_r], weight [A_r], cuts)``."""
    rank, world = _world(group)
    cuts = _owner_cuts(num_nodes, world).to(rows.device)
    owner = torch.searchsorted(cuts[1:].contiguous(), rows, right=True).clamp_(max=world - 1)
    payload = torch.stack((rows, cols, weight.to(torch.float64).view(torch.int64) if weight.dtype == torch.float64
                           else weight.to(torch.float32).view(torch.int32).to(torch.int64)), dim=1)
    buckets = [payload[owner == r] for r in range(world)]
    mine = _exchange(buckets, group)
    ei = mine[:, :2].t().contiguous()
    w = mine[:, 2].to(torch.int32).view(torch.float32) if weight.dtype != torch.float64 else mine[:, 2].view(torch.float64)
    merged_index, merged_weight = _dispatch.coalesce(ei, w.contiguous(), num_nodes, "sum")
    return merged_index, merged_weight, cuts.cpu()


def unique_pairs_sharded(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, group=None):
    """Global lexicographic numbering of the distinct (src, dst) pairs of events scattered over the ranks (= the order-2
    De Bruijn nodes).  Returns ``(all_pairs [U, 2] on every rank, ids)`` with ``ids[e]`` the global id of local event e."""
    rank, world = _world(group)
    ones = torch.ones(src.numel(), dtype=torch.float32, device=src.device)
    local_pairs, _, _ = coalesce_sharded(src, dst, ones, num_nodes, group)          # my key range, sorted, distinct
    counts = torch.tensor([local_pairs.size(1)], dtype=torch.int64, device=src.device)
    if world > 1:
        sizes = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(sizes, counts, group=group)
        cap = max(int(s.item()) for s in sizes)
        padded = torch.zeros((2, cap), dtype=torch.int64, device=src.device)
        padded[:, : local_pairs.size(1)] = local_pairs
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        all_pairs = torch.cat([p[:, : int(s.item())] for p, s in zip(parts, sizes)], dim=1)
    else:
        all_pairs = local_pairs
    keys = all_pairs[0] * num_nodes + all_pairs[1]                                   # ascending: ranks own ascending row ranges
    ids = torch.searchsorted(keys.contiguous(), (src * num_nodes + dst).contiguous())
    return all_pairs.t().contiguous(), ids


def second_order_layer_sharded(g, delta=1, group=None, edge_weight: torch.Tensor | None = None) -> dict:
    """Order-2 De Bruijn layer of a temporal graph replicated on every rank, computed cooperatively: edge-range lift (no
    exchange), global pair numbering (one exchange), global coalesce of the lifted pairs (one exchange).
    Rank r returns its slice of the layer: the aggregated edges whose source node id lies in its row range."""
    rank, world = _world(group)
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    n = int(data.num_nodes)
    w = edge_weight if edge_weight is not None else torch.ones(ei.size(1), device=ei.device)
    local, _, total = lift_order_temporal_sharded(g, delta, group)                 # (i, j) with global event ids
    lo, hi = event_ranges(ei.size(1), world)[rank]
    pairs, own_ids = unique_pairs_sharded(ei[0, lo:hi], ei[1, lo:hi], n, group)     # every rank contributes its own events
    num_ho = pairs.size(0)
    # node id of EVERY event the local pairs refer to (own events and halo): look the pair up in the global list
    pair_keys = (pairs[:, 0] * n + pairs[:, 1]).contiguous()
    i, j = local[0], local[1]
    u = torch.searchsorted(pair_keys, (ei[0][i] * n + ei[1][i]).contiguous())
    v = torch.searchsorted(pair_keys, (ei[0][j] * n + ei[1][j]).contiguous())
    edges, weights, cuts = coalesce_sharded(u, v, w[i], num_ho, group)
    return {"edge_index": edges, "edge_weight": weights, "node_sequence": pairs, "num_nodes": num_ho, "row_cuts": cuts,
            "instance_pairs_total": total, "own_event_ids": own_ids}
