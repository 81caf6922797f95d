"""Destination-partitioned DBGNN: one process per GPU, every graph split by DESTINATION ROWS (the reference
``pathpyG.nn.dbgnn`` is single-process, nn/dbgnn.py:121-151).

Rank r owns a contiguous range of the rows of every feature matrix.  Its share of a graph is a RECTANGULAR plan: the owned
destination rows aggregate from a local source space ``[owned rows | halo rows]`` where the halo rows are exactly the source nodes
owned by peers that one of its edges points from (:class:`GraphShard`).  Per GCN layer ONE embedding exchange fills the halo
rows (all-to-all of the rows each peer asked for — a sparse all-gather) and in the backward pass ONE exchange returns the
halo rows' gradient contributions to their owners.  On a De Bruijn graph whose cuts follow first-order node boundaries every
higher-order row ``(a, b)`` is needed by exactly one peer (the owner of the block ``(b, .)``), so the exchange moves each row once
instead of ``R - 1`` times as an all-gather would; on graphs without that structure it degrades gracefully to an all-gather.
The bipartite projection sums, per rank, the owned higher-order rows into all first-order rows and reduce-scatters the
``[N, H]`` partials (cheaper than exchanging the ``U >> N`` higher-order rows).  The layers themselves are the same fused HIP
kernels as on one GPU (``pp_gcn_forward_f32`` / ``pp_gcn_backward_f32`` / ``pp_gcn_input_grad_f32`` on rectangular plans); weights are
replicated and their gradients summed with one flattened all-reduce.

All device work goes through an ``ops`` object (:class:`HipOps`, the HIP kernels).  Tests inject a CPU stand-in to exercise the
sharding logic and the collectives under ``gloo`` without a GPU; the product has no CPU path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _hip

TAG_FO, TAG_FO_OUT, TAG_HO, TAG_HO_OUT, TAG_HEAD = 0, 32, 64, 96, 128          # dropout call sites (same numbering as nn.dbgnn)
OWNED_ROW_BACKWARD = True          # _ShardedTrunk: exchange A^T dpre and run the layer's GEMMs on the owned rows only (False: the fused kernel over
#                                    owned + halo rows, then exchange of its output — kept for A/B measurements, bench.py --halo-row-backward)


class HipOps:
    """The device operations of the sharded DBGNN, all HIP (see the module docstring for why this is an object)."""

    name = "hip"

    # ---- plans
    gcn_plan = staticmethod(_hip.gcn_plan)
    gcn_plan_partition = staticmethod(_hip.gcn_plan_partition)
    bipartite_plan = staticmethod(_hip.bipartite_plan)
    check_plan_status = staticmethod(_hip.check_plan_status)
    bipartite_from_grouping = staticmethod(_hip.bipartite_plan_from_edge_grouping)

    # ---- lift + aggregation
    debruijn2 = staticmethod(_hip.debruijn2)
    debruijn2_part_count = staticmethod(_hip.debruijn2_part_count)
    debruijn2_part_fill = staticmethod(_hip.debruijn2_part_fill)
    temporal_lift = staticmethod(_hip.temporal_lift)
    coalesce = staticmethod(_hip.coalesce)

    @staticmethod
    def coalesce_and_lift(coalesce_args: tuple, lift_args: tuple):
        """Layer-1 coalesce and the temporal lift of the same stream are independent: their count phases are launched back to back and their
        sizes come back with one read (``(coalesce result, lift result)``)."""
        return tuple(_hip.run_together(_hip.coalesce_steps(*coalesce_args), _hip.temporal_lift_steps(*lift_args)))
    ptr_from_sorted = staticmethod(_hip.ptr_from_sorted)
    degree = staticmethod(_hip.degree)

    @staticmethod
    def group_rows(keys: torch.Tensor, num_rows: int):
        """(ptr int32 [num_rows+1], order int32 [n]): stable grouping of positions by ``keys`` (values in [0, num_rows))."""
        keys32 = keys.to(torch.int32).contiguous()
        bits = max(int(max(num_rows - 1, 1)).bit_length(), 1)
        sorted_keys, order = _hip.sort_pairs(keys32, None, 0, bits)
        ptr = _hip.ptr_from_sorted(sorted_keys, num_rows).to(torch.int32)
        return ptr, order

    # ---- one GCN layer on a (possibly rectangular) plan
    @staticmethod
    def drop_fusable(weight: torch.Tensor) -> bool:
        """Layer shapes whose fused kernels apply dropout in their epilogues (no extra pass)."""
        k = weight.size(1)
        return k % 4 == 0 and _hip.gcn_fused_supported(k, weight.size(0)) > 0 and _hip.gcn_drop_supported(k, weight.size(0))

    @staticmethod
    def layer_forward(plan, x_full: torch.Tensor, weight: torch.Tensor, bias, first: bool, out: torch.Tensor, drop=None):
        """``out[:] = ELU((A x_full + diag(self) x_full[:n_dst]) W^T + b)`` for the plan's ``n_dst`` destination rows; returns what
        :meth:`layer_backward` wants to see again.  ``drop = (p, seed, tag, row0)``: ``out`` leaves dropped (in the kernel's epilogue where
        the shape allows, by one in-place pass otherwise)."""
        kind = _hip.gcn_fused_supported(weight.size(1), weight.size(0)) if x_full.size(1) % 4 == 0 else 0
        in_kernel = drop is not None and HipOps.drop_fusable(weight)
        if kind:
            want_agg = first or kind == 2
            res = _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, x_full, plan.self_coef, weight, bias, True, want_agg,
                                   heavy=plan.fwd_heavy, out=out, drop=drop if in_kernel else None)
            saved = res[1] if want_agg else None
        else:
            with torch.no_grad():                                      # widths without a fused layer kernel: padded / blocked dense kernel + CSR kernel
                from .dbgnn import dense_w
                t = dense_w(x_full, weight)
            out.copy_(_hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, t, plan.self_coef, t, bias, True, heavy=plan.fwd_heavy))
            saved = None
        if drop is not None and not in_kernel:
            _hip.dropout(out, drop[0], drop[1], drop[2], drop[3], None, out)
        return saved

    @staticmethod
    def layer_backward(plan, dpre: torch.Tensor, x_full: torch.Tensor, weight: torch.Tensor, saved, need_input_grad: bool, fuse_below, drop=None):
        """Backward of :meth:`layer_forward` from the gradient w.r.t. its pre-activation: ``(d_lin [n_src, K] or None, colsum or None,
        dW)`` with ``d_lin = (A^T dpre + diag(self) dpre) W``.  ``fuse_below`` (world size 1 only): the layer input IS the stored
        activation of the layer below — its ELU' and bias gradient are folded into the same kernel and ``d_lin`` is final; with
        ``drop`` (needs :meth:`drop_fusable`) that activation was stored DROPPED and the mask goes into the same epilogue."""
        m, k = weight.shape
        kind = _hip.gcn_fused_supported(k, m) if k % 4 == 0 else 0
        fuse = fuse_below is not None
        if not need_input_grad and saved is not None:
            return None, None, _hip.weight_grad(dpre, saved, want_bias=False)[0]
        if kind == 1:
            d_lin, colsum, dw = _hip.gcn_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, x_full, weight,
                                                  fuse, fuse, heavy=plan.bwd_heavy, n_self=plan.n_dst, drop=drop if fuse else None)
            return (d_lin if need_input_grad else None), colsum, dw
        if kind == 2:
            dw = _hip.weight_grad(dpre, saved, want_bias=False)[0]
            d_lin, colsum = _hip.gcn_input_grad(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, weight,
                                                fuse_below, fuse, heavy=plan.bwd_heavy, n_self=plan.n_dst, drop=drop if fuse else None)
            return d_lin, colsum, dw
        g = _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, heavy=plan.bwd_heavy)
        g[: plan.n_dst].addcmul_(dpre, plan.self_coef.unsqueeze(1))
        dw = _hip.weight_grad(g, x_full, want_bias=False)[0]
        if not need_input_grad:
            return None, None, dw
        with torch.no_grad():
            from .dbgnn import dense_w
            d_lin = dense_w(g, weight.t().contiguous())
        if fuse:
            d_lin, colsum = _hip.act_backward(d_lin, fuse_below, True, want_dpre=True, want_dbias=True)
            return d_lin, colsum, dw
        return d_lin, None, dw

    # ---- backward of a layer with the matrix work on the OWNED rows only (world size > 1): the halo rows' partial sums A^T dpre travel back
    # to their owners BEFORE the product with W, so both GEMMs of the layer (input gradient, weight gradient) run once per row
    @staticmethod
    def owned_backward_ok(weight: torch.Tensor) -> bool:
        m, k = weight.shape
        return m % 4 == 0 and k % 4 == 0 and _hip.gcn_fused_supported(k, m) in (1, 2)

    @staticmethod
    def transposed_sum(plan, dpre: torch.Tensor) -> torch.Tensor:
        """``A^T dpre`` over the local source space ``[owned | halo]`` (no self term, no product): ``[n_src, M]``."""
        return _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, heavy=plan.bwd_heavy)

    @staticmethod
    def owned_backward(gs, t_own: torch.Tensor, recv: torch.Tensor, dpre: torch.Tensor, x_own: torch.Tensor, weight: torch.Tensor, saved):
        """``t_own`` = this rank's own partial sums for its owned rows, ``recv`` = the partial sums its consumers returned (send-list order).
        One pass folds them + the self-loop term (pp_halo_fold_f32); then ``(d_lin * ELU'(x_own), its column sums, dW)`` on the owned rows."""
        m, k = weight.shape
        slot = extra = None
        if recv.size(0):
            if gs.send_unique:
                slot = gs.send_slot
            else:
                extra = _hip.spmm(gs.back_ptr, gs.back_idx, None, gs.n_own, recv)
        total = _hip.halo_fold(t_own, recv if slot is not None else None, slot, extra, gs.plan.self_coef, dpre)
        if _hip.gcn_fused_supported(k, m) == 1:
            d_in, colsum, dw, _ = _hip.dense_backward(total, x_own, weight, True, True, True, False)
            return d_in, colsum, dw
        dw = _hip.weight_grad(dpre, saved, want_bias=False)[0] if saved is not None else _hip.weight_grad(total, x_own, want_bias=False)[0]
        d_in, colsum = _hip.dense(total, weight, False, None, x_own, True)
        return d_in, colsum, dw

    # ---- backward of a layer on a shard whose plan carries AUGMENTED source-major rows (node-range partition, _hip.debruijn2_part_fill): the
    # halo sums the peers return are stored behind dpre and enter as one more CSR entry of the sent rows — the fused single-GPU kernels run unchanged
    @staticmethod
    def aug_backward_ok(gs, weight: torch.Tensor) -> bool:
        return getattr(gs.plan, "aug", None) is not None and HipOps.owned_backward_ok(weight)

    @staticmethod
    def halo_transposed_sum(plan, dpre: torch.Tensor) -> torch.Tensor:
        """``A^T dpre`` for the HALO source rows only: ``[n_src - n_dst, M]`` (what travels back to the owners)."""
        return _hip.spmm(plan.bwd_ptr[plan.n_dst:], plan.bwd_idx, plan.bwd_val, plan.n_src - plan.n_dst, dpre, heavy=None)

    @staticmethod
    def aug_backward(gs, dbuf: torch.Tensor, x_own: torch.Tensor, weight: torch.Tensor, saved):
        """``dbuf = [dpre (n_own rows) | returned halo sums (n_send rows)]`` -> ``(d_lin * ELU'(x_own), its column sums, dW)`` on the owned rows."""
        m, k = weight.shape
        aug_ptr, aug_idx, aug_val = gs.plan.aug
        n_own = gs.n_own
        if _hip.gcn_fused_supported(k, m) == 1:
            return _hip.gcn_backward(aug_ptr, aug_idx, aug_val, n_own, dbuf, gs.plan.self_coef, x_own, weight, True, True, n_self=n_own)
        dw = _hip.weight_grad(dbuf[:n_own], saved, want_bias=False)[0]
        d_in, colsum = _hip.gcn_input_grad(aug_ptr, aug_idx, aug_val, n_own, dbuf, gs.plan.self_coef, weight, x_own, True, n_self=n_own)
        return d_in, colsum, dw

    @staticmethod
    def act_combine(d_lin_own: torch.Tensor, extra, y_below: torch.Tensor):
        """``((d_lin_own + extra) * ELU'(y_below), column sums)``: gradient w.r.t. the pre-activation of the layer below and its bias."""
        d = d_lin_own if extra is None else d_lin_own + extra
        return _hip.act_backward(d, y_below, True, want_dpre=True, want_dbias=True)

    # ---- CSR segment sums
    spmm = staticmethod(_hip.spmm)

    @staticmethod
    def spmm_act_backward(ptr, idx, val, n_rows, d, z, want_colsum, drop=None, out=None):
        """``drop = (p, seed, tag, row0[, rows])``: ``rows`` = explicit global row ids (shards whose local row order is not the global one).
        ``out``: write the result there (the head of a ``[dpre | returned halo sums]`` buffer)."""
        rows = drop[4] if (drop is not None and len(drop) > 4) else None
        if d.size(1) % 4 == 0 and d.size(1) <= 256 and rows is None:
            return _hip.spmm_act_backward(ptr, idx, val, n_rows, d, z, want_colsum, drop, out=out)
        g = _hip.spmm(ptr, idx, val, n_rows, d)                                                                          # odd widths / explicit rows: two kernels
        if drop is not None:
            return _hip.dropout_act_backward(g, z, drop[0], drop[1], drop[2], 0 if rows is not None else drop[3], rows, True, want_colsum)
        return _hip.act_backward(g, z, True, want_dpre=True, want_dbias=want_colsum)

    # ---- dense layers of the head (first-order rows only) and the loss
    # ---- dropout (counter-based masks keyed by the global row id: pp_dropout_f32 / pp_dropout_act_backward_f32)
    dropout = staticmethod(_hip.dropout)
    dropout_act_backward = staticmethod(_hip.dropout_act_backward)

    @staticmethod
    def drop_act(y, act_bias, p, seed, tag, row0, act: bool, applied: bool = False):
        """Autograd dropout of the owned rows ``row0 ..``; ``act``: ``y`` is a stored activation whose producer expects the gradient w.r.t. its
        pre-activation (see dbgnn._DropAct)."""
        from .dbgnn import _DropAct
        return _DropAct.apply(y, act_bias, p, seed, tag, row0, None, act, applied)

    @staticmethod
    def dense(x, linear, fuse_act: bool = False, act_bias=None):
        from .dbgnn import dense
        return dense(x, linear, fuse_act, act_bias)

    @staticmethod
    def dense_nobias(x, weight):
        from .dbgnn import dense_w
        return dense_w(x, weight)

    @staticmethod
    def bip_combine(agg_lin, per_node, deg, bias):
        """``ELU(agg_lin + deg[:, None] * (per_node + bias))`` — one kernel each way (pp_bip_combine_f32 / _backward_f32)."""
        from .dbgnn import bip_combine
        return bip_combine(agg_lin, per_node, deg, bias)

    @staticmethod
    def cross_entropy_mean(logits, target):
        from .dbgnn import cross_entropy
        return cross_entropy(logits, target)


class GraphShard:
    """One rank's share of a graph under a destination-row partition (see the module docstring).

    ``plan``: rectangular CsrPlan over local ids (destinations ``[0, n_own)``, sources ``[0, n_own + n_halo)``, owned node i is
    both source i and destination i); ``halo_ids``: global ids of the halo rows (ascending, hence grouped by owner);
    ``send_idx`` / ``send_counts``: owned rows each peer asked for; ``recv_counts``: halo rows coming from each peer;
    ``back_ptr`` / ``back_idx``: CSR over the owned rows into the ``[n_send]`` buffer of returned gradient rows; ``send_unique``: every
    owned row is sent to at most one peer (De Bruijn layers cut at first-order node boundaries): returned rows are added in place;
    ``send_slot`` (``send_unique`` shards): int32 ``[n_own]``, the position of an owned row in the send list or -1 — the inverse of ``send_idx``."""

    __slots__ = ("lo", "hi", "n_own", "n_halo", "n_src", "num_nodes", "cuts", "plan", "halo_ids", "send_idx", "send_counts", "recv_counts",
                 "back_ptr", "back_idx", "send_unique", "send_slot", "halo_fetch", "dense", "send_prefix", "own_ids")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    @property
    def n_send(self) -> int:
        if self.send_prefix is not None:
            return int(self.send_prefix)
        return 0 if self.send_idx is None else int(self.send_idx.numel())

    def send_rows(self, buf: torch.Tensor) -> torch.Tensor:
        """The rows of ``buf[:n_own]`` the peers gather from, grouped by peer.  Shards of the node-by-node builder number their rows in send
        order (``send_prefix`` = rows sent): the send list is a VIEW of the first rows, no pack."""
        if self.send_prefix is not None:
            return buf[: self.send_prefix]
        return buf[: self.n_own].index_select(0, self.send_idx)

    def local_rows(self) -> torch.Tensor:
        """Global ids of the local source space ``[owned | halo]`` (int64).  Shards of the node-by-node builder never exchange ids: the halo
        ids are fetched on first use (``halo_fetch``: one exchange of 8 bytes per halo row — a COLLECTIVE, every rank calls it alike)."""
        if self.halo_ids is None and self.halo_fetch is not None:
            self.halo_ids = self.halo_fetch()
        own = torch.arange(self.lo, self.hi, device=self.plan.fwd_ptr.device) if self.own_ids is None else self.own_ids
        return own if self.n_halo == 0 else torch.cat((own, self.halo_ids))


class _DenseHalo:
    """Pending all-gather of a DENSE-halo shard (every foreign node is a halo row: the first-order graph of a dense stream): ``wait()``
    moves the gathered blocks into the halo rows ``[ids below lo | ids from hi on]`` — no pack of the owned rows per peer, and the collective
    is the ring / direct all-gather instead of an all-to-all of 7 copies."""

    def __init__(self, shard, handle, buf):
        self.shard, self.handle, self.buf = shard, handle, buf

    def wait(self):
        gs, out = self.shard, self.handle.wait()
        world = len(gs.cuts) - 1
        cap = out.size(0) // world
        at = gs.n_own
        for r in range(world):
            size = gs.cuts[r + 1] - gs.cuts[r]
            if size == 0 or gs.cuts[r] == gs.lo:
                continue
            self.buf[at: at + size] = out[r * cap: r * cap + size]
            at += size
        return self.buf[gs.n_own:]


def halo_fill_async(shard: GraphShard, comm, buf: torch.Tensor):
    """Start the embedding exchange of one layer (halo rows ``buf[n_own:]`` <- the owners' rows); ``.wait()`` completes it."""
    if getattr(shard, "dense", False) and comm.world > 1:
        world = len(shard.cuts) - 1
        cap = max(max(shard.cuts[r + 1] - shard.cuts[r] for r in range(world)), 1)
        own = buf[: shard.n_own]
        if shard.n_own != cap:
            own = F.pad(own, (0, 0, 0, cap - shard.n_own))
        return _DenseHalo(shard, comm.all_gather_rows_async(own), buf)
    return comm.exchange_rows_async(shard.send_rows(buf), shard.send_counts, shard.recv_counts, out=buf[shard.n_own:])


def halo_fill(shard: GraphShard, comm, buf: torch.Tensor) -> None:
    """Fill the halo rows ``buf[n_own:]`` with the owners' rows of ``buf[:n_own]`` — the embedding exchange of one layer."""
    if comm.world == 1:          # (a rank without halo rows still takes part: the exchange is a collective)
        return
    if getattr(shard, "dense", False):
        halo_fill_async(shard, comm, buf).wait()
        return
    comm.exchange_rows(shard.send_rows(buf), shard.send_counts, shard.recv_counts, out=buf[shard.n_own:])


def halo_reduce(shard: GraphShard, comm, ops, d_halo: torch.Tensor, d_own: torch.Tensor | None = None):
    """Return the halo rows' gradient contributions to their owners; result: ``[n_own, K]`` sums of what the peers sent (or None).
    ``send_unique`` shards with ``d_own`` given: the returned rows are added to ``d_own`` in place (every row has one consumer) -> None."""
    if comm.world == 1:
        return None
    recv = comm.exchange_rows(d_halo.contiguous(), shard.recv_counts, shard.send_counts)
    if recv.size(0) == 0:
        return None
    if shard.send_unique:
        if shard.send_prefix is not None:          # rows in send order: the returned rows line up with the first rows
            if d_own is not None:
                d_own[: shard.send_prefix] += recv
                return None
            out = torch.zeros((shard.n_own, recv.size(1)), dtype=recv.dtype, device=recv.device)
            out[: shard.send_prefix] = recv
            return out
        if d_own is not None:
            d_own.index_add_(0, shard.send_idx, recv)
            return None
        return torch.zeros((shard.n_own, recv.size(1)), dtype=recv.dtype, device=recv.device).index_add_(0, shard.send_idx, recv)
    return ops.spmm(shard.back_ptr, shard.back_idx, None, shard.n_own, recv)


def dropout_mask(rows: torch.Tensor, width: int, p: float, seed: int, tag: int) -> torch.Tensor:
    """Dropout factors ``[len(rows), width]`` in {0, 1/(1-p)} as a pure function of (seed, tag, GLOBAL row id, column): every rank —
    whatever the world size — derives the same mask for the same row, so the halo copy of a row is dropped exactly like the owner's
    row and a run is reproducible across partitionings.  Integer hash (two multiply-xorshift rounds on 32 bits) in int64 arithmetic."""
    m32 = 0xFFFFFFFF
    idx = rows.to(torch.int64).unsqueeze(1) * width + torch.arange(width, device=rows.device, dtype=torch.int64)
    key = (seed * 0x9E3779B1 + tag * 0x85EBCA6B + 0x27D4EB2F) & m32
    x = ((idx & m32) * 2654435761 + (idx >> 32) * 40503 + key) & m32
    x = (((x >> 16) ^ x) * 0x45D9F3B) & m32
    x = (((x >> 16) ^ x) * 0x45D9F3B) & m32
    x = (x >> 16) ^ x
    return (x >= int(p * 4294967296.0)).to(torch.float32) * (1.0 / (1.0 - p))


class _ShardedGcnStack(torch.autograd.Function):
    """A stack of GCN layers ``h_{l+1} = ELU(A_hat h_l W_l^T + b_l)`` on this rank's destination rows.

    Input ``x_full`` = ``[n_own + n_halo, F]`` features of the local source space (no gradient).  Output = the last layer's
    activation on the owned rows.  Contract as :class:`pathpyg_amd.nn.dbgnn._GcnLayer`: the consumer hands back the gradient
    w.r.t. the last layer's PRE-activation and computes that layer's bias gradient itself."""

    @staticmethod
    def forward(ctx, shard: GraphShard, comm, ops, drop, x_full: torch.Tensor, *params):
        """``drop``: None, or ``(p, seed, tag, out_tag)`` — training-mode dropout on the INPUT of every layer (sites tag + layer) and, with
        ``out_tag``, on the stack's output (the consumer's backward then owns that mask) — the former was: dropout on the INPUT of every layer (reference dbgnn.py:131-140), with the
        reproducible masks of :func:`dropout_mask`: the owner drops its rows before they are exchanged, the first layer's replicated input
        rows are dropped locally (same mask on every rank)."""
        n_layers = len(params) // 2
        plan, n_own = shard.plan, shard.n_own
        inputs, saved = [], []
        h = x_full
        own_rows = getattr(shard, "own_ids", None)          # shards numbered in send order: masks by explicit global row ids, one extra pass
        if drop is not None:
            p_drop, seed, tag, out_tag = drop
            h = ops.dropout(x_full, p_drop, seed, tag, 0, shard.local_rows() if (shard.n_halo or shard.lo or own_rows is not None) else None)
        for layer in range(n_layers):
            weight, bias = params[2 * layer], params[2 * layer + 1]
            last = layer == n_layers - 1
            buf = torch.empty((n_own if last else shard.n_src, weight.size(0)), dtype=torch.float32, device=x_full.device)
            inputs.append(h)
            # the next layer's input dropout is applied by the owner before the exchange — in the layer kernel's epilogue where the shape allows
            site_out = None
            if drop is not None and (not last or out_tag is not None):      # (the last layer's output: the dropout in front of the bipartite layer)
                site_out = (p_drop, seed, out_tag if last else tag + layer + 1, shard.lo)
            saved.append(ops.layer_forward(plan, h, weight, bias, layer == 0, buf[:n_own], site_out if own_rows is None else None))
            if site_out is not None and own_rows is not None:
                ops.dropout(buf[:n_own], site_out[0], site_out[1], site_out[2], 0, own_rows, buf[:n_own])
            if not last:
                halo_fill(shard, comm, buf)
            h = buf
        ctx.shard, ctx.comm, ctx.ops, ctx.n_layers = shard, comm, ops, n_layers
        ctx.inputs, ctx.saved, ctx.drop = inputs, saved, drop
        ctx.save_for_backward(*params)
        return h

    @staticmethod
    def backward(ctx, dpre):
        shard, comm, ops, n_layers = ctx.shard, ctx.comm, ctx.ops, ctx.n_layers
        params = ctx.saved_tensors
        plan, n_own = shard.plan, shard.n_own
        grads = [None] * (2 * n_layers)
        d = dpre.contiguous()
        for layer in range(n_layers - 1, -1, -1):
            weight = params[2 * layer]
            x_in = ctx.inputs[layer]
            if layer == 0:
                grads[0] = ops.layer_backward(plan, d, x_in, weight, ctx.saved[0], False, None)[2]
                break
            # world size 1: the input IS the stored (possibly dropped) activation of the layer below, its backward rides in the same kernel
            fuse_below = x_in if (comm.world == 1 and (ctx.drop is None or ops.drop_fusable(weight))) else None
            site_in = None if ctx.drop is None else (ctx.drop[0], ctx.drop[1], ctx.drop[2] + layer, shard.lo)
            d_lin, colsum, grads[2 * layer] = ops.layer_backward(plan, d, x_in, weight, ctx.saved[layer], True, fuse_below,
                                                                 site_in if fuse_below is not None else None)
            if fuse_below is not None:
                d = d_lin
            else:
                extra = halo_reduce(shard, comm, ops, d_lin[n_own:], d_lin[:n_own])
                if ctx.drop is None:
                    d, colsum = ops.act_combine(d_lin[:n_own], extra, x_in[:n_own])
                else:
                    # x_in holds the DROPPED activation y * keep / (1 - p): the gradient passes the mask, and ELU' is taken at y = x_in * (1 - p)
                    # (where the mask is 0 the gradient is 0 whatever ELU' says) — one pass (pp_dropout_act_backward_f32)
                    d_own = d_lin[:n_own] if extra is None else d_lin[:n_own] + extra
                    p_drop, seed, tag = ctx.drop[:3]
                    own_rows = getattr(shard, "own_ids", None)
                    d, colsum = ops.dropout_act_backward(d_own, x_in[:n_own], p_drop, seed, tag + layer, 0 if own_rows is not None else shard.lo, own_rows,
                                                         True, True)
            grads[2 * layer - 1] = colsum                      # bias gradient of the layer below
        ctx.inputs = ctx.saved = None
        return (None, None, None, None, None, *grads)


class _ShardedBipartite(torch.autograd.Function):
    """``agg[i] = sum of the last higher-order activations y_h[j] over the higher-order nodes j mapped to first-order node i``, for
    the first-order rows this rank owns: local partial sums over ALL first-order rows (rank-major, padded to ``cap`` rows per
    rank) + one reduce-scatter.  Backward: all-gather of the gradient rows, then ``(B^T d) * ELU'(y_h)`` and the last higher-order
    layer's bias gradient in one kernel."""

    @staticmethod
    def forward(ctx, plan, comm, ops, cap: int, n_own_fo: int, y_h: torch.Tensor, act_bias, fuse_act: bool = True, drop=None):
        """``fuse_act=False``: ``y_h`` is not a raw ELU activation (dropout sits in between): plain transposed aggregation backward."""
        ctx.plan, ctx.comm, ctx.ops, ctx.cap, ctx.n_own_fo = plan, comm, ops, cap, n_own_fo
        ctx.fuse_act, ctx.drop = fuse_act, drop          # drop: y_h is stored dropped (site (p, seed, tag, first global row))
        ctx.has_bias = act_bias is not None
        ctx.save_for_backward(y_h)
        partial = ops.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, y_h, heavy=plan.fwd_heavy)       # [world * cap, H]
        return comm.reduce_scatter_rows(partial, cap)[:n_own_fo]

    @staticmethod
    def backward(ctx, d_agg):
        plan, comm, ops = ctx.plan, ctx.comm, ctx.ops
        (y_h,) = ctx.saved_tensors
        d_own = d_agg.contiguous()
        if ctx.cap != ctx.n_own_fo:
            d_own = F.pad(d_own, (0, 0, 0, ctx.cap - ctx.n_own_fo))
        d_full = comm.all_gather_rows(d_own)                                                                     # [world * cap, H]
        if not ctx.fuse_act:
            return None, None, None, None, None, ops.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, d_full, heavy=plan.bwd_heavy), None, None, None
        want = ctx.has_bias and ctx.needs_input_grad[6]
        dpre, colsum = ops.spmm_act_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, d_full, y_h, want, ctx.drop)
        return None, None, None, None, None, dpre, colsum, None, None


class _ShardedTrunk(torch.autograd.Function):
    """Both GCN stacks and the bipartite sum of a partitioned DBGNN as ONE schedule (world size > 1, no dropout), so that every exchange has
    independent kernels to hide behind — the layer functions above run a stack at a time and wait for each exchange where it is issued:

    forward   per layer l:  [wait the higher-order halo of layer l]  HIGHER-ORDER layer l  -> its halo exchange starts
                            [wait the first-order halo of layer l]   FIRST-ORDER layer l   -> its halo exchange starts
              (the first-order layer runs under the higher-order exchange, the next higher-order layer under the first-order one);
              after the last higher-order layer: bipartite partial sums -> reduce-scatter starts, the last first-order layer runs under it;
    backward  the mirror image: all-gather of the bipartite gradient rows under the last first-order layer's backward kernel, every
              stack's halo-gradient exchange under the other stack's backward kernel.

    Inputs: the shard, then the first-order stack's (weight, bias) pairs, then the higher-order stack's.  Outputs ``(y_fo, agg)``: the
    first-order stack's activation and the reduce-scattered bipartite sums on the owned first-order rows.  Contract as the layer functions:
    the consumer of ``y_fo`` hands back the gradient w.r.t. the last first-order PRE-activation and owns that layer's bias gradient; the
    last higher-order layer's bias gradient comes out of the bipartite backward kernel here."""

    @staticmethod
    def forward(ctx, shard, comm, ops, n_layers: int, *params):
        fo, ho = shard.fo, shard.ho
        p_fo, p_ho = params[: 2 * n_layers], params[2 * n_layers:]
        state = {}
        for name, gs, x_full, prm in (("ho", ho, shard.x_h, p_ho), ("fo", fo, shard.x, p_fo)):
            state[name] = {"gs": gs, "h": x_full, "prm": prm, "inputs": [], "saved": [], "pending": None}
        agg_pending = None

        def layer(name, l):
            st = state[name]
            gs, prm = st["gs"], st["prm"]
            last = l == n_layers - 1
            if st["pending"] is not None:
                st["pending"].wait()
                st["pending"] = None
            weight, bias = prm[2 * l], prm[2 * l + 1]
            buf = torch.empty((gs.n_own if last else gs.n_src, weight.size(0)), dtype=torch.float32, device=st["h"].device)
            st["inputs"].append(st["h"])
            st["saved"].append(ops.layer_forward(gs.plan, st["h"], weight, bias, l == 0, buf[: gs.n_own], None))
            if not last:
                st["pending"] = halo_fill_async(gs, comm, buf)
            st["h"] = buf

        mark = getattr(comm, "mark", lambda label: None)
        mark("step: other")
        for l in range(n_layers):
            layer("ho", l)
            mark("fwd: higher-order layers")
            if l == n_layers - 1:
                bip = shard.bip
                partial = ops.spmm(bip.fwd_ptr, bip.fwd_idx, bip.fwd_val, bip.n_dst, state["ho"]["h"], heavy=bip.fwd_heavy)      # [world * cap, H]
                agg_pending = comm.reduce_scatter_rows_async(partial, shard.cap)
                mark("fwd: bipartite sums")
            layer("fo", l)
            mark("fwd: first-order layers")
        agg = agg_pending.wait()[: fo.n_own]
        y_fo, y_ho = state["fo"]["h"], state["ho"]["h"]
        ctx.shard, ctx.comm, ctx.ops, ctx.n_layers = shard, comm, ops, n_layers
        ctx.state = {k: (v["inputs"], v["saved"]) for k, v in state.items()}
        ctx.y_ho = y_ho
        ctx.save_for_backward(*params)
        return y_fo, agg

    @staticmethod
    def backward(ctx, dpre_fo, d_agg):
        shard, comm, ops, n_layers = ctx.shard, ctx.comm, ctx.ops, ctx.n_layers
        params = ctx.saved_tensors
        fo, ho, bip = shard.fo, shard.ho, shard.bip
        grads_fo, grads_ho = [None] * (2 * n_layers), [None] * (2 * n_layers)
        st = {"fo": {"gs": fo, "prm": params[: 2 * n_layers], "grads": grads_fo, "d": dpre_fo.contiguous(), "pending": None},
              "ho": {"gs": ho, "prm": params[2 * n_layers:], "grads": grads_ho, "d": None, "pending": None}}
        # the bipartite gradient rows of all ranks travel while the first-order stack starts its backward pass
        d_own = d_agg.contiguous()
        if shard.cap != fo.n_own:
            d_own = F.pad(d_own, (0, 0, 0, shard.cap - fo.n_own))
        full_pending = comm.all_gather_rows_async(d_own)

        def finish_exchange(name, l):
            """Halo-gradient rows of layer l are back: fold them into the owned rows, ELU' of the layer below, its bias gradient."""
            s = st[name]
            gs = s["gs"]
            if s["pending"][0] == "aug":
                _, dbuf, x_in, sv, weight = s["pending"]
                s["handle"].wait()                                  # (the returned halo sums landed in dbuf[n_own:])
                s["d"], colsum, s["grads"][2 * l] = ops.aug_backward(gs, dbuf, x_in[: gs.n_own], weight, sv)
                s["grads"][2 * l - 1] = colsum
                s["pending"] = None
                return
            if s["pending"][0] == "owned":
                _, t_sum, x_in, dpre_l, sv, weight = s["pending"]
                recv = s["handle"].wait()
                s["d"], colsum, s["grads"][2 * l] = ops.owned_backward(gs, t_sum[: gs.n_own], recv, dpre_l, x_in[: gs.n_own], weight, sv)
                s["grads"][2 * l - 1] = colsum
                s["pending"] = None
                return
            d_lin, x_in = s["pending"]
            recv = s["handle"].wait()
            own = d_lin[: gs.n_own]
            if recv.size(0):
                if gs.send_unique and gs.send_prefix is not None:
                    own[: gs.send_prefix] += recv
                elif gs.send_unique:
                    own.index_add_(0, gs.send_idx, recv)
                else:
                    own = own + ops.spmm(gs.back_ptr, gs.back_idx, None, gs.n_own, recv)
            s["d"], colsum = ops.act_combine(own, None, x_in[: gs.n_own])
            s["grads"][2 * l - 1] = colsum
            s["pending"] = None

        def layer_backward(name, l):
            s = st[name]
            gs = s["gs"]
            inputs, saved = ctx.state[name]
            weight, x_in = s["prm"][2 * l], inputs[l]
            if l == 0:
                s["grads"][0] = ops.layer_backward(gs.plan, s["d"], x_in, weight, saved[0], False, None)[2]
                return
            if OWNED_ROW_BACKWARD and gs.send_prefix is not None and getattr(ops, "aug_backward_ok", lambda g_, w_: False)(gs, weight):
                # the peers' partial sums come back into the tail of [dpre | recv]; the fused backward kernel then runs on the owned rows alone
                d = s["d"]
                dbuf = s.pop("dbuf", None)
                if dbuf is None or dbuf.data_ptr() != d.data_ptr():
                    dbuf = torch.empty((gs.n_own + gs.send_prefix, d.size(1)), dtype=d.dtype, device=d.device)
                    dbuf[: gs.n_own] = d
                t_halo = ops.halo_transposed_sum(gs.plan, dbuf[: gs.n_own])
                s["pending"] = ("aug", dbuf, x_in, saved[l], weight)
                s["handle"] = comm.exchange_rows_async(t_halo, gs.recv_counts, gs.send_counts, out=dbuf[gs.n_own:])
                return
            if OWNED_ROW_BACKWARD and getattr(ops, "owned_backward_ok", lambda w: False)(weight):
                # gather-only pass over owned + halo rows; the halo rows' sums go home before the product with W (see HipOps.owned_backward)
                t_sum = ops.transposed_sum(gs.plan, s["d"])
                s["pending"] = ("owned", t_sum, x_in, s["d"], saved[l], weight)
                s["handle"] = comm.exchange_rows_async(t_sum[gs.n_own:], gs.recv_counts, gs.send_counts)
                return
            d_lin, _, s["grads"][2 * l] = ops.layer_backward(gs.plan, s["d"], x_in, weight, saved[l], True, None)
            s["pending"] = (d_lin, x_in)
            s["handle"] = comm.exchange_rows_async(d_lin[gs.n_own:].contiguous(), gs.recv_counts, gs.send_counts)

        mark = getattr(comm, "mark", lambda label: None)
        mark("step: head + loss (fwd + bwd)")
        for l in range(n_layers - 1, -1, -1):
            if l < n_layers - 1:
                finish_exchange("fo", l + 1)
            layer_backward("fo", l)
            mark("bwd: first-order layers")
            if l == n_layers - 1:
                d_full = full_pending.wait()
                want = ctx.needs_input_grad[4 + 2 * n_layers + 2 * n_layers - 1]
                out_d = None
                if l > 0 and ho.send_prefix is not None and getattr(ops, "aug_backward_ok", lambda g_, w_: False)(ho, st["ho"]["prm"][2 * l]):
                    st["ho"]["dbuf"] = torch.empty((ho.n_own + ho.send_prefix, ctx.y_ho.size(1)), dtype=ctx.y_ho.dtype, device=ctx.y_ho.device)
                    out_d = st["ho"]["dbuf"][: ho.n_own]           # dpre of the last layer is born at the head of its [dpre | recv] buffer
                    st["ho"]["d"], colsum_h = ops.spmm_act_backward(bip.bwd_ptr, bip.bwd_idx, bip.bwd_val, bip.n_src, d_full, ctx.y_ho, want, None, out_d)
                else:
                    st["ho"]["d"], colsum_h = ops.spmm_act_backward(bip.bwd_ptr, bip.bwd_idx, bip.bwd_val, bip.n_src, d_full, ctx.y_ho, want, None)
                grads_ho[2 * n_layers - 1] = colsum_h
            else:
                finish_exchange("ho", l + 1)
            layer_backward("ho", l)
            mark("bwd: higher-order layers (+ bipartite)")
        ctx.state = ctx.y_ho = None
        return (None, None, None, None, *grads_fo, *grads_ho)


class DbgnnShard:
    """Everything one rank needs for DBGNN steps on its partition: the two graph shards, the bipartite plan, the local input features
    (owned + halo rows) and the labels of the owned first-order nodes."""

    __slots__ = ("fo", "ho", "bip", "cap", "indeg", "x", "x_h", "y", "n_fo", "n_ho", "sizes", "pending")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def resolve(self) -> "DbgnnShard":
        """Read the report of the plans whose check was deferred (``pending = (ops, entries)``: bad-index status, hub rows of the
        higher-order graph).  build_dbgnn_shard leaves it to the consumer at world size 1 so that independent kernels (the first-order
        layers) can be queued in front of the read-back; idempotent."""
        if self.pending is not None:
            ops, entries = self.pending
            self.pending = None
            ops.check_plan_status(entries)
        return self


class ShardedDBGNN(torch.nn.Module):
    """Runs a :class:`pathpyg_amd.nn.DBGNN` on a destination-row partition (one process per GPU).

    ``forward(shard)`` returns the logits of the first-order nodes this rank owns; ``loss(shard)`` the rank's share of the
    mean cross-entropy over ALL first-order nodes, so that summing the per-rank weight gradients
    (``all_reduce_gradients(model, average=False)``) reproduces the single-process gradient.  Shards come from
    :func:`pathpyg_amd.distributed.shard_dbgnn_bundle` (a replicated ``to_dbgnn_data`` bundle) or
    :func:`pathpyg_amd.distributed.build_dbgnn_shard` (straight from the event stream: sharded lift + aggregation)."""

    def __init__(self, model, group=None, ops=None, overlap: bool = True):
        super().__init__()
        from ..distributed import Comm
        self.model = model
        self.overlap = overlap          # world size > 1 without dropout: the interleaved schedule of _ShardedTrunk
        self.comm = group if isinstance(group, Comm) else Comm(group)
        self.ops = ops if ops is not None else HipOps()
        self.rank, self.world = self.comm.rank, self.comm.world

    def prepare(self, data, **kw) -> DbgnnShard:
        from ..distributed import shard_dbgnn_bundle
        return shard_dbgnn_bundle(data, self.comm, self.ops, **kw)

    def forward(self, shard: DbgnnShard) -> torch.Tensor:
        m, ops, comm = self.model, self.ops, self.comm
        dropping = m.p_dropout > 0 and m.training
        seed = 0
        if dropping:            # one seed per forward pass, the same on every rank (the masks are functions of the global row id)
            pick = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(shard.x.device)
            seed = int(comm.all_reduce_(pick, torch.distributed.ReduceOp.MAX).item()) if comm.world > 1 else int(pick.item())

        def stack(layers, graph_shard, x_full, tag, out_tag):
            params = []
            for layer in layers:
                params += [layer.lin.weight, layer.bias]
            drop = (m.p_dropout, seed, tag, out_tag) if dropping else None
            return _ShardedGcnStack.apply(graph_shard, comm, ops, drop, x_full, *params), layers[-1].bias

        bl = m.bipartite_layer
        if comm.world > 1:
            shard.resolve()
        if comm.world > 1 and not dropping and len(m.first_order_layers) == len(m.higher_order_layers) and self.overlap:
            # one schedule for both stacks + the bipartite sum: every exchange runs beside kernels of the other stack (_ShardedTrunk)
            params = []
            for layers in (m.first_order_layers, m.higher_order_layers):
                for layer in layers:
                    params += [layer.lin.weight, layer.bias]
            x, agg = _ShardedTrunk.apply(shard, comm, ops, len(m.first_order_layers), *params)
            x = ops.bip_combine(ops.dense_nobias(agg, bl.lin1.weight), ops.dense(x, bl.lin2, True, m.first_order_layers[-1].bias), shard.indeg,
                                bl.lin1.bias)
            return ops.dense(x, m.lin)
        x, bias_fo = stack(m.first_order_layers, shard.fo, shard.x, TAG_FO, TAG_FO_OUT)
        shard.resolve()                       # (higher-order plan report: read while the first-order layers run)
        x_h, bias_ho = stack(m.higher_order_layers, shard.ho, shard.x_h, TAG_HO, TAG_HO_OUT)
        if dropping:            # dropout after both stacks and after the bipartite ELU (reference dbgnn.py:136,142,148), masks as above
            p = m.p_dropout
            # both stacks hand their outputs over DROPPED (last layer's epilogue); what is left of those two sites is their backward: one
            # element-wise pass on the first-order rows, nothing on the higher-order ones (the bipartite backward kernel takes the mask)
            x = ops.drop_act(x, bias_fo, p, seed, TAG_FO_OUT, shard.fo.lo, True, True)
            ho_site = (p, seed, TAG_HO_OUT, shard.ho.lo) if getattr(shard.ho, "own_ids", None) is None else (p, seed, TAG_HO_OUT, 0, shard.ho.own_ids)
            agg = _ShardedBipartite.apply(shard.bip, comm, ops, shard.cap, shard.fo.n_own, x_h, bias_ho, True, ho_site)
            per_edge = ops.dense(x, bl.lin2) + bl.lin1.bias
            x = F.elu(torch.addcmul(ops.dense_nobias(agg, bl.lin1.weight), shard.indeg.unsqueeze(1), per_edge))
            return ops.dense(ops.drop_act(x, None, p, seed, TAG_HEAD, shard.fo.lo, False), m.lin)
        # sum_j (W1 y_h[j] + b1) = W1 (sum_j y_h[j]) + deg * b1 (linearity, as in DBGNN.forward): only [N, H] partials cross xGMI
        agg = _ShardedBipartite.apply(shard.bip, comm, ops, shard.cap, shard.fo.n_own, x_h, bias_ho)
        x = ops.bip_combine(ops.dense_nobias(agg, bl.lin1.weight), ops.dense(x, bl.lin2, True, bias_fo), shard.indeg, bl.lin1.bias)
        return ops.dense(x, m.lin)

    def loss(self, shard: DbgnnShard) -> torch.Tensor:
        out = self.forward(shard)
        n_own = out.size(0)
        if n_own == 0:
            return out.sum() * 0.0
        return self.ops.cross_entropy_mean(out, shard.y) * (n_own / shard.n_fo)
