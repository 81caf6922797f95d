"""TEST-ONLY stand-in for :class:`pathpyg_amd.nn.sharded.HipOps`: the same operations on torch-CPU (and the CPU oracle), so that
the sharding logic, the halo bookkeeping and the collectives of the partitioned lift + DBGNN can run under ``gloo`` in a container
without a GPU.  Nothing under ``pathpyg_amd/`` imports this file; the product's only ops object is HipOps (HIP kernels)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import aggregate as oa
from oracle import lift as ol


class _Plan:
    fwd_heavy = bwd_heavy = None
    dst_order = None


def _csr(group_key: torch.Tensor, n_groups: int):
    order = torch.sort(group_key, stable=True).indices
    ptr = torch.zeros(n_groups + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(group_key, minlength=n_groups), 0)
    return ptr, order


def _elu_grad(y):
    return torch.where(y > 0, torch.ones_like(y), y + 1)


class _ActGrad(torch.autograd.Function):
    """Identity whose backward turns the gradient w.r.t. a stored ELU activation into the gradient w.r.t. its pre-activation and hands
    the column sums to ``act_bias`` — the contract of ``dense(..., fuse_act=True, act_bias=b)`` on the device."""

    @staticmethod
    def forward(ctx, y, act_bias):
        ctx.save_for_backward(y)
        ctx.has_bias = act_bias is not None
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        d = g * _elu_grad(y)
        return d, (d.sum(0) if ctx.has_bias else None)


class _DroppedGrad(torch.autograd.Function):
    """Identity on an activation its producer stored DROPPED (site (p, seed, tag, row0)): the backward pass returns the gradient w.r.t. the
    producer's pre-activation, g * mask / (1 - p) * ELU'(y_dropped * (1 - p)), and the column sums for ``act_bias``."""

    @staticmethod
    def forward(ctx, y, act_bias, p, seed, tag, row0):
        ctx.save_for_backward(y)
        ctx.site, ctx.has_bias = (p, seed, tag, row0), act_bias is not None
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        p, seed, tag, row0 = ctx.site
        d = g * CpuOps._mask(y.size(0), y.size(1), p, seed, tag, row0, None) * _elu_grad(y * (1.0 - p))
        return d, (d.sum(0) if ctx.has_bias else None), None, None, None, None


class CpuOps:
    name = "cpu-test-standin"

    # ---- plans
    @staticmethod
    def gcn_plan_partition(edge_index_local, edge_weight, n_src, n_dst, halo_dinv, row_sorted=False, status_out=None, want_dst_order=False):
        src, dst = edge_index_local[0].long(), edge_index_local[1].long()
        w = torch.ones(src.numel()) if edge_weight is None else edge_weight.float()
        if src.numel() and (int(src.max()) >= n_src or int(dst.max()) >= n_dst or int(src.min()) < 0 or int(dst.min()) < 0):
            raise IndexError("node index out of range")
        loop = src == dst
        loop_w = torch.ones(n_dst)
        loop_w[dst[loop]] = w[loop]                         # an existing self loop keeps its weight (last wins)
        deg = torch.zeros(n_dst).index_add_(0, dst[~loop], w[~loop]) + loop_w
        dinv_own = deg.pow(-0.5)
        dinv_own[torch.isinf(dinv_own)] = 0
        dinv = torch.cat((dinv_own, halo_dinv(dinv_own))) if halo_dinv is not None else dinv_own
        val = torch.where(loop, torch.zeros_like(w), dinv[src] * w * dinv[dst])
        plan = _Plan()
        plan.n_dst, plan.n_src = n_dst, n_src
        plan.fwd_ptr, by_dst = _csr(dst, n_dst)
        plan.bwd_ptr, by_src = _csr(src, n_src)
        plan.fwd_idx, plan.fwd_val = src[by_dst].int(), val[by_dst]
        plan.bwd_idx, plan.bwd_val = dst[by_src].int(), val[by_src]
        plan.self_coef = dinv_own * loop_w * dinv_own
        if want_dst_order:
            plan.dst_order = by_dst.int()
        return plan

    @staticmethod
    def gcn_plan(edge_index, edge_weight, num_nodes, row_sorted=None, status_out=None, want_dst_order=False):
        return CpuOps.gcn_plan_partition(edge_index, edge_weight, num_nodes, num_nodes, None, want_dst_order=want_dst_order)

    @staticmethod
    def bipartite_plan(bipartite_index, n_src, n_dst, pair_value=None, src_sorted=None, status_out=None):
        src, dst = bipartite_index[0].long(), bipartite_index[1].long()
        plan = _Plan()
        plan.n_dst, plan.n_src = n_dst, n_src
        plan.fwd_ptr, by_dst = _csr(dst, n_dst)
        plan.bwd_ptr, by_src = _csr(src, n_src)
        plan.fwd_idx, plan.bwd_idx = src[by_dst].int(), dst[by_src].int()
        plan.fwd_val = None if pair_value is None else pair_value[by_dst].float()
        plan.bwd_val = None if pair_value is None else pair_value[by_src].float()
        plan.self_coef = torch.bincount(dst, minlength=n_dst).float()
        return plan

    @staticmethod
    def bipartite_from_grouping(plan_fo, edge_dst, n_ho):
        plan = _Plan()
        plan.n_dst, plan.n_src = plan_fo.n_dst, n_ho
        plan.fwd_ptr, plan.fwd_idx, plan.fwd_val = plan_fo.fwd_ptr, plan_fo.dst_order, None
        plan.bwd_ptr, plan.bwd_idx, plan.bwd_val = torch.arange(n_ho + 1, dtype=torch.int32), edge_dst.int(), None
        plan.self_coef = (plan_fo.fwd_ptr[1:] - plan_fo.fwd_ptr[:-1]).float()
        return plan

    @staticmethod
    def check_plan_status(entries, what="DBGNN"):
        return None

    @staticmethod
    def group_rows(keys, num_rows):
        ptr, order = _csr(keys.long(), num_rows)
        return ptr, order.int()

    # ---- lift + aggregation (the oracle)
    @staticmethod
    def temporal_lift(edge_index, time, num_nodes, delta, n_own=None, id_offset=0):
        out = ol.temporal_lift_sorted(edge_index, time, delta, num_nodes)
        if n_own is not None:
            out = out[:, out[0] < n_own]
        return out + id_offset

    @staticmethod
    def coalesce_and_lift(coalesce_args, lift_args):
        return CpuOps.coalesce(*coalesce_args), CpuOps.temporal_lift(*lift_args)

    @staticmethod
    def coalesce(edge_index, weight, num_nodes, reduce="sum", remap=None, want_inverse=False, col_block=None):
        ei = edge_index if remap is None else remap[edge_index]
        if isinstance(weight, str):                       # pathpyg_amd._hip.UNIT: unit weights
            weight = torch.ones(ei.size(1))
        if ei.numel() and int(ei.max()) >= num_nodes:
            raise ValueError("node id >= number of nodes")
        merged_index, merged_weight = oa.coalesce(ei, weight, num_nodes, reduce)
        if not want_inverse:
            return merged_index, merged_weight
        keys = merged_index[0] * num_nodes + merged_index[1]
        return merged_index, merged_weight, torch.searchsorted(keys.contiguous(), (ei[0] * num_nodes + ei[1]).contiguous())

    @staticmethod
    def degree(index, num_bins):
        return torch.bincount(index, minlength=num_bins).to(torch.int32)

    @staticmethod
    def ptr_from_sorted(sorted_index, num_rows):
        ptr = torch.zeros(num_rows + 1, dtype=torch.int64)
        ptr[1:] = torch.cumsum(torch.bincount(sorted_index, minlength=num_rows), 0)
        return ptr

    # ---- CSR segment sums
    @staticmethod
    def spmm(ptr, idx, val, n_rows, x, self_coef=None, s=None, bias=None, act=False, heavy=None):
        counts = (ptr[1:] - ptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(n_rows), counts)
        contrib = x[idx.long()] * (1.0 if val is None else val.unsqueeze(1))
        y = torch.zeros(n_rows, x.size(1)).index_add_(0, rows, contrib)
        if self_coef is not None:
            y = y + self_coef.unsqueeze(1) * (x if s is None else s)[:n_rows]
        if bias is not None:
            y = y + bias
        return F.elu(y) if act else y

    @staticmethod
    def spmm_act_backward(ptr, idx, val, n_rows, d, z, want_colsum, drop=None):
        g = CpuOps.spmm(ptr, idx, val, n_rows, d)
        if drop is not None:
            rows = drop[4] if len(drop) > 4 else None          # explicit global row ids (shards numbered in send order)
            g = g * CpuOps._mask(g.size(0), g.size(1), drop[0], drop[1], drop[2], drop[3], rows) * _elu_grad(z * (1.0 - drop[0]))
        else:
            g = g * _elu_grad(z)
        return g, (g.sum(0) if want_colsum else None)

    # ---- one GCN layer on a rectangular plan
    @staticmethod
    def drop_fusable(weight):
        return True

    @staticmethod
    def layer_forward(plan, x_full, weight, bias, first, out, drop=None):
        agg = CpuOps.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, x_full, plan.self_coef, x_full)
        y = F.elu(agg @ weight.t() + bias)
        if drop is not None:
            y = y * CpuOps._mask(y.size(0), y.size(1), drop[0], drop[1], drop[2], drop[3], None)
        out.copy_(y)
        return agg if first else None

    @staticmethod
    def layer_backward(plan, dpre, x_full, weight, saved, need_input_grad, fuse_below, drop=None):
        if not need_input_grad and saved is not None:
            return None, None, dpre.t() @ saved
        g = CpuOps.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre)
        g[: plan.n_dst] += plan.self_coef.unsqueeze(1) * dpre
        dw = g.t() @ x_full
        if not need_input_grad:
            return None, None, dw
        d_lin = g @ weight
        if fuse_below is not None:
            if drop is not None:                       # fuse_below is the dropped activation: mask / (1 - p), ELU' at y = dropped * (1 - p)
                d_lin = d_lin * CpuOps._mask(d_lin.size(0), d_lin.size(1), drop[0], drop[1], drop[2], drop[3], None) * _elu_grad(fuse_below * (1.0 - drop[0]))
            else:
                d_lin = d_lin * _elu_grad(fuse_below)
            return d_lin, d_lin.sum(0), dw
        return d_lin, None, dw

    @staticmethod
    def owned_backward_ok(weight):
        return True

    @staticmethod
    def transposed_sum(plan, dpre):
        return CpuOps.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre)

    @staticmethod
    def owned_backward(gs, t_own, recv, dpre, x_own, weight, saved):
        total = t_own + gs.plan.self_coef.unsqueeze(1) * dpre
        if recv.size(0):
            if gs.send_unique:
                slot = gs.send_slot.long()
                if gs.send_idx is not None:
                    assert torch.equal(torch.nonzero(slot >= 0).flatten().sort().values, gs.send_idx.long().sort().values)
                else:          # shards numbered in send order: the send list is the prefix
                    assert torch.equal(slot[: gs.n_send], torch.arange(gs.n_send)) and bool((slot[gs.n_send:] < 0).all())
                total = total + torch.where((slot >= 0).unsqueeze(1), recv[slot.clamp(min=0)], torch.zeros_like(total))
            else:
                total = total + CpuOps.spmm(gs.back_ptr, gs.back_idx, None, gs.n_own, recv)
        d_in = (total @ weight) * _elu_grad(x_own)
        dw = dpre.t() @ saved if saved is not None else total.t() @ x_own
        return d_in, d_in.sum(0), dw

    @staticmethod
    def act_combine(d_lin_own, extra, y_below):
        d = (d_lin_own if extra is None else d_lin_own + extra) * _elu_grad(y_below)
        return d, d.sum(0)

    # ---- dropout (the torch evaluation of the same counter-based masks)
    @staticmethod
    def _mask(n, f, p, seed, tag, row0, rows):
        from pathpyg_amd.nn.sharded import dropout_mask
        return dropout_mask(torch.arange(row0, row0 + n) if rows is None else rows, f, p, seed, tag)

    @staticmethod
    def dropout(x, p, seed, tag, row0=0, rows=None, out=None):
        y = x * CpuOps._mask(x.size(0), x.size(1), p, seed, tag, row0, rows)
        if out is None:
            return y
        out.copy_(y)
        return out

    @staticmethod
    def dropout_act_backward(dy, y_dropped, p, seed, tag, row0=0, rows=None, act=True, want_dbias=False):
        g = dy * CpuOps._mask(dy.size(0), dy.size(1), p, seed, tag, row0, rows)
        if act:
            g = g * _elu_grad(y_dropped * (1.0 - p))
        return g, (g.sum(0) if want_dbias else None)

    @staticmethod
    def drop_act(y, act_bias, p, seed, tag, row0, act, applied=False):
        if applied:
            return _DroppedGrad.apply(y, act_bias, p, seed, tag, row0)
        if act:
            y = _ActGrad.apply(y, act_bias)
        return y * CpuOps._mask(y.size(0), y.size(1), p, seed, tag, row0, None)

    # ---- head

    @staticmethod
    def dense(x, linear, fuse_act=False, act_bias=None):
        if fuse_act:
            x = _ActGrad.apply(x, act_bias)
        return linear(x)

    @staticmethod
    def bip_combine(agg_lin, per_node, deg, bias):
        return torch.nn.functional.elu(torch.addcmul(agg_lin, deg.unsqueeze(1), per_node + bias))

    @staticmethod
    def dense_nobias(x, weight):
        return x @ weight.t()

    @staticmethod
    def cross_entropy_mean(logits, target):
        return F.cross_entropy(logits, target)


# =====================================================================================================
# Stand-in for the node-range partition on the node-by-node builder (pathpyg_amd._hip.debruijn2_part_count / _fill; csrc/pp_debruijn.hip):
# an independent torch-CPU evaluation of the same CONTRACT — owned rows numbered in send order, halo rows in (owner, head, source) order, the
# order-2 plan over [owned | halo], the first-order shard over the dense local source space — so that pathpyg_amd.distributed
# ._build_partitioned_by_node, its collectives and the ShardedDBGNN schedule on such shards run under gloo / ThreadWorld without a GPU.
# =====================================================================================================
class _PartCount:
    pass


def _plan_from_edges(src, dst, val, n_src, n_dst, self_coef):
    plan = _Plan()
    plan.n_dst, plan.n_src = n_dst, n_src
    plan.fwd_ptr, by_dst = _csr(dst, n_dst)
    plan.bwd_ptr, by_src = _csr(src, n_src)
    plan.fwd_idx, plan.fwd_val = src[by_dst].int(), val[by_dst]
    plan.bwd_idx, plan.bwd_val = dst[by_src].int(), val[by_src]
    plan.self_coef = self_coef
    return plan


class CpuOpsNode(CpuOps):
    name = "cpu-test-standin (node-range partition)"
    MAX_LIST = 64           # the builder's limit: events per node and side

    @staticmethod
    def debruijn2_part_count(edge_index, time, num_nodes, node_lo, node_hi, cuts_dev, rank, delta, weight=None, pad_rows=None):
        src, dst = edge_index[0].long(), edge_index[1].long()
        n, lo, hi = int(num_nodes), int(node_lo), int(node_hi)
        cuts = [int(v) for v in cuts_dev.tolist()]
        world = len(cuts) - 1
        m = int(src.numel())
        w = torch.ones(m) if weight is None else weight.float()
        own_src, own_dst = (src >= lo) & (src < hi), (dst >= lo) & (dst < hi)
        c = _PartCount()
        c.m, c.n, c.lo, c.n_own, c.world = m, n, lo, hi - lo, world
        c.status = 0
        outdeg = torch.bincount(src[own_src] - lo, minlength=hi - lo) if hi > lo else torch.zeros(0, dtype=torch.long)
        indeg = torch.bincount(dst[own_dst] - lo, minlength=hi - lo) if hi > lo else torch.zeros(0, dtype=torch.long)
        if (outdeg.numel() and int(outdeg.max()) > CpuOpsNode.MAX_LIST) or (indeg.numel() and int(indeg.max()) > CpuOpsNode.MAX_LIST):
            c.status = 4
        # ---- owned order-2 rows = distinct (b, c) of the out-events, lexicographic; local order = send order
        ev_out = torch.nonzero(own_src).flatten()
        keys_out = src[ev_out] * n + dst[ev_out]
        rows_key, inv_out = torch.unique(keys_out, return_inverse=True)
        u2 = int(rows_key.numel())
        w1 = torch.zeros(u2).index_add_(0, inv_out, w[ev_out])
        succ_lex = rows_key % n
        sent = (succ_lex < lo) | (succ_lex >= hi)
        order = torch.sort(torch.where(sent, succ_lex, torch.full_like(succ_lex, n)), stable=True).indices          # local row -> lexicographic row
        perm = torch.empty(u2, dtype=torch.long)
        perm[order] = torch.arange(u2)
        n_send = int(sent.sum())
        succ = succ_lex[order]
        c.u2, c.n_send, c.row_of, c.succ = u2, n_send, order.int(), succ.int()
        c.send_counts = [0 if r == rank else int(((succ[:n_send] >= cuts[r]) & (succ[:n_send] < cuts[r + 1])).sum()) for r in range(world)]
        c.send_slot = torch.where(torch.arange(u2) < n_send, torch.arange(u2), torch.full((u2,), -1)).int()
        node_of_out = torch.full((m,), -1, dtype=torch.long)          # event -> local row of its (src, dst) pair (owned sources)
        node_of_out[ev_out] = perm[inv_out]
        # ---- halo: distinct (a, b) with a foreign a among the in-events, in (owner of a, b, a) order behind the owned rows
        ev_in = torch.nonzero(own_dst).flatten()
        a_in, b_in = src[ev_in], dst[ev_in]
        foreign = ~own_src[ev_in]
        cuts_t = torch.tensor(cuts[1:-1], dtype=torch.long)
        owner = torch.searchsorted(cuts_t, a_in, right=True) if world > 1 else torch.zeros_like(a_in)
        hkey = (owner * n + b_in) * n + a_in
        halo_key, inv_halo = torch.unique(hkey[foreign], return_inverse=True)
        n_halo = int(halo_key.numel())
        halo_owner = halo_key // (n * n)
        c.n_halo = n_halo
        c.recv_counts = [int((halo_owner == r).sum()) for r in range(world)]
        node_as_source = node_of_out.clone()                           # event -> local SOURCE row (owned row, or halo row for foreign sources)
        node_as_source[ev_in[foreign]] = u2 + inv_halo
        # ---- lifted pairs with an owned middle node
        pairs = ol.temporal_lift_sorted(edge_index, time, delta, n)
        pairs = pairs[:, own_dst[pairs[0]]]
        c.e2 = int(pairs.size(1))
        u, v = node_as_source[pairs[0]], node_of_out[pairs[1]]
        n_src = u2 + n_halo
        if pairs.size(1):
            merged, mw = oa.coalesce(torch.stack((u, v)), w[pairs[0]], max(n_src, 1), "sum")
        else:
            merged, mw = torch.zeros((2, 0), dtype=torch.long), torch.zeros(0)
        c.a2 = int(merged.size(1))
        loop = merged[0] == merged[1]
        lw = torch.ones(u2)
        lw[merged[1][loop]] = mw[loop]
        c.ho_deg = torch.zeros(max(n_src, 1))
        c.ho_deg[:u2] = torch.zeros(u2).index_add_(0, merged[1][~loop], mw[~loop]) + lw
        # ---- first-order in-edges of the owned nodes, their weighted degrees
        fkey, finv = torch.unique(a_in * n + b_in, return_inverse=True)
        fa, fb = fkey // n, fkey % n
        fw = torch.zeros(fkey.numel()).index_add_(0, finv, w[ev_in])
        floop = fa == fb
        flw = torch.ones(hi - lo)
        flw[fb[floop] - lo] = fw[floop]
        fo_deg = torch.zeros(max(n, 1))
        fo_deg[lo:hi] = torch.zeros(hi - lo).index_add_(0, fb[~floop] - lo, fw[~floop]) + flw
        c.a1 = int(fkey.numel())
        c.bufs = {"fo_deg": fo_deg}
        c._ho = (merged, mw, lw)
        c._fo = (fa, fb, fw, flw)
        return c

    @staticmethod
    def debruijn2_part_fill(c):
        def inv_sqrt(d):
            out = d.pow(-0.5)
            out[torch.isinf(out)] = 0
            return out
        merged, mw, lw = c._ho
        n_src = c.u2 + c.n_halo
        dinv = inv_sqrt(c.ho_deg[:max(n_src, 1)])
        val = torch.where(merged[0] == merged[1], torch.zeros_like(mw), dinv[merged[0]] * mw * dinv[merged[1]])
        ho = _plan_from_edges(merged[0], merged[1], val, n_src, c.u2, dinv[: c.u2] * lw * dinv[: c.u2])
        fa, fb, fw, flw = c._fo
        lo, hi, n_own = c.lo, c.lo + c.n_own, c.n_own
        d1 = inv_sqrt(c.bufs["fo_deg"])
        src_local = torch.where(fa < lo, fa + n_own, torch.where(fa >= hi, fa, fa - lo))
        fval = torch.where(fa == fb, torch.zeros_like(fw), d1[fa] * fw * d1[fb])
        fo = _plan_from_edges(src_local, fb - lo, fval, c.n, n_own, d1[lo:hi] * flw * d1[lo:hi])
        indeg = (fo.fwd_ptr[1:] - fo.fwd_ptr[:-1]).float()
        return ho, fo, indeg
