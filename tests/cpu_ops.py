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
            g = g * CpuOps._mask(g.size(0), g.size(1), drop[0], drop[1], drop[2], drop[3], None) * _elu_grad(z * (1.0 - drop[0]))
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
                assert torch.equal(torch.nonzero(slot >= 0).flatten().sort().values, gs.send_idx.long().sort().values)
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
