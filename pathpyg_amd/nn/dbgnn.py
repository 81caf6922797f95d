"""De Bruijn Graph Neural Network on hand-written HIP message passing (reference ``pathpyG.nn.dbgnn``).

Same constructor, ``forward(data)`` contract and ``state_dict`` layout as the reference model
(src/pathpyG/nn/dbgnn.py:32-151; its GCN layers are torch_geometric 2.7.0 ``GCNConv``):

    first_order_layers.{i}.lin.weight [out,in]   first_order_layers.{i}.bias [out]
    higher_order_layers.{i}.lin.weight           higher_order_layers.{i}.bias
    bipartite_layer.lin1.{weight,bias}           bipartite_layer.lin2.{weight,bias}
    lin.{weight,bias}

so reference checkpoints load unchanged.  What differs is the execution: each graph's GCN normalisation is
computed once and cached on the ``data`` object (PyG recomputes it on every call), every propagation is an
atomics-free CSR segment reduction, a whole GCN layer (aggregation + the dense product on ``v_mfma_f32_16x16x4_f32`` +
bias + ELU) is ONE hand-written kernel for layer widths 16/32/64/128/256 (``csrc/pp_gcn_fused.hip``, ``csrc/pp_gcn_wide.hip``), and
the backward pass runs the same structure over the transposed CSR.  No library GEMM is called for ANY width: widths without a kernel of
their own are zero-padded to the next kernel width (<= 256) or split into 256-wide blocks (:func:`dense_w`), and the reference's default
one-hot features (``torch.eye``, multi_order_model.py:532-533) never meet a GEMM at all — ``I W^T`` is ``W^T``, the first layer is one CSR
segment reduction over the rows of ``W^T``.  All tensors must live on the GPU; fp32 only.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch.nn import Linear, Module, ModuleList, Parameter

from .. import _hip
from .._dispatch import plain as _dispatch_plain


class _Propagate(torch.autograd.Function):
    """y = act(A x + diag(self_coef) s + bias) with A given by a CsrPlan; s is x itself (GCN) or a second input."""

    @staticmethod
    def forward(ctx, plan, x, s, bias, act: bool, grad_is_pre: bool = False):
        """``grad_is_pre``: the (single) consumer of ``y`` is a :class:`_Dense` with ``fuse_act`` — it hands back the gradient
        w.r.t. the PRE-activation and computes this layer's bias gradient itself, so nothing of that is redone here."""
        y = _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, x, plan.self_coef, s, bias, act, heavy=plan.fwd_heavy)
        ctx.plan, ctx.act, ctx.separate_self = plan, act, s is not None
        ctx.has_bias = bias is not None
        ctx.grad_is_pre = grad_is_pre
        ctx.save_for_backward(y if (act and not grad_is_pre) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        plan = ctx.plan
        (y,) = ctx.saved_tensors
        need_x, need_s, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3] and ctx.has_bias
        if ctx.grad_is_pre:
            dpre, dbias = dy.contiguous(), None
        elif ctx.act or need_b:
            dpre, dbias = _hip.act_backward(dy, y, ctx.act, want_dpre=ctx.act, want_dbias=need_b)
            if not ctx.act:
                dpre = dy.contiguous()
        else:
            dpre, dbias = dy.contiguous(), None
        dx = ds = None
        if ctx.separate_self:
            if need_x:
                dx = _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, heavy=plan.bwd_heavy)
            if need_s:
                ds = _hip.scale_rows(dpre, plan.self_coef)
        elif need_x:      # the self term acts on x itself: A^T dpre + diag(self_coef) dpre in one pass
            dx = _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, dpre, heavy=plan.bwd_heavy)
        return None, dx, ds, dbias, None, None


class _Dense(torch.autograd.Function):
    """y = x W^T (+ b) on the matrix cores — hand-written fp32 MFMA kernels for every width up to 64 and for 64/128/256; only other
    shapes (e.g. 100 -> 300) go to the library GEMM.

    ``fuse_act``: ``x`` is the stored activation ELU(pre) of the layer below and that layer was told ``grad_is_pre``: the
    input-gradient GEMM multiplies ELU'(pre) into its epilogue and accumulates the lower layer's bias gradient (returned as
    the gradient of ``act_bias``), replacing a separate three-pass ELU-backward kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, fuse_act: bool, act_bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.fuse_act, ctx.has_act_bias = bias is not None, fuse_act, act_bias is not None
        kind = _hip.dense_supported(weight.size(1), weight.size(0))
        if kind == 0:
            raise ValueError(f"_Dense: no kernel for a {weight.size(1)} -> {weight.size(0)} layer (use dense_w, which pads / blocks such widths)")
        ctx.one_pass = kind == 1
        return _hip.dense(x, weight, True, bias)[0]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = dact = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_sum = ctx.fuse_act and ctx.has_act_bias and ctx.needs_input_grad[4]
        if ctx.one_pass and ctx.needs_input_grad[0] and need_w:
            # everything in one pass over dy and x: input gradient (+ fused ELU' and the lower layer's bias gradient), dW, db
            dx, dact, dw, db = _hip.dense_backward(dy, x, weight, ctx.fuse_act, True, want_sum, ctx.has_bias)
            return dx, dw, db, None, dact
        if ctx.needs_input_grad[0]:
            if ctx.fuse_act:
                dx, dact = _hip.dense(dy, weight, False, None, grad_act=x, want_colsum=want_sum)
            else:
                dx = _hip.dense(dy, weight, False)[0]
        if need_w:
            dw, db = _hip.weight_grad(dy, x, want_bias=ctx.has_bias)
        return dx, dw, db, None, dact


class _GcnLayer(torch.autograd.Function):
    """A whole GCN layer ``y = ELU((A x + diag(self_coef) x) W^T + b)`` in ONE kernel (``pp_gcn_forward_f32``): the aggregation
    runs on the layer INPUT and the 16-row tile it produces is multiplied by ``W`` on the matrix cores before it ever leaves
    the CU, so the transformed matrix ``x W^T`` of the reference's ``A (x W^T)`` order is neither written nor gathered back.

    Contract (as ``_Propagate(grad_is_pre=True)``): the consumer of ``y`` hands back the gradient w.r.t. the PRE-activation and
    computes this layer's bias gradient; ``fuse_act`` / ``act_bias`` describe the layer BELOW exactly as in :class:`_Dense`.
    Backward is the same math as the unfused path — ``G = A^T dpre``, input gradient ``G W`` with the ELU' of the layer below,
    its bias gradient and ``dW = G^T x`` — in one kernel as well (``pp_gcn_backward_f32``): ``G`` never reaches HBM."""

    @staticmethod
    def supported(plan, x, weight) -> bool:
        # any size: from 4 GiB per matrix on the kernels switch to 64-bit row offsets (10^8-row layers, 25.6 GB matrices: 218 ms per
        # step against 243 ms on the two-kernel path; 2*10^7 rows: 41 against 46 ms)
        return (plan.self_coef is not None and plan.n_dst == plan.n_src == x.size(0) and x.dtype == torch.float32
                and _hip.gcn_fused_supported(weight.size(1), weight.size(0)) > 0)

    @staticmethod
    def forward(ctx, plan, x, weight, bias, fuse_act: bool, act_bias, drop_in=None, drop_out=None):
        """``drop_out = (p, seed, tag, row0)``: the dropout that follows this layer's ELU, applied in the kernel's epilogue (``y`` leaves
        the kernel dropped).  ``drop_in`` (with ``fuse_act``): ``x`` is the DROPPED activation of the layer below; the backward kernel puts
        its mask, ``1 / (1 - p)`` and the ELU' at ``x * (1 - p)`` into the gradient it hands down."""
        ctx.plan, ctx.fuse_act, ctx.has_act_bias = plan, fuse_act, act_bias is not None
        ctx.drop_in = drop_in if fuse_act else None
        # first layer of a stack (its input needs no gradient): keep the aggregated input A x; the only gradient left is
        # dW = dpre^T (A x), so the backward pass needs no aggregation at all
        ctx.keep_agg = not ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        # 128-wide layers: the 128 x 128 weight gradient does not fit in registers beside the gather, so A x is always kept for it and the
        # input gradient comes from the forward kernel with a gradient epilogue (pp_gcn_input_grad_f32)
        ctx.wide = _hip.gcn_fused_supported(weight.size(1), weight.size(0)) == 2
        want_agg = ctx.keep_agg or (ctx.wide and ctx.needs_input_grad[2])
        out = _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, x, plan.self_coef, weight, bias, True, want_agg,
                               heavy=plan.fwd_heavy, drop=drop_out)
        if ctx.keep_agg:
            ctx.save_for_backward(out[1], weight)
            return out[0]
        if ctx.wide:
            y, agg = out if want_agg else (out, None)
            ctx.save_for_backward(x, weight, agg)
            return y
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, dpre):
        plan = ctx.plan
        dpre = dpre.contiguous()
        dx = dw = dact = None
        if ctx.keep_agg:
            agg, weight = ctx.saved_tensors
            dw, _ = _hip.weight_grad(dpre, agg, want_bias=False)
            return None, None, dw, None, None, None, None, None
        want_sum = ctx.fuse_act and ctx.has_act_bias and ctx.needs_input_grad[5]
        if ctx.wide:
            x, weight, agg = ctx.saved_tensors
            if ctx.needs_input_grad[2]:
                dw, _ = _hip.weight_grad(dpre, agg, want_bias=False)
            if ctx.needs_input_grad[1]:
                dx, dact = _hip.gcn_input_grad(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, weight,
                                               x if ctx.fuse_act else None, want_sum, heavy=plan.bwd_heavy, drop=ctx.drop_in)
            return None, dx, dw, None, None, dact, None, None
        x, weight = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            # aggregation over the transposed graph, input gradient (+ ELU' and the bias gradient of the layer below) and dW: one kernel
            dx, dact, dw = _hip.gcn_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, x, weight,
                                             ctx.fuse_act, want_sum, heavy=plan.bwd_heavy, drop=ctx.drop_in)
        elif ctx.needs_input_grad[2]:
            g = _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, dpre, plan.self_coef, dpre, heavy=plan.bwd_heavy)
            dw, _ = _hip.weight_grad(g, x, want_bias=False)
        return None, dx, dw, None, None, dact, None, None


class _AggregateAct(torch.autograd.Function):
    """``agg = A y`` for a stored activation ``y = ELU(pre)`` of a layer below that was told ``grad_is_pre``: the backward pass
    returns the gradient w.r.t. that layer's PRE-activation, ``(A^T d_agg) * ELU'(pre)``, and its bias gradient (as the
    gradient of ``act_bias``) from one kernel (``pp_spmm_act_backward_f32``)."""

    @staticmethod
    def forward(ctx, plan, y, act_bias, drop=None):
        """``drop = (p, seed, tag, row0)``: ``y`` was stored DROPPED by its producer's epilogue; the backward kernel applies the mask too."""
        ctx.plan, ctx.has_act_bias, ctx.drop = plan, act_bias is not None, drop
        ctx.save_for_backward(y)
        return _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, y, heavy=plan.fwd_heavy)

    @staticmethod
    def backward(ctx, d_agg):
        plan = ctx.plan
        (y,) = ctx.saved_tensors
        want_sum = ctx.has_act_bias and ctx.needs_input_grad[2]
        dpre, colsum = _hip.spmm_act_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, d_agg.contiguous(), y, want_sum, ctx.drop)
        return None, dpre, colsum, None


class _ActBoundary(torch.autograd.Function):
    """Identity on a stored activation ``y = ELU(pre)`` whose producer follows the ``grad_is_pre`` contract (:class:`_GcnLayer`): the
    backward pass turns the gradient w.r.t. ``y`` into the gradient w.r.t. ``pre`` and hands the column sums to ``act_bias`` (one
    ``pp_act_backward_f32`` pass).  Used where something that is not a fused consumer sits behind the activation — dropout."""

    @staticmethod
    def forward(ctx, y, act_bias):
        ctx.has_bias = act_bias is not None
        ctx.save_for_backward(y)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        want = ctx.has_bias and ctx.needs_input_grad[1]
        dpre, dbias = _hip.act_backward(dy, y, True, want_dpre=True, want_dbias=want)
        return dpre, dbias


class _Aggregate(torch.autograd.Function):
    """``A y`` over a CsrPlan without self term, bias or activation (the bipartite sum of already dropped-out rows)."""

    @staticmethod
    def forward(ctx, plan, y):
        ctx.plan = plan
        return _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, plan.n_dst, y, heavy=plan.fwd_heavy)

    @staticmethod
    def backward(ctx, d):
        plan = ctx.plan
        return None, _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, plan.n_src, d.contiguous(), heavy=plan.bwd_heavy)


class _BipCombine(torch.autograd.Function):
    """``ELU(agg_lin + deg * (per_node + b1))``: the element-wise tail of the bipartite layer (reference nn/dbgnn.py:66-69,143-144 after the
    re-association of lin1) as ONE kernel each way instead of the add / addcmul / elu chain and its three backward kernels."""

    @staticmethod
    def forward(ctx, agg_lin, per_node, deg, bias):
        y = _hip.bip_combine(agg_lin, per_node, deg, bias)
        ctx.save_for_backward(y, deg)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, deg = ctx.saved_tensors
        da, dp, db = _hip.bip_combine_backward(dy, y, deg, ctx.has_bias and ctx.needs_input_grad[3])
        return da, dp, None, db


def bip_combine(agg_lin, per_node, deg, bias):
    return _BipCombine.apply(agg_lin, per_node, deg, bias)


class _DropAct(torch.autograd.Function):
    """Training-mode dropout of a matrix whose rows are the GLOBAL rows ``row0 .. row0 + n`` (or ``rows``), with the counter-based masks of
    ``pp_dropout_f32`` (no mask tensor; the same decision for the same (seed, tag, row, column) on every rank of a partitioned run).
    ``act=True``: the input is a stored activation ``ELU(pre)`` whose producer follows the ``grad_is_pre`` contract — the backward pass
    returns the gradient w.r.t. ``pre`` (dropout backward and ELU backward in ONE pass, ``pp_dropout_act_backward_f32``) and hands the
    column sums to ``act_bias``."""

    @staticmethod
    def forward(ctx, y, act_bias, p: float, seed: int, tag: int, row0: int, rows, act: bool, applied: bool = False):
        # applied: the producing kernel dropped `y` already (fused epilogue): identity forward, the same backward
        out = y.view_as(y) if applied else _hip.dropout(y, p, seed, tag, row0, rows)
        ctx.args = (p, seed, tag, row0, rows, act)
        ctx.has_bias = act_bias is not None
        ctx.save_for_backward(out if act else None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        p, seed, tag, row0, rows, act = ctx.args
        (dropped,) = ctx.saved_tensors
        want = act and ctx.has_bias and ctx.needs_input_grad[1]
        dpre, dbias = _hip.dropout_act_backward(d_out, dropped, p, seed, tag, row0, rows, act, want)
        return dpre, dbias, None, None, None, None, None, None, None


def _draw_seed() -> int:
    """One dropout seed per forward pass (from torch's CPU generator, so ``torch.manual_seed`` makes a run reproducible)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).item())


# dropout call sites of DBGNN.forward -> mask tags (shared with the partitioned model: both paths drop the same elements for a given seed)
TAG_FO, TAG_FO_OUT, TAG_HO, TAG_HO_OUT, TAG_HEAD = 0, 32, 64, 96, 128


_KERNEL_WIDTHS = (16, 32, 64, 128, 256)


def _kernel_shape(p: int, q: int) -> tuple[int, int]:
    """Smallest (P', Q') >= (p, q) the dense kernels take (both <= 256)."""
    pp_, qq = next(c for c in _KERNEL_WIDTHS if c >= p), next(c for c in _KERNEL_WIDTHS if c >= q)
    if not _hip.dense_supported(pp_, qq):              # e.g. 32 x 256: too large for the register-resident form, too narrow for the streamed one
        pp_, qq = max(pp_, 64), max(qq, 64)
    return pp_, qq


def dense_w(x: torch.Tensor, weight: torch.Tensor, bias=None, fuse_act: bool = False, act_bias=None) -> torch.Tensor:
    """``x @ weight.T + bias`` on the hand-written MFMA kernels for EVERY width (no library GEMM):

    * widths with a kernel (``_hip.dense_supported``): :class:`_Dense` directly;
    * other widths up to 256 (e.g. 100 -> 100, 20 -> 128): input, weight and bias are zero-padded to the next kernel widths and the
      result is sliced — the padded output columns are exactly 0 (bias 0, ELU' of a stored 0 is 1 and their gradient is sliced away);
    * wider layers (e.g. 300 -> 16): 256-wide blocks of the input / output widths, partial products summed.

    The padding and blocking are ordinary autograd ops around the kernels.  ``fuse_act`` / ``act_bias``: the contract of :class:`_Dense`."""
    q, p = weight.shape
    if _hip.dense_supported(p, q):
        return _Dense.apply(x, weight, bias, fuse_act, act_bias)
    if p <= 256 and q <= 256:
        pp_, qq = _kernel_shape(p, q)
        y = _Dense.apply(F.pad(x, (0, pp_ - p)), F.pad(weight, (0, pp_ - p, 0, qq - q)), None if bias is None else F.pad(bias, (0, qq - q)), fuse_act,
                         None if act_bias is None else F.pad(act_bias, (0, pp_ - p)))
        return y[:, :q].contiguous() if qq != q else y
    if fuse_act:                   # the blocks below read slices of x: settle the activation contract of its producer first (one pass)
        x = _ActBoundary.apply(x, act_bias)
    cols = []
    for q0 in range(0, q, 256):
        acc = None
        for p0 in range(0, p, 256):
            part = dense_w(x[:, p0: p0 + 256].contiguous(), weight[q0: q0 + 256, p0: p0 + 256].contiguous(),
                           bias[q0: q0 + 256] if (bias is not None and p0 == 0) else None)
            acc = part if acc is None else acc + part
        cols.append(acc)
    return cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)


def dense(x, linear: Linear, fuse_act: bool = False, act_bias=None):
    if x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda:
        return dense_w(x, linear.weight, linear.bias, fuse_act, act_bias)
    return linear(x)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, grad = _hip.cross_entropy(logits, target, want_grad=True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean softmax cross-entropy (``F.cross_entropy`` semantics for class-index targets) with its gradient computed in the
    same HIP pass; falls back to torch for CPU tensors or more than 64 classes."""
    if (logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and logits.size(1) <= 64 and logits.size(0) > 0
            and target.dim() == 1 and target.dtype == torch.int64):
        # the kernel has no ignore_index / out-of-range handling: check the class indices first (one tiny reduction + 16-byte read);
        # anything else (soft labels, ignore_index = -100, empty batches) keeps torch's semantics through torch
        # validated once per (storage, slice, in-place version) — the record lives on the base tensor, so the per-step views of one label
        # vector do not cost a read-back each
        holder = target._base if target._base is not None else target
        key = (target.storage_offset(), target.numel(), tuple(target.stride()), target._version)
        seen = getattr(holder, "_pp_class_range", None)
        if not isinstance(seen, dict):
            seen = {}
            try:
                holder._pp_class_range = seen
            except Exception:
                pass
        if key not in seen:
            if len(seen) > 64:
                seen.clear()
            seen[key] = tuple(_hip.minmax(target))
        lo, hi = seen[key]
        if lo >= 0 and hi < logits.size(1):
            return _CrossEntropy.apply(logits, target)
    return F.cross_entropy(logits, target)


def _plan_cache(data) -> dict:
    cache = getattr(data, "_pp_plan_cache", None)
    if cache is None:
        cache = {}
        try:
            object.__setattr__(data, "_pp_plan_cache", cache)     # beside, not inside, the attribute store
        except Exception:          # exotic containers: no caching, still correct
            pass
    return cache


def _stamp(tensors):
    """Identity + in-place version of the tensors a cached object was derived from.  The tensors themselves are kept (not their
    addresses: the caching allocator hands a freed block to the next tensor of the same size)."""
    return tuple((t, t._version) for t in tensors if t is not None)


def _stamp_matches(stamp, tensors) -> bool:
    live = [t for t in tensors if t is not None]
    return len(stamp) == len(live) and all(a is t and v == t._version for (a, v), t in zip(stamp, live))


def _cached(data, key, tensors, build):
    cache = _plan_cache(data)
    hit = cache.get(key)
    if hit is None or not _stamp_matches(hit[0], tensors):
        hit = (_stamp(tensors), build())
        cache[key] = hit
    return hit[1]


def _valid_hints(data) -> dict:
    """Hints attached by ``MultiOrderModel.to_dbgnn_data`` — honoured only while the tensors they describe are still the very
    objects (and in-place versions) they were made for; a replaced or edited edge index falls back to the on-device checks."""
    hints = getattr(data, "_pp_hints", None)
    if not hints:
        return {}
    names = ("edge_index", "edge_weights", "edge_index_higher_order", "edge_weights_higher_order", "bipartite_edge_index")
    if not _stamp_matches(hints.get("stamp", ()), [getattr(data, n, None) for n in names]):
        return {}
    return hints


def _valid_plans(data):
    """``(first-order plan, higher-order plan, bipartite plan)`` handed over by ``MultiOrderModel.to_dbgnn_data`` when the layers came out of
    the order-2 builder (``pp_debruijn2_*`` builds the GCN normalisation of both graphs and the bipartite grouping with the layers) — honoured
    only while the bundle's five graph tensors are what the plans were made for: still deferred, or resolved to the deferred value and not
    edited in place since; a replaced or edited tensor sends ``forward`` to the plans built from the tensors."""
    rec = getattr(data, "_pp_plans", None)
    peek = getattr(data, "peek", None)
    if not rec or peek is None:
        return None
    for name, made in rec["stamp"].items():
        cur = peek(name)
        if isinstance(made, torch.Tensor):
            if not (cur is made and cur._version == rec["versions"][name]):
                return None
        elif not (cur is made or (made.value is not None and cur is made.value and cur._version == made.version)):
            return None
    if int(data.num_nodes) != rec["fo"].n_dst or int(data.num_ho_nodes) != rec["ho"].n_dst:
        return None
    return rec["fo"], rec["ho"], rec["bi"]


def _is_hinted_eye(data, name: str) -> bool:
    """``data.x`` / ``data.x_h`` is the identity matrix ``MultiOrderModel.to_dbgnn_data`` created (its own stamp: assigning other features
    later leaves the other hints alone)."""
    hints = getattr(data, "_pp_hints", None) or {}
    stamp = hints.get(name + "_eye")
    t = getattr(data, name, None)
    return stamp is not None and t is not None and stamp[0] is t and stamp[1] == t._version


class GCNConv(Module):
    """Graph convolution ``D^-1/2 (A + I) D^-1/2 X W^T + b`` with PyG's defaults (self loops added,
    symmetric normalisation by weighted in-degree, bias) — parameters ``lin.weight`` (glorot) and ``bias`` (zeros)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = Linear(in_channels, out_channels, bias=False)
        self.bias = Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        with torch.no_grad():
            self.lin.weight.uniform_(-bound, bound)
            self.bias.zero_()

    def forward(self, x, edge_index, edge_weight=None, *, plan=None, activation: bool = False):
        if plan is None:
            plan = _hip.gcn_plan(edge_index, edge_weight, x.size(0))
        return _Propagate.apply(plan, dense(x, self.lin), None, self.bias, activation)


class BipartiteGraphOperator(Module):
    """Sum, for every first-order node ``i``, of ``lin2(x)[i] + lin1(x_h)[j]`` over its higher-order
    neighbours ``j`` (reference dbgnn.py:32-69)."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.lin1 = Linear(in_ch, out_ch)
        self.lin2 = Linear(in_ch, out_ch)

    def forward(self, x: tuple, bipartite_index: torch.Tensor, n_ho: int, n_fo: int, *, plan=None, activation: bool = False):
        if plan is None:
            plan = _hip.bipartite_plan(bipartite_index, n_ho, n_fo)
        return _Propagate.apply(plan, dense(x[0], self.lin1), dense(x[1], self.lin2), None, activation)

    def message(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        """Per-pair message of the reference's ``MessagePassing("add")`` operator (dbgnn.py:67-69): destination row + source row.
        ``forward`` never materialises the per-pair tensor (the sum over pairs is a CSR segment reduction); kept for callers and
        subclasses that use the reference's hook."""
        return x_i + x_j


class DBGNN(Module):
    """Time-aware GNN over a first-order graph, a higher-order De Bruijn graph and the bipartite map
    between them (Qarkaxhija, Perri, Scholtes 2022; reference dbgnn.py:72-151)."""

    def __init__(self, num_classes: int, num_features: tuple[int, int], hidden_dims: list[int], p_dropout: float = 0.0):
        super().__init__()
        self.num_features = num_features
        self.num_classes = num_classes
        self.hidden_dims = hidden_dims
        self.p_dropout = p_dropout

        self.higher_order_layers = ModuleList([GCNConv(num_features[1], hidden_dims[0])])
        self.first_order_layers = ModuleList([GCNConv(num_features[0], hidden_dims[0])])
        for d in range(1, len(hidden_dims) - 1):
            self.higher_order_layers.append(GCNConv(hidden_dims[d - 1], hidden_dims[d]))
            self.first_order_layers.append(GCNConv(hidden_dims[d - 1], hidden_dims[d]))
        self.bipartite_layer = BipartiteGraphOperator(hidden_dims[-2], hidden_dims[-1])
        self.lin = Linear(hidden_dims[-1], num_classes)

    def forward(self, data) -> torch.Tensor:
        x, x_h = data.x, data.x_h
        n_fo, n_ho = int(data.num_nodes), int(data.num_ho_nodes)
        # bundles made by MultiOrderModel.to_dbgnn_data carry hints (every Graph's edge index is row-sorted, the bipartite
        # sources are arange) that save three device round trips; foreign bundles are checked on the device instead
        handed = _valid_plans(data)
        if handed is not None:
            # the layers came out of the fused order-2 builder with their plans: nothing to normalise, sort or check here
            plan_fo, plan_ho, plan_bi = handed
        else:
            hints = _valid_hints(data)
            rows_sorted = True if hints.get("rows_sorted") else None
            bip_sorted = True if hints.get("bipartite_sources_sorted") else None
            from_edges = bool(hints.get("bipartite_is_fo_edge_heads"))     # order-2 temporal model, "last" mapping: no bipartite sort
            pending = []
            plan_fo = _cached(data, "fo", (data.edge_index, data.edge_weights),
                              lambda: _hip.gcn_plan(data.edge_index, data.edge_weights, n_fo, rows_sorted, pending, want_dst_order=from_edges))
            plan_ho = _cached(data, "ho", (data.edge_index_higher_order, data.edge_weights_higher_order),
                              lambda: _hip.gcn_plan(data.edge_index_higher_order, data.edge_weights_higher_order, n_ho, rows_sorted, pending))
            if not from_edges:
                plan_bi = _cached(data, "bi", (data.bipartite_edge_index,),
                                  lambda: _hip.bipartite_plan(data.bipartite_edge_index, n_ho, n_fo, None, bip_sorted, pending))
            _hip.check_plan_status(pending)
            if from_edges:
                plan_bi = _cached(data, "bi", (data.edge_index, data.edge_weights),
                                  lambda: _hip.bipartite_plan_from_edge_grouping(plan_fo, _dispatch_plain(data.edge_index)[1], n_ho))

        if self.p_dropout > 0 and self.training:
            # dropout -> GCNConv -> ELU (reference dbgnn.py:131-146).  The layers stay on the fused kernels (aggregation + MFMA product + bias +
            # ELU in one launch, one-kernel backward); each dropout is one pass forward and one pass backward that also carries the ELU
            # backward of the layer underneath (_DropAct: counter-based masks, no mask tensor).
            p, seed = self.p_dropout, _draw_seed()

            def stack_drop(layers, h, plan, tag, out_tag, finish=True):
                # `h` is raw input first, then a layer's activation.  contract: its producer wants the gradient w.r.t. its pre-activation;
                # applied: the producer's epilogue dropped `h` already (site tag + i), so the dropout costs no pass in either direction
                pending_bias, contract, applied = None, False, False
                for i, layer in enumerate(layers):
                    weight = layer.lin.weight
                    site_in = (p, seed, tag + i, 0)
                    site_out = (p, seed, tag + i + 1 if i + 1 < len(layers) else out_tag, 0)
                    fused = _GcnLayer.supported(plan, h, weight)
                    in_kernel = fused and _hip.gcn_drop_supported(weight.size(1), weight.size(0))
                    if applied and in_kernel:
                        h = _GcnLayer.apply(plan, h, weight, layer.bias, True, pending_bias, site_in, site_out)
                        pending_bias, contract, applied = layer.bias, True, True
                        continue
                    h = _DropAct.apply(h, pending_bias, *site_in, None, contract, applied)
                    if fused:
                        h = _GcnLayer.apply(plan, h, weight, layer.bias, False, None, None, site_out if in_kernel else None)
                        pending_bias, contract, applied = layer.bias, True, in_kernel
                    else:
                        h, pending_bias, contract, applied = _Propagate.apply(plan, dense(h, layer.lin), None, layer.bias, True), None, False, False
                if not finish and contract and applied:            # the consumer's backward kernel takes the out_tag mask itself
                    return h, pending_bias
                return _DropAct.apply(h, pending_bias, p, seed, out_tag, 0, None, contract, applied), None

            x, _ = stack_drop(self.first_order_layers, x, plan_fo, TAG_FO, TAG_FO_OUT)
            bl = self.bipartite_layer
            if plan_bi.fwd_val is None and self.higher_order_layers[-1].lin.weight.size(0) % 4 == 0 and self.higher_order_layers[-1].lin.weight.size(0) <= 256:
                # sum_j (W1 x_h[j] + b1) = W1 (sum_j x_h[j]) + deg * b1, as below: the dense layers run on the N first-order rows only
                x_h, pending_ho = stack_drop(self.higher_order_layers, x_h, plan_ho, TAG_HO, TAG_HO_OUT, finish=False)
                if pending_ho is not None:
                    agg = _AggregateAct.apply(plan_bi, x_h, pending_ho, (p, seed, TAG_HO_OUT, 0))
                else:
                    agg = _Aggregate.apply(plan_bi, x_h)
                per_edge = dense(x, bl.lin2) + bl.lin1.bias
                x = F.elu(torch.addcmul(dense_w(agg, bl.lin1.weight), plan_bi.self_coef.unsqueeze(1), per_edge))
            else:
                x_h, _ = stack_drop(self.higher_order_layers, x_h, plan_ho, TAG_HO, TAG_HO_OUT)
                x = self.bipartite_layer((x_h, x), data.bipartite_edge_index, n_ho=n_ho, n_fo=n_fo, plan=plan_bi, activation=True)
            return dense(_DropAct.apply(x, None, p, seed, TAG_HEAD, 0, None, False), self.lin)

        # No dropout between an activation and the dense layer that consumes it: every ELU backward is fused into the
        # epilogue of that dense layer's input-gradient GEMM (see _Dense / _Propagate.grad_is_pre).
        def stack(layers, h, plan, one_hot=False):
            below = None                                            # bias of the layer whose activation `h` is
            for i, layer in enumerate(layers):
                if i == 0 and one_hot and h.size(0) == h.size(1) == layer.lin.weight.size(1):
                    # h is the identity (the reference's default features, multi_order_model.py:532-533): I W^T = W^T, so the layer is
                    # ELU(A_hat W^T + b) — one CSR segment reduction over the rows of W^T, no n x n matrix product; its backward pass
                    # (A_hat^T dpre) is the gradient of W^T
                    h = _Propagate.apply(plan, layer.lin.weight.t().contiguous(), None, layer.bias, True, True)
                elif _GcnLayer.supported(plan, h, layer.lin.weight):
                    h = _GcnLayer.apply(plan, h, layer.lin.weight, layer.bias, i > 0, below)
                else:
                    t = dense(h, layer.lin, fuse_act=i > 0, act_bias=below)
                    h = _Propagate.apply(plan, t, None, layer.bias, True, True)
                below = layer.bias
            return h, below

        x, bias_fo = stack(self.first_order_layers, x, plan_fo, _is_hinted_eye(data, "x"))
        x_h, bias_ho = stack(self.higher_order_layers, x_h, plan_ho, _is_hinted_eye(data, "x_h"))
        bl = self.bipartite_layer
        if plan_bi.fwd_val is None and x_h.size(1) % 4 == 0 and x_h.size(1) <= 256:
            # sum_j (W1 x_h[j] + b1) = W1 (sum_j x_h[j]) + deg * b1: aggregate the U higher-order rows FIRST (one gather pass over
            # x_h), then everything else lives on the N first-order rows (N << U) — instead of a dense layer over all U rows
            # forward and its three-matrix backward.  Same sum, re-associated (linearity of lin1).
            agg = _AggregateAct.apply(plan_bi, x_h, bias_ho)
            # (dense(): its weight gradients contract over all N rows on the MFMA kernel; the library GEMM is 4x slower there)
            x = bip_combine(dense_w(agg, bl.lin1.weight), dense(x, bl.lin2, True, bias_fo), plan_bi.self_coef, bl.lin1.bias)
            return dense(x, self.lin)
        x = _Propagate.apply(plan_bi, dense(x_h, bl.lin1, True, bias_ho), dense(x, bl.lin2, True, bias_fo), None, True, True)
        return dense(x, self.lin, True, None)
