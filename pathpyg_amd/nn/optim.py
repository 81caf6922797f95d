"""Adam for the DBGNN train step as ONE HIP launch over all parameter tensors (``pp_adam_f32``, csrc/pp_dbgnn.hip).

Drop-in for ``torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)`` as the reference's training loops use it
(docs/tutorial/netzschleuder.ipynb:2480): same constructor arguments, ``step`` / ``zero_grad`` / ``state_dict`` of
``torch.optim.Optimizer``, the same update (amsgrad and maximize are not offered).  ``torch.optim.Adam`` issues ~10 multi-tensor launches
and a few dozen host-side device queries per step — 1 ms of host time, which is a third of the step of a small graph or of one rank's share
of a partitioned stream; this class issues one launch per 24 tensors and no query.  fp32 parameters on the GPU only; there is no CPU path.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _hip
from .._lib import check, lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            # pass 1: validate every parameter of the group; nothing is modified before all of them passed (an exception must not leave
            # some step counters advanced with no update applied)
            todo = []
            dev = None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("pathpyg_amd.nn.optim.Adam does not support sparse gradients")
                if p.dtype != torch.float32 or g.dtype != torch.float32:
                    raise RuntimeError("pathpyg_amd.nn.optim.Adam: fp32 parameters only")
                dev = _hip.require_device(p, g) if dev is None else dev
                if p.device != dev:
                    raise RuntimeError(f"parameters of one group on different devices: {dev} and {p.device}")
                if not p.is_contiguous():
                    raise RuntimeError("pathpyg_amd.nn.optim.Adam: parameters must be contiguous")
                todo.append((p, g if g.is_contiguous() else g.contiguous()))
            # pass 2: state + launches (parameters that joined later have their own step number, hence their own launch)
            ps, gs, ms, vs, ns, sts = [], [], [], [], [], []
            step_no = None

            def flush():
                self._launch(group, ps, gs, ms, vs, ns, step_no, dev)
                for st_ in sts:                                # the counters advance once their update has been queued
                    st_["step"] = step_no

            for p, g in todo:
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                mine = int(state["step"]) + 1
                if step_no is not None and step_no != mine:
                    flush()
                    ps, gs, ms, vs, ns, sts = [], [], [], [], [], []
                step_no = mine
                sts.append(state)
                ps.append(p)
                gs.append(g)
                ms.append(state["exp_avg"])
                vs.append(state["exp_avg_sq"])
                ns.append(p.numel())
            if ps:
                flush()
        return loss

    @staticmethod
    def _launch(group, ps, gs, ms, vs, ns, step_no, dev):
        k = len(ps)
        ptrs = ctypes.c_void_p * k
        beta1, beta2 = group["betas"]
        with torch.cuda.device(dev):
            check(lib().pp_adam_f32(k, ptrs(*[t.data_ptr() for t in ps]), ptrs(*[t.data_ptr() for t in gs]), ptrs(*[t.data_ptr() for t in ms]),
                                    ptrs(*[t.data_ptr() for t in vs]), (ctypes.c_int64 * k)(*ns), float(group["lr"]), float(beta1), float(beta2),
                                    float(group["eps"]), float(group["weight_decay"]), int(step_no), _hip._stream()), "pp_adam_f32")
