"""The 64 x 64 layer kernel on the order-2 layer of the headline stream, plain (k_gcn_forward) and staged (k_gcn_forward_staged): time per launch,
time of the stage plan, and the largest deviation of both from a float64 torch evaluation of the same layer."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402
from bench import synth_stream  # noqa: E402

dev = torch.device("cuda:0")
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
if len(sys.argv) > 2:
    m, n = int(sys.argv[1]), int(sys.argv[2])
ei, t = synth_stream(m, n, span, seed=1, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
b = _hip.debruijn2(tg.data.edge_index, tg.data.time, n, delta, None)
plan = b.ho
rows = plan.fwd_ptr.numel() - 1
gen = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(rows, 64, generator=gen, device=dev)
w = torch.randn(64, 64, generator=gen, device=dev) / 8
bias = torch.randn(64, generator=gen, device=dev)
sp = _hip.gcn_stage_plan(plan.fwd_ptr, plan.fwd_idx, rows)
cnt = sp.grp_cnt.long()
print(f"stage plan: {cnt.numel()} groups, {int((cnt == 255).sum())} on the ordinary path, distinct sources {int(cnt[cnt < 255].sum())} for {plan.fwd_idx.numel()} entries, "
      f"most in one group {int(cnt[cnt < 255].max())}")


def run(kind, out=None):
    if kind == "staged":
        return _hip.gcn_forward_staged(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, rows, x, plan.self_coef, w, bias, True, sp, out=out)
    return _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, rows, x, plan.self_coef, w, bias, True, out=out)


ys = {kind: run(kind) for kind in ("plain", "staged")}
print(f"staged vs plain: max abs difference {(ys['staged'] - ys['plain']).abs().max().item():.3e}")
check = min(rows, 400_000)
for lo in (0, rows - check):
    ptr = plan.fwd_ptr[lo:lo + check + 1].long()
    cnt = ptr[1:] - ptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(check, device=dev), cnt)
    e0, e1 = int(ptr[0]), int(ptr[-1])
    agg = plan.self_coef[lo:lo + check].double()[:, None] * x[lo:lo + check].double()
    agg.index_add_(0, row_of, plan.fwd_val[e0:e1].double()[:, None] * x[plan.fwd_idx[e0:e1].long()].double())
    ref = torch.nn.functional.elu(agg @ w.double().T + bias.double())
    for kind, y in ys.items():
        err = (y[lo:lo + check].double() - ref).abs().max().item()
        print(f"{kind}: rows [{lo}, {lo + check}): max abs deviation from float64 {err:.3e} (rms of the reference {ref.pow(2).mean().sqrt().item():.3f})")
reps = 20
for kind, y in ys.items():
    for _ in range(3):
        run(kind, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(kind, y)
    torch.cuda.synchronize()
    print(f"{kind}: rows {rows}, entries {plan.fwd_idx.numel()}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per launch")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    _hip.gcn_stage_plan(plan.fwd_ptr, plan.fwd_idx, rows)
torch.cuda.synchronize()
print(f"stage plan: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
