#!/usr/bin/env python3
"""Per-op timings at the headline workload (HIP events, median of N) — the optimisation loop's A/B harness.

    python tools/bench_kernels.py [--events 10000000] [--iters 10] [--ops spmm,lift,agg,plan,dense]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathpyg_amd as pp  # noqa: E402
from pathpyg_amd import _hip  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--nodes", type=int, default=500_000)
    ap.add_argument("--span", type=int, default=10_000_000)
    ap.add_argument("--delta", type=int, default=1_000_000)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ops", default="lift,agg,plan,spmm,dense")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    ei = torch.randint(0, a.nodes, (2, a.events), generator=g, device=dev)
    t = torch.randint(0, a.span, (a.events,), generator=g, device=dev)
    tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=a.nodes))
    ei, t = tg.data.edge_index, tg.data.time
    ops = set(a.ops.split(","))
    rep = []

    def report(name, fn, gbytes=None):
        med, best = timed(fn, a.iters)
        extra = f"  {gbytes / (med * 1e-3):8.1f} GB/s algorithmic" if gbytes else ""
        rep.append(f"{name:<44} median {med:8.3f} ms   best {best:8.3f} ms{extra}")
        print(rep[-1], flush=True)

    ho = _hip.temporal_lift(ei, t, a.nodes, a.delta)
    e2 = ho.size(1)
    if "lift" in ops:
        report(f"temporal_lift (count+scan+fill) E2={e2}", lambda: _hip.temporal_lift(ei, t, a.nodes, a.delta), (24 * a.events + 16 * e2) / 1e9)
        L = _hip.lib()
        ws = _hip._workspace(L.pp_temporal_ws_bytes(a.events, a.nodes), dev)
        st = torch.cuda.current_stream().cuda_stream
        cnt = lambda: L.pp_temporal_count(ei.data_ptr(), t.data_ptr(), 1, a.events, a.events, a.nodes, 0, a.delta, 0.0, ws.data_ptr(), ws.numel(), st)
        cnt()
        out = torch.empty((2, e2), dtype=torch.int64, device=dev)
        report("  pp_temporal_count (sort+rowptr+count+scan)", cnt)
        report("  pp_temporal_fill (k_expand)", lambda: L.pp_temporal_fill(a.events, a.nodes, e2, 0, out.data_ptr(), ws.data_ptr(), ws.numel(), st), 16 * e2 / 1e9)
        e3 = _hip.linegraph_lift(ho, a.events).size(1)
        report(f"linegraph_lift of the event graph E3={e3}", lambda: _hip.linegraph_lift(ho, a.events), (16 * e2 + 16 * e3) / 1e9)
        ws2 = _hip._workspace(L.pp_linegraph_ws_bytes(e2, a.events), dev)
        L.pp_linegraph_count(ho.data_ptr(), e2, 0, e2, a.events, ws2.data_ptr(), ws2.numel(), st)
        out3 = torch.empty((2, e3), dtype=torch.int64, device=dev)
        report("  pp_linegraph_count", lambda: L.pp_linegraph_count(ho.data_ptr(), e2, 0, e2, a.events, ws2.data_ptr(), ws2.numel(), st))
        report("  pp_linegraph_fill (k_expand, no list)", lambda: L.pp_linegraph_fill(e2, a.events, e3, out3.data_ptr(), ws2.data_ptr(), ws2.numel(), st), 16 * e3 / 1e9)
        big = torch.empty(e3 * 2, dtype=torch.int64, device=dev)
        report("  torch fill_ of the same bytes (write BW reference)", lambda: big.fill_(7), 16 * e3 / 1e9)
        src_copy = torch.empty_like(big)
        report("  torch copy_ of the same bytes (read+write reference)", lambda: src_copy.copy_(big), 32 * e3 / 1e9)
    ns2 = _hip.extend_node_sequence(ei, torch.arange(a.nodes, device=dev).unsqueeze(1))
    if "agg" in ops:
        report("unique_rows [m,2]", lambda: _hip.unique_rows(ns2, (0, a.nodes - 1)))
        w = torch.ones(a.events, device=dev)
        report("coalesce layer 1 (m edges)", lambda: _hip.coalesce(ei, w, a.nodes, "sum"))
        uniq, inv = _hip.unique_rows(ns2, (0, a.nodes - 1))
        w2 = torch.ones(e2, device=dev)
        report("coalesce layer 2 (E2 edges, remap)", lambda: _hip.coalesce(ho, w2, uniq.size(0), "sum", remap=inv))
        report("from_temporal_graph(max_order=2)", lambda: pp.MultiOrderModel.from_temporal_graph(tg, delta=a.delta, max_order=2))
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=a.delta, max_order=2)
    g2 = mom.layers[2]
    n_ho, f = g2.n, a.features
    if "plan" in ops:
        report("gcn_plan higher-order graph", lambda: _hip.gcn_plan(g2.data.edge_index, g2.data.edge_weight, n_ho))
        report("gcn_plan first-order graph", lambda: _hip.gcn_plan(mom.layers[1].data.edge_index, mom.layers[1].data.edge_weight, a.nodes))
    plan = _hip.gcn_plan(g2.data.edge_index, g2.data.edge_weight, n_ho)
    x = torch.randn(n_ho, f, device=dev)
    bias = torch.randn(f, device=dev)
    nnz = plan.fwd_idx.numel()
    alg = (4 * (n_ho + 1) + 8 * nnz + 4 * f * 2 * n_ho + 4 * n_ho) / 1e9
    if "spmm" in ops:
        report(f"spmm fwd ho (rows={n_ho}, nnz={nnz}, F={f})", lambda: _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, x, plan.self_coef, None, bias, True), alg)
        report("spmm bwd ho (transposed)", lambda: _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n_ho, x, plan.self_coef, x), alg)
        p1 = _hip.gcn_plan(mom.layers[1].data.edge_index, mom.layers[1].data.edge_weight, a.nodes)
        x1 = torch.randn(a.nodes, f, device=dev)
        report("spmm fwd fo", lambda: _hip.spmm(p1.fwd_ptr, p1.fwd_idx, p1.fwd_val, a.nodes, x1, p1.self_coef, None, bias, True))
    if "gcn" in ops:
        wq = torch.randn(f, f, device=dev) / 8
        narrow = _hip.gcn_fused_supported(f, f) == 1

        def lin(m):
            return _hip.dense(m, wq, True)[0] if narrow else m @ wq.t()

        fused = _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, x, plan.self_coef, wq, bias, True)
        split = _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, lin(x), plan.self_coef, None, bias, True)
        print("gcn_forward fused vs dense+spmm: max abs diff", float((fused - split).abs().max()), "max abs", float(split.abs().max()))
        del fused, split
        report("gcn_forward fused (gather + MFMA + ELU)", lambda: _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, x, plan.self_coef, wq, bias, True), alg)
        report("gcn_forward fused, keeping A x (first layer / 128-wide layers)", lambda: _hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, x, plan.self_coef, wq, bias, True, True), alg + 4 * f * n_ho / 1e9)
        zp = torch.zeros_like(plan.fwd_ptr)
        report("gcn_forward fused on an EMPTY graph (self rows only: MFMA stage cost)", lambda: _hip.gcn_forward(zp, plan.fwd_idx, plan.fwd_val, n_ho, x, plan.self_coef, wq, bias, True), 8 * f * n_ho / 1e9)
        report("dense + spmm (what it replaces)", lambda: _hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n_ho, lin(x), plan.self_coef, None, bias, True), alg)
        dpre = torch.randn(n_ho, f, device=dev)
        xa = torch.nn.functional.elu(torch.randn(n_ho, f, device=dev))
        if narrow:
            report("gcn_backward fused (gather + d_in + ELU' + colsum + dW)", lambda: _hip.gcn_backward(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n_ho, dpre, plan.self_coef, xa, wq, True, True))
            report("spmm bwd + dense_backward (what it replaces)", lambda: _hip.dense_backward(_hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n_ho, dpre, plan.self_coef, dpre), xa, wq, True, True, True, False))
        else:
            report("gcn_input_grad fused (gather + d_in + ELU' + colsum)", lambda: _hip.gcn_input_grad(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n_ho, dpre, plan.self_coef, wq, xa, True), alg + 4 * f * n_ho / 1e9)
            report("weight_grad(dpre, A x) (the other half of the 128-wide backward)", lambda: _hip.weight_grad(dpre, xa, False), 8 * f * n_ho / 1e9)
            report("spmm bwd + library GEMM + act_backward (what they replace, without dW)", lambda: _hip.act_backward(_hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n_ho, dpre, plan.self_coef, dpre) @ wq, xa, True, True, True))
        del dpre, xa
    if "dense" in ops:
        w = torch.randn(f, f, device=dev)
        y = torch.randn(n_ho, f, device=dev)
        report("F.linear fwd (rocBLAS)", lambda: torch.nn.functional.linear(x, w), 8 * f * n_ho / 1e9)
        report("dy @ W (rocBLAS)", lambda: y @ w, 8 * f * n_ho / 1e9)
        report("weight_grad (MFMA)", lambda: _hip.weight_grad(y, x, True), 8 * f * n_ho / 1e9)
        report("pp_dense fwd (x @ W^T + b)", lambda: _hip.dense(x, w, True, bias), 8 * f * n_ho / 1e9)
        report("pp_dense bwd (dy @ W) * elu'(y) + colsum", lambda: _hip.dense(y, w, False, None, x, True), 12 * f * n_ho / 1e9)
        if _hip.dense_supported(f, f) == 1:
            report("pp_dense_backward (d_in + colsum + dW + db)", lambda: _hip.dense_backward(y, x, w, True, True, True, True), 12 * f * n_ho / 1e9)
        report("act_backward", lambda: _hip.act_backward(y, x, True, True, True), 12 * f * n_ho / 1e9)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.txt", "w") as fh:
        fh.write("\n".join(rep) + "\n")


if __name__ == "__main__":
    main()
