"""GPU parity of the HIP DBGNN (forward, loss, every parameter gradient) against the CPU oracle.
Tolerance: 1e-5 relative fp32 (BASELINE.json north_star) with a 1e-6 absolute floor for values near zero.
The oracle itself is unpinned by the reference (its DBGNN test asserts only ``out is not None``); it is
cross-checked against a dense evaluation in tests/test_oracle_dbgnn.py."""
import numpy as np
import pytest

from tests.tolerance import assert_gradients_close
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-5, 2e-6


@pytest.fixture(scope="module")
def pp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import pathpyg_amd
    return pathpyg_amd


def _bundle(seed, n, e, n_ho, e_ho, f, mapping="last", loops=True):
    g = torch.Generator().manual_seed(seed)

    def graph(nn, ee):
        ei = torch.randint(0, nn, (2, ee), generator=g)
        if loops and ee >= 10:
            ei[:, : ee // 10] = torch.randint(0, nn, (1, ee // 10), generator=g).repeat(2, 1)
        key = torch.unique(ei[0] * nn + ei[1])                      # layers are coalesced: distinct, (row, col)-sorted
        ei = torch.stack((key // nn, key % nn))
        return ei, torch.randint(1, 6, (ei.size(1),), generator=g).float()

    ei, w = graph(n, e)
    ei_h, w_h = graph(n_ho, e_ho)
    ns = torch.randint(0, n, (n_ho, 2), generator=g)
    from oracle import model as om
    data = {
        "num_nodes": n, "num_ho_nodes": n_ho,
        "x": torch.randn(n, f[0], generator=g), "x_h": torch.randn(n_ho, f[1], generator=g),
        "edge_index": ei, "edge_weights": w, "edge_index_higher_order": ei_h, "edge_weights_higher_order": w_h,
        "bipartite_edge_index": om.bipartite_edge_index(ns, mapping),
    }
    return data, torch.randint(0, 3, (n,), generator=g)


def _to_module(pp, params, num_classes, num_features, hidden):
    model = pp.nn.DBGNN(num_classes=num_classes, num_features=num_features, hidden_dims=hidden, p_dropout=0.0)
    missing = model.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(DEV)


@pytest.mark.parametrize("seed,n,e,n_ho,e_ho,f,hidden,mapping", [
    (0, 30, 120, 70, 200, (8, 12), [16, 32, 8], "last"),
    (1, 200, 3000, 900, 4000, (64, 64), [64, 64, 64], "last"),
    (2, 50, 20, 40, 0, (5, 7), [6, 10, 3], "first"),                # scalar-width kernels, empty higher-order graph
    (3, 120, 900, 500, 1500, (16, 16), [32, 16], "both"),            # single GCN layer per order
    (4, 3000, 40000, 20000, 60000, (64, 64), [128, 256, 64], "last"),
    (5, 2500, 30000, 12000, 40000, (128, 128), [128, 128, 128], "last"),   # BASELINE configs[3] widths: 128-wide fused layers
    (6, 800, 9000, 3000, 9000, (64, 128), [128, 64, 32], "last"),          # 64 -> 128 -> 64 stacks mix both kinds of fused layer
    (7, 1500, 20000, 6000, 25000, (256, 256), [256, 256, 256], "last"),    # BASELINE configs[4] widths: every layer on the LDS-streamed kernel
    (8, 700, 8000, 2500, 9000, (64, 256), [256, 128, 256], "both"),        # every combination of 64/128/256 in one model
])
def test_dbgnn_forward_backward_matches_oracle(pp, seed, n, e, n_ho, e_ho, f, hidden, mapping):
    from oracle import dbgnn as od
    data, y = _bundle(seed, n, e, n_ho, e_ho, f, mapping)
    params = od.init_params(3, f, hidden, seed=seed)
    want_out, want_loss, want_grads = od.loss_and_grads(params, data, y)

    model = _to_module(pp, params, 3, f, hidden)
    gdata = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()})
    out = model(gdata)
    loss = F.cross_entropy(out, y.to(DEV))
    loss.backward()
    torch.testing.assert_close(out.detach().cpu(), want_out, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(loss.detach().cpu(), want_loss, rtol=RTOL, atol=ATOL)
    for name, p in model.named_parameters():
        scale = float(want_grads[name].abs().max()) + 1e-12
        torch.testing.assert_close(p.grad.cpu(), want_grads[name], rtol=RTOL * 10, atol=max(ATOL, 2e-5 * scale)), name
    # cached plans: second call reuses them and gives the same numbers
    out2 = model(gdata)
    assert torch.equal(out2, out)


def test_baseline_config0_one_hot_dbgnn_matches_oracle_without_a_library_gemm(pp):
    """BASELINE configs[0]: 100 walks of 5 nodes over a 20-node alphabet, k = 2 De Bruijn model, DBGNN with the reference's
    DEFAULT one-hot features (``to_dbgnn_data()`` without x / x_h: torch.eye, multi_order_model.py:532-533), num_features = (20, U_2),
    hidden_dims [16, 32, 8], 2 classes — forward, loss and every gradient against the oracle (which multiplies by the identity for real);
    and no rocBLAS / hipBLASLt kernel in the step (the first layers read W^T through the CSR)."""
    from oracle import dbgnn as od
    gen = torch.Generator().manual_seed(0)
    walks = torch.randint(0, 20, (100, 5), generator=gen)
    paths = pp.PathData(pp.IndexMap([str(i) for i in range(20)]), device=DEV)
    paths.append_walks([tuple(str(int(v)) for v in row) for row in walks], weights=[1.0] * 100)
    mom = pp.MultiOrderModel.from_path_data(paths, max_order=2)
    data = mom.to_dbgnn_data(max_order=2, mapping="last")
    n, n_ho = mom.layers[1].n, mom.layers[2].n
    assert tuple(data.x.shape) == (n, n) and tuple(data.x_h.shape) == (n_ho, n_ho) and torch.equal(data.x_h, torch.eye(n_ho, device=DEV))
    y = torch.randint(0, 2, (n,), generator=gen)
    hidden = [16, 32, 8]
    params = od.init_params(2, (n, n_ho), hidden, seed=1)
    cpu_data = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in data.to_dict().items()}
    want_out, want_loss, want_grads = od.loss_and_grads(params, cpu_data, y)
    model = pp.nn.DBGNN(num_classes=2, num_features=(n, n_ho), hidden_dims=hidden).to(DEV)
    model.load_state_dict(params)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        out = model(data)
        loss = F.cross_entropy(out, y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
    torch.testing.assert_close(out.detach().cpu(), want_out, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(loss.detach().cpu(), want_loss, rtol=RTOL, atol=ATOL)
    for name, prm in model.named_parameters():
        scale = float(want_grads[name].abs().max()) + 1e-12
        torch.testing.assert_close(prm.grad.cpu(), want_grads[name], rtol=RTOL * 10, atol=max(ATOL, 2e-5 * scale), msg=lambda s_: f"{name}: {s_}")
    names = [e.key for e in prof.key_averages()]
    assert not [k for k in names if "Cijk" in k or "gemm" in k.lower() or "addmm" in k.lower() or "aten::mm" in k], names
    # features assigned afterwards are no longer the identity: the hint must not survive
    data.x_h = torch.eye(n_ho, device=DEV) * 2.0
    out2 = model(data)
    cpu_data["x_h"] = torch.eye(n_ho) * 2.0
    torch.testing.assert_close(out2.detach().cpu(), od.forward(params, cpu_data), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("f,hidden", [((100, 100), [100, 72, 40]), ((300, 20), [16, 32, 8]), ((20, 128), [96, 200, 130]), ((7, 520), [300, 24, 5])])
def test_dbgnn_widths_without_a_kernel_of_their_own_are_padded_or_blocked(pp, f, hidden):
    """Layer widths outside {<= 64, 64, 128, 256}: zero-padded to the next kernel width or split into 256-wide blocks (nn.dbgnn.dense_w) —
    same numbers as the oracle, still no library GEMM."""
    from oracle import dbgnn as od
    data, y = _bundle(21, 150, 1200, 400, 1500, f, "last")
    params = od.init_params(3, f, hidden, seed=3)
    want_out, want_loss, want_grads = od.loss_and_grads(params, data, y)
    model = _to_module(pp, params, 3, f, hidden)
    gdata = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()})
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        out = model(gdata)
        loss = F.cross_entropy(out, y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
    torch.testing.assert_close(out.detach().cpu(), want_out, rtol=RTOL, atol=ATOL)
    for name, prm in model.named_parameters():
        scale = float(want_grads[name].abs().max()) + 1e-12
        torch.testing.assert_close(prm.grad.cpu(), want_grads[name], rtol=RTOL * 10, atol=max(ATOL, 2e-5 * scale), msg=lambda s_: f"{name}: {s_}")
    names = [e.key for e in prof.key_averages()]
    assert not [k for k in names if "Cijk" in k or "gemm" in k.lower() or "addmm" in k.lower() or "aten::mm" in k], names


def test_gcn_conv_layer_alone_with_self_loops_and_isolated_nodes(pp):
    from oracle import dbgnn as od
    ei = torch.tensor([[0, 0, 1, 3, 3], [0, 1, 0, 3, 1]])          # node 2 and 4 isolated; loops on 0 and 3
    w = torch.tensor([3.0, 2.0, 5.0, 0.5, 1.5])
    x = torch.randn(5, 4, generator=torch.Generator().manual_seed(0))
    conv = pp.nn.GCNConv(4, 6).to(DEV)
    want = od.gcn_conv(x, ei, w, conv.lin.weight.detach().cpu(), conv.bias.detach().cpu())
    got = conv(x.to(DEV), ei.to(DEV), w.to(DEV))
    torch.testing.assert_close(got.detach().cpu(), want, rtol=RTOL, atol=ATOL)


def test_reference_dbgnn_smoke(pp):
    # reference tests/nn/test_dbgnn.py:33-43 (one-hot features, dropout 0.4, train mode): output exists, right shape
    paths = pp.PathData(pp.IndexMap(["A", "B", "C", "D", "E"]), device=DEV)
    for wk in (("A", "C", "D"), ("A", "C", "D"), ("B", "C", "E"), ("B", "C", "E")):
        paths.append_walk(wk)
    m = pp.MultiOrderModel.from_path_data(paths, max_order=2)
    data = m.to_dbgnn_data()
    g1, g2 = m.layers[1], m.layers[2]
    data.y = torch.tensor([0, 0, 1, 1, 1], device=DEV)
    model = pp.nn.DBGNN(num_features=[g1.n, g2.n], num_classes=2, hidden_dims=[16, 32, 8], p_dropout=0.4).to(DEV)
    out = model(data)
    assert out is not None and out.shape == (5, 2) and torch.isfinite(out).all()
    # and deterministic parity in eval mode against the oracle
    from oracle import dbgnn as od
    model.eval()
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cpu_data = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in data.to_dict().items()}
    torch.testing.assert_close(model(data).detach().cpu(), od.forward(params, cpu_data), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("n,m,k", [(1, 1, 1), (100, 8, 64), (5000, 64, 64), (70_001, 64, 64), (3000, 40, 100), (2000, 256, 128), (999, 3, 70),
                                   (40_003, 128, 128), (7, 128, 64), (1234, 64, 256), (9001, 256, 256), (513, 128, 256)])
def test_weight_grad_kernel(pp, n, m, k):
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + m + k)
    dy = torch.randn(n, m, generator=g)
    x = torch.randn(n, k, generator=g)
    dw, db = _hip.weight_grad(dy.to(DEV), x.to(DEV), want_bias=True)
    want = (dy.double().t() @ x.double())
    scale = float(want.abs().max()) + 1e-9
    torch.testing.assert_close(dw.cpu().double(), want, rtol=1e-5, atol=1e-5 * scale)
    torch.testing.assert_close(db.cpu().double(), dy.double().sum(0), rtol=1e-5, atol=1e-5 * float(dy.abs().sum(0).max()))
    dw2, none = _hip.weight_grad(dy.to(DEV), x.to(DEV), want_bias=False)
    assert none is None and torch.equal(dw2, dw)          # two-stage reduction is order-fixed: bitwise reproducible


def test_gcn_plan_unsorted_edges_equal_sorted(pp):
    """The plan must not depend on the edge order: row-sorted fast path vs the general (sorting) path."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(9)
    n, e = 500, 6000
    ei = torch.randint(0, n, (2, e), generator=g)
    w = torch.rand(e, generator=g) + 0.5
    x = torch.randn(n, 16, generator=g).to(DEV)
    perm = torch.sort(ei[0], stable=True).indices
    outs = []
    for idx, ww in ((ei, w), (ei[:, perm], w[perm])):
        plan = _hip.gcn_plan(idx.to(DEV), ww.to(DEV), n)
        outs.append((_hip.spmm(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, n, x, plan.self_coef),
                     _hip.spmm(plan.bwd_ptr, plan.bwd_idx, plan.bwd_val, n, x, plan.self_coef)))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)


def test_sharded_dbgnn_world1_matches_oracle_on_gpu(pp):
    """The destination-partitioned code path (rectangular plans with per-pair coefficients, local propagate) on real
    kernels; the multi-rank logic itself is covered by the gloo tests in tests/test_distributed_cpu.py."""
    from oracle import dbgnn as od
    from pathpyg_amd import distributed as pd
    data, y = _bundle(5, 300, 4000, 1500, 6000, (32, 32))
    params = od.init_params(3, (32, 32), [64, 32, 16], seed=5)
    want_out, want_loss, want_grads = od.loss_and_grads(params, data, y)
    net = _to_module(pp, params, 3, (32, 32), [64, 32, 16])
    gdata = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}, y=y.to(DEV))
    sharded = pd.ShardedDBGNN(net)
    shard = sharded.prepare(gdata)
    out = sharded(shard)
    torch.testing.assert_close(out.detach().cpu(), want_out, rtol=RTOL, atol=ATOL)
    loss = sharded.loss(shard)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), want_loss, rtol=RTOL, atol=ATOL)
    for name, p in net.named_parameters():
        scale = float(want_grads[name].abs().max()) + 1e-12
        torch.testing.assert_close(p.grad.cpu(), want_grads[name], rtol=RTOL * 10, atol=max(ATOL, 2e-5 * scale)), name


@pytest.mark.parametrize("n,p,q", [(1, 16, 16), (15, 64, 64), (16, 32, 64), (1000, 64, 32), (70_001, 64, 64), (4097, 16, 64),
                                   (1000, 64, 8), (1000, 8, 64), (333, 12, 10), (5000, 3, 3), (70_001, 64, 13),       # zero-padded narrow widths
                                   (3000, 256, 8), (3000, 8, 256), (500, 128, 32), (300, 200, 10), (300, 10, 200),   # one small side, up to 256
                                   (1, 256, 256), (17, 256, 256), (5000, 256, 256), (3000, 64, 256), (3000, 256, 64), (4097, 128, 256),
                                   (2000, 256, 128), (2000, 128, 128), (70_001, 128, 64)])                              # weights streamed through LDS
def test_dense_mfma_kernel(pp, n, p, q):
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + p + q)
    a = torch.randn(n, p, generator=g)
    w_t = torch.randn(q, p, generator=g)          # forward layout [Q, P]
    w_n = torch.randn(p, q, generator=g)          # gradient layout [P, Q]
    bias = torch.randn(q, generator=g)
    y = F.elu(torch.randn(n, q, generator=g))     # a stored activation
    tol = dict(rtol=2e-5, atol=2e-5 * max(1.0, p / 64) ** 0.5)          # fp32 dot products of length p against a float64 evaluation
    fwd = (a.double() @ w_t.double().t())
    out, none = _hip.dense(a.to(DEV), w_t.to(DEV), True)
    assert none is None
    torch.testing.assert_close(out.cpu(), fwd.float(), **tol)
    out, _ = _hip.dense(a.to(DEV), w_t.to(DEV), True, bias.to(DEV))
    torch.testing.assert_close(out.cpu(), (fwd + bias.double()).float(), **tol)
    out, colsum = _hip.dense(a.to(DEV), w_n.to(DEV), False, None, y.to(DEV), True)
    want = ((a.double() @ w_n.double()) * torch.where(y > 0, torch.ones_like(y), y + 1).double()).float()
    torch.testing.assert_close(out.cpu(), want, **tol)
    torch.testing.assert_close(colsum.cpu(), want.sum(0), rtol=1e-4, atol=1e-4 * float(want.abs().sum(0).max() + 1))
    assert _hip.dense_supported(8, 64) == 2 and _hip.dense_supported(64, 16) == 1 and _hip.dense_supported(256, 256) == 3
    assert _hip.dense_supported(100, 300) == 0 and _hip.dense_supported(64, 96) == 0 and _hip.dense_supported(256, 8) == 2


@pytest.mark.parametrize("n,c", [(1, 2), (1000, 8), (100_003, 8), (5000, 13), (300, 64)])
def test_cross_entropy_kernel(pp, n, c):
    g = torch.Generator().manual_seed(n + c)
    z = (torch.randn(n, c, generator=g) * 3).requires_grad_(True)
    y = torch.randint(0, c, (n,), generator=g)
    want = F.cross_entropy(z, y)
    want.backward()
    zg = z.detach().to(DEV).requires_grad_(True)
    got = pp.nn.cross_entropy(zg, y.to(DEV))
    (got * 2).backward()
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(zg.grad.cpu(), 2 * z.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("n,m,k", [(1, 16, 16), (17, 64, 64), (1000, 32, 64), (70_001, 64, 64), (4097, 64, 16)])
def test_dense_backward_kernel(pp, n, m, k):
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + m + k)
    dy = torch.randn(n, m, generator=g)
    x = F.elu(torch.randn(n, k, generator=g))
    w = torch.randn(m, k, generator=g)
    d_in, colsum, dw, db = _hip.dense_backward(dy.to(DEV), x.to(DEV), w.to(DEV), True, True, True, True)
    want_in = (dy @ w) * torch.where(x > 0, torch.ones_like(x), x + 1)
    torch.testing.assert_close(d_in.cpu(), want_in, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(colsum.cpu(), want_in.sum(0), rtol=1e-4, atol=1e-4 * float(want_in.abs().sum(0).max() + 1))
    ref = dy.double().t() @ x.double()
    torch.testing.assert_close(dw.cpu().double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max() + 1e-9))
    torch.testing.assert_close(db.cpu().double(), dy.double().sum(0), rtol=1e-5, atol=1e-5 * float(dy.abs().sum(0).max() + 1e-9))
    d2, c2, dw2, db2 = _hip.dense_backward(dy.to(DEV), x.to(DEV), w.to(DEV), False, True, False, False)
    torch.testing.assert_close(d2.cpu(), dy @ w, rtol=2e-5, atol=2e-5)
    assert c2 is None and db2 is None and torch.equal(dw2, dw)
    d3, c3, dw3, _ = _hip.dense_backward(dy.to(DEV), x.to(DEV), w.to(DEV), False, False, False, False)
    assert d3 is None and torch.equal(dw3, dw)


@pytest.mark.parametrize("n,e,p,q,with_self,weighted", [
    (1, 0, 16, 16, True, True), (17, 40, 64, 64, True, True), (1000, 5000, 32, 64, True, False), (4097, 9000, 64, 16, True, True),
    (70_001, 200_000, 64, 64, True, True), (300, 20_000, 64, 32, True, True),          # last: rows far longer than one index chunk
    (500, 700, 16, 32, False, True), (2000, 3000, 64, 64, False, False),
    (17, 40, 128, 128, True, True), (5000, 30_000, 128, 128, True, True), (3000, 9000, 64, 128, True, False), (4097, 9000, 128, 64, True, True),
    (1, 0, 128, 128, True, True), (900, 5000, 128, 128, False, True),             # 128-wide shapes: 8 waves per workgroup around one W in LDS
    (1, 0, 256, 256, True, True), (17, 40, 256, 256, True, True), (5000, 30_000, 256, 256, True, True), (3000, 9000, 64, 256, True, False),
    (4097, 9000, 256, 64, True, True), (900, 5000, 256, 128, False, True), (2000, 9000, 128, 256, True, True),
    (300, 20_000, 256, 256, True, True),                                          # a side of 256: weights streamed through LDS in 16-column chunks
    (63, 300, 256, 256, True, True), (64, 300, 256, 256, True, False), (65, 300, 256, 256, False, True), (129, 900, 256, 256, True, True),
    (70_001, 200_000, 256, 256, True, True),           # 256 x 256: weights stationary in registers, 64-row tiles (edges of a tile; several tiles per workgroup)
])
def test_fused_gcn_layer_kernel(pp, n, e, p, q, with_self, weighted):
    """pp_gcn_forward_f32 (aggregate, then multiply on the matrix cores) against a float64 evaluation of the reference order
    A (x W^T) + b; isolated rows, rows longer than the first gather batch / the index chunk, no self term, unit weights."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + e + p)
    dst = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    if n > 10 and e > 100:
        dst[: min(e // 4, 3000)] = dst[min(e // 4, 3000)]           # one long row (fp32 accumulation: keep it to thousands)
        dst = torch.sort(dst).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0).int()
    idx = torch.randint(0, n, (e,), generator=g, dtype=torch.int32)
    val = torch.rand(e, generator=g) if weighted else None
    x = torch.randn(n, p, generator=g)
    w = torch.randn(q, p, generator=g) / p ** 0.5
    bias = torch.randn(q, generator=g)
    self_coef = torch.rand(n, generator=g) if with_self else None
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((dst, idx.long()), (val if weighted else torch.ones(e)).double(), accumulate=True)
    if with_self:
        a += torch.diag(self_coef.double())
    for act in (True, False):
        pre = a @ (x.double() @ w.double().t()) + bias.double()
        want = (F.elu(pre) if act else pre).float()
        got = _hip.gcn_forward(ptr.to(DEV), idx.to(DEV), val.to(DEV) if weighted else None, n, x.to(DEV),
                               self_coef.to(DEV) if with_self else None, w.to(DEV), bias.to(DEV), act).cpu()
        scale = float(want.abs().max()) + 1e-12
        torch.testing.assert_close(got, want, rtol=RTOL, atol=max(ATOL, 1e-6 * scale))


@pytest.mark.parametrize("n,n_src,n_self,e", [(1, 1, 1, 3), (64, 64, 64, 200), (1000, 1700, 1000, 4000), (5000, 9000, 3000, 30_000), (20_000, 20_000, 20_000, 0)])
def test_wide_256_layer_on_rectangular_shards_and_the_kept_aggregate(pp, n, n_src, n_self, e):
    """The 256 x 256 layer kernel (k_wide_ws) on what a partition shard hands it: more source rows than destination rows, the self term
    on the first n_self rows only (input gradient), and the aggregated input A x kept for the weight gradient (forward)."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + e)
    dst = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0).int()
    idx = torch.randint(0, n_src, (max(e, 1),), generator=g, dtype=torch.int32)[:e]
    val = torch.rand(e, generator=g) + 0.1
    x = torch.randn(n_src, 256, generator=g)
    w = torch.randn(256, 256, generator=g) / 16
    bias = torch.randn(256, generator=g)
    sc = torch.rand(n, generator=g)
    agg = torch.zeros(n, 256, dtype=torch.float64)
    agg.index_add_(0, dst, val.double().unsqueeze(1) * x.double()[idx.long()])
    agg += sc.double().unsqueeze(1) * x.double()[:n]
    want = F.elu(agg @ w.double().t() + bias.double())
    got, got_agg = _hip.gcn_forward(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, x.to(DEV), sc.to(DEV), w.to(DEV), bias.to(DEV), True, want_agg=True)
    torch.testing.assert_close(got.cpu(), want.float(), rtol=RTOL, atol=max(ATOL, 1e-6 * float(want.abs().max())))
    torch.testing.assert_close(got_agg.cpu(), agg.float(), rtol=RTOL, atol=max(ATOL, 1e-6 * float(agg.abs().max())))
    d = torch.randn(n_src, 256, generator=g)
    act = F.elu(torch.randn(n, 256, generator=g))
    gsum = torch.zeros(n, 256, dtype=torch.float64)
    gsum.index_add_(0, dst, val.double().unsqueeze(1) * d.double()[idx.long()])
    gsum[:n_self] += sc[:n_self].double().unsqueeze(1) * d.double()[:n_self]
    want_g = (gsum @ w.double()) * torch.where(act > 0, torch.ones_like(act), act + 1).double()
    got_g, got_cs = _hip.gcn_input_grad(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, d.to(DEV), sc.to(DEV), w.to(DEV), act.to(DEV), True, n_self=n_self)
    torch.testing.assert_close(got_g.cpu(), want_g.float(), rtol=RTOL, atol=max(ATOL, 1e-6 * float(want_g.abs().max())))
    torch.testing.assert_close(got_cs.cpu(), want_g.sum(0).float(), rtol=1e-4, atol=1e-5 * float(want_g.abs().sum(0).max() + 1))


def test_wide_256_layer_takes_hub_rows_from_the_chunked_prepass(pp):
    """Rows longer than HEAVY_ROW_ENTRIES are summed by pp_spmm_heavy_f32; the 256 x 256 layer kernel reads the finished sums (heavy.slot /
    heavy.sum) in both directions — with a threshold low enough that hub and ordinary rows share tiles."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(77)
    n, e = 3000, 40_000
    dst = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    dst[:6000] = 17                                             # one hub row of ~6000 entries
    dst[6000:7000] = 1500
    dst = torch.sort(dst).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0).int()
    idx = torch.randint(0, n, (e,), generator=g, dtype=torch.int32)
    val = torch.rand(e, generator=g) * 0.01
    x = torch.randn(n, 256, generator=g)
    w = torch.randn(256, 256, generator=g) / 16
    bias = torch.randn(256, generator=g)
    sc = torch.rand(n, generator=g)
    heavy = _hip.HeavyRows(ptr.to(DEV), n, threshold=300)
    assert heavy.n_heavy >= 2
    agg = torch.zeros(n, 256, dtype=torch.float64)
    agg.index_add_(0, dst, val.double().unsqueeze(1) * x.double()[idx.long()])
    agg += sc.double().unsqueeze(1) * x.double()
    want = F.elu(agg @ w.double().t() + bias.double())
    got = _hip.gcn_forward(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, x.to(DEV), sc.to(DEV), w.to(DEV), bias.to(DEV), True, heavy=heavy)
    torch.testing.assert_close(got.cpu(), want.float(), rtol=RTOL, atol=max(ATOL, 2e-6 * float(want.abs().max())))
    act = F.elu(torch.randn(n, 256, generator=g))
    want_g = (agg @ w.double()) * torch.where(act > 0, torch.ones_like(act), act + 1).double()
    got_g, got_cs = _hip.gcn_input_grad(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, x.to(DEV), sc.to(DEV), w.to(DEV), act.to(DEV), True, heavy=heavy)
    torch.testing.assert_close(got_g.cpu(), want_g.float(), rtol=RTOL, atol=max(ATOL, 2e-6 * float(want_g.abs().max())))
    torch.testing.assert_close(got_cs.cpu(), want_g.sum(0).float(), rtol=1e-4, atol=1e-5 * float(want_g.abs().sum(0).max() + 1))


def test_fused_gcn_layer_rejects_unsupported_shapes(pp):
    from pathpyg_amd import _hip
    ptr = torch.zeros(5, dtype=torch.int32, device=DEV)
    idx = torch.zeros(1, dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):
        _hip.gcn_forward(ptr, idx, None, 4, torch.zeros(4, 48, device=DEV), None, torch.zeros(64, 48, device=DEV), None, True)


@pytest.mark.parametrize("n_rows,n_src,e,f", [(1, 1, 1, 4), (1000, 300, 1000, 64), (5000, 5000, 12000, 32), (70_001, 900, 70_001, 64), (333, 50, 0, 8)])
def test_spmm_act_backward_kernel(pp, n_rows, n_src, e, f):
    """(A d) * elu'(z) + column sums in one pass against float64."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n_rows + e + f)
    row = torch.sort(torch.randint(0, n_rows, (e,), generator=g)).values
    ptr = torch.zeros(n_rows + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n_rows), 0).int()
    idx = torch.randint(0, n_src, (max(e, 1),), generator=g, dtype=torch.int32)[:e]
    val = torch.rand(e, generator=g)
    d = torch.randn(n_src, f, generator=g)
    z = F.elu(torch.randn(n_rows, f, generator=g))
    a = torch.zeros(n_rows, n_src, dtype=torch.float64)
    a.index_put_((row, idx.long()), val.double(), accumulate=True)
    want = (a @ d.double()) * torch.where(z > 0, torch.ones_like(z), z + 1).double()
    for use_val in (True, False):
        if not use_val:
            a1 = torch.zeros(n_rows, n_src, dtype=torch.float64)
            a1.index_put_((row, idx.long()), torch.ones(e, dtype=torch.float64), accumulate=True)
            want = (a1 @ d.double()) * torch.where(z > 0, torch.ones_like(z), z + 1).double()
        got, colsum = _hip.spmm_act_backward(ptr.to(DEV), idx.to(DEV), val.to(DEV) if use_val else None, n_rows, d.to(DEV), z.to(DEV), True)
        torch.testing.assert_close(got.cpu(), want.float(), rtol=RTOL, atol=ATOL)
        cs = want.sum(0).float()
        torch.testing.assert_close(colsum.cpu(), cs, rtol=1e-4, atol=1e-5 * float(want.abs().sum(0).max() + 1))


@pytest.mark.parametrize("n,e,m,k,fuse", [(1, 0, 16, 16, True), (17, 40, 64, 64, True), (1000, 5000, 32, 64, False), (4097, 9000, 64, 16, True),
                                          (70_001, 200_000, 64, 64, True), (300, 20_000, 64, 32, True)])
def test_fused_gcn_backward_kernel(pp, n, e, m, k, fuse):
    """pp_gcn_backward_f32 against float64: G = A^T dpre + self*dpre, d_in = (G W) * elu'(x), column sums, dW = G^T x."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + e + m + k)
    row = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    if n > 10 and e > 100:
        row[: min(e // 4, 2000)] = row[min(e // 4, 2000)]
        row = torch.sort(row).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0).int()
    idx = torch.randint(0, n, (max(e, 1),), generator=g, dtype=torch.int32)[:e]
    val = torch.rand(e, generator=g)
    self_coef = torch.rand(n, generator=g)
    dpre = torch.randn(n, m, generator=g)
    x = F.elu(torch.randn(n, k, generator=g))
    w = torch.randn(m, k, generator=g) / m ** 0.5
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((row, idx.long()), val.double(), accumulate=True)
    a += torch.diag(self_coef.double())
    gmat = a @ dpre.double()
    d_in = gmat @ w.double()
    if fuse:
        d_in = d_in * torch.where(x > 0, torch.ones_like(x), x + 1).double()
    dw = gmat.t() @ x.double()
    got_in, got_sum, got_w = _hip.gcn_backward(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, dpre.to(DEV), self_coef.to(DEV), x.to(DEV),
                                               w.to(DEV), fuse, True)
    scale = float(d_in.abs().max()) + 1e-12
    torch.testing.assert_close(got_in.cpu(), d_in.float(), rtol=RTOL, atol=max(ATOL, 1e-6 * scale))
    torch.testing.assert_close(got_sum.cpu(), d_in.sum(0).float(), rtol=1e-4, atol=1e-5 * float(d_in.abs().sum(0).max() + 1))
    wscale = float((gmat.abs().t() @ x.double().abs()).max()) + 1e-12
    torch.testing.assert_close(got_w.cpu(), dw.float(), rtol=1e-4, atol=2e-6 * wscale)


@pytest.mark.parametrize("n,e,m,k,fuse", [(1, 0, 128, 128, True), (17, 40, 128, 128, True), (5000, 30_000, 128, 128, True), (3000, 9000, 64, 128, False),
                                          (4097, 9000, 128, 64, True), (300, 20_000, 128, 128, True),
                                          (1, 0, 256, 256, True), (17, 40, 256, 256, True), (5000, 30_000, 256, 256, True), (3000, 9000, 64, 256, False),
                                          (4097, 9000, 256, 64, True), (2000, 9000, 256, 128, True), (300, 20_000, 256, 256, False),
                                          (63, 300, 256, 256, True), (64, 300, 256, 256, False), (65, 300, 256, 256, True), (70_001, 200_000, 256, 256, True)])
def test_fused_gcn_input_grad_kernel(pp, n, e, m, k, fuse):
    """pp_gcn_input_grad_f32 (128-wide layers) against float64: d_in = ((A^T dpre + self*dpre) W) * elu'(x) and its column sums."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + e + m + k)
    row = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    if n > 10 and e > 100:
        row[: min(e // 4, 2000)] = row[min(e // 4, 2000)]
        row = torch.sort(row).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0).int()
    idx = torch.randint(0, n, (max(e, 1),), generator=g, dtype=torch.int32)[:e]
    val = torch.rand(e, generator=g)
    self_coef = torch.rand(n, generator=g)
    dpre = torch.randn(n, m, generator=g)
    x = F.elu(torch.randn(n, k, generator=g))
    w = torch.randn(m, k, generator=g) / m ** 0.5
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((row, idx.long()), val.double(), accumulate=True)
    a += torch.diag(self_coef.double())
    d_in = (a @ dpre.double()) @ w.double()
    if fuse:
        d_in = d_in * torch.where(x > 0, torch.ones_like(x), x + 1).double()
    assert _hip.gcn_fused_supported(m, k) == 2 and _hip.gcn_fused_supported(64, 64) == 1 and _hip.gcn_fused_supported(48, 64) == 0
    got_in, got_sum = _hip.gcn_input_grad(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, dpre.to(DEV), self_coef.to(DEV), w.to(DEV),
                                          x.to(DEV) if fuse else None, True)
    scale = float(d_in.abs().max()) + 1e-12
    torch.testing.assert_close(got_in.cpu(), d_in.float(), rtol=RTOL, atol=max(ATOL, 1e-6 * scale))
    torch.testing.assert_close(got_sum.cpu(), d_in.sum(0).float(), rtol=1e-4, atol=1e-5 * float(d_in.abs().sum(0).max() + 1))
    none_in, none_sum = _hip.gcn_input_grad(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, dpre.to(DEV), self_coef.to(DEV), w.to(DEV),
                                            x.to(DEV) if fuse else None, False)
    assert none_sum is None and torch.equal(none_in, got_in)


def _hub_bundle(seed, n, e, n_ho, e_ho, f, hub_share=0.4):
    """Like _bundle, but a large share of all edges points at (and leaves from) a handful of hub nodes: rows far beyond
    HEAVY_ROW_ENTRIES in both CSR directions of both graphs and of the bipartite map."""
    from oracle import model as om
    g = torch.Generator().manual_seed(seed)

    def graph(nn, ee):
        ei = torch.randint(0, nn, (2, ee), generator=g)
        k = int(ee * hub_share)
        ei[1, :k] = torch.randint(0, 3, (k,), generator=g)                   # three hubs collect 40 % of the in-edges
        ei[0, k:2 * k] = torch.randint(3, 5, (k,), generator=g)              # two others emit 40 % of the out-edges
        key = torch.unique(ei[0] * nn + ei[1])
        ei = torch.stack((key // nn, key % nn))
        return ei, torch.randint(1, 4, (ei.size(1),), generator=g).float()

    ei, w = graph(n, e)
    ei_h, w_h = graph(n_ho, e_ho)
    ns = torch.randint(0, n, (n_ho, 2), generator=g)
    ns[: n_ho // 2, 1] = 7                                                    # half of the higher-order nodes map to ONE first-order node
    data = {"num_nodes": n, "num_ho_nodes": n_ho, "x": torch.randn(n, f, generator=g), "x_h": torch.randn(n_ho, f, generator=g),
            "edge_index": ei, "edge_weights": w, "edge_index_higher_order": ei_h, "edge_weights_higher_order": w_h,
            "bipartite_edge_index": om.bipartite_edge_index(ns, "last")}
    return data, torch.randint(0, 3, (n,), generator=g)


@pytest.mark.parametrize("f,hidden", [(32, [32, 32, 16]), (20, [24, 40, 8]), (128, [128, 128, 64])])   # fused / generic SpMM + library GEMM / 128-wide fused
def test_dbgnn_with_hub_rows_matches_float64_oracle(pp, f, hidden):
    """Scale-free shape: rows with 10^4 entries go through the chunked hub pre-pass (pp_spmm_heavy_f32) in every kernel that walks
    CSR rows.  fp32 sums of 10^4 terms in a different order: compared with the float64 oracle at 2e-5 of the largest entry."""
    from oracle import dbgnn as od
    from pathpyg_amd import _hip
    data, y = _hub_bundle(11, 4000, 60_000, 9000, 90_000, f)
    params = od.init_params(3, (f, f), hidden, seed=2)
    ref = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in data.items()}
    want_out, want_loss, want_grads = od.loss_and_grads({k: v.double() for k, v in params.items()}, ref, y)
    net = _to_module(pp, params, 3, (f, f), hidden)
    gdata = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()})
    out = net(gdata)
    plans = gdata._pp_plan_cache
    assert plans["fo"][1].fwd_heavy is not None and plans["fo"][1].bwd_heavy is not None
    assert plans["ho"][1].fwd_heavy is not None and plans["bi"][1].fwd_heavy is not None
    assert plans["fo"][1].fwd_heavy.n_heavy >= 3 and plans["fo"][1].fwd_heavy.n_chunks > plans["fo"][1].fwd_heavy.n_heavy
    loss = F.cross_entropy(out, y.to(DEV))
    loss.backward()
    scale = float(want_out.abs().max())
    torch.testing.assert_close(out.detach().cpu().double(), want_out, rtol=2e-5, atol=2e-5 * scale)
    torch.testing.assert_close(loss.detach().cpu().double(), want_loss, rtol=2e-5, atol=1e-6)
    for name, p in net.named_parameters():
        gs = float(want_grads[name].abs().max()) + 1e-30
        torch.testing.assert_close(p.grad.cpu().double(), want_grads[name], rtol=1e-4, atol=1e-4 * gs), name


@pytest.mark.parametrize("f", [4, 20, 40, 64, 100, 256, 7])
def test_spmm_with_hub_rows(pp, f):
    """pp_spmm_f32 with the chunked hub pre-pass against float64 for widths that do and do not fill the lane groups (7: the
    scalar kernel, where hub rows fall back to the plain row walk)."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(f)
    n, n_src, e = 3000, 2500, 120_000
    row = torch.randint(0, n, (e,), generator=g)
    row[:30_000] = 5                                       # 30 000 entries in one row (15 chunks), 9 000 in another
    row[30_000:39_000] = 77
    row = torch.sort(row).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0).int()
    idx = torch.randint(0, n_src, (e,), generator=g, dtype=torch.int32)
    val = torch.rand(e, generator=g)
    x = torch.randn(n_src, f, generator=g)
    a = torch.zeros(n, n_src, dtype=torch.float64)
    a.index_put_((row, idx.long()), val.double(), accumulate=True)
    want = a @ x.double()
    heavy = _hip.HeavyRows(ptr.to(DEV), n)
    assert heavy.n_heavy == 2 and heavy.n_chunks == 15 + 5
    got = _hip.spmm(ptr.to(DEV), idx.to(DEV), val.to(DEV), n, x.to(DEV), heavy=heavy).cpu().double()
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6 * scale)


def test_fuzz_dbgnn_on_models_built_from_streams_and_paths(pp):
    """End to end through the public API on 24 seeded configurations: temporal streams (the path with derived bipartite plans and
    layer hints) and walk data, mappings last / first / both, one-hot default features and given ones, supported and unsupported
    layer widths, order 2 and 3 as the higher order: logits, loss and every gradient against the oracle on the oracle's own layers."""
    from oracle import dbgnn as od
    from oracle import model as om
    rng = np.random.default_rng(77)
    for case in range(24):
        temporal = case % 3 != 2
        order = 2 if case % 4 else 3
        mapping = ("last", "first", "both")[case % 3] if order == 2 else "last"
        n = int(rng.integers(4, 40))
        if temporal:
            m = int(rng.integers(30, 1500))
            ei = torch.from_numpy(rng.integers(0, n, (2, m)))
            t = torch.from_numpy(rng.integers(0, 200, m))
            delta = int(rng.integers(2, 30))
            sei, st, _ = om.stable_time_sort(ei, t)
            want = om.layers_from_temporal(sei, st, n, delta=delta, max_order=order)
            if want[order]["edge_index"].size(1) == 0 and order == 3:
                continue
            g = pp.TemporalGraph(pp.Data(edge_index=ei.to(DEV), time=t.to(DEV), num_nodes=n))
            model = pp.MultiOrderModel.from_temporal_graph(g, delta=delta, max_order=order)
        else:
            walks = [rng.integers(0, n, int(rng.integers(2, 7))).tolist() for _ in range(int(rng.integers(5, 120)))]
            walks.append(list(range(n)))                             # every node occurs: dense first-order ids
            weights = rng.integers(1, 4, len(walks)).astype(np.float32).tolist()
            want = om.layers_from_paths(om.walks_to_path_tensors(walks, weights), max_order=order)
            paths = pp.PathData(device=DEV)
            paths.append_walks(walks, weights)
            model = pp.MultiOrderModel.from_path_data(paths, max_order=order)
        n_ho = want[order]["num_nodes"]
        widths = [(16, 32, 8), (12, 20, 6), (64, 64, 16), (8, 8, 8)][case % 4]
        one_hot = case % 5 == 0 and n_ho < 400
        gen = torch.Generator().manual_seed(case)
        fx, fh = (n, n_ho) if one_hot else (int(rng.integers(3, 20)), int(rng.integers(3, 20)))
        x = None if one_hot else torch.randn(n, fx, generator=gen)
        x_h = None if one_hot else torch.randn(n_ho, fh, generator=gen)
        y = torch.randint(0, 3, (n,), generator=gen)
        params = od.init_params(3, (fx, fh), list(widths), seed=case)
        ref = om.dbgnn_inputs(want, order, mapping, x=x, x_h=x_h)
        want_out, want_loss, want_grads = od.loss_and_grads(params, ref, y)
        data = model.to_dbgnn_data(max_order=order, mapping=mapping, x=None if x is None else x.to(DEV), x_h=None if x_h is None else x_h.to(DEV))
        net = _to_module(pp, params, 3, (fx, fh), list(widths))
        out = net(data)
        loss = F.cross_entropy(out, y.to(DEV))
        loss.backward()
        scale = float(want_out.abs().max()) + 1e-12
        torch.testing.assert_close(out.detach().cpu(), want_out, rtol=RTOL * 10, atol=max(ATOL, 2e-6 * scale)), case
        torch.testing.assert_close(loss.detach().cpu(), want_loss, rtol=RTOL * 10, atol=ATOL)
        for name, p in net.named_parameters():
            assert_gradients_close(p.grad, want_grads[name], f"{case} {name}")


@pytest.mark.parametrize("f,hidden", [((16, 16), [32, 16, 8]), ((64, 64), [64, 64, 64]), ((128, 64), [128, 128, 64]), ((5, 7), [6, 10, 3])])
def test_dbgnn_training_with_dropout_matches_masked_reference(pp, f, hidden):
    """p_dropout > 0 in training mode (reference dbgnn.py:131-150; its own test uses 0.4): the layers stay on the fused kernels; dropout and
    the dropout + ELU backward are one element-wise kernel each (pp_dropout_f32 / pp_dropout_act_backward_f32) whose keep decision is a pure
    function of (seed, call site, row, column).  Logits, loss and every gradient must match a torch-CPU evaluation of the reference forward
    with the same masks (pathpyg_amd.nn.sharded.dropout_mask is the torch statement of the hash)."""
    from oracle import dbgnn as od
    import pathpyg_amd.nn.dbgnn as mod
    from pathpyg_amd.nn.sharded import dropout_mask
    p = 0.4
    data, y = _bundle(11, 300, 4000, 1200, 5000, f, "last")
    params = od.init_params(3, f, hidden, seed=2)
    model = _to_module(pp, params, 3, f, hidden)
    model.p_dropout = p
    model.train()
    torch.manual_seed(77)
    seed = mod._draw_seed()
    torch.manual_seed(77)                      # the model's forward draws the same seed
    gdata = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()})
    out = model(gdata)
    loss = F.cross_entropy(out, y.to(DEV))
    loss.backward()

    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    site = {"fo": mod.TAG_FO, "ho": mod.TAG_HO}
    drop = lambda t, tag: t * dropout_mask(torch.arange(t.size(0)), t.size(1), p, seed, tag)
    x, x_h = data["x"], data["x_h"]
    n_gcn = len(hidden) - 1
    for i in range(n_gcn):
        x = F.elu(od.gcn_conv(drop(x, site["fo"] + i), data["edge_index"], data["edge_weights"], leaves[f"first_order_layers.{i}.lin.weight"],
                              leaves[f"first_order_layers.{i}.bias"]))
    x = drop(x, mod.TAG_FO_OUT)
    for i in range(n_gcn):
        x_h = F.elu(od.gcn_conv(drop(x_h, site["ho"] + i), data["edge_index_higher_order"], data["edge_weights_higher_order"],
                                leaves[f"higher_order_layers.{i}.lin.weight"], leaves[f"higher_order_layers.{i}.bias"]))
    x_h = drop(x_h, mod.TAG_HO_OUT)
    x = F.elu(od.bipartite_op(x_h, x, data["bipartite_edge_index"], data["num_nodes"], leaves["bipartite_layer.lin1.weight"],
                              leaves["bipartite_layer.lin1.bias"], leaves["bipartite_layer.lin2.weight"], leaves["bipartite_layer.lin2.bias"]))
    want_out = drop(x, mod.TAG_HEAD) @ leaves["lin.weight"].t() + leaves["lin.bias"]
    want_loss = F.cross_entropy(want_out, y)
    want_loss.backward()
    scale = float(want_out.detach().abs().max()) + 1e-12
    torch.testing.assert_close(out.detach().cpu(), want_out.detach(), rtol=RTOL, atol=max(ATOL, 2e-6 * scale))
    torch.testing.assert_close(loss.detach().cpu(), want_loss.detach(), rtol=RTOL, atol=ATOL)
    for name, prm in model.named_parameters():
        gs = float(leaves[name].grad.abs().max()) + 1e-12
        torch.testing.assert_close(prm.grad.cpu(), leaves[name].grad, rtol=RTOL * 10, atol=max(ATOL, 2e-5 * gs), msg=lambda s_: f"{name}: {s_}")
    # and the fused layer kernels really ran (not the dense + spmm pair) for the supported widths
    if f[0] % 16 == 0:
        assert mod._GcnLayer.supported(mod._hip.gcn_plan(gdata.edge_index, gdata.edge_weights, data["num_nodes"]), gdata.x,
                                       model.first_order_layers[0].lin.weight)


@pytest.mark.parametrize("n,e,p_in,q_out", [(17, 40, 16, 32), (3001, 9000, 64, 64), (4097, 20_000, 32, 16), (2000, 9000, 128, 128), (1500, 6000, 64, 128)])
def test_fused_layer_kernels_with_dropout_in_their_epilogues(pp, n, e, p_in, q_out):
    """pp_gcn_forward_drop_f32 / pp_gcn_backward_drop_f32 / pp_gcn_input_grad_drop_f32: the dropout fused into the epilogue drops exactly the
    elements pp_dropout_f32 drops (same counter-based mask, bit-identical values), and the backward kernels' d_in equals the unfused
    composition (linear input gradient, then pp_dropout_act_backward_f32)."""
    from pathpyg_amd import _hip
    g = torch.Generator().manual_seed(n + e + p_in)
    row = torch.sort(torch.randint(0, n, (e,), generator=g)).values
    ptr = torch.zeros(n + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0).int()
    idx = torch.randint(0, n, (e,), generator=g, dtype=torch.int32)
    val = torch.rand(e, generator=g)
    sc = torch.rand(n, generator=g)
    x = torch.randn(n, p_in, generator=g)
    w = torch.randn(q_out, p_in, generator=g) / p_in ** 0.5
    b = torch.randn(q_out, generator=g)
    site = (0.4, 987654321, 65, 2 ** 33 + 5)
    assert _hip.gcn_drop_supported(p_in, q_out) and not _hip.gcn_drop_supported(256, 256)
    dev = lambda t: t.to(DEV)
    y0 = _hip.gcn_forward(dev(ptr), dev(idx), dev(val), n, dev(x), dev(sc), dev(w), dev(b), True)
    y1 = _hip.gcn_forward(dev(ptr), dev(idx), dev(val), n, dev(x), dev(sc), dev(w), dev(b), True, drop=site)
    assert torch.equal(y1, _hip.dropout(y0, *site))
    # backward of a layer [q_out -> p_in here: weight m x k] whose INPUT is the dropped activation y1 (k = q_out columns)
    m, k = p_in, q_out
    dpre = torch.randn(n, m, generator=g)
    wb = torch.randn(m, k, generator=g) / m ** 0.5
    if _hip.gcn_fused_supported(k, m) == 1:
        lin, _, dw0 = _hip.gcn_backward(dev(ptr), dev(idx), dev(val), n, dev(dpre), dev(sc), y1, dev(wb), False, False)
        got, colsum, dw1 = _hip.gcn_backward(dev(ptr), dev(idx), dev(val), n, dev(dpre), dev(sc), y1, dev(wb), True, True, drop=site)
        torch.testing.assert_close(dw1, dw0, rtol=1e-6, atol=1e-6)
    else:
        lin, _ = _hip.gcn_input_grad(dev(ptr), dev(idx), dev(val), n, dev(dpre), dev(sc), dev(wb), None, False)
        got, colsum = _hip.gcn_input_grad(dev(ptr), dev(idx), dev(val), n, dev(dpre), dev(sc), dev(wb), y1, True, drop=site)
    want, want_sum = _hip.dropout_act_backward(lin, y1, *site, None, True, True)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(colsum, want_sum, rtol=1e-4, atol=1e-4 * float(want.abs().sum(0).max() + 1))


@pytest.mark.parametrize("n,f", [(1, 4), (1000, 8), (4097, 64), (777, 37), (300, 256)])
def test_bip_combine_kernels_match_the_elementwise_chain(n, f):
    """pp_bip_combine_f32 / _backward_f32 against the reference's element-wise tail (nn/dbgnn.py:66-69,143-144 after the lin1 re-association):
    ELU(a + deg * (p + b)), its input gradients and the bias gradient."""
    from pathpyg_amd.nn.dbgnn import bip_combine
    g = torch.Generator().manual_seed(n * 131 + f)
    a, p = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
    deg = torch.randint(0, 9, (n,), generator=g).float()
    b = torch.randn(f, generator=g)
    dy = torch.randn(n, f, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (a, p, b)]
    ref = torch.nn.functional.elu(torch.addcmul(ref_in[0], deg.double().unsqueeze(1), ref_in[1] + ref_in[2]))
    ref.backward(dy.double())
    got_in = [t.to(DEV).requires_grad_(True) for t in (a, p, b)]
    got = bip_combine(got_in[0], got_in[1], deg.to(DEV), got_in[2])
    got.backward(dy.to(DEV))
    torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=2e-6)
    for name, x, y in zip("apb", got_in, ref_in):
        scale = float(y.grad.abs().max()) + 1e-30
        torch.testing.assert_close(x.grad.cpu().double(), y.grad, rtol=1e-5, atol=2e-6 * max(scale, 1.0), msg=lambda s_: f"d{name}: {s_}")
    no_bias = bip_combine(a.to(DEV), p.to(DEV), deg.to(DEV), None)
    torch.testing.assert_close(no_bias.cpu().double(), torch.nn.functional.elu(torch.addcmul(a.double(), deg.double().unsqueeze(1), p.double())), rtol=1e-5, atol=2e-6)


# ---------------------------------------------------------------------------------------------------------------------------------
# pathpyg_amd.nn.optim.Adam (pp_adam_f32: one launch over all parameter tensors) against torch.optim.Adam, the optimizer of the
# reference's training loops (docs/tutorial/netzschleuder.ipynb:2480 — with weight decay).  fp32: 2e-6 relative + 1e-6 absolute after 6 steps.
@pytest.mark.parametrize("weight_decay", [0.0, 5e-4])
@pytest.mark.parametrize("n_tensors", [3, 30])            # 30 > the 24 tensors of one launch
def test_adam_matches_torch_adam(pp, weight_decay, n_tensors):
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 64), (64,), (8, 64), (1,), (0,), (5, 7, 3)]
    shapes = [shapes[i % len(shapes)] for i in range(n_tensors)]
    init = [torch.randn(s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    opt = pp.nn.optim.Adam(mine, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    opt_ref = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    for step in range(6):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if step == 2 and i == 1:
                a.grad = b.grad = None                      # a parameter without a gradient is skipped (and keeps its own step count)
                continue
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** (i % 3 - 1))
            a.grad, b.grad = gr.clone(), gr.clone()
        opt.step()
        opt_ref.step()
        for a, b in zip(mine, ref):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=1e-6)      # (parameters are O(1): a few ulps)
    sd = opt.state_dict()
    assert len(sd["state"]) == n_tensors and sd["param_groups"][0]["lr"] == 1e-2


def test_adam_trains_the_dbgnn_like_torch_adam(pp):
    """Three train steps of the DBGNN with either optimizer: same losses, same parameters."""
    data, y = _bundle(5, 40, 200, 120, 500, (16, 16))
    losses = {}
    params = {}
    for name in ("hip", "torch"):
        torch.manual_seed(3)
        net = pp.nn.DBGNN(num_classes=3, num_features=(16, 16), hidden_dims=[16, 32, 8], p_dropout=0.0).to(DEV)
        opt = (pp.nn.optim.Adam if name == "hip" else torch.optim.Adam)(net.parameters(), lr=5e-3, weight_decay=5e-4)
        d = pp.Data(**{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}, y=y.to(DEV))
        losses[name] = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(net(d), d.y)
            loss.backward()
            opt.step()
            losses[name].append(float(loss.detach()))
        params[name] = [p.detach().clone() for p in net.parameters()]
    assert losses["hip"] == pytest.approx(losses["torch"], rel=1e-5)
    for a, b in zip(params["hip"], params["torch"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
