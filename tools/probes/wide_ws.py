"""256 x 256 GCN layer (gather + X W^T + ELU) and its input gradient on a De-Bruijn-shaped CSR: check against float64 torch at a small
size, time at 10^7 rows.  usage: python tools/probes/wide_ws.py [rows] [avg degree]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip

dev = torch.device("cuda:0")


def csr(n, deg, seed, n_src=None):
    g = torch.Generator(device=dev).manual_seed(seed)
    n_src = n if n_src is None else n_src
    counts = torch.poisson(torch.full((n,), float(deg), device=dev), generator=g).to(torch.int64)
    counts[:: 1000] = 70                                      # some rows beyond the first-four batch and beyond one 64-lane chunk
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] = counts.cumsum(0)
    nnz = int(ptr[-1])
    idx = torch.randint(0, n_src, (nnz,), generator=g, device=dev)
    val = torch.rand(nnz, generator=g, device=dev) + 0.1
    return ptr.to(torch.int32), idx.to(torch.int32), val, nnz


def check(n, n_src, n_self):
    ptr, idx, val, nnz = csr(n, 1.9, 3, n_src)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n_src, 256, generator=g, device=dev)
    w = torch.randn(256, 256, generator=g, device=dev) / 16
    b = torch.randn(256, generator=g, device=dev)
    sc = torch.rand(n, generator=g, device=dev)
    sc[n_self:] = 0                                           # (forward: one coefficient per row; the gradient call stops its self term at n_self)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())
    agg = torch.zeros(n, 256, dtype=torch.float64, device=dev)
    agg.index_add_(0, rows, val.double().unsqueeze(1) * x.double()[idx.long()])
    agg += sc.double().unsqueeze(1) * x.double()[:n]
    want = torch.nn.functional.elu(agg @ w.double().t() + b.double())
    y, a = _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True, want_agg=True)
    e1 = float((y.double() - want).abs().max() / want.abs().max())
    e2 = float((a.double() - agg).abs().max() / agg.abs().max())
    # input gradient: (A d + sc d) @ W * elu'(act)
    d = torch.randn(n_src, 256, generator=g, device=dev)
    act = torch.nn.functional.elu(torch.randn(n, 256, generator=g, device=dev))
    gsum = torch.zeros(n, 256, dtype=torch.float64, device=dev)
    gsum.index_add_(0, rows, val.double().unsqueeze(1) * d.double()[idx.long()])
    gsum[:n_self] += sc[:n_self].double().unsqueeze(1) * d.double()[:n_self]
    fac = torch.where(act > 0, torch.ones_like(act), act + 1).double()
    want_g = (gsum @ w.double()) * fac
    got, cs = _hip.gcn_input_grad(ptr, idx, val, n, d, sc, w, act, True, n_self=n_self)
    e3 = float((got.double() - want_g).abs().max() / want_g.abs().max())
    e4 = float((cs.double() - want_g.sum(0)).abs().max() / want_g.sum(0).abs().max())
    print(f"check n={n} n_src={n_src} n_self={n_self}: forward {e1:.2e} agg {e2:.2e} input-grad {e3:.2e} colsum {e4:.2e}")
    assert max(e1, e2, e3) < 2e-6 and e4 < 2e-5


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


if __name__ == "__main__":
    for shape in ((1, 1, 1), (63, 63, 63), (64, 64, 64), (1000, 1000, 1000), (70001, 70001, 70001), (50000, 90000, 50000), (50000, 90000, 30000)):
        check(*shape)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    deg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.9
    ptr, idx, val, nnz = csr(n, deg, 1)
    x = torch.randn(n, 256, device=dev)
    w = torch.randn(256, 256, device=dev) / 16
    b = torch.randn(256, device=dev)
    sc = torch.rand(n, device=dev)
    out = torch.empty(n, 256, device=dev)
    flops = 2.0 * n * 256 * 256
    t = timeit(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True, out=out))
    print(f"gcn_forward 256x256 rows={n} nnz={nnz}: {t:.3f} ms  {flops / t / 1e9:.1f} TFLOP/s ({flops / t / 1e9 / 157.3:.2f} of the fp32 matrix peak)")
    t = timeit(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True, want_agg=True))
    print(f"gcn_forward keeping A x: {t:.3f} ms")
    act = torch.nn.functional.elu(torch.randn(n, 256, device=dev))
    t = timeit(lambda: _hip.gcn_input_grad(ptr, idx, val, n, x, sc, w, act, True))
    print(f"gcn_input_grad 256x256: {t:.3f} ms  {flops / t / 1e9:.1f} TFLOP/s")
