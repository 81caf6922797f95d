import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip
torch.manual_seed(0)
for f in (16, 32, 64, 128):
    n = 70
    deg = torch.tensor([0, 1, 2, 3, 4, 5, 600, 9, 17, 33, 40, 70, 130, 2000] * 5)
    ptr = torch.zeros(n + 1, dtype=torch.int32); ptr[1:] = torch.cumsum(deg, 0)
    nnz = int(ptr[-1])
    idx = torch.randint(0, n, (nnz,)).to(torch.int32)
    val = torch.rand(nnz) + 0.5
    x = torch.randn(n, f); w = torch.eye(f); sc = torch.rand(n) + 0.5
    h = _hip.HeavyRows(ptr.cuda(), n)
    y = _hip.gcn_forward(ptr.cuda(), idx.cuda(), val.cuda(), n, x.cuda(), sc.cuda(), w.cuda(), None, False, heavy=h).cpu()
    want = (sc[:, None] * x).double()
    for r in range(n):
        sl = slice(int(ptr[r]), int(ptr[r + 1]))
        want[r] += (val[sl].double()[:, None] * x[idx[sl].long()].double()).sum(0)
    err = (y.double() - want).abs().max(1).values
    print(f, "n_heavy", h.n_heavy, "bad rows (row:deg:err):", [f"{r}:{int(deg[r])}:{float(e):.3f}" for r, e in enumerate(err) if e > 2e-3])
