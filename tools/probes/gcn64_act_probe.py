"""Is k_gcn_forward<64,64> issue-bound?  Same launch with and without the ELU epilogue (~130 of ~700 VALU instructions per tile)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip
n, f = 10_000_000, 64
g = torch.Generator(device="cuda").manual_seed(0)
# De-Bruijn-like locality: rows in groups of 20 share 20 sources
grp = torch.arange(n, device="cuda") // 20
deg = torch.randint(0, 4, (n,), generator=g, device="cuda")
ptr = torch.zeros(n + 1, dtype=torch.int32, device="cuda"); ptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
nnz = int(ptr[-1])
row_of = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
perm_grp = torch.randperm(n // 20 + 1, generator=g, device="cuda")
idx = ((perm_grp[grp[row_of]] * 20 + torch.randint(0, 20, (nnz,), generator=g, device="cuda")) % n).to(torch.int32)
val = torch.rand(nnz, generator=g, device="cuda")
x = torch.randn(n, f, generator=g, device="cuda"); w = torch.randn(f, f, generator=g, device="cuda") / 8
b = torch.randn(f, generator=g, device="cuda"); sc = torch.rand(n, generator=g, device="cuda")
def t(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(k):
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
print("act=True ", t(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True)))
print("act=False", t(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, False)))
