"""Calibrate rocprofv3 FETCH_SIZE for the SpMM gather pattern: every row gathers exactly ONE distinct random source row
(a permutation), no self term -> the kernel must fetch each 256-byte row once: 4*(n+1) + 8n + 256n bytes."""
import sys, torch
sys.path.insert(0, '/root/repo')
from pathpyg_amd import _hip
dev = torch.device('cuda:0')
n, f = 10_000_000, 64
g = torch.Generator(device=dev).manual_seed(0)
perm = torch.randperm(n, generator=g, device=dev).int()
ptr = torch.arange(n + 1, device=dev, dtype=torch.int32)
val = torch.ones(n, device=dev)
x = torch.randn(n, f, device=dev)
for _ in range(3):
    y = _hip.spmm(ptr, perm, val, n, x)
torch.cuda.synchronize()
print("expected fetch bytes", 4 * (n + 1) + 8 * n + 4 * f * n, "write", 4 * f * n)
