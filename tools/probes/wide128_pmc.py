"""rocprofv3 --pmc target: a few launches of the 128 x 128 fused layer (forward) on the bench's higher-order graph shape (synthetic CSR:
10^6..10^7 rows, ~1.9 neighbours per row).  PP_WIDE_STREAMED=1 routes the shape to k_wide_layer (resident weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
f = 128
g = torch.Generator(device="cuda").manual_seed(0)
deg = torch.randint(0, 4, (n,), generator=g, device="cuda")
ptr = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
ptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
nnz = int(ptr[-1])
idx = torch.randint(0, n, (nnz,), generator=g, device="cuda").to(torch.int32)
val = torch.rand(nnz, generator=g, device="cuda")
x = torch.randn(n, f, generator=g, device="cuda")
w = torch.randn(f, f, generator=g, device="cuda") / 8
b = torch.randn(f, generator=g, device="cuda")
sc = torch.rand(n, generator=g, device="cuda")
for _ in range(3):
    y = _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True)
torch.cuda.synchronize()
print(float(y.abs().mean()))
