"""Is the first-order layer kernel (5*10^5 rows x 20 neighbours) bound by where its gathers land, or by its own structure?
Same CSR shape, three neighbour patterns: random (the ER graph), a window of 64 rows around the destination, the same row 20 times."""
import sys
import torch
sys.path.insert(0, ".")
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
n, deg, f = 500_000, 20, 64
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, f, generator=g, device=dev)
w = torch.randn(f, f, generator=g, device=dev) * 0.1
b = torch.zeros(f, device=dev)
ptr = (torch.arange(n + 1, device=dev) * deg).to(torch.int32)
val = torch.rand(n * deg, generator=g, device=dev)
selfc = torch.rand(n, generator=g, device=dev)
rows = torch.arange(n, device=dev).repeat_interleave(deg)
pats = {"random": torch.randint(0, n, (n * deg,), generator=g, device=dev),
        "window64": (rows + torch.randint(-32, 32, (n * deg,), generator=g, device=dev)).clamp(0, n - 1),
        "same row": rows.clone(),
        "none (deg 0)": None}
for name, idx in pats.items():
    if idx is None:
        p, i32, v = torch.zeros(n + 1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
    else:
        p, i32, v = ptr, idx.to(torch.int32), val
    for _ in range(3):
        _hip.gcn_forward(p, i32, v, n, x, selfc, w, b, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _hip.gcn_forward(p, i32, v, n, x, selfc, w, b, True)
    e1.record()
    torch.cuda.synchronize()
    print(f"gcn_forward 5e5 x 20, neighbours {name:14s}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
