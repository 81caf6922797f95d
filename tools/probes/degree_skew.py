"""pp_degree_i64 (PyG degree) on uniform and on skewed index vectors (a scale-free stream's targets: a tenth of 2 * 10^7 entries in ONE bin)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
n, m = 1_000_000, 20_000_000
uni = torch.randint(0, n, (m,), generator=g, device=dev)
zipf = (n * torch.rand(m, generator=g, device=dev, dtype=torch.float64).pow(6.0)).long().clamp_(max=n - 1)
srt = torch.sort(uni).values
for name, idx in (("uniform", uni), ("scale-free (one bin with a tenth)", zipf), ("sorted", srt)):
    ref = torch.bincount(idx, minlength=n)
    for _ in range(2):
        out = _hip.degree(idx, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = _hip.degree(idx, n)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 100:.3f} ms, equal to torch.bincount: {bool(torch.equal(out.long(), ref))}", flush=True)
