"""dense_w (padded / blocked dense layers) against torch on a grid of widths: which (p, q) disagree?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd.nn.dbgnn import dense_w
from pathpyg_amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bad = []
for p in (7, 8, 16, 20, 44, 64, 100, 128, 200, 256, 300, 520):
    for q in (5, 16, 24, 44, 64, 100, 128, 256, 300):
        x = torch.randn(333, p, generator=g, device=dev)
        w = torch.randn(q, p, generator=g, device=dev) / p ** 0.5
        b = torch.randn(q, generator=g, device=dev)
        got = dense_w(x, w, b)
        want = (x.double() @ w.double().t() + b.double()).float()
        err = float((got - want).abs().max())
        if err > 1e-4:
            bad.append((p, q, _hip.dense_supported(p, q), err))
print("bad:", bad)
