#!/usr/bin/env python3
"""Time the classifier head's weight gradient (rows x 64 against rows x 8/16) — the shape the general k_weight_grad serves."""
import sys, torch
sys.path.insert(0, ".")
from pathpyg_amd import _hip
dev = torch.device("cuda:0")
for n, m, k in ((500_000, 8, 64), (500_000, 16, 256), (10_000_000, 8, 64)):
    dh = torch.randn(n, m, device=dev); x = torch.randn(n, k, device=dev)
    for _ in range(3): _hip.weight_grad(dh, x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _hip.weight_grad(dh, x, True)
    e1.record(); torch.cuda.synchronize()
    print(f"median-ish weight_grad {n} x {m} x {k}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")
