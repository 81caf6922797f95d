"""Time and check pp_weight_grad_f32 on the wide shapes (dW = dH^T X, dH [n, M], X [n, K]).  Run on the GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for m, k in ((256, 256), (128, 128), (256, 128), (128, 256), (64, 256)):
    n_small = 70_001
    dh, x = torch.randn(n_small, m, generator=g, device=dev), torch.randn(n_small, k, generator=g, device=dev)
    dw, db = _hip.weight_grad(dh, x, want_bias=True)
    ref = dh.double().t() @ x.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    errb = float((db.double() - dh.double().sum(0)).abs().max())
    n = 10_000_000 if m * k >= 256 * 128 else 10_000_000
    dh, x = torch.randn(n, m, generator=g, device=dev), torch.randn(n, k, generator=g, device=dev)
    for _ in range(2):
        _hip.weight_grad(dh, x, want_bias=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _hip.weight_grad(dh, x, want_bias=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"weight_grad {m}x{k} n={n}: {ms:7.3f} ms  {2 * n * m * k / ms / 1e9:6.1f} TFLOP/s  {4 * n * (m + k) / ms / 1e6:7.1f} GB/s   rel err {err:.2e} bias err {errb:.2e}")
    del dh, x
