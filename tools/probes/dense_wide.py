"""Time and check the dense mode of the wide layers (pp_dense_f32 -> pp_wide_layer_f32 with ptr == NULL): forward x W^T + b and the
input-gradient form (g W) * ELU'(y) + column sums."""
import sys
import torch
sys.path.insert(0, ".")
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for p, q in ((256, 256), (128, 128), (256, 128), (128, 256), (64, 256), (256, 64)):
    n_small = 70_001
    a = torch.randn(n_small, p, generator=g, device=dev)
    w = torch.randn(q, p, generator=g, device=dev) * 0.1
    b = torch.randn(q, generator=g, device=dev)
    y, _ = _hip.dense(a, w, True, b)
    ref = a.double() @ w.double().t() + b.double()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    act = torch.randn(n_small, q, generator=g, device=dev)
    wk = torch.randn(p, q, generator=g, device=dev) * 0.1          # [P, Q]: k-major, the input-gradient case
    d, cs = _hip.dense(a, wk, False, None, act, True)
    refd = (a.double() @ wk.double()) * torch.where(act > 0, torch.ones_like(act), act + 1).double()
    errd = float((d.double() - refd).abs().max() / refd.abs().max())
    errc = float((cs.double() - refd.sum(0)).abs().max() / refd.sum(0).abs().max())
    n = 10_000_000
    a = torch.randn(n, p, generator=g, device=dev)
    for _ in range(2):
        _hip.dense(a, w, True, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _hip.dense(a, w, True, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"dense {p}->{q} n={n}: {ms:7.3f} ms  {2 * n * p * q / ms / 1e9:6.1f} TFLOP/s  {4 * n * (p + q) / ms / 1e6:7.1f} GB/s   rel err fwd {err:.2e} grad {errd:.2e} colsum {errc:.2e}")
    del a
