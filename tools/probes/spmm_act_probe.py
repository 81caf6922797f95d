"""Times pp_spmm_act_backward_f32 on the bench's bipartite shape (10^7 order-2 rows with one first-order target each, F = 64) and
checks it against torch.  Run on the GPU box: python tools/probes/spmm_act_probe.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
n, n_fo, f = 10_000_000, 500_000, 64
g = torch.Generator(device=dev).manual_seed(0)
ptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
idx = torch.randint(0, n_fo, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
d = torch.randn(n_fo, f, generator=g, device=dev)
z = torch.nn.functional.elu(torch.randn(n, f, generator=g, device=dev))
for _ in range(3):
    dx, cs = _hip.spmm_act_backward(ptr, idx, None, n, d, z, True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    a.record(); dx, cs = _hip.spmm_act_backward(ptr, idx, None, n, d, z, True); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
want = d[idx.long()] * torch.where(z > 0, torch.ones_like(z), z + 1)
print(f"spmm_act_backward median {ts[len(ts)//2]:.3f} ms best {ts[0]:.3f} ms  {(8 * n * f + 4 * n_fo * f) / ts[len(ts)//2] / 1e6:.0f} GB/s  max err {float((dx - want).abs().max()):.2e} colsum rel err {float(((cs - want.sum(0)).abs() / want.sum(0).abs().clamp_min(1)).max()):.2e}")
