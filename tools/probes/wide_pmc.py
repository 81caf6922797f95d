"""rocprofv3 --pmc target: a few dense-mode launches of k_wide_layer<256,256,0> (pp_dense_f32 at 10^7 x 256 x 256) and rocBLAS on the same product."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
x = torch.randn(n, 256, device="cuda")
w = torch.randn(256, 256, device="cuda") / 16
b = torch.randn(256, device="cuda")
for _ in range(3):
    y = _hip.dense(x, w, True, b)[0]
    z = torch.nn.functional.linear(x, w, b)
torch.cuda.synchronize()
print(float((y - z).abs().max()))
