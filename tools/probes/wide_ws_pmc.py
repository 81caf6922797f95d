"""rocprofv3 --pmc target: six forward launches of the 256 x 256 GCN layer (k_wide_ws<0>) on a De-Bruijn-shaped CSR, no checks (the
measurement variants of the library compute nothing meaningful).  usage: python tools/probes/wide_ws_pmc.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathpyg_amd import _hip  # noqa: E402
from tools.probes.wide_ws import csr, timeit  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ptr, idx, val, nnz = csr(n, 1.9, 1)
x = torch.randn(n, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
b = torch.randn(256, device=dev)
sc = torch.rand(n, device=dev)
out = torch.empty(n, 256, device=dev)
t = timeit(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, w, b, True, out=out))
print(f"gcn_forward 256x256 rows={n} nnz={nnz}: {t:.3f} ms  {2.0 * n * 65536 / t / 1e9:.1f} TFLOP/s")
