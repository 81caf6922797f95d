import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("w", "/root/repo/tools/probes/wide_ws.py"); w = importlib.util.module_from_spec(spec); spec.loader.exec_module(w)
from pathpyg_amd import _hip
dev = w.dev
n = 10_000_000
ptr, idx, val, nnz = w.csr(n, float(os.environ.get("DEG", "0")), 1)
x = torch.randn(n, 256, device=dev) * float(os.environ.get("XSCALE", "1")); wt = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev); sc = torch.rand(n, device=dev)
out = torch.empty(n, 256, device=dev)
t = w.timeit(lambda: _hip.gcn_forward(ptr, idx, val, n, x, sc, wt, b, True, out=out))
print("dbg", os.environ.get("PP_WS_DBG"), "deg", os.environ.get("DEG", "0"), f"{t:.3f} ms")
