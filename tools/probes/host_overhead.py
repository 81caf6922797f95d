"""Host cost of one C-ABI call through the Python shims (what bounds small graphs and the per-rank step of a partitioned run)."""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
from pathpyg_amd import _hip

dev = torch.device("cuda:0")
x = torch.randn(64, 64, device=dev)
idx = torch.randint(0, 16, (64,), device=dev)
ptr = (torch.arange(17, device=dev) * 4).to(torch.int32)
i32 = idx.to(torch.int32)
val = torch.rand(64, device=dev)


def many(n):
    for _ in range(n):
        _hip.spmm(ptr, i32, val, 16, x)


many(200)
torch.cuda.synchronize()
t0 = time.perf_counter()
many(2000)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"_hip.spmm on a 16-row graph: {(t1 - t0) / 2000 * 1e6:.1f} us of host time per call")
t0 = time.perf_counter()
for _ in range(2000):
    torch.empty((16, 64), device=dev)
t1 = time.perf_counter()
print(f"torch.empty: {(t1 - t0) / 2000 * 1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(2000):
    with torch.cuda.device(dev):
        pass
t1 = time.perf_counter()
print(f"with torch.cuda.device(dev): {(t1 - t0) / 2000 * 1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(2000):
    torch.cuda.current_stream().cuda_stream
t1 = time.perf_counter()
print(f"current_stream().cuda_stream: {(t1 - t0) / 2000 * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
many(2000)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
