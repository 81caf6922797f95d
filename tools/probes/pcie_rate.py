"""PCIe-inclusive rate of the lift when the boundary hands over HOST buffers: CPU tensors in, CPU tensor out."""
import sys, time, torch
sys.path.insert(0, ".")
import pathpyg_amd as pp
m, n, span, delta = 10_000_000, 500_000, 10_000_000, 1_000_000
g = torch.Generator().manual_seed(0)
ei = torch.randint(0, n, (2, m), generator=g)
t = torch.sort(torch.randint(0, span, (m,), generator=g)).values
tg_dev = pp.TemporalGraph(pp.Data(edge_index=ei.cuda(), time=t.cuda(), num_nodes=n))
ei_s, t_s = tg_dev.data.edge_index.cpu(), tg_dev.data.time.cpu()
data = pp.Data(edge_index=ei_s, time=t_s, num_nodes=n)
class G: pass
gh = G(); gh.data = data
for pinned in (False, True):
    if pinned:
        gh.data = pp.Data(edge_index=ei_s.pin_memory(), time=t_s.pin_memory(), num_nodes=n)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ho = pp.algorithms.lift_order_temporal(gh, delta)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    e2 = ho.size(1)
    print(f"host buffers ({'pinned' if pinned else 'pageable'} inputs): E2={e2} best {min(ts)*1e3:.1f} ms -> {e2/min(ts)/1e9:.3f} G lifted k-edges/s; result on {ho.device}")
torch.cuda.synchronize(); t0 = time.perf_counter(); ho = pp.algorithms.lift_order_temporal(tg_dev, delta); torch.cuda.synchronize()
t0 = time.perf_counter(); ho = pp.algorithms.lift_order_temporal(tg_dev, delta); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"device-resident: {dt*1e3:.2f} ms -> {e2/dt/1e9:.2f} G lifted k-edges/s")
