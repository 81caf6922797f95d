import sys, torch
sys.path.insert(0, '/root/repo')
import pathpyg_amd as pp
from pathpyg_amd import _hip
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
m, n = 10_000_000, 500_000
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, 10_000_000, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
ho = _hip.temporal_lift(tg.data.edge_index, tg.data.time, n, 1_000_000)
for name, cnt in (("temporal", torch.bincount(ho[0], minlength=m)), ("linegraph", torch.bincount(ho[0], minlength=m)[ho[1]])):
    off = torch.zeros(cnt.numel() + 1, dtype=torch.long, device=dev); off[1:] = torch.cumsum(cnt, 0)
    total = int(off[-1])
    tiles = torch.arange(0, total, 512, device=dev)
    ts = torch.searchsorted(off, tiles, right=True) - 1
    nb = ts[1:] - ts[:-1] + 1
    print(name, "total", total, "tiles", tiles.numel(), "n_bound mean %.1f max %d frac>513 %.4f" % (nb.float().mean().item(), int(nb.max()), float((nb > 513).float().mean())), "max count", int(cnt.max()))
