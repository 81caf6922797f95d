"""How do the DBGNN kernels behave on a scale-free stream (a few hubs with 10^5+ in-events)?"""
import sys, time, torch
sys.path.insert(0, ".")
import pathpyg_amd as pp
dev = "cuda:0"
import os
m, n, span, delta, f = (int(x) for x in os.environ.get("HUB_SHAPE", "2000000,100000,2000000,20000,64").split(","))
for zipf in ((True,) if os.environ.get("HUB_ONLY") else (False, True)):
    g = torch.Generator(device=dev).manual_seed(1)
    src = torch.randint(0, n, (m,), generator=g, device=dev)
    if zipf:
        u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
        dst = (n * u.pow(6.0)).long().clamp_(max=n - 1)
    else:
        dst = torch.randint(0, n, (m,), generator=g, device=dev)
    t = torch.randint(0, span, (m,), generator=g, device=dev)
    tg = pp.TemporalGraph(pp.Data(edge_index=torch.stack((src, dst)), time=t, num_nodes=n))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=delta, max_order=2)
    torch.cuda.synchronize(); t_lift = time.perf_counter() - t0
    n_ho = mom.layers[2].n
    data = mom.to_dbgnn_data(max_order=2, x=torch.randn(n, f, device=dev), x_h=torch.randn(n_ho, f, device=dev))
    net = pp.nn.DBGNN(num_classes=8, num_features=(f, f), hidden_dims=[f, f, f]).to(dev)
    y = torch.randint(0, 8, (n,), device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = pp.nn.dbgnn.cross_entropy(net(data), y); loss.backward()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    deg1 = torch.bincount(mom.layers[1].data.edge_index[1], minlength=n).max().item()
    deg2 = torch.bincount(mom.layers[2].data.edge_index[1], minlength=n_ho).max().item()
    print(f"zipf={zipf}: layers {t_lift*1e3:.1f} ms, E2={mom.layers[2].data.edge_index.size(1)}, U2={n_ho}, max in-degree fo={deg1} ho={deg2}, train step {dt*1e3:.1f} ms")
    from pathpyg_amd import _hip
    for name, gl in (("fo", mom.layers[1]), ("ho", mom.layers[2])):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _hip.gcn_plan(gl.data.edge_index, gl.data.edge_weight, gl.n)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"   gcn_plan {name}: {min(ts)*1e3:.2f} ms")
