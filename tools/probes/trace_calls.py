import sys, torch
sys.path.insert(0, "/root/repo")
import pathpyg_amd as pp
from pathpyg_amd import _lib
L = _lib.lib()
names = ["pp_gcn_forward_f32", "pp_gcn_backward_f32", "pp_spmm_f32", "pp_dense_f32", "pp_dense_backward_f32", "pp_weight_grad_f32", "pp_spmm_act_backward_f32", "pp_act_backward_f32"]
log = []
for n in names:
    orig = getattr(L, n)
    def mk(n, orig):
        def w(*a):
            log.append((n, [x for x in a[3:5] if isinstance(x, int)][:1]))
            return orig(*a)
        return w
    setattr(L, n, mk(n, orig))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
m, N = 200000, 10000
ei = torch.randint(0, N, (2, m), generator=g, device=dev); t = torch.randint(0, 200000, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=N))
mom = pp.MultiOrderModel.from_temporal_graph(tg, delta=20000, max_order=2)
x = torch.randn(N, 64, device=dev); xh = torch.randn(mom.layers[2].n, 64, device=dev)
data = mom.to_dbgnn_data(max_order=2, x=x, x_h=xh)
net = pp.nn.DBGNN(num_classes=8, num_features=(64, 64), hidden_dims=[64, 64, 64]).to(dev)
out = net(data); print("fwd:", [(n, a) for n, a in log]); log.clear()
pp.nn.dbgnn.cross_entropy(out, torch.randint(0, 8, (N,), device=dev)).backward(); print("bwd:", [(n, a) for n, a in log])
