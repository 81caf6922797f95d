"""What does running the first-order kernels of a step on a SECOND HIP stream beside the higher-order kernels buy?  
Pairs (A on the main stream, B on a side stream) timed serially and concurrently, for several launch shares of A's persistent grid
(pp_set_launch_share): headline stream, real plans of build_dbgnn_shard at world size 1."""
import sys
import torch
sys.path.insert(0, ".")
import pathpyg_amd as pp
from pathpyg_amd import _hip, distributed as ppd
from pathpyg_amd._lib import lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
m, n, span, delta, f = 10_000_000, 500_000, 10_000_000, 1_000_000, 64
g = torch.Generator(device=dev).manual_seed(1)
ei = torch.randint(0, n, (2, m), generator=g, device=dev)
t = torch.randint(0, span, (m,), generator=g, device=dev)
tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
n_ho = int(pp.MultiOrderModel.from_temporal_graph(tg, delta=1, max_order=1).layers[1].m)
x = torch.randn(n, f, generator=g, device=dev)
x_h = torch.randn(n_ho, f, generator=g, device=dev)
y = torch.randint(0, 8, (n,), generator=g, device=dev)
shard = ppd.build_dbgnn_shard(tg, delta, x, x_h, y, ppd.Comm()).resolve()
fo, ho, bip = shard.fo.plan, shard.ho.plan, shard.bip
w = torch.randn(f, f, generator=g, device=dev) * 0.1
b = torch.zeros(f, device=dev)
d_ho = torch.randn(n_ho, f, generator=g, device=dev)
d_fo = torch.randn(n, f, generator=g, device=dev)
L = lib()
side = torch.cuda.Stream()


def ho_fwd():
    _hip.gcn_forward(ho.fwd_ptr, ho.fwd_idx, ho.fwd_val, ho.n_dst, x_h, ho.self_coef, w, b, True)


def fo_fwd():
    _hip.gcn_forward(fo.fwd_ptr, fo.fwd_idx, fo.fwd_val, fo.n_dst, x, fo.self_coef, w, b, True)


def ho_bwd():
    _hip.gcn_backward(ho.bwd_ptr, ho.bwd_idx, ho.bwd_val, ho.n_src, d_ho, ho.self_coef, x_h, w, True, True)


def fo_bwd():
    _hip.gcn_backward(fo.bwd_ptr, fo.bwd_idx, fo.bwd_val, fo.n_src, d_fo, fo.self_coef, x, w, True, True)


def bip_bwd():
    _hip.spmm_act_backward(bip.bwd_ptr, bip.bwd_idx, bip.bwd_val, bip.n_src, d_fo, x_h, True)


def wgrad():
    _hip.weight_grad(d_ho, x_h, want_bias=False)


def bip_fwd():
    _hip.spmm(bip.fwd_ptr, bip.fwd_idx, bip.fwd_val, bip.n_dst, x_h)


# first-order edge list (row-sorted): sources from the row pointers of the source-major CSR, destinations = bwd_idx
src_fo = torch.repeat_interleave(torch.arange(n, device=dev), (fo.bwd_ptr[1:] - fo.bwd_ptr[:-1]).long())
ei_fo = torch.stack((src_fo, fo.bwd_idx.long())).contiguous()
w_fo = torch.rand(ei_fo.size(1), generator=g, device=dev)
src_ho = torch.repeat_interleave(torch.arange(n_ho, device=dev), (ho.bwd_ptr[1:] - ho.bwd_ptr[:-1]).long())
ei_ho = torch.stack((src_ho, ho.bwd_idx.long())).contiguous()
w_ho = torch.rand(ei_ho.size(1), generator=g, device=dev)


def fo_plan():
    _hip.gcn_plan(ei_fo, w_fo, n, True, [], want_dst_order=True)


def ho_plan():
    _hip.gcn_plan(ei_ho, w_ho, n_ho, True, [])


def timed(fn_main, fn_side, share_main, share_side, concurrent, reps=6):
    def once():
        if concurrent:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                L.pp_set_launch_share(share_side)
                fn_side()
            L.pp_set_launch_share(share_main)
            fn_main()
            torch.cuda.current_stream().wait_stream(side)
        else:
            L.pp_set_launch_share(1000)
            fn_side()
            fn_main()
        L.pp_set_launch_share(1000)
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def alone(fn, reps=6):
    return timed(fn, lambda: None, 1000, 1000, False, reps)


pairs = [("ho fwd layer | 2 x fo fwd layer", ho_fwd, lambda: (fo_fwd(), fo_fwd())),
         ("2 x ho fwd layer | 2 x fo fwd layer", lambda: (ho_fwd(), ho_fwd()), lambda: (fo_fwd(), fo_fwd())),
         ("ho bwd layer | fo bwd layer", ho_bwd, fo_bwd),
         ("bipartite bwd | fo bwd layer", bip_bwd, fo_bwd),
         ("weight_grad64 ho | fo bwd layer", wgrad, fo_bwd),
         ("bipartite fwd (spmm) | fo fwd layer", bip_fwd, fo_fwd),
         ("ho plan | fo plan", ho_plan, fo_plan),
         ("ho plan | 2 x fo fwd layer", ho_plan, lambda: (fo_fwd(), fo_fwd()))]
for name, a, bfn in pairs:
    ta, tb = alone(a), alone(bfn)
    ser = timed(a, bfn, 1000, 1000, False)
    line = f"{name:42s} A {ta:8.1f} us  B {tb:8.1f} us  serial {ser:8.1f} us | concurrent:"
    for sm, ss in ((1000, 1000), (875, 1000), (750, 1000), (750, 250), (625, 375), (500, 500)):
        line += f"  {sm}/{ss}: {timed(a, bfn, sm, ss, True):8.1f}"
    print(line, flush=True)
