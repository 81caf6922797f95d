"""GPU tests of the staged 64 x 64 layer kernel (pp_gcn_stage_plan_i32 / pp_gcn_forward_staged_f32): the stage plan against a torch
restatement of its definition (distinct sources of every 64-row group in order of first appearance), the layer against the plain kernel and
against a float64 evaluation of ``ELU((A x + diag(self) x) W^T + b)`` (reference: GCNConv inside nn/dbgnn.py:131-140) at the 1e-5 bar of
tests/tolerance.py.  The cases cover groups the plan cannot stage (more than 256 entries, more than pp_gcn_stage_slots() = 96 distinct sources), empty rows,
a ragged last group, rectangular graphs, missing values / self terms."""
import pytest
import torch

from tests.tolerance import assert_embeddings_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from pathpyg_amd import _hip
    return _hip


def _csr(seed, n_rows, n_src, mean, dense_rows=(), dense_len=300, dup_window=None):
    """Random CSR (int32 ptr / idx, fp32 val): row lengths Poisson(mean); ``dense_rows`` get ``dense_len`` entries; ``dup_window`` draws the
    sources of a 64-row group from a window of that many rows (repeats inside a group, as on a De Bruijn layer)."""
    g = torch.Generator().manual_seed(seed)
    length = torch.poisson(torch.full((n_rows,), float(mean)), generator=g).long()
    for r in dense_rows:
        length[r] = dense_len
    ptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(length, 0)
    nnz = int(ptr[-1])
    row_of = torch.repeat_interleave(torch.arange(n_rows), length)
    if dup_window:
        base = (row_of // 64) * 37 % max(n_src - dup_window, 1)
        idx = base + torch.randint(0, dup_window, (nnz,), generator=g)
    else:
        idx = torch.randint(0, n_src, (nnz,), generator=g)
    idx = idx.clamp_(max=n_src - 1)
    val = torch.rand(nnz, generator=g) + 0.1
    return ptr.to(torch.int32).to(DEV), idx.to(torch.int32).to(DEV), val.to(DEV), row_of.to(DEV)


def _reference(ptr, idx, val, row_of, n_rows, x, self_coef, w, b, act):
    agg = torch.zeros(n_rows, 64, dtype=torch.float64, device=DEV)
    if self_coef is not None:
        agg += self_coef.double()[:, None] * x[:n_rows].double()
    v = val.double() if val is not None else torch.ones(idx.numel(), dtype=torch.float64, device=DEV)
    agg.index_add_(0, row_of, v[:, None] * x[idx.long()].double())
    out = agg @ w.double().T + (b.double() if b is not None else 0.0)
    return torch.nn.functional.elu(out) if act else out


CASES = {
    # name: (n_rows, n_src, mean entries per row, dense rows, dup window)
    "repeats-inside-groups": (5000, 5000, 2.0, (), 40),
    "ragged-last-group": (1000 + 37, 1500, 1.9, (), 30),
    "no-repeats": (4096, 200000, 2.5, (), None),
    "dense-rows-fall-back": (3000, 3000, 1.5, (5, 700, 701, 2999), 25),
    "many-distinct-fall-back": (2048, 100000, 3.4, (), None),          # groups of ~218 entries, nearly all distinct: more sources than the stage has slots
    "tiny": (5, 9, 1.0, (), None),
    "empty-rows": (700, 700, 0.0, (), None),
}


@pytest.mark.parametrize("name", list(CASES))
def test_stage_plan_is_the_distinct_sources_of_every_group(hip, name):
    n_rows, n_src, mean, dense, window = CASES[name]
    ptr, idx, _val, row_of = _csr(11, n_rows, n_src, mean, dense, dup_window=window)
    sp = hip.gcn_stage_plan(ptr, idx, n_rows)
    sp2 = hip.gcn_stage_plan(ptr, idx, n_rows)
    slots = sp.grp_list.size(1)
    groups = (n_rows + 63) // 64
    cnt = sp.grp_cnt.cpu().long()
    lst, slot = sp.grp_list.cpu().long(), sp.slot.cpu().long()
    p, j = ptr.cpu().long(), idx.cpu().long()
    n_fb = int(sp.fallback[0])
    marked = set(sp.fallback[1:1 + n_fb].cpu().tolist())
    assert marked == set(torch.nonzero(cnt == 255).flatten().tolist())
    for gi in range(groups):
        e0, e1 = int(p[gi * 64]), int(p[min(gi * 64 + 64, n_rows)])
        seen = list(dict.fromkeys(j[e0:e1].tolist()))                       # distinct, in order of first appearance
        if e1 - e0 > 256 or len(seen) > slots:
            assert int(cnt[gi]) == 255, (name, gi)
            continue
        assert int(cnt[gi]) == len(seen), (name, gi)
        assert lst[gi, :len(seen)].tolist() == seen, (name, gi)
        assert torch.equal(lst[gi][slot[e0:e1]], j[e0:e1]), (name, gi)
    # the same plan on every run (the numbering does not depend on which lane wins a hash slot)
    keep = cnt < 255
    assert torch.equal(sp.grp_cnt, sp2.grp_cnt)
    assert torch.equal(sp.grp_list.cpu()[keep][torch.arange(slots)[None, :] < cnt[keep][:, None]],
                       sp2.grp_list.cpu()[keep][torch.arange(slots)[None, :] < cnt[keep][:, None]])


@pytest.mark.parametrize("act", [True, False], ids=["elu", "linear"])
@pytest.mark.parametrize("name", list(CASES))
def test_staged_layer_equals_the_plain_kernel_and_float64(hip, name, act):
    n_rows, n_src, mean, dense, window = CASES[name]
    ptr, idx, val, row_of = _csr(3, n_rows, n_src, mean, dense, dup_window=window)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(max(n_src, n_rows), 64, generator=g, device=DEV)
    w = torch.randn(64, 64, generator=g, device=DEV) / 8
    b = torch.randn(64, generator=g, device=DEV)
    self_coef = torch.rand(n_rows, generator=g, device=DEV)
    sp = hip.gcn_stage_plan(ptr, idx, n_rows)
    y = hip.gcn_forward_staged(ptr, idx, val, n_rows, x, self_coef, w, b, act, sp)
    plain = hip.gcn_forward(ptr, idx, val, n_rows, x, self_coef, w, b, act)
    assert_embeddings_close(y, plain, rtol=2e-6, what=f"{name}: staged vs plain kernel")
    # (rows of 300 terms: the fp32 row sum itself carries ~300 * 2^-24 of the terms' magnitude — the plain kernel deviates from float64 by the
    # same amount; staged against plain stays at 2e-6 above)
    rtol64 = 1e-4 if dense else 1e-5
    assert_embeddings_close(y, _reference(ptr, idx, val, row_of, n_rows, x, self_coef, w, b, act), rtol=rtol64, what=f"{name}: staged vs float64")


def test_staged_layer_without_values_self_terms_and_bias(hip):
    n_rows, n_src = 3000, 3000
    ptr, idx, val, row_of = _csr(8, n_rows, n_src, 2.0, (17,), dup_window=50)
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(n_src, 64, generator=g, device=DEV)
    w = torch.randn(64, 64, generator=g, device=DEV) / 8
    sp = hip.gcn_stage_plan(ptr, idx, n_rows)
    for v, sc, b in ((None, None, None), (val, None, None), (None, torch.rand(n_rows, device=DEV), torch.randn(64, device=DEV))):
        y = hip.gcn_forward_staged(ptr, idx, v, n_rows, x, sc, w, b, True, sp)
        assert_embeddings_close(y, _reference(ptr, idx, v, row_of, n_rows, x, sc, w, b, True), rtol=1e-4, what="staged layer, optional operands (one 300-term row)")
        assert_embeddings_close(y, hip.gcn_forward(ptr, idx, v, n_rows, x, sc, w, b, True), rtol=2e-6, what="staged vs plain kernel, optional operands")


def test_staged_layer_on_a_de_bruijn_layer_into_a_given_output(hip):
    import pathpyg_amd as pp
    g = torch.Generator(device=DEV).manual_seed(2)
    n, m = 3000, 60000
    ei = torch.randint(0, n, (2, m), generator=g, device=DEV)
    t = torch.randint(0, 60000, (m,), generator=g, device=DEV)
    tg = pp.TemporalGraph(pp.Data(edge_index=ei, time=t, num_nodes=n))
    plan = hip.debruijn2(tg.data.edge_index, tg.data.time, n, 6000, None).ho
    rows = plan.fwd_ptr.numel() - 1
    assert hip.gcn_stage_wanted(rows, rows, plan.fwd_idx.numel(), 64, 64)
    x = torch.randn(rows, 64, generator=g, device=DEV)
    w = torch.randn(64, 64, generator=g, device=DEV) / 8
    b = torch.randn(64, generator=g, device=DEV)
    sp = hip.gcn_stage_plan(plan.fwd_ptr, plan.fwd_idx, rows)
    assert int(sp.grp_cnt[sp.grp_cnt < 255].long().sum()) < plan.fwd_idx.numel()          # repeats inside the groups: fewer rows fetched than gathered
    out = torch.full((rows + 3, 64), float("nan"), device=DEV)
    y = hip.gcn_forward_staged(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, rows, x, plan.self_coef, w, b, True, sp, out=out[:rows])
    assert y.data_ptr() == out.data_ptr() and bool(torch.isnan(out[rows:]).all())
    plain = hip.gcn_forward(plan.fwd_ptr, plan.fwd_idx, plan.fwd_val, rows, x, plan.self_coef, w, b, True)
    assert_embeddings_close(y, plain, rtol=2e-6, what="staged vs plain kernel on an order-2 layer")


def test_staged_layer_rejects_what_it_cannot_address(hip):
    ptr, idx, val, _ = _csr(1, 128, 128, 2.0)
    x = torch.randn(128, 32, device=DEV)
    sp = hip.gcn_stage_plan(ptr, idx, 128)
    with pytest.raises(ValueError):
        hip.gcn_forward_staged(ptr, idx, val, 128, x, None, torch.randn(64, 32, device=DEV), None, True, sp)
    assert not hip.gcn_stage_wanted(10**7, 2 * 10**7, 2 * 10**7, 64, 64)                   # X of 4 GiB or more
    assert not hip.gcn_stage_wanted(10**6, 10**6, 2 * 10**7, 64, 64)                       # long rows
    assert not hip.gcn_stage_wanted(10**6, 10**6, 2 * 10**6, 64, 32)
