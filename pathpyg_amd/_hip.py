"""Device-tensor level wrappers over the C ABI (include/pathpyg_amd.h).

Every function here takes tensors that already live on an MI355X (``tensor.is_cuda``), allocates
outputs and workspaces through PyTorch's caching allocator, launches on the current stream and
returns new tensors on the same device.  PyTorch is only plumbing (memory, streams); all arithmetic
happens in the HIP kernels.  There is no CPU implementation behind these calls.
"""
from __future__ import annotations

import threading

import torch

from ._lib import check, lib

_DTYPE_CODE = {torch.int32: 0, torch.int64: 1, torch.float32: 2, torch.float64: 3}
_EDGE_AGGR = {"src": 0, "dst": 1, "max": 2, "mul": 3, "add": 4}
_REDUCE = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3}
DELTA_I64, DELTA_F32, DELTA_F64 = 0, 1, 2


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of the current device's current stream (the raw handle: ~0.4 us instead of the ~2.7 us of building a Stream object —
    a fifth of a shim call's host time, which is what small graphs and the per-rank step of a partitioned run are bound by)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _result(ws: torch.Tensor) -> tuple[int, int]:
    """{size, status} that *_count left at the start of its workspace (one 16-byte D2H copy)."""
    size, status = ws[:16].view(torch.int64).tolist()
    return size, status


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pathpyg_amd kernels need tensors on an MI355X device (got a CPU tensor); there is no CPU path")
        if dev is not None and t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
        dev = t.device
    return dev


def _edge_index(edge_index: torch.Tensor) -> torch.Tensor:
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    ei = torch.as_tensor(edge_index)
    if type(ei) is not torch.Tensor:          # tensor subclasses (e.g. an EdgeIndex) -> plain view
        ei = ei.as_subclass(torch.Tensor)
    if ei.dtype != torch.int64:
        ei = ei.to(torch.int64)
    return ei.contiguous()


def _bad_index(status: int, what: str, error=IndexError) -> None:
    if status & 1:
        raise error(f"{what}: node index out of range")


# ------------------------------------------------------------------ primitives
def minmax(a: torch.Tensor) -> tuple[int, int]:
    a = a.contiguous()
    dev = require_device(a)
    with torch.cuda.device(dev):
        out = torch.empty(2, dtype=torch.int64, device=dev)
        check(lib().pp_minmax_i64(_p(a), a.numel(), _p(out), _stream()), "pp_minmax_i64")
        lo, hi = out.tolist()
    return lo, hi


def degree(index: torch.Tensor, num_bins: int) -> torch.Tensor:
    index = index.contiguous()
    dev = require_device(index)
    with torch.cuda.device(dev):
        bins = torch.empty(num_bins, dtype=torch.int32, device=dev)
        check(lib().pp_degree_i64(_p(index), index.numel(), num_bins, _p(bins), _stream()), "pp_degree_i64")
    return bins


def exclusive_scan(values: torch.Tensor) -> torch.Tensor:
    """[0, v0, v0+v1, ..., total] as int64 (PyG ``cumsum``)."""
    values = values.contiguous()
    dev = require_device(values)
    n = values.numel()
    with torch.cuda.device(dev):
        out = torch.empty(n + 1, dtype=torch.int64, device=dev)
        ws = _workspace(lib().pp_scan_ws_bytes(n), dev)
        if values.dtype == torch.int32:
            fn, name = lib().pp_exclusive_scan_i32, "pp_exclusive_scan_i32"
        elif values.dtype == torch.int64:
            fn, name = lib().pp_exclusive_scan_i64, "pp_exclusive_scan_i64"
        else:
            raise TypeError("exclusive_scan: int32 or int64 input")
        check(fn(_p(values), n, _p(out), _p(ws), ws.numel(), _stream()), name)
    return out


def sort_pairs(keys: torch.Tensor, values: torch.Tensor | None, begin_bit: int, end_bit: int):
    """Stable radix sort of (key, value) pairs; keys int32/int64 holding NON-NEGATIVE values."""
    keys = keys.contiguous()
    dev = require_device(keys, values)
    n = keys.numel()
    with torch.cuda.device(dev):
        keys_out = torch.empty_like(keys)
        vals_out = torch.empty(n, dtype=torch.int32, device=dev)
        if keys.dtype == torch.int32:
            fn, name, kb = lib().pp_sort_pairs_u32, "pp_sort_pairs_u32", 4
        elif keys.dtype == torch.int64:
            fn, name, kb = lib().pp_sort_pairs_u64, "pp_sort_pairs_u64", 8
        else:
            raise TypeError("sort_pairs: int32 or int64 keys")
        if values is not None:
            values = values.contiguous()
            if values.dtype != torch.int32:
                raise TypeError("sort_pairs: int32 values")
        ws = _workspace(lib().pp_sort_ws_bytes(n, kb), dev)
        check(fn(_p(keys), _p(values), _p(keys_out), _p(vals_out), n, begin_bit, end_bit, _p(ws), ws.numel(), _stream()), name)
    return keys_out, vals_out


# ------------------------------------------------------------------ lifts
def resolve_delta(time_dtype: torch.dtype, delta) -> tuple[int, int, float]:
    """(kind, delta_i, delta_f) describing ``torch.tensor(delta)`` the way the reference's
    ``lift_order_temporal`` sees it (src/pathpyG/algorithms/temporal.py:30,43)."""
    d = delta.detach().cpu() if isinstance(delta, torch.Tensor) else torch.tensor(delta)
    if d.dim() != 0:
        raise ValueError("delta must be a scalar")
    if time_dtype == torch.float64:
        return DELTA_F64, 0, float(d)                  # int / float32 / float64 delta all widen to float64
    if d.dtype.is_floating_point:
        if d.dtype == torch.float64:
            return DELTA_F64, 0, float(d)
        return DELTA_F32, 0, float(d.to(torch.float32))  # python float -> float32 tensor in the reference
    return DELTA_I64, int(d), 0.0


def _drive(gen):
    """Runs a count -> read-back -> fill generator on its own: the generator yields its workspace after the count launch and is sent
    ``(size, status)``."""
    ws = next(gen)
    try:
        gen.send(_result(ws))
    except StopIteration as done:
        return done.value
    raise RuntimeError("count/fill generator yielded twice")


def run_together(*gens):
    """Launches the COUNT phases of several independent count -> fill operations back to back, reads all their {size, status} pairs with
    ONE device-to-host copy (instead of one synchronising read-back each), then runs the fills.  Returns the results in order."""
    try:
        workspaces = [next(g) for g in gens]
        if len({ws.device for ws in workspaces}) > 1:
            raise RuntimeError("run_together: the operations live on different devices")
        if len(workspaces) == 1:
            heads = [_result(workspaces[0])]
        else:
            flat = torch.cat([ws[:16] for ws in workspaces]).view(torch.int64).tolist()
            heads = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(workspaces))]
        results = []
        for g, head in zip(gens, heads):
            try:
                g.send(head)
            except StopIteration as done:
                results.append(done.value)
            else:
                raise RuntimeError("count/fill generator yielded twice")
        return results
    finally:
        for g in gens:          # an error in one operation must not leave the others suspended
            g.close()


def temporal_lift(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta, n_own: int | None = None,
                  id_offset: int = 0) -> torch.Tensor:
    """Second-order event graph of a TIME-SORTED event list: all (i, j) with head(i) == tail(j) and
    t_i < t_j <= t_i + delta, lexicographic, int64 [2, E2].  ``n_own`` / ``id_offset``: edge-range shard —
    only the first ``n_own`` events are sources and ``id_offset`` is added to every id of the result."""
    return _drive(temporal_lift_steps(edge_index, time, num_nodes, delta, n_own, id_offset))


def temporal_lift_steps(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta, n_own: int | None = None, id_offset: int = 0):
    """:func:`temporal_lift` as a count -> fill generator for :func:`run_together`."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, time)
    if time.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        time = time.to(torch.int64)
    if time.dtype not in (torch.int64, torch.float64):
        raise TypeError(f"timestamps must be int64 or float64, got {time.dtype}")
    time = time.contiguous()
    m = ei.size(1)
    if time.numel() != m:
        raise ValueError("time and edge_index disagree on the number of events")
    kind, di, df = resolve_delta(time.dtype, delta)
    L = lib()
    # (the device guard is NOT held across the yield: interleaved generators would exit their guards out of order)
    with torch.cuda.device(dev):
        ws = _workspace(L.pp_temporal_ws_bytes(m, num_nodes), dev)
        check(L.pp_temporal_count(_p(ei), _p(time), _DTYPE_CODE[time.dtype], m, m if n_own is None else int(n_own), num_nodes, kind, di, df,
                                  _p(ws), ws.numel(), _stream()), "pp_temporal_count")
    total, status = yield ws
    _bad_index(status, "lift_order_temporal")
    if status & 2:
        raise ValueError("lift_order_temporal: the events are not sorted by time (TemporalGraph sorts them on construction; "
                         "data.time / data.edge_index were modified afterwards)")
    with torch.cuda.device(dev):
        out = torch.empty((2, total), dtype=torch.int64, device=dev)
        check(L.pp_temporal_fill(m, num_nodes, total, int(id_offset), _p(out), _p(ws), ws.numel(), _stream()), "pp_temporal_fill")
    return out


def linegraph_lift(edge_index: torch.Tensor, num_nodes: int, edge_range: tuple[int, int] | None = None) -> torch.Tensor:
    """Line-graph lift of a source-sorted edge list; ``edge_range = (lo, hi)``: edge-range shard — only the edges ``lo .. hi-1`` are
    sources (the block of the global result that starts at the first pair of edge ``lo``), ids stay global."""
    ei = _edge_index(edge_index)
    dev = require_device(ei)
    e = ei.size(1)
    lo, hi = (0, e) if edge_range is None else edge_range
    L = lib()
    with torch.cuda.device(dev):
        ws = _workspace(L.pp_linegraph_ws_bytes(e, num_nodes), dev)
        check(L.pp_linegraph_count(_p(ei), e, lo, hi, num_nodes, _p(ws), ws.numel(), _stream()), "pp_linegraph_count")
        total, status = _result(ws)
        _bad_index(status, "lift_order_edge_index")
        out = torch.empty((2, total), dtype=torch.int64, device=dev)
        check(L.pp_linegraph_fill(e, num_nodes, total, _p(out), _p(ws), ws.numel(), _stream()), "pp_linegraph_fill")
    return out


def edge_attr(edge_index: torch.Tensor, attr: torch.Tensor, aggr: str) -> torch.Tensor:
    if aggr not in _EDGE_AGGR:
        raise ValueError(f"Unknown aggregation method {aggr}")
    ei = _edge_index(edge_index)
    dev = require_device(ei, attr)
    if attr.dtype not in _DTYPE_CODE:
        raise TypeError(f"aggregate_node_attributes: unsupported attribute dtype {attr.dtype}")
    attr = attr.contiguous()
    n = attr.size(0)
    width = attr.numel() // n if n else 1
    e = ei.size(1)
    with torch.cuda.device(dev):
        out = torch.empty((e,) + tuple(attr.shape[1:]), dtype=attr.dtype, device=dev)
        status = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib().pp_edge_attr(_p(ei), e, _p(attr), _DTYPE_CODE[attr.dtype], n, max(width, 1), _EDGE_AGGR[aggr], _p(out), _p(status),
                                 _stream()), "pp_edge_attr")
        _bad_index(int(status.item()), "aggregate_node_attributes")
    return out


def extend_node_sequence(edge_index: torch.Tensor, node_sequence: torch.Tensor) -> torch.Tensor:
    ei = _edge_index(edge_index)
    dev = require_device(ei, node_sequence)
    ns = node_sequence.to(torch.int64).contiguous()
    n_rows, k = ns.shape
    e = ei.size(1)
    with torch.cuda.device(dev):
        out = torch.empty((e, k + 1), dtype=torch.int64, device=dev)
        status = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib().pp_extend_node_sequence(_p(ei), e, _p(ns), n_rows, k, _p(out), _p(status), _stream()), "pp_extend_node_sequence")
        _bad_index(int(status.item()), "node sequence extension")
    return out


def gather_concat(rows: torch.Tensor, idx: torch.Tensor, suffix: torch.Tensor) -> torch.Tensor:
    """``cat([rows[idx], suffix[:, None]], 1)`` for int64 ``rows [R, k]`` -> ``[n, k+1]``."""
    dev = require_device(rows, idx, suffix)
    rows = rows.to(torch.int64).contiguous()
    idx = idx.to(torch.int64).contiguous()
    suffix = suffix.to(torch.int64).contiguous()
    n, k = idx.numel(), rows.size(1)
    with torch.cuda.device(dev):
        out = torch.empty((n, k + 1), dtype=torch.int64, device=dev)
        status = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib().pp_gather_concat(_p(rows), rows.size(0), k, _p(idx), _p(suffix), n, _p(out), _p(status), _stream()), "pp_gather_concat")
        _bad_index(int(status.item()), "node sequence gather")
    return out


# ------------------------------------------------------------------ aggregation
def unique_rows(rows: torch.Tensor, value_range: tuple[int, int] | None = None):
    """``torch.unique(rows, dim=0, return_inverse=True)`` for int64 ``[M, k]`` rows."""
    dev = require_device(rows)
    rows = rows.to(torch.int64).contiguous()
    m, k = rows.shape
    L = lib()
    with torch.cuda.device(dev):
        inverse = torch.empty(m, dtype=torch.int64, device=dev)
        if m == 0:
            return rows.new_empty((0, k)), inverse
        lo, hi = value_range if value_range is not None else minmax(rows)
        ws = _workspace(L.pp_unique_rows_ws_bytes(m), dev)
        check(L.pp_unique_rows_count(_p(rows), m, k, lo, hi, _p(inverse), _p(ws), ws.numel(), _stream()), "pp_unique_rows_count")
        n_unique, _ = _result(ws)
        uniq = torch.empty((n_unique, k), dtype=torch.int64, device=dev)
        check(L.pp_unique_rows_fill(_p(rows), m, k, n_unique, _p(uniq), _p(ws), ws.numel(), _stream()), "pp_unique_rows_fill")
    return uniq, inverse


UNIT = "unit"           # `weight=UNIT`: every instance edge weighs 1.0 (the reference's default torch.ones) without materialising the vector


def coalesce(edge_index: torch.Tensor, weight, num_nodes: int, reduce: str = "sum",
             remap: torch.Tensor | None = None, want_inverse: bool = False, col_block: tuple | None = None):
    """PyG ``coalesce`` of ``remap[edge_index]`` (or ``edge_index``): (row, col)-sorted distinct edges + reduced weights.
    ``weight``: a vector, ``None`` (no weights) or :data:`UNIT` (float32 ones: the merged weight is the run length, no gather).
    ``want_inverse`` additionally returns, per input edge, the index of the merged edge it went into.
    ``col_block = (col_base [num_nodes] int64, col_bits)``: every column of row r lies in ``[col_base[r], col_base[r] + 2**col_bits)``
    (De Bruijn layers): shorter sort keys, same result."""
    return _drive(coalesce_steps(edge_index, weight, num_nodes, reduce, remap, want_inverse, col_block))


def coalesce_steps(edge_index: torch.Tensor, weight, num_nodes: int, reduce: str = "sum",
                   remap: torch.Tensor | None = None, want_inverse: bool = False, col_block: tuple | None = None):
    """:func:`coalesce` as a count -> fill generator for :func:`run_together`."""
    if reduce not in _REDUCE:
        raise ValueError(f"unknown reduce {reduce}")
    ei = _edge_index(edge_index)
    unit = isinstance(weight, str)
    if unit:
        if weight != UNIT:
            raise ValueError(f"edge weights must be a tensor, None or the marker {UNIT!r}, got {weight!r}")
        weight = None
    dev = require_device(ei, weight, remap)
    e = ei.size(1)
    if weight is not None:
        if weight.dtype not in _DTYPE_CODE:
            raise TypeError(f"edge weights of dtype {weight.dtype} are not supported")
        if weight.dim() != 1 or weight.numel() != e:
            raise ValueError("edge_weight must be a vector with one entry per edge")
        weight = weight.contiguous()
    if remap is not None:
        remap = remap.to(torch.int64).contiguous()
    col_base, col_bits = (None, 0) if col_block is None else (col_block[0].to(torch.int64).contiguous(), int(col_block[1]))
    if col_base is not None and col_base.numel() != num_nodes:
        raise ValueError("col_base must hold one entry per node")
    L = lib()
    with torch.cuda.device(dev):
        ws = _workspace(L.pp_coalesce_ws_bytes(e), dev)
        check(L.pp_coalesce_count(_p(ei), e, _p(remap), 0 if remap is None else remap.numel(), num_nodes, _p(col_base), col_bits,
                                  _p(ws), ws.numel(), _stream()), "pp_coalesce_count")
    n_out, status = yield ws
    # the reference fails in EdgeIndex.validate() with a ValueError when an index exceeds the number of distinct nodes
    # (lift_order.py:133-147: layer-1 quirk, node ids are used as given while num_nodes counts the distinct ones)
    _bad_index(status, "aggregate_edge_index (an edge refers to a node id >= number of distinct nodes)", ValueError)
    with torch.cuda.device(dev):
        out_index = torch.empty((2, n_out), dtype=torch.int64, device=dev)
        out_weight = (torch.empty(n_out, dtype=torch.float32, device=dev) if unit else None) if weight is None else \
            torch.empty(n_out, dtype=weight.dtype, device=dev)
        check(L.pp_coalesce_fill(_p(weight), 2 if weight is None else _DTYPE_CODE[weight.dtype], _REDUCE[reduce], e, n_out, num_nodes,
                                 _p(col_base), col_bits, _p(out_index), _p(out_weight), _p(ws), ws.numel(), _stream()), "pp_coalesce_fill")
        if want_inverse:
            inverse = torch.empty(e, dtype=torch.int64, device=dev)
            check(L.pp_coalesce_inverse(e, _p(inverse), _p(ws), ws.numel(), _stream()), "pp_coalesce_inverse")
            return out_index, out_weight, inverse
    return out_index, out_weight


# ------------------------------------------------------------------ Graph bookkeeping
def _ordered_dtype(a: torch.Tensor) -> torch.Tensor:
    if a.dtype in (torch.int64, torch.float64):
        return a.contiguous()
    if a.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool):
        return a.to(torch.int64).contiguous()
    if a.dtype in (torch.float32, torch.float16, torch.bfloat16):
        return a.to(torch.float64).contiguous()      # exact widening keeps the order
    raise TypeError(f"cannot order values of dtype {a.dtype}")


def is_sorted(a: torch.Tensor) -> bool:
    a = _ordered_dtype(a)
    dev = require_device(a)
    with torch.cuda.device(dev):
        out = torch.empty(1, dtype=torch.int64, device=dev)
        if a.dtype == torch.float64:
            check(lib().pp_count_descents_f64(_p(a), a.numel(), _p(out), _stream()), "pp_count_descents_f64")
        else:
            check(lib().pp_count_descents_i64(_p(a), a.numel(), _p(out), _stream()), "pp_count_descents_i64")
        return int(out.item()) == 0


def time_stats(time: torch.Tensor) -> tuple[int, int, int]:
    """``(descents, min, max)`` of a timestamp vector with ONE read-back (float64: min = max = 0)."""
    time = _ordered_dtype(time)
    dev = require_device(time)
    with torch.cuda.device(dev):
        out = torch.empty(3, dtype=torch.int64, device=dev)
        check(lib().pp_time_stats(_p(time), _DTYPE_CODE[time.dtype], time.numel(), _p(out), _stream()), "pp_time_stats")
        descents, lo, hi = out.tolist()
    return descents, lo, hi


def gather_events(edge_index: torch.Tensor, time: torch.Tensor, perm: torch.Tensor):
    """``(edge_index[:, perm], time[perm])`` in one pass (time int64 or float64)."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, time, perm)
    if time.dtype not in (torch.int64, torch.float64):
        raise TypeError("gather_events: timestamps must be int64 or float64")
    time, perm = time.contiguous(), perm.to(torch.int64).contiguous()
    m = ei.size(1)
    with torch.cuda.device(dev):
        out_ei, out_t = torch.empty_like(ei), torch.empty_like(time)
        status = torch.empty(1, dtype=torch.int64, device=dev)
        check(lib().pp_gather_events(_p(ei), _p(time), _p(perm), m, _p(out_ei), _p(out_t), _p(status), _stream()), "pp_gather_events")
    return out_ei, out_t


def argsort(keys: torch.Tensor, value_range: tuple[int, int] | None = None) -> torch.Tensor:
    """Stable argsort of an integer or floating-point vector -> int64 permutation."""
    keys = _ordered_dtype(keys)
    dev = require_device(keys)
    n = keys.numel()
    with torch.cuda.device(dev):
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        if n == 0:
            return perm
        if keys.dtype == torch.float64:
            ws = _workspace(lib().pp_argsort_ws_bytes(n), dev)
            check(lib().pp_argsort_f64(_p(keys), n, _p(perm), _p(ws), ws.numel(), _stream()), "pp_argsort_f64")
            return perm
        lo, hi = value_range if value_range is not None else minmax(keys)
        ws = _workspace(lib().pp_argsort_ws_bytes(n), dev)
        check(lib().pp_argsort_i64(_p(keys), n, lo, hi, _p(perm), _p(ws), ws.numel(), _stream()), "pp_argsort_i64")
    return perm


def ptr_from_sorted(sorted_index: torch.Tensor, num_rows: int) -> torch.Tensor:
    sorted_index = sorted_index.to(torch.int64).contiguous()
    dev = require_device(sorted_index)
    with torch.cuda.device(dev):
        ptr = torch.empty(num_rows + 1, dtype=torch.int64, device=dev)
        check(lib().pp_ptr_from_sorted_i64(_p(sorted_index), sorted_index.numel(), num_rows, _p(ptr), _stream()), "pp_ptr_from_sorted_i64")
    return ptr


def temporal_bfs(edge_index: torch.Tensor, num_nodes: int, event_graph: torch.Tensor):
    """All-pairs shortest time-respecting paths over a lifted event graph (``event_graph`` = ``temporal_lift`` output of the same
    time-sorted ``edge_index``): ``(dist int32 [n,n] with -1 = unreachable, pred int64 [n,n])``."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, event_graph)
    m, n = ei.size(1), int(num_nodes)
    succ_ptr = ptr_from_sorted(event_graph[0], m)
    succ = event_graph[1].contiguous()
    by_src = argsort(ei[0], (0, max(n - 1, 0)))
    by_src_ptr = ptr_from_sorted(ei[0][by_src], n)
    L = lib()
    with torch.cuda.device(dev):
        dist = torch.empty((n, n), dtype=torch.int32, device=dev)
        pred = torch.empty((n, n), dtype=torch.int64, device=dev)
        ws = _workspace(L.pp_temporal_bfs_ws_bytes(m, n), dev)
        check(L.pp_temporal_bfs(_p(ei), m, n, _p(succ_ptr), _p(succ), _p(by_src_ptr), _p(by_src), _p(dist), _p(pred), _p(ws), ws.numel(),
                                _stream()), "pp_temporal_bfs")
    return dist, pred


def temporal_betweenness(edge_index: torch.Tensor, num_nodes: int, event_graph: torch.Tensor) -> torch.Tensor:
    """Temporal betweenness per node index (float64 [n]) over a lifted event graph of the time-sorted ``edge_index``."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, event_graph)
    m, n = ei.size(1), int(num_nodes)
    succ_ptr = ptr_from_sorted(event_graph[0], m)
    succ = event_graph[1].contiguous()
    by_src = argsort(ei[0], (0, max(n - 1, 0)))
    by_src_ptr = ptr_from_sorted(ei[0][by_src], n)
    by_dst = argsort(ei[1], (0, max(n - 1, 0)))
    by_dst_ptr = ptr_from_sorted(ei[1][by_dst], n)
    L = lib()
    with torch.cuda.device(dev):
        parts = int(L.pp_temporal_betweenness_parts(m, n))
        partial = torch.zeros((parts, n), dtype=torch.float64, device=dev)
        ws = _workspace(L.pp_temporal_betweenness_ws_bytes(m, n), dev)
        check(L.pp_temporal_betweenness(_p(ei), m, n, _p(succ_ptr), _p(succ), _p(by_src_ptr), _p(by_src), _p(by_dst_ptr), _p(by_dst),
                                        _p(partial), _p(ws), ws.numel(), _stream()), "pp_temporal_betweenness")
        return partial.sum(dim=0)


# ------------------------------------------------------------------ DBGNN message passing
class CsrPlan:
    """CSR pair of one propagation: ``fwd`` rows are the destinations, ``bwd`` rows the sources (transposed)."""

    __slots__ = ("n_dst", "n_src", "fwd_ptr", "fwd_idx", "fwd_val", "bwd_ptr", "bwd_idx", "bwd_val", "self_coef", "fwd_heavy", "bwd_heavy",
                 "dst_order", "edge_ordered", "aug")          # edge_ordered: bwd_idx[e] is the destination of edge e (plan of a row-sorted edge list)

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


HEAVY_ROW_ENTRIES = 512      # CSR rows longer than this (hubs of scale-free graphs) are summed by the chunked pre-pass


class HeavyRows:
    """Hub rows of one CSR direction: ``slot`` [n_rows] int32 (-1 = ordinary row) and the chunk table of ``pp_spmm_heavy_f32``."""

    __slots__ = ("slot", "chunk_begin", "chunk_end", "chunk_ptr", "n_heavy", "n_chunks")

    def __init__(self, ptr: torch.Tensor, n_rows: int, threshold: int = HEAVY_ROW_ENTRIES):
        dev = ptr.device
        length = ptr[1:] - ptr[:-1]
        rows = torch.nonzero(length > threshold).flatten()
        self.n_heavy = int(rows.numel())
        chunk = int(lib().pp_heavy_chunk_entries())
        per_row = (length[rows].to(torch.int64) + chunk - 1) // chunk
        chunk_ptr = torch.zeros(self.n_heavy + 1, dtype=torch.int64, device=dev)
        chunk_ptr[1:] = torch.cumsum(per_row, 0)
        self.n_chunks = int(chunk_ptr[-1])
        owner = torch.repeat_interleave(torch.arange(self.n_heavy, device=dev), per_row)
        within = torch.arange(self.n_chunks, device=dev) - chunk_ptr[owner]
        begin = ptr[rows].to(torch.int64)[owner] + within * chunk
        end = torch.minimum(begin + chunk, ptr[rows + 1].to(torch.int64)[owner])
        self.chunk_begin, self.chunk_end = begin.to(torch.int32), end.to(torch.int32)
        self.chunk_ptr = chunk_ptr.to(torch.int32)
        self.slot = torch.full((n_rows,), -1, dtype=torch.int32, device=dev)
        self.slot[rows] = torch.arange(self.n_heavy, dtype=torch.int32, device=dev)

    def aggregate(self, idx: torch.Tensor, val: torch.Tensor | None, x: torch.Tensor) -> torch.Tensor:
        """``[n_heavy, F]`` neighbour sums of the hub rows over ``x`` (chunked, fixed summation order)."""
        f = x.size(1)
        L = lib()
        with torch.cuda.device(x.device):
            out = torch.empty((self.n_heavy, f), dtype=torch.float32, device=x.device)
            ws = _workspace(L.pp_spmm_heavy_ws_bytes(self.n_chunks, f), x.device)
            check(L.pp_spmm_heavy_f32(_p(idx), _p(val), _p(x), f, self.n_chunks, _p(self.chunk_begin), _p(self.chunk_end), self.n_heavy,
                                      _p(self.chunk_ptr), _p(out), _p(ws), ws.numel(), _stream()), "pp_spmm_heavy_f32")
        return out


def _heavy_args(heavy: "HeavyRows | None", idx, val, x):
    """(slot, sums) to hand to a row kernel; ``(None, None)`` for plans without hub rows or feature widths the pre-pass cannot take."""
    if heavy is None or x.size(1) % 4 != 0 or x.size(1) > 256:
        return None, None
    return heavy.slot, heavy.aggregate(idx, val, x.contiguous())


def _plan_report(ws: torch.Tensor) -> torch.Tensor:
    """Device int64 [3] = {status bits, longest forward row, longest backward row} of the plan a builder just queued on workspace ``ws``
    (pp_plan_result_ptr(ws)[1..3]: the kernels of the plan report the row lengths themselves, no extra launches)."""
    return ws[8:32].view(torch.int64).clone()


def _finish_plans(entries: list, what: str) -> list:
    """ONE device-to-host copy for everything the host must know about freshly built plans: the bad-index status and the longest
    rows (hub rows get their chunk tables here — the rare path, a few torch ops).  Returns ``[(status, longest forward row, longest
    backward row)]`` per plan."""
    if not entries:
        return []
    host = (entries[0][0] if len(entries) == 1 else torch.cat([report for report, _ in entries])).tolist()
    if any(int(host[3 * k]) & 1 for k in range(len(entries))):
        raise IndexError(f"{what}: node index out of range")
    for k, (_, plan) in enumerate(entries):
        if host[3 * k + 1] > HEAVY_ROW_ENTRIES:
            plan.fwd_heavy = HeavyRows(plan.fwd_ptr, plan.n_dst)
        if host[3 * k + 2] > HEAVY_ROW_ENTRIES:
            plan.bwd_heavy = HeavyRows(plan.bwd_ptr, plan.n_src)
    return [tuple(int(v) for v in host[3 * k: 3 * k + 3]) for k in range(len(entries))]


def gcn_plan(edge_index: torch.Tensor, edge_weight: torch.Tensor | None, num_nodes: int, row_sorted: bool | None = None,
             status_out: list | None = None, want_dst_order: bool = False) -> CsrPlan:
    """GCN normalisation of a weighted graph, once per graph (see pp_gcn_plan in the C header).
    ``row_sorted=None`` checks on the device whether the sources are non-decreasing (one tiny kernel + 8-byte read).
    ``status_out``: append the device status word instead of reading it now (the caller checks several plans with ONE
    device-to-host copy, see :func:`check_plan_status`)."""
    ei = _edge_index(edge_index)
    if row_sorted is None:
        row_sorted = ei.size(1) < 2 or is_sorted(ei[0])
    dev = require_device(ei, edge_weight)
    e = ei.size(1)
    if edge_weight is not None:
        edge_weight = edge_weight.to(torch.float32).contiguous()
        if edge_weight.numel() != e:
            raise ValueError("edge_weights must hold one value per edge")
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        plan = CsrPlan(n_dst=num_nodes, n_src=num_nodes,
                       fwd_ptr=torch.empty(num_nodes + 1, **i32), fwd_idx=torch.empty(e, **i32), fwd_val=torch.empty(e, **f32),
                       bwd_ptr=torch.empty(num_nodes + 1, **i32), bwd_idx=torch.empty(e, **i32), bwd_val=torch.empty(e, **f32),
                       self_coef=torch.empty(num_nodes, **f32))
        ws = _workspace(L.pp_gcn_plan_ws_bytes(e, num_nodes), dev)
        if want_dst_order:
            plan.dst_order = torch.empty(e, **i32)
        check(L.pp_gcn_plan(_p(ei), _p(edge_weight), e, num_nodes, 1 if row_sorted else 0, _p(plan.fwd_ptr), _p(plan.fwd_idx), _p(plan.fwd_val), _p(plan.bwd_ptr),
                            _p(plan.bwd_idx), _p(plan.bwd_val), _p(plan.self_coef), _p(plan.dst_order), _p(ws), ws.numel(), _stream()), "pp_gcn_plan")
        plan.edge_ordered = bool(row_sorted)
        entry = (_plan_report(ws), plan)
        if status_out is None:
            _finish_plans([entry], "GCNConv")
        else:
            status_out.append(entry)
    return plan


def gcn_plan_partition(edge_index_local: torch.Tensor, edge_weight: torch.Tensor | None, n_src: int, n_dst: int, halo_dinv,
                       row_sorted: bool = False, status_out: list | None = None, want_dst_order: bool = False) -> CsrPlan:
    """GCN plan of a DESTINATION-ROW PARTITION (multi-GPU DBGNN): ``edge_index_local`` [2, E] holds local ids — row 0 (sources) below
    ``n_src`` = owned + halo rows, row 1 (destinations) below ``n_dst`` = owned rows, owned node i is source i and destination i.
    ``halo_dinv(dinv_own [n_dst]) -> dinv_halo [n_src - n_dst]`` is called between the two phases of the plan (pp_gcn_plan_begin /
    pp_gcn_plan_finish) and returns the d^-1/2 of the halo rows from their owners (one 4-byte-per-row halo exchange)."""
    ei = _edge_index(edge_index_local)
    dev = require_device(ei, edge_weight)
    e = ei.size(1)
    if edge_weight is not None:
        edge_weight = edge_weight.to(torch.float32).contiguous()
        if edge_weight.numel() != e:
            raise ValueError("edge_weights must hold one value per edge")
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        plan = CsrPlan(n_dst=n_dst, n_src=n_src,
                       fwd_ptr=torch.empty(n_dst + 1, **i32), fwd_idx=torch.empty(e, **i32), fwd_val=torch.empty(e, **f32),
                       bwd_ptr=torch.empty(n_src + 1, **i32), bwd_idx=torch.empty(e, **i32), bwd_val=torch.empty(e, **f32),
                       self_coef=torch.empty(n_dst, **f32))
        dinv = torch.empty(n_src, **f32)
        ws = _workspace(L.pp_gcn_plan_ws_bytes(e, n_src), dev)
        if want_dst_order:
            plan.dst_order = torch.empty(e, **i32)
        rs = 1 if row_sorted else 0
        check(L.pp_gcn_plan_begin(_p(ei), _p(edge_weight), e, n_src, n_dst, rs, _p(plan.fwd_ptr), _p(plan.fwd_idx), _p(plan.fwd_val),
                                  _p(plan.bwd_ptr), _p(plan.self_coef), _p(dinv), _p(plan.dst_order), _p(ws), ws.numel(), _stream()),
              "pp_gcn_plan_begin")
        if halo_dinv is not None:            # a collective: called even when this rank has no halo rows
            dinv[n_dst:] = halo_dinv(dinv[:n_dst])
        check(L.pp_gcn_plan_finish(_p(ei), _p(edge_weight), e, n_src, n_dst, rs, _p(dinv), _p(plan.fwd_idx), _p(plan.fwd_val), _p(plan.bwd_ptr),
                                   _p(plan.bwd_idx), _p(plan.bwd_val), _p(ws), ws.numel(), _stream()), "pp_gcn_plan_finish")
        entry = (_plan_report(ws), plan)
        if status_out is None:
            _finish_plans([entry], "GCNConv (partition)")
        else:
            status_out.append(entry)
    return plan


_IDENTITY_PTR: dict = {}


def _identity_ptr(n: int, dev) -> torch.Tensor:
    """``arange(n + 1)`` as int32 row pointers (one entry per row), kept for the last size asked for on a device: the order-2 node count of
    consecutive batches of one stream rarely changes, and the array is read-only."""
    hit = _IDENTITY_PTR.get(dev)
    if hit is None or hit.numel() != n + 1:
        hit = torch.arange(n + 1, dtype=torch.int32, device=dev)
        _IDENTITY_PTR[dev] = hit
    return hit


def bipartite_plan_from_edge_grouping(plan_fo: CsrPlan, edge_dst: torch.Tensor, n_ho: int) -> CsrPlan:
    """Bipartite "last" plan of an order-2 De Bruijn model WITHOUT another sort: the higher-order nodes are the first-order graph's
    edges (same order), so "the higher-order nodes that end in node v" = "the edges into v" = the destination grouping the
    first-order GCN plan already holds (``gcn_plan(..., want_dst_order=True)``).  ``edge_dst`` = ``edge_index[1]`` of that graph."""
    dev = plan_fo.fwd_ptr.device
    ptr = plan_fo.fwd_ptr
    # (the plan of a row-sorted edge list already holds the destinations in edge order as its source-major index: no int32 copy)
    bwd_idx = plan_fo.bwd_idx if plan_fo.edge_ordered else edge_dst.to(torch.int32).contiguous()
    plan = CsrPlan(n_dst=plan_fo.n_dst, n_src=n_ho, fwd_ptr=ptr, fwd_idx=plan_fo.dst_order, fwd_val=None,
                   bwd_ptr=_identity_ptr(n_ho, dev), bwd_idx=bwd_idx, bwd_val=None,
                   self_coef=(ptr[1:] - ptr[:-1]).to(torch.float32))
    plan.fwd_heavy = plan_fo.fwd_heavy              # same row pointers: the same hub rows
    return plan


def check_plan_status(entries: list, what: str = "DBGNN") -> list:
    """One device-to-host copy for the status words (and longest rows) collected by several plan builders; returns
    ``[(status, longest forward row, longest backward row)]``."""
    return _finish_plans(entries, what)


def bipartite_plan(bipartite_index: torch.Tensor, n_ho: int, n_fo: int, pair_value: torch.Tensor | None = None,
                   src_sorted: bool | None = None, status_out: list | None = None) -> CsrPlan:
    """CSR pair of a [2, n_pairs] (source, destination) index between two node sets (``n_ho`` sources, ``n_fo``
    destinations); ``self_coef`` = in-degree of every destination.  ``pair_value``: optional coefficient per pair."""
    bi = _edge_index(bipartite_index)
    dev = require_device(bi, pair_value)
    nb = bi.size(1)
    if pair_value is not None:
        pair_value = pair_value.to(torch.float32).contiguous()
    if src_sorted is None:
        src_sorted = nb < 2 or is_sorted(bi[0])
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        plan = CsrPlan(n_dst=n_fo, n_src=n_ho,
                       fwd_ptr=torch.empty(n_fo + 1, **i32), fwd_idx=torch.empty(nb, **i32),
                       bwd_ptr=torch.empty(n_ho + 1, **i32), bwd_idx=torch.empty(nb, **i32),
                       self_coef=torch.empty(n_fo, **f32))
        if pair_value is not None:
            plan.fwd_val, plan.bwd_val = torch.empty(nb, **f32), torch.empty(nb, **f32)
        ws = _workspace(L.pp_gcn_plan_ws_bytes(nb, max(n_ho, n_fo)), dev)
        check(L.pp_bipartite_plan(_p(bi), nb, n_ho, n_fo, 1 if src_sorted else 0, _p(pair_value), _p(plan.fwd_ptr), _p(plan.fwd_idx), _p(plan.fwd_val),
                                  _p(plan.self_coef), _p(plan.bwd_ptr), _p(plan.bwd_idx), _p(plan.bwd_val), _p(ws), ws.numel(), _stream()),
              "pp_bipartite_plan")
        entry = (_plan_report(ws), plan)
        if status_out is None:
            _finish_plans([entry], "BipartiteGraphOperator")
        else:
            status_out.append(entry)
    return plan


_INT32_ROWS = 0x7ffffff0           # sizes the int32 CSR arrays of the order-2 builder can hold


class DeBruijn2:
    """Result of :func:`debruijn2`: the GCN plans of the first-order and the order-2 graph of a temporal stream, the bipartite "last" plan
    and the layer sizes (``m`` events, ``E2`` lifted instance pairs, ``U2`` order-2 nodes = first-order edges, ``A2`` order-2 edges)."""

    __slots__ = ("fo", "ho", "bip", "sizes", "fo_weight", "fo_dst", "ho_fwd_weight")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


_HUB_STATS = 8 + 2 * (64 + 1)          # index of the hub block inside the builder's int64 result header (csrc/pp_debruijn.hip: kDb2HubStats)
_HUB_PART_BYTES_MAX = 24 << 30         # per-task partial results of the hub nodes (768 bytes per part column): beyond this the generic kernels
_tls = threading.local()


_A2_GUESS: dict = {}                    # (m, n, want_weights, delta) -> A2 of the last build of a stream of that shape at that delta (debruijn2 allocates ahead of its read-back;
#                                         delta is part of the key: a sweep from a large to a small delta would otherwise keep plans that are views of buffers many times too long)


def _a2_buffers(a2: int, want_weights: bool, i32: dict, f32: dict) -> tuple:
    """The A2-sized outputs of :func:`debruijn2`: forward / backward index + value arrays, the 2 * A2 pack scratch, the raw weights."""
    return (torch.empty(a2, **i32), torch.empty(a2, **f32), torch.empty(a2, **i32), torch.empty(a2, **f32), torch.empty(2 * a2, **i32),
            torch.empty(a2, **f32) if want_weights else None)


def _pinned_stats() -> torch.Tensor:
    """Pinned int64 [16 + 154] of the calling thread: where pp_debruijn2_lists copies the hub statistics ([:16]) and pp_debruijn2_count the
    result header ([16:]), both asynchronously."""
    buf = getattr(_tls, "stats", None)
    if buf is None:
        buf = _tls.stats = torch.empty(16 + _HUB_STATS + 16, dtype=torch.int64).pin_memory()
    return buf


DENSE_EVENTS_PER_NODE = 2048


def debruijn2_wanted(m: int, num_nodes: int) -> bool:
    """Whether the node-by-node order-2 builder is the one to use for a stream of ``m`` events over ``num_nodes`` nodes.  It works node by node —
    in-events x out-events of the middle node — and is built for nodes with tens of events per side (hubs are chunked).  On a contact-shaped
    stream (tens of nodes, 10^4 events per node and side: 96 nodes / 2 * 10^6 events) every node is a hub and the per-node scans cost 1.9 ms where
    lift -> coalesce -> coalesce -> plans takes 1.0 ms (``hub_streams`` in the bench line): from ``DENSE_EVENTS_PER_NODE`` events per node on average
    the callers that choose (``MultiOrderModel.from_temporal_graph``, ``distributed.build_dbgnn_shard``) take the generic kernels."""
    return m <= DENSE_EVENTS_PER_NODE * max(int(num_nodes), 1)


def debruijn2(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta, weight: torch.Tensor | None = None,
              want_weights: bool = False, unsorted_ok: bool = False):
    """Order-2 De Bruijn model of a TIME-SORTED event stream, fused (pp_debruijn2_lists / _count / _fill, csrc/pp_debruijn.hip): what
    ``coalesce`` (layer 1) + ``temporal_lift`` + ``coalesce`` (layer 2) + ``gcn_plan`` x 2 + ``bipartite_plan_from_edge_grouping`` build,
    identical array by array, without the event graph.  Two read-backs: the hub statistics (behind the sorts; the wait runs under the
    out-side kernel) and the layer sizes.  Nodes with more than 64 in- / out-events (hubs) are worked on chunk-wise by the same builder;
    rows of more than 512 entries get the chunk tables of the DBGNN kernels (``HeavyRows``).  ``weight``: None (unit weights) or float32 [m].
    ``want_weights``: also keep the merged order-2 weights themselves (``ho_fwd_weight`` [A2], destination-major order, beside the normalised
    coefficients of the plan) — what ``MultiOrderModel.layers[2].data.edge_weight`` is derived from when it is read.
    Returns a :class:`DeBruijn2`, or ``None`` when the builder does not apply (an empty stream, a weight vector that is not float32, more than
    2^31 order-2 edges, hub partials beyond 24 GiB; with ``unsorted_ok`` an unsorted stream): the caller then takes the generic path."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, time, weight)
    if time.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        time = time.to(torch.int64)
    if time.dtype not in (torch.int64, torch.float64):
        raise TypeError(f"timestamps must be int64 or float64, got {time.dtype}")
    time = time.contiguous()
    m, n = ei.size(1), int(num_nodes)
    if time.numel() != m:
        raise ValueError("time and edge_index disagree on the number of events")
    if m == 0 or n == 0:
        return None
    if weight is not None:
        if weight.dtype != torch.float32 or weight.numel() != m:
            return None
        weight = weight.contiguous()
    kind, di, df = resolve_delta(time.dtype, delta)
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        ws = _workspace(L.pp_debruijn2_ws_bytes(m, n), dev)
        tcode = _DTYPE_CODE[time.dtype]
        stats = _pinned_stats()
        check(L.pp_debruijn2_lists(_p(ei), _p(time), tcode, m, n, _p(weight), _p(ws), ws.numel(), stats.data_ptr(), _stream()), "pp_debruijn2_lists")
        fo_bwd_ptr, fo_fwd_ptr = torch.empty(n + 1, **i32), torch.empty(n + 1, **i32)       # (allocated while the GPU sorts)
        fo_bwd_idx, fo_w = torch.empty(m, **i32), torch.empty(m, **f32)
        ho_fwd_ptr, ho_bwd_ptr = torch.empty(m + 1, **i32), torch.empty(m + 1, **i32)
        ho_deg, fo_deg = torch.empty(m, **f32), torch.empty(n, **f32)
        check(L.pp_debruijn2_wait(), "pp_debruijn2_wait")                                     # read-back 1 of 2: is there a hub node, how large
        hubs, packed, tasks, parts = (int(v) for v in stats[:4].tolist())
        out_hubs, out_events = packed >> 32, packed & 0xFFFFFFFF
        hub_ws = None
        if hubs:
            if parts * 768 > _HUB_PART_BYTES_MAX:
                return None
            hub_ws = _workspace(L.pp_debruijn2_hub_ws_bytes(out_events, out_hubs, parts), dev)
        hub_args = (hubs, out_hubs, out_events, tasks, parts, _p(hub_ws), hub_ws.numel() if hub_ws is not None else 0)
        check(L.pp_debruijn2_count(tcode, m, n, kind, di, df, _p(weight), _p(fo_bwd_ptr), _p(fo_bwd_idx), _p(fo_w), _p(fo_fwd_ptr), _p(ho_fwd_ptr),
                                   _p(ho_bwd_ptr), _p(ho_deg), _p(fo_deg), _p(ws), ws.numel(), *hub_args, stats.data_ptr() + 128, _stream()),
              "pp_debruijn2_count")
        # The outputs are allocated WHILE the count pass runs, not behind the read-back of its sizes (the GPU idles through everything the host
        # does between that read-back and the fill call): U2 = A1 <= m is a bound, A2 is guessed from the last build of a stream of this shape
        # (an epoch loop, a rolling window) and re-allocated when the guess is short or more than a quarter too long.
        ho_self, fo_fwd_idx, fo_fwd_val = torch.empty(m, **f32), torch.empty(m, **i32), torch.empty(m, **f32)
        fo_dst_order, fo_bwd_val, fo_self = torch.empty(m, **i32), torch.empty(m, **f32), torch.empty(n, **f32)
        guess_key = (m, n, want_weights, kind, di, df)
        guess = _A2_GUESS.get(guess_key, 0)
        a2_bufs = _a2_buffers(guess, want_weights, i32, f32) if guess else None
        launched = False
        if a2_bufs is not None:
            # .. and with a guess the fill is launched by the SAME C call that waits for the sizes (when the stream is good and A2 fits): no host
            # code between read-back and fill
            flag = stats[15:16]
            check(L.pp_debruijn2_fill_ready(tcode, m, n, kind, di, df, _p(weight), _p(fo_bwd_ptr), _p(fo_bwd_idx), _p(fo_w), _p(fo_fwd_ptr), _p(ho_fwd_ptr),
                                            _p(ho_bwd_ptr), _p(ho_deg), _p(fo_deg), guess, _p(a2_bufs[0]), _p(a2_bufs[1]), _p(a2_bufs[2]), _p(a2_bufs[3]),
                                            _p(ho_self), _p(fo_fwd_idx), _p(fo_fwd_val), _p(fo_dst_order), _p(fo_bwd_val), _p(fo_self), _p(a2_bufs[5]),
                                            _p(a2_bufs[4]), _p(ws), ws.numel(), *hub_args, 1, stats.data_ptr() + 128, flag.data_ptr(), _stream()),
                  "pp_debruijn2_fill_ready")
            launched = bool(int(flag[0]))
        else:
            check(L.pp_debruijn2_wait(), "pp_debruijn2_wait")                                 # read-back 2 of 2: the layer sizes (+ the hubs' share)
        head = stats[16:].tolist()
        u2, status, a2, e2, a1 = head[:5]
        e2 += head[_HUB_STATS + 4]
        longest = head[_HUB_STATS + 5: _HUB_STATS + 9] if hubs else (0, 0, 0, 0)
        _bad_index(status, "MultiOrderModel.from_temporal_graph")
        if status & 2:
            if unsorted_ok:           # (the caller sorts and takes the generic path: MultiOrderModel.from_temporal_graph, multi_order_model.py:148-151)
                return None
            raise ValueError("lift_order_temporal: the events are not sorted by time (TemporalGraph sorts them on construction; "
                             "data.time / data.edge_index were modified afterwards)")
        if status & 4:
            return None
        if a2 >= _INT32_ROWS:           # the builder's row pointers are int32 (the scan's TOTAL is int64, so this is the true A2): generic kernels
            return None
        if not launched and (a2_bufs is None or not (a2 <= guess <= a2 + a2 // 4 + 1024)):
            a2_bufs = _a2_buffers(a2, want_weights, i32, f32)
        if len(_A2_GUESS) >= 64 and guess_key not in _A2_GUESS:      # (a handful of stream shapes at a time: drop the oldest)
            _A2_GUESS.pop(next(iter(_A2_GUESS)))
        _A2_GUESS[guess_key] = a2
        ho_fwd_idx, ho_fwd_val, ho_bwd_idx, ho_bwd_val, pack, ho_fwd_w = (None if b is None else b[:k * a2] for b, k in zip(a2_bufs, (1, 1, 1, 1, 2, 1)))
        ho = CsrPlan(n_dst=u2, n_src=u2, fwd_ptr=ho_fwd_ptr[: u2 + 1], fwd_idx=ho_fwd_idx, fwd_val=ho_fwd_val,
                     bwd_ptr=ho_bwd_ptr[: u2 + 1], bwd_idx=ho_bwd_idx, bwd_val=ho_bwd_val, self_coef=ho_self[:u2], edge_ordered=True)
        fo = CsrPlan(n_dst=n, n_src=n, fwd_ptr=fo_fwd_ptr, fwd_idx=fo_fwd_idx[:a1], fwd_val=fo_fwd_val[:a1],
                     bwd_ptr=fo_bwd_ptr, bwd_idx=fo_bwd_idx[:u2], bwd_val=fo_bwd_val[:u2], self_coef=fo_self,
                     dst_order=fo_dst_order[:a1], edge_ordered=True)
        if not launched:
            check(L.pp_debruijn2_fill(tcode, m, n, kind, di, df, _p(weight), _p(fo_bwd_ptr), _p(fo_bwd_idx), _p(fo_w), _p(fo_fwd_ptr), _p(ho_fwd_ptr),
                                      _p(ho_bwd_ptr), _p(ho_deg), _p(fo_deg), a2, _p(ho.fwd_idx), _p(ho.fwd_val), _p(ho.bwd_idx), _p(ho.bwd_val),
                                      _p(ho.self_coef), _p(fo.fwd_idx), _p(fo.fwd_val), _p(fo.dst_order), _p(fo.bwd_val), _p(fo.self_coef), _p(ho_fwd_w),
                                      _p(pack), _p(ws), ws.numel(), *hub_args, 1, _stream()),
                  "pp_debruijn2_fill")
        # hub rows of the plans (more than 512 entries): the chunk tables of the row kernels' pre-pass, as pp_gcn_plan's report triggers them
        if longest[0] > HEAVY_ROW_ENTRIES:
            ho.fwd_heavy = HeavyRows(ho.fwd_ptr, u2)
        if longest[1] > HEAVY_ROW_ENTRIES:
            ho.bwd_heavy = HeavyRows(ho.bwd_ptr, u2)
        if longest[2] > HEAVY_ROW_ENTRIES:
            fo.fwd_heavy = HeavyRows(fo.fwd_ptr, n)
        if longest[3] > HEAVY_ROW_ENTRIES:
            fo.bwd_heavy = HeavyRows(fo.bwd_ptr, n)
    bip = bipartite_plan_from_edge_grouping(fo, None, u2)
    return DeBruijn2(fo=fo, ho=ho, bip=bip, fo_weight=fo_w[:u2], fo_dst=fo.bwd_idx, ho_fwd_weight=ho_fwd_w,
                     sizes={"m": m, "N": n, "E2": e2, "U2": u2, "A1": a1, "A2": a2, "hub_nodes": hubs, "hub_tasks": tasks})


class MultiOrderLayer:
    """One De Bruijn layer as source-major CSR (:func:`multi_order_temporal`): ``row_ptr`` int32 [n_nodes + 1], ``col`` int32 [n_edges], ``weight``
    float32 [n_edges] (merged weights), ``last`` int32 [n_edges] (last first-order node of every edge = of every node of the next layer; ``None``
    for the top layer of a build)."""

    __slots__ = ("n_nodes", "n_edges", "n_instances", "row_ptr", "col", "weight", "last")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def multi_order_temporal(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, delta, weight: torch.Tensor | None, max_order: int,
                         clock: list | None = None, event_graph: torch.Tensor | None = None):
    """All De Bruijn layers 1..max_order of a TIME-SORTED event stream, level by level (pp_multiorder_prepare / _step, csrc/pp_multiorder.hip):
    no instance graph ``[2, E_k]``, no per-instance node sequences, no global sort beyond the two of the first order; one read-back per order.
    Returns ``[MultiOrderLayer]`` (index k - 1 = layer k; ``n_instances`` = E_k, the instance edges the reference would have lifted) or ``None``
    when the generic kernels have to take over: a layer without edges, a node sequence with more than 4096 continuations (dense contact
    streams), 2^31 or more instances at some order, an unsorted stream.  ``clock``: a list that receives one ``(name, start event, end event)``
    per phase — the windows and level 1 ("prepare"), then every step ("layer k") — for measurements (bench.py).  ``event_graph``: a given
    ``lift_order_temporal(g, delta)`` ([2, E2] int64, sorted by source) instead of ``time`` / ``delta`` (pp_multiorder_prepare_graph)."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, time, weight, event_graph)

    def tick():
        if clock is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev
    if time.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        time = time.to(torch.int64)
    if time.dtype not in (torch.int64, torch.float64):
        raise TypeError(f"timestamps must be int64 or float64, got {time.dtype}")
    time = time.contiguous()
    m, n = ei.size(1), int(num_nodes)
    if time.numel() != m:
        raise ValueError("time and edge_index disagree on the number of events")
    if m == 0 or n == 0 or m >= _INT32_ROWS:
        return None
    if weight is not None:
        if weight.dtype != torch.float32 or weight.numel() != m:
            return None
        weight = weight.contiguous()
    kind, di, df = resolve_delta(time.dtype, delta)
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        t0 = tick()
        if event_graph is None:
            lift_ws = _workspace(L.pp_temporal_ws_bytes(m, n), dev)
            check(L.pp_temporal_windows(_p(ei), _p(time), _DTYPE_CODE[time.dtype], m, n, kind, di, df, _p(lift_ws), lift_ws.numel(), _stream()),
                  "pp_temporal_windows")
            n_list = m
        else:
            eg = _edge_index(event_graph)
            n_list = eg.size(1)
            if n_list == 0 or n_list >= _INT32_ROWS:
                return None
            lift_ws = _workspace(L.pp_multiorder_graph_ws_bytes(m, n_list), dev)
        tab = torch.empty((n_list, 4), **i32)
        inst = torch.empty((m, 4), **i32)
        tptr, ibase = torch.empty(m + 1, **i32), torch.empty(m + 1, **i32)
        tlast, w = torch.empty(m, **i32), torch.empty(m, **f32)
        row_ptr = torch.empty(n + 1, **i32)
        ws = _workspace(L.pp_multiorder_prepare_ws_bytes(m), dev)
        # the (source, target, time) order of the events: every node's out-list sorted in LDS; a node with more than 4096 out-events (status bit 4:
        # its list is left unsorted) sends the stream to the radix sort — contact-shaped streams go there at once
        for radix in ((True,) if (event_graph is not None or m > 1024 * n) else (False, True)):
            if event_graph is None:
                check(L.pp_multiorder_prepare(_p(ei), m, n, _p(weight), _p(lift_ws), lift_ws.numel(), 1 if radix else 0, _p(tab), _p(inst), _p(tptr),
                                              _p(ibase), _p(tlast), _p(w), _p(row_ptr), _p(ws), ws.numel(), _stream()), "pp_multiorder_prepare")
            else:
                check(L.pp_multiorder_prepare_graph(_p(ei), m, n, _p(weight), _p(eg), n_list, _p(lift_ws), lift_ws.numel(), _p(tab), _p(inst), _p(tptr),
                                                    _p(ibase), _p(tlast), _p(w), _p(row_ptr), _p(ws), ws.numel(), _stream()), "pp_multiorder_prepare_graph")
            types, status, children, _ = ws[:32].view(torch.int64).tolist()
            if not status & 16:
                break
        if clock is not None:
            clock.append(("prepare", t0, tick()))
        del lift_ws, ws
        _bad_index(status, "MultiOrderModel.from_temporal_graph")
        if status & (2 | 4):
            return None                  # (unsorted: the caller sorts and takes the generic path; 2^31 continuations of one node pair's events)
        if event_graph is not None and children != n_list:      # the instances of level 2 ARE the event graph's edges
            return None
        layers = [MultiOrderLayer(n_nodes=n, n_edges=types, n_instances=m, row_ptr=row_ptr, col=tlast[:types], weight=w[:types], last=tlast[:types])]
        col, cand_ptr, cand_last = tlast, row_ptr, tlast
        for k in range(2, max_order + 1):
            if types == 0 or children == 0 or children >= _INT32_ROWS:
                return None
            last = k == max_order
            child = torch.empty((children, (1 if weight is None else 2) if last else 4), **i32)      # (the top layer's children are nobody's parents)
            row_next = torch.empty(types + 1, **i32)
            col_next, w_next = torch.empty(children, **i32), torch.empty(children, **f32)
            tptr_next = ibase_next = tlast_next = None
            if not last:
                tptr_next, ibase_next, tlast_next = torch.empty(children + 1, **i32), torch.empty(children + 1, **i32), torch.empty(children, **i32)
            ws = _workspace(L.pp_multiorder_step_ws_bytes(types, children), dev)
            t0 = tick()
            check(L.pp_multiorder_step(types, children, _p(tptr), _p(ibase), _p(col), _p(inst), _p(cand_ptr), _p(cand_last), _p(tab),
                                       0 if weight is None else 1, 1 if last else 0, _p(child), _p(row_next), _p(tptr_next), _p(ibase_next),
                                       _p(tlast_next), _p(col_next), _p(w_next), _p(ws), ws.numel(), _stream()), "pp_multiorder_step")
            if clock is not None:
                clock.append((f"layer {k}", t0, tick()))
            new_types, status, new_children, _ = ws[:32].view(torch.int64).tolist()
            del ws
            if status & 4:
                return None
            if 2 * new_types < children:                    # far fewer types than instances: do not keep the instance-sized buffers alive
                col_next, w_next = col_next[:new_types].clone(), w_next[:new_types].clone()
                if not last:
                    tlast_next = tlast_next[:new_types].clone()
            layers.append(MultiOrderLayer(n_nodes=types, n_edges=new_types, n_instances=children, row_ptr=row_next, col=col_next[:new_types],
                                          weight=w_next[:new_types], last=None if last else tlast_next[:new_types]))
            tptr, ibase, col, inst = tptr_next, ibase_next, col_next, child
            cand_ptr, cand_last = row_next, tlast_next
            types, children = new_types, new_children
    return layers


class DeBruijn2Part:
    """Count phase of the order-2 builder on ONE RANK's node range (:func:`debruijn2_part_count`): sizes on the host, everything else on
    the device until :func:`debruijn2_part_fill`."""

    __slots__ = ("m", "n", "lo", "n_own", "world", "args", "ws", "bufs", "u2", "status", "a2", "e2", "a1", "n_halo", "n_send", "recv_counts", "send_counts",
                 "row_of", "send_slot", "ho_deg", "succ", "bip", "pad_rows")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def debruijn2_part_count(edge_index: torch.Tensor, time: torch.Tensor, num_nodes: int, node_lo: int, node_hi: int, cuts_dev: torch.Tensor,
                         rank: int, delta, weight: torch.Tensor | None = None, pad_rows: int | None = None) -> "DeBruijn2Part | None":
    """The order-2 builder for the rank that owns the nodes ``[node_lo, node_hi)`` of a partitioned stream: ``edge_index`` / ``time`` hold the
    (time-sorted) events that start or end in that range.  Everything :func:`debruijn2` counts, restricted to the owned middle nodes, plus the
    halo numbering (sources (a, b) with a foreign a, in (owner of a, b, a) order behind the owned rows) and the LOCAL ROW ORDER: owned rows (b, c)
    whose c another rank owns come first, grouped by that rank, ordered by (c, b) = the receiver's halo order — the send list of every exchange
    is the prefix ``[0, n_send)`` of a row matrix; ``row_of[local row]`` = lexicographic row.  ONE read-back.  ``None``: inputs the builder does
    not take — a shard without events or without nodes, a weight that is not float32 (the caller reports it in its agreement collective and
    every rank falls back).  Bad input is NOT raised here: a rank sees only its own events, so the caller gathers ``status`` (bit 0: node index
    out of range, bit 1: events not sorted by time, bit 2: a node with more than 64 in- / out-events) from all ranks and raises everywhere
    (a rank-local raise in front of a collective leaves the other ranks blocked in it)."""
    ei = _edge_index(edge_index)
    dev = require_device(ei, time, weight, cuts_dev)
    if time.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        time = time.to(torch.int64)
    if time.dtype not in (torch.int64, torch.float64):
        raise TypeError(f"timestamps must be int64 or float64, got {time.dtype}")
    time = time.contiguous()
    m, n = ei.size(1), int(num_nodes)
    world = int(cuts_dev.numel()) - 1
    n_own = int(node_hi) - int(node_lo)
    if m == 0 or n_own <= 0:            # (pp_debruijn2_part_count leaves the shard-level outputs of such a rank unwritten: ADVICE r4)
        return None
    if weight is not None:
        if weight.dtype != torch.float32 or weight.numel() != m:
            return None
        weight = weight.contiguous()
    pad_rows = max(int(pad_rows) if pad_rows is not None else n, 1)          # rows per rank of the padded first-order layout (>= the largest range)
    kind, di, df = resolve_delta(time.dtype, delta)
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        cap = max(m, 1)
        ws = _workspace(L.pp_debruijn2_ws_bytes(m, n), dev)
        bufs = {"fo_bwd_ptr": torch.empty(n_own + 1, **i32), "fo_fwd_ptr": torch.empty(n_own + 1, **i32), "fo_bwd_idx": torch.empty(cap, **i32),
                "fo_w": torch.empty(cap, **f32), "ho_fwd_ptr": torch.empty(cap + 1, **i32), "ho_bwd_ptr": torch.empty(cap + 1, **i32),
                "ho_deg": torch.empty(cap, **f32), "fo_deg": torch.empty(max(n, 1), **f32), "row_of": torch.empty(cap, **i32),
                "send_slot": torch.empty(cap, **i32), "fo_shard_bwd_ptr": torch.empty(n + 1, **i32),
                "bip_fwd_ptr": torch.empty(world * pad_rows + 1, **i32), "bip_fwd_idx": torch.empty(cap, **i32), "bip_bwd_ptr": torch.empty(cap + 1, **i32),
                "bip_bwd_idx": torch.empty(cap, **i32), "bip_self": torch.empty(world * pad_rows, **f32)}
        tcode = _DTYPE_CODE[time.dtype]
        check(L.pp_debruijn2_part_count(_p(ei), _p(time), tcode, m, n, int(node_lo), n_own, _p(cuts_dev), world, int(rank), kind, di, df, _p(weight),
                                        _p(bufs["fo_bwd_ptr"]), _p(bufs["fo_bwd_idx"]), _p(bufs["fo_w"]), _p(bufs["fo_fwd_ptr"]), _p(bufs["ho_fwd_ptr"]),
                                        _p(bufs["ho_bwd_ptr"]), _p(bufs["ho_deg"]), _p(bufs["fo_deg"]), _p(bufs["send_slot"]), _p(bufs["row_of"]), _p(bufs["fo_shard_bwd_ptr"]), pad_rows, _p(bufs["bip_fwd_ptr"]),
                                        _p(bufs["bip_fwd_idx"]), _p(bufs["bip_bwd_ptr"]), _p(bufs["bip_bwd_idx"]), _p(bufs["bip_self"]),
                                        _p(ws), ws.numel(), _stream()), "pp_debruijn2_part_count")
        head = ws[: 8 * (8 + 2 * (world + 1))].view(torch.int64).tolist()                 # the ONE read-back of this rank's graph construction
    u2, status, a2, e2, a1, n_halo, n_send = head[:7]
    if a2 >= _INT32_ROWS or u2 + n_halo >= _INT32_ROWS:          # int32 row pointers / local ids would wrap: reported like a hub node, every rank falls back
        status |= 4
    recv_ptr, send_ptr = head[8: 8 + world + 1], head[8 + world + 1: 8 + 2 * (world + 1)]
    return DeBruijn2Part(m=m, n=n, lo=int(node_lo), n_own=n_own, world=world, args=(tcode, kind, di, df, weight), ws=ws, bufs=bufs, u2=u2, status=status,
                         a2=a2, e2=e2, a1=a1, n_halo=n_halo, n_send=n_send, recv_counts=[recv_ptr[r + 1] - recv_ptr[r] for r in range(world)],
                         send_counts=[send_ptr[r + 1] - send_ptr[r] for r in range(world)], row_of=bufs["row_of"][:u2],
                         send_slot=bufs["send_slot"][:u2], ho_deg=bufs["ho_deg"], succ=bufs["fo_bwd_idx"][:u2], pad_rows=pad_rows,
                         bip=CsrPlan(n_dst=world * pad_rows, n_src=u2, fwd_ptr=bufs["bip_fwd_ptr"], fwd_idx=bufs["bip_fwd_idx"][:u2], fwd_val=None,
                                     bwd_ptr=bufs["bip_bwd_ptr"][: u2 + 1], bwd_idx=bufs["bip_bwd_idx"][:u2], bwd_val=None, self_coef=bufs["bip_self"]))


def debruijn2_part_fill(c: DeBruijn2Part):
    """Fill phase (after the weighted degrees of the halo rows arrived in ``c.ho_deg[u2: u2 + n_halo]`` and ``c.bufs["fo_deg"]`` holds the degrees of
    ALL first-order nodes): ``(ho CsrPlan over [owned | halo] sources, fo CsrPlan: destinations = owned nodes, sources = all nodes in the dense
    local order [owned | ids below lo | ids from hi on], order-2 nodes per owned first-order node)``."""
    dev = c.ws.device
    tcode, kind, di, df, weight = c.args
    b = c.bufs
    L = lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        n_src = c.u2 + c.n_halo
        # source-major arrays with room for n_send entries in front: the AUGMENTED rows of the backward pass (see `aug` below) are
        # [one unit entry per sent row | the plain entries], i.e. the same arrays read from their start
        bwd_idx_all, bwd_val_all = torch.empty(c.n_send + c.a2, **i32), torch.empty(c.n_send + c.a2, **f32)
        ho = CsrPlan(n_dst=c.u2, n_src=n_src, fwd_ptr=b["ho_fwd_ptr"][: c.u2 + 1], fwd_idx=torch.empty(c.a2, **i32), fwd_val=torch.empty(c.a2, **f32),
                     bwd_ptr=b["ho_bwd_ptr"][: n_src + 1], bwd_idx=bwd_idx_all[c.n_send:], bwd_val=bwd_val_all[c.n_send:],
                     self_coef=torch.empty(c.u2, **f32))
        fo = CsrPlan(n_dst=c.n_own, n_src=c.n, fwd_ptr=b["fo_fwd_ptr"], fwd_idx=torch.empty(c.a1, **i32), fwd_val=torch.empty(c.a1, **f32),
                     bwd_ptr=b["fo_shard_bwd_ptr"], bwd_idx=torch.empty(c.a1, **i32), bwd_val=torch.empty(c.a1, **f32),
                     self_coef=torch.empty(c.n_own, **f32))
        if c.m > 0 and c.n_own > 0:
            check(L.pp_debruijn2_part_fill(tcode, c.m, c.n, c.lo, c.n_own, kind, di, df, _p(weight), _p(b["fo_bwd_ptr"]), _p(b["fo_fwd_ptr"]),
                                           _p(b["ho_fwd_ptr"]), _p(b["ho_bwd_ptr"]), _p(b["ho_deg"]), _p(b["fo_deg"]), c.a2, _p(ho.fwd_idx), _p(ho.fwd_val),
                                           _p(ho.bwd_idx), _p(ho.bwd_val), _p(ho.self_coef), _p(fo.fwd_idx), _p(fo.fwd_val), _p(fo.self_coef),
                                           _p(b["fo_shard_bwd_ptr"]), _p(fo.bwd_idx), _p(fo.bwd_val), _p(torch.empty(2 * c.a2, **i32)), _p(c.ws),
                                           c.ws.numel(), _stream()), "pp_debruijn2_part_fill")
        indeg = (b["fo_fwd_ptr"][1:] - b["fo_fwd_ptr"][:-1]).to(torch.float32)
        # Augmented source-major rows of the OWNED rows for the backward pass: a sent row (local id k < n_send) has no local out-edge — its
        # out-edges live on the rank that owns its head node, whose partial sum comes back as row k of the returned halo sums.  With those
        # stored behind dpre ([dpre | recv]) the returned sum is ONE more CSR entry (n_own + k, 1.0) of row k, and the fused backward kernels
        # (pp_gcn_backward_f32 / pp_gcn_input_grad_f32) do the fold, the self term, both products and the weight gradient in one pass.
        torch.arange(c.u2, c.u2 + c.n_send, out=bwd_idx_all[: c.n_send])
        bwd_val_all[: c.n_send] = 1.0
        aug_ptr = b["ho_bwd_ptr"][: c.u2 + 1] + torch.arange(c.u2 + 1, **i32).clamp_(max=c.n_send)
        ho.aug = (aug_ptr, bwd_idx_all, bwd_val_all)
    return ho, fo, indeg


def spmm(ptr, idx, val, n_rows: int, x: torch.Tensor, self_coef=None, s=None, bias=None, act: bool = False,
         heavy: HeavyRows | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Y[r] = act(sum_p val[p] * x[idx[p]] + self_coef[r] * s[r] + bias) — fp32, rows of width F.  ``heavy``: the hub rows of this
    CSR (``plan.fwd_heavy`` / ``plan.bwd_heavy``), summed by the chunked pre-pass."""
    dev = require_device(x, s, bias)
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise TypeError("DBGNN kernels are fp32")
    f = x.size(1)
    if f > 256:          # the row kernels own at most 64 lanes x 4 columns per row: wider matrices go through in 256-column blocks
        blocks = [spmm(ptr, idx, val, n_rows, x[:, c: c + 256], self_coef, None if s is None else s[:, c: c + 256],
                       None if bias is None else bias[c: c + 256], act, heavy) for c in range(0, f, 256)]
        y = torch.cat(blocks, dim=1)
        return y if out is None else out.copy_(y)
    if s is not None:
        s = s.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    slot, sums = _heavy_args(heavy, idx, val, x)
    with torch.cuda.device(dev):
        y = torch.empty((n_rows, f), dtype=torch.float32, device=dev) if out is None else out
        if out is not None and (tuple(out.shape) != (n_rows, f) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise ValueError("spmm: out must be a contiguous fp32 [n_rows, F] tensor")
        check(lib().pp_spmm_f32(_p(ptr), _p(idx), _p(val), n_rows, _p(x), f, _p(self_coef), _p(s), _p(bias), 1 if act else 0,
                                _p(slot), _p(sums), _p(y), _stream()), "pp_spmm_f32")
    return y


def bip_combine(a: torch.Tensor, p: torch.Tensor, deg: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """``ELU(a + deg[:, None] * (p + bias))`` in one pass (pp_bip_combine_f32)."""
    dev = require_device(a, p, deg, bias)
    a, p, deg = a.contiguous(), p.contiguous(), deg.to(torch.float32).contiguous()
    if a.dtype != torch.float32 or p.shape != a.shape or deg.numel() != a.size(0):
        raise ValueError("bip_combine: a, p [n, F] fp32 and deg [n] expected")
    with torch.cuda.device(dev):
        y = torch.empty_like(a)
        check(lib().pp_bip_combine_f32(_p(a), _p(p), _p(deg), _p(None if bias is None else bias.contiguous()), a.size(0), a.size(1), _p(y), _stream()),
              "pp_bip_combine_f32")
    return y


def bip_combine_backward(dy: torch.Tensor, y: torch.Tensor, deg: torch.Tensor, want_bias: bool):
    """Backward of :func:`bip_combine`: ``(d_a, d_p, d_bias or None)`` in one pass (pp_bip_combine_backward_f32)."""
    dev = require_device(dy, y, deg)
    dy, deg = dy.contiguous(), deg.to(torch.float32).contiguous()
    with torch.cuda.device(dev):
        da, dp = torch.empty_like(y), torch.empty_like(y)
        db = torch.empty(y.size(1), dtype=torch.float32, device=dev) if want_bias else None
        check(lib().pp_bip_combine_backward_f32(_p(dy), _p(y), _p(deg), y.size(0), y.size(1), _p(da), _p(dp), _p(db), _stream()),
              "pp_bip_combine_backward_f32")
    return da, dp, db


def act_backward(dy: torch.Tensor, y: torch.Tensor | None, act: bool, want_dpre: bool = True, want_dbias: bool = False):
    dev = require_device(dy, y)
    dy = dy.contiguous()
    n, f = dy.shape
    with torch.cuda.device(dev):
        dpre = torch.empty_like(dy) if want_dpre else None
        dbias = torch.empty(f, dtype=torch.float32, device=dev) if want_dbias else None
        check(lib().pp_act_backward_f32(_p(dy), _p(y), n, f, 1 if act else 0, _p(dpre), _p(dbias), _stream()), "pp_act_backward_f32")
    return dpre, dbias


def dropout(x: torch.Tensor, p: float, seed: int, tag: int, row0: int = 0, rows: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """Training-mode dropout with counter-based masks (see pp_dropout_f32): ``x * keep / (1 - p)``; ``out`` may be ``x`` itself."""
    dev = require_device(x, rows)
    x = x.contiguous()
    n, f = x.shape
    if rows is not None:
        rows = rows.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty_like(x)
        check(lib().pp_dropout_f32(_p(x), n, f, float(p), int(seed), int(tag), int(row0), _p(rows), _p(out), _stream()), "pp_dropout_f32")
    return out


def dropout_act_backward(dy: torch.Tensor, y_dropped: torch.Tensor | None, p: float, seed: int, tag: int, row0: int = 0,
                         rows: torch.Tensor | None = None, act: bool = True, want_dbias: bool = False):
    """``(dpre, dbias or None)``: backward of :func:`dropout` fused with the ELU backward of the activation underneath (``act``)."""
    dev = require_device(dy, y_dropped, rows)
    dy = dy.contiguous()
    n, f = dy.shape
    if y_dropped is not None:
        y_dropped = y_dropped.contiguous()
    if rows is not None:
        rows = rows.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        dpre = torch.empty_like(dy)
        dbias = torch.empty(f, dtype=torch.float32, device=dev) if want_dbias else None
        check(lib().pp_dropout_act_backward_f32(_p(dy), _p(y_dropped), n, f, float(p), int(seed), int(tag), int(row0), _p(rows), 1 if act else 0,
                                                _p(dpre), _p(dbias), _stream()), "pp_dropout_act_backward_f32")
    return dpre, dbias


def halo_fold(own: torch.Tensor, recv: torch.Tensor | None = None, slot: torch.Tensor | None = None, extra: torch.Tensor | None = None,
              self_coef: torch.Tensor | None = None, dpre: torch.Tensor | None = None, inplace: bool = True) -> torch.Tensor:
    """``own + recv[slot] (where slot >= 0) + extra + self_coef[:, None] * dpre`` in one pass (pp_halo_fold_f32); written into ``own`` when
    ``inplace``.  ``slot`` int32 [n]; fp32 matrices of one width (a multiple of 4)."""
    dev = require_device(own, recv, slot, extra, self_coef, dpre)
    if not own.is_contiguous():
        raise ValueError("halo_fold: own must be contiguous")
    if own.numel() == 0:
        return own if inplace else own.clone()
    if recv is not None and (slot is None or recv.size(0) == 0):
        recv = slot = None
    if slot is not None and recv is None:
        slot = None
    recv = None if recv is None else recv.contiguous()
    extra = None if extra is None else extra.contiguous()
    dpre = None if dpre is None else dpre.contiguous()
    if slot is not None and slot.dtype != torch.int32:
        raise TypeError("halo_fold: slot must be int32")
    with torch.cuda.device(dev):
        out = own if inplace else torch.empty_like(own)
        check(lib().pp_halo_fold_f32(_p(own), _p(recv), _p(slot), _p(extra), _p(self_coef), _p(dpre), own.size(0), own.size(1), _p(out), _stream()),
              "pp_halo_fold_f32")
    return out


def gather_rows(x: torch.Tensor, rows: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """``x[rows]`` for an fp32 matrix whose width is a multiple of 4 (pp_spmm_f32 with one unit entry per output row)."""
    rows32 = rows.to(torch.int32).contiguous()
    n = int(rows32.numel())
    ptr = torch.arange(n + 1, dtype=torch.int32, device=rows32.device)
    return spmm(ptr, rows32, None, n, x, out=out)


def scale_rows(x: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    dev = require_device(x, coef)
    x = x.contiguous()
    with torch.cuda.device(dev):
        out = torch.empty_like(x)
        check(lib().pp_scale_rows_f32(_p(x), _p(coef), x.size(0), x.size(1), _p(out), _stream()), "pp_scale_rows_f32")
    return out


def weight_grad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool):
    """(dW [M,K], db [M] or None) of y = x W^T (+ b) from dy [N,M] and x [N,K] — MFMA kernel, fp32."""
    dev = require_device(dy, x)
    dy, x = dy.contiguous(), x.contiguous()
    n, m = dy.shape
    k = x.size(1)
    L = lib()
    with torch.cuda.device(dev):
        dw = torch.empty((m, k), dtype=torch.float32, device=dev)
        db = torch.empty(m, dtype=torch.float32, device=dev) if want_bias else None
        ws = _workspace(L.pp_weight_grad_ws_bytes(n, m, k), dev)
        check(L.pp_weight_grad_f32(_p(dy), _p(x), n, m, k, _p(dw), _p(db), _p(ws), ws.numel(), _stream()), "pp_weight_grad_f32")
    return dw, db


def dense_supported(p: int, q: int) -> int:
    """0: no HIP kernel for this layer shape (library GEMM); 1: widths 16/32/64 (also :func:`dense_backward`); 2: other widths whose
    zero-padded product fits the register-resident kernel (anything up to 64 x 64, 256 x 8, 8 x 256, ...); 3: 64/128/256 with a side > 64 (weights streamed through LDS)."""
    return int(lib().pp_dense_supported(int(p), int(q)))


def dense(a: torch.Tensor, weight: torch.Tensor, transposed: bool, bias: torch.Tensor | None = None,
          grad_act: torch.Tensor | None = None, want_colsum: bool = False):
    """``out = (a @ (weight.T if transposed else weight) + bias) * elu'`` on the matrix cores (fp32) where ``elu'`` is the ELU
    derivative recovered from the stored activation ``grad_act`` (``y > 0 ? 1 : y + 1``); returns ``(out, colsum or None)``.
    Shapes must satisfy :func:`dense_supported`."""
    dev = require_device(a, weight, bias, grad_act)
    a, weight = a.contiguous(), weight.contiguous()
    n, p = a.shape
    q = weight.size(0) if transposed else weight.size(1)
    if (weight.size(1) if transposed else weight.size(0)) != p:
        raise ValueError("dense: inner dimensions do not match")
    if bias is not None:
        bias = bias.contiguous()
    if grad_act is not None:
        grad_act = grad_act.contiguous()
    L = lib()
    with torch.cuda.device(dev):
        out = torch.empty((n, q), dtype=torch.float32, device=dev)
        colsum = torch.empty(q, dtype=torch.float32, device=dev) if want_colsum else None
        ws = _workspace(L.pp_wide_layer_ws_bytes(p, q), dev) if L.pp_dense_supported(p, q) == 3 else None       # (chunk-major / transposed W)
        check(L.pp_dense_f32(_p(a), _p(weight), 1 if transposed else 0, n, p, q, _p(bias), _p(grad_act), _p(colsum), _p(out), _p(ws),
                             0 if ws is None else ws.numel(), _stream()), "pp_dense_f32")
    return out, colsum


def spmm_act_backward(ptr, idx, val, n_rows: int, d: torch.Tensor, z: torch.Tensor, want_colsum: bool, drop: tuple | None = None,
                      out: torch.Tensor | None = None):
    """``dx = (A d) * elu'(z)`` with ``elu'`` from the stored activation ``z`` (``z > 0 ? 1 : z + 1``) and, optionally, the
    column sums of ``dx`` — one pass.  ``drop = (p, seed, tag, row0)``: ``z`` is stored dropped; mask and ``1 / (1 - p)`` go into ``dx``."""
    dev = require_device(d, z)
    dp, dseed, dtag, drow0 = _drop_args(drop)
    d, z = d.contiguous(), z.contiguous()
    f = d.size(1)
    with torch.cuda.device(dev):
        dx = torch.empty((n_rows, f), dtype=torch.float32, device=dev) if out is None else out
        if out is not None and (tuple(out.shape) != (n_rows, f) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise ValueError("spmm_act_backward: out must be a contiguous fp32 [n_rows, F] tensor")
        colsum = torch.empty(f, dtype=torch.float32, device=dev) if want_colsum else None
        check(lib().pp_spmm_act_backward_drop_f32(_p(ptr), _p(idx), _p(val), n_rows, _p(d), f, _p(z), _p(colsum), _p(dx), dp, dseed, dtag, drow0,
                                                  _stream()), "pp_spmm_act_backward_drop_f32")
    return dx, colsum


def _drop_args(drop):
    if drop is None:
        return 0.0, 0, 0, 0
    p, seed, tag, row0 = drop
    return float(p), int(seed), int(tag), int(row0)


def gcn_drop_supported(p: int, q: int) -> bool:
    """Layer shapes whose fused kernels take the ``drop`` argument (16/32/64 and the 128-wide shapes)."""
    return bool(lib().pp_gcn_drop_supported(int(p), int(q)))


def gcn_forward(ptr: torch.Tensor, idx: torch.Tensor, val: torch.Tensor | None, n_rows: int, x: torch.Tensor,
                self_coef: torch.Tensor | None, weight: torch.Tensor, bias: torch.Tensor | None, act: bool, want_agg: bool = False,
                heavy: HeavyRows | None = None, out: torch.Tensor | None = None, drop: tuple | None = None):
    """``act((A x + diag(self_coef) x) @ weight.T + bias)`` in one kernel (aggregation fused with the MFMA product);
    ``want_agg``: returns ``(y, A x + diag(self_coef) x)``.  ``out``: a contiguous ``[n_rows, Q]`` fp32 tensor to write ``y`` into
    (e.g. the head of a buffer whose tail receives halo rows).  ``drop = (p, seed, tag, row0)``: training-mode dropout of ``y`` fused into
    the epilogue (counter-based mask of :func:`dropout`; needs :func:`gcn_drop_supported`)."""
    dev = require_device(ptr, idx, val, x, self_coef, weight, bias)
    dp, dseed, dtag, drow0 = _drop_args(drop)
    x, weight = x.contiguous(), weight.contiguous()
    q, p = weight.shape
    if x.size(1) != p:
        raise ValueError("gcn_forward: inner dimensions do not match")
    if bias is not None:
        bias = bias.contiguous()
    with torch.cuda.device(dev):
        if out is None:
            y = torch.empty((n_rows, q), dtype=torch.float32, device=dev)
        else:
            if tuple(out.shape) != (n_rows, q) or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError("gcn_forward: out must be a contiguous fp32 [n_rows, Q] tensor")
            y = out
        agg = torch.empty((n_rows, p), dtype=torch.float32, device=dev) if want_agg else None
        slot, sums = _heavy_args(heavy, idx, val, x)
        check(lib().pp_gcn_forward_drop_f32(_p(ptr), _p(idx), _p(val), n_rows, x.size(0), _p(x), p, _p(self_coef), _p(weight), q, _p(bias),
                                            1 if act else 0, _p(slot), _p(sums), _p(agg), _p(y), dp, dseed, dtag, drow0, _stream()),
              "pp_gcn_forward_drop_f32")
    return (y, agg) if want_agg else y


def gcn_backward(ptr, idx, val, n_rows: int, dpre: torch.Tensor, self_coef, x: torch.Tensor, weight: torch.Tensor, fuse_act: bool,
                 want_colsum: bool, heavy: HeavyRows | None = None, n_self: int | None = None, drop: tuple | None = None):
    """Backward of :func:`gcn_forward` in one kernel: ``(d_in, colsum_in or None, dW)`` from the gradient ``dpre`` w.r.t. the
    layer's pre-activation, over the SOURCE-major CSR (``ptr``/``idx``/``val`` = the plan's ``bwd_*`` arrays).
    ``n_self`` (default ``n_rows``): partition plans — ``n_rows`` = owned + halo source rows, ``dpre`` has the ``n_self`` owned rows.
    ``drop = (p, seed, tag, row0)`` (with ``fuse_act``): ``x`` is the DROPPED activation of the layer below; its mask and ``1 / (1 - p)`` go
    into ``d_in`` together with the ELU'."""
    dev = require_device(ptr, idx, val, dpre, self_coef, x, weight)
    dp, dseed, dtag, drow0 = _drop_args(drop)
    dpre, x, weight = dpre.contiguous(), x.contiguous(), weight.contiguous()
    m, k = weight.shape
    n_self = n_rows if n_self is None else int(n_self)
    # (dpre may hold MORE rows than n_self: an augmented CSR of a partition shard also gathers the returned halo sums stored behind the owned rows)
    if dpre.size(1) != m or x.size(1) != k or x.size(0) != n_rows or dpre.size(0) < n_self or n_self > n_rows:
        raise ValueError("gcn_backward: shapes do not match")
    L = lib()
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        d_in = torch.empty((n_rows, k), **f32)
        colsum = torch.empty(k, **f32) if want_colsum else None
        dw = torch.empty((m, k), **f32)
        slot, sums = _heavy_args(heavy, idx, val, dpre)
        ws = _workspace(L.pp_gcn_backward_ws_bytes(n_rows), dev)
        check(L.pp_gcn_backward_nnz_f32(_p(ptr), _p(idx), _p(val), n_rows, n_self, int(idx.numel()), _p(dpre), m, _p(self_coef), _p(x), k, _p(weight),
                                        1 if fuse_act else 0, _p(slot), _p(sums), _p(d_in), _p(colsum), _p(dw), _p(ws), ws.numel(), dp, dseed, dtag,
                                        drow0, _stream()), "pp_gcn_backward_nnz_f32")
    return d_in, colsum, dw


def gcn_fused_supported(p: int, q: int) -> int:
    """1: fused forward + one-kernel backward (widths 16/32/64); 2: fused forward + input-gradient kernel (128-wide shapes); 0: neither."""
    return int(lib().pp_gcn_fused_supported(int(p), int(q)))


def gcn_input_grad(ptr, idx, val, n_rows: int, dpre: torch.Tensor, self_coef, weight: torch.Tensor, x_act: torch.Tensor | None,
                   want_colsum: bool, heavy: HeavyRows | None = None, n_self: int | None = None, drop: tuple | None = None):
    """``((A^T dpre + diag(self_coef) dpre) @ weight) * elu'(x_act)`` (no activation factor when ``x_act`` is None) and optionally its
    column sums, over the SOURCE-major CSR — the input gradient of a 128-wide fused layer (``drop``: as in :func:`gcn_backward`)."""
    dev = require_device(ptr, idx, val, dpre, self_coef, weight, x_act)
    dp, dseed, dtag, drow0 = _drop_args(drop)
    dpre, weight = dpre.contiguous(), weight.contiguous()
    m, k = weight.shape
    n_self = n_rows if n_self is None else int(n_self)
    if dpre.size(1) != m or dpre.size(0) < n_self or n_self > n_rows or (x_act is not None and tuple(x_act.shape) != (n_rows, k)):
        raise ValueError("gcn_input_grad: shapes do not match")
    if x_act is not None:
        x_act = x_act.contiguous()
    L = lib()
    with torch.cuda.device(dev):
        d_in = torch.empty((n_rows, k), dtype=torch.float32, device=dev)
        colsum = torch.empty(k, dtype=torch.float32, device=dev) if want_colsum else None
        slot, sums = _heavy_args(heavy, idx, val, dpre)
        ws = _workspace(L.pp_wide_layer_ws_bytes(m, k), dev)              # (only the shapes with a side of 256 use it: W^T)
        check(L.pp_gcn_input_grad_drop_f32(_p(ptr), _p(idx), _p(val), n_rows, n_self, _p(dpre), m, _p(self_coef), _p(weight), k, _p(x_act),
                                           1 if x_act is not None else 0, _p(slot), _p(sums), _p(d_in), _p(colsum), _p(ws), ws.numel(), dp, dseed,
                                           dtag, drow0, _stream()), "pp_gcn_input_grad_drop_f32")
    return d_in, colsum


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, want_grad: bool = True):
    """(mean cross-entropy [scalar tensor], d loss / d logits or None) in one pass; logits [N, C<=64] fp32, target int64 [N]."""
    dev = require_device(logits, target)
    logits = logits.contiguous()
    target = target.to(torch.int64).contiguous()
    n, c = logits.shape
    with torch.cuda.device(dev):
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(logits) if want_grad else None
        L = lib()
        ws = _workspace(L.pp_cross_entropy_ws_bytes(), dev)
        check(L.pp_cross_entropy_f32(_p(logits), _p(target), n, c, _p(loss), _p(grad), _p(ws), ws.numel(), _stream()), "pp_cross_entropy_f32")
    return loss[0], grad


def dense_backward(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, fuse_act: bool, want_input_grad: bool, want_colsum: bool,
                   want_bias: bool):
    """Backward of ``y = x @ weight.T (+ b)`` in one pass: ``(d_in or None, colsum_in or None, dW, db or None)`` with
    ``d_in = (dy @ weight) * elu'(x)`` when ``fuse_act``.  Shapes must satisfy :func:`dense_supported`."""
    dev = require_device(dy, x, weight)
    dy, x, weight = dy.contiguous(), x.contiguous(), weight.contiguous()
    n, m = dy.shape
    k = x.size(1)
    L = lib()
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        d_in = torch.empty((n, k), **f32) if want_input_grad else None
        colsum = torch.empty(k, **f32) if (want_input_grad and want_colsum) else None
        dw = torch.empty((m, k), **f32)
        db = torch.empty(m, **f32) if want_bias else None
        ws = _workspace(L.pp_dense_backward_ws_bytes(n), dev)
        check(L.pp_dense_backward_f32(_p(dy), _p(x), _p(weight), n, m, k, 1 if fuse_act else 0, _p(d_in), _p(colsum), _p(dw), _p(db),
                                      _p(ws), ws.numel(), _stream()), "pp_dense_backward_f32")
    return d_in, colsum, dw, db
