"""Bridge between the reference-shaped Python API (tensors on any device) and the HIP kernels.

Rule: results come back on the device of the inputs, like the reference does
(src/pathpyG/algorithms/temporal.py:30-31, lift_order.py:76).  CPU inputs are staged to the current
MI355X, processed by the HIP kernels and copied back (a PCIe round trip — keep data on the GPU for
throughput).  Without a GPU or without the built library every call raises; nothing is computed on the CPU.
"""
from __future__ import annotations

import torch

from . import _hip


def compute_device(*tensors) -> torch.device:
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError(
            "pathpyg_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm: its lifts, sorts and DBGNN layers "
            "run as HIP kernels only and there is no CPU fallback"
        )
    return torch.device("cuda", torch.cuda.current_device())


def plain(t):
    """View a tensor subclass (e.g. an edge index wrapper) as a plain torch.Tensor."""
    if isinstance(t, torch.Tensor) and type(t) is not torch.Tensor:
        return t.as_subclass(torch.Tensor)
    return t


def _stage(dev, *tensors):
    return tuple(t if (t is None or isinstance(t, str)) else plain(t).to(dev) for t in tensors)      # (str: the _hip.UNIT weight marker)


def _back(like: torch.Tensor, *outs):
    res = tuple(o if (o is None or o.device == like.device) else o.to(like.device) for o in outs)
    return res[0] if len(res) == 1 else res


def temporal_lift(edge_index, time, num_nodes: int, delta, n_own=None, id_offset: int = 0):
    dev = compute_device(edge_index, time)
    ei, t = _stage(dev, edge_index, time)
    return _back(edge_index, _hip.temporal_lift(ei, t, num_nodes, delta, n_own, id_offset))


def temporal_bfs(edge_index, time, num_nodes: int, delta):
    dev = compute_device(edge_index, time)
    ei, t = _stage(dev, edge_index, time)
    event_graph = _hip.temporal_lift(ei, t, num_nodes, delta)
    return _hip.temporal_bfs(ei, num_nodes, event_graph)


def temporal_betweenness(edge_index, time, num_nodes: int, delta):
    dev = compute_device(edge_index, time)
    ei, t = _stage(dev, edge_index, time)
    event_graph = _hip.temporal_lift(ei, t, num_nodes, delta)
    return _hip.temporal_betweenness(ei, num_nodes, event_graph)


def linegraph_lift(edge_index, num_nodes: int, edge_range=None):
    dev = compute_device(edge_index)
    (ei,) = _stage(dev, edge_index)
    return _back(edge_index, _hip.linegraph_lift(ei, num_nodes, edge_range))


def edge_attr(edge_index, attr, aggr: str):
    dev = compute_device(edge_index, attr)
    ei, a = _stage(dev, edge_index, attr)
    return _back(edge_index, _hip.edge_attr(ei, a, aggr))


def extend_node_sequence(edge_index, node_sequence):
    dev = compute_device(edge_index, node_sequence)
    ei, ns = _stage(dev, edge_index, node_sequence)
    return _back(edge_index, _hip.extend_node_sequence(ei, ns))


def gather_concat(rows, idx, suffix):
    dev = compute_device(rows, idx, suffix)
    r, i, sfx = _stage(dev, rows, idx, suffix)
    return _back(rows, _hip.gather_concat(r, i, sfx))


def unique_rows(rows, value_range=None):
    dev = compute_device(rows)
    (r,) = _stage(dev, rows)
    return _back(rows, *_hip.unique_rows(r, value_range))


def coalesce(edge_index, weight, num_nodes: int, reduce: str = "sum", remap=None, want_inverse: bool = False, col_block=None):
    dev = compute_device(edge_index, None if isinstance(weight, str) else weight, remap)
    ei, w, rm = _stage(dev, edge_index, weight, remap)
    if col_block is not None:
        col_block = (_stage(dev, col_block[0])[0], col_block[1])
    return _back(edge_index, *_hip.coalesce(ei, w, num_nodes, reduce, rm, want_inverse, col_block))


def successor_blocks(block_key, num_blocks: int, node_block):
    """Column blocks for :func:`coalesce` on a De Bruijn layer: the nodes are numbered lexicographically, ``block_key[u]`` (sorted,
    non-decreasing) is the id of node u's prefix and ``node_block[u]`` the prefix id that every SUCCESSOR of u shares.  Returns
    ``(col_base [U], col_bits)``: successors of u have ids in ``[col_base[u], col_base[u] + 2**col_bits)``."""
    dev = compute_device(block_key, node_block)
    key, blk = _stage(dev, block_key, node_block)
    ptr = _hip.ptr_from_sorted(key, num_blocks)
    widest = int((ptr[1:] - ptr[:-1]).max().item()) if num_blocks > 0 else 1
    bits = max(int(widest - 1).bit_length(), 1)
    return ptr[blk], bits


def minmax(a):
    dev = compute_device(a)
    (x,) = _stage(dev, a)
    return _hip.minmax(x)


def is_sorted(a) -> bool:
    if a.numel() < 2:
        return True
    dev = compute_device(a)
    (x,) = _stage(dev, a)
    return _hip.is_sorted(x)


def time_stats(time):
    dev = compute_device(time)
    (t,) = _stage(dev, time)
    return _hip.time_stats(t)


def gather_events(edge_index, time, perm):
    dev = compute_device(edge_index, time, perm)
    ei, t, p = _stage(dev, edge_index, time, perm)
    return _back(edge_index, *_hip.gather_events(ei, t, p))


def stable_argsort(keys, value_range=None):
    dev = compute_device(keys)
    (k,) = _stage(dev, keys)
    return _back(keys, _hip.argsort(k, value_range))


def ptr_from_sorted(sorted_index, num_rows: int):
    dev = compute_device(sorted_index)
    (s,) = _stage(dev, sorted_index)
    return _back(sorted_index, _hip.ptr_from_sorted(s, num_rows))
