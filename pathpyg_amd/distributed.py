"""Multi-GPU sharding of the lift: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL on ROCm).

The reference is single-process; this is new design following SURVEY §8(e):

* temporal lift — **edge-range shards with a forward halo, no data exchange**.  Events are time-sorted, so the
  continuations of the source events ``[lo, hi)`` all lie in ``[lo+1, halo_end)`` with
  ``halo_end = first event with t > t[hi-1] + delta``.  Rank ``r`` lifts its slice ``[lo_r, halo_end_r)`` with only
  the first ``hi_r - lo_r`` events acting as sources (``n_own``) and global ids restored by ``id_offset``; the
  concatenation of the per-rank results in rank order IS the global lexicographic result.  The only collective is
  one all-gather of the per-rank pair counts (8 bytes per rank) for global output offsets.
* ranges are balanced on OUTPUT size when a per-event count estimate is supplied, else on event count.

Everything below the collectives runs through ``pathpyg_amd._dispatch`` (HIP kernels); tests exercise the
planner and the collectives on CPU with the ``gloo`` backend by substituting the local lift.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _dispatch


def _world(group=None) -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def event_ranges(num_events: int, world_size: int, weights: torch.Tensor | None = None) -> list[tuple[int, int]]:
    """Contiguous source-event ranges ``[(lo, hi)] * world_size`` covering ``[0, num_events)``.

    Without ``weights`` the events are split evenly; with per-event ``weights`` (e.g. continuation counts from a
    counting pre-pass) the split equalises the weight per rank (each cut is the first prefix-sum crossing)."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if weights is None or num_events == 0:
        cuts = [(num_events * r) // world_size for r in range(world_size + 1)]
    else:
        if weights.numel() != num_events:
            raise ValueError("weights must hold one value per event")
        prefix = torch.cumsum(weights.to(torch.float64).cpu(), 0)
        total = float(prefix[-1])
        targets = torch.tensor([total * r / world_size for r in range(1, world_size)], dtype=torch.float64)
        inner = torch.searchsorted(prefix, targets, right=False).tolist() if world_size > 1 else []
        cuts = [0] + [min(int(c) + 1, num_events) for c in inner] + [num_events]
        for i in range(1, len(cuts)):                       # keep the cuts monotone
            cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def halo_end(time: torch.Tensor, hi: int, delta) -> int:
    """First event id after ``hi - 1`` that can no longer continue any source event < ``hi``:
    events are time-sorted, so it is the first id with ``t > t[hi-1] + delta`` (evaluated like temporal.py:43).
    A conservative (never too small) bound is all correctness needs; the kernel re-checks every pair."""
    m = int(time.numel())
    if hi <= 0 or hi >= m:
        return m if hi > 0 else 0
    last = time[hi - 1].cpu()
    thr = last + torch.tensor(delta)
    cmp_dtype = torch.result_type(time, thr)
    # searchsorted on a tiny CPU copy would need the whole array; a bisection with O(log m) scalar reads is enough
    lo, up = hi, m
    thr_c = thr.to(cmp_dtype)
    while lo < up:
        mid = (lo + up) // 2
        if bool(time[mid].to(cmp_dtype).cpu() <= thr_c):
            lo = mid + 1
        else:
            up = mid
    return lo


def lift_order_temporal_sharded(g, delta=1, group=None, weights: torch.Tensor | None = None):
    """This rank's part of ``lift_order_temporal(g, delta)`` for a stream replicated on every rank.

    Returns ``(local_index [2, E_r] int64 with GLOBAL event ids, global_offset, global_total)``: rank ``r``'s block
    occupies columns ``[global_offset, global_offset + E_r)`` of the full lexicographic result."""
    rank, world = _world(group)
    data = g.data
    m = int(data.edge_index.size(1))
    lo, hi = event_ranges(m, world, weights)[rank]
    end = halo_end(data.time, hi, delta) if hi > lo else lo
    ei = _dispatch.plain(data.edge_index)[:, lo:end].contiguous()
    t = data.time[lo:end].contiguous()
    local = _dispatch.temporal_lift(ei, t, int(data.num_nodes), delta, n_own=hi - lo, id_offset=lo)
    counts = torch.tensor([local.size(1)], dtype=torch.int64, device=local.device)
    if world > 1:
        gathered = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts, group=group)
        sizes = [int(c.item()) for c in gathered]
    else:
        sizes = [int(counts.item())]
    return local, sum(sizes[:rank]), sum(sizes)


def gather_lifted(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate the per-rank blocks (rank order) on every rank — for tests and small graphs only."""
    rank, world = _world(group)
    if world == 1:
        return local
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.size(1)], dtype=torch.int64, device=local.device), group=group)
    cap = max(int(s.item()) for s in sizes)
    padded = torch.zeros((2, cap), dtype=torch.int64, device=local.device)
    padded[:, : local.size(1)] = local
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:, : int(s.item())] for p, s in zip(parts, sizes)], dim=1)


def all_reduce_gradients(module: torch.nn.Module, group=None) -> None:
    """Average the (small) DBGNN weight gradients across ranks in ONE flattened all-reduce (latency-bound:
    ~20 k floats; per-tensor calls would pay the xGMI launch latency 14 times)."""
    _, world = _world(group)
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= world
    at = 0
    for g in grads:
        g.copy_(flat[at: at + g.numel()].view_as(g))
        at += g.numel()
