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

import math

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
    # one searchsorted on the device and ONE read-back (this used to be a host bisection: ~log2(m) scalar reads per step and rank)
    delta_t = delta.to(time.device) if isinstance(delta, torch.Tensor) else torch.as_tensor(delta, device=time.device)
    thr = time[hi - 1] + delta_t                                     # promoted like the reference's comparison
    if thr.dtype != time.dtype and not time.dtype.is_floating_point:
        # a float threshold over integer times: float(t) <= thr admits no integer above ceil(thr) + one rounding step of the float type
        t64 = thr.double()
        step = t64.abs() * (2.0 ** -23 if thr.dtype == torch.float32 else 2.0 ** -52) + 1.0
        bound = torch.ceil(t64 + step).clamp(max=float(2 ** 62)).to(torch.int64)
    else:
        bound = thr.to(time.dtype)
    return hi + int(torch.searchsorted(time[hi:], bound, right=True).item())


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


def lift_order_edge_index_sharded(edge_index: torch.Tensor, num_nodes: int, group=None, weights: torch.Tensor | None = None):
    """Edge-range sharded line-graph lift of a source-sorted edge list replicated on every rank: rank r expands
    the edges ``[lo_r, hi_r)``; out-degrees and row pointers are derived locally from the replicated list, so the only collective
    is the all-gather of the per-rank pair counts.  Returns ``(local [2, E'_r] with global ids, ranges, total E')``; the rank-order
    concatenation of the blocks is the single-process result."""
    rank, world = _world(group)
    ei = _dispatch.plain(edge_index)
    ranges = event_ranges(ei.size(1), world, weights)
    lo, hi = ranges[rank]
    local = _dispatch.linegraph_lift(ei, num_nodes, (lo, hi))
    total = local.size(1)
    if world > 1:
        counts = torch.tensor([total], dtype=torch.int64, device=local.device)
        dist.all_reduce(counts, group=group)
        total = int(counts.item())
    return local, ranges, total


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


def all_reduce_gradients(module: torch.nn.Module, group=None, average: bool = True, comm: "Comm | None" = None,
                         inplace_views: bool = False) -> None:
    """Sum (``average=False``) or average the (small) DBGNN weight gradients across ranks in ONE flattened all-reduce
    (latency-bound: ~20 k floats; per-tensor calls would pay the xGMI launch latency 14 times).  ``comm``: count the bytes on (and take
    turns through) this :class:`Comm`.  The reduced values are copied back into the existing ``.grad`` tensors (references taken before the
    call — clipping lists, hooks — see them).  ``inplace_views=True`` skips that copy: every ``.grad`` is REBOUND to a view of the one flat
    buffer (earlier references keep the unreduced values, and any surviving view keeps the whole buffer alive) — for loops that only hand the
    gradients to the optimizer (bench.py)."""
    _, world = _world(group) if comm is None else (comm.rank, comm.world)
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if comm is not None:
        comm.all_reduce_(flat)
    else:
        dist.all_reduce(flat, group=group)
    if average:
        flat /= world
    at = 0
    for p in module.parameters():
        if p.grad is None:
            continue
        n_ = p.grad.numel()
        if inplace_views:                  # the reduced gradients stay where they are: .grad becomes a view of the flat buffer (no copies)
            p.grad = flat[at: at + n_].view_as(p.grad)
        else:
            p.grad.copy_(flat[at: at + n_].view_as(p.grad))
        at += n_


# =====================================================================================================
# Collectives of the partitioned path.  RCCL ("nccl") moves device buffers directly over xGMI; any other backend (gloo in the CPU
# tests, or gloo with several ranks sharing one GPU in the single-GPU hardware tests) is served by staging through host memory —
# test transport only, the driver's multi-GPU runs use RCCL.
# =====================================================================================================
class ThreadWorld:
    """R ranks as R THREADS of one process on ONE GPU (``bench.py --emulate-ranks``, tests): the transport of a :class:`Comm` whose collectives
    are device-to-device copies between the ranks' tensors.  The ranks take turns: a baton travels rank 0 -> 1 -> .. -> R-1 between
    collectives, so a rank computes from one collective to the next while the others wait, and the wall time of its turns (GPU drained at
    the end of each, copies of the collectives excluded) is the compute time it would need on a GPU of its own.  One process = one HIP
    context: turns cost no context switch (R processes sharing a GPU pay ~1 ms per switch — measured: 130 ms per step of pure overhead).
    Needs ``torch.autograd.set_multithreading_enabled(False)`` (backward functions then run on the calling thread; the engine's one
    device thread would otherwise block in the first rank's collective)."""

    def __init__(self, world: int, clock: str = "drain"):
        import threading
        if clock not in ("drain", "events"):
            raise ValueError("ThreadWorld: clock must be 'drain' or 'events'")
        self.world = world
        self.clock = clock                # how a rank's turns are timed, see Comm._clock_start
        self.cv = threading.Condition()
        self.baton = 0
        self.slots = [[None] * world, [None] * world]
        self.arrived = {}                 # collective sequence number -> ranks that deposited
        self.plain_barrier = threading.Barrier(world)
        self.failed = None

    def fail(self, exc):
        with self.cv:
            self.failed = exc
            self.cv.notify_all()
        self.plain_barrier.abort()

    def _wait(self, cond, limit_s: float = 300.0):
        import time as _time
        t0 = _time.monotonic()
        while not cond():
            if self.failed is not None:
                raise RuntimeError(f"another emulated rank failed: {self.failed!r}")
            if _time.monotonic() - t0 > limit_s:
                raise RuntimeError("ThreadWorld: a rank waited for its turn for more than 5 minutes (the ranks no longer call the same collectives?)")
            self.cv.wait(timeout=1.0)

    def take_turn(self, comm):
        """Block until this rank holds the baton (start of its first turn)."""
        with self.cv:
            self._wait(lambda: self.baton == comm.rank)
        comm._holding = True
        comm._clock_start()

    def collective(self, comm, payload, collect):
        """Deposit ``payload``, hand the baton on, wait until every rank has deposited and the baton is back, then ``collect(slots)``."""
        rank, world = comm.rank, self.world
        if not comm._holding:
            self.take_turn(comm)
        comm._clock_stop()
        seq = comm._seq
        comm._seq += 1
        with self.cv:
            self.slots[seq % 2][rank] = payload
            self.arrived[seq] = self.arrived.get(seq, 0) + 1
            self.baton = (rank + 1) % world
            comm._holding = False
            self.cv.notify_all()
            self._wait(lambda: self.arrived.get(seq, 0) == world and self.baton == rank)
            slots = list(self.slots[seq % 2])
        comm._holding = True
        out = collect(slots) if collect is not None else None
        if rank == world - 1:
            with self.cv:
                self.arrived.pop(seq - 1, None)
        if collect is not None and torch.cuda.is_available() and self.clock == "drain":
            torch.cuda.synchronize()          # the copies that stand in for the collective are queued asynchronously: drain them BEFORE the turn's clock
        comm._clock_start()                   # starts, or their GPU time is billed to this rank's compute (and priced again on the link model)
        return out                            # ("events" clock: the start event is queued behind the copies, nothing to drain)

    def release(self, comm):
        """End this rank's turn without a collective (before a plain barrier)."""
        if comm._holding:
            comm._clock_stop()
            with self.cv:
                self.baton = (comm.rank + 1) % self.world
                comm._holding = False
                self.cv.notify_all()


def run_thread_world(world: int, body, device=None, clock: str = "drain"):
    """Run ``body(comm) -> result`` on ``world`` emulated ranks (threads of this process, :class:`ThreadWorld`); returns the results in rank
    order.  An exception in one rank stops the others and is re-raised.  ``clock``: how the ranks' turns are timed (:meth:`Comm._clock_start`)."""
    import threading
    tw = ThreadWorld(world, clock)
    results, errors = [None] * world, [None] * world
    previous = torch.autograd.is_multithreading_enabled() if hasattr(torch.autograd, "is_multithreading_enabled") else True
    torch.autograd.set_multithreading_enabled(False)

    def run(rank):
        try:
            torch.autograd.set_multithreading_enabled(False)       # (thread-local, like the grad mode: every rank thread sets it for itself)
            if device is not None and torch.device(device).type == "cuda":
                torch.cuda.set_device(device)
            comm = Comm(thread_world=tw, rank=rank)
            results[rank] = body(comm)
            tw.release(comm)
        except BaseException as exc:          # noqa: BLE001 - handed to the caller
            errors[rank] = exc
            tw.fail(exc)

    threads = [threading.Thread(target=run, args=(r,), name=f"pp-rank-{r}") for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        torch.autograd.set_multithreading_enabled(previous)
    first = next((e for e in errors if e is not None and not isinstance(e, RuntimeError)), None) or next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results


class Comm:
    """Process-group handle with the collectives the partitioned lift + DBGNN need, byte counters per kind, and a per-collective log
    (``events``: kind, bytes to the busiest peer, overlapped or not) from which ``bench.py`` prices a step on xGMI.

    ``thread_world`` / ``rank``: this rank is a THREAD of an emulated world (:class:`ThreadWorld`; ``bench.py --emulate-ranks``): collectives
    are copies between the ranks' device tensors, the ranks take turns, ``compute_s`` accumulates the wall time of this rank's turns."""

    def __init__(self, group=None, thread_world: "ThreadWorld | None" = None, rank: int | None = None):
        self.group = group
        self.tw = thread_world
        if thread_world is not None:          # R ranks as threads of this process (ThreadWorld): device-to-device collectives, timed turns
            self.rank, self.world, self.backend, self.native = int(rank), thread_world.world, "threads", False
        else:
            self.rank, self.world = _world(group)
            self.backend = dist.get_backend(group) if (self.world > 1 or (dist.is_available() and dist.is_initialized())) else None
            self.native = self.backend == "nccl"
        self._holding = False
        self.sent_bytes = {"exchange": 0, "all_gather": 0, "reduce_scatter": 0, "all_reduce": 0}
        self.events = []               # (kind, bytes to / from the busiest peer, issued asynchronously)
        self.windows = []              # emulation: (index into events, lap at issue, lap at wait) of every asynchronous collective
        self._compute_s = 0.0
        self.host_s = 0.0              # emulation: host time of this rank's turns ("events" clock: CPU time of its thread; "drain": the drained wall time)
        self._turn_start = None
        self._seq = 0
        self.trace, self._mark_at = None, 0.0
        # "events" clock of the emulation: a turn is bracketed by two device events instead of a drained wall clock
        self._clock = thread_world.clock if (thread_world is not None and torch.cuda.is_available()) else "drain"
        self._intervals, self._open, self._resolved, self._marks = [], None, 0, []
        if self._clock == "events":
            self._mark_at = 0

    # ---- emulation turns.  Two clocks:
    #   "drain":  the GPU is drained at the end of every turn and the turn's wall time counts — every collective of a step (~45) costs the rank
    #             an empty queue and the launch latency behind it: an UPPER bound of what a process of its own would need;
    #   "events": a turn is bracketed by two events in the (shared, in-order) stream; nothing is drained, the ranks' data hand-offs are ordered by
    #             the stream itself.  A turn then costs its kernels plus whatever the GPU waited for this rank's host inside the turn — what a
    #             stream-ordered RCCL run pays: there the host blocks only where it reads a value back (one size read-back per step), not at the
    #             collectives.  While one rank's kernels run, the next rank's thread already enqueues (as a process of its own runs ahead of its
    #             own queue), so a rank whose host needs longer than its GPU is flattered: ``host_s`` (CPU time of the rank's thread in its turns) is reported next
    #             to it and the projection takes max(host, device) per rank.
    def _clock_start(self):
        import time as _time
        # "events": the thread's CPU time — what the rank's host side costs; its waits (for the baton, for a read-back behind the OTHER
        # ranks' queued kernels) are the emulation's, not the rank's
        self._turn_start = _time.thread_time() if self._clock == "events" else _time.perf_counter()
        if self._clock == "events":
            self._open = torch.cuda.Event(enable_timing=True)
            self._open.record()

    def _clock_stop(self):
        import time as _time
        if self._clock == "events":
            if self._turn_start is not None:
                self.host_s += _time.thread_time() - self._turn_start
                self._turn_start = None
            if self._open is not None:
                end = torch.cuda.Event(enable_timing=True)
                end.record()
                self._intervals.append((self._open, end))
                self._open = None
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self._turn_start is not None:
            dt = _time.perf_counter() - self._turn_start
            self._compute_s += dt
            self.host_s += dt
            self._turn_start = None

    @property
    def compute_s(self) -> float:
        """Emulation: compute time of this rank's finished turns ("events" clock: drains the device to read the events)."""
        if self._clock == "events" and self._resolved < len(self._intervals):
            torch.cuda.synchronize()
            for a, b in self._intervals[self._resolved:]:
                self._compute_s += a.elapsed_time(b) * 1e-3
            self._resolved = len(self._intervals)
        return self._compute_s

    def position(self, lap) -> float:
        """Compute seconds from the last :meth:`reset_counters` to a :meth:`lap` result."""
        return self.between(0, lap) if self._clock == "events" else float(lap)

    def between(self, a, b) -> float:
        """Compute seconds between two :meth:`lap` results."""
        if self._clock == "events":
            torch.cuda.synchronize()
            return sum(x.elapsed_time(y) for x, y in self._intervals[a:b]) * 1e-3
        return b - a

    def mark(self, label: str) -> None:
        """Emulation only (``trace`` enabled by bench.py --emulate-ranks --trace): compute time since the previous mark, under ``label``."""
        if self.trace is None or self.tw is None:
            return
        now = self.lap()
        if self._clock == "events":
            self._marks.append((label, self._mark_at, now))
        else:
            self.trace[label] = self.trace.get(label, 0.0) + (now - self._mark_at)
        self._mark_at = now

    def resolve_trace(self):
        """"events" clock: turn the recorded marks into ``trace`` (seconds per label)."""
        if self.trace is not None and self._marks:
            for label, a, b in self._marks:
                self.trace[label] = self.trace.get(label, 0.0) + self.between(a, b)
            self._marks = []
        return self.trace

    def lap(self):
        """Emulation: close the running turn's interval and open the next; returns a position for :meth:`between` ("drain" clock: the compute
        seconds so far, GPU drained; "events" clock: an index, nothing is drained)."""
        if self._turn_start is not None or self._open is not None:
            self._clock_stop()
            self._clock_start()
        return len(self._intervals) if self._clock == "events" else self._compute_s

    def barrier(self):
        """A collective without payload (step boundaries of the bench; closes / opens a turn in the emulation)."""
        if self.world == 1:
            return
        if self.tw is not None:
            self.tw.collective(self, None, None)
            return
        dist.barrier(group=self.group)

    def end_turns(self):
        """Emulation: close this rank's turn and pass the baton on without waiting for it again."""
        if self.tw is not None:
            self.tw.release(self)

    def reset_counters(self):
        self.sent_bytes = {k: 0 for k in self.sent_bytes}
        self.events = []
        self.windows = []
        self._compute_s, self.host_s = 0.0, 0.0
        reopen = self._open is not None
        self._intervals, self._open, self._resolved, self._marks = [], None, 0, []
        self._mark_at = 0 if self._clock == "events" else 0.0
        if reopen:
            self._clock_start()
        if self.trace is not None:
            self.trace = {}

    def _count_exchange(self, send_counts, row_bytes: int, recv_counts=None, overlapped: bool = False):
        self.sent_bytes["exchange"] += (int(sum(send_counts)) - int(send_counts[self.rank])) * row_bytes
        out_peer = max([int(c) for r, c in enumerate(send_counts) if r != self.rank] or [0])
        in_peer = max([int(c) for r, c in enumerate(recv_counts) if r != self.rank] or [0]) if recv_counts is not None else 0
        self.events.append(("exchange", max(out_peer, in_peer) * row_bytes, overlapped))

    def _peer_bytes(self, kind: str, block_bytes: int, overlapped: bool = False):
        self.events.append((kind, int(block_bytes), overlapped))

    def _stage(self, t: torch.Tensor) -> torch.Tensor:
        return t if (self.native or not t.is_cuda) else t.cpu()

    def exchange_counts(self, send_counts: list[int], device) -> list[int]:
        """counts[r] rows go to rank r -> how many rows come from each rank (one all-to-all of ``world`` integers)."""
        if self.world == 1:
            return list(send_counts)
        if self.tw is not None:
            self.events.append(("counts", 8, False))
            return self.tw.collective(self, [int(c) for c in send_counts], lambda slots: [slots[q][self.rank] for q in range(self.world)])
        dev = device if self.native else torch.device("cpu")
        send = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        out = [int(v) for v in recv.tolist()]
        self.events.append(("counts", 8, False))
        return out

    def exchange_rows(self, send: torch.Tensor, send_counts: list[int], recv_counts: list[int], out: torch.Tensor | None = None) -> torch.Tensor:
        """Variable all-to-all along dim 0: ``send`` holds the rows for rank 0, 1, .. back to back; returns (or fills ``out`` with) the
        rows received from rank 0, 1, .. back to back."""
        n_recv = int(sum(recv_counts))
        shape = (n_recv,) + tuple(send.shape[1:])
        if self.world == 1:
            if out is None:
                return send
            out.copy_(send)
            return out
        row_bytes = send.element_size() * math.prod(send.shape[1:])
        self._count_exchange(send_counts, row_bytes, recv_counts)
        if self.tw is not None:
            offs = [0]
            for c in send_counts:
                offs.append(offs[-1] + int(c))

            def collect(slots):
                pieces = [slots[q][0][slots[q][1][self.rank]: slots[q][1][self.rank + 1]] for q in range(self.world)]
                got = torch.cat(pieces) if pieces else send.new_empty(shape)
                if out is None:
                    return got
                out.copy_(got)
                return out
            return self.tw.collective(self, (send.detach().contiguous(), offs), collect)
        if self.native:
            if out is None:
                out = torch.empty(shape, dtype=send.dtype, device=send.device)
            dist.all_to_all_single(out, send.contiguous(), list(recv_counts), list(send_counts), group=self.group)
            return out
        staged = torch.empty(shape, dtype=send.dtype)
        dist.all_to_all_single(staged, send.detach().cpu().contiguous(), list(recv_counts), list(send_counts), group=self.group)
        if out is None:
            out = staged.to(send.device)
        else:
            out.copy_(staged)
        return out

    # ---- asynchronous forms: the collective is queued behind the work already on the current stream and runs beside what is launched next
    # (RCCL executes on its own stream; `wait()` makes the current stream wait for it).  Host-staged backends complete eagerly — the log
    # still marks them as overlapped: the schedule, not the transport, decides what can hide behind what.
    class _Done:
        """A collective that completed at issue (host-staged transports, the emulation).  In the emulation it remembers WHERE on the rank's
        compute clock it was issued and where it is waited for (``Comm.windows``): bench.py prices what a link-bound transfer would still
        have in flight at the wait."""

        def __init__(self, value, comm=None):
            self.value, self.comm = value, comm
            if comm is not None:
                self.idx, self.issued = len(comm.events) - 1, comm.lap()

        def wait(self):
            if self.comm is not None:
                self.comm.windows.append((self.idx, self.issued, self.comm.lap()))
                self.comm = None
            return self.value

    class _Pending:
        def __init__(self, work, value, keep=()):
            self.work, self.value, self.keep = work, value, keep

        def wait(self):
            self.work.wait()
            self.keep = ()
            return self.value

    def _mark_overlapped(self):
        if self.events:
            kind, nbytes, _ = self.events[-1]
            self.events[-1] = (kind, nbytes, True)

    def exchange_rows_async(self, send: torch.Tensor, send_counts: list[int], recv_counts: list[int], out: torch.Tensor | None = None):
        """:meth:`exchange_rows` as a handle whose ``wait()`` returns the received rows."""
        if not self.native or self.world == 1:
            value = self.exchange_rows(send, send_counts, recv_counts, out)
            if self.world > 1:
                self._mark_overlapped()
            return Comm._Done(value, self if (self.tw is not None and self.world > 1) else None)
        row_bytes = send.element_size() * math.prod(send.shape[1:])
        self._count_exchange(send_counts, row_bytes, recv_counts, overlapped=True)
        if out is None:
            out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        src = send.contiguous()
        work = dist.all_to_all_single(out, src, list(recv_counts), list(send_counts), group=self.group, async_op=True)
        return Comm._Pending(work, out, (src,))

    def reduce_scatter_rows_async(self, x_full: torch.Tensor, rows_per_rank: int):
        if not self.native or self.world == 1:
            value = self.reduce_scatter_rows(x_full, rows_per_rank)
            if self.world > 1:
                self._mark_overlapped()
            return Comm._Done(value, self if (self.tw is not None and self.world > 1) else None)
        block = rows_per_rank * x_full[0].numel() * x_full.element_size()
        self.sent_bytes["reduce_scatter"] += (self.world - 1) * block
        self._peer_bytes("reduce_scatter", block, overlapped=True)
        src = x_full.contiguous()
        out = torch.empty((rows_per_rank,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        work = dist.reduce_scatter_tensor(out, src, group=self.group, async_op=True)
        return Comm._Pending(work, out, (src,))

    def all_gather_rows_async(self, x_local: torch.Tensor):
        if not self.native or self.world == 1:
            value = self.all_gather_rows(x_local)
            if self.world > 1:
                self._mark_overlapped()
            return Comm._Done(value, self if (self.tw is not None and self.world > 1) else None)
        block = x_local.numel() * x_local.element_size()
        self.sent_bytes["all_gather"] += (self.world - 1) * block
        self._peer_bytes("all_gather", block, overlapped=True)
        src = x_local.contiguous()
        out = torch.empty((self.world * src.size(0),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        work = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        return Comm._Pending(work, out, (src,))

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """Equal-sized row blocks of all ranks, rank order: ``[world * rows, ...]``."""
        if self.world == 1:
            return x_local
        block = x_local.numel() * x_local.element_size()
        self.sent_bytes["all_gather"] += (self.world - 1) * block
        self._peer_bytes("all_gather", block)
        if self.tw is not None:
            return self.tw.collective(self, x_local.detach().contiguous(), lambda slots: torch.cat(slots))
        src = self._stage(x_local.contiguous())
        out = torch.empty((self.world * src.size(0),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=self.group)
        out = out.to(x_local.device)
        return out

    def reduce_scatter_rows(self, x_full: torch.Tensor, rows_per_rank: int) -> torch.Tensor:
        """Sum ``[world * rows_per_rank, ...]`` over the ranks, keep this rank's block."""
        if self.world == 1:
            return x_full
        block = rows_per_rank * x_full[0].numel() * x_full.element_size()
        self.sent_bytes["reduce_scatter"] += (self.world - 1) * block
        self._peer_bytes("reduce_scatter", block)
        if self.tw is not None:
            lo_, hi_ = self.rank * rows_per_rank, (self.rank + 1) * rows_per_rank

            def collect(slots):
                acc = slots[0][lo_:hi_].clone()
                for q in range(1, self.world):          # rank order: the same sum on every run
                    acc += slots[q][lo_:hi_]
                return acc
            return self.tw.collective(self, x_full.detach().contiguous(), collect)
        src = self._stage(x_full.contiguous())
        out = torch.empty((rows_per_rank,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.reduce_scatter_tensor(out, src, group=self.group)
        out = out.to(x_full.device)
        return out

    def all_reduce_(self, t: torch.Tensor, op=None) -> torch.Tensor:
        if self.world == 1:
            return t
        self.sent_bytes["all_reduce"] += t.numel() * t.element_size()
        self._peer_bytes("all_reduce", t.numel() * t.element_size())
        op = dist.ReduceOp.SUM if op is None else op
        if self.tw is not None:
            def collect(slots):
                acc = slots[0].clone()
                for q in range(1, self.world):
                    acc = torch.maximum(acc, slots[q]) if op == dist.ReduceOp.MAX else acc + slots[q]
                t.copy_(acc)
                return t
            return self.tw.collective(self, t.detach().clone(), collect)
        if self.native or not t.is_cuda:
            dist.all_reduce(t, op=op, group=self.group)
        else:
            staged = t.cpu()
            dist.all_reduce(staged, op=op, group=self.group)
            t.copy_(staged)
        return t

    def all_gather_ints_dev(self, mine: torch.Tensor) -> list[list[int]]:
        """Every rank's small int64 DEVICE vector (same length everywhere) -> ``[world][len]`` on the host: one collective + one read-back."""
        if self.world == 1:
            return [mine.tolist()]
        if self.tw is not None:
            self.events.append(("counts", 8 * mine.numel(), False))
            return self.tw.collective(self, mine.tolist(), lambda slots: [list(v) for v in slots])
        src = mine.to(torch.int64).contiguous()
        src = src if (self.native or not src.is_cuda) else src.cpu()
        out = torch.empty(self.world * src.numel(), dtype=torch.int64, device=src.device)
        dist.all_gather_into_tensor(out, src, group=self.group)
        host = out.view(self.world, -1).tolist()
        self.events.append(("counts", 8 * src.numel(), False))
        return host

    def all_gather_ints(self, values: list[int], device) -> list[list[int]]:
        """Every rank's small integer vector (same length everywhere): ``[world][len(values)]`` on the host."""
        if self.world == 1:
            return [list(values)]
        if self.tw is not None:
            self.events.append(("counts", 8 * len(values), False))
            return self.tw.collective(self, [int(v) for v in values], lambda slots: [list(v) for v in slots])
        dev = device if self.native else torch.device("cpu")
        mine = torch.tensor(values, dtype=torch.int64, device=dev)
        out = torch.empty(self.world * mine.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        host = out.view(self.world, -1).tolist()
        self.events.append(("counts", 8 * mine.numel(), False))
        return host


def node_ranges(num_nodes: int, world_size: int) -> list[tuple[int, int]]:
    return [((num_nodes * r) // world_size, (num_nodes * (r + 1)) // world_size) for r in range(world_size)]


def _even_cuts(num_nodes: int, world: int) -> list[int]:
    return [(num_nodes * r) // world for r in range(world + 1)]


def _debruijn2_wanted(m: int, n: int) -> bool:
    from ._hip import debruijn2_wanted
    return debruijn2_wanted(m, n)


def _hip_unit():
    from ._hip import UNIT
    return UNIT


def _ops_default(ops):
    if ops is not None:
        return ops
    from .nn.sharded import HipOps
    return HipOps()


# =====================================================================================================
# Graph shards (see pathpyg_amd.nn.sharded for the layout): halo discovery, the request exchange, the rectangular GCN plan with its
# one d^-1/2 halo exchange, and the CSR that folds returned gradient rows into the owned rows.
# =====================================================================================================
_DENSE_BOOK = {}


def _dense_halo_book(lo: int, hi: int, num_nodes: int, cuts: list[int], rank: int, dev):
    """Bookkeeping of a DENSE-halo shard (every foreign node is a halo row): halo ids, send list, counts, return CSR — pure arithmetic on the cuts,
    cached (the same tensors at every step of a stream)."""
    key = (lo, hi, num_nodes, tuple(cuts), rank, str(dev))
    book = _DENSE_BOOK.get(key)
    if book is None:
        world = len(cuts) - 1
        n_own = hi - lo
        halo_ids = torch.cat((torch.arange(0, lo, dtype=torch.int64, device=dev), torch.arange(hi, num_nodes, dtype=torch.int64, device=dev)))
        recv_counts = [0 if r == rank else int(cuts[r + 1] - cuts[r]) for r in range(world)]
        send_counts = [0 if r == rank else n_own for r in range(world)]
        peers = world - 1
        own_rows = torch.arange(n_own, dtype=torch.int32, device=dev)
        send_idx = own_rows.repeat(peers)                                                                   # every owned row to every peer
        # returned gradient rows arrive peer by peer: row i of peer block p sits at p * n_own + i
        back_ptr = torch.arange(0, n_own * peers + 1, max(peers, 1), dtype=torch.int32, device=dev)[: n_own + 1]
        back_idx = (own_rows.unsqueeze(1) + (torch.arange(peers, dtype=torch.int32, device=dev) * n_own).unsqueeze(0)).reshape(-1)
        if len(_DENSE_BOOK) > 64:
            _DENSE_BOOK.clear()
        book = _DENSE_BOOK[key] = (halo_ids, send_idx, send_counts, recv_counts, back_ptr, back_idx)
    return book


def sorted_halo(src: torch.Tensor, lo: int, hi: int, cuts_t: torch.Tensor):
    """Halo of an edge list whose global source ids ``src`` are ASCENDING (what the exchanges of :func:`build_dbgnn_shard` deliver): the
    distinct sources outside ``[lo, hi)`` are the run heads of the list — no flag array over all nodes, no sort.  Returns ``(src_local
    int64 [E] in the local source space [owned | halo], halo_ids int64 ascending, rows needed from every rank)``."""
    world = int(cuts_t.numel()) - 1
    e = int(src.numel())
    if e == 0:
        return src, src.new_empty(0), [0] * world
    foreign = (src < lo) | (src >= hi)
    head = foreign.clone()
    head[1:] &= src[1:] != src[:-1]
    slot = torch.cumsum(head, 0) - 1                                       # halo slot of every foreign edge
    src_local = torch.where(foreign, slot + (hi - lo), src - lo)
    halo_ids = src[head]                                                   # (one size read-back)
    bounds = torch.searchsorted(halo_ids, cuts_t.to(halo_ids.device)).tolist()
    return src_local, halo_ids, [int(bounds[r + 1] - bounds[r]) for r in range(world)]


def build_graph_shard(src: torch.Tensor, dst: torch.Tensor, weight: torch.Tensor | None, num_nodes: int, cuts: list[int], comm: Comm,
                      ops=None, row_sorted: bool = False, status_out: list | None = None, want_dst_order: bool = False,
                      edge_index: torch.Tensor | None = None, src_sorted: bool = False, dense_halo: bool = False):
    """This rank's :class:`~pathpyg_amd.nn.sharded.GraphShard` of a graph with ``num_nodes`` nodes from the edges (GLOBAL ids) that
    point into its destination range ``[cuts[rank], cuts[rank+1])`` — GCN normalisation with PyG ``gcn_norm`` semantics
    (reference nn/dbgnn.py:104-114 through GCNConv): every in-edge of an owned node is local, so the weighted in-degree is too;
    the d^-1/2 of the halo sources comes from their owners in one 4-byte-per-row exchange.  ``edge_index``: the same edges as one
    contiguous [2, E] tensor when the caller has it (saves a copy at world size 1); ``src_sorted``: ``src`` ascends (:func:`sorted_halo`).
    ``dense_halo`` (the SAME value on every rank): the halo is taken to be EVERY node of the other ranks — no discovery, no request round, no
    read-back: local ids, counts and the return CSR are arithmetic on the cuts.  For graphs whose ranks gather from nearly all nodes anyway
    (the first-order graph of a dense stream: 20 in-edges per node, 8 ranks -> 92 % of the nodes are sources of every rank)."""
    from .nn.sharded import GraphShard
    ops = _ops_default(ops)
    rank, world = comm.rank, comm.world
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    n_own = hi - lo
    dev = src.device
    if world == 1:
        whole = edge_index if edge_index is not None else torch.stack((src, dst))
        plan = ops.gcn_plan(whole, weight, num_nodes, row_sorted, status_out, want_dst_order=want_dst_order)
        return GraphShard(lo=0, hi=num_nodes, n_own=num_nodes, n_halo=0, n_src=num_nodes, num_nodes=num_nodes, cuts=list(cuts), plan=plan,
                          send_counts=[0], recv_counts=[0])
    if dense_halo:
        # local source space [owned | ids below lo | ids from hi on]: id + n_own below the range, id itself above it
        src_local = torch.where(src < lo, src + n_own, torch.where(src >= hi, src, src - lo))
        halo_ids, send_idx, send_counts, recv_counts, back_ptr, back_idx = _dense_halo_book(lo, hi, num_nodes, cuts, rank, dev)
        gs = _finish_graph_shard(torch.stack((src_local, dst - lo)), weight, lo, hi, num_nodes, cuts, halo_ids, send_idx, send_counts, recv_counts,
                                 comm, ops, status_out, unique_send=False, want_dst_order=want_dst_order, back=(back_ptr, back_idx))
        gs.dense = True              # (layer exchanges of this shard: all-gather + block copies, nn.sharded.halo_fill_async)
        return gs
    cuts_t = torch.tensor(cuts, dtype=torch.int64, device=dev)
    if src_sorted:
        src_local, halo_ids, recv_counts = sorted_halo(src, lo, hi, cuts_t)
    else:
        # halo = the distinct foreign sources, ascending (= grouped by owner: owners hold ascending id ranges)
        need = torch.zeros(num_nodes, dtype=torch.bool, device=dev)
        need[src] = True
        need[lo:hi] = False
        halo_ids = torch.nonzero(need).flatten()
        halo_rank = torch.cumsum(need, 0, dtype=torch.int32) - 1
        foreign = (src < lo) | (src >= hi)
        src_local = torch.where(foreign, halo_rank[src].to(torch.int64) + n_own, src - lo)
        del need, halo_rank, foreign
        bounds = torch.searchsorted(halo_ids, cuts_t).tolist()
        recv_counts = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
    send_counts = comm.exchange_counts(recv_counts, dev)
    requests = comm.exchange_rows(halo_ids, recv_counts, send_counts)             # the rows each peer wants from me (global ids)
    send_idx = (requests - lo).contiguous()        # (peers derive their requests from the same cuts: every id lies in [lo, hi))
    return _finish_graph_shard(torch.stack((src_local, dst - lo)), weight, lo, hi, num_nodes, cuts, halo_ids, send_idx, send_counts, recv_counts,
                               comm, ops, status_out, unique_send=False, want_dst_order=want_dst_order)


def _finish_graph_shard(ei_local: torch.Tensor, weight, lo: int, hi: int, num_nodes: int, cuts: list[int], halo_ids: torch.Tensor,
                        send_idx: torch.Tensor, send_counts: list[int], recv_counts: list[int], comm: Comm, ops, status_out,
                        unique_send: bool, want_dst_order: bool = False, back: tuple | None = None):
    """Rectangular GCN plan over the local source space ``[owned | halo]`` (one d^-1/2 exchange between its two phases) + the bookkeeping of
    the layer exchanges.  ``unique_send``: every owned row goes to at most one peer (De Bruijn layers): returned gradient rows are added in
    place, no CSR over the send list is needed."""
    from .nn.sharded import GraphShard
    n_own, n_halo = hi - lo, int(halo_ids.numel())
    dev = ei_local.device

    def halo_dinv(dinv_own: torch.Tensor) -> torch.Tensor:
        return comm.exchange_rows(dinv_own.index_select(0, send_idx), send_counts, recv_counts)

    plan = ops.gcn_plan_partition(ei_local, weight, n_own + n_halo, n_own, halo_dinv, row_sorted=False, status_out=status_out,
                                  want_dst_order=want_dst_order)
    back_ptr = back_idx = send_slot = None
    if unique_send:
        send_slot = torch.full((n_own,), -1, dtype=torch.int32, device=dev)
        if send_idx.numel():
            send_slot[send_idx] = torch.arange(int(send_idx.numel()), dtype=torch.int32, device=dev)
    if back is not None:
        back_ptr, back_idx = back
    elif not unique_send:
        back_ptr, back_idx = ops.group_rows(send_idx, n_own) if send_idx.numel() else (torch.zeros(n_own + 1, dtype=torch.int32, device=dev),
                                                                                       torch.zeros(0, dtype=torch.int32, device=dev))
    return GraphShard(lo=lo, hi=hi, n_own=n_own, n_halo=n_halo, n_src=n_own + n_halo, num_nodes=num_nodes, cuts=list(cuts), plan=plan,
                      halo_ids=halo_ids, send_idx=send_idx, send_counts=send_counts, recv_counts=recv_counts, back_ptr=back_ptr,
                      back_idx=back_idx, send_unique=unique_send, send_slot=send_slot)


def _bipartite_shard(ho_local: torch.Tensor, fo_global: torch.Tensor, n_ho_own: int, fo_cuts: list[int], comm: Comm, ops, src_sorted: bool):
    """Plan of the partial bipartite sums: sources = the owned higher-order rows (local ids), destinations = ALL first-order nodes in
    rank-major padded layout (node v of rank r's range sits at ``r * cap + v - fo_cuts[r]``) so that the ``[world * cap, H]`` partials
    reduce-scatter without a copy.  Returns ``(plan, cap)``."""
    world = comm.world
    cap = max(max(fo_cuts[r + 1] - fo_cuts[r] for r in range(world)), 1)
    if world == 1:
        padded = fo_global
    else:
        cuts_t = torch.tensor(fo_cuts, dtype=torch.int64, device=fo_global.device)
        owner = torch.searchsorted(cuts_t[1:].contiguous(), fo_global, right=True).clamp_(max=world - 1)
        padded = owner * cap + fo_global - cuts_t[owner]
    plan = ops.bipartite_plan(torch.stack((ho_local, padded)), n_ho_own, world * cap, None, src_sorted)
    return plan, cap


def shard_dbgnn_bundle(data, comm: Comm, ops=None, fo_cuts: list[int] | None = None, ho_cuts: list[int] | None = None):
    """Partition a REPLICATED ``MultiOrderModel.to_dbgnn_data`` bundle (reference multi_order_model.py:511-554) for this rank: the edges
    into its first-order / higher-order destination ranges, the bipartite pairs of its higher-order rows, its feature rows (owned +
    halo, taken from the replicated inputs: no exchange for the first layer) and labels.  ``fo_cuts`` / ``ho_cuts``: row cuts
    (default: equal node counts; for a De Bruijn layer cut the higher-order ids at first-order node boundaries to get the
    one-peer-per-row exchange)."""
    from .nn.sharded import DbgnnShard
    ops = _ops_default(ops)
    rank, world = comm.rank, comm.world
    n_fo, n_ho = int(data.num_nodes), int(data.num_ho_nodes)
    fo_cuts = _even_cuts(n_fo, world) if fo_cuts is None else [int(c) for c in fo_cuts]
    ho_cuts = _even_cuts(n_ho, world) if ho_cuts is None else [int(c) for c in ho_cuts]

    def in_edges(edge_index, weights, cuts):
        ei = _dispatch.plain(edge_index)
        if world == 1:
            return ei[0], ei[1], weights
        mine = torch.nonzero((ei[1] >= cuts[rank]) & (ei[1] < cuts[rank + 1])).flatten()
        return ei[0].index_select(0, mine), ei[1].index_select(0, mine), (None if weights is None else weights.index_select(0, mine))

    pending = []
    from .nn.dbgnn import _valid_hints
    hints = _valid_hints(data)             # honoured only while they still describe these very tensors; None -> checked on the device
    sorted_rows = True if (world == 1 and hints.get("rows_sorted")) else (None if world == 1 else False)
    fo = build_graph_shard(*in_edges(data.edge_index, data.edge_weights, fo_cuts), n_fo, fo_cuts, comm, ops, sorted_rows, pending)
    ho = build_graph_shard(*in_edges(data.edge_index_higher_order, data.edge_weights_higher_order, ho_cuts), n_ho, ho_cuts, comm, ops,
                           sorted_rows, pending)
    bi = _dispatch.plain(data.bipartite_edge_index)
    if world == 1:
        bi_ho, bi_fo = bi[0], bi[1]
    else:
        mine = (bi[0] >= ho_cuts[rank]) & (bi[0] < ho_cuts[rank + 1])
        bi_ho, bi_fo = bi[0][mine] - ho_cuts[rank], bi[1][mine]
    bip, cap = _bipartite_shard(bi_ho, bi_fo, ho.n_own, fo_cuts, comm, ops, src_sorted=False)
    ops.check_plan_status(pending)
    indeg = torch.bincount(bi[1], minlength=n_fo)[fo_cuts[rank]: fo_cuts[rank + 1]].to(torch.float32)
    x = data.x.index_select(0, fo.local_rows()) if world > 1 else data.x
    x_h = data.x_h.index_select(0, ho.local_rows()) if world > 1 else data.x_h
    y = None if data.y is None else data.y[fo_cuts[rank]: fo_cuts[rank + 1]]
    return DbgnnShard(fo=fo, ho=ho, bip=bip, cap=cap, indeg=indeg, x=x.contiguous(), x_h=x_h.contiguous(), y=y, n_fo=n_fo, n_ho=n_ho)


FUSED_BUILDER = True   # build_dbgnn_shard at world size 1: the node-by-node order-2 builder (pp_debruijn2_*, one read-back, no event graph) where it
#                        applies (no node with more than 64 in- / out-events, unit or float32 weights); False: always the generic kernels (A/B, tests)
FO_DENSE_HALO = None   # first-order shard of build_dbgnn_shard: None = dense halo (all foreign nodes, no discovery) when a rank's in-edges exceed
#                        1.5 x the node count (U2 >= 1.5 * world * N), True / False force a mode (tests, A/B)
ROW_COST = 2          # cost of a higher-order row relative to one of its in-edges when the cuts are balanced (a row is read and written once)


def _rows_of(source, rows: torch.Tensor | None, lo: int = 0, hi: int = 0, device=None):
    """Feature / label rows from a replicated tensor or from a ROW LOADER ``source(rows int64) -> tensor`` (a rank of a partitioned run only
    ever touches its owned + halo rows: a 10^8 x 256 feature matrix need not exist on any single GPU)."""
    if source is None:
        return None
    if callable(source):
        return source(rows if rows is not None else torch.arange(lo, hi, device=device))
    return source.index_select(0, rows) if rows is not None else source[lo:hi]


def _shard_rows(source, gs, comm: "Comm | None" = None) -> torch.Tensor:
    """Feature rows ``[owned | halo]`` of a graph shard.  A source with a ``shard_rows(lo, hi, halo_ids) -> [n_own + n_halo, F]`` method decides
    itself how to provide them — a RESIDENT row store keeps a rank's owned rows where they are and only fetches the halo rows (what a deployment
    does: the owned rows of the input features live on their rank; `bench.py` ``ResidentRows``); tensors and plain row loaders see
    ``gs.local_rows()``."""
    fetch = getattr(source, "shard_rows", None)
    if fetch is not None:
        static = getattr(source, "shard_rows_static", None)
        if getattr(gs, "dense", False) and static is not None:
            # dense halo = EVERY foreign node, whatever the graph of this step looks like: the halo rows of the (static) input features do not
            # change between steps — the store keeps them next to the owned rows (the first-order input features are replicated, 4 N F bytes)
            return static(gs.lo, gs.hi, gs.halo_ids)
        rows = fetch(gs.lo, gs.hi, gs.halo_ids)
        if comm is not None and comm.world > 1 and gs.n_halo:
            # a resident store serves the halo rows from the other ranks' memories: in a deployment that IS an exchange over xGMI (recv_counts
            # rows from every peer) — logged so that byte counters and the link-model pricing of bench.py --emulate-ranks see it
            comm._count_exchange(gs.send_counts, rows.element_size() * rows.size(1), gs.recv_counts)
        return rows
    return _rows_of(source, gs.local_rows())


def _window_ends(time: torch.Tensor, his: torch.Tensor, delta) -> torch.Tensor:
    """:func:`halo_end` for a vector of range ends ``his`` (device tensor), without a host read-back."""
    m = int(time.numel())
    idx = (his - 1).clamp(min=0, max=max(m - 1, 0))
    delta_t = delta.to(time.device) if isinstance(delta, torch.Tensor) else torch.as_tensor(delta, device=time.device)
    thr = time[idx] + delta_t
    if thr.dtype != time.dtype and not time.dtype.is_floating_point:
        t64 = thr.double()
        step = t64.abs() * (2.0 ** -23 if thr.dtype == torch.float32 else 2.0 ** -52) + 1.0
        bound = torch.ceil(t64 + step).clamp(max=float(2 ** 62)).to(torch.int64)
    else:
        bound = thr.to(time.dtype)
    ends = torch.searchsorted(time, bound.contiguous(), right=True)
    ends = torch.maximum(ends, his)
    return torch.where(his <= 0, torch.zeros_like(his), torch.where(his >= m, torch.full_like(his, m), ends))


def _balanced_cuts(weights: torch.Tensor, world: int) -> torch.Tensor:
    """Device int64 ``[world + 1]``: cut positions over ``len(weights)`` items so that every part carries about the same weight (exact integer
    arithmetic: every rank derives the SAME cuts from the same replicated input, no collective)."""
    n = int(weights.numel())
    prefix = torch.cumsum(weights.to(torch.int64), 0)
    total = prefix[-1] if n else torch.zeros((), dtype=torch.int64, device=weights.device)
    targets = (total * torch.arange(1, world, device=weights.device, dtype=torch.int64)) // world
    inner = torch.searchsorted(prefix, targets, right=True).clamp_(max=n) if n else torch.zeros(world - 1, dtype=torch.int64, device=weights.device)
    inner = torch.cummax(inner, 0).values if world > 1 else inner
    ends = torch.tensor([0, n], dtype=torch.int64, device=weights.device)
    return torch.cat((ends[:1], inner, ends[1:]))


def partition_plan(ei: torch.Tensor, time: torch.Tensor, n: int, delta, world: int, ops):
    """Everything the ranks must agree on BEFORE they split up, derived from the replicated stream with exact integer arithmetic (identical on
    every rank, no collective) and read back with ONE device-to-host copy:

    * ``fo_cuts``: first-order node ranges.  Rank r owns the nodes ``[fo_cuts[r], fo_cuts[r+1])``, the order-2 nodes ``(a, .)`` of its nodes a
      and runs layer 1 on the events that START in its range.  Balanced by estimated work: a node b costs ``ROW_COST`` per outgoing event
      (its rows ``(b, .)``) plus its expected in-edges ``in(b) * out(b) * delta / span`` (SURVEY §8e "balanced by nnz");
    * ``ev_cuts``: source-event ranges of the edge-range lift, balanced by the expected OUTPUT of an event (the out-degree of its head node;
      SURVEY §8e "balanced by output count") and ``ev_ends``: where each range's forward halo ends;
    * ``order`` / ``owner_ptr``: the event ids grouped by the owner of their START node (stable: ascending inside a group) — rank r's layer-1
      events are ``order[owner_ptr[r]:owner_ptr[r+1]]``, and every rank knows every other rank's list (needed to place the gathered
      event -> order-2 node maps)."""
    dev = ei.device
    m = int(ei.size(1))
    outdeg, indeg = ops.degree(ei[0], n).to(torch.int64), ops.degree(ei[1], n).to(torch.int64)
    if m > 0:
        span = (time[-1] - time[0]).to(torch.float64).clamp(min=1e-300)
        delta_t = (delta.to(dev) if isinstance(delta, torch.Tensor) else torch.as_tensor(delta, device=dev)).to(torch.float64)
        frac_q = torch.floor((delta_t / span).clamp(min=0.0, max=1.0) * 1024.0).to(torch.int64)      # window share of the stream, in 1/1024
    else:
        frac_q = torch.zeros((), dtype=torch.int64, device=dev)
    node_w = outdeg * (ROW_COST * 1024 + indeg * frac_q) + 1                                          # (+1: empty nodes still get spread)
    fo_cuts_t = _balanced_cuts(node_w, world)
    ev_w = outdeg.index_select(0, ei[1]) + 1 if m else torch.zeros(0, dtype=torch.int64, device=dev)
    ev_cuts_t = _balanced_cuts(ev_w, world)
    ev_ends_t = _window_ends(time, ev_cuts_t[1:], delta) if m else torch.zeros(world, dtype=torch.int64, device=dev)
    owner = torch.searchsorted(fo_cuts_t[1:-1].contiguous(), ei[0].contiguous(), right=True) if world > 1 else torch.zeros(m, dtype=torch.int64, device=dev)
    owner_ptr_t, order = ops.group_rows(owner, world)
    host = torch.cat((fo_cuts_t, ev_cuts_t, ev_ends_t, owner_ptr_t.to(torch.int64))).tolist()
    w1 = world + 1
    return {"fo_cuts": host[:w1], "ev_cuts": host[w1: 2 * w1], "ev_ends": host[2 * w1: 2 * w1 + world], "owner_ptr": host[2 * w1 + world:],
            "fo_cuts_t": fo_cuts_t, "order": order}


def _route(keys_owner: torch.Tensor, world: int, ops):
    """Stable grouping of items by destination rank: ``(ptr device int32 [world+1], order int64 [n])``."""
    ptr, order = ops.group_rows(keys_owner, world)
    return ptr, order.long()


def build_dbgnn_shard(g, delta, x, x_h, y, comm: Comm, ops=None, weight: str = "edge_weight", defer_status: bool = False):
    """The north-star split, straight from a time-sorted event stream replicated on every rank (what it replaces on one process:
    ``MultiOrderModel.from_temporal_graph(g, delta, max_order=2)`` + ``to_dbgnn_data`` + the plans of ``DBGNN.forward``; reference
    multi_order_model.py:124-192, 511-554, algorithms/temporal.py:17-54, lift_order.py:109-152).

    World size 1: the single-GPU kernels back to back (layer-1 coalesce and the lift share one size read-back).  World size R > 1
    (:func:`_build_partitioned`): every stage is sharded —

    0. :func:`partition_plan`: node ranges balanced by estimated nnz, event ranges balanced by expected output, no collective;
    1. LAYER 1 BY START-NODE RANGE: rank r coalesces the events that start in its node range — they yield exactly the order-2 nodes
       ``(a, .)`` it owns, in global lexicographic order.  Two all-gathers make the result global: the per-node block sizes (N ints: global
       ids, ``row_ptr``) and the event -> order-2 node map (m ints);
    2. EDGE-RANGE LIFT with a forward halo, no exchange (``pp_temporal_count/_fill`` with ``n_own`` / ``id_offset``);
    3. the lifted pairs (u, v) go to the OWNER OF THEIR DESTINATION v in one all-to-all (8 bytes per pair, + the weight when the stream is
       weighted) and are coalesced there: every rank ends up with exactly the in-edges of its order-2 rows.  Pairs arrive in (source
       rank, local) = global instance order, so merged weights are summed in the single-process order;
    4. the order-2 nodes ``(a, b)`` — which ARE the first-order edges — go to the owner of b in one all-to-all (16 bytes per node): the
       receiver gets the in-edges of its first-order rows and, by the De Bruijn property, exactly the higher-order rows its layers will
       gather from (``(a, b)`` feeds only rows ``(b, .)``): the higher-order halo needs no request round, every owned row has ONE consumer;
    5. graph shards (rectangular GCN plans with one d^-1/2 halo exchange each), the bipartite "last" plan, local feature rows.

    ``x`` [N, F] / ``x_h`` [U_2, F] / ``y`` [N]: replicated tensors, or ROW LOADERS ``f(rows int64) -> tensor`` (``x_h`` may also be a
    callable of the node count, the round-2 form, when it takes an ``int``) — a rank only ever reads its owned + halo rows.
    ``defer_status=True`` (world size 1, generic kernels): the higher-order plan's report (bad-index status, hub-row tables) is left to
    ``DbgnnShard.resolve()`` — which :class:`~pathpyg_amd.nn.sharded.ShardedDBGNN` calls after it has queued the first-order layers, so the
    read-back costs no idle GPU time; a caller that uses ``shard.ho.plan`` directly must call ``resolve()`` first.  Default: resolved here.
    Returns a :class:`~pathpyg_amd.nn.sharded.DbgnnShard`; ``shard.sizes`` reports the global layer sizes."""
    ops = _ops_default(ops)
    if comm.world > 1:
        if FUSED_BUILDER and getattr(ops, "debruijn2_part_count", None) is not None:
            shard = _build_partitioned_by_node(g, delta, x, x_h, y, comm, ops, weight)
            if shard is not None:
                return shard
        return _build_partitioned(g, delta, x, x_h, y, comm, ops, weight)
    from .nn.sharded import DbgnnShard
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    dev = ei.device
    n, m = int(data.num_nodes), int(ei.size(1))
    unit_weights = weight not in data
    fused = getattr(ops, "debruijn2", None) if FUSED_BUILDER else None
    if fused is not None and m > 0 and n > 0 and (unit_weights or data[weight].dtype == torch.float32) and (FUSED_BUILDER == "always" or _debruijn2_wanted(m, n)):
        built = fused(ei, data.time, n, delta, None if unit_weights else data[weight])
        if built is not None:              # (None: a hub node — the generic path below)
            from .nn.sharded import GraphShard
            n_ho = built.sizes["U2"]

            def whole(plan, nodes):
                return GraphShard(lo=0, hi=nodes, n_own=nodes, n_halo=0, n_src=nodes, num_nodes=nodes, cuts=[0, nodes], plan=plan, send_counts=[0],
                                  recv_counts=[0])
            xh_loc = x_h(n_ho) if (callable(x_h) and _takes_count(x_h)) else _rows_of(x_h, None, 0, n_ho, dev)
            return DbgnnShard(fo=whole(built.fo, n), ho=whole(built.ho, n_ho), bip=built.bip, cap=n, indeg=built.bip.self_coef,
                              x=_rows_of(x, None, 0, n, dev).contiguous(), x_h=xh_loc.contiguous(), y=_rows_of(y, None, 0, n, dev), n_fo=n, n_ho=n_ho,
                              sizes={"m": m, "N": n, "E2": built.sizes["E2"], "E2_local": built.sizes["E2"], "U2": n_ho, "A1": built.sizes["A1"],
                                     "A2": built.sizes["A2"], "A2_local": built.sizes["A2"], "fo_cuts": [0, n], "ho_cuts": [0, n_ho], "fo_halo": 0,
                                     "ho_halo": 0, "builder": "fused"})
    w = _hip_unit() if unit_weights else data[weight]                       # (unit weights: merged weight = run length, no ones vector, no gather)
    # layer 1 and the lift of the same stream are independent: their count phases are queued together, their sizes cost ONE read-back
    (fo, fo_w, inv1), local = ops.coalesce_and_lift((ei, w, n, "sum", None, True), (ei, data.time.contiguous(), n, delta, m, 0))
    n_ho = int(fo.size(1))
    fo_cuts, ho_cuts = [0, n], [0, n_ho]
    # the first-order graph needs nothing that follows: its plan kernels are queued NOW; the plan's own report (status, longest rows) is
    # the one read-back this stage needs — its longest source row IS the widest successor block of the order-2 id space
    pending_fo, pending_ho = [], []
    fo_shard = build_graph_shard(fo[0], fo[1], fo_w.to(torch.float32), n, fo_cuts, comm, ops, True, pending_fo, want_dst_order=True, edge_index=fo)
    row_ptr = fo_shard.plan.bwd_ptr.to(torch.int64)                          # [n+1]: order-2 nodes (a, .) = ids row_ptr[a] .. row_ptr[a+1]
    report = ops.check_plan_status(pending_fo)
    widest = report[0][2] if report else (int((row_ptr[1:] - row_ptr[:-1]).max().item()) if n > 0 else 1)
    col_block = (row_ptr[fo[1]], max(int(max(widest, 1) - 1).bit_length(), 1)) if n_ho else None     # successors of (a, b): the id block of b
    e2 = int(local.size(1))
    w_pairs = w if unit_weights else w.index_select(0, local[0])               # lifted weight = weight of the source event
    ho_ei, ho_w = ops.coalesce(local, w_pairs, n_ho, "sum", inv1, False, col_block)
    ho = build_graph_shard(ho_ei[0], ho_ei[1], ho_w.to(torch.float32), n_ho, ho_cuts, comm, ops, True, pending_ho, edge_index=ho_ei)
    bip = ops.bipartite_from_grouping(fo_shard.plan, fo[1], n_ho)
    indeg = bip.self_coef                                                       # order-2 nodes (., b) per first-order node b
    x_loc = _rows_of(x, None, 0, n, dev)
    xh_loc = x_h(n_ho) if (callable(x_h) and _takes_count(x_h)) else _rows_of(x_h, None, 0, n_ho, dev)
    a2 = int(ho_ei.size(1))
    # defer_status: the higher-order plan's report is read by DbgnnShard.resolve() — ShardedDBGNN queues the first-order layers in front of it
    shard = DbgnnShard(fo=fo_shard, ho=ho, bip=bip, cap=n, indeg=indeg, x=x_loc.contiguous(), x_h=xh_loc.contiguous(), y=_rows_of(y, None, 0, n, dev),
                      n_fo=n, n_ho=n_ho, pending=(ops, pending_ho),
                      sizes={"m": m, "N": n, "E2": e2, "E2_local": e2, "U2": n_ho, "A1": n_ho, "A2": a2, "A2_local": a2, "fo_cuts": fo_cuts,
                             "ho_cuts": ho_cuts, "fo_halo": 0, "ho_halo": 0})
    return shard if defer_status else shard.resolve()


def global_sizes(shard, comm: Comm) -> dict:
    """``shard.sizes`` with the global number of aggregated order-2 edges ``A2`` (a collective: every rank calls it; kept out of
    :func:`build_dbgnn_shard` so that a step pays no all-reduce for a number only reports need)."""
    sizes = dict(shard.sizes)
    if "A2" not in sizes:
        total = torch.tensor([sizes["A2_local"]], dtype=torch.float64, device=shard.x.device)
        comm.all_reduce_(total)
        sizes["A2"] = int(total.item())
    return sizes


def _takes_count(fn) -> bool:
    """Round-2 form of the ``x_h`` argument: ``x_h(num_ho_nodes) -> [U, F]`` (marked by the attribute ``takes_count``, or by a parameter
    named ``num_ho_nodes`` / ``n_ho``)."""
    if getattr(fn, "takes_count", False):
        return True
    try:
        import inspect
        names = list(inspect.signature(fn).parameters)
    except (TypeError, ValueError):
        return False
    return bool(names) and names[0] in ("num_ho_nodes", "n_ho", "num_nodes")


class StreamShard:
    """One rank's share of a time-sorted event stream under :func:`partition_plan` — what a multi-GPU deployment holds after it has
    DISTRIBUTED the stream (done once per stream, like the time sort of ``TemporalGraph.__init__``; every step then starts from resident,
    already distributed inputs): the events that start in its node range (layer 1), its edge range + forward halo (lift), and where the
    events of its lift slice sit in the all-gathered event -> order-2 node maps."""

    __slots__ = ("plan", "world", "rank", "n", "m", "ei_l1", "w_l1", "ei_lift", "time_lift", "w_lift", "n_own_lift", "slot", "slot_owner",
                 "cap_n", "cap_m", "stamp", "delta")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def distribute_stream(g, delta, comm: Comm, ops=None, weight: str = "edge_weight") -> StreamShard:
    """Partition plan + this rank's slices of the (replicated) stream ``g``.  Cached on ``g`` for as long as its tensors, ``delta`` and the
    world stay the same; :func:`build_dbgnn_shard` calls it on first use."""
    ops = _ops_default(ops)
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    w_all = data[weight] if weight in data else None
    stamp = (comm.world, comm.rank, repr(delta), tuple((t_, t_._version) for t_ in (data.edge_index, data.time, w_all) if t_ is not None))
    cache = getattr(g, "_pp_stream_shards", None)           # {(world, rank): shard} — emulated ranks share the graph object
    cached = cache.get((comm.world, comm.rank)) if isinstance(cache, dict) else None
    if cached is not None and len(cached.stamp) == len(stamp) and cached.stamp[:3] == stamp[:3] and len(cached.stamp[3]) == len(stamp[3]) and \
            all(a is b and va == vb for (a, va), (b, vb) in zip(cached.stamp[3], stamp[3])):
        return cached
    rank, world = comm.rank, comm.world
    dev = ei.device
    time = data.time.contiguous()
    n, m = int(data.num_nodes), int(ei.size(1))
    plan = partition_plan(ei, time, n, delta, world, ops)
    owner_ptr, order = plan["owner_ptr"], plan["order"].long()
    lo_e, hi_e = plan["ev_cuts"][rank], plan["ev_cuts"][rank + 1]
    end_e = max(plan["ev_ends"][rank], hi_e) if hi_e > lo_e else lo_e
    mine = order[owner_ptr[rank]: owner_ptr[rank + 1]]
    cap_n = max(max(plan["fo_cuts"][r + 1] - plan["fo_cuts"][r] for r in range(world)), 1)
    cap_m = max(max(owner_ptr[r + 1] - owner_ptr[r] for r in range(world)), 1)
    # where the events of my lift slice sit in the [world, cap_n + cap_m] buffer of all-gathered (block sizes | event -> local order-2 node) rows
    pos_of_event = torch.empty(m, dtype=torch.int64, device=dev)
    pos_of_event[order] = torch.arange(m, dtype=torch.int64, device=dev)
    k = pos_of_event[lo_e:end_e]
    ptr_t = torch.tensor(owner_ptr, dtype=torch.int64, device=dev)
    q = torch.searchsorted(ptr_t[1:-1].contiguous(), k.contiguous(), right=True) if world > 1 else torch.zeros_like(k)
    shard = StreamShard(plan=plan, world=world, rank=rank, n=n, m=m, ei_l1=ei.index_select(1, mine).contiguous(),
                        w_l1=None if w_all is None else w_all.index_select(0, mine).contiguous(), ei_lift=ei[:, lo_e:end_e].contiguous(),
                        time_lift=time[lo_e:end_e].contiguous(), w_lift=None if w_all is None else w_all[lo_e:end_e].contiguous(),
                        n_own_lift=hi_e - lo_e, slot=(q * (cap_n + cap_m) + cap_n + (k - ptr_t[q])).contiguous(), slot_owner=q.contiguous(), cap_n=cap_n, cap_m=cap_m,
                        stamp=stamp, delta=delta)
    try:
        if not isinstance(cache, dict):
            cache = {}
            object.__setattr__(g, "_pp_stream_shards", cache)
        cache[(comm.world, comm.rank)] = shard
    except Exception:          # exotic containers: no caching, still correct
        pass
    return shard


class NodeShard:
    """One rank's share of a time-sorted stream under a NODE-RANGE partition (round 4): the events that start or end in its node range, in
    stream order — what the node-by-node order-2 builder wants (pp_debruijn2_part_*).  Built once per stream (cached on the graph)."""

    __slots__ = ("world", "rank", "n", "m", "fo_cuts", "cuts_t", "ei", "time", "w", "stamp")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def distribute_stream_by_node(g, delta, comm: Comm, ops=None, weight: str = "edge_weight") -> NodeShard:
    """Node ranges balanced as in :func:`partition_plan` (estimated nnz per node; exact integer arithmetic on the replicated input, identical on
    every rank, no collective) + this rank's events: those with the tail OR the head node in its range.  Cached on ``g``."""
    ops = _ops_default(ops)
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    w_all = data[weight] if weight in data else None
    stamp = (comm.world, comm.rank, repr(delta), tuple((t_, t_._version) for t_ in (data.edge_index, data.time, w_all) if t_ is not None))
    cache = getattr(g, "_pp_node_shards", None)
    cached = cache.get((comm.world, comm.rank)) if isinstance(cache, dict) else None
    if cached is not None and len(cached.stamp) == len(stamp) and cached.stamp[:3] == stamp[:3] and len(cached.stamp[3]) == len(stamp[3]) and \
            all(a is b and va == vb for (a, va), (b, vb) in zip(cached.stamp[3], stamp[3])):
        return cached
    rank, world = comm.rank, comm.world
    dev = ei.device
    time = data.time.contiguous()
    n, m = int(data.num_nodes), int(ei.size(1))
    outdeg, indeg = ops.degree(ei[0], n).to(torch.int64), ops.degree(ei[1], n).to(torch.int64)
    if m > 0:
        span = (time[-1] - time[0]).to(torch.float64).clamp(min=1e-300)
        delta_t = (delta.to(dev) if isinstance(delta, torch.Tensor) else torch.as_tensor(delta, device=dev)).to(torch.float64)
        frac_q = torch.floor((delta_t / span).clamp(min=0.0, max=1.0) * 1024.0).to(torch.int64)
    else:
        frac_q = torch.zeros((), dtype=torch.int64, device=dev)
    # a node b costs its events on both sides (sorted, gathered) + its rows (b, .) + its expected in-edges in(b) * out(b) * delta / span
    node_w = (outdeg + indeg) * 1024 + outdeg * (ROW_COST * 1024 + indeg * frac_q) + 1
    cuts_t = _balanced_cuts(node_w, world)
    fo_cuts = cuts_t.tolist()
    lo, hi = fo_cuts[rank], fo_cuts[rank + 1]
    mine = torch.nonzero(((ei[0] >= lo) & (ei[0] < hi)) | ((ei[1] >= lo) & (ei[1] < hi))).flatten()
    shard = NodeShard(world=world, rank=rank, n=n, m=m, fo_cuts=fo_cuts, cuts_t=cuts_t.contiguous(), ei=ei.index_select(1, mine).contiguous(),
                      time=time.index_select(0, mine).contiguous(), w=None if w_all is None else w_all.index_select(0, mine).contiguous(), stamp=stamp)
    try:
        if not isinstance(cache, dict):
            cache = {}
            object.__setattr__(g, "_pp_node_shards", cache)
        cache[(comm.world, comm.rank)] = shard
    except Exception:
        pass
    return shard


def _own_rows_buffer(source, lo: int, hi: int, row_of: torch.Tensor, n_halo: int, device) -> torch.Tensor:
    """``[n_own + n_halo, F]``: the rows ``lo + row_of`` of a feature source in front (the owned rows of a partition shard in its LOCAL row
    order), room for the halo rows (which arrive by exchange) behind them.  Row loaders are asked for exactly those rows; matrices go through
    the row-gather kernel; a resident row store may offer ``rows_buffer(lo, hi, row_of, n_halo)`` itself."""
    fetch = getattr(source, "rows_buffer", None)
    if fetch is not None:
        return fetch(lo, hi, row_of, n_halo)
    n_own = hi - lo
    if callable(source):
        own = source(row_of.to(torch.int64) + lo)
        buf = torch.empty((n_own + n_halo, own.size(1)), dtype=own.dtype, device=own.device)
        buf[:n_own] = own
        return buf
    part = source[lo:hi]
    buf = torch.empty((n_own + n_halo, part.size(1)), dtype=part.dtype, device=part.device)
    if n_own:
        if part.is_cuda and part.dtype == torch.float32 and part.size(1) % 4 == 0:
            from . import _hip
            _hip.gather_rows(part.contiguous(), row_of, out=buf[:n_own])
        else:
            torch.index_select(part, 0, row_of.to(torch.int64), out=buf[:n_own])
    return buf


def _build_partitioned_by_node(g, delta, x, x_h, y, comm: Comm, ops, weight: str):
    """World size > 1, round 4: NODE-RANGE partition on the node-by-node order-2 builder.  Rank r owns the first-order nodes of its range, is
    handed the events that touch them (:func:`distribute_stream_by_node`, once per stream) and builds — with one read-back and without any
    exchange of pairs, node ids or counts — the order-2 rows (b, .) of its nodes b: their in-edges come from in-events x out-events of b, all
    local.  What crosses links per step: 2 integers per rank (sizes), the d^-1/2 degrees and the input feature rows of the halo rows, the
    first-order degrees (inside pp_gcn_plan_begin / _finish).  Returns ``None`` when some rank holds a node with more than 64 in- / out-events
    (every rank then takes :func:`_build_partitioned`, the generic kernels)."""
    from .nn.sharded import DbgnnShard, GraphShard
    rank, world = comm.rank, comm.world
    ns = distribute_stream_by_node(g, delta, comm, ops, weight)
    comm.mark("build: 0 stream distribution (first step only)")
    n, fo_cuts = ns.n, ns.fo_cuts
    lo_n, hi_n = fo_cuts[rank], fo_cuts[rank + 1]
    dev = ns.ei.device
    if ns.w is not None and ns.w.dtype != torch.float32:
        return None
    cap_n = max(max(fo_cuts[r + 1] - fo_cuts[r] for r in range(world)), 1)          # rows per rank of the padded first-order layout
    c = ops.debruijn2_part_count(ns.ei, ns.time, n, lo_n, hi_n, ns.cuts_t, rank, delta, ns.w, cap_n)
    comm.mark("build: 1 order-2 builder, count pass (sorts, successor blocks, halo numbering, send lists)")
    # sizes of all ranks: global order-2 id ranges, E2, and whether everybody can stay on this path (one tiny collective)
    sizes_all = comm.all_gather_ints([c.u2 if c is not None else 0, c.e2 if c is not None else 0, 1 if (c is None or c.status & 4) else 0,
                                      (c.status & 3) if c is not None else 0], dev)
    # bad input seen by ANY rank is raised by EVERY rank, after the collective (a rank holds only its own events)
    if any(row[3] & 1 for row in sizes_all):
        raise IndexError("MultiOrderModel.from_temporal_graph (partition): node index out of range")
    if any(row[3] & 2 for row in sizes_all):
        raise ValueError("lift_order_temporal: the events are not sorted by time")
    if any(row[2] for row in sizes_all):
        return None
    ho_cuts = [0]
    for row in sizes_all:
        ho_cuts.append(ho_cuts[-1] + row[0])
    n_ho, e2 = ho_cuts[-1], sum(row[1] for row in sizes_all)
    lo_h, hi_h = ho_cuts[rank], ho_cuts[rank + 1]
    n_own, n_halo = c.u2, c.n_halo
    # input features of the halo rows travel under the rest of the build; the halo rows' degrees are needed by the fill pass
    if callable(x_h) and _takes_count(x_h):
        x_h = x_h(n_ho)
    xh_buf = _own_rows_buffer(x_h, lo_h, hi_h, c.row_of, n_halo, dev)          # (owned rows in the shard's local order = send order: no pack below)
    xh_pending = comm.exchange_rows_async(xh_buf[: c.n_send], c.send_counts, c.recv_counts, out=xh_buf[n_own: n_own + n_halo])
    comm.exchange_rows(c.ho_deg[: c.n_send], c.send_counts, c.recv_counts, out=c.ho_deg[n_own: n_own + n_halo])
    comm.mark("build: 2 halo exchanges (degrees; feature rows asynchronously)")
    # first-order degrees of ALL nodes (the count pass filled my slice): one all-gather of N floats
    fo_deg = c.bufs["fo_deg"]
    mine = fo_deg[lo_n:hi_n]
    gathered = comm.all_gather_rows(mine if hi_n - lo_n == cap_n else torch.nn.functional.pad(mine, (0, cap_n - (hi_n - lo_n))))
    for r in range(world):
        if r != rank and fo_cuts[r + 1] > fo_cuts[r]:
            fo_deg[fo_cuts[r]: fo_cuts[r + 1]] = gathered[r * cap_n: r * cap_n + fo_cuts[r + 1] - fo_cuts[r]]
    ho_plan, fo_plan, indeg = ops.debruijn2_part_fill(c)
    comm.mark("build: 3 order-2 builder, fill pass (both plans)")
    own_ids = c.row_of.to(torch.int64) + lo_h                                   # global ids of the owned rows, local order

    def fetch_halo_ids():
        return comm.exchange_rows(own_ids[: c.n_send].contiguous(), c.send_counts, c.recv_counts)

    ho = GraphShard(lo=lo_h, hi=hi_h, n_own=n_own, n_halo=n_halo, n_src=n_own + n_halo, num_nodes=n_ho, cuts=ho_cuts, plan=ho_plan, halo_ids=None,
                    send_idx=None, send_counts=c.send_counts, recv_counts=c.recv_counts, send_unique=True, send_slot=c.send_slot,
                    halo_fetch=fetch_halo_ids, send_prefix=c.n_send, own_ids=own_ids)
    pending = []
    # first-order shard: straight from the builder too (dense halo: every foreign node is a source row; layer exchanges = all-gathers)
    halo_ids, fo_send_idx, fo_send_counts, fo_recv_counts, back_ptr, back_idx = _dense_halo_book(lo_n, hi_n, n, fo_cuts, rank, dev)
    fo_shard = GraphShard(lo=lo_n, hi=hi_n, n_own=hi_n - lo_n, n_halo=n - (hi_n - lo_n), n_src=n, num_nodes=n, cuts=list(fo_cuts), plan=fo_plan,
                          halo_ids=halo_ids, send_idx=fo_send_idx, send_counts=fo_send_counts, recv_counts=fo_recv_counts, back_ptr=back_ptr,
                          back_idx=back_idx, send_unique=False, send_slot=None, dense=True)
    comm.mark("build: first-order shard")
    bip, cap = getattr(c, "bip", None), cap_n                  # (the builder's count pass wrote the bipartite plan: the local row order groups the rows by successor)
    if bip is None:
        bip, cap = _bipartite_shard(torch.arange(n_own, dtype=torch.int64, device=dev), c.succ.to(torch.int64), n_own, fo_cuts, comm, ops, src_sorted=True)
    ops.check_plan_status(pending)
    comm.mark("build: bipartite plan")
    x_loc = _shard_rows(x, fo_shard, comm)
    xh_pending.wait()
    comm.mark("build: feature rows (owned + halo)")
    return DbgnnShard(fo=fo_shard, ho=ho, bip=bip, cap=cap, indeg=indeg, x=x_loc.contiguous(), x_h=xh_buf[: n_own + n_halo],
                      y=_rows_of(y, None, lo_n, hi_n, dev), n_fo=n, n_ho=n_ho,
                      sizes={"m": ns.m, "N": n, "E2": e2, "E2_local": c.e2, "U2": n_ho, "A1": n_ho, "A2_local": c.a2, "fo_cuts": fo_cuts,
                             "ho_cuts": ho_cuts, "fo_halo": fo_shard.n_halo, "ho_halo": n_halo, "events_local": int(ns.ei.size(1)), "builder": "fused"})


def _build_partitioned(g, delta, x, x_h, y, comm: Comm, ops, weight: str):
    """World size > 1 branch of :func:`build_dbgnn_shard` (see there for the scheme)."""
    from .nn.sharded import DbgnnShard
    rank, world = comm.rank, comm.world
    ss = distribute_stream(g, delta, comm, ops, weight)              # (cached: the stream is distributed once, not once per step)
    comm.mark("build: 0 stream distribution (first step only)")
    plan = ss.plan
    n, m = ss.n, ss.m
    dev = ss.ei_l1.device
    unit_weights = ss.w_l1 is None
    i64 = dict(dtype=torch.int64, device=dev)
    fo_cuts, owner_ptr, fo_cuts_t = plan["fo_cuts"], plan["owner_ptr"], plan["fo_cuts_t"]
    lo_n, hi_n = fo_cuts[rank], fo_cuts[rank + 1]
    n_fo_own = hi_n - lo_n
    m_l1 = int(ss.ei_l1.size(1))
    # ---- 1. layer 1 on the events that start in my node range  +  2. the edge-range lift (count phases queued together: one read-back)
    w_r = _hip_unit() if unit_weights else ss.w_l1
    (fo_r, fo_w_r, inv_r), local = ops.coalesce_and_lift((ss.ei_l1, w_r, n, "sum", None, True),
                                                         (ss.ei_lift, ss.time_lift, n, delta, ss.n_own_lift, 0))       # (ids relative to my slice)
    n_ho_own = int(fo_r.size(1))
    comm.mark("build: 1+2 layer-1 coalesce + lift")
    # global ids: per-node block sizes of all ranks (N ints over the wire) -> row_ptr; event -> order-2 node map of all ranks (m ints)
    ptr_r = ops.ptr_from_sorted(fo_r[0] - lo_n, n_fo_own)                                       # int64 [n_fo_own + 1], local
    # (one all-gather for both: [block sizes | event -> local node map] per rank)
    both = torch.zeros(ss.cap_n + ss.cap_m, dtype=torch.int32, device=dev)
    both[:n_fo_own] = ptr_r[1:] - ptr_r[:-1]
    both[ss.cap_n: ss.cap_n + m_l1] = inv_r
    gathered = comm.all_gather_rows(both).view(world, ss.cap_n + ss.cap_m)
    blocks_all = gathered[:, : ss.cap_n]
    inv_all = gathered.reshape(-1)                                                                # rank q's map starts at q * (cap_n + cap_m) + cap_n
    own_blocks = blocks_all.sum(dim=1, dtype=torch.int64)                                       # order-2 nodes per rank
    ho_cuts_t = torch.zeros(world + 1, **i64)
    torch.cumsum(own_blocks, 0, out=ho_cuts_t[1:])
    e2_local = int(local.size(1))
    ho_cuts = ho_cuts_t.tolist()                                                                # (read-back: global order-2 id ranges)
    n_ho = ho_cuts[-1]
    lo_h, hi_h = ho_cuts[rank], ho_cuts[rank + 1]
    assert hi_h - lo_h == n_ho_own, "partition_plan: the ranks disagree on the order-2 node ranges"
    inv_slice = inv_all.index_select(0, ss.slot).to(torch.int64) + ho_cuts_t.index_select(0, ss.slot_owner)     # event of my lift slice -> global node id
    del inv_all
    comm.mark("build: global ids (2 all-gathers, event->node map of my slice)")
    # ---- 3. lifted pairs to the owner of their destination  +  4. order-2 nodes (= first-order edges) to the owner of their head node:
    #         both send-count vectors travel in ONE all-gather, one read-back
    u, v = inv_slice.index_select(0, local[0]), inv_slice.index_select(0, local[1])
    p_ptr, p_order = _route(torch.searchsorted(ho_cuts_t[1:-1].contiguous(), v, right=True), world, ops)
    f_ptr, f_order = _route(torch.searchsorted(fo_cuts_t[1:-1].contiguous(), fo_r[1].contiguous(), right=True), world, ops)
    counts = comm.all_gather_ints_dev(torch.cat((p_ptr[1:] - p_ptr[:-1], f_ptr[1:] - f_ptr[:-1], torch.tensor([e2_local], dtype=torch.int32, device=dev)))
                                      .to(torch.int64))                                                                  # [world][2 * world + 1]
    e2 = sum(counts[r][2 * world] for r in range(world))
    p_send, f_send = counts[rank][:world], counts[rank][world: 2 * world]
    p_recv, f_recv = [counts[r][rank] for r in range(world)], [counts[r][world + rank] for r in range(world)]
    pairs = torch.stack((u, v), dim=1).to(torch.int32).index_select(0, p_order)
    pairs_in = comm.exchange_rows(pairs, p_send, p_recv)
    if unit_weights:
        w_in = _hip_unit()
    else:
        w_in = comm.exchange_rows(ss.w_lift.index_select(0, local[0]).index_select(0, p_order), p_send, p_recv)     # weight of the source event
    own_ids = torch.arange(lo_h, hi_h, **i64)
    nodes = torch.stack((fo_r[0], fo_r[1], own_ids, fo_w_r.to(torch.float32).view(torch.int32).to(torch.int64)), dim=1).to(torch.int32)
    nodes_out = nodes.index_select(0, f_order)
    nodes_in = comm.exchange_rows(nodes_out, f_send, f_recv)                                     # (a, b, global id, weight bits), sorted by id
    del u, v, pairs, nodes
    comm.mark("build: 3+4 routing (pairs, first-order edges)")
    # ---- layer 2: the in-edges of my order-2 rows
    ho_ei, ho_w = ops.coalesce(pairs_in.t().to(torch.int64).contiguous(), w_in, n_ho, "sum", None, False, None)
    comm.mark("build: layer-2 coalesce")
    # ---- 5. shards.  Higher-order graph: the halo is structural — the order-2 nodes (a, b) with b in my node range that other ranks own
    pending = []
    f_off, s_off = [0], [0]
    for c in f_recv:
        f_off.append(f_off[-1] + c)
    for c in f_send:
        s_off.append(s_off[-1] + c)
    ids_in = nodes_in[:, 2]
    ho_halo_ids = torch.cat((ids_in[: f_off[rank]], ids_in[f_off[rank + 1]:])).to(torch.int64)     # ascending (senders hold ascending id ranges)
    ho_send_idx = torch.cat((f_order[: s_off[rank]], f_order[s_off[rank + 1]:])).contiguous()     # my rows, grouped by the rank that gathers them
    ho_send = [0 if r == rank else f_send[r] for r in range(world)]
    ho_recv = [0 if r == rank else f_recv[r] for r in range(world)]
    src = ho_ei[0]
    own = (src >= lo_h) & (src < hi_h)
    src_local = torch.where(own, src - lo_h, n_ho_own + torch.searchsorted(ho_halo_ids, src.contiguous())) if ho_halo_ids.numel() else src - lo_h
    ho = _finish_graph_shard(torch.stack((src_local, ho_ei[1] - lo_h)), ho_w.to(torch.float32), lo_h, hi_h, n_ho, ho_cuts, ho_halo_ids, ho_send_idx,
                             ho_send, ho_recv, comm, ops, pending, unique_send=True)
    comm.mark("build: higher-order shard + plan")
    # first-order graph: my in-edges (a -> b, b in my range) arrived sorted by a; its halo = the distinct foreign a (request round)
    dense_fo = FO_DENSE_HALO if FO_DENSE_HALO is not None else n_ho >= 1.5 * world * n       # (global quantities: every rank decides alike)
    fo_shard = build_graph_shard(nodes_in[:, 0].to(torch.int64), nodes_in[:, 1].to(torch.int64), nodes_in[:, 3].contiguous().view(torch.float32), n, fo_cuts,
                                 comm, ops, False, pending, src_sorted=True, dense_halo=bool(dense_fo))
    comm.mark("build: first-order shard + plan")
    bip, cap = _bipartite_shard(torch.arange(n_ho_own, **i64), fo_r[1], n_ho_own, fo_cuts, comm, ops, src_sorted=True)
    ops.check_plan_status(pending)
    fptr = fo_shard.plan.fwd_ptr
    indeg = (fptr[1:] - fptr[:-1]).to(torch.float32)                            # order-2 nodes (., b) per owned first-order node b
    comm.mark("build: bipartite plan + status read-back")
    x_loc = _shard_rows(x, fo_shard, comm)
    if callable(x_h) and _takes_count(x_h):
        x_h = x_h(n_ho)
    xh_loc = _shard_rows(x_h, ho, comm)
    comm.mark("build: feature rows (owned + halo)")
    return DbgnnShard(fo=fo_shard, ho=ho, bip=bip, cap=cap, indeg=indeg, x=x_loc.contiguous(), x_h=xh_loc.contiguous(),
                      y=_rows_of(y, None, lo_n, hi_n, dev), n_fo=n, n_ho=n_ho,
                      sizes={"m": m, "N": n, "E2": e2, "E2_local": e2_local, "U2": n_ho, "A1": n_ho, "A2_local": int(ho_ei.size(1)),
                             "fo_cuts": fo_cuts, "ho_cuts": ho_cuts, "ev_cuts": plan["ev_cuts"], "fo_halo": fo_shard.n_halo, "ho_halo": ho.n_halo,
                             "lift_events_local": int(ss.ei_lift.size(1)), "layer1_events_local": m_l1})


from .nn.sharded import ShardedDBGNN  # noqa: E402,F401  (historic import location)

# =====================================================================================================
# Distributed De Bruijn aggregation: global lexicographic unique / coalesce = a distributed sort with ONE
# exchange step per layer.  Keys are range-partitioned (rank r owns the keys in [cut_r, cut_{r+1})), every rank sends each key
# (+ weight) to its owner, the owner coalesces its range with the single-GPU kernels, and an all-gather of the per-rank counts
# turns local ranks into global ids.  Concatenating the ranks' outputs in rank order IS the single-process result.
# =====================================================================================================
def _exchange(buckets: list[torch.Tensor], group=None) -> torch.Tensor:
    """Send ``buckets[r]`` to rank r, return the concatenation of what the other ranks sent here (rank order).
    RCCL: one all_to_all of the sizes + one of the payload; gloo (tests) has no all_to_all: all-gather and pick."""
    rank, world = _world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return buckets[0]
    if dist.get_backend(group) == "nccl":
        send_sizes = torch.tensor([b.size(0) for b in buckets], dtype=torch.int64, device=buckets[0].device)
        recv_sizes = torch.empty_like(send_sizes)
        dist.all_to_all_single(recv_sizes, send_sizes, group=group)
        recv = list(torch.empty((int(recv_sizes.sum()),) + tuple(buckets[0].shape[1:]), dtype=buckets[0].dtype,
                                device=buckets[0].device).split(recv_sizes.tolist()))
        dist.all_to_all(recv, [b.contiguous() for b in buckets], group=group)
        return torch.cat(recv, dim=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, [b.cpu() for b in buckets], group=group)
    return torch.cat([gathered[src][rank] for src in range(world)], dim=0).to(buckets[0].device)


def _owner_cuts(num_keys_space: int, world: int) -> torch.Tensor:
    """Equal-width key ranges over ``[0, num_keys_space)``: ``cuts[r] .. cuts[r+1]`` is owned by rank r."""
    return torch.tensor([(num_keys_space * r) // world for r in range(world + 1)], dtype=torch.int64)


def coalesce_sharded(rows: torch.Tensor, cols: torch.Tensor, weight: torch.Tensor, num_nodes: int, group=None):
    """Global ``coalesce`` of edges scattered over the ranks.  Rank r ends up with the distinct edges whose ROW lies in
    ``[cuts[r], cuts[r+1])`` (sorted by (row, col), weights summed); returns ``(edge_index [2, A_r], weight [A_r], cuts)``."""
    rank, world = _world(group)
    cuts = _owner_cuts(num_nodes, world).to(rows.device)
    owner = torch.searchsorted(cuts[1:].contiguous(), rows, right=True).clamp_(max=world - 1)
    # ids and weights travel separately: the weights keep their dtype (integer weights stay exact beyond 2^24)
    order = torch.sort(owner, stable=True).indices
    counts = torch.bincount(owner, minlength=world).tolist()
    ids = torch.stack((rows, cols), dim=1).index_select(0, order)
    wsorted = weight.index_select(0, order)
    mine = _exchange(list(ids.split(counts)), group)
    w = _exchange(list(wsorted.split(counts)), group)
    ei = mine.t().contiguous()
    merged_index, merged_weight = _dispatch.coalesce(ei, w.contiguous(), num_nodes, "sum")
    return merged_index, merged_weight, cuts.cpu()


def unique_pairs_sharded(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, group=None):
    """Global lexicographic numbering of the distinct (src, dst) pairs of events scattered over the ranks (= the order-2
    De Bruijn nodes).  Returns ``(all_pairs [U, 2] on every rank, ids)`` with ``ids[e]`` the global id of local event e."""
    rank, world = _world(group)
    ones = torch.ones(src.numel(), dtype=torch.float32, device=src.device)
    local_pairs, _, _ = coalesce_sharded(src, dst, ones, num_nodes, group)          # my key range, sorted, distinct
    counts = torch.tensor([local_pairs.size(1)], dtype=torch.int64, device=src.device)
    if world > 1:
        sizes = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(sizes, counts, group=group)
        cap = max(int(s.item()) for s in sizes)
        padded = torch.zeros((2, cap), dtype=torch.int64, device=src.device)
        padded[:, : local_pairs.size(1)] = local_pairs
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        all_pairs = torch.cat([p[:, : int(s.item())] for p, s in zip(parts, sizes)], dim=1)
    else:
        all_pairs = local_pairs
    keys = all_pairs[0] * num_nodes + all_pairs[1]                                   # ascending: ranks own ascending row ranges
    ids = torch.searchsorted(keys.contiguous(), (src * num_nodes + dst).contiguous())
    return all_pairs.t().contiguous(), ids


def second_order_layer_sharded(g, delta=1, group=None, edge_weight: torch.Tensor | None = None) -> dict:
    """Order-2 De Bruijn layer of a temporal graph replicated on every rank, computed cooperatively: edge-range lift (no
    exchange), global pair numbering (one exchange), global coalesce of the lifted pairs (one exchange).
    Rank r returns its slice of the layer: the aggregated edges whose source node id lies in its row range."""
    rank, world = _world(group)
    data = g.data
    ei = _dispatch.plain(data.edge_index)
    n = int(data.num_nodes)
    w = edge_weight if edge_weight is not None else torch.ones(ei.size(1), device=ei.device)
    local, _, total = lift_order_temporal_sharded(g, delta, group)                 # (i, j) with global event ids
    lo, hi = event_ranges(ei.size(1), world)[rank]
    pairs, own_ids = unique_pairs_sharded(ei[0, lo:hi], ei[1, lo:hi], n, group)     # every rank contributes its own events
    num_ho = pairs.size(0)
    # node id of EVERY event the local pairs refer to (own events and halo): look the pair up in the global list
    pair_keys = (pairs[:, 0] * n + pairs[:, 1]).contiguous()
    i, j = local[0], local[1]
    u = torch.searchsorted(pair_keys, (ei[0][i] * n + ei[1][i]).contiguous())
    v = torch.searchsorted(pair_keys, (ei[0][j] * n + ei[1][j]).contiguous())
    edges, weights, cuts = coalesce_sharded(u, v, w[i], num_ho, group)
    return {"edge_index": edges, "edge_weight": weights, "node_sequence": pairs, "num_nodes": num_ho, "row_cuts": cuts,
            "instance_pairs_total": total, "own_event_ids": own_ids}
