"""Temporal event-graph lift, HIP-backed (reference ``pathpyG.algorithms.temporal``)."""
from __future__ import annotations

import torch

from .. import _dispatch


def lift_order_temporal(g, delta: float | int = 1) -> torch.Tensor:
    """Lift a temporal graph to its second-order event graph (reference src/pathpyG/algorithms/temporal.py:17-54).

    Returns the int64 ``[2, E2]`` index of all event pairs ``(i, j)`` with ``dst(i) == src(j)`` and
    ``t_i < t_j <= t_i + delta``, in lexicographic order, on the device of ``g.data.edge_index``.
    ``delta`` is interpreted exactly like ``torch.tensor(delta)`` in the reference: an ``int`` compares
    in int64, a Python ``float`` against int64 timestamps compares in float32, ``np.float64`` in float64.

    Differences from the reference, both deliberate: the per-timestamp Python loop is replaced by a
    sort/search/scan/fill pipeline of HIP kernels, and an event list without any admissible pair yields an
    empty ``[2, 0]`` tensor instead of the reference's ``torch.cat([])`` error (temporal.py:53).
    Supported timestamp dtypes: integer types (widened to int64) and float64.
    """
    data = g.data
    return _dispatch.temporal_lift(data.edge_index, data.time, int(data.num_nodes), delta)
