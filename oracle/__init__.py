"""CPU oracle for the pathpyG hot path (k-th order De Bruijn lift + DBGNN).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a CPU (torch-CPU / numpy) restatement of the algorithms the reference
implements in
    src/pathpyG/algorithms/lift_order.py:10-152
    src/pathpyG/algorithms/temporal.py:17-54
    src/pathpyG/core/multi_order_model.py:83-241,511-554
    src/pathpyG/core/graph.py:79-119
    src/pathpyG/utils/dbgnn.py:33-44
    src/pathpyG/nn/dbgnn.py:39-151
plus the torch_geometric 2.7.0 semantics those call into (``degree``,
``cumsum``, ``coalesce``, ``GCNConv``/``gcn_norm``, ``MessagePassing("add")``),
which are NOT vendored in the reference tree (pinned by its ``uv.lock``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported baseline.  Nothing under
``pathpyg_amd/`` imports it; the product path raises if the HIP library or a
GPU is missing instead of falling back to this code.

Pinning status
--------------
* integer path (line-graph lift, weighted lift, node-attribute aggregation,
  temporal event-graph lift): PINNED.  ``tests/golden/make_golden.py`` runs the
  reference's own function source (AST-loaded from /root/reference in the build
  container, where ``import pathpyG`` is impossible because torch_geometric is
  absent) on seeded inputs and stores inputs+outputs under ``tests/golden/``;
  ``tests/test_oracle_golden.py`` checks this oracle against them bit-for-bit
  and against every known-answer value of the reference's unit tests
  (tests/algorithms/test_lift_order.py, tests/algorithms/test_temporal.py,
  tests/core/test_multi_order_model.py, tests/nn/test_dbgnn.py).
* De Bruijn aggregation (``torch.unique(dim=0)`` + PyG ``coalesce``): pinned by
  the reference's known-answer tests and by running real ``torch.unique``.
* DBGNN numerics: PARITY UNPINNED by the reference (its only test asserts
  ``out is not None``).  The oracle follows the published GCN formula and is
  cross-checked against a dense-matrix evaluation in tests/test_oracle_dbgnn.py.
"""
