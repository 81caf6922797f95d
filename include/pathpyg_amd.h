/* pathpyg_amd — C ABI of the MI355X-native (gfx950) higher-order graph engine.
 *
 * This is the drop-in boundary for ONE hot path of pathpy/pathpyG: the k-th order De Bruijn lift
 * of temporal edge streams / walk data and the DBGNN forward/backward that consumes it.  The
 * reference has no FFI seam of its own (it is pure Python on torch + torch_geometric), so every
 * entry point below names the reference function (file:line, relative to the pathpyG repository
 * root) whose tensor work it replaces; the Python shims in pathpyg_amd/ keep the reference's
 * signatures and call these through ctypes.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers, element counts, a hipStream_t passed as void*.
 *   - every function returns PP_OK (0) or a negative PP_ERR_* code; pp_last_error() gives the text.
 *   - index tensors at the boundary are int64 row-major [2,E] exactly like the reference;
 *     internal ids are 32-bit, so element counts per call must stay below 2^31 (PP_ERR_TOO_LARGE).
 *   - all memory is owned by the caller (PyTorch's caching allocator in the Python shims).
 *     Scratch comes from a caller-provided workspace sized by the matching *_ws_bytes() query.
 *   - data-dependent output sizes use two calls: *_count leaves its state in the workspace and
 *     writes the size to a device scalar; the caller reads it (one 8-byte D2H copy), allocates the
 *     output and calls *_fill with the SAME workspace.
 *   - everything is asynchronous on `stream`; nothing here synchronises the device.
 */
#ifndef PATHPYG_AMD_H
#define PATHPYG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_VERSION 100 /* 0.1.0 */

typedef void* pp_stream_t; /* hipStream_t */

enum pp_status { PP_OK = 0, PP_ERR_HIP = -1, PP_ERR_ARG = -2, PP_ERR_WORKSPACE = -3, PP_ERR_TOO_LARGE = -4 };

/* element types of caller tensors */
enum pp_dtype { PP_I32 = 0, PP_I64 = 1, PP_F32 = 2, PP_F64 = 3 };

/* per-edge attribute from the two endpoints: reference aggregate_node_attributes,
 * src/pathpyG/algorithms/lift_order.py:33-44 */
enum pp_edge_aggr { PP_AGGR_SRC = 0, PP_AGGR_DST = 1, PP_AGGR_MAX = 2, PP_AGGR_MUL = 3, PP_AGGR_ADD = 4 };

/* duplicate-edge reduction: reference aggregate_edge_index(aggr=...), lift_order.py:139-144 (PyG coalesce) */
enum pp_reduce { PP_REDUCE_SUM = 0, PP_REDUCE_MEAN = 1, PP_REDUCE_MIN = 2, PP_REDUCE_MAX = 3 };

/* how `delta` reached torch.tensor(delta) in the reference (src/pathpyG/algorithms/temporal.py:30,43):
 * the threshold t+delta and the <= comparison are evaluated in torch's promoted dtype. */
enum pp_delta_kind { PP_DELTA_I64 = 0, PP_DELTA_F32 = 1, PP_DELTA_F64 = 2 };

int pp_version(void);
const char* pp_last_error(void);
/* Concurrency between two HIP streams of one process (no reference counterpart: the reference runs nn/dbgnn.py:130-145 — the first-order
 * stack, then the higher-order stack — as one serial chain of torch ops).  The fused layer kernels launch PERSISTENT grids sized to the
 * workgroups that are resident at once, which leaves no slot for a kernel of another stream.  pp_set_launch_share(s) lets the persistent
 * launches of the CALLING THREAD take only s per mille of those slots (1..1000, default 1000) and returns the previous value. */
int pp_set_launch_share(int per_mille);

/* ------------------------------------------------------------------ primitives (pp_scan.hip, pp_sort.hip) */

/* torch_geometric.utils.cumsum as used at lift_order.py:74,77: out[0..n] = exclusive prefix sums, out[n] = total */
size_t pp_scan_ws_bytes(int64_t n);
int pp_exclusive_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_exclusive_scan_i64(const int64_t* in, int64_t n, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream);

/* torch_geometric.utils.degree(index, num_nodes) as used at lift_order.py:65: bins[v] = #occurrences of v */
int pp_degree_i64(const int64_t* index, int64_t n, int64_t num_bins, int32_t* bins, pp_stream_t stream);

/* out2 = {min, max} of a (INT64_MAX, INT64_MIN when n == 0) */
int pp_minmax_i64(const int64_t* a, int64_t n, int64_t* out2, pp_stream_t stream);

/* stable LSD radix sort of (key, 32-bit value) pairs on key bits [begin_bit, end_bit);
 * vals_in == NULL means values 0..n-1 (argsort).  Replaces torch.argsort / index_sort / torch.unique's sort
 * (temporal_graph.py:58, lift_order.py:133,139, graph.py:103,115).  in/out buffers must differ. */
size_t pp_sort_ws_bytes(int64_t n, int key_bytes);
int pp_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
                      int begin_bit, int end_bit, void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_sort_pairs_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                      int begin_bit, int end_bit, void* ws, size_t ws_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------ fused order-2 De Bruijn builder (pp_debruijn.hip) */

/* MultiOrderModel.from_temporal_graph(g, delta, max_order=2) (src/pathpyG/core/multi_order_model.py:124-192: lift_order_temporal,
 * algorithms/temporal.py:17-54, + aggregate_edge_index for layers 1 and 2, algorithms/lift_order.py:109-152) together with what
 * DBGNN.forward derives from the two layers on every call (gcn_norm through GCNConv, src/pathpyG/nn/dbgnn.py:104-114,130-140) and the
 * bipartite "last" index (utils/dbgnn.py:10-46) — node by node, without materialising the event graph, in three calls:
 *   pp_debruijn2_lists  the two sorts of the events (out-lists, in-lists), hub classification -> 16 int64 statistics copied to `host_stats`
 *                       (pinned host memory; the copy is asynchronous and followed by the out-side kernel of the ordinary nodes, so that
 *                       pp_debruijn2_wait — the ONE call of this library that blocks — returns while the GPU is still busy);
 *   pp_debruijn2_count  everything up to the sizes (the first five int64 of ws).  `host_result` (pinned int64 [154], may be NULL): the whole result
 *                       header is copied there asynchronously, the first size-independent kernel of the fill pass is queued behind the copy,
 *                       pp_debruijn2_wait returns when the copy has landed (NULL: the caller reads the header from ws itself);
 *   pp_debruijn2_fill   both plans (`rows_packed` = 1 when the count call was given a host_result).
 * Same inputs as pp_temporal_count (time-sorted events, delta as torch.tensor(delta) sees it); weight: NULL (every event weighs 1: the
 * reference's default torch.ones) or float32 [m].  Results are identical, array by array, to pp_coalesce_* (layer 1) -> pp_temporal_* ->
 * pp_coalesce_* (layer 2) -> pp_gcn_plan x 2 on the same stream (bit for bit where the partial sums are exact: see "hub nodes" in
 * pp_debruijn.hip for streams with non-integer weights on nodes of more than 64 events).
 * HUB NODES: a node with more than 64 in- or more than 64 out-events is worked on by several waves (chunks of 256 in-events) instead of
 * one; host_stats = {hub nodes, (out-hubs << 32) | their out-events, tasks, part columns, -, longest rows (4), longest in-list, longest
 * out-list, ...} sizes the extra workspace: pp_debruijn2_hub_ws_bytes(out-hub events, out-hubs, part columns); the same five numbers go to
 * count and fill (all 0 and hub_ws NULL when there is no hub).  After count the hub block (ws int64 [138 .. 154)) also holds [4] the lifted
 * pairs through hub nodes (E2 = ws[3] + that) and [5..8] the longest order-2 destination- / source-major and first-order destination- /
 * source-major row among the hubs' rows (the caller's chunked pre-pass for rows beyond 512 entries, pp_spmm_heavy_f32).
 * Order-2 node u = first-order edge u (lexicographic (src, dst) order).  Outputs, all int32 / float32:
 *   count phase (capacities in brackets; U2 = order-2 nodes, A2 = order-2 edges, A1 = U2 = first-order edges):
 *     fo_bwd_ptr [N+1], fo_bwd_idx [m], fo_w [m]     source-major first-order graph: successors of every node + merged weights (first U2 entries)
 *     fo_fwd_ptr [N+1]                                destination-major row pointers of the first-order graph
 *     ho_fwd_ptr [m+1], ho_bwd_ptr [m+1]              row pointers of the order-2 graph, destination- / source-major (first U2+1 entries; constant after)
 *     ho_deg [m], fo_deg [N]                          weighted in-degree incl. the self loop of gcn_norm (add_remaining_self_loops, fill 1)
 *   the first five int64 of ws = {U2, status, A2, E2 (without the hubs' pairs), A1}; status bit 0: node index outside [0, N); bit 1: time
 *     not ascending; bit 2: (partition shards only) a node has more than 64 in- or out-events.
 *   fill phase: ho_fwd_idx/val [A2] (in-edges of every order-2 node, ascending source), ho_bwd_idx/val [A2], ho_self [U2];
 *     fo_fwd_idx/val [A1], fo_dst_order [A1] (edge id = order-2 node id of every in-edge: the bipartite "last" grouping),
 *     fo_bwd_val [U2], fo_self [N].  Values are the normalised coefficients d_src^-1/2 w d_dst^-1/2 (0 on self-loop entries).
 *     num_ho_edges = A2; pair_scratch: 8 * A2 bytes of device scratch (the source-major rows are scattered as 8-byte pairs, then split).
 *     ho_fwd_w [A2] (optional, may be NULL): the merged weights THEMSELVES (lift_order.py:139, coalesce "sum") in destination-major order —
 *     what MultiOrderModel.layers[2].data.edge_weight is derived from when a caller reads it. */
size_t pp_debruijn2_ws_bytes(int64_t m, int64_t num_nodes);
size_t pp_debruijn2_hub_ws_bytes(int64_t hub_out_events, int64_t out_hubs, int64_t hub_parts);
int pp_debruijn2_lists(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, const float* weight, void* ws,
                       size_t ws_bytes, int64_t* host_stats, pp_stream_t stream);
int pp_debruijn2_wait(void);
int pp_debruijn2_count(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                       int32_t* fo_bwd_ptr, int32_t* fo_bwd_idx, float* fo_w, int32_t* fo_fwd_ptr, int32_t* ho_fwd_ptr, int32_t* ho_bwd_ptr,
                       float* ho_deg, float* fo_deg, void* ws, size_t ws_bytes, int64_t hub_nodes, int64_t out_hubs, int64_t hub_out_events,
                       int64_t hub_tasks, int64_t hub_parts, void* hub_ws, size_t hub_ws_bytes, int64_t* host_result, pp_stream_t stream);
int pp_debruijn2_fill(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                      const int32_t* fo_bwd_ptr, const int32_t* fo_bwd_idx, const float* fo_w, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                      const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t num_ho_edges, int32_t* ho_fwd_idx, float* ho_fwd_val,
                      int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val, int32_t* fo_dst_order,
                      float* fo_bwd_val, float* fo_self, float* ho_fwd_w, void* pair_scratch, void* ws, size_t ws_bytes, int64_t hub_nodes,
                      int64_t out_hubs, int64_t hub_out_events, int64_t hub_tasks, int64_t hub_parts, void* hub_ws, size_t hub_ws_bytes,
                      int rows_packed, pp_stream_t stream);
/* pp_debruijn2_wait + pp_debruijn2_fill in ONE call, for a caller that allocated the A2-sized outputs ahead of the size read-back with a guessed
 * capacity (ho_edge_capacity entries; pair_scratch 8 * capacity bytes): waits for the header pp_debruijn2_count copied to host_result, and launches
 * the fill at once when the stream is good (status 0) and A2 fits — *launched = 1 — so that nothing of the caller's host code sits between the
 * read-back and the fill; otherwise *launched = 0 and nothing was queued (the caller reads the header, allocates exactly and calls
 * pp_debruijn2_fill, or reports the status). */
int pp_debruijn2_fill_ready(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                            const int32_t* fo_bwd_ptr, const int32_t* fo_bwd_idx, const float* fo_w, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                            const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t ho_edge_capacity, int32_t* ho_fwd_idx,
                            float* ho_fwd_val, int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val,
                            int32_t* fo_dst_order, float* fo_bwd_val, float* fo_self, float* ho_fwd_w, void* pair_scratch, void* ws, size_t ws_bytes,
                            int64_t hub_nodes, int64_t out_hubs, int64_t hub_out_events, int64_t hub_tasks, int64_t hub_parts, void* hub_ws,
                            size_t hub_ws_bytes, int rows_packed, const int64_t* host_result, int64_t* launched, pp_stream_t stream);

/* The same builder on ONE RANK of a node-range partition (the lift shards by edge range, the DBGNN by destination-node
 * partition; no reference counterpart — the reference is single-process).  Rank `rank` owns the first-order nodes
 * [node_lo, node_lo + n_own) = cuts[rank] .. cuts[rank + 1] (cuts: device int64 [world + 1]) and is handed the time-sorted events that start
 * or end in that range (node ids stay global).  Every order-2 edge (a,b) -> (b,c) is born on the owner of its MIDDLE node b, which owns the
 * rows (b, .): no lifted pair and no order-2 node id ever crosses a link.  Sources (a, b) with a foreign a are halo rows, numbered behind
 * the U2 owned rows in (owner of a, b, a) order.  The owned rows are numbered in SEND ORDER: the rows (b, c) other ranks gather from come
 * first, grouped by the owner of c and ordered by (c, b) — the receiver's halo order —, so the send list of every layer exchange is the
 * contiguous prefix [0, rows sent) of a row matrix (no pack, no ids, no request round: one all-to-all of rows fills the peers' halos); rows
 * nobody gathers from follow.  fo_bwd_idx / fo_w are in that local order (fo_bwd_ptr stays lexicographic: block sizes per node);
 * send_slot [m]: local row -> its position in the prefix, -1 behind it; row_of [m]: local row -> lexicographic row (global id - first owned id).
 * In that order ALL rows are grouped by their successor c, so the bipartite "last" plan (utils/dbgnn.py:10-46) of the shard needs no sort either:
 * destinations = all first-order nodes in the rank-major padded layout (node c of rank r at r * pad_rows + c - cuts[r]; pad_rows >= the largest
 * range), bip_fwd_ptr [world * pad_rows + 1] / bip_fwd_idx [m] (local rows per destination), bip_bwd_ptr [m + 1] / bip_bwd_idx [m] (one
 * destination per row), bip_self [world * pad_rows] (in-degrees) — all written by the count phase.  The first 8 int64 of ws = {U2, status, A2, E2, A1, halo rows, rows sent, -}, then recv_ptr [world + 1] and
 * send_ptr [world + 1] (rows from / to every rank, as offsets).  Between count and fill the caller fetches ho_deg of its halo rows
 * (ho_deg[U2 ..]) from their owners and all-gathers fo_deg (the count pass fills the owned entries of the [num_nodes] array).  Fill: the
 * order-2 plan over the local source space [owned | halo] as in pp_debruijn2_fill, and the rank's FIRST-ORDER SHARD with a dense halo — local
 * source space [owned | ids below node_lo | ids from node_lo + n_own on] (num_nodes rows): destination-major rows fo_fwd_ptr / fo_fwd_idx /
 * fo_fwd_val / fo_self of the owned nodes, source-major rows fo_shard_bwd_ptr [num_nodes + 1] (count phase) / _idx / _val [A1] — the successor
 * runs of ALL nodes that fall into the owned range (the out-lists of foreign nodes hold exactly their events into it). */
int pp_debruijn2_part_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, int64_t node_lo, int64_t n_own,
                            const int64_t* cuts, int world, int rank, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                            int32_t* fo_bwd_ptr, int32_t* fo_bwd_idx, float* fo_w, int32_t* fo_fwd_ptr, int32_t* ho_fwd_ptr, int32_t* ho_bwd_ptr,
                            float* ho_deg, float* fo_deg, int32_t* send_slot, int32_t* row_of, int32_t* fo_shard_bwd_ptr, int64_t pad_rows,
                            int32_t* bip_fwd_ptr, int32_t* bip_fwd_idx, int32_t* bip_bwd_ptr, int32_t* bip_bwd_idx, float* bip_self, void* ws,
                            size_t ws_bytes, pp_stream_t stream);
int pp_debruijn2_part_fill(int time_dtype, int64_t m, int64_t num_nodes, int64_t node_lo, int64_t n_own, int delta_kind, int64_t delta_i, double delta_f,
                           const float* weight, const int32_t* fo_bwd_ptr, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                           const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t num_ho_edges, int32_t* ho_fwd_idx,
                           float* ho_fwd_val, int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val,
                           float* fo_self, const int32_t* fo_shard_bwd_ptr, int32_t* fo_shard_bwd_idx, float* fo_shard_bwd_val, void* pair_scratch,
                           void* ws, size_t ws_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------ all orders of a temporal stream, level by level (pp_multiorder.hip)
 *
 * MultiOrderModel.from_temporal_graph(g, delta, max_order >= 3), src/pathpyG/core/multi_order_model.py:124-192: lift_order_temporal
 * (algorithms/temporal.py:17-54) followed, per order, by iterate_lift_order (multi_order_model.py:83-122) = lift_order_edge_index(_weighted,
 * aggr="src") (algorithms/lift_order.py:48-106) + the node-sequence extension (:114) + aggregate_edge_index (lift_order.py:109-152).
 * The layers come out as source-major CSR (int32 row pointers / columns, float32 merged weights) — the reference's edge_index is
 * (row of every entry, column), its edge_weight the weights; neither the instance graphs [2, E_k] nor the [E_k, k] node sequences exist.
 *
 * A LEVEL k holds the order-(k+1) instances (time-respecting paths of k events) grouped by type (= edge of layer k = node of layer k+1),
 * types in lexicographic order:  tptr [types + 1] instance range of every type;  ibase [types + 1] first child of every type's instances
 * (ibase[types] = instances of level k+1);  col [types] column of the type in layer k;  tlast [types] its last node;
 * inst [instances] 16-byte records {window first, window count | head bit, last node, float32 weight of the first event}.
 *
 * pp_multiorder_prepare: level 1 from a finished pp_temporal_count or pp_temporal_windows (same stream, same delta; `lift_ws` is its workspace).  All outputs
 *   have capacity m (tptr / ibase: m + 1, rowptr: num_nodes + 1); `tab` [m] 16-byte records is the continuation table every step reads.
 *   Layer 1 = (rowptr, tlast as columns, w).  pp_multiorder_result_ptr(ws) = {types, status, instances of level 2 (= E2), long runs};
 *   status: bit 0 node index out of range, bit 1 time not ascending (both from pp_temporal_count).  The (source, target, time) order of the events:
 *   radix_sort == 0 sorts every node's time-ordered out-list (pp_temporal_count's) by target in LDS — no second global sort; a node with more than
 *   4096 out-events sets status bit 4 and leaves the outputs incomplete: call again with radix_sort != 0 (one stable radix sort of the stream by the
 *   (source, target) key).
 * pp_multiorder_step: level k+1 from level k.  cand_ptr / cand_last: row pointers of layer k (over ITS nodes) and the last nodes of
 *   level k's types (the candidates a column is looked up in; for k = 1: rowptr and tlast of pp_multiorder_prepare).  Outputs with capacity
 *   n_children (tptr_out / ibase_out: n_children + 1): child = inst of level k+1, row_ptr [n_types + 1] = row pointers of layer k+1,
 *   col_out / w_out = its columns and merged weights, tlast_out, tptr_out, ibase_out as above.  last != 0: the top layer — tptr_out,
 *   ibase_out, tlast_out are not written (may be NULL) and `child` is scratch of 4 (weighted: 8) bytes per child instead of 16.  weighted == 0: every event weighs 1 (merged weight = run length).
 *   pp_multiorder_result_ptr(ws) = {types of level k+1, status, instances of level k+2, types handled by workgroups};
 *   status bit 2: a type with more than 4096 children — the outputs are incomplete, use the generic kernels (pp_linegraph_*, pp_coalesce_*). */
size_t pp_multiorder_prepare_ws_bytes(int64_t m);
int pp_multiorder_prepare(const int64_t* edge_index, int64_t m, int64_t num_nodes, const float* weight, void* lift_ws, size_t lift_ws_bytes,
                          int radix_sort, void* tab, void* inst, int32_t* tptr, int32_t* ibase, int32_t* tlast, float* w, int32_t* rowptr, void* ws,
                          size_t ws_bytes, pp_stream_t stream);
/* The same from a GIVEN event graph (from_temporal_graph(..., event_graph=lift_order_temporal(g, delta)), multi_order_model.py:124-192 with
 * `event_graph` set): event_graph [2, num_event_edges] int64, sorted by source (as lift_order_temporal / lift_order_edge_index leave it; the order of
 * a source's targets is kept: it is the reference's instance order).  graph_ws: pp_multiorder_graph_ws_bytes; `tab` has num_event_edges records.
 * Status bit 0: an event id outside [0, m); bit 1: the event graph is not sorted by source. */
size_t pp_multiorder_graph_ws_bytes(int64_t m, int64_t num_event_edges);
int pp_multiorder_prepare_graph(const int64_t* edge_index, int64_t m, int64_t num_nodes, const float* weight, const int64_t* event_graph,
                                int64_t num_event_edges, void* graph_ws, size_t graph_ws_bytes, void* tab, void* inst, int32_t* tptr,
                                int32_t* ibase, int32_t* tlast, float* w, int32_t* rowptr, void* ws, size_t ws_bytes, pp_stream_t stream);
const int64_t* pp_multiorder_result_ptr(void* ws);
size_t pp_multiorder_step_ws_bytes(int64_t n_types, int64_t n_children);
int pp_multiorder_step(int64_t n_types, int64_t n_children, const int32_t* tptr, const int32_t* ibase, const int32_t* col, const void* inst,
                       const int32_t* cand_ptr, const int32_t* cand_last, const void* tab, int weighted, int last, void* child, int32_t* row_ptr,
                       int32_t* tptr_out, int32_t* ibase_out, int32_t* tlast_out, int32_t* col_out, float* w_out, void* ws, size_t ws_bytes,
                       pp_stream_t stream);

/* ------------------------------------------------------------------ order lifts (pp_lift.hip) */

/* lift_order_temporal(g, delta) -> [2,E2] int64, src/pathpyG/algorithms/temporal.py:17-54.
 *   edge_index : [2,m] int64, events in time order (TemporalGraph.__init__ sorted them, temporal_graph.py:58-63)
 *   time       : [m] int64 (PP_I64) or float64 (PP_F64), ascending
 *   delta      : as torch.tensor(delta) sees it: PP_DELTA_I64 -> delta_i; PP_DELTA_F32 / PP_DELTA_F64 -> delta_f
 *                (for PP_DELTA_F32 pass the float32-rounded value).  With float64 time only delta_f is used.
 * pp_temporal_count builds the per-node event lists, counts each event's continuations and scans them;
 * pp_lift_result_ptr(ws)[0] = E2, [1] = status (bit 0: node index outside [0,num_nodes); bit 1: time is not ascending —
 * the result is then meaningless; the reference's mask-based loop has no such precondition, temporal.py:37-43).
 * pp_temporal_fill writes out[0][p] = i, out[1][p] = j for all pairs in lexicographic (i,j) order.
 * Edge-range sharding (one shard per GPU): pass the shard's events followed by its forward halo (all later events
 * with t <= t_last_owned + delta); only the first n_own events act as sources (n_own = m, or < 0, for the whole
 * stream) and id_offset (the global id of the shard's first event) is added to both rows of the result. */
size_t pp_temporal_ws_bytes(int64_t m, int64_t num_nodes);
int pp_temporal_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t n_own, int64_t num_nodes,
                      int delta_kind, int64_t delta_i, double delta_f, void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_temporal_fill(int64_t m, int64_t num_nodes, int64_t total, int64_t id_offset, int64_t* out, void* ws, size_t ws_bytes,
                     pp_stream_t stream);
/* Steps 1-2 of pp_temporal_count only — the per-node event lists and every event's continuation window, no output offsets (result[0] stays 0,
 * pp_temporal_fill must not follow): what pp_multiorder_prepare reads (the multi-order builder never writes the event graph). */
int pp_temporal_windows(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i,
                        double delta_f, void* ws, size_t ws_bytes, pp_stream_t stream);

/* lift_order_edge_index(edge_index, num_nodes) -> [2,E'] int64, src/pathpyG/algorithms/lift_order.py:48-79.
 * edge_index must be grouped by source like the reference demands (:52,55). */
size_t pp_linegraph_ws_bytes(int64_t n_edges, int64_t num_nodes);
int pp_linegraph_count(const int64_t* edge_index, int64_t n_edges, int64_t e_begin, int64_t e_end, int64_t num_nodes, void* ws,
                       size_t ws_bytes, pp_stream_t stream);   /* [e_begin, e_end): the source edges of this call (edge-range shard);
                                                                 out-degrees / row pointers always come from all n_edges edges */
int pp_linegraph_fill(int64_t n_edges, int64_t num_nodes, int64_t total, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream);
const int64_t* pp_lift_result_ptr(void* ws); /* device pointer to {size, status} */

/* aggregate_node_attributes(edge_index, node_attribute, aggr), src/pathpyG/algorithms/lift_order.py:10-45.
 * attr is [num_nodes, width] row-major of `dtype`; out is [n_edges, width]; *status as above (device int64). */
int pp_edge_attr(const int64_t* edge_index, int64_t n_edges, const void* attr, int dtype, int64_t num_nodes, int64_t width, int aggr,
                 void* out, int64_t* status, pp_stream_t stream);

/* cat([ns[ei[0]], ns[ei[1]][:, -1:]], 1), src/pathpyG/core/multi_order_model.py:114,165: rows [n_rows,k] -> out [n_edges,k+1] */
int pp_extend_node_sequence(const int64_t* edge_index, int64_t n_edges, const int64_t* rows, int64_t n_rows, int k, int64_t* out,
                            int64_t* status, pp_stream_t stream);

/* out[i,:k] = rows[idx[i],:], out[i,k] = suffix[i]: the order-(k+1) node sequence of a node whose first k entries are the
 * order-k node idx[i] (multi_order_model.py:114 applied to DISTINCT nodes only, so instance sequences are never materialised) */
int pp_gather_concat(const int64_t* rows, int64_t n_rows, int k, const int64_t* idx, const int64_t* suffix, int64_t n, int64_t* out,
                     int64_t* status, pp_stream_t stream);

/* ------------------------------------------------------------------ De Bruijn aggregation (pp_aggregate.hip) */

/* torch.unique(node_sequence, dim=0, return_inverse=True), src/pathpyG/algorithms/lift_order.py:133.
 * rows [n_rows,k] int64 with every value in [min_value, max_value].  _count writes inverse[n_rows] and
 * {U, 0} to pp_aggregate_result_ptr(ws); _fill writes the U unique rows in lexicographic order. */
size_t pp_unique_rows_ws_bytes(int64_t n_rows);
int pp_unique_rows_count(const int64_t* rows, int64_t n_rows, int k, int64_t min_value, int64_t max_value, int64_t* inverse, void* ws,
                         size_t ws_bytes, pp_stream_t stream);
int pp_unique_rows_fill(const int64_t* rows, int64_t n_rows, int k, int64_t n_unique, int64_t* unique_rows, void* ws, size_t ws_bytes,
                        pp_stream_t stream);

/* coalesce(remap[edge_index], edge_attr, num_nodes, reduce), src/pathpyG/algorithms/lift_order.py:135-144.
 * remap may be NULL (edges already hold node ids).  _count -> {A, status}; _fill writes out_index [2,A]
 * sorted by (row, col) and the reduced weights (weight may be NULL; weight NULL with out_weight given = UNIT weights, the reference's
 * default `torch.ones` (lift_order.py:130-131): out_weight (float32 [A], whatever `dtype` says) = run length for sum, 1 otherwise, without the per-instance gather).
 * col_base [num_nodes] / col_bits (NULL / 0 = off; pass the same pair to _count and _fill): the caller knows that every column of row r
 * lies in [col_base[r], col_base[r] + 2^col_bits) - true for De Bruijn layers, where the successors of a node form one
 * contiguous id block - and the sort key shrinks from 2*bits(num_nodes) to bits(num_nodes) + col_bits bits (fewer radix passes);
 * an edge outside its block sets the bad-index status bit. */
size_t pp_coalesce_ws_bytes(int64_t n_edges);
int pp_coalesce_count(const int64_t* edge_index, int64_t n_edges, const int64_t* remap, int64_t remap_len, int64_t num_nodes,
                      const int64_t* col_base, int col_bits, void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_coalesce_fill(const void* weight, int dtype, int reduce, int64_t n_edges, int64_t n_out, int64_t num_nodes, const int64_t* col_base,
                     int col_bits, int64_t* out_index,
                     void* out_weight, void* ws, size_t ws_bytes, pp_stream_t stream);
/* inverse[n_edges] = position of every input edge's merged edge (call between _count and the end of the workspace's life).
 * For a first-order event list this equals the inverse_idx of torch.unique over the (src,dst) rows, lift_order.py:133, i.e.
 * layer 2's node ids come for free from layer 1's coalesce. */
int pp_coalesce_inverse(int64_t n_edges, int64_t* inverse, void* ws, size_t ws_bytes, pp_stream_t stream);
const int64_t* pp_aggregate_result_ptr(void* ws);

/* Graph.__init__ helpers, src/pathpyG/core/graph.py:103-115 (EdgeIndex.sort_by("row"), get_csr, get_csc) */
int pp_count_descents_i64(const int64_t* a, int64_t n, int64_t* descents, pp_stream_t stream);
int pp_count_descents_f64(const double* a, int64_t n, int64_t* descents, pp_stream_t stream);
size_t pp_argsort_ws_bytes(int64_t n);
int pp_argsort_i64(const int64_t* keys, int64_t n, int64_t min_value, int64_t max_value, int64_t* perm_out, void* ws, size_t ws_bytes,
                   pp_stream_t stream);
/* stable replacement of torch.argsort(time) for float64 timestamps, src/pathpyG/core/temporal_graph.py:58 */
int pp_argsort_f64(const double* keys, int64_t n, int64_t* perm_out, void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_ptr_from_sorted_i64(const int64_t* sorted, int64_t n, int64_t num_rows, int64_t* ptr, pp_stream_t stream);
/* TemporalGraph.__init__, src/pathpyG/core/temporal_graph.py:58-63 (argsort of the timestamps, then edge_index / time indexed by it):
 * out3 = {descents, min, max} of the timestamps in one device buffer (float64: descents only) — one read-back says whether a sort is
 * needed and how many key bits it takes; pp_gather_events applies the permutation to the three 8-byte columns of every event in one
 * pass (time_dtype PP_I64 or PP_F64; status bit 1: permutation entry out of range). */
int pp_time_stats(const void* time, int time_dtype, int64_t n, int64_t* out3, pp_stream_t stream);
int pp_gather_events(const int64_t* edge_index, const void* time, const int64_t* perm, int64_t m, int64_t* edge_index_out, void* time_out,
                     int64_t* status, pp_stream_t stream);

/* ------------------------------------------------------------------ DBGNN message passing (pp_dbgnn.hip) */

/* What PyG's gcn_norm computes on every GCNConv call of DBGNN.forward (src/pathpyG/nn/dbgnn.py:133,139):
 * add_remaining_self_loops(fill 1), weighted in-degree, d^-1/2, norm_e = d[row] w_e d[col] — built ONCE per graph:
 *   in_*  : CSR over DESTINATION nodes (in_ptr [N+1], in_idx = source of each incoming edge, in_val = norm)   forward
 *   out_* : CSR over SOURCE nodes      (out_idx = destination, out_val = norm)                               backward
 *   self_coef[i] = d[i]^2 * (weight of node i's self loop, 1 if it had none); existing self-loop edges get norm 0.
 * edge_weight may be NULL (all ones).  row_sorted != 0: the caller guarantees edge_index[0] is non-decreasing (every Graph's
 * edge index is, graph.py:103) and the source-major grouping reuses the edge order instead of sorting.
 * pp_plan_result_ptr(ws)[1] = status (bit 0: index out of range); [2] / [3] = longest row of the destination-major / source-major CSR (plans
 * with rows above a few hundred entries want pp_spmm_heavy_f32's chunk tables) — written by every plan builder below, one read-back for all. */
size_t pp_gcn_plan_ws_bytes(int64_t n_edges, int64_t n_nodes);
int pp_gcn_plan(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_nodes, int row_sorted, int32_t* in_ptr,
                int32_t* in_idx, float* in_val, int32_t* out_ptr, int32_t* out_idx, float* out_val, float* self_coef, int32_t* dst_order,
                void* ws, size_t ws_bytes, pp_stream_t stream);   /* dst_order [E] or NULL: the edge ids grouped by destination (the
                                                                    permutation behind in_*); with in_ptr it IS the bipartite "last"
                                                                    plan of an order-2 model, whose nodes are this graph's edges */

/* The same plan for a DESTINATION-ROW PARTITION of the graph (multi-GPU DBGNN; the reference is single-process): this rank owns
 * n_dst destination rows; its local source space has n_src >= n_dst rows, the first n_dst being the owned nodes themselves (source i ==
 * destination i) and the rest halo rows owned by peers.  edge_index holds LOCAL ids (row 0 < n_src, row 1 < n_dst).  Two phases around
 * the one exchange the normalisation needs:
 *   pp_gcn_plan_begin : destination grouping, self loops, weighted in-degree -> dinv[0..n_dst), self_coef[n_dst]; with row_sorted
 *                       also out_ptr [n_src+1]
 *   (the caller fills dinv[n_dst..n_src) with the owners' values — one halo exchange of 4 bytes per halo row)
 *   pp_gcn_plan_finish: coefficients dinv[src] w dinv[dst] of both groupings (in_val in place; out_*).
 * The workspace (pp_gcn_plan_ws_bytes(n_edges, n_src)) carries the state between the two calls and must not be touched in between.
 * pp_gcn_plan == begin + finish with n_src == n_dst. */
int pp_gcn_plan_begin(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_src, int64_t n_dst, int row_sorted,
                      int32_t* in_ptr, int32_t* in_idx, float* in_val, int32_t* out_ptr, float* self_coef, float* dinv, int32_t* dst_order,
                      void* ws, size_t ws_bytes, pp_stream_t stream);
int pp_gcn_plan_finish(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_src, int64_t n_dst, int row_sorted,
                       const float* dinv, const int32_t* in_idx, float* in_val, int32_t* out_ptr, int32_t* out_idx, float* out_val, void* ws,
                       size_t ws_bytes, pp_stream_t stream);

/* CSR views of DBGNN's bipartite_edge_index [2,n_pairs] (row 0: higher-order node, row 1: first-order node),
 * src/pathpyG/nn/dbgnn.py:64-69: in_* grouped by first-order node (+ its in-degree as float), out_* by higher-order node.
 * src_sorted != 0: row 0 is non-decreasing (arange for the reference's "last"/"first" mappings): no source-major sort.
 * pair_value (optional, with in_val/out_val): a coefficient per pair carried into both groupings — this makes the call a
 * general RECTANGULAR plan builder (sources x destinations), used for the destination-partitioned multi-GPU DBGNN where a rank
 * owns a slice of the destination rows but aggregates from all source rows.
 * Workspace: pp_gcn_plan_ws_bytes(n_pairs, max(n_ho, n_fo)). */
int pp_bipartite_plan(const int64_t* bipartite_index, int64_t n_pairs, int64_t n_ho, int64_t n_fo, int src_sorted, const float* pair_value,
                      int32_t* in_ptr, int32_t* in_idx, float* in_val, float* in_degree, int32_t* out_ptr, int32_t* out_idx, float* out_val,
                      void* ws, size_t ws_bytes, pp_stream_t stream);
const int64_t* pp_plan_result_ptr(void* ws);

/* Y[r,:] = act( sum_{p in [ptr[r],ptr[r+1])} val[p] * X[idx[p],:] + self_coef[r] * S[r,:] + bias ),  X:[*,F], Y:[n_rows,F] fp32.
 * val NULL = 1, self_coef NULL = no self term, S NULL = X, bias NULL = none, act 0 = identity / 1 = ELU.
 * GCNConv.propagate + bias + F.elu (dbgnn.py:133,139), the bipartite propagate + elu (:143-144) and all their transposes. */
int pp_spmm_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* X, int F, const float* self_coef,
                const float* S, const float* bias, int act, const int32_t* heavy_slot, const float* heavy_sum, float* Y, pp_stream_t stream);

/* Hub rows (scale-free graphs: 10^5+ entries in one CSR row).  The row kernels (pp_spmm_f32, pp_gcn_forward_f32, pp_gcn_backward_f32)
 * walk a row with ONE lane group; for rows the caller marks as heavy (heavy_slot [n_rows] int32: -1 = ordinary, else h) they read the
 * neighbour sum from heavy_sum [n_heavy, F] instead, which pp_spmm_heavy_f32 computes with a whole workgroup per chunk of
 * pp_heavy_chunk_entries() entries and a fixed-order combine (bit-reproducible).  heavy_slot NULL = no heavy rows.
 * pp_max_row_length_i32: out_max[0] (device int64) = longest row of a CSR pointer array, to decide whether a plan needs this. */
int pp_max_row_length_i32(const int32_t* ptr, int64_t n_rows, int64_t* out_max, pp_stream_t stream);
int pp_heavy_chunk_entries(void);
size_t pp_spmm_heavy_ws_bytes(int64_t n_chunks, int F);
int pp_spmm_heavy_f32(const int32_t* idx, const float* val, const float* X, int F, int64_t n_chunks, const int32_t* chunk_begin,
                      const int32_t* chunk_end, int64_t n_heavy, const int32_t* heavy_chunk_ptr, float* heavy_sum, void* ws, size_t ws_bytes,
                      pp_stream_t stream);

/* Transposed aggregation fused with the ELU backward of the layer that produced its input and with that layer's bias gradient:
 *   dX[r,:] = ( sum_{p in [ptr[r],ptr[r+1])} val[p] * D[idx[p],:] ) * ELU'(Z[r,:]),   colsum[F] (may be NULL) = column sums of dX
 * with Z the stored activation ELU(pre) and ELU' = (Z > 0 ? 1 : Z + 1).  F a multiple of 4, <= 256.  Used for the backward of the
 * bipartite aggregation sum_j x_h[j] (dbgnn.py:50-69 re-associated: lin1 is applied AFTER the sum over the higher-order nodes). */
int pp_spmm_act_backward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* D, int F, const float* Z,
                             float* colsum, float* dX, pp_stream_t stream);
/* the same with Z stored DROPPED (F.dropout between the last higher-order layer and the bipartite layer, dbgnn.py:142; masks of pp_dropout_f32):
 * dX = (A D) * mask / (1 - p) * ELU'(Z * (1 - p)) */
int pp_spmm_act_backward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* D, int F, const float* Z,
                                  float* colsum, float* dX, double drop_p, int64_t drop_seed, int64_t drop_tag, int64_t drop_row0,
                                  pp_stream_t stream);

/* dpre = dY * elu'(.) from the stored OUTPUT y (1 if y > 0 else y + 1); act 0: dpre = dY; dbias[F] = column sums of dpre.
 * dpre or dbias may be NULL. */
int pp_act_backward_f32(const float* dY, const float* Y, int64_t n_rows, int F, int act, float* dpre, float* dbias, pp_stream_t stream);

/* The element-wise tail of BipartiteGraphOperator.forward + F.elu (src/pathpyG/nn/dbgnn.py:66-69,143-144) on the first-order rows, after the
 * re-association sum_j lin1(x_h[j]) = lin1.weight (sum_j x_h[j]) + deg * lin1.bias:  Y = ELU(A + deg[r] * (P + bias)),  A = lin1.weight applied
 * to the summed higher-order rows, P = lin2(x), deg [n_rows] = incoming pairs per first-order node, bias [F] or NULL.
 * _backward: dpre = dY * ELU'(Y);  dA = dpre;  dP = deg[r] * dpre;  dbias[c] = sum_r dP[r][c] (NULL: not wanted). */
int pp_bip_combine_f32(const float* A, const float* P, const float* deg, const float* bias, int64_t n_rows, int F, float* Y, pp_stream_t stream);
int pp_bip_combine_backward_f32(const float* dY, const float* Y, const float* deg, int64_t n_rows, int F, float* dA, float* dP, float* dbias,
                                pp_stream_t stream);

/* Partitioned DBGNN backward (no single-process counterpart in the reference; the sum it completes is the transposed aggregation of
 * GCNConv's backward, nn/dbgnn.py:131-140 under autograd): out[r,:] = own[r,:] + recv[slot[r],:] (slot[r] >= 0) + extra[r,:] + self_coef[r] * dpre[r,:],
 * every addend optional (NULL); recv/slot and self_coef/dpre come in pairs; F a multiple of 4, 16-byte aligned rows; out may alias own. */
int pp_halo_fold_f32(const float* own, const float* recv, const int32_t* slot, const float* extra, const float* self_coef, const float* dpre,
                     int64_t n_rows, int F, float* out, pp_stream_t stream);

/* out[r,:] = coef[r] * X[r,:] (gradient of the bipartite self term) */
int pp_scale_rows_f32(const float* X, const float* coef, int64_t n_rows, int F, float* out, pp_stream_t stream);

/* F.dropout(x, p, training=True) of DBGNN.forward (nn/dbgnn.py:132,136,138,142,148) with COUNTER-BASED masks: element (r, c) of a [n_rows, F] matrix
 * is kept iff hash(seed, tag, global row, c) >= p * 2^32 (two multiply-xorshift rounds on 32 bits), global row = rows[r] if rows != NULL else
 * row0 + r.  No mask tensor: the backward pass regenerates it, and the ranks of a partitioned run agree on every row.
 *   pp_dropout_f32               out = x * keep / (1 - p)                                   (out may alias x)
 *   pp_dropout_act_backward_f32  dpre = dY * keep / (1 - p) * (act ? ELU'(y) : 1), y = Ydrop * (1 - p) where kept;  dbias[F] (optional) = column sums
 * i.e. the backward of dropout and the ELU backward of the layer underneath in one pass over (dY, Ydrop). */
int pp_dropout_f32(const float* X, int64_t n_rows, int F, double p, int64_t seed, int64_t tag, int64_t row0, const int64_t* rows, float* out,
                   pp_stream_t stream);
int pp_dropout_act_backward_f32(const float* dY, const float* Ydrop, int64_t n_rows, int F, double p, int64_t seed, int64_t tag, int64_t row0,
                                const int64_t* rows, int act, float* dpre, float* dbias, pp_stream_t stream);

/* Weight gradient of a dense layer of the DBGNN (autograd of lin / lin1 / lin2 / GCNConv.lin, dbgnn.py:64,133,139,149):
 * dW[M,K] = dH[N,M]^T X[N,K] and, when db != NULL, db[M] = column sums of dH.  fp32 on the matrix cores
 * (v_mfma_f32_32x32x2_f32, operands read straight from the row-major inputs), deterministic two-stage reduction. */
size_t pp_weight_grad_ws_bytes(int64_t n_rows, int M, int K);
int pp_weight_grad_f32(const float* dH, const float* X, int64_t n_rows, int M, int K, float* dW, float* db, void* ws, size_t ws_bytes,
                       pp_stream_t stream);

/* Dense layer on the matrix cores (the Linear / GCNConv.lin calls of dbgnn.py:64,133,139,149 and their input gradients):
 *   out[N,Q] = ( A[N,P] . B + bias ) (*) g'      B = W^T for w_transposed != 0 (W is [Q,P]: forward), W otherwise (W is [P,Q])
 *   grad_act : [N,Q] or NULL: the stored activation y = ELU(pre) of the layer below; the result is multiplied by
 *              ELU'(pre) = (y > 0 ? 1 : y + 1) and, with colsum [Q], its column sums (= that layer's bias gradient) accumulate.
 * fp32 on v_mfma_f32_16x16x4_f32.  pp_dense_supported(P, Q): 1 = P, Q in {16, 32, 64} (weights in registers; also pp_dense_backward_f32),
 * 2 = other widths whose padded product is <= 4096, e.g. anything up to 64 x 64 or 256 x 8 (zero-padded: pp_dense_narrow_f32), 3 = 64/128/256 with a side > 64 (pp_wide_layer_f32 in dense mode; with
 * w_transposed == 0 it needs ws = pp_wide_layer_ws_bytes(P, Q) bytes), 0 = not supported (the Python side then uses the library GEMM). */
int pp_dense_supported(int P, int Q);
int pp_dense_f32(const float* A, const float* W, int w_transposed, int64_t n_rows, int P, int Q, const float* bias,
                 const float* grad_act, float* colsum, float* out, void* ws, size_t ws_bytes, pp_stream_t stream);

/* Mean softmax cross-entropy of logits [n,C] (C <= 64) against int64 targets [n] and its gradient dlogits [n,C] (may be NULL) in
 * one pass - the loss of the train step bench.py times (the reference ships no training loop). */
size_t pp_cross_entropy_ws_bytes(void);       /* per-workgroup partial sums, added in a fixed order: the loss is bitwise reproducible */
int pp_cross_entropy_f32(const float* logits, const int64_t* target, int64_t n, int C, float* loss, float* dlogits, void* ws, size_t ws_bytes,
                         pp_stream_t stream);

/* One Adam step (step = 1, 2, ...) over ALL n_tensors fp32 parameter tensors in one launch per 24 tensors: HOST arrays of DEVICE pointers
 * (params, grads, exp_avg, exp_avg_sq) and of element counts.  The update of torch.optim.Adam (amsgrad off, maximize off; weight_decay is
 * the L2 form g + wd*p).  No reference counterpart: pathpyG ships the model (nn/dbgnn.py:72-151); its only training loop is the upstream tutorial
 * docs/tutorial/dbgnn.ipynb, which is absent from /root/reference (.MISSING_LARGE_BLOBS:2).  This is the optimizer step such a loop takes from
 * torch.optim.Adam (15 small launches there, one here). */
int pp_adam_f32(int n_tensors, void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, pp_stream_t stream);

/* Whole backward of a dense layer y = x W^T (+ b) in one pass over dH [N,M] and x [N,K] (W is [M,K]; M, K in {16,32,64}):
 *   d_in[N,K] = (dH . W) (*) ELU'(x) when fuse_act (x is then the stored activation of the layer below), colsum_in[K] = its column
 *   sums (that layer's bias gradient); dW[M,K] = dH^T x; db[M] = column sums of dH.  d_in/colsum_in/db may be NULL.
 * Replaces pp_dense_f32(input gradient) + pp_weight_grad_f32: 3 instead of 5 passes over N x 64 matrices. */
size_t pp_dense_backward_ws_bytes(int64_t n_rows);
int pp_dense_backward_f32(const float* dH, const float* X, const float* W, int64_t n_rows, int M, int K, int fuse_act, float* d_in,
                          float* colsum_in, float* dW, float* db, void* ws, size_t ws_bytes, pp_stream_t stream);

/* A whole GCNConv layer (dbgnn.py:131-140) in one kernel, re-associated as (A_hat X) W^T so that the transformed matrix never
 * makes a round trip through HBM:
 *   Y[r, :Q] = act( (sum_e val[e] X[idx[e], :P] + self_coef[r] X[r, :P]) . W^T + bias ),   W is [Q,P] (Linear layout), act 0/1 (ELU)
 * over the destination-major CSR of a pp_gcn_plan (self_coef may be NULL: no self term); X has n_src rows (rows are addressed by
 * 32-bit byte offsets below 4 GiB, by 64-bit ones above).  P, Q in {16,32,64}; the 128-wide shapes 64x128, 128x64, 128x128 (W then
 * fills 32-64 KB of LDS: one workgroup of 8 waves per CU); every other combination of 64/128/256 goes to pp_wide_layer_f32 (weights
 * streamed through LDS).  pp_gcn_fused_supported(P, Q): 1 = forward + pp_gcn_backward_f32, 2 = forward + pp_gcn_input_grad_f32
 * (a side of 128 or 256), 0 = unsupported shape.
 * agg_out [n_rows,P] or NULL: also store the aggregated input A_hat X; the weight gradient of a layer whose input needs no
 * gradient is then dW = dpre^T agg_out (pp_weight_grad_f32) without any backward aggregation. */
int pp_gcn_fused_supported(int P, int Q);
int pp_gcn_forward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_src, const float* X, int P,
                       const float* self_coef, const float* W, int Q, const float* bias, int act, const int32_t* heavy_slot, const float* heavy_sum,
                       float* agg_out, float* Y, pp_stream_t stream);

/* Backward of that layer in one kernel (pp_spmm_f32 over the source-major CSR + pp_dense_backward_f32 without the round trip of the
 * aggregated gradient through HBM):  G = A^T D + diag(self_coef) D with D = dpre [n_rows,M];
 *   d_in[n_rows,K] = (G . W) (*) ELU'(X) when fuse_act (X [n_rows,K] is then the stored activation of the layer below),
 *   colsum_in[K] (may be NULL) = column sums of d_in,  dW[M,K] = G^T X.   W is [M,K]; M, K in {16,32,64}.
 * n_self <= n_rows: only the first n_self rows carry the self term (D then has n_self rows).  n_self == n_rows for a whole graph;
 * a destination-row partition (multi-GPU) has n_rows = owned + halo source rows and n_self = owned rows. */
size_t pp_gcn_backward_ws_bytes(int64_t n_rows);
int pp_gcn_backward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                        const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                        const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, pp_stream_t stream);

/* Input gradient of a 128-wide fused layer (GCNConv backward, reference nn/dbgnn.py:131-140 through autograd): the forward kernel over the
 * source-major CSR with a gradient epilogue,  d_in[n_rows,K] = ((A^T D + diag(self_coef) D) . W) (*) ELU'(X_act) when fuse_act,
 * colsum_in[K] (may be NULL) = column sums of d_in.  D = dpre [n_rows,M], W [M,K]; shapes 64x128, 128x64, 128x128.  The weight gradient
 * of such a layer is dW = dpre^T (A_hat X) with the agg_out of its forward call (pp_weight_grad_f32): 64 accumulator registers per
 * 64x64 block do not fit beside the gather here. */
int pp_gcn_input_grad_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                          const float* self_coef, const float* W, int K, const float* X_act, int fuse_act, const int32_t* heavy_slot,
                          const float* heavy_sum, float* d_in, float* colsum_in, void* ws, size_t ws_bytes, pp_stream_t stream);
/* The same three calls with the reference's training-mode F.dropout (src/pathpyG/nn/dbgnn.py:132,138,142) fused into their epilogues
 * (counter-based masks, see pp_dropout_f32; drop_p == 0: exactly the calls above; pp_gcn_drop_supported(P, Q): the 16/32/64 and 128-wide
 * kernels).  forward: Y is dropped — mask(drop_seed, drop_tag, drop_row0 + row, column) — before it is stored, i.e. the NEXT layer's input
 * dropout costs no pass.  backward / input_grad (with fuse_act): X / X_act is the dropped activation of the layer below (site drop_tag);
 * d_in becomes the gradient w.r.t. that layer's pre-activation, (G W) * mask / (1 - p) * ELU'(X * (1 - p)). */
int pp_gcn_drop_supported(int P, int Q);
int pp_gcn_forward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_src, const float* X, int P,
                            const float* self_coef, const float* W, int Q, const float* bias, int act, const int32_t* heavy_slot,
                            const float* heavy_sum, float* agg_out, float* Y, double drop_p, int64_t drop_seed, int64_t drop_tag, int64_t drop_row0,
                            pp_stream_t stream);
int pp_gcn_backward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                             const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                             const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, double drop_p,
                             int64_t drop_seed, int64_t drop_tag, int64_t drop_row0, pp_stream_t stream);
/* The same call with the number of CSR entries, when the caller knows it (nnz < 0: unknown): graphs with short rows (nnz <= 8 n_rows, a
 * De Bruijn layer) run a register-capped variant of the 64 x 64 kernel (3 waves per SIMD); long-row graphs keep the 2-wave one. */
int pp_gcn_backward_nnz_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, int64_t nnz, const float* D, int M,
                            const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                            const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, double drop_p,
                            int64_t drop_seed, int64_t drop_tag, int64_t drop_row0, pp_stream_t stream);
int pp_gcn_input_grad_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                               const float* self_coef, const float* W, int K, const float* X_act, int fuse_act, const int32_t* heavy_slot,
                               const float* heavy_sum, float* d_in, float* colsum_in, void* ws, size_t ws_bytes, double drop_p, int64_t drop_seed,
                               int64_t drop_tag, int64_t drop_row0, pp_stream_t stream);

                          /* ws: pp_wide_layer_ws_bytes(M, K) for the shapes served by pp_wide_layer_f32 (a side of 256), else unused */

/* Layers too wide for a weight matrix in LDS (pp_gcn_wide.hip): P, Q in {64, 128, 256} with a side > 64 — DBGNN with 256-dim features
 * (BASELINE configs[4]; GCNConv / Linear of nn/dbgnn.py:104-119 and their backward).  One kernel per layer:
 *   Y[n, :Q] = epi( tile[n, :P] . B ),  tile[n] = sum_e val[e] X[idx[e]] + self_coef[n] X[n]  (rows >= n_self: no self term)
 *                                       or X[n] when ptr == NULL (a dense layer; idx/val/self_coef ignored)
 *   B = W^T with W [Q,P] (w_is_kq == 0: Linear layout, the forward) or B = W with W [P,Q] (w_is_kq == 1: the input gradient; transposed
 *   once into ws, pp_wide_layer_ws_bytes(P, Q) bytes)
 *   epilogue 0: Y = act(.. + bias), agg_out [n_rows,P] (optional) receives the aggregated tile;
 *   epilogue 1: Y = (..) (*) ELU'(act_in) when act (act_in [n_rows,Q] = stored activation), colsum[Q] (optional) = column sums of Y.
 * The 16 x P tile lives in registers in MFMA A layout, the weights stream through LDS 16 output columns at a time. */
int pp_wide_layer_supported(int P, int Q);
size_t pp_wide_layer_ws_bytes(int P, int Q);
int pp_wide_layer_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, int64_t n_src, const float* X, int P,
                      const float* self_coef, const float* W, int w_is_kq, int Q, const float* bias, int act, int epilogue, const float* act_in,
                      const int32_t* heavy_slot, const float* heavy_sum, float* agg_out, float* Y, float* colsum, void* ws, size_t ws_bytes,
                      pp_stream_t stream);

/* Dense layers with one small side (classifier head: 64 -> 8 or 256 -> 8 classes and its input gradient; odd hidden widths): the
 * pp_dense_f32 scheme with both sides zero-padded to 16/32/64/128/256 (padded P*Q <= 4096) and guarded scalar I/O on the true widths.
 * Same arguments as pp_dense_f32. */
int pp_dense_narrow_supported(int P, int Q);
int pp_dense_narrow_f32(const float* A, const float* W, int w_transposed, int64_t n_rows, int P, int Q, const float* bias, const float* grad_act,
                        float* colsum, float* out, pp_stream_t stream);

/* All-pairs shortest time-respecting paths (temporal_shortest_paths, src/pathpyG/algorithms/temporal.py:57-107: scipy Dijkstra with
 * unit weights on the event DAG augmented by a virtual source and sink per node) as a frontier BFS per source node on the event graph:
 *   edge_index [2,m] time-sorted events;  succ_ptr [m+1] / succ [E2]: CSR of lift_order_temporal's result (row i -> events j);
 *   by_src_ptr [n+1] / by_src [m]: event ids grouped by their source node;
 *   dist [n,n] int32: number of events on a shortest path, -1 = unreachable, 0 on the diagonal;
 *   pred [n,n] int64: source node of the latest event that completes a shortest path into the column node (-1 none, s on the diagonal). */
size_t pp_temporal_bfs_ws_bytes(int64_t m, int64_t n);
int pp_temporal_bfs(const int64_t* edge_index, int64_t m, int64_t n, const int64_t* succ_ptr, const int64_t* succ, const int64_t* by_src_ptr,
                    const int64_t* by_src, int32_t* dist, int64_t* pred, void* ws, size_t ws_bytes, pp_stream_t stream);

/* Temporal betweenness centrality (temporal_betweenness_centrality, src/pathpyG/algorithms/centrality.py:164-297: Brandes on the event
 * DAG) in level-synchronous form, one workgroup per source node.  Inputs as pp_temporal_bfs plus the events grouped by their head
 * node (by_dst_ptr [n+1] / by_dst [m]).  partial [pp_temporal_betweenness_parts(m,n), n] float64: one row per workgroup; the
 * centrality of node v is the sum of column v over the rows (summed by the caller in row order: bit-reproducible). */
int64_t pp_temporal_betweenness_parts(int64_t m, int64_t n);
size_t pp_temporal_betweenness_ws_bytes(int64_t m, int64_t n);
int pp_temporal_betweenness(const int64_t* edge_index, int64_t m, int64_t n, const int64_t* succ_ptr, const int64_t* succ,
                            const int64_t* by_src_ptr, const int64_t* by_src, const int64_t* by_dst_ptr, const int64_t* by_dst,
                            double* partial, void* ws, size_t ws_bytes, pp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PATHPYG_AMD_H */
