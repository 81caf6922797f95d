// pathpyg_amd — order lifts on gfx950: temporal event-graph lift and line-graph lift.
//
// Reference functions replaced (paths relative to the pathpyG repository root):
//   lift_order_temporal             src/pathpyG/algorithms/temporal.py:17-54
//   lift_order_edge_index           src/pathpyG/algorithms/lift_order.py:48-79
//   aggregate_node_attributes       src/pathpyG/algorithms/lift_order.py:10-45
//   node-sequence extension         src/pathpyG/core/multi_order_model.py:114,165
//
// Both lifts are "count -> exclusive scan -> fill" over a CSR of continuations; none of the reference's
// per-timestamp loop, masks or cartesian products exist here.
//
// Temporal lift, events i = 0..m-1 in (stable) time order:
//   1. radix-sort event ids by tail node (pp_sort.hip; stable => ids ascend inside a node's list),
//      row pointers from the run boundaries of the sorted keys.
//   2. k_temporal_count: the admissible continuations of event i are the event ids in
//      [g_lo, g_hi) = {j : t_j > t_i and t_j <= t_i + delta}, a contiguous id range because events are
//      time-sorted; both ends come from binary searches on the time array with the comparison done in
//      the dtype torch promotes to (temporal.py:30,43).  Inside the id list of head(i) the ids of that
//      range are again contiguous: two more binary searches give (first position, count).
//   3. exclusive scan of the counts (pp_scan.hip) -> output offsets and E2.
//   4. k_expand: one WAVE per 512 output slots (no workgroup barriers); the sources overlapping the tile are
//      staged in LDS, a running maximum maps slots to sources, every lane stores 16 bytes per row into the
//      row-major [2,E2] result (lexicographic (i,j) order by construction).
// HBM algorithmic bytes: 24*m (tail, head, time) + 16*E2 (result).
#include "pp_internal.h"
#include "pp_window.h"

namespace pp {

// status word bits reported next to the size (see pp_*_count)
constexpr int64_t kBadIndex = 1;
constexpr int64_t kUnsortedTime = 2;

// ------------------------------------------------------------------ small element-wise kernels
__global__ __launch_bounds__(kBlock) void k_tail_keys(const int64_t* __restrict__ tail, int64_t m, int64_t num_nodes,
                                                     uint32_t* __restrict__ keys, int64_t* __restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    int64_t v = tail[i];
    if (v < 0 || v >= num_nodes) { atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex); v = 0; }
    keys[i] = (uint32_t)v;
}

// rowptr[v] = first position p with sorted_keys[p] >= v, for v in [0, num_rows]; one thread per boundary.
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_rowptr_from_sorted(const KeyT* __restrict__ sorted_keys, int64_t n, int64_t num_rows,
                                                              uint32_t* __restrict__ rowptr) {
    int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    int64_t a = p == 0 ? -1 : (int64_t)sorted_keys[p - 1];
    int64_t b = p == n ? num_rows : (int64_t)sorted_keys[p];
    if (b > num_rows) b = num_rows;
    for (int64_t v = a + 1; v <= b; ++v) rowptr[v] = (uint32_t)p;
}

// marks[w] = first event id whose time the window of event 64*w (the first lane of wave w of k_temporal_count) no longer admits, over the
// whole stream; marks[n_waves] = m.  The stream is time-sorted, so the window end of every event of wave w lies in
// [marks[w], marks[w + 1]]: the count kernel bisects ~6 levels inside one or two cache lines shared by the wave instead of 23 levels
// over the stream.  One thread per wave: m/64 full bisections, 47 us at m = 10^7 (neighbouring threads share the lines of the upper
// levels; a 17-ary search with 16 lanes per mark — 6 round trips instead of 23 — took 94 us: every round touches 16 scattered lines per
// mark, and scattered lines per load instruction, not round trips, are what these searches pay for).
template <typename TimeT, int kMode>
__global__ __launch_bounds__(kBlock) void k_window_marks(const TimeT* __restrict__ time, int64_t m, int64_t n_waves, int64_t delta_i, double delta_f,
                                                        uint32_t* __restrict__ marks) {
    using W = Window<TimeT, kMode>;
    const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (w > n_waves) return;
    if (w == n_waves) { marks[w] = (uint32_t)m; return; }
    const typename W::Thr thr = W::threshold(time[w * kWave], delta_i, delta_f);
    int64_t lo = w * kWave, hi = m;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (W::admits(time[mid], thr)) lo = mid + 1; else hi = mid;
    }
    marks[w] = (uint32_t)lo;
}

// Tail of the list search for lists longer than the part `covered` that was counted in registers (hub nodes): finish both searches on
// memory beyond it and fetch the head continuations that lie there.  In: pos / end = s0 + (ids below g_lo / g_hi among the covered part),
// `taken` = head slots already filled.  Out: the final position, the count, the completed head slots.
constexpr int kHeadSlots = 4;
__device__ __forceinline__ void list_tail(const uint32_t* __restrict__ ids_by_tail, uint32_t covered, uint32_t s1, uint32_t glo, uint32_t ghi,
                                          int taken, uint32_t& pos, uint32_t& end, int32_t& c, uint4& h) {
    if (covered < s1) {
        if (pos == covered) pos = lower_bound_dev<uint32_t, uint32_t>(ids_by_tail, covered, s1, glo);
        if (end == covered) end = lower_bound_dev<uint32_t, uint32_t>(ids_by_tail, pos > covered ? pos : covered, s1, ghi);
        c = (int32_t)(end - pos);
        if (c > taken) {         // some of the first 4 continuations lie beyond the register window
            if (taken < 1 && c > 0) h.x = ids_by_tail[pos];
            if (taken < 2 && c > 1) h.y = ids_by_tail[pos + 1];
            if (taken < 3 && c > 2) h.z = ids_by_tail[pos + 2];
            if (taken < 4 && c > 3) h.w = ids_by_tail[pos + 3];
        }
    } else {
        c = (int32_t)(end - pos);
    }
    if (c == 0) pos = 0;
}

// Per event i: the admissible id window [g_lo, g_hi) and, inside the id list of head(i), (first position, count) of the ids in that window
// + the first 4 of them.
// Phase A, one lane per event: times, list bounds (ONE 8-byte load at a 4-byte aligned address: rowptr[v], rowptr[v + 1] as two
//   instructions touch the same 64 random lines twice), g_lo (= i + 1 without timestamp ties: probe, gallop, bisect), g_hi by bisection
//   between the two window marks of the wave (k_window_marks: ~6 levels inside one or two cache lines all 64 lanes share instead of 23
//   dependent levels over the stream; 0.635 -> 0.466 ms).  Every search relies on a time-sorted stream (the reference's mask loop does
//   not, temporal.py:37-43): a descent sets a status bit instead of returning a wrong event graph.
// Phase B, EIGHT LANES PER EVENT: the list is short (the node's out-degree) and sits in one or two cache lines nothing else on this CU
//   touches again.  A group of 8 lanes reads ONE aligned 128-byte line of its event's list per instruction (the wave covers 8 events at
//   a time, 8 rounds), counts the ids below g_lo / g_hi in registers, sums over the group with 3 xor-shuffles; the first 4 continuations
//   and (position, end) go back to the owning lane through a wave-private LDS row.  Lists longer than 3 lines (hubs) finish on memory.
// History (same-box A/B): two bisections per event on memory 0.74 ms; one lane per event fetching 12 sixteen-byte pieces
//   back to back 0.447 ms; this form 0.43 ms.  The kernel moves ~3 GB over the fabric per launch (1.6 random 128-byte lines per event
//   out of a 40 MB array that lives in the Infinity Cache, 10 % of it in an XCD's L2): ~7 TB/s, the same rate the row gathers of the
//   GCN kernels reach from that cache — it is bound by that traffic, not by instruction issue.  Measured and rejected: a wave-cooperative
//   64-ary search (first steps touch 64 scattered lines per wave), LDS-staged bisection (98 VGPRs), `nt` loads of the lists.
constexpr int kGroupLanes = 8;
constexpr int kLineIds = 32;                   // ids per aligned 128-byte line
constexpr int kGroupLines = 3;                 // lines a group reads before the (hub) fallback on memory: >= 65 ids from s0
#ifndef PP_ROUND_BATCH
#define PP_ROUND_BATCH 1
#endif
constexpr int kRoundBatch = PP_ROUND_BATCH;      // rounds of phase B whose loads share one round trip

template <typename TimeT, int kMode>
__global__ __launch_bounds__(kBlock) void k_temporal_count(const int64_t* __restrict__ head, const TimeT* __restrict__ time, int64_t m,
                                                             int64_t n_own, int64_t num_nodes, int64_t delta_i, double delta_f,
                                                             const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids_by_tail,
                                                             const uint32_t* __restrict__ marks, uint32_t* __restrict__ first_pos,
                                                             int32_t* __restrict__ count, uint32_t* __restrict__ head4, int64_t* __restrict__ status) {
    using W = Window<TimeT, kMode>;
    __shared__ __attribute__((aligned(16))) uint32_t s_head[kWavesPerBlock][kWave][4];
    __shared__ uint2 s_pc[kWavesPerBlock][kWave];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < m;
    const bool source = live && i < n_own;
    const int64_t ic = live ? i : m - 1;
    const TimeT ti = time[ic];
    const TimeT t_next = ic + 1 < m ? time[ic + 1] : ti;
    if (live && t_next < ti) atomicOr((unsigned long long*)status, (unsigned long long)kUnsortedTime);
    if (__ballot(source) == 0ull) {
        if (live) { first_pos[i] = 0; count[i] = 0; }
        return;
    }
    const int64_t v = source ? load_stream(head + i) : 0;
    const bool ok = v >= 0 && v < num_nodes;
    struct __attribute__((packed, aligned(4))) Bounds { uint32_t begin, end; };
    Bounds bounds = {0u, 0u};
    if (source && ok) bounds = *reinterpret_cast<const Bounds*>(rowptr + v);
    int64_t g_lo = ic + 1;
    if (g_lo < m && !(t_next > ti)) {
        int64_t step = 2;
        while (ic + step < m && !(time[ic + step] > ti)) step <<= 1;
        int64_t hi = ic + step < m ? ic + step : m;
        g_lo = upper_bound_dev<TimeT, int64_t>(time, ic + (step >> 1) + 1, hi, ti);
    }
    const typename W::Thr thr = W::threshold(ti, delta_i, delta_f);
    int64_t g_hi;
    {
        const int64_t wv = i >> 6;
        int64_t lo = marks[wv], hi = marks[wv + 1];
        if (lo < g_lo) lo = g_lo;
        if (lo > m) lo = m;
        if (hi > m) hi = m;
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (W::admits(time[mid], thr)) lo = mid + 1; else hi = mid;
        }
        g_hi = lo;
    }
    if (g_hi < g_lo) g_hi = g_lo;
    if (source && !ok) atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex);
    const bool need = source && ok && g_hi > g_lo && bounds.end > bounds.begin;
    // ---- phase B: 8 lanes per event
    const int l = lane_id(), w = wave_id();
    const int grp = l >> 3, q = l & 7;
    const uint32_t my_s0 = need ? bounds.begin : 0u, my_s1 = need ? bounds.end : 0u;
    const uint32_t my_glo = (uint32_t)g_lo, my_ghi = (uint32_t)g_hi;
    *reinterpret_cast<uint4*>(s_head[w][l]) = make_uint4(0u, 0u, 0u, 0u);
    s_pc[w][l] = make_uint2(0u, 0u);
    __builtin_amdgcn_wave_barrier();
    // kRoundBatch rounds issue their line loads back to back before the first compare.  Measured (round 3, same box, count phase of the
    // 10^7-event stream): 1 round per trip 0.754 ms, 2: 0.765, 4: 0.917, 8: 1.31 — MORE loads in flight make the kernel slower: it is not
    // waiting for round trips, it is bound by the rate at which the memory pipeline takes instructions that touch 8 distinct lines each
#pragma unroll 1
    for (int round0 = 0; round0 < kWave / kGroupLanes; round0 += kRoundBatch) {
        uint32_t s0v[kRoundBatch], s1v[kRoundBatch];
        bool any = false;
#pragma unroll
        for (int rr = 0; rr < kRoundBatch; ++rr) {
            const int e = (round0 + rr) * kGroupLanes + grp;    // the event (lane) this group serves in this round
            s0v[rr] = (uint32_t)__shfl((int)my_s0, e, kWave);
            s1v[rr] = (uint32_t)__shfl((int)my_s1, e, kWave);
            any = any || s1v[rr] > s0v[rr];
        }
        if (__ballot(any) == 0ull) continue;                    // (wave-uniform)
        uint4 ch[kRoundBatch][kGroupLines];
#pragma unroll
        for (int rr = 0; rr < kRoundBatch; ++rr) {
            const uint32_t a0 = s0v[rr] & ~(uint32_t)(kLineIds - 1);
#pragma unroll
            for (int k = 0; k < kGroupLines; ++k) {
                const uint32_t at = a0 + (uint32_t)(k * kLineIds + 4 * q);
                ch[rr][k] = at < s1v[rr] ? *reinterpret_cast<const uint4*>(ids_by_tail + at) : make_uint4(0u, 0u, 0u, 0u);   // (the list array is padded to 4)
            }
        }
#pragma unroll
        for (int rr = 0; rr < kRoundBatch; ++rr) {
            const int e = (round0 + rr) * kGroupLanes + grp;
            const uint32_t s0 = s0v[rr], s1 = s1v[rr];
            const uint32_t glo = (uint32_t)__shfl((int)my_glo, e, kWave), ghi = (uint32_t)__shfl((int)my_ghi, e, kWave);
            const uint32_t a0 = s0 & ~(uint32_t)(kLineIds - 1);
            uint32_t below = 0;                                  // below_lo in the low half, below_hi in the high half
#pragma unroll
            for (int k = 0; k < kGroupLines; ++k) {
                const uint32_t v4[4] = {ch[rr][k].x, ch[rr][k].y, ch[rr][k].z, ch[rr][k].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t at = a0 + (uint32_t)(k * kLineIds + 4 * q + t);
                    const bool in = at >= s0 && at < s1;
                    below += ((in && v4[t] < glo) ? 1u : 0u) + ((in && v4[t] < ghi) ? 0x10000u : 0u);
                }
            }
            below += (uint32_t)__shfl_xor((int)below, 1, kWave);
            below += (uint32_t)__shfl_xor((int)below, 2, kWave);
            below += (uint32_t)__shfl_xor((int)below, 4, kWave);
            const uint32_t pos = s0 + (below & 0xffffu), end = s0 + (below >> 16);
            // the first 4 continuations are the list entries pos .. pos+3 (below `end`): whoever holds one writes it to the owner's LDS row
#pragma unroll
            for (int k = 0; k < kGroupLines; ++k) {
                const uint32_t v4[4] = {ch[rr][k].x, ch[rr][k].y, ch[rr][k].z, ch[rr][k].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t at = a0 + (uint32_t)(k * kLineIds + 4 * q + t);
                    if (at >= pos && at < end && at - pos < (uint32_t)kHeadSlots && at < s1) s_head[w][e][at - pos] = v4[t];
                }
            }
            if (q == 0 && s1 > s0) s_pc[w][e] = make_uint2(pos, end);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (!live) return;
    uint32_t pos = 0;
    int32_t c = 0;
    uint4 h = make_uint4(0u, 0u, 0u, 0u);
    if (need) {
        const uint2 pe = s_pc[w][l];
        h = *reinterpret_cast<const uint4*>(s_head[w][l]);
        pos = pe.x;
        uint32_t end = pe.y;
        const int have = (int)(end - pos) < kHeadSlots ? (int)(end - pos) : kHeadSlots;      // head slots filled from the lines read above
        list_tail(ids_by_tail, (my_s0 & ~(uint32_t)(kLineIds - 1)) + (uint32_t)(kGroupLines * kLineIds), my_s1, my_glo, my_ghi, have, pos, end, c, h);
    }
    store_stream(first_pos + i, pos);
    store_stream(count + i, c);
    if (source) store_stream_u4(head4 + i * 4, h);
}

// ------------------------------------------------------------------ line-graph count
// (the out-degree of the head node is read as rowptr[v + 1] - rowptr[v] with ONE load: a separate degree array is a second random
//  line per edge — 0.38 -> 0.17 ms per launch at 1.9e7 edges)
__global__ __launch_bounds__(kBlock) void k_linegraph_count(const int64_t* __restrict__ head, int64_t n_edges, int64_t num_nodes,
                                                           const uint32_t* __restrict__ rowptr,
                                                           uint32_t* __restrict__ first_pos, int32_t* __restrict__ count,
                                                           int64_t e_begin, int64_t e_end, int64_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t v = head[e];
    uint32_t pos = 0;
    int32_t c = 0;
    if (v < 0 || v >= num_nodes) {
        atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex);
    } else if (e >= e_begin && e < e_end) {          // edge-range shard: only the own edges are sources
        struct __attribute__((packed, aligned(4))) Bounds { uint32_t begin, end; };
        const Bounds bounds = *reinterpret_cast<const Bounds*>(rowptr + v);       // one 8-byte load at a 4-byte aligned address
        pos = bounds.begin;
        c = (int32_t)(bounds.end - bounds.begin);
    }
    first_pos[e] = pos;
    count[e] = c;
}

// ------------------------------------------------------------------ fill: load-balanced expansion
// Source s owns output slots [offset[s], offset[s+1]).  Slot p of source s, rank r = p - offset[s]:
//   out[0][p] = s ;  out[1][p] = kList ? list[first_pos[s] + r] : first_pos[s] + r
//
// The unit of work is ONE WAVE and 512 consecutive output slots; waves never wait for each other (no workgroup
// barrier: measured 2x faster than a 2048-slot workgroup tile, whose five barriers serialised on the slowest wave).
//   (scan)          the offset scan of the count phase also records, per tile, the source owning the tile's first slot
//                   (pp_scan.hip, `slot_owner`): no separate search kernel.
//   k_expand        the wave loads the boundaries + first positions of its <= 513 sources with 9 back-to-back
//                   coalesced loads (a run-time loop here costs one HBM round trip per iteration), marks every
//                   non-empty source at its first slot in an LDS slot array, spreads the marks with an inclusive
//                   running maximum (8 slots per lane + wave shuffles), transposes (source, position) through LDS
//                   so that each lane owns slot PAIRS, gathers the list entries and stores 16 bytes per lane and row.
constexpr int kWaveTile = 512;                       // output slots per wave
constexpr int kWaveCap = kWaveTile + 1;              // sources whose boundaries are staged (else: per-slot search)
constexpr int kStageIters = (kWaveCap + 1 + kWave - 1) / kWave;

struct alignas(16) I64x2 { int64_t a, b; };
struct alignas(16) I32x4 { int32_t x, y, z, w; };

// kList: the second row comes from `list` (temporal lift) instead of being the position itself (line-graph lift).  With
// `head4` (the first kHead continuations of every source, written by the count kernel while it had their cache lines in
// hand) the common case r < kHead is a near-sequential 4-byte read instead of a random gather into the node lists.
constexpr int kHead = 4;

template <bool kList>
__global__ __launch_bounds__(kBlock) void k_expand(const int64_t* __restrict__ offset, const uint32_t* __restrict__ first_pos,
                                                  const uint32_t* __restrict__ list, const uint32_t* __restrict__ head4, int64_t n_src,
                                                  int64_t total, int64_t id_offset, const int64_t* __restrict__ tile_src,
                                                  int64_t* __restrict__ out) {
    __shared__ int32_t s_rel_all[kWavesPerBlock][kWaveCap + 3];                                   // boundaries relative to the tile start
    __shared__ __attribute__((aligned(16))) uint32_t s_pos_all[kWavesPerBlock][kWaveCap + 3];    // first positions, later slot -> position
    __shared__ __attribute__((aligned(16))) int32_t s_src_all[kWavesPerBlock][kWaveTile];        // slot -> local source index
    const int w = wave_id(), l = lane_id();
    int32_t* s_rel = s_rel_all[w];
    uint32_t* s_pos = s_pos_all[w];
    int32_t* s_src = s_src_all[w];
    // tiles are taken from the END of the output first: the stretches of empty sources (slow path below) sit at the
    // end of a temporal stream, and starting them first keeps them off the kernel's tail
    const int64_t tile = ((int64_t)gridDim.x - 1 - blockIdx.x) * kWavesPerBlock + w;
    const int64_t p0 = tile * kWaveTile;
    if (p0 >= total) return;
    const int64_t p1 = p0 + kWaveTile < total ? p0 + kWaveTile : total;
    int64_t s_first, s_beyond;          // boundaries s_first .. s_beyond + 1 bracket every slot of the tile
    if (tile_src) {
        s_first = tile_src[tile];
        s_beyond = tile_src[tile + 1];
    } else {
        s_first = upper_bound_dev<int64_t, int64_t>(offset, 0, n_src + 1, p0) - 1;
        s_beyond = upper_bound_dev<int64_t, int64_t>(offset, s_first, n_src + 1, p1 - 1) - 1;
    }
    const bool staged = (s_beyond - s_first + 1) <= kWaveCap;       // else: a sea of empty sources, resolve per slot
    int64_t src[8];
    uint32_t at2[8];
    if (staged) {
        const int n_bound = (int)(s_beyond - s_first + 1);
        const int64_t rel0 = offset[s_first] - p0;          // <= 0, exact (the first source may start before the tile)
        int64_t g_off[kStageIters];
        uint32_t g_pos[kStageIters];
#pragma unroll
        for (int it = 0; it < kStageIters; ++it) {
            const int k = it * kWave + l;
            g_off[it] = k <= n_bound ? offset[s_first + k] : 0;
            g_pos[it] = k < n_bound ? first_pos[s_first + k] : 0u;
        }
#pragma unroll
        for (int it = 0; it < kStageIters; ++it) {
            const int k = it * kWave + l;
            if (k <= n_bound) {
                const int64_t rel = g_off[it] - p0;
                s_rel[k] = rel < 0 ? -1 : (rel > kWaveTile ? kWaveTile + 1 : (int32_t)rel);
                s_pos[k] = g_pos[it];
            }
        }
        *(I32x4*)(s_src + 8 * l) = I32x4{0, 0, 0, 0};
        *(I32x4*)(s_src + 8 * l + 4) = I32x4{0, 0, 0, 0};
        __builtin_amdgcn_wave_barrier();
        // every non-empty source that starts inside the tile marks its first slot with its index ...
#pragma unroll
        for (int it = 0; it < kStageIters; ++it) {
            const int k = it * kWave + l;
            if (k >= 1 && k < n_bound) {
                const int rel = s_rel[k];
                if (rel < kWaveTile && s_rel[k + 1] > rel) s_src[rel] = k;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ... and an inclusive running maximum spreads it over the source's slots (lane l: slots 8l .. 8l+7)
        const I32x4 lo4 = *(const I32x4*)(s_src + 8 * l), hi4 = *(const I32x4*)(s_src + 8 * l + 4);
        int kk[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 1; j < 8; ++j) kk[j] = kk[j] > kk[j - 1] ? kk[j] : kk[j - 1];
        int upto = kk[7];
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(upto, d, kWave);
            if (l >= d) upto = o > upto ? o : upto;
        }
        int before = __shfl_up(upto, 1, kWave);
        if (l == 0) before = 0;
        uint32_t at[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kk[j] > before ? kk[j] : before;
            const int q = 8 * l + j;
            kk[j] = k;
            const uint32_t rank = (uint32_t)(k == 0 ? (int64_t)q - rel0 : (int64_t)(q - s_rel[k]));
            at[j] = (kList && head4) ? rank : s_pos[k] + rank;          // with head4 the slot carries its RANK, else its position
        }
        __builtin_amdgcn_wave_barrier();                   // all reads of s_pos done: reuse it as slot -> position / rank
        *(I32x4*)(s_src + 8 * l) = I32x4{kk[0], kk[1], kk[2], kk[3]};
        *(I32x4*)(s_src + 8 * l + 4) = I32x4{kk[4], kk[5], kk[6], kk[7]};
        *(I32x4*)(s_pos + 8 * l) = I32x4{(int)at[0], (int)at[1], (int)at[2], (int)at[3]};
        *(I32x4*)(s_pos + 8 * l + 4) = I32x4{(int)at[4], (int)at[5], (int)at[6], (int)at[7]};
        __builtin_amdgcn_wave_barrier();
        // transposed read: lane l owns slot pairs (2u, 2u+1), u = j * 64 + l
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 2 * (u * kWave + l);
            const int2 k2 = *(const int2*)(s_src + q);
            const uint2 a2 = *(const uint2*)(s_pos + q);
            src[2 * u] = s_first + k2.x;
            src[2 * u + 1] = s_first + k2.y;
            at2[2 * u] = a2.x;
            at2[2 * u + 1] = a2.y;
        }
    } else {
        // More than 513 sources for 512 slots: a stretch of (mostly) empty sources, e.g. the end of a stream whose
        // events have no continuation left.  Same marking + running maximum, but the boundaries are streamed
        // through registers (4 independent coalesced loads per step) instead of being staged in LDS, and the two
        // per-slot lookups go to global memory.
        const int64_t n_bound = s_beyond - s_first + 1;
        *(I32x4*)(s_src + 8 * l) = I32x4{0, 0, 0, 0};
        *(I32x4*)(s_src + 8 * l + 4) = I32x4{0, 0, 0, 0};
        __builtin_amdgcn_wave_barrier();
        // an extremely sparse result (far more sources than output slots: tens of thousands of empty sources per tile) is
        // cheaper to resolve with one binary search per slot than by streaming every boundary past one wave
        const bool search = n_bound > 16384;
        if (search) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t p = p0 + 8 * l + j;
                if (p < p1) s_src[8 * l + j] = (int32_t)(upper_bound_dev<int64_t, int64_t>(offset, s_first, s_beyond + 2, p) - 1 - s_first);
            }
        }
        for (int64_t base = 1; base < n_bound && !search; base += 4 * kWave) {
            int64_t o[4], o1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t k = base + c * kWave + l;
                o[c] = k < n_bound ? offset[s_first + k] : 0;
                o1[c] = k < n_bound ? offset[s_first + k + 1] : 0;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t k = base + c * kWave + l;
                const int64_t rel = o[c] - p0;
                if (k < n_bound && o1[c] > o[c] && rel < kWaveTile) s_src[rel] = (int32_t)k;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const I32x4 lo4 = *(const I32x4*)(s_src + 8 * l), hi4 = *(const I32x4*)(s_src + 8 * l + 4);
        int kk[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 1; j < 8; ++j) kk[j] = kk[j] > kk[j - 1] ? kk[j] : kk[j - 1];
        int upto = kk[7];
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(upto, d, kWave);
            if (l >= d) upto = o > upto ? o : upto;
        }
        int before = __shfl_up(upto, 1, kWave);
        if (l == 0) before = 0;
        uint32_t at[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kk[j] > before ? kk[j] : before;
            const int64_t p = p0 + 8 * l + j;
            kk[j] = k;
            const uint32_t rank = p < p1 ? (uint32_t)(p - offset[s_first + k]) : 0u;
            at[j] = (kList && head4) ? rank : (p < p1 ? first_pos[s_first + k] + rank : 0u);
        }
        *(I32x4*)(s_src + 8 * l) = I32x4{kk[0], kk[1], kk[2], kk[3]};
        *(I32x4*)(s_src + 8 * l + 4) = I32x4{kk[4], kk[5], kk[6], kk[7]};
        *(I32x4*)(s_pos + 8 * l) = I32x4{(int)at[0], (int)at[1], (int)at[2], (int)at[3]};
        *(I32x4*)(s_pos + 8 * l + 4) = I32x4{(int)at[4], (int)at[5], (int)at[6], (int)at[7]};
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 2 * (u * kWave + l);
            const int2 k2 = *(const int2*)(s_src + q);
            const uint2 a2 = *(const uint2*)(s_pos + q);
            src[2 * u] = s_first + k2.x;
            src[2 * u + 1] = s_first + k2.y;
            at2[2 * u] = a2.x;
            at2[2 * u + 1] = a2.y;
        }
    }
    int64_t dst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t p = p0 + 2 * ((j >> 1) * kWave + l) + (j & 1);
        int64_t v;
        if (!kList) {
            v = (int64_t)at2[j];
        } else if (p >= p1) {
            v = 0;
        } else if (head4) {
            const uint32_t rank = at2[j];
            v = rank < (uint32_t)kHead ? (int64_t)head4[src[j] * kHead + rank] : (int64_t)list[first_pos[src[j]] + rank];
        } else {
            v = (int64_t)list[at2[j]];
        }
        dst[j] = v + id_offset;
    }
    const bool row1_aligned = (total & 1) == 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + 2 * (u * kWave + l);
        if (p + 1 < p1) {
            *(I64x2*)(out + p) = I64x2{src[2 * u] + id_offset, src[2 * u + 1] + id_offset};
            if (row1_aligned) {
                *(I64x2*)(out + total + p) = I64x2{dst[2 * u], dst[2 * u + 1]};
            } else {
                out[total + p] = dst[2 * u];
                out[total + p + 1] = dst[2 * u + 1];
            }
        } else if (p < p1) {
            out[p] = src[2 * u] + id_offset;
            out[total + p] = dst[2 * u];
        }
    }
}

// ------------------------------------------------------------------ aggregate_node_attributes
template <typename T>
__device__ __forceinline__ T combine(T a, T b, int aggr) {
    switch (aggr) {
        case PP_AGGR_SRC: return a;
        case PP_AGGR_DST: return b;
        case PP_AGGR_MAX: return a > b ? a : b;
        case PP_AGGR_MUL: return a * b;
        default: return a + b;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_edge_attr(const int64_t* __restrict__ edge_index, int64_t n_edges, const T* __restrict__ attr,
                                                     int64_t num_nodes, int64_t width, int aggr, T* __restrict__ out,
                                                     int64_t* __restrict__ status) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n_edges * width) return;
    const int64_t e = idx / width, c = idx - e * width;
    int64_t a = edge_index[e], b = edge_index[n_edges + e];
    if (a < 0 || a >= num_nodes || b < 0 || b >= num_nodes) {
        atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex);
        return;
    }
    T va = aggr == PP_AGGR_DST ? (T)0 : attr[a * width + c];
    T vb = aggr == PP_AGGR_SRC ? (T)0 : attr[b * width + c];
    out[idx] = combine<T>(va, vb, aggr);
}

// order-(k+1) instance sequences: row e = node_sequence[tail(e)] ++ last element of node_sequence[head(e)]
__global__ __launch_bounds__(kBlock) void k_extend_rows(const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                       const int64_t* __restrict__ rows, int64_t n_rows, int k,
                                                       int64_t* __restrict__ out, int64_t* __restrict__ status) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int kk = k + 1;
    if (idx >= n_edges * kk) return;
    const int64_t e = idx / kk;
    const int c = (int)(idx - e * kk);
    const int64_t r = c < k ? edge_index[e] : edge_index[n_edges + e];
    if (r < 0 || r >= n_rows) {
        atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex);
        return;
    }
    out[idx] = rows[r * k + (c < k ? c : k - 1)];
}

// out[i, :k] = rows[idx[i], :],  out[i, k] = suffix[i]   (rows [n_rows, k], out [n, k+1])
__global__ __launch_bounds__(kBlock) void k_gather_concat(const int64_t* __restrict__ rows, int64_t n_rows, int k, const int64_t* __restrict__ idx,
                                                         const int64_t* __restrict__ suffix, int64_t n, int64_t* __restrict__ out,
                                                         int64_t* __restrict__ status) {
    const int64_t at = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int kk = k + 1;
    if (at >= n * kk) return;
    const int64_t i = at / kk;
    const int c = (int)(at - i * kk);
    if (c == k) { out[at] = suffix[i]; return; }
    const int64_t r = idx[i];
    if (r < 0 || r >= n_rows) { atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex); return; }
    out[at] = rows[r * k + c];
}

// ------------------------------------------------------------------ workspace layouts
struct LiftWs {
    int64_t* result;        // [2]: {total, status}
    int64_t* offset;        // [n_src + 1]
    uint32_t* first_pos;    // [n_src]
    int32_t* count;         // [n_src]
    uint32_t* rowptr;       // [num_nodes + 1]
    uint32_t* ids;          // temporal: event ids grouped by tail [n_src]
    uint32_t* keys;         // temporal: tail keys [n_src];  line graph: outdeg (as int32) [num_nodes]
    uint32_t* sorted_keys;  // temporal only [n_src]
    uint32_t* head4;        // temporal only [n_src * 4]: the first 4 continuations of every event
    uint32_t* marks;        // temporal only [n_src / 64 + 2]: window end of the first event of every wave (k_window_marks)
    int64_t* tile_src;      // first source of every 512-slot output tile (written by the offset scan) [tile_cap + 1]
    int64_t tile_cap;
    void* scratch;          // sort / scan workspace
    size_t scratch_bytes;
    size_t total_bytes;
};

static LiftWs carve_lift(void* ws, int64_t n_src, int64_t num_nodes, bool temporal) {
    Arena a(ws, (size_t)-1);
    LiftWs w;
    w.result = a.take<int64_t>(2);
    w.offset = a.take<int64_t>(n_src + 1);
    w.first_pos = a.take<uint32_t>(n_src);
    w.count = a.take<int32_t>(n_src);
    w.rowptr = a.take<uint32_t>(num_nodes + 1);
    w.ids = temporal ? a.take<uint32_t>(n_src + 4) : nullptr;          // + 4: k_temporal_count reads it in aligned 16-byte pieces
    w.keys = a.take<uint32_t>(temporal ? n_src : num_nodes);
    w.sorted_keys = temporal ? a.take<uint32_t>(n_src) : nullptr;
    w.head4 = temporal ? a.take<uint32_t>(n_src * 4) : nullptr;
    w.marks = temporal ? a.take<uint32_t>(n_src / kWave + 2) : nullptr;
    w.tile_cap = n_src / 16 > 65536 ? n_src / 16 : 65536;       // covers results up to 32x the number of sources
    w.tile_src = a.take<int64_t>(w.tile_cap + 1);
    size_t sb = scan_ws_bytes(n_src > num_nodes ? n_src : num_nodes);
    if (temporal) { size_t s2 = sort_ws_bytes(n_src, 4); sb = s2 > sb ? s2 : sb; }
    w.scratch_bytes = sb;
    w.scratch = a.take<char>((int64_t)sb);
    w.total_bytes = a.used;
    return w;
}

// the per-node event lists and per-event continuation windows pp_temporal_count left in its workspace (pp_multiorder.hip builds on them)
TemporalLists temporal_lists(void* ws, int64_t m, int64_t num_nodes) {
    const LiftWs w = carve_lift(ws, m, num_nodes, true);
    return TemporalLists{w.ids, w.rowptr, w.first_pos, w.count, w.result, w.total_bytes};
}

// the per-tile start sources come from the count phase's scan when the result is at most 32x the source count; beyond that every
// wave searches for its own start (amortised by the amount of output per source)
template <bool kList>
static int launch_expand(const LiftWs& w, int64_t n_src, int64_t total, int64_t id_offset, int64_t* out, hipStream_t st) {
    const int64_t n_tiles = ceil_div(total, kWaveTile);
    int64_t* tile_src = n_tiles <= w.tile_cap ? w.tile_src : nullptr;          // (written by the count phase's scan)
    k_expand<kList><<<(unsigned)ceil_div(n_tiles, kWavesPerBlock), kBlock, 0, st>>>(w.offset, w.first_pos, kList ? w.ids : nullptr,
                                                                                       kList ? w.head4 : nullptr, n_src, total, id_offset,
                                                                                       tile_src, out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <typename TimeT, int kMode>
static int launch_temporal_count_mode(unsigned grid, hipStream_t st, const int64_t* head, const TimeT* time, int64_t m, int64_t n_own, int64_t n,
                                      int64_t di, double df, const LiftWs& w) {
    const int64_t n_waves = ceil_div(m, kWave);
    k_window_marks<TimeT, kMode><<<(unsigned)ceil_div(n_waves + 1, kBlock), kBlock, 0, st>>>(time, m, n_waves, di, df, w.marks);
    PP_LAUNCH_CHECK();
    k_temporal_count<TimeT, kMode><<<grid, kBlock, 0, st>>>(head, time, m, n_own, n, di, df, w.rowptr, w.ids, w.marks, w.first_pos, w.count,
                                                            w.head4, w.result + 1);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <typename TimeT>
static int launch_temporal_count(int delta_kind, unsigned grid, hipStream_t st, const int64_t* head, const TimeT* time, int64_t m,
                                 int64_t n_own, int64_t n, int64_t di, double df, const LiftWs& w);

template <>
int launch_temporal_count<int64_t>(int delta_kind, unsigned grid, hipStream_t st, const int64_t* head, const int64_t* time, int64_t m,
                                   int64_t n_own, int64_t n, int64_t di, double df, const LiftWs& w) {
    if (delta_kind == PP_DELTA_I64) return launch_temporal_count_mode<int64_t, 0>(grid, st, head, time, m, n_own, n, di, df, w);
    if (delta_kind == PP_DELTA_F32) return launch_temporal_count_mode<int64_t, 1>(grid, st, head, time, m, n_own, n, di, df, w);
    return launch_temporal_count_mode<int64_t, 2>(grid, st, head, time, m, n_own, n, di, df, w);
}
template <>
int launch_temporal_count<double>(int, unsigned grid, hipStream_t st, const int64_t* head, const double* time, int64_t m, int64_t n_own,
                                  int64_t n, int64_t di, double df, const LiftWs& w) {
    return launch_temporal_count_mode<double, 0>(grid, st, head, time, m, n_own, n, di, df, w);
}

}  // namespace pp

using namespace pp;

extern "C" {

// ---------------------------------------------------------------- temporal lift
size_t pp_temporal_ws_bytes(int64_t m, int64_t num_nodes) { return carve_lift(nullptr, m, num_nodes, true).total_bytes; }

static int temporal_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t n_own, int64_t num_nodes,
                          int delta_kind, int64_t delta_i, double delta_f, void* ws, size_t ws_bytes, bool offsets, hipStream_t st);

int pp_temporal_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t n_own, int64_t num_nodes,
                      int delta_kind, int64_t delta_i, double delta_f, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return temporal_count(edge_index, time, time_dtype, m, n_own, num_nodes, delta_kind, delta_i, delta_f, ws, ws_bytes, true, (hipStream_t)stream);
}

/* see include/pathpyg_amd.h: the lists and windows of pp_temporal_count without the offsets of the fill */
int pp_temporal_windows(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i,
                        double delta_f, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return temporal_count(edge_index, time, time_dtype, m, m, num_nodes, delta_kind, delta_i, delta_f, ws, ws_bytes, false, (hipStream_t)stream);
}

static int temporal_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t n_own, int64_t num_nodes,
                          int delta_kind, int64_t delta_i, double delta_f, void* ws, size_t ws_bytes, bool offsets, hipStream_t st) {
    PP_REQUIRE(m >= 0 && num_nodes >= 0, PP_ERR_ARG, "pp_temporal_count: negative size");
    if (n_own < 0 || n_own > m) n_own = m;
    PP_REQUIRE(m < (int64_t)0x7fffffff && num_nodes < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_temporal_count: m or num_nodes >= 2^31");
    PP_REQUIRE(time_dtype == PP_I64 || time_dtype == PP_F64, PP_ERR_ARG, "pp_temporal_count: time must be int64 or float64");
    PP_REQUIRE(delta_kind >= PP_DELTA_I64 && delta_kind <= PP_DELTA_F64, PP_ERR_ARG, "pp_temporal_count: bad delta kind");
    LiftWs w = carve_lift(ws, m, num_nodes, true);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_temporal_count: workspace too small");
    PP_HIP(hipMemsetAsync(w.result, 0, 2 * sizeof(int64_t), st));
    if (m == 0) return PP_OK;
    const int64_t* tail = edge_index;
    const int64_t* head = edge_index + m;
    const unsigned grid = (unsigned)ceil_div(m, kBlock);
    // 1. CSR of event ids by tail node
    k_tail_keys<<<grid, kBlock, 0, st>>>(tail, m, num_nodes, w.keys, w.result + 1);
    PP_LAUNCH_CHECK();
    const int key_bits = bits_for((uint64_t)(num_nodes > 0 ? num_nodes - 1 : 0));
    int rc = sort_pairs<uint32_t>(w.keys, nullptr, w.sorted_keys, w.ids, m, 0, key_bits, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_rowptr_from_sorted<uint32_t><<<(unsigned)ceil_div(m + 1, kBlock), kBlock, 0, st>>>(w.sorted_keys, m, num_nodes, w.rowptr);
    PP_LAUNCH_CHECK();
    // 2. per-event continuation window
    if (time_dtype == PP_I64)
        rc = launch_temporal_count<int64_t>(delta_kind, grid, st, head, (const int64_t*)time, m, n_own, num_nodes, delta_i, delta_f, w);
    else
        rc = launch_temporal_count<double>(delta_kind, grid, st, head, (const double*)time, m, n_own, num_nodes, delta_i, delta_f, w);
    if (rc != PP_OK) return rc;
    if (!offsets) return PP_OK;
    // 3. offsets + total
    return exclusive_scan<int32_t, int64_t>(w.count, m, w.offset, true, w.result, w.scratch, w.scratch_bytes, st, w.tile_src, kWaveTile, w.tile_cap);
}

int pp_temporal_fill(int64_t m, int64_t num_nodes, int64_t total, int64_t id_offset, int64_t* out, void* ws, size_t ws_bytes,
                     pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    LiftWs w = carve_lift(ws, m, num_nodes, true);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_temporal_fill: workspace too small");
    if (total <= 0) return PP_OK;
    return launch_expand<true>(w, m, total, id_offset, out, st);
}

// ---------------------------------------------------------------- line-graph lift
size_t pp_linegraph_ws_bytes(int64_t n_edges, int64_t num_nodes) { return carve_lift(nullptr, n_edges, num_nodes, false).total_bytes; }

int pp_linegraph_count(const int64_t* edge_index, int64_t n_edges, int64_t e_begin, int64_t e_end, int64_t num_nodes, void* ws,
                       size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_edges >= 0 && num_nodes >= 0, PP_ERR_ARG, "pp_linegraph_count: negative size");
    PP_REQUIRE(e_begin >= 0 && e_begin <= e_end && e_end <= n_edges, PP_ERR_ARG, "pp_linegraph_count: edge range [%lld, %lld) outside [0, %lld]",
               (long long)e_begin, (long long)e_end, (long long)n_edges);
    PP_REQUIRE(n_edges < (int64_t)0x7fffffff && num_nodes < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_linegraph_count: E or N >= 2^31");
    LiftWs w = carve_lift(ws, n_edges, num_nodes, false);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_linegraph_count: workspace too small");
    PP_HIP(hipMemsetAsync(w.result, 0, 2 * sizeof(int64_t), st));
    if (n_edges == 0) return PP_OK;
    int32_t* outdeg = (int32_t*)w.keys;
    int rc = histogram<int64_t>(edge_index, n_edges, num_nodes, outdeg, st);      // degree(edge_index[0])
    if (rc != PP_OK) return rc;
    rc = exclusive_scan<int32_t, int32_t>(outdeg, num_nodes, (int32_t*)w.rowptr, true, nullptr, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_linegraph_count<<<(unsigned)ceil_div(n_edges, kBlock), kBlock, 0, st>>>(edge_index + n_edges, n_edges, num_nodes, w.rowptr,
                                                                             w.first_pos, w.count, e_begin, e_end, w.result + 1);
    PP_LAUNCH_CHECK();
    return exclusive_scan<int32_t, int64_t>(w.count, n_edges, w.offset, true, w.result, w.scratch, w.scratch_bytes, st, w.tile_src, kWaveTile, w.tile_cap);
}

int pp_linegraph_fill(int64_t n_edges, int64_t num_nodes, int64_t total, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    LiftWs w = carve_lift(ws, n_edges, num_nodes, false);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_linegraph_fill: workspace too small");
    if (total <= 0) return PP_OK;
    return launch_expand<false>(w, n_edges, total, 0, out, st);
}

// {total, status} written by the last *_count on this workspace (two int64 at the start of the workspace)
const int64_t* pp_lift_result_ptr(void* ws) { return (const int64_t*)ws; }

// ---------------------------------------------------------------- aggregate_node_attributes
int pp_edge_attr(const int64_t* edge_index, int64_t n_edges, const void* attr, int dtype, int64_t num_nodes, int64_t width, int aggr,
                 void* out, int64_t* status, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(aggr >= PP_AGGR_SRC && aggr <= PP_AGGR_ADD, PP_ERR_ARG, "Unknown aggregation method %d", aggr);
    PP_REQUIRE(width >= 1, PP_ERR_ARG, "pp_edge_attr: width must be >= 1");
    PP_HIP(hipMemsetAsync(status, 0, sizeof(int64_t), st));
    const int64_t total = n_edges * width;
    if (total == 0) return PP_OK;
    const unsigned grid = (unsigned)ceil_div(total, kBlock);
    switch (dtype) {
        case PP_I32: k_edge_attr<int32_t><<<grid, kBlock, 0, st>>>(edge_index, n_edges, (const int32_t*)attr, num_nodes, width, aggr, (int32_t*)out, status); break;
        case PP_I64: k_edge_attr<int64_t><<<grid, kBlock, 0, st>>>(edge_index, n_edges, (const int64_t*)attr, num_nodes, width, aggr, (int64_t*)out, status); break;
        case PP_F32: k_edge_attr<float><<<grid, kBlock, 0, st>>>(edge_index, n_edges, (const float*)attr, num_nodes, width, aggr, (float*)out, status); break;
        case PP_F64: k_edge_attr<double><<<grid, kBlock, 0, st>>>(edge_index, n_edges, (const double*)attr, num_nodes, width, aggr, (double*)out, status); break;
        default: PP_REQUIRE(false, PP_ERR_ARG, "pp_edge_attr: unsupported dtype %d", dtype);
    }
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ---------------------------------------------------------------- prefix-row gather + suffix column
int pp_gather_concat(const int64_t* rows, int64_t n_rows, int k, const int64_t* idx, const int64_t* suffix, int64_t n, int64_t* out,
                     int64_t* status, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(k >= 1 && n >= 0, PP_ERR_ARG, "pp_gather_concat: bad shape");
    PP_HIP(hipMemsetAsync(status, 0, sizeof(int64_t), st));
    if (n == 0) return PP_OK;
    k_gather_concat<<<(unsigned)ceil_div(n * (k + 1), kBlock), kBlock, 0, st>>>(rows, n_rows, k, idx, suffix, n, out, status);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ---------------------------------------------------------------- node-sequence extension
int pp_extend_node_sequence(const int64_t* edge_index, int64_t n_edges, const int64_t* rows, int64_t n_rows, int k, int64_t* out,
                            int64_t* status, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(k >= 1, PP_ERR_ARG, "pp_extend_node_sequence: k must be >= 1");
    PP_HIP(hipMemsetAsync(status, 0, sizeof(int64_t), st));
    const int64_t total = n_edges * (k + 1);
    if (total == 0) return PP_OK;
    k_extend_rows<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, st>>>(edge_index, n_edges, rows, n_rows, k, out, status);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
