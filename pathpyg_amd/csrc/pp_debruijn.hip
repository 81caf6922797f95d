// pathpyg_amd — the order-2 De Bruijn model of a temporal event stream, built NODE BY NODE (round 4).
//
// Reference work replaced, fused into one count -> read-back -> fill pair (paths relative to the pathpyG repository root):
//   MultiOrderModel.from_temporal_graph(g, delta, max_order=2)   src/pathpyG/core/multi_order_model.py:124-192
//     lift_order_temporal                                        src/pathpyG/algorithms/temporal.py:17-54
//     aggregate_edge_index (layers 1 and 2)                      src/pathpyG/algorithms/lift_order.py:109-152
//   gcn_norm of both layers + the bipartite "last" index         src/pathpyG/nn/dbgnn.py:104-114 (through GCNConv), utils/dbgnn.py:10-46
//
// The generic path (pp_lift.hip -> pp_aggregate.hip -> pp_gcn_plan) materialises the event graph ([2, E2] int64), sorts it twice
// (coalesce by (source, destination), then the plan's destination grouping) and sorts the events twice more (tail lists, layer 1): five
// global radix sorts and ~1.3e8 random accesses per step of the headline stream.  Here the structure of a De Bruijn graph does the work:
// every order-2 edge (a,b) -> (b,c) has a MIDDLE NODE b, and everything about it is decided by b's in-events (., b, t) and out-events
// (b, ., t).  Two sorts of the m events (by tail; then that sequence by head, so that a node's in-events arrive ordered by (source, time);
// 32-bit keys) put both lists of every node next to each other; after that ONE WAVE PER NODE works on ~20 + ~20 events in registers:
//   k_db2_out   out-events of b ranked by (c, time): the distinct successors c = the order-2 nodes (b, .) = the first-order out-edges of b
//               (block sizes -> scan -> ids), their weights (run lengths / left-to-right sums), the out-events stored in that order;
//   k_db2_mid   in-events of b ranked by (a, time); for every run (a, b) = source node u and every instance i of it ONE ballot over the
//               out-events gives the continuations t_i < t_j <= t_i + delta (the window test of temporal.py:43 in torch's promoted dtype),
//               popcounts against the successor runs give the merged weights of the edges u -> (b, c): the destination-major CSR of the
//               rows (b, .) is written contiguously, the source-major CSR row of u at its scanned offset.  First pass: counts, weighted
//               in-degrees (left to right in ascending source order, as the generic plan sums them), E2; second pass: the normalised
//               coefficients d^-1/2 w d^-1/2 of both CSRs, the first-order graph's destination-major CSR and the bipartite index.
// The event graph never exists in HBM; E2 (the number of lifted instance pairs) is the sum of the popcounts.  Results are IDENTICAL to the
// generic path (same ids, same order inside every row, same fp32 sums) — tests/test_gpu_builder.py compares them array by array.
// HUB NODES (round 5): a node with more than 64 in- or out-events does not fit one wave.  pp_debruijn2_lists classifies the nodes after the two
// sorts and reports {hub nodes, their out-events, tasks, part columns} (the caller reads them back while the out-side kernel of the other
// nodes runs); such nodes are then worked on by k_db2_hub — see "hub nodes" below — and every other node stays on the one-wave kernels: no
// whole-stream fallback.  Node-range partitions (pp_debruijn2_part_*) still report hubs through status bit 2 (kDb2Overflow) and their
// caller falls back.  Event weights float32 or absent (unit weights: the reference's default torch.ones).
#include "pp_internal.h"
#include "pp_window.h"

namespace pp {

constexpr int64_t kDb2BadIndex = 1, kDb2Unsorted = 2, kDb2Overflow = 4;
constexpr int kHubChunk = 256;                        // in-events per hub task (one wave) ..
// .. except for a node with FEW in-events and a LONG out-list: a task of k_db2_hubx streams the out-list's window once per in-run, so the work of
// one task grows with the out-list while the node has one or two tasks (a node pair that carries a third of a 2*10^6-event stream: 20 in-runs
// over a 6.7*10^5-event out-list in ONE wave, 6.5 ms per pass).  Such a node gets smaller chunks — more tasks for the same in-events; the per-task
// partial results add up as before (an order-2 edge belongs to exactly one in-run).  At most 256 tasks per such node, and there are at most
// m / kHubLongOut of them: db2_task_cap.
constexpr int64_t kHubLongOut = 4096, kHubFewIn = 4096;
__host__ __device__ __forceinline__ int hub_chunk(int64_t ni, int64_t no) {
    if (no <= kHubLongOut || ni > kHubFewIn) return kHubChunk;
    int c = kHubChunk;
    for (int64_t x = kHubLongOut; x < no && c > 1; x <<= 1) c >>= 1;
    const int least = ni <= 256 ? 1 : 16;
    return c < least ? least : c;
}
constexpr uint8_t kHubOut = 1, kHubIn = 2;            // hub_flag bits: more than 64 out-events / in-events
constexpr uint32_t kDb2Foreign = 0xFFFFFFFEu;         // order-2 node of an event whose source node another rank owns: numbered on the head side

struct alignas(16) Db2Rec {
    uint64_t t;       // timestamp bits (int64 or float64)
    uint32_t a;       // src
    uint32_t c;       // dst
};

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ float rl_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int lane) {
    return ((uint64_t)rl_u((uint32_t)(v >> 32), lane) << 32) | (uint64_t)rl_u((uint32_t)v, lane);
}
// value pushed to lane `dest` (a permutation of the lanes)
__device__ __forceinline__ uint32_t push_u(int dest, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)v); }
__device__ __forceinline__ uint64_t push_u64(int dest, uint64_t v) {
    return ((uint64_t)push_u(dest, (uint32_t)(v >> 32)) << 32) | (uint64_t)push_u(dest, (uint32_t)v);
}
__device__ __forceinline__ float push_f(int dest, float v) { return __builtin_bit_cast(float, push_u(dest, __builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ uint64_t lanes_below(int l) { return (1ull << l) - 1ull; }            // l in [0, 63]
__device__ __forceinline__ uint64_t lanes_upto(int l) { return (2ull << l) - 1ull; }             // lanes 0 .. l (l = 63: all)
__device__ __forceinline__ float inv_sqrt_deg(float deg) {
    float d = 1.0f / sqrtf(deg);                                   // deg^-1/2, inf -> 0 (gcn_norm's masked_fill_)
    return isinf(d) ? 0.0f : d;
}
template <typename TimeT>
__device__ __forceinline__ TimeT time_of(uint64_t bits) { return __builtin_bit_cast(TimeT, bits); }
__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ flat (one thread per event) kernels: every random access of the builder
// lives here, at full memory-level parallelism; the per-node kernels below read and write contiguous ranges only
template <typename TimeT>
__global__ __launch_bounds__(kBlock) void k_db2_keys(const int64_t* __restrict__ ei, const TimeT* __restrict__ time, int64_t m, int64_t n,
                                                    uint32_t* __restrict__ tkeys, Db2Rec* __restrict__ rec, int64_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= m) return;
    int64_t s = ei[e], d = ei[m + e];
    if (s < 0 || s >= n || d < 0 || d >= n) { atomicOr((unsigned long long*)status, (unsigned long long)kDb2BadIndex); s = 0; d = 0; }
    const TimeT t = time[e];
    if (e + 1 < m && time[e + 1] < t) atomicOr((unsigned long long*)status, (unsigned long long)kDb2Unsorted);
    tkeys[e] = (uint32_t)s;
    Db2Rec r;
    r.t = __builtin_bit_cast(uint64_t, t);
    r.a = (uint32_t)s;
    r.c = (uint32_t)d;
    rec[e] = r;
}

// rowptr[v] = first position p with sorted_keys[p] >= v, v in [0, n]
__global__ __launch_bounds__(kBlock) void k_db2_rowptr(const uint32_t* __restrict__ sorted_keys, int64_t m, int64_t n, uint32_t* __restrict__ rowptr) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > m) return;
    const int64_t a = p == 0 ? -1 : (int64_t)sorted_keys[p - 1];
    int64_t b = p == m ? n : (int64_t)sorted_keys[p];
    if (b > n) b = n;
    for (int64_t v = a + 1; v <= b; ++v) rowptr[v] = (uint32_t)p;
}

// out-events in list order (source, time): (successor, time [, weight]) next to each other.  The successors are also the keys of the SECOND
// sort: the list sequence, stably sorted by head node, gives every node its in-events ordered by (source node, time)
__global__ __launch_bounds__(kBlock) void k_db2_gather_out(int64_t m, const uint32_t* __restrict__ tl, const Db2Rec* __restrict__ rec,
                                                          const float* __restrict__ w, uint32_t* __restrict__ oc_t, uint64_t* __restrict__ ot_t,
                                                          float* __restrict__ ow_t) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t e = tl[p];
    const Db2Rec r = rec[e];
    oc_t[p] = r.c;
    ot_t[p] = r.t;
    if (w) ow_t[p] = w[e];
}

struct alignas(16) Db2Src {
    uint64_t t;       // timestamp bits
    uint32_t u;       // order-2 node (source node, head node) of the event
    uint32_t a;       // source node
};

// in-events of every node in (source, time) order: one 16-byte record per event from its position in the out-lists
__global__ __launch_bounds__(kBlock) void k_db2_gather_in(int64_t m, const uint32_t* __restrict__ hl, const Db2Src* __restrict__ src_t,
                                                         const float* __restrict__ ow_t, uint64_t* __restrict__ is_t, uint32_t* __restrict__ is_a,
                                                         uint32_t* __restrict__ is_u, float* __restrict__ is_w) {
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= m) return;
    const uint32_t p = hl[q];
    const Db2Src r = src_t[p];
    is_t[q] = r.t;
    is_a[q] = r.a;
    is_u[q] = r.u;
    if (ow_t) is_w[q] = ow_t[p];
}

// per order-2 node: (weighted degree, start of its source-major row) next to each other — ONE random access per in-event in the gather below
__global__ __launch_bounds__(kBlock) void k_db2_pack_rows(int64_t m, const float* __restrict__ ho_deg, const int32_t* __restrict__ ho_bwd_ptr,
                                                         uint2* __restrict__ row_pack) {
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (u >= m) return;
    row_pack[u] = make_uint2(__float_as_uint(inv_sqrt_deg(ho_deg[u])), (uint32_t)ho_bwd_ptr[u]);
}

// the source-major rows were scattered as (destination, coefficient) pairs: one 8-byte random store per entry instead of two 4-byte ones
__global__ __launch_bounds__(kBlock) void k_db2_unzip(int64_t n, const uint2* __restrict__ pack, int32_t* __restrict__ idx, float* __restrict__ val) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const uint2 r = pack[j];
    idx[j] = (int32_t)r.x;
    val[j] = __uint_as_float(r.y);
}

// ------------------------------------------------------------------ out side: the successors of every node
// kN nodes per wave, one after the other: the loads of all of them are issued before the first is worked on (the per-node work is a
// chain of short dependent steps; 8 waves per SIMD alone do not hide the memory latency under it)
#ifndef PP_DB2_OUT_NODES
#define PP_DB2_OUT_NODES 2
#endif
#ifndef PP_DB2_MID_NODES
#define PP_DB2_MID_NODES 1
#endif
constexpr int kDb2OutNodes = PP_DB2_OUT_NODES;      // nodes per wave of k_db2_out
constexpr int kDb2Nodes = PP_DB2_MID_NODES;         // nodes per wave of k_db2_mid

template <bool kW>
struct Db2OutIn {
    uint32_t p0;
    int cnt;
    uint32_t c;
    uint64_t tb;
    float wv;
};

template <bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_out(int64_t n, int64_t lo, int64_t n_own, const uint32_t* __restrict__ tp, const uint32_t* __restrict__ oc_t,
                                                   const uint64_t* __restrict__ ot_t, const float* __restrict__ ow_t, uint64_t* __restrict__ ot_s,
                                                   uint32_t* __restrict__ oc_s, float* __restrict__ ow_s, uint8_t* __restrict__ ocr_s,
                                                   uint8_t* __restrict__ ocr_t, int32_t* __restrict__ blk, int32_t* __restrict__ fblk,
                                                   uint8_t* __restrict__ fskip, int64_t* __restrict__ status, int hubs_handled) {
    // n nodes from node 0 (one GPU: all of them, all owned; partition shard: all of them, [lo, lo + n_own) owned — the lists of the FOREIGN
    // nodes hold their events into the owned range: their successor runs are the source-major rows of the rank's first-order shard)
    const int64_t node0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * kDb2OutNodes;
    if (node0 >= n) return;
    const int l = lane_id();
    Db2OutIn<kW> in[kDb2OutNodes];
#pragma unroll
    for (int s = 0; s < kDb2OutNodes; ++s) {
        const int64_t node = node0 + s;
        const uint32_t b0 = node < n ? tp[node] : 0u, b1 = node < n ? tp[node + 1] : 0u;
        in[s].p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);          // (wave-uniform by construction: scalar loop bounds, scalar addresses)
        in[s].cnt = __builtin_amdgcn_readfirstlane((int)(b1 - b0));
    }
#pragma unroll
    for (int s = 0; s < kDb2OutNodes; ++s) {
        const bool live = l < in[s].cnt && in[s].cnt <= kWave;
        in[s].c = live ? oc_t[in[s].p0 + l] : 0xFFFFFFFFu;
        in[s].tb = live ? ot_t[in[s].p0 + l] : 0ull;
        in[s].wv = (kW && live) ? ow_t[in[s].p0 + l] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < kDb2OutNodes; ++s) {
        const int64_t node = node0 + s;
        if (node >= n) break;
        const uint32_t p0 = in[s].p0;
        const int cnt = in[s].cnt;
        const int64_t nl = node - lo;
        const bool own = nl >= 0 && nl < n_own;
        const int64_t jloc = own ? nl : (node < lo ? n_own + node : node);          // dense local source space [owned | ids below lo | ids from hi on]
        if (cnt > kWave && hubs_handled) continue;             // (an out-hub: k_db2_hub_out_* rank its out-events and write its block size)
        if (cnt > kWave || cnt == 0) {
            if (l == 0) {
                if (own) blk[nl] = 0;
                if (fblk) { fblk[jloc] = 0; fskip[node] = 0; }
                if (cnt > kWave) atomicOr((unsigned long long*)status, (unsigned long long)kDb2Overflow);
            }
            continue;
        }
        const bool live = l < cnt;
        const uint32_t c = in[s].c;
        int r = 0;
        for (int kk = 0; kk < cnt; ++kk) {
            const uint32_t ck = rl_u(c, kk);
            r += (ck < c || (ck == c && kk < l)) ? 1 : 0;
        }
        const int dest = live ? r : l;                 // a permutation of the lanes: live lanes fill 0 .. cnt-1
        const uint32_t sc = push_u(dest, c);
        const uint64_t st = push_u64(dest, in[s].tb);
        const float sw = kW ? push_f(dest, in[s].wv) : 0.0f;
        const uint32_t prev = (uint32_t)__shfl_up((int)sc, 1, kWave);
        const bool head = live && (l == 0 || sc != prev);
        const uint64_t hm = __ballot(head);
        const int crank = (int)__popcll(hm & lanes_upto(l)) - 1;
        const uint64_t later = hm & ~lanes_upto(l);
        const int end = later ? __ffsll((long long)later) - 1 : cnt;
        const int len = end - l;
        float weight = (float)len;
        if (kW) {
            const int mx = wave_max(head ? len : 0);
            float acc = 0.0f;
            for (int p = 0; p < mx; ++p) {
                const float v = __shfl(sw, (l + p) & (kWave - 1), kWave);
                if (head && p < len) acc += v;         // left to right = instance (time) order, as the segment reduce of the generic coalesce
            }
            weight = acc;
        }
        const int mine = lane_read_i(dest << 2, crank);      // successor rank of the event at LIST position l (its slot holds it after the push)
        if (live) {
            ot_s[p0 + l] = st;
            oc_s[p0 + l] = sc;
            ocr_s[p0 + l] = (uint8_t)crank;
            ocr_t[p0 + l] = (uint8_t)mine;
            ow_s[p0 + l] = head ? weight : 0.0f;
        }
        if (l == 0 && own) blk[nl] = (int32_t)__popcll(hm);
        if (fblk) {
            const uint64_t below = __ballot(head && (int64_t)sc < lo), inside = __ballot(head && (int64_t)sc >= lo && (int64_t)sc < lo + n_own);
            if (l == 0) { fblk[jloc] = (int32_t)__popcll(inside); fskip[node] = (uint8_t)__popcll(below); }
        }
    }
}

// per successor run: the first-order edge (destination + weight per order-2 node, in lexicographic row order)
__global__ __launch_bounds__(kBlock) void k_db2_out_heads(int64_t m, int64_t lo, int64_t n_own, const uint32_t* __restrict__ tp, const uint32_t* __restrict__ tkeys_s,
                                                         const uint32_t* __restrict__ oc_s, const float* __restrict__ ow_s,
                                                         const uint8_t* __restrict__ ocr_s, const int32_t* __restrict__ row_ptr,
                                                         int32_t* __restrict__ fo_bwd_idx, float* __restrict__ fo_w,
                                                         const uint32_t* __restrict__ hoff, const uint32_t* __restrict__ rank_s) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t b = tkeys_s[p];
    const int64_t bl = (int64_t)b - lo;
    if (bl < 0 || bl >= n_own) return;
    const uint32_t p0 = tp[b];
    if (tp[b + 1] - p0 > (uint32_t)kWave) {
        if (rank_s == nullptr) return;                         // (partition shard: overflow node, the caller falls back)
        const uint32_t j = hoff[b] + ((uint32_t)p - p0);       // out-hub: 32-bit successor ranks in the hub scratch
        const uint32_t cr = rank_s[j];
        if (p == p0 || rank_s[j - 1] != cr) {
            const uint32_t u = (uint32_t)row_ptr[bl] + cr;
            fo_bwd_idx[u] = (int32_t)oc_s[p];
            fo_w[u] = ow_s[p];
        }
        return;
    }
    const uint8_t cr = ocr_s[p];
    if (p == p0 || ocr_s[p - 1] != cr) {
        const uint32_t u = (uint32_t)row_ptr[bl] + cr;
        fo_bwd_idx[u] = (int32_t)oc_s[p];
        fo_w[u] = ow_s[p];
    }
}

// per list position: the (time, order-2 node, source) record the head-side gather reads.  `perm` (partition shards): local id of every
// lexicographic row — rows other ranks gather from come first, grouped by that rank (k_db2_apply_perm)
__global__ __launch_bounds__(kBlock) void k_db2_out_fill(int64_t m, int64_t lo, int64_t n_own, const uint32_t* __restrict__ tp, const uint32_t* __restrict__ tkeys_s,
                                                        const uint64_t* __restrict__ ot_t, const uint8_t* __restrict__ ocr_t,
                                                        const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ perm,
                                                        Db2Src* __restrict__ src_t, const uint32_t* __restrict__ hoff,
                                                        const uint32_t* __restrict__ rank_t) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t b = tkeys_s[p];
    Db2Src r;
    r.t = ot_t[p];
    r.a = b;
    r.u = 0xFFFFFFFFu;
    const int64_t bl = (int64_t)b - lo;
    if (bl < 0 || bl >= n_own) {
        r.u = kDb2Foreign;
    } else if (tp[b + 1] - tp[b] <= (uint32_t)kWave) {         // (else an overflow node: its events keep the id 0xFFFFFFFF, the caller falls back)
        const uint32_t row = (uint32_t)row_ptr[bl] + ocr_t[p];
        r.u = perm ? (uint32_t)perm[row] : row;
    } else if (rank_t != nullptr) {                            // out-hub (one GPU): rank of the event's successor from the hub scratch
        r.u = (uint32_t)row_ptr[bl] + rank_t[hoff[b] + ((uint32_t)p - tp[b])];
    }
    src_t[p] = r;
}

// one GPU: k_db2_out_heads + k_db2_out_fill in ONE pass over the list positions (same tail node, same row pointers, same ranks; round 5)
__global__ __launch_bounds__(kBlock) void k_db2_out_ids(int64_t m, const uint32_t* __restrict__ tp, const uint32_t* __restrict__ tkeys_s,
                                                       const uint32_t* __restrict__ oc_s, const float* __restrict__ ow_s,
                                                       const uint8_t* __restrict__ ocr_s, const uint64_t* __restrict__ ot_t,
                                                       const uint8_t* __restrict__ ocr_t, const int32_t* __restrict__ row_ptr,
                                                       int32_t* __restrict__ fo_bwd_idx, float* __restrict__ fo_w, Db2Src* __restrict__ src_t,
                                                       const uint32_t* __restrict__ hoff, const uint32_t* __restrict__ rank_s,
                                                       const uint32_t* __restrict__ rank_t) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t b = tkeys_s[p];
    const uint32_t p0 = tp[b], cnt = tp[b + 1] - p0;
    const uint32_t row0 = (uint32_t)row_ptr[b];
    Db2Src r;
    r.t = ot_t[p];
    r.a = b;
    r.u = 0xFFFFFFFFu;
    if (cnt <= (uint32_t)kWave) {
        const uint8_t cr = ocr_s[p];
        if (p == p0 || ocr_s[p - 1] != cr) {               // head of a successor run: the first-order edge b -> c = order-2 node row0 + rank
            fo_bwd_idx[row0 + cr] = (int32_t)oc_s[p];
            fo_w[row0 + cr] = ow_s[p];
        }
        r.u = row0 + ocr_t[p];
    } else if (rank_s != nullptr) {                        // out-hub: 32-bit ranks in the hub scratch
        const uint32_t j = hoff[b] + ((uint32_t)p - p0);
        const uint32_t cr = rank_s[j];
        if (p == p0 || rank_s[j - 1] != cr) {
            fo_bwd_idx[row0 + cr] = (int32_t)oc_s[p];
            fo_w[row0 + cr] = ow_s[p];
        }
        r.u = row0 + rank_t[j];
    }
    src_t[p] = r;
}

// partition shards: the source-major rows of the first-order shard (sources: ALL nodes in the dense local order, destinations: owned nodes) are the
// successor runs that fall into the owned range; coefficients d_b^-1/2 w d_c^-1/2 from the all-gathered degrees
__global__ __launch_bounds__(kBlock) void k_db2_fo_part_bwd(int64_t m, int64_t lo, int64_t n_own, const uint32_t* __restrict__ tp,
                                                           const uint32_t* __restrict__ tkeys_s, const uint32_t* __restrict__ oc_s,
                                                           const uint8_t* __restrict__ ocr_s, const float* __restrict__ ow_s,
                                                           const uint8_t* __restrict__ fskip, const int32_t* __restrict__ fo2_ptr,
                                                           const float* __restrict__ fo_deg, int32_t* __restrict__ idx, float* __restrict__ val) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t b = tkeys_s[p];
    const uint32_t p0 = tp[b];
    if (tp[b + 1] - p0 > (uint32_t)kWave) return;
    const uint8_t cr = ocr_s[p];
    if (!(p == p0 || ocr_s[p - 1] != cr)) return;
    const int64_t c = oc_s[p];
    if (c < lo || c >= lo + n_own) return;
    const int64_t bl = (int64_t)b - lo;
    const int64_t jloc = (bl >= 0 && bl < n_own) ? bl : ((int64_t)b < lo ? n_own + b : b);
    const int64_t pos = (int64_t)fo2_ptr[jloc] + cr - fskip[b];
    idx[pos] = (int32_t)(c - lo);
    val[pos] = (int64_t)b == c ? 0.0f : inv_sqrt_deg(fo_deg[b]) * ow_s[p] * inv_sqrt_deg(fo_deg[c]);
}

// ------------------------------------------------------------------ middle-node pass
struct Db2Mid {
    int64_t lo;                      // first owned node (0 on one GPU); arrays indexed by node: tp, hp, fo_deg GLOBAL ids, everything else owned-local
    int64_t n_own;                   // partition shard (> 0): first-order sources are written as dense local ids [owned | below lo | from hi on]
    int part;
    const uint32_t *tp, *hp;
    const uint64_t* ot_s;            // out-events in (successor, time) order: time, successor rank
    const uint8_t* ocr_s;
    const int32_t* row_ptr;
    const int32_t* perm;             // partition shards: local id of the lexicographic row (nullptr: the identity)
    // in-events in (source, time) order
    const uint64_t* is_t;
    const uint32_t *is_a, *is_u;
    const float* is_w;
    // count pass on one GPU: the in-events are taken straight from their positions in the out-lists (hl = head-sorted order, src_t = one 16-byte
    // record per out-list position) — the random reads of what was k_db2_gather_in, hidden behind the other waves — and written for the fill pass
    const uint32_t* hl;
    const Db2Src* src_t;
    const float* ow_t;
    uint64_t* is_t_out;
    uint32_t *is_a_out, *is_u_out;
    float* is_w_out;
    // count pass: outputs; fill pass: inputs
    int32_t *indeg2, *outdeg2;
    float *ho_deg, *ho_lw, *fo_deg, *fo_lw;
    int32_t *nu, *pc;
    int64_t* status;
    // fill pass
    const uint32_t* oc_s;            // fill pass on one GPU: successor node of every out-event, merged first-order weights -> the source-major
    const float* fo_w;               // first-order coefficients are written by the owner node's wave (what was k_db2_fo_bwd_val)
    float* fo_bwd_val;
    const uint2* row_pack;           // per order-2 row: (d^-1/2 bits, start of its source-major row): gathered by the fill pass's lanes themselves
    const int32_t *ho_fwd_ptr, *fo_fwd_ptr;
    int32_t *in_idx2, *fwd_idx1, *dst_order;
    float *in_val2, *self2, *fwd_val1, *self1;
    float* in_w2;                    // optional: the merged weights themselves in destination-major order (MultiOrderModel.layers[2].data.edge_weight)
    uint2* out_pack;
    // count -> fill: a node whose in-runs are single events that reach no successor run twice is SIMPLE: run_em[position of the in-event] = the
    // successor runs it reaches (bit = lane of the run's first out-event); the fill pass then handles all its runs at once (no window tests)
    uint64_t* run_em;
    uint8_t* node_simple;
    int hubs_handled;                // one GPU: nodes with more than 64 in- / out-events are left to k_db2_hub (no overflow status)
};

template <bool kFill, bool kW>
struct Db2MidIn {
    uint32_t p0, q0;
    int no, ni;
    uint64_t tj, ti;
    int cr;
    uint32_t ia, iu;
    float wi, du, da;
    int32_t ob;
    int32_t row0;
    int32_t rid;               // local id of the row with successor rank l (lane l)
    float rdeg, rlw;           // fill: per successor RANK (lane r = row row0 + r): weighted degree, self-loop weight, row start
    int32_t rip;
    float d1, lw1;
    int32_t fp;
    uint64_t em;               // fill, simple nodes: run_em of the in-event at lane l
    int simple;
};

template <typename TimeT, int kMode, bool kFill, bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_mid(int64_t n, int64_t delta_i, double delta_f, Db2Mid a) {
    using W = Window<TimeT, kMode>;
    const int64_t node0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * kDb2Nodes;
    if (node0 >= n) return;
    const int l = lane_id();
    Db2MidIn<kFill, kW> in[kDb2Nodes];
#pragma unroll
    for (int s = 0; s < kDb2Nodes; ++s) {
        const int64_t node = node0 + s;
        const bool there = node < n;
        const int64_t gn = a.lo + node;
        const uint32_t b0 = there ? a.tp[gn] : 0u, b1 = there ? a.tp[gn + 1] : 0u, h0 = there ? a.hp[gn] : 0u, h1 = there ? a.hp[gn + 1] : 0u;
        const int32_t r0 = there ? a.row_ptr[node] : 0;
        in[s].p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);          // (wave-uniform by construction: scalar loop bounds, scalar addresses)
        in[s].q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0);
        in[s].no = __builtin_amdgcn_readfirstlane((int)(b1 - b0));
        in[s].ni = __builtin_amdgcn_readfirstlane((int)(h1 - h0));
        in[s].row0 = __builtin_amdgcn_readfirstlane(r0);
        if (kFill) {
            in[s].d1 = there ? inv_sqrt_deg(a.fo_deg[gn]) : 0.0f;
            in[s].lw1 = there ? a.fo_lw[node] : 1.0f;
            in[s].fp = __builtin_amdgcn_readfirstlane(there ? a.fo_fwd_ptr[node] : 0);
            in[s].simple = __builtin_amdgcn_readfirstlane(there ? (int)a.node_simple[node] : 0);
        }
    }
#pragma unroll
    for (int s = 0; s < kDb2Nodes; ++s) {
        const bool fits = in[s].no <= kWave && in[s].ni <= kWave;
        const bool lo_ = fits && l < in[s].no, li = fits && l < in[s].ni;
        in[s].tj = lo_ ? a.ot_s[in[s].p0 + l] : 0ull;
        in[s].cr = lo_ ? (int)a.ocr_s[in[s].p0 + l] : 255;
        in[s].rid = (lo_ && a.perm) ? a.perm[in[s].row0 + l] : in[s].row0 + l;      // (at most `no` successor rows: a row beyond the node's block is never used)
        if (!kFill && a.hl != nullptr) {
            const uint32_t q = in[s].q0 + l;
            const uint32_t e = li ? a.hl[q] : 0u;
            Db2Src r;
            r.t = 0ull; r.u = 0xFFFFFFFFu; r.a = 0xFFFFFFFFu;
            if (li) r = a.src_t[e];
            in[s].ti = r.t;
            in[s].ia = r.a;
            in[s].iu = r.u;
            in[s].wi = (kW && li) ? a.ow_t[e] : 1.0f;
            if (li) {
                a.is_t_out[q] = r.t;
                a.is_a_out[q] = r.a;
                a.is_u_out[q] = r.u;
                if (kW) a.is_w_out[q] = in[s].wi;
            }
        } else {
            in[s].ti = li ? a.is_t[in[s].q0 + l] : 0ull;
            in[s].ia = li ? a.is_a[in[s].q0 + l] : 0xFFFFFFFFu;
            in[s].iu = li ? a.is_u[in[s].q0 + l] : 0xFFFFFFFFu;
            in[s].wi = (kW && li) ? a.is_w[in[s].q0 + l] : 1.0f;
        }
        if (kFill) {
            in[s].em = (li && in[s].simple) ? a.run_em[in[s].q0 + l] : 0ull;
        }
    }
    if (kFill) {
#pragma unroll
        for (int s = 0; s < kDb2Nodes; ++s) {
            // what the fill pass needs per in-event: d^-1/2 of its order-2 row and of its source node, the start of that row's source-major entries — two
            // random reads per lane, issued for all four nodes at once and hidden behind the other waves' instruction issue (round 4: they were a
            // kernel of their own, 0.25 ms of pure latency)
            const bool ok = in[s].no <= kWave && in[s].ni <= kWave && l < in[s].ni && in[s].iu < kDb2Foreign;
            const uint2 rp = ok ? a.row_pack[in[s].iu] : make_uint2(0u, 0u);
            in[s].du = __uint_as_float(rp.x);
            in[s].ob = (int32_t)rp.y;
            in[s].da = ok ? inv_sqrt_deg(a.fo_deg[in[s].ia]) : 0.0f;
            if (a.fo_bwd_val != nullptr) {
                // source-major coefficients of the node's first-order out-edges (one per successor run): val(b -> c) = d_b^-1/2 w d_c^-1/2, 0 on a self loop
                const bool lo_ = in[s].no <= kWave && in[s].ni <= kWave && l < in[s].no;
                const int prev = __shfl_up(in[s].cr, 1, kWave);
                if (lo_ && (l == 0 || in[s].cr != prev)) {
                    const uint32_t u = (uint32_t)in[s].row0 + (uint32_t)in[s].cr;
                    const uint32_t c = a.oc_s[in[s].p0 + l];
                    a.fo_bwd_val[u] = (uint32_t)(a.lo + node0 + s) == c ? 0.0f : in[s].d1 * a.fo_w[u] * inv_sqrt_deg(a.fo_deg[c]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < kDb2Nodes; ++s) {
            const bool lo_ = in[s].no <= kWave && in[s].ni <= kWave && l < in[s].no;
            in[s].rdeg = lo_ ? a.ho_deg[in[s].rid] : 1.0f;
            in[s].rlw = lo_ ? a.ho_lw[in[s].rid] : 1.0f;
            in[s].rip = lo_ ? a.ho_fwd_ptr[in[s].rid] : 0;
        }
    }
#pragma unroll
    for (int s = 0; s < kDb2Nodes; ++s) {
        const int64_t node = node0 + s;
        if (node >= n) break;
        const uint32_t gnode = (uint32_t)(a.lo + node);
        const int no = in[s].no, ni = in[s].ni;
        if (no > kWave || ni > kWave) {
            if (a.hubs_handled) continue;
            if (!kFill && l == 0) {
                atomicOr((unsigned long long*)a.status, (unsigned long long)kDb2Overflow);
                a.nu[node] = 0; a.pc[node] = 0; a.fo_deg[gnode] = 1.0f; a.fo_lw[node] = 1.0f; a.node_simple[node] = 0;
            }
            continue;
        }
        // ---- out side: lanes in (successor, time) order
        const bool lo_ = l < no;
        const TimeT tj = time_of<TimeT>(in[s].tj);
        const int cr = in[s].cr;
        const int prevcr = __shfl_up(cr, 1, kWave);
        const bool ohead = lo_ && (l == 0 || cr != prevcr);
        const uint64_t ohm = __ballot(ohead);
        const uint64_t olater = ohm & ~lanes_upto(l);
        const int oend = olater ? __ffsll((long long)olater) - 1 : no;
        const uint64_t myrun = ohead ? ((oend >= kWave ? ~0ull : lanes_below(oend)) & ~lanes_below(l)) : 0ull;
        const uint32_t v = (uint32_t)lane_read_i((ohead ? cr : l) << 2, in[s].rid);       // local id of this run's row (lane cr holds the row of rank cr)
        // ---- in side: lanes in (source node, time) order
        const bool li = l < ni;
        const uint32_t sa = in[s].ia, su = in[s].iu;
        const uint64_t sti = in[s].ti;
        const float swi = in[s].wi;
        const uint32_t preva = (uint32_t)__shfl_up((int)sa, 1, kWave);
        const bool ihead = li && (l == 0 || sa != preva);
        const uint64_t ihm = __ballot(ihead);
        float dv = 0.0f, lwv = 1.0f;
        int32_t ip = 0;
        if (kFill) {                               // the successor run with rank cr takes its row's values from lane cr
            const int from = (ohead ? cr : l) << 2;
            dv = inv_sqrt_deg(lane_read_f(from, in[s].rdeg));
            lwv = lane_read_f(from, in[s].rlw);
            ip = lane_read_i(from, in[s].rip);
        }
        const float d1b = in[s].d1;
        if (kFill && in[s].simple) {
            // ---- all in-runs of the node at once (lane r = in-event r = in-run r): no loop over the runs, no window test
            const uint64_t em = li ? in[s].em : 0ull;
            uint64_t col = 0ull;                                   // head lane c: the in-runs that reach successor run c (the transposed masks)
            for (uint64_t hm2 = ohm; hm2 != 0; hm2 &= hm2 - 1) {
                const int c = __ffsll((long long)hm2) - 1;
                const uint64_t reach = __ballot((em >> c) & 1ull);
                if (l == c) col = reach;
            }
            if (li) {
                uint32_t run_a = sa;
                if (a.part) run_a = (int64_t)sa < a.lo ? (uint32_t)(a.n_own + sa) : ((int64_t)sa >= a.lo + a.n_own ? sa : (uint32_t)(sa - a.lo));
                const float w1 = kW ? swi : 1.0f;
                a.fwd_idx1[in[s].fp + l] = (int32_t)run_a;
                a.fwd_val1[in[s].fp + l] = sa == gnode ? 0.0f : in[s].da * w1 * d1b;
                if (a.dst_order) a.dst_order[in[s].fp + l] = (int32_t)su;
            }
            const int col_lo = (int)(uint32_t)col, col_hi = (int)(uint32_t)(col >> 32);
            for (uint64_t left = em; __ballot(left != 0ull) != 0ull;) {          // (wave-uniform trip count: the lane reads below need every lane)
                const bool act = left != 0ull;
                const int c = act ? __ffsll((long long)left) - 1 : 0;
                const uint32_t vc = (uint32_t)lane_read_i(c << 2, (int)v);
                const float dvc = lane_read_f(c << 2, dv);
                const int32_t ipc = lane_read_i(c << 2, ip);
                const uint64_t reach = ((uint64_t)(uint32_t)lane_read_i(c << 2, col_hi) << 32) | (uint64_t)(uint32_t)lane_read_i(c << 2, col_lo);
                if (act) {
                    const int pos = (int)__popcll(reach & lanes_below(l));
                    const int rank = (int)__popcll(em & lanes_below(c));
                    const float wgt = kW ? swi : 1.0f;
                    const float val = su == vc ? 0.0f : in[s].du * wgt * dvc;
                    a.in_idx2[ipc + pos] = (int32_t)su;
                    a.in_val2[ipc + pos] = val;
                    if (a.in_w2) a.in_w2[ipc + pos] = wgt;
                    a.out_pack[in[s].ob + rank] = make_uint2(vc, __float_as_uint(val));
                    left &= left - 1;
                }
            }
            if (ohead) a.self2[v] = dv * lwv * dv;
            if (l == 0 && a.self1) a.self1[node] = d1b * in[s].lw1 * d1b;
            continue;
        }
        int cnt = 0, pairs = 0, nuc = 0;
        bool node_simple = true, twice = false;                    // count pass: every in-run one event / some successor run reached twice by one in-run (per lane)
        uint64_t run_em = 0ull;
        float deg = 0.0f, lw = -1.0f, deg1 = 0.0f, lw1 = -1.0f;
        if (!kFill && (int)__popcll(ihm) == ni) {
            // ---- count pass, every in-event from its own source node (an ER stream: all nodes): one window ballot per in-event, then all runs at once
            uint64_t my_em = 0ull;                                 // lane r: successor runs (head lanes) in-event r reaches
            int all_pairs = 0;
            bool again = false;
            for (int z = 0; z < ni; ++z) {
                const TimeT ti = time_of<TimeT>(rl_u64(sti, z));
                const typename W::Thr thr = W::threshold(ti, delta_i, delta_f);
                const uint64_t win = __ballot(lo_ && tj > ti && W::admits(tj, thr));
                all_pairs += (int)__popcll(win);
                const int h = (int)__popcll(win & myrun);          // (myrun: the head lane's own run, 0 elsewhere)
                again = again || h > 1;
                const uint64_t reach = __ballot(h > 0);
                if (l == z) my_em = reach;
            }
            if (__ballot(again) == 0ull) {
                uint64_t col = 0ull;                               // head lane c: the in-events that reach successor run c
                for (uint64_t hm2 = ohm; hm2 != 0; hm2 &= hm2 - 1) {
                    const int c = __ffsll((long long)hm2) - 1;
                    const uint64_t reach = __ballot((my_em >> c) & 1ull);
                    if (l == c) col = reach;
                }
                if (li) {
                    const int od = (int)__popcll(my_em);
                    if (od != 0 && su != 0xFFFFFFFFu) a.outdeg2[su] = od;      // (no id: the source's node overflowed)
                    a.run_em[in[s].q0 + l] = my_em;
                }
                // the successor rows: in-degree, weighted degree and self-loop weight, summed in the order of the in-events (as the general loop does)
                const int su_i = (int)su;
                for (uint64_t left = ohead ? col : 0ull; __ballot(left != 0ull) != 0ull;) {      // (wave-uniform trip count: lane reads need every lane)
                    const bool act = left != 0ull;
                    const int r = act ? __ffsll((long long)left) - 1 : 0;
                    const uint32_t ur = (uint32_t)lane_read_i(r << 2, su_i);
                    const float wr = kW ? lane_read_f(r << 2, swi) : 1.0f;
                    if (act) {
                        if (ur == v) lw = wr; else deg += wr;
                        left &= left - 1;
                    }
                }
                if (ohead) {
                    const float l2 = lw < 0.0f ? 1.0f : lw;
                    a.indeg2[v] = (int)__popcll(col);
                    a.ho_deg[v] = deg + l2;
                    a.ho_lw[v] = l2;
                }
                // the node's first-order in-edges, one per in-event, summed in their order
                for (int z = 0; z < ni; ++z) {
                    const float wz = kW ? rl_f(swi, z) : 1.0f;
                    if (rl_u(sa, z) == gnode) lw1 = wz; else deg1 += wz;
                }
                if (l == 0) {
                    const float l1 = lw1 < 0.0f ? 1.0f : lw1;
                    a.node_simple[node] = 1;
                    a.nu[node] = ni;
                    a.pc[node] = all_pairs;
                    a.fo_deg[gnode] = deg1 + l1;
                    a.fo_lw[node] = l1;
                }
                continue;
            }
            // (a successor run reached twice by one in-event: the general loop below)
        }
        // per in-run results are parked in lane `run index` and stored after the loop with one instruction each
        uint32_t run_u = 0xFFFFFFFFu, run_a = 0u;
        int32_t run_od = 0;
        float run_val = 0.0f;
        for (uint64_t hm = ihm; hm != 0; hm &= hm - 1) {
            const int z0 = __ffsll((long long)hm) - 1;
            const uint64_t nxt = hm & (hm - 1);
            const int z1 = nxt ? __ffsll((long long)nxt) - 1 : ni;
            const uint32_t acur = rl_u(sa, z0), ucur = rl_u(su, z0);
            int hits = 0;
            float facc = 0.0f, w1run = 0.0f;
            for (int z = z0; z < z1; ++z) {
                const TimeT ti = time_of<TimeT>(rl_u64(sti, z));
                const typename W::Thr thr = W::threshold(ti, delta_i, delta_f);
                const uint64_t win = __ballot(lo_ && tj > ti && W::admits(tj, thr));
                if (!kFill) pairs += (int)__popcll(win);
                const int h = (int)__popcll(win & myrun);
                hits += h;
                if (kW) {
                    const float wz = rl_f(swi, z);
                    for (int x = 0; x < h; ++x) facc += wz;      // instance pairs in lexicographic order carry the weight of their source event
                    w1run += wz;
                }
            }
            if (!kW) w1run = (float)(z1 - z0);
            const float wgt = kW ? facc : (float)hits;
            const bool emit = ohead && hits > 0;
            const uint64_t em = __ballot(emit);
            if (!kFill) {
                if (emit) {
                    ++cnt;
                    if (ucur == v) lw = wgt; else deg += wgt;
                }
                node_simple = node_simple && z1 - z0 == 1;
                twice = twice || (ohead && hits > 1);
                if (l == nuc) { run_u = ucur; run_od = (int32_t)__popcll(em); run_em = em; }
                if (acur == gnode) lw1 = w1run; else deg1 += w1run;
            } else {
                const float du_ = rl_f(in[s].du, z0), da_ = rl_f(in[s].da, z0);
                const int32_t ob_ = rl_i(in[s].ob, z0);
                if (emit) {
                    const float val = ucur == v ? 0.0f : du_ * wgt * dv;
                    a.in_idx2[ip + cnt] = (int32_t)ucur;
                    a.in_val2[ip + cnt] = val;
                    if (a.in_w2) a.in_w2[ip + cnt] = wgt;
                    const int rank = (int)__popcll(em & lanes_below(l));
                    a.out_pack[ob_ + rank] = make_uint2(v, __float_as_uint(val));
                    ++cnt;
                }
                if (l == nuc) {
                    run_u = ucur;
                    run_a = acur;
                    if (a.part) run_a = (int64_t)acur < a.lo ? (uint32_t)(a.n_own + acur) : ((int64_t)acur >= a.lo + a.n_own ? acur : (uint32_t)(acur - a.lo));
                    run_val = acur == gnode ? 0.0f : da_ * w1run * d1b;
                }
            }
            ++nuc;
        }
        if (!kFill) {
            if (l < nuc && run_od != 0 && run_u != 0xFFFFFFFFu) a.outdeg2[run_u] = run_od;      // (no id: the source's node overflowed)
            if (ohead) {
                const float l2 = lw < 0.0f ? 1.0f : lw;              // an existing self loop keeps its weight, every other node gets one of weight 1
                a.indeg2[v] = cnt;
                a.ho_deg[v] = deg + l2;
                a.ho_lw[v] = l2;
            }
            node_simple = node_simple && __ballot(twice) == 0ull;
            if (node_simple && l < nuc) a.run_em[in[s].q0 + l] = run_em;       // (simple: run r = in-event r)
            if (l == 0) {
                const float l1 = lw1 < 0.0f ? 1.0f : lw1;
                a.node_simple[node] = node_simple ? 1 : 0;
                a.nu[node] = nuc;
                a.pc[node] = pairs;
                a.fo_deg[gnode] = deg1 + l1;
                a.fo_lw[node] = l1;
            }
        } else {
            if (l < nuc) {
                a.fwd_idx1[in[s].fp + l] = (int32_t)run_a;
                a.fwd_val1[in[s].fp + l] = run_val;
                if (a.dst_order) a.dst_order[in[s].fp + l] = (int32_t)run_u;
            }
            if (ohead) a.self2[v] = dv * lwv * dv;
            if (l == 0 && a.self1) a.self1[node] = d1b * in[s].lw1 * d1b;
        }
    }
}

// ------------------------------------------------------------------ hub nodes (round 5)
// A node with more than 64 in-events or more than 64 out-events.  Every order-2 edge through it is still decided by its in-events x
// out-events; what changes is who works on them:
//   k_db2_hub_classify   one thread per node, after both row pointers: hub flags, and for every hub its tasks (chunks of kHubChunk in-events,
//                        one wave each), its slice of the per-task partial results and — for more than 64 out-events — its slice of the
//                        out-hub scratch.  The totals {hubs, out-hub events, tasks, part columns} are what pp_debruijn2_lists reports.
//   k_db2_hub_out_*      out-events of the out-hubs in (successor, time) order by ONE radix sort of (slice, successor) keys over exactly
//                        those events (the list sequence is in time order and the sort is stable), then run heads -> scan -> 32-bit
//                        successor ranks, run starts, block sizes, merged first-order weights — what k_db2_out does in registers.
//   k_db2_hub            the middle-node pass of one task of a node with at most 64 out-events.  LANES ARE SUCCESSOR RUNS, in-events are walked
//                        one by one: the window ballot over the out-events in registers, popcount against the run's lane mask (as k_db2_mid).
//                        At the end of an in-run (a, b): ballot of the runs it reached = its out-degree / its source-major row, every
//                        reached run takes one more entry of its destination-major row.  A run that starts in the chunk is finished by this
//                        task however far it reaches; a chunk that starts inside a run skips to its end (bisection: sources ascend).
//   k_db2_hubx           the same for a node with MORE than 64 out-events, an in-run at a time (see there).
//                        Count pass: per task and run the number of entries, the weighted-degree share and the self-loop weight (a part
//                        column of 64 lanes); k_db2_hub_combine turns them into in-degrees, degrees and EXCLUSIVE prefixes per task, in task
//                        order (fixed summation order, bit-reproducible); the fill pass starts every row at its task's prefix.
// Sums: in-run weights and degrees are summed in (in-event, task) order — identical to the one-wave kernels and the generic path whenever
// the partial sums are exactly representable (unit and integer-valued weights below 2^24), a different association of the same fp32 terms
// otherwise (as the chunked reduction of long runs in pp_coalesce_*); with more than 64 out-events a merged weight is the sum, over the
// successor's out-events in time order, of the weights of the instances each continues.
struct Db2Hub {
    const uint8_t* flag;             // [n] kHubOut | kHubIn
    const uint32_t *hoff, *oslot;    // [n] out-hubs: first slot of the out-hub scratch, index among the out-hubs
    const uint32_t *tbase;           // [n] first task of a hub
    const int64_t* pbase;            // [n] first part column of a hub
    const uint32_t *task_node, *hub_list;
    const uint32_t* run_start;       // out-hub scratch: [hoff + oslot + r] = first out-event (list-relative) of successor run r, one sentinel behind
    int32_t* part_cnt;               // [P * 64] per (task, round, run): entries (count pass), then their exclusive prefix over the tasks
    float *part_deg, *part_lw;
    int32_t *task_runs, *task_runbase;
    float *task_deg1, *task_lw1;
    int64_t* stats;                  // result + kDb2HubStats
    int64_t num_nodes;
    const int32_t* succ;             // fill: successor node of every order-2 row (fo_bwd_idx)
};
constexpr int kDb2HubStats = 8 + 2 * (64 + 1);        // index of the hub block inside the result header (behind recv_ptr / send_ptr of 64 ranks)
// stats[0] hubs, [1] (out-hubs << 32) | their out-events, [2] tasks, [3] part columns, [4] lifted pairs through hubs,
// [5..8] longest row: order-2 destination-major / source-major, first-order destination-major / source-major, [9] longest in-list, [10] out-list

__global__ __launch_bounds__(kBlock) void k_db2_hub_classify(int64_t n, const uint32_t* __restrict__ tp, const uint32_t* __restrict__ hp,
                                                            uint8_t* __restrict__ flag, uint32_t* __restrict__ hoff, uint32_t* __restrict__ oslot,
                                                            uint32_t* __restrict__ tbase, int64_t* __restrict__ pbase,
                                                            uint32_t* __restrict__ hub_list, int64_t* __restrict__ stats) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= n) return;
    const int64_t no = (int64_t)tp[b + 1] - tp[b], ni = (int64_t)hp[b + 1] - hp[b];
    const uint8_t f = (uint8_t)((no > kWave ? kHubOut : 0) | (ni > kWave ? kHubIn : 0));
    flag[b] = f;
    if (!f) return;
    const int64_t ntask = ni > 0 ? ceil_div_dev(ni, (int64_t)hub_chunk(ni, no)) : 1;
    const int64_t nrc = no > 0 ? ceil_div_dev(no < n ? no : n, kWave) : 1;          // rounds of 64 successor runs, upper bound (runs <= min(no, n))
    hub_list[atomicAdd((unsigned long long*)&stats[0], 1ull)] = (uint32_t)b;
    const uint32_t t0 = (uint32_t)atomicAdd((unsigned long long*)&stats[2], (unsigned long long)ntask);
    tbase[b] = t0;
    pbase[b] = (int64_t)atomicAdd((unsigned long long*)&stats[3], (unsigned long long)(ntask * nrc));
    if (no > kWave) {           // ONE atomic for (index among the out-hubs, first event slot): both follow the same arrival order
        const unsigned long long old = atomicAdd((unsigned long long*)&stats[1], (1ull << 32) | (unsigned long long)no);
        oslot[b] = (uint32_t)(old >> 32);
        hoff[b] = (uint32_t)old;
    }
    atomicMax((unsigned long long*)&stats[9], (unsigned long long)ni);
    atomicMax((unsigned long long*)&stats[10], (unsigned long long)no);
}

// task -> node, one workgroup per hub (a node with 2*10^6 in-events has 7 800 tasks: written by ONE thread of the classification they took 0.13 ms)
__global__ __launch_bounds__(kBlock) void k_db2_hub_tasks(int64_t n_hubs, const uint32_t* __restrict__ hub_list, const uint32_t* __restrict__ hp,
                                                         const uint32_t* __restrict__ tp, const uint32_t* __restrict__ tbase, uint32_t* __restrict__ task_node) {
    if ((int64_t)blockIdx.x >= n_hubs) return;
    const uint32_t b = hub_list[blockIdx.x];
    const int64_t ni = (int64_t)hp[b + 1] - hp[b], no = (int64_t)tp[b + 1] - tp[b];
    const int64_t ntask = ni > 0 ? ceil_div_dev(ni, (int64_t)hub_chunk(ni, no)) : 1;
    const uint32_t t0 = tbase[b];
    for (int64_t k = threadIdx.x; k < ntask; k += kBlock) task_node[t0 + k] = b;
}

// in-events of the hub nodes in (source, time) order: what k_db2_mid's count prologue gathers for the other nodes
__global__ __launch_bounds__(kBlock) void k_db2_hub_gather_in(int64_t m, const uint32_t* __restrict__ hkeys_s, const uint8_t* __restrict__ flag,
                                                             const uint32_t* __restrict__ hl, const Db2Src* __restrict__ src_t,
                                                             const float* __restrict__ ow_t, uint64_t* __restrict__ is_t, uint32_t* __restrict__ is_a,
                                                             uint32_t* __restrict__ is_u, float* __restrict__ is_w) {
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= m || !flag[hkeys_s[q]]) return;
    const uint32_t p = hl[q];
    const Db2Src r = src_t[p];
    is_t[q] = r.t;
    is_a[q] = r.a;
    is_u[q] = r.u;
    if (ow_t) is_w[q] = ow_t[p];
}

// out-hubs, step 1: (slice, successor) key + list position of every out-event, compacted into the node's slice
__global__ __launch_bounds__(kBlock) void k_db2_hub_out_keys(int64_t m, const uint32_t* __restrict__ tkeys_s, const uint32_t* __restrict__ tp,
                                                            const uint8_t* __restrict__ flag, const uint32_t* __restrict__ hoff,
                                                            const uint32_t* __restrict__ oc_t, int key_bits, uint64_t* __restrict__ keys,
                                                            uint32_t* __restrict__ pos) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= m) return;
    const uint32_t b = tkeys_s[p];
    if (!(flag[b] & kHubOut)) return;
    const uint32_t j = hoff[b] + ((uint32_t)p - tp[b]);
    keys[j] = ((uint64_t)hoff[b] << key_bits) | (uint64_t)oc_t[p];
    pos[j] = (uint32_t)p;
}

// step 2 (after the sort): run heads.  The slices are sorted by their first slot, so element j of the sorted sequence lies in the slice of the
// node its event belongs to, at `j - hoff`
__global__ __launch_bounds__(kBlock) void k_db2_hub_out_heads(int64_t h_out, const uint64_t* __restrict__ keys_s, const uint32_t* __restrict__ pos_s,
                                                             const uint32_t* __restrict__ tkeys_s, const uint32_t* __restrict__ hoff,
                                                             uint32_t* __restrict__ head) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= h_out) return;
    const uint32_t b = tkeys_s[pos_s[j]];
    head[j] = ((uint32_t)j == hoff[b] || keys_s[j] != keys_s[j - 1]) ? 1u : 0u;
}

// step 3 (after the scan of the heads): successor ranks, the (successor, time)-ordered out-list, run starts, block sizes
template <bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_hub_out_write(int64_t h_out, const uint64_t* __restrict__ keys_s, const uint32_t* __restrict__ pos_s,
                                                             const uint32_t* __restrict__ head, const uint32_t* __restrict__ head_scan,
                                                             const uint32_t* __restrict__ tkeys_s, const uint32_t* __restrict__ tp,
                                                             const uint32_t* __restrict__ hoff, const uint32_t* __restrict__ oslot, int key_bits,
                                                             const uint64_t* __restrict__ ot_t, const float* __restrict__ ow_t,
                                                             uint64_t* __restrict__ ot_s, uint32_t* __restrict__ oc_s, float* __restrict__ ow_s,
                                                             uint32_t* __restrict__ rank_s, uint32_t* __restrict__ rank_t,
                                                             uint32_t* __restrict__ run_start, int32_t* __restrict__ blk, int64_t* __restrict__ stats) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= h_out) return;
    const uint32_t p = pos_s[j];
    const uint32_t b = tkeys_s[p];
    const uint32_t h0 = hoff[b], p0 = tp[b], no = tp[b + 1] - p0;
    const uint32_t within = (uint32_t)j - h0;
    const uint32_t r = head_scan[j] + head[j] - 1u - head_scan[h0];      // rank of this event's successor among the node's distinct successors
    const uint32_t c = (uint32_t)(keys_s[j] & ((1ull << key_bits) - 1ull));
    ot_s[p0 + within] = ot_t[p];
    oc_s[p0 + within] = c;
    ow_s[p0 + within] = kW ? ow_t[p] : 0.0f;                             // (weighted: the instance weight for now, k_db2_hub_out_runs sums the runs)
    rank_s[j] = r;
    rank_t[h0 + (p - p0)] = r;
    uint32_t* rs = run_start + (h0 + oslot[b]);
    if (head[j]) rs[r] = within;
    if (within == no - 1u) {
        rs[r + 1u] = no;
        blk[b] = (int32_t)(r + 1u);
        atomicMax((unsigned long long*)&stats[8], (unsigned long long)(r + 1u));
    }
}

// step 4: merged first-order weight of every successor run at its head (0 elsewhere, as k_db2_out leaves them): run length, or the
// left-to-right sum of the instance weights
template <bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_hub_out_runs(int64_t h_out, const uint32_t* __restrict__ pos_s, const uint32_t* __restrict__ head,
                                                            const uint32_t* __restrict__ rank_s, const uint32_t* __restrict__ tkeys_s,
                                                            const uint32_t* __restrict__ tp, const uint32_t* __restrict__ hoff,
                                                            const uint32_t* __restrict__ oslot, const uint32_t* __restrict__ run_start,
                                                            float* __restrict__ ow_s) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= h_out) return;
    const uint32_t b = tkeys_s[pos_s[j]];
    const uint32_t h0 = hoff[b], p0 = tp[b];
    const uint32_t within = (uint32_t)j - h0;
    if (!head[j]) {
        if (!kW) ow_s[p0 + within] = 0.0f;
        return;
    }
    const uint32_t* rs = run_start + (h0 + oslot[b]);
    const uint32_t end = rs[rank_s[j] + 1u];
    if (!kW) { ow_s[p0 + within] = (float)(end - within); return; }
    float acc = 0.0f;
    for (uint32_t x = within; x < end; ++x) acc += ow_s[p0 + x];          // (the run's own entries: nobody else touches them)
    ow_s[p0 + within] = acc;
    for (uint32_t x = within + 1u; x < end; ++x) ow_s[p0 + x] = 0.0f;
}

template <typename TimeT, int kMode, bool kFill, bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_hub(int64_t n_tasks, int64_t delta_i, double delta_f, Db2Mid a, Db2Hub h) {
    using W = Window<TimeT, kMode>;
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    if (task >= n_tasks) return;
    const int l = lane_id();
    const uint32_t b = h.task_node[task];
    if (h.flag[b] & kHubOut) return;                               // (more than 64 out-events: k_db2_hubx takes the node run by run)
    const int64_t k = task - (int64_t)h.tbase[b];
    const uint32_t p0 = a.tp[b], q0 = a.hp[b];
    const int64_t no = (int64_t)a.tp[b + 1] - p0, ni = (int64_t)a.hp[b + 1] - q0;
    const int32_t row0 = a.row_ptr[b];
    const int64_t R = (int64_t)a.row_ptr[b + 1] - row0;                                  // successor runs = order-2 rows (b, .)
    const int64_t rounds = R > 0 ? (R + kWave - 1) / kWave : 1;
    const int64_t nrc = no > 0 ? ((no < h.num_nodes ? no : h.num_nodes) + kWave - 1) / kWave : 1;      // part columns per task (k_db2_hub_classify)
    const int64_t part0 = h.pbase[b] + k * nrc;
    const int64_t qend = (int64_t)q0 + ni;
    int64_t qa = (int64_t)q0 + k * kHubChunk;
    const int64_t qb = qa + kHubChunk < qend ? qa + kHubChunk : qend;
    if (k > 0 && qa < qend) {                                     // the chunk starts inside a run of an earlier chunk: skip to the run's end
        const uint32_t prev = a.is_a[qa - 1];
        if (a.is_a[qa] == prev) qa = upper_bound_dev<uint32_t, int64_t>(a.is_a, qa, qend, prev);
    }
    // the (at most 64) out-events in registers: lanes = out-events in (successor, time) order for the window ballot, lane r also = run r
    uint64_t runmask = 0ull;
    const bool lo_ = l < no;
    const TimeT tj = time_of<TimeT>(lo_ ? a.ot_s[p0 + l] : 0ull);
    {
        const int cr = lo_ ? (int)a.ocr_s[p0 + l] : 255;
        const int prevcr = __shfl_up(cr, 1, kWave);
        const uint64_t ohm = __ballot(lo_ && (l == 0 || cr != prevcr));
        uint64_t from = ohm;
        for (int i = 0; i < l && from; ++i) from &= from - 1;      // heads from the l-th on
        if (from) {
            const int rs = __ffsll((long long)from) - 1;
            const uint64_t nxt = from & (from - 1);
            const int re = nxt ? __ffsll((long long)nxt) - 1 : (int)no;
            runmask = (re >= kWave ? ~0ull : lanes_below(re)) & ~lanes_below(rs);
        }
    }
    const float d1b = kFill ? inv_sqrt_deg(a.fo_deg[b]) : 0.0f;
    const int32_t fp = kFill ? a.fo_fwd_ptr[b] + h.task_runbase[task] : 0;
    if (kFill && k == 0 && l == 0 && a.self1) a.self1[b] = d1b * a.fo_lw[b] * d1b;
    long long pairs = 0;
    int runs = 0, longest_src = 0;
    float deg1 = 0.0f, lw1 = -1.0f;
    int od0 = 0, od1 = 0, od2 = 0, od3 = 0;                        // entries of the source-major row of in-run `ord` so far: lane ord % 64, register ord / 64
    for (int64_t c = 0; c < rounds; ++c) {
        const int64_t rr = c * kWave + l;
        const bool valid = rr < R;
        const uint32_t v = (uint32_t)(row0 + rr);
        int cnt = 0;
        float deg = 0.0f, lw = -1.0f, dv = 0.0f;
        int32_t ip = 0;
        if (kFill && valid) {
            dv = inv_sqrt_deg(a.ho_deg[v]);
            ip = a.ho_fwd_ptr[v] + h.part_cnt[(part0 + c) * kWave + l];
            if (k == 0) {
                const float lwv = a.ho_lw[v];
                a.self2[v] = dv * lwv * dv;
                if (a.fo_bwd_val != nullptr) {
                    const uint32_t cn = (uint32_t)h.succ[v];
                    a.fo_bwd_val[v] = cn == b ? 0.0f : d1b * a.fo_w[v] * inv_sqrt_deg(a.fo_deg[cn]);
                }
            }
        }
        // ---- the in-runs that start in [qa, qb).  One virtual event behind the node's last in-event closes the last run: the run epilogue
        // exists once, inside the loop (a lambda over the kernel's argument structs would force them into scratch memory)
        uint32_t cur_a = 0u, cur_u = 0xFFFFFFFFu;
        int ord = -1, hits = 0;
        float facc = 0.0f, w1run = 0.0f, du_r = 0.0f, da_r = 0.0f;
        int32_t ob_r = 0;
        bool open = false, stop = qa >= qend;
        for (int64_t qq = qa; !stop; qq += kWave) {
            const int64_t q = qq + l;
            const bool li = q < qend;
            const uint64_t sti = li ? a.is_t[q] : 0ull;
            const uint32_t sa = li ? a.is_a[q] : 0xFFFFFFFFu, su = li ? a.is_u[q] : 0xFFFFFFFFu;
            const float swi = (kW && li) ? a.is_w[q] : 1.0f;
            float du_l = 0.0f, da_l = 0.0f;
            int32_t ob_l = 0;
            if (kFill) {                                             // d^-1/2 of the in-event's order-2 row and source node, start of its source-major row
                const bool ok = li && su < kDb2Foreign;
                const uint2 rp = ok ? a.row_pack[su] : make_uint2(0u, 0u);
                du_l = __uint_as_float(rp.x);
                ob_l = (int32_t)rp.y;
                da_l = ok ? inv_sqrt_deg(a.fo_deg[sa]) : 0.0f;
            }
            const int nz = qend - qq < kWave ? (int)(qend - qq) : kWave;
            if (open && nz == kWave && rl_u(sa, 0) == cur_a && rl_u(sa, kWave - 1) == cur_a) {
                // A LONG IN-RUN: all 64 events of this sub-chunk (and maybe many more) continue the open run.  Walking them one by one is serial
                // in the run's length (a node pair carrying a third of a 2*10^6-event stream: 260 ms); the rest of the run [qq, qe2) is counted
                // FROM THE OUT-EVENTS' SIDE instead: lane j = out-event j counts the instances it continues by two bisections over the run's
                // (ascending) instance times — t_i < t_j and t_j <= t_i + delta are monotone in t_i — and every run sums its lanes' counts.
                const int64_t qe2 = upper_bound_dev<uint32_t, int64_t>(a.is_a, qq + kWave, qend, cur_a);
                int64_t lb = qq, pp = qq;
                if (lo_) {
                    int64_t lo = qq, hi = qe2;
                    while (lo < hi) {                                  // instances earlier than t_j
                        const int64_t mid = lo + ((hi - lo) >> 1);
                        if (time_of<TimeT>(a.is_t[mid]) < tj) lo = mid + 1; else hi = mid;
                    }
                    lb = lo;
                    lo = qq; hi = lb;
                    while (lo < hi) {                                  // ... of which the first whose window still reaches t_j
                        const int64_t mid = lo + ((hi - lo) >> 1);
                        if (!W::admits(tj, W::threshold(time_of<TimeT>(a.is_t[mid]), delta_i, delta_f))) lo = mid + 1; else hi = mid;
                    }
                    pp = lo;
                }
                const int cj = (int)(lb - pp);                         // (0 on lanes beyond the out-events)
                float wj = 0.0f;
                if (kW) {
                    // the weights of those instances, summed by the whole wave per out-event (ascending instance order inside every lane's share)
                    for (int x = 0; x < (int)no; ++x) {
                        const int64_t w0 = rl_i((int)(pp - qq), x) + qq, w1 = rl_i((int)(lb - qq), x) + qq;
                        float part = 0.0f;
                        for (int64_t y = w0 + l; y < w1; y += kWave) part += a.is_w[y];
                        part = wave_sum(part);
                        if (l == x) wj = part;
                    }
                    float wall = 0.0f;
                    for (int64_t y = qq + l; y < qe2; y += kWave) wall += a.is_w[y];
                    w1run += wave_sum(wall);
                } else {
                    w1run += (float)(qe2 - qq);
                }
                int hsum = 0;
                float fsum = 0.0f;
                for (int x = 0; x < (int)no; ++x) {                    // every run takes its lanes' counts (uniform loop: lane reads need every lane)
                    const int cx = rl_i(cj, x);
                    const float wx = kW ? rl_f(wj, x) : 0.0f;
                    if ((runmask >> x) & 1ull) { hsum += cx; fsum += wx; }
                }
                hits += hsum;
                pairs += hsum;
                if (kW) facc += fsum;
                qq = qe2 - kWave;                                      // (the loop's step brings it to the run's end: the next event closes the run)
                continue;
            }
            const int nsteps = qq + nz >= qend ? nz + 1 : nz;          // + the virtual event behind the last one
            for (int z = 0; z < nsteps; ++z) {
                const bool real = z < nz;
                const uint32_t az = real ? rl_u(sa, z & (kWave - 1)) : 0xFFFFFFFEu;
                if (!open || az != cur_a) {                          // position qq + z begins a run (or is the end)
                    if (open) {
                        // ---- epilogue of in-run (cur_a, b) = order-2 node cur_u
                        const float wgt = kW ? facc : (float)hits;
                        const bool emit = valid && hits > 0;
                        const uint64_t em = __ballot(emit);
                        const int reached = (int)__popcll(em);
                        const int slot = ord >> 6, olane = ord & (kWave - 1);
                        const int odsel = slot == 0 ? od0 : (slot == 1 ? od1 : (slot == 2 ? od2 : od3));
                        const int before = rl_i(odsel, olane);        // entries of this in-run's row from earlier rounds
                        if (l == olane) {
                            if (slot == 0) od0 += reached; else if (slot == 1) od1 += reached; else if (slot == 2) od2 += reached; else od3 += reached;
                        }
                        if (!kFill) {
                            if (emit) {
                                ++cnt;
                                if (cur_u == v) lw = wgt; else deg += wgt;
                            }
                            if (c == rounds - 1) {
                                const int total = before + reached;
                                if (l == 0 && total != 0 && cur_u < kDb2Foreign) a.outdeg2[cur_u] = total;
                                longest_src = total > longest_src ? total : longest_src;
                            }
                            if (c == 0) {
                                ++runs;
                                if (cur_a == b) lw1 = w1run; else deg1 += w1run;
                            }
                        } else {
                            if (emit) {
                                const float val = cur_u == v ? 0.0f : du_r * wgt * dv;
                                a.in_idx2[ip + cnt] = (int32_t)cur_u;
                                a.in_val2[ip + cnt] = val;
                                if (a.in_w2) a.in_w2[ip + cnt] = wgt;
                                const int rank = before + (int)__popcll(em & lanes_below(l));
                                a.out_pack[ob_r + rank] = make_uint2(v, __float_as_uint(val));
                                ++cnt;
                            }
                            if (c == 0 && l == 0) {
                                a.fwd_idx1[fp + ord] = (int32_t)cur_a;
                                a.fwd_val1[fp + ord] = cur_a == b ? 0.0f : da_r * w1run * d1b;
                                if (a.dst_order) a.dst_order[fp + ord] = (int32_t)cur_u;
                            }
                        }
                        open = false;
                    }
                    if (!real || qq + z >= qb) { stop = true; break; }     // the end, or a run of the next chunk's task
                    open = true;
                    cur_a = az;
                    cur_u = rl_u(su, z);
                    ++ord;
                    hits = 0; facc = 0.0f; w1run = 0.0f;
                    if (kFill) { du_r = rl_f(du_l, z); da_r = rl_f(da_l, z); ob_r = rl_i(ob_l, z); }
                }
                const TimeT ti = time_of<TimeT>(rl_u64(sti, z));
                const typename W::Thr thr = W::threshold(ti, delta_i, delta_f);
                const uint64_t win = __ballot(lo_ && tj > ti && W::admits(tj, thr));
                const int ci = (int)__popcll(win & runmask);
                if (kW) {
                    const float wz = rl_f(swi, z);
                    for (int x = 0; x < ci; ++x) facc += wz;          // as k_db2_mid: instance pairs in lexicographic order carry the weight of their source event
                    w1run += wz;
                }
                if (!kW) w1run += 1.0f;
                hits += ci;
                pairs += ci;
            }
        }
        if (!kFill) {
            const int64_t pi = (part0 + c) * kWave + l;
            h.part_cnt[pi] = cnt;
            h.part_deg[pi] = deg;
            h.part_lw[pi] = lw;
        }
        // the next round walks the same runs again: `ord` restarts, the row prefixes od0..3 stay
    }
    if (!kFill) {
        const long long total = wave_sum(pairs);
        const int longest = wave_max(longest_src);
        if (l == 0) {
            h.task_runs[task] = runs;
            h.task_deg1[task] = deg1;
            h.task_lw1[task] = lw1;
            if (total) atomicAdd((unsigned long long*)&h.stats[4], (unsigned long long)total);
            atomicMax((unsigned long long*)&h.stats[6], (unsigned long long)longest);
        }
    }
}

// OUT-HUBS, run at a time (round 5): the middle-node pass for nodes with more than 64 out-events — the shape of the reference's documented
// contact streams (tens of nodes, 10^4 events per node and side).  Walking the in-events one by one with each lane chasing its own run of
// out-events through memory costs one dependent load per step (the first form of this round: 26 ms on such a stream).  Here an in-run
// (a, b) is taken as a whole: its instance times go to LDS (pieces of 512), the out-events of b between the first instance's window start and
// the last instance's window end are streamed ONCE in time order (coalesced: timestamp + 32-bit successor rank), and every out-event counts
// the instances it continues by two bisections over the LDS times (t_i < t_j and t_j <= t_i + delta are monotone in t_i) and adds that count
// to its successor's slot of an LDS histogram (integer atomics: exact, order-free).  The histogram of 512 runs is the `hits` vector of eight
// 64-lane columns at once; nodes with more runs take several rounds.  Same tasks, same part columns, same epilogue as k_db2_hub.
#ifndef PP_HUBX_RUNS
#define PP_HUBX_RUNS 512
#endif
#ifndef PP_HUBX_AHEAD
#define PP_HUBX_AHEAD 4
#endif
constexpr int kHubxRuns = PP_HUBX_RUNS, kHubxInst = 512, kHubxCols = kHubxRuns / kWave;

template <typename TimeT, int kMode, bool kFill, bool kW>
__global__ __launch_bounds__(kBlock) void k_db2_hubx(int64_t n_tasks, int64_t delta_i, double delta_f, Db2Mid a, Db2Hub h, const uint64_t* __restrict__ ot_t,
                                                    const uint32_t* __restrict__ rank_t) {
    using W = Window<TimeT, kMode>;
    __shared__ uint32_t s_hist[kWavesPerBlock][kHubxRuns];
    __shared__ uint64_t s_time[kWavesPerBlock][kHubxInst];
    // weighted streams: the instance weights beside their times, and a second histogram of the weight sums (fp32 LDS atomics of ONE wave:
    // the order is the scan's — out-events in time order, lanes ascending — the same on every run)
    __shared__ float s_wt[kWavesPerBlock][kW ? kHubxInst : 1];
    __shared__ float s_histw[kWavesPerBlock][kW ? kHubxRuns : 1];
    float* wts = s_wt[wave_id()];
    float* histw = s_histw[wave_id()];
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    if (task >= n_tasks) return;
    const uint32_t b = h.task_node[task];
    if (!(h.flag[b] & kHubOut)) return;                          // (k_db2_hub's node)
    const int l = lane_id();
    uint32_t* hist = s_hist[wave_id()];
    uint64_t* tms = s_time[wave_id()];
    const int64_t k = task - (int64_t)h.tbase[b];
    const uint32_t p0 = a.tp[b], q0 = a.hp[b];
    const int64_t no = (int64_t)a.tp[b + 1] - p0, ni = (int64_t)a.hp[b + 1] - q0;
    const int32_t row0 = a.row_ptr[b];
    const int64_t R = (int64_t)a.row_ptr[b + 1] - row0;
    const int64_t nrc = no > 0 ? ((no < h.num_nodes ? no : h.num_nodes) + kWave - 1) / kWave : 1;
    const int64_t part0 = h.pbase[b] + k * nrc;
    const int64_t qend = (int64_t)q0 + ni;
    const int64_t chunk = hub_chunk(ni, no);
    int64_t qa = (int64_t)q0 + k * chunk;
    const int64_t qb = qa + chunk < qend ? qa + chunk : qend;
    if (k > 0 && qa < qend) {
        const uint32_t prev = a.is_a[qa - 1];
        if (a.is_a[qa] == prev) qa = upper_bound_dev<uint32_t, int64_t>(a.is_a, qa, qend, prev);
    }
    const uint64_t* otb = ot_t + p0;                              // out-events of b in time order
    const uint32_t* rkb = rank_t + h.hoff[b];                     // their successor ranks
    for (int i = l; i < kHubxRuns; i += kWave) {
        hist[i] = 0u;
        if (kW) histw[i] = 0.0f;
    }
    const float d1b = kFill ? inv_sqrt_deg(a.fo_deg[b]) : 0.0f;
    const int32_t fp = kFill ? a.fo_fwd_ptr[b] + h.task_runbase[task] : 0;
    if (kFill && k == 0 && l == 0 && a.self1) a.self1[b] = d1b * a.fo_lw[b] * d1b;
    long long pairs = 0;
    int runs = 0, longest_src = 0;
    float deg1 = 0.0f, lw1 = -1.0f;
    int od0 = 0, od1 = 0, od2 = 0, od3 = 0;
    const int64_t rounds = R > 0 ? (R + kHubxRuns - 1) / kHubxRuns : 1;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t rbase = rd * kHubxRuns;
        const int ncol = (int)((R - rbase + kWave - 1) / kWave < kHubxCols ? (R - rbase + kWave - 1) / kWave : kHubxCols);       // (0 when the node has no successor)
        int cnt[kHubxCols];
        float deg[kHubxCols], lw[kHubxCols], dv[kHubxCols];
        int32_t ip[kHubxCols];
#pragma unroll
        for (int cc = 0; cc < kHubxCols; ++cc) {
            cnt[cc] = 0; deg[cc] = 0.0f; lw[cc] = -1.0f; dv[cc] = 0.0f; ip[cc] = 0;
            const int64_t rr = rbase + cc * kWave + l;
            if (kFill && cc < ncol && rr < R) {
                const uint32_t v = (uint32_t)(row0 + rr);
                dv[cc] = inv_sqrt_deg(a.ho_deg[v]);
                ip[cc] = a.ho_fwd_ptr[v] + h.part_cnt[(part0 + rbase / kWave + cc) * kWave + l];
                if (k == 0) {
                    const float lwv = a.ho_lw[v];
                    a.self2[v] = dv[cc] * lwv * dv[cc];
                    if (a.fo_bwd_val != nullptr) {
                        const uint32_t cn = (uint32_t)h.succ[v];
                        a.fo_bwd_val[v] = cn == b ? 0.0f : d1b * a.fo_w[v] * inv_sqrt_deg(a.fo_deg[cn]);
                    }
                }
            }
        }
        int ord = -1;
        for (int64_t q = qa; q < qb;) {                          // q: first in-event of a run that begins in this chunk
            const uint32_t cur_a = a.is_a[q], cur_u = a.is_u[q];
            ++ord;
            float du_r = 0.0f, da_r = 0.0f;
            int32_t ob_r = 0;
            if (kFill && cur_u < kDb2Foreign) {
                const uint2 rp = a.row_pack[cur_u];
                du_r = __uint_as_float(rp.x);
                ob_r = (int32_t)rp.y;
                da_r = inv_sqrt_deg(a.fo_deg[cur_a]);
            }
            int64_t qp = q;
            bool run_done = false;
            float w1sum = 0.0f;                                       // weighted: the run's first-order weight (sum of its instances)
            while (!run_done) {
                // ---- one piece: up to kHubxInst instance times of the run -> LDS
                int got = 0;
                while (got < kHubxInst) {
                    const int64_t qx = qp + got + l;
                    const bool li = qx < qend;
                    const uint32_t sa = li ? a.is_a[qx] : 0xFFFFFFFFu;
                    const uint64_t st_ = li ? a.is_t[qx] : 0ull;
                    const float sw_ = (kW && li) ? a.is_w[qx] : 0.0f;
                    const uint64_t mism = __ballot(!li || sa != cur_a);
                    const int upto = mism ? __ffsll((long long)mism) - 1 : kWave;
                    const int room = kHubxInst - got;
                    const int take = upto < room ? upto : room;
                    if (l < take) {
                        tms[got + l] = st_;
                        if (kW) wts[got + l] = sw_;
                    }
                    if (kW) w1sum += wave_sum(l < take ? sw_ : 0.0f);
                    got += take;
                    if (upto < kWave && upto <= room) { run_done = true; break; }
                    if (take < kWave) break;                          // the piece is full
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ninst = got;
                if (ninst == 0) break;
                qp += ninst;
                if (ncol > 0) {
                    const TimeT t_first = time_of<TimeT>(tms[0]), t_last = time_of<TimeT>(tms[ninst - 1]);
                    const typename W::Thr thr_last = W::threshold(t_last, delta_i, delta_f);
                    int64_t g0 = 0, hi = no;
                    while (g0 < hi) {                                 // first out-event later than the first instance
                        const int64_t mid = g0 + ((hi - g0) >> 1);
                        if (time_of<TimeT>(otb[mid]) > t_first) hi = mid; else g0 = mid + 1;
                    }
                    int64_t g1 = g0;
                    hi = no;
                    while (g1 < hi) {                                 // first one beyond the last instance's window
                        const int64_t mid = g1 + ((hi - g1) >> 1);
                        if (!W::admits(time_of<TimeT>(otb[mid]), thr_last)) hi = mid; else g1 = mid + 1;
                    }
                    // the out-events ascend in time and so do the instances: ONE cursor into the LDS times follows the scan — `plo`, the first
                    // instance whose window has not ended before the batch's first out-event (its time and threshold sit in registers).  A
                    // batch whose last out-event is not later than that instance is dropped after two compares (most batches when delta is
                    // small against the spacing of the instances); otherwise the few instances that begin before the batch ends are
                    // tested against every lane's out-event directly: c_j = #{i : t_i < t_j <= t_i + delta}.  The scan ends with the last window.
                    int plo = 0;
                    TimeT t_plo = time_of<TimeT>(tms[0]);
                    typename W::Thr thr_plo = W::threshold(t_plo, delta_i, delta_f);
                    constexpr int kAhead = PP_HUBX_AHEAD;                 // batches whose loads are in flight together
                    bool done = false;
                    for (int64_t j00 = g0; j00 < g1 && !done; j00 += kAhead * kWave) {
                        uint64_t tbq[kAhead];
                        uint32_t rkq[kAhead];
#pragma unroll
                        for (int x = 0; x < kAhead; ++x) {
                            const int64_t jx = j00 + x * kWave + l;
                            tbq[x] = jx < g1 ? otb[jx] : 0ull;
                            rkq[x] = jx < g1 ? rkb[jx] : 0xFFFFFFFFu;
                        }
#pragma unroll
                        for (int x = 0; x < kAhead; ++x) {
                            const int64_t j0 = j00 + x * kWave;
                            if (j0 >= g1 || done) break;
                            const uint64_t tb = tbq[x];
                            const int nv = g1 - j0 < kWave ? (int)(g1 - j0) : kWave;
                            const TimeT t0 = time_of<TimeT>(rl_u64(tb, 0)), tl = time_of<TimeT>(rl_u64(tb, nv - 1));
                            while (!W::admits(t0, thr_plo)) {             // windows that ended before this batch
                                if (++plo >= ninst) break;
                                t_plo = time_of<TimeT>(tms[plo]);
                                thr_plo = W::threshold(t_plo, delta_i, delta_f);
                            }
                            if (plo >= ninst) { done = true; break; }
                            if (!(t_plo < tl)) continue;                  // the next open instance begins after the batch
                            const TimeT tj = time_of<TimeT>(tb);
                            int c = 0;
                            float wsum = 0.0f;
                            for (int ii = plo; ii < ninst; ++ii) {
                                const TimeT ti = time_of<TimeT>(tms[ii]);
                                if (!(ti < tl)) break;
                                const bool hit = ti < tj && W::admits(tj, W::threshold(ti, delta_i, delta_f));
                                c += hit ? 1 : 0;
                                if (kW) wsum += hit ? wts[ii] : 0.0f;
                            }
                            const uint32_t rk = rkq[x] - (uint32_t)rbase;
                            if (c > 0 && j0 + l < g1 && rk < (uint32_t)kHubxRuns) {
                                atomicAdd(&hist[rk], (uint32_t)c);
                                if (kW) atomicAdd(&histw[rk], wsum);
                                pairs += c;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const int64_t ninst_run = qp - q;
            const float w1run = kW ? w1sum : (float)ninst_run;
            // ---- epilogue of in-run (cur_a, b) = order-2 node cur_u, one 64-run column after the other (ascending successor)
            const int slot = ord >> 6, olane = ord & (kWave - 1);
#pragma unroll
            for (int cc = 0; cc < kHubxCols; ++cc) {
                if (cc < ncol) {
                    const int64_t rr = rbase + cc * kWave + l;
                    const bool valid = rr < R;
                    const uint32_t v = (uint32_t)(row0 + rr);
                    const int hits = (int)hist[cc * kWave + l];
                    hist[cc * kWave + l] = 0u;
                    float wgt = (float)hits;
                    if (kW) {
                        wgt = histw[cc * kWave + l];
                        histw[cc * kWave + l] = 0.0f;
                    }
                    const bool emit = valid && hits > 0;
                    const uint64_t em = __ballot(emit);
                    const int reached = (int)__popcll(em);
                    const int odsel = slot == 0 ? od0 : (slot == 1 ? od1 : (slot == 2 ? od2 : od3));
                    const int before = rl_i(odsel, olane);
                    if (l == olane) {
                        if (slot == 0) od0 += reached; else if (slot == 1) od1 += reached; else if (slot == 2) od2 += reached; else od3 += reached;
                    }
                    if (!kFill) {
                        if (emit) {
                            ++cnt[cc];
                            if (cur_u == v) lw[cc] = wgt; else deg[cc] += wgt;
                        }
                    } else if (emit) {
                        const float val = cur_u == v ? 0.0f : du_r * wgt * dv[cc];
                        a.in_idx2[ip[cc] + cnt[cc]] = (int32_t)cur_u;
                        a.in_val2[ip[cc] + cnt[cc]] = val;
                        if (a.in_w2) a.in_w2[ip[cc] + cnt[cc]] = wgt;
                        const int rank = before + (int)__popcll(em & lanes_below(l));
                        a.out_pack[ob_r + rank] = make_uint2(v, __float_as_uint(val));
                        ++cnt[cc];
                    }
                }
            }
            if (!kFill && rd == rounds - 1) {
                const int odsel = slot == 0 ? od0 : (slot == 1 ? od1 : (slot == 2 ? od2 : od3));
                const int total = rl_i(odsel, olane);
                if (l == 0 && total != 0 && cur_u < kDb2Foreign) a.outdeg2[cur_u] = total;
                longest_src = total > longest_src ? total : longest_src;
            }
            if (rd == 0) {
                if (!kFill) {
                    ++runs;
                    if (cur_a == b) lw1 = w1run; else deg1 += w1run;
                } else if (l == 0) {
                    a.fwd_idx1[fp + ord] = (int32_t)cur_a;
                    a.fwd_val1[fp + ord] = cur_a == b ? 0.0f : da_r * w1run * d1b;
                    if (a.dst_order) a.dst_order[fp + ord] = (int32_t)cur_u;
                }
            }
            q = qp;
        }
        if (!kFill) {
#pragma unroll
            for (int cc = 0; cc < kHubxCols; ++cc) {
                if (cc < ncol) {
                    const int64_t pi = (part0 + rbase / kWave + cc) * kWave + l;
                    h.part_cnt[pi] = cnt[cc];
                    h.part_deg[pi] = deg[cc];
                    h.part_lw[pi] = lw[cc];
                }
            }
        }
    }
    if (!kFill) {
        if (R == 0 && l < kWave) {                                // (no successor: the one part column k_db2_hub_combine reads)
            h.part_cnt[part0 * kWave + l] = 0;
            h.part_deg[part0 * kWave + l] = 0.0f;
            h.part_lw[part0 * kWave + l] = -1.0f;
        }
        const long long total = wave_sum(pairs);
        const int longest = wave_max(longest_src);
        if (l == 0) {
            h.task_runs[task] = runs;
            h.task_deg1[task] = deg1;
            h.task_lw1[task] = lw1;
            if (total) atomicAdd((unsigned long long*)&h.stats[4], (unsigned long long)total);
            atomicMax((unsigned long long*)&h.stats[6], (unsigned long long)longest);
        }
    }
}

// per hub node: the tasks' partial results in task order -> in-degrees, weighted degrees, self-loop weights of its order-2 rows and the
// exclusive prefix of every task (where its entries start inside a destination-major row); first-order in-degree / degree of the node.
// One workgroup of 8 waves per hub: the task list is cut into 8 contiguous pieces, every wave sums its piece, the pieces' totals meet in LDS,
// every wave rewrites its piece's counts as prefixes (a node with 2*10^6 in-events has 7 800 tasks: 1.7 -> 0.3 ms on BASELINE configs[2]'s
// stream).  Sums in (piece, task) order: fixed, bit-reproducible.
constexpr int kCombineWaves = 8;
__global__ __launch_bounds__(kCombineWaves* kWave) void k_db2_hub_combine(int64_t n_hubs, Db2Mid a, Db2Hub h) {
    __shared__ int s_cnt[kCombineWaves][kWave];
    __shared__ float s_deg[kCombineWaves][kWave], s_lw[kCombineWaves][kWave];
    const int64_t slot = blockIdx.x;
    if (slot >= n_hubs) return;
    const int l = lane_id(), w = wave_id();
    const uint32_t b = h.hub_list[slot];
    const int64_t no = (int64_t)a.tp[b + 1] - a.tp[b], ni = (int64_t)a.hp[b + 1] - a.hp[b];
    const int32_t row0 = a.row_ptr[b];
    const int64_t R = (int64_t)a.row_ptr[b + 1] - row0;
    const int64_t rounds = R > 0 ? (R + kWave - 1) / kWave : 1;
    const int64_t nrc = no > 0 ? ((no < h.num_nodes ? no : h.num_nodes) + kWave - 1) / kWave : 1;
    const int64_t chunk = hub_chunk(ni, no);
    const int64_t ntask = ni > 0 ? (ni + chunk - 1) / chunk : 1;
    const int64_t t0 = h.tbase[b], pb = h.pbase[b];
    const int64_t per = (ntask + kCombineWaves - 1) / kCombineWaves;
    const int64_t k_lo = w * per < ntask ? w * per : ntask, k_hi = k_lo + per < ntask ? k_lo + per : ntask;
    int longest = 0;
    constexpr int kBatch = 16;
    for (int64_t c = 0; c < rounds; ++c) {
        int total = 0;
        float deg = 0.0f, lw = -1.0f;
        for (int64_t k0 = k_lo; k0 < k_hi; k0 += kBatch) {
            int x[kBatch];
            float d[kBatch], t[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const bool in = k0 + j < k_hi;
                const int64_t pi = (pb + (k0 + j) * nrc + c) * kWave + l;
                x[j] = in ? h.part_cnt[pi] : 0;
                d[j] = in ? h.part_deg[pi] : 0.0f;
                t[j] = in ? h.part_lw[pi] : -1.0f;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                total += x[j];
                deg += d[j];
                if (t[j] >= 0.0f) lw = t[j];
            }
        }
        s_cnt[w][l] = total;
        s_deg[w][l] = deg;
        s_lw[w][l] = lw;
        __syncthreads();
        int running = 0;
        for (int ww = 0; ww < w; ++ww) running += s_cnt[ww][l];
        for (int64_t k0 = k_lo; k0 < k_hi; k0 += kBatch) {
            int x[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) x[j] = k0 + j < k_hi ? h.part_cnt[(pb + (k0 + j) * nrc + c) * kWave + l] : 0;
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                if (k0 + j < k_hi) {
                    h.part_cnt[(pb + (k0 + j) * nrc + c) * kWave + l] = running;
                    running += x[j];
                }
            }
        }
        if (w == kCombineWaves - 1) {
            const int64_t rr = c * kWave + l;
            if (rr < R) {
                float dsum = 0.0f, lwf = -1.0f;
                for (int ww = 0; ww < kCombineWaves; ++ww) {
                    dsum += s_deg[ww][l];
                    if (s_lw[ww][l] >= 0.0f) lwf = s_lw[ww][l];
                }
                const uint32_t v = (uint32_t)(row0 + rr);
                const float l2 = lwf < 0.0f ? 1.0f : lwf;          // an existing self loop keeps its weight, every other node gets one of weight 1
                a.indeg2[v] = running;
                a.ho_deg[v] = dsum + l2;
                a.ho_lw[v] = l2;
                longest = running > longest ? running : longest;
            }
        }
        __syncthreads();
    }
    if (w == kCombineWaves - 1) {
        longest = wave_max(longest);
        if (l == 0) atomicMax((unsigned long long*)&h.stats[5], (unsigned long long)longest);
    }
    if (w != 0) return;
    // the node itself: in-runs per task -> first-order in-degree + where every task's first-order entries start; weighted degree
    int base = 0;
    float deg1 = 0.0f, lw1 = -1.0f;
    for (int64_t k0 = 0; k0 < ntask; k0 += kWave) {
        const bool in = k0 + l < ntask;
        const int r = in ? h.task_runs[t0 + k0 + l] : 0;
        const int inc = wave_inclusive_sum(r);
        if (in) h.task_runbase[t0 + k0 + l] = base + inc - r;
        base += rl_i(inc, kWave - 1);
        const float dl = in ? h.task_deg1[t0 + k0 + l] : 0.0f, tl_ = in ? h.task_lw1[t0 + k0 + l] : -1.0f;
        const float tmax = wave_max(tl_);                          // (at most one task met the node's own self loop)
        if (tmax >= 0.0f) lw1 = tmax;
        deg1 += wave_sum(dl);                                      // a fixed-order tree over the 64 tasks, the blocks of 64 in order
    }
    if (l == 0) {
        const float l1 = lw1 < 0.0f ? 1.0f : lw1;
        a.nu[b] = base;
        a.pc[b] = 0;                                               // (the hubs' lifted pairs are counted in stats[4])
        a.fo_deg[b] = deg1 + l1;
        a.fo_lw[b] = l1;
        a.node_simple[b] = 0;
        atomicMax((unsigned long long*)&h.stats[7], (unsigned long long)base);
    }
}

// ------------------------------------------------------------------ partition shards: halo numbering and send lists
// One rank owns the nodes [lo, lo + n_own) and holds the events that touch them.  Its order-2 rows are the nodes (b, .) of its b; the
// sources (a, b) with a foreign a are HALO rows, numbered behind the owned rows in the order (owner of a, b, a): the owner q of a sends its
// rows (a, b) for this rank ordered by (b, a) (k_db2_send_keys + a stable sort by b), so the rows of one all-to-all arrive exactly in
// halo order — no ids travel, no request round.
__device__ __forceinline__ int owner_of(int64_t node, const int64_t* __restrict__ cuts, int world) {
    int r = 0;
    while (r + 1 < world && node >= cuts[r + 1]) ++r;
    return r;
}

// key of every in-event position: the owner of its source node at the first event of a foreign (source, head) run, `world` elsewhere
__global__ __launch_bounds__(kBlock) void k_db2_halo_keys(int64_t m, int64_t lo, int64_t n_own, const uint32_t* __restrict__ hkeys_s,
                                                         const uint32_t* __restrict__ hp, const uint32_t* __restrict__ is_a,
                                                         const uint32_t* __restrict__ is_u, const int64_t* __restrict__ cuts, int world,
                                                         uint32_t* __restrict__ keys) {
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= m) return;
    const uint32_t b = hkeys_s[q];
    const int64_t bl = (int64_t)b - lo;
    uint32_t key = (uint32_t)world;
    if (bl >= 0 && bl < n_own && is_u[q] == kDb2Foreign && (q == (int64_t)hp[b] || is_a[q - 1] != is_a[q])) key = (uint32_t)owner_of(is_a[q], cuts, world);
    keys[q] = key;
}

// the k-th foreign run in (owner, head, source) order is halo row k = local source id U2_own + k
__global__ __launch_bounds__(kBlock) void k_db2_halo_assign(int64_t m, const uint32_t* __restrict__ keys_s, const uint32_t* __restrict__ order,
                                                           int world, const int64_t* __restrict__ result, uint32_t* __restrict__ is_u) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= m || keys_s[k] >= (uint32_t)world) return;
    is_u[order[k]] = (uint32_t)result[0] + (uint32_t)k;
}

// sort key of every owned order-2 row (b, c): c when another rank owns c (that rank gathers from the row), num_nodes otherwise
__global__ __launch_bounds__(kBlock) void k_db2_send_keys(int64_t cap, const int64_t* __restrict__ result, const int32_t* __restrict__ fo_bwd_idx,
                                                         int64_t lo, int64_t n_own, int64_t num_nodes, uint32_t* __restrict__ keys) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= cap) return;
    uint32_t key = (uint32_t)(num_nodes + n_own);           // (rows beyond U2: behind everything)
    if (r < result[0]) {
        const int64_t c = fo_bwd_idx[r];
        key = (c < lo || c >= lo + n_own) ? (uint32_t)c : (uint32_t)(num_nodes + (c - lo));      // kept rows: behind the sent ones, by successor as well
    }
    keys[r] = key;
}

// bipartite "last" plan of a partition shard without another sort: in the local row order ALL rows are grouped by their successor c (the send
// prefix by foreign c, the kept rows by own c), so the rows that end in c are one contiguous range.  Destinations live in the rank-major
// padded layout (node c of rank r at r * cap + c - cuts[r]: the partial sums reduce-scatter without a copy).
__global__ __launch_bounds__(kBlock) void k_db2_bip_count(int64_t cap_m, const int64_t* __restrict__ result, const int32_t* __restrict__ succ,
                                                         const int64_t* __restrict__ cuts, int world, int64_t cap_n, int32_t* __restrict__ bwd_idx,
                                                         int32_t* __restrict__ bwd_ptr, int32_t* __restrict__ cnt) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k > cap_m || k > result[0]) return;
    bwd_ptr[k] = (int32_t)k;                               // every order-2 row ends in exactly one first-order node
    if (k == result[0]) return;
    const int64_t c = succ[k];
    const int r = owner_of(c, cuts, world);
    const int32_t pd = (int32_t)(r * cap_n + (c - cuts[r]));
    bwd_idx[k] = pd;
    atomicAdd(&cnt[pd], 1);
}

// fwd_idx: the local rows in padded-destination order = [rows for ranks below me | kept rows | rows for ranks above me]; self_coef = in-degree
__global__ __launch_bounds__(kBlock) void k_db2_bip_fill(int64_t total, const int64_t* __restrict__ result, const uint32_t* __restrict__ send_keys_s,
                                                        int64_t cap_m, int64_t lo, int64_t num_nodes, int64_t n_pad, void* cnt_then_self_coef,
                                                        int32_t* __restrict__ fwd_idx) {
    // the per-destination counts (int32) are converted IN PLACE into the self coefficients (float): one buffer, one pointer
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= total) return;
    if (j < n_pad) {
        const int32_t c = ((const int32_t*)cnt_then_self_coef)[j];
        ((float*)cnt_then_self_coef)[j] = (float)c;
    }
    const int64_t u2 = result[0];
    if (j >= u2) return;
    const int64_t n_below = lower_bound_dev<uint32_t, int64_t>(send_keys_s, 0, cap_m, (uint32_t)lo);              // rows whose successor lies below my range
    const int64_t n_send = lower_bound_dev<uint32_t, int64_t>(send_keys_s, n_below, cap_m, (uint32_t)num_nodes);
    const int64_t n_kept = u2 - n_send;
    fwd_idx[j] = (int32_t)(j < n_below ? j : (j < n_below + n_kept ? n_send + (j - n_below) : j - n_kept));
}

// send_ptr[r] = first position of the sorted send keys that belongs to rank r (r = 0 .. world); recv_ptr from the sorted halo keys; both
// published as int64 behind the sizes: result[8 ..] = recv_ptr[world + 1], then send_ptr[world + 1]; result[5] = halo rows, [6] = rows sent
__global__ void k_db2_publish(int64_t m, const uint32_t* __restrict__ halo_keys_s, const uint32_t* __restrict__ send_keys_s,
                              const int64_t* __restrict__ cuts, int world, int64_t num_nodes, int64_t* __restrict__ result) {
    const int r = threadIdx.x;
    if (r > world) return;
    const uint32_t hk = (uint32_t)r;                                        // halo keys are owners
    const uint32_t sk = r == world ? (uint32_t)num_nodes : (uint32_t)cuts[r];   // send keys are head nodes
    const int64_t hpos = lower_bound_dev<uint32_t, int64_t>(halo_keys_s, 0, m, hk);
    const int64_t spos = lower_bound_dev<uint32_t, int64_t>(send_keys_s, 0, m, sk);
    result[8 + r] = hpos;
    result[8 + world + 1 + r] = spos;
    if (r == world) { result[5] = hpos; result[6] = spos; }
}

// The sorted send keys define the LOCAL ROW ORDER of a partition shard: rows other ranks gather from come first, grouped by that rank and
// ordered by (head node, source node) — the send list of every layer exchange is the contiguous prefix [0, rows sent) of every row matrix,
// no pack; rows nobody gathers from follow.  perm[lexicographic row] = local row; the first-order edge list moves to the local order;
// send_slot[local row] = the row itself inside the prefix, -1 behind it.
__global__ __launch_bounds__(kBlock) void k_db2_apply_perm(int64_t m, const int64_t* __restrict__ result, const uint32_t* __restrict__ send_keys_s,
                                                          const uint32_t* __restrict__ order, int64_t num_nodes, const int32_t* __restrict__ succ_old,
                                                          const float* __restrict__ w_old, int32_t* __restrict__ perm, int32_t* __restrict__ fo_bwd_idx,
                                                          float* __restrict__ fo_w, int32_t* __restrict__ send_slot, int32_t* __restrict__ row_of) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= m || k >= result[0]) return;
    const uint32_t old = order[k];
    perm[old] = (int32_t)k;
    row_of[k] = (int32_t)old;
    fo_bwd_idx[k] = succ_old[old];
    fo_w[k] = w_old[old];
    send_slot[k] = send_keys_s[k] < (uint32_t)num_nodes ? (int32_t)k : -1;
}

// ------------------------------------------------------------------ workspace
constexpr int kDb2MaxWorld = 64;
constexpr int kDb2HubStatCount = 16;
constexpr int kDb2Result = 8 + 2 * (kDb2MaxWorld + 1) + kDb2HubStatCount;
static_assert(kDb2HubStats == 8 + 2 * (kDb2MaxWorld + 1), "hub block of the result header");

struct Db2Ws {
    int64_t* result;         // [kDb2Result]: {U2, status, A2, E2, A1 (first-order in-edges), halo rows, rows sent, -, recv_ptr[world+1], send_ptr[world+1]}
    uint32_t *xkeys, *xkeys_s, *xorder;      // partition shards: halo / send-list sort keys
    int32_t *perm, *succ_old, *fblk;
    float* w_old;
    uint8_t* fskip;
    uint32_t *tkeys, *tkeys_s, *hkeys_s, *tl, *hl, *tp, *hp;
    Db2Rec* rec;
    Db2Src* src_t;
    uint32_t *oc_t, *oc_s;
    uint64_t *ot_t, *ot_s;
    float *ow_t, *ow_s;
    uint8_t *ocr_s, *ocr_t;
    uint64_t* is_t;
    uint32_t *is_a, *is_u;
    float *is_w, *du_s, *da_s;
    uint64_t* run_em;         // [m] count -> fill, see Db2Mid
    uint8_t* node_simple;     // [n]
    int32_t* ob_s;
    uint2* row_pack;
    int32_t *blk, *nu, *pc, *indeg2, *outdeg2;
    float *ho_lw, *fo_lw;
    int64_t* pc_scan;
    // hub nodes: tables sized by (m, n); everything sized by the hubs themselves lives in the caller's hub workspace (Db2HubWs)
    uint8_t* hub_flag;
    uint32_t *hoff, *oslot, *tbase, *task_node, *hub_list;
    int64_t* pbase;
    int32_t *task_runs, *task_runbase;
    float *task_deg1, *task_lw1;
    void* scratch;
    size_t scratch_bytes, total_bytes;
};

static inline int64_t db2_task_cap(int64_t m, int64_t n) { return m / kHubChunk + n + 1 + 256 * (m / kHubLongOut); }      // sum over hubs of max(1, ceil(in-events / chunk)); hub_chunk

static Db2Ws carve_db2(void* ws, int64_t m, int64_t n) {
    Arena a(ws, (size_t)-1);
    Db2Ws w;
    w.result = a.take<int64_t>(kDb2Result);
    w.xkeys = a.take<uint32_t>(m);
    w.xkeys_s = a.take<uint32_t>(m);
    w.xorder = a.take<uint32_t>(m);
    w.perm = a.take<int32_t>(m);
    w.succ_old = a.take<int32_t>(m);
    w.w_old = a.take<float>(m);
    w.fblk = a.take<int32_t>(n);
    w.fskip = a.take<uint8_t>(n + 16);
    w.tkeys = a.take<uint32_t>(m);
    w.tkeys_s = a.take<uint32_t>(m);
    w.hkeys_s = a.take<uint32_t>(m);
    w.tl = a.take<uint32_t>(m);
    w.hl = a.take<uint32_t>(m);
    w.tp = a.take<uint32_t>(n + 2);
    w.hp = a.take<uint32_t>(n + 2);
    w.rec = a.take<Db2Rec>(m);
    w.src_t = a.take<Db2Src>(m);
    w.oc_t = a.take<uint32_t>(m);
    w.oc_s = a.take<uint32_t>(m);
    w.ot_t = a.take<uint64_t>(m);
    w.ot_s = a.take<uint64_t>(m);
    w.ow_t = a.take<float>(m);
    w.ow_s = a.take<float>(m);
    w.ocr_s = a.take<uint8_t>(m + 16);
    w.ocr_t = a.take<uint8_t>(m + 16);
    w.is_t = a.take<uint64_t>(m);
    w.is_a = a.take<uint32_t>(m);
    w.is_u = a.take<uint32_t>(m);
    w.is_w = a.take<float>(m);
    w.du_s = a.take<float>(m);
    w.da_s = a.take<float>(m);
    w.ob_s = a.take<int32_t>(m);
    w.run_em = a.take<uint64_t>(m);
    w.node_simple = a.take<uint8_t>(n);
    w.row_pack = a.take<uint2>(m);
    w.blk = a.take<int32_t>(n);
    w.nu = a.take<int32_t>(n);
    w.pc = a.take<int32_t>(n);
    w.indeg2 = a.take<int32_t>(m);
    w.outdeg2 = a.take<int32_t>(m);
    w.ho_lw = a.take<float>(m);
    w.fo_lw = a.take<float>(n);
    w.pc_scan = a.take<int64_t>(n + 1);
    w.hub_flag = a.take<uint8_t>(n + 16);
    w.hoff = a.take<uint32_t>(n);
    w.oslot = a.take<uint32_t>(n);
    w.tbase = a.take<uint32_t>(n);
    w.hub_list = a.take<uint32_t>(n);
    w.pbase = a.take<int64_t>(n);
    const int64_t tcap = db2_task_cap(m, n);
    w.task_node = a.take<uint32_t>(tcap);
    w.task_runs = a.take<int32_t>(tcap);
    w.task_runbase = a.take<int32_t>(tcap);
    w.task_deg1 = a.take<float>(tcap);
    w.task_lw1 = a.take<float>(tcap);
    size_t sb = scan_ws_bytes(m > n ? m : n), s2 = sort_ws_bytes(m, 4);
    w.scratch_bytes = s2 > sb ? s2 : sb;
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

struct Db2HubWs {
    uint64_t *keys, *keys_s;         // [H]
    uint32_t *pos, *pos_s, *head, *head_scan, *rank_s, *rank_t;      // [H] ([H + 1] for the scan)
    uint32_t* run_start;             // [H + out-hubs + 1]
    int32_t* part_cnt;               // [P * 64]
    float *part_deg, *part_lw;
    void* scratch;
    size_t scratch_bytes, total_bytes;
};

static Db2HubWs carve_db2_hub(void* ws, int64_t h_out, int64_t out_hubs, int64_t parts) {
    Arena a(ws, (size_t)-1);
    Db2HubWs w;
    w.keys = a.take<uint64_t>(h_out);
    w.keys_s = a.take<uint64_t>(h_out);
    w.pos = a.take<uint32_t>(h_out);
    w.pos_s = a.take<uint32_t>(h_out);
    w.head = a.take<uint32_t>(h_out);
    w.head_scan = a.take<uint32_t>(h_out + 1);
    w.rank_s = a.take<uint32_t>(h_out);
    w.rank_t = a.take<uint32_t>(h_out);
    w.run_start = a.take<uint32_t>(h_out + out_hubs + 1);
    w.part_cnt = a.take<int32_t>(parts * kWave);
    w.part_deg = a.take<float>(parts * kWave);
    w.part_lw = a.take<float>(parts * kWave);
    const size_t sb = scan_ws_bytes(h_out), s2 = sort_ws_bytes(h_out, 8);
    w.scratch_bytes = s2 > sb ? s2 : sb;
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

struct Db2HubSizes {                 // what pp_debruijn2_lists reported (0 everywhere: no hub node)
    int64_t hubs, out_hubs, out_events, tasks, parts;
};

static void hub_common(Db2Hub& h, const Db2Ws& w, const Db2HubWs& hw, int64_t n) {
    h.flag = w.hub_flag; h.hoff = w.hoff; h.oslot = w.oslot; h.tbase = w.tbase; h.pbase = w.pbase; h.task_node = w.task_node; h.hub_list = w.hub_list;
    h.run_start = hw.run_start; h.part_cnt = hw.part_cnt; h.part_deg = hw.part_deg; h.part_lw = hw.part_lw;
    h.task_runs = w.task_runs; h.task_runbase = w.task_runbase; h.task_deg1 = w.task_deg1; h.task_lw1 = w.task_lw1;
    h.stats = w.result + kDb2HubStats; h.num_nodes = n; h.succ = nullptr;
}

template <typename TimeT, int kMode, bool kFill>
static void launch_hub(bool weighted, unsigned grid, hipStream_t st, int64_t tasks, int64_t di, double df, const Db2Mid& a, const Db2Hub& h) {
    if (weighted) k_db2_hub<TimeT, kMode, kFill, true><<<grid, kBlock, 0, st>>>(tasks, di, df, a, h);
    else k_db2_hub<TimeT, kMode, kFill, false><<<grid, kBlock, 0, st>>>(tasks, di, df, a, h);
}

template <typename TimeT, int kMode, bool kFill>
static void launch_hubx(bool weighted, unsigned grid, hipStream_t st, int64_t tasks, int64_t di, double df, const Db2Mid& a, const Db2Hub& h,
                        const uint64_t* ot_t, const uint32_t* rank_t) {
    if (weighted) k_db2_hubx<TimeT, kMode, kFill, true><<<grid, kBlock, 0, st>>>(tasks, di, df, a, h, ot_t, rank_t);
    else k_db2_hubx<TimeT, kMode, kFill, false><<<grid, kBlock, 0, st>>>(tasks, di, df, a, h, ot_t, rank_t);
}

template <bool kFill>
static int launch_hubx_any(int time_dtype, int delta_kind, bool weighted, hipStream_t st, int64_t tasks, int64_t di, double df, const Db2Mid& a,
                           const Db2Hub& h, const uint64_t* ot_t, const uint32_t* rank_t) {
    const unsigned grid = (unsigned)ceil_div(tasks, kWavesPerBlock);
    if (time_dtype == PP_F64) launch_hubx<double, 0, kFill>(weighted, grid, st, tasks, di, df, a, h, ot_t, rank_t);
    else if (delta_kind == PP_DELTA_I64) launch_hubx<int64_t, 0, kFill>(weighted, grid, st, tasks, di, df, a, h, ot_t, rank_t);
    else if (delta_kind == PP_DELTA_F32) launch_hubx<int64_t, 1, kFill>(weighted, grid, st, tasks, di, df, a, h, ot_t, rank_t);
    else launch_hubx<int64_t, 2, kFill>(weighted, grid, st, tasks, di, df, a, h, ot_t, rank_t);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <bool kFill>
static int launch_hub_any(int time_dtype, int delta_kind, bool weighted, hipStream_t st, int64_t tasks, int64_t di, double df, const Db2Mid& a,
                          const Db2Hub& h) {
    const unsigned grid = (unsigned)ceil_div(tasks, kWavesPerBlock);
    if (time_dtype == PP_F64) launch_hub<double, 0, kFill>(weighted, grid, st, tasks, di, df, a, h);
    else if (delta_kind == PP_DELTA_I64) launch_hub<int64_t, 0, kFill>(weighted, grid, st, tasks, di, df, a, h);
    else if (delta_kind == PP_DELTA_F32) launch_hub<int64_t, 1, kFill>(weighted, grid, st, tasks, di, df, a, h);
    else launch_hub<int64_t, 2, kFill>(weighted, grid, st, tasks, di, df, a, h);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

static void mid_common(Db2Mid& a, const Db2Ws& w, const int32_t* row_ptr, bool weighted) {
    a.tp = w.tp; a.hp = w.hp; a.ot_s = w.ot_s; a.ocr_s = w.ocr_s; a.row_ptr = row_ptr;
    a.is_t = w.is_t; a.is_a = w.is_a; a.is_u = w.is_u; a.is_w = weighted ? w.is_w : nullptr;
    a.ho_lw = w.ho_lw; a.fo_lw = w.fo_lw; a.run_em = w.run_em; a.node_simple = w.node_simple;
}

template <typename TimeT, int kMode, bool kFill>
static void launch_mid(bool weighted, unsigned grid, hipStream_t st, int64_t n, int64_t di, double df, const Db2Mid& a) {
    if (weighted) k_db2_mid<TimeT, kMode, kFill, true><<<grid, kBlock, 0, st>>>(n, di, df, a);
    else k_db2_mid<TimeT, kMode, kFill, false><<<grid, kBlock, 0, st>>>(n, di, df, a);
}

template <bool kFill>
static int launch_mid_any(int time_dtype, int delta_kind, bool weighted, unsigned grid, hipStream_t st, int64_t n, int64_t di, double df, const Db2Mid& a) {
    if (time_dtype == PP_F64) launch_mid<double, 0, kFill>(weighted, grid, st, n, di, df, a);
    else if (delta_kind == PP_DELTA_I64) launch_mid<int64_t, 0, kFill>(weighted, grid, st, n, di, df, a);
    else if (delta_kind == PP_DELTA_F32) launch_mid<int64_t, 1, kFill>(weighted, grid, st, n, di, df, a);
    else launch_mid<int64_t, 2, kFill>(weighted, grid, st, n, di, df, a);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp

using namespace pp;

extern "C" {

size_t pp_debruijn2_ws_bytes(int64_t m, int64_t num_nodes) { return carve_db2(nullptr, m > 0 ? m : 0, num_nodes > 0 ? num_nodes : 0).total_bytes; }

size_t pp_debruijn2_hub_ws_bytes(int64_t hub_out_events, int64_t out_hubs, int64_t hub_parts) {
    return carve_db2_hub(nullptr, hub_out_events > 0 ? hub_out_events : 0, out_hubs > 0 ? out_hubs : 0, hub_parts > 0 ? hub_parts : 0).total_bytes;
}

}  // extern "C"

namespace pp {

struct Db2Part {                   // node range of a partition shard (one GPU: lo = 0, n_own = num_nodes, world = 1, the rest unused)
    int64_t lo, n_own;
    const int64_t* cuts;           // device int64 [world + 1]: node ranges of all ranks
    int world, me;
    int32_t *send_slot, *row_of;   // [m] each: local row -> its position in the send prefix (-1 behind it) / its lexicographic row
    int32_t* fo2_bwd_ptr;          // [num_nodes + 1]: source-major row pointers of the first-order shard (dense local source order)
    int64_t cap_n;                 // rows per rank of the padded first-order layout (bipartite partial sums)
    int32_t *bip_fwd_ptr, *bip_fwd_idx, *bip_bwd_ptr, *bip_bwd_idx;      // [world * cap_n + 1], [m], [m + 1], [m]
    float* bip_self;               // [world * cap_n]
};

// Recorded behind the asynchronous copies of the builder's statistics / sizes.  An event belongs to the device that was current when it was made and can
// only be recorded on a stream of that device: one event per (thread, device), made on first use (one thread may build on several devices);
// `tls_stats_event` is the one recorded last — what pp_debruijn2_wait waits for.
static constexpr int kDb2MaxDevices = 64;
static thread_local hipEvent_t tls_stats_events[kDb2MaxDevices] = {};
static thread_local hipEvent_t tls_stats_event = nullptr;
static int record_stats_event(hipStream_t st) {
    int dev = 0;
    PP_HIP(hipGetDevice(&dev));
    PP_REQUIRE(dev >= 0 && dev < kDb2MaxDevices, PP_ERR_ARG, "pp_debruijn2: device ordinal %d out of range", dev);
    if (!tls_stats_events[dev]) PP_HIP(hipEventCreateWithFlags(&tls_stats_events[dev], hipEventDisableTiming));
    tls_stats_event = tls_stats_events[dev];
    PP_HIP(hipEventRecord(tls_stats_event, st));
    return PP_OK;
}

// 1. event records; out-lists (stable sort by tail: time order inside a list), then the list SEQUENCE sorted by head: in-lists in (source, time)
//    order; row pointers of both; one GPU: hub classification, its totals copied to `host_stats` (asynchronously), then the out-side kernel of
//    the other nodes — the caller's wait for the totals runs under it
static int db2_lists(const char* who, const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t n, const Db2Part& pt,
                     const float* weight, void* ws, size_t ws_bytes, int64_t* host_stats, hipStream_t st) {
    const bool part = pt.world > 1;
    const int64_t n_own = pt.n_own;
    PP_REQUIRE(m >= 0 && n >= 0, PP_ERR_ARG, "%s: negative size", who);
    PP_REQUIRE(m < (int64_t)0x7ffffff0 && n < (int64_t)0x7ffffff0, PP_ERR_TOO_LARGE, "%s: m or num_nodes >= 2^31", who);
    PP_REQUIRE(time_dtype == PP_I64 || time_dtype == PP_F64, PP_ERR_ARG, "%s: time must be int64 or float64", who);
    PP_REQUIRE(pt.lo >= 0 && n_own >= 0 && pt.lo + n_own <= n && pt.world >= 1 && pt.world <= kDb2MaxWorld && pt.me >= 0 && pt.me < pt.world, PP_ERR_ARG,
               "%s: bad node range / world", who);
    Db2Ws w = carve_db2(ws, m, n);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "%s: workspace too small", who);
    PP_HIP(hipMemsetAsync(w.result, 0, kDb2Result * sizeof(int64_t), st));
    if (m == 0 || n == 0) {
        if (host_stats) {
            PP_HIP(hipMemcpyAsync(host_stats, w.result + kDb2HubStats, kDb2HubStatCount * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            { const int erc = record_stats_event(st); if (erc != PP_OK) return erc; }
        }
        return PP_OK;
    }
    const unsigned egrid = (unsigned)ceil_div(m, kBlock);
    const unsigned agrid = (unsigned)ceil_div(n, kWavesPerBlock * kDb2OutNodes);
    if (time_dtype == PP_I64) k_db2_keys<int64_t><<<egrid, kBlock, 0, st>>>(edge_index, (const int64_t*)time, m, n, w.tkeys, w.rec, w.result + 1);
    else k_db2_keys<double><<<egrid, kBlock, 0, st>>>(edge_index, (const double*)time, m, n, w.tkeys, w.rec, w.result + 1);
    PP_LAUNCH_CHECK();
    const int key_bits = bits_for((uint64_t)(n > 0 ? n - 1 : 0));
    int rc = sort_pairs<uint32_t>(w.tkeys, nullptr, w.tkeys_s, w.tl, m, 0, key_bits, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_db2_gather_out<<<egrid, kBlock, 0, st>>>(m, w.tl, w.rec, weight, w.oc_t, w.ot_t, w.ow_t);
    PP_LAUNCH_CHECK();
    rc = sort_pairs<uint32_t>(w.oc_t, nullptr, w.hkeys_s, w.hl, m, 0, key_bits, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_db2_rowptr<<<(unsigned)ceil_div(m + 1, kBlock), kBlock, 0, st>>>(w.tkeys_s, m, n, w.tp);
    PP_LAUNCH_CHECK();
    k_db2_rowptr<<<(unsigned)ceil_div(m + 1, kBlock), kBlock, 0, st>>>(w.hkeys_s, m, n, w.hp);
    PP_LAUNCH_CHECK();
    if (!part) {
        k_db2_hub_classify<<<(unsigned)ceil_div(n, kBlock), kBlock, 0, st>>>(n, w.tp, w.hp, w.hub_flag, w.hoff, w.oslot, w.tbase, w.pbase, w.hub_list,
                                                                           w.result + kDb2HubStats);
        PP_LAUNCH_CHECK();
        if (host_stats) {
            PP_HIP(hipMemcpyAsync(host_stats, w.result + kDb2HubStats, kDb2HubStatCount * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            { const int erc = record_stats_event(st); if (erc != PP_OK) return erc; }
        }
    }
    // 2. successors of every node with at most 64 out-events -> block sizes of the order-2 node ids
    int32_t* fblk = part ? w.fblk : nullptr;
    const int hubs_handled = part ? 0 : 1;
    if (weight) k_db2_out<true><<<agrid, kBlock, 0, st>>>(n, pt.lo, n_own, w.tp, w.oc_t, w.ot_t, w.ow_t, w.ot_s, w.oc_s, w.ow_s, w.ocr_s, w.ocr_t, w.blk, fblk, w.fskip, w.result + 1, hubs_handled);
    else k_db2_out<false><<<agrid, kBlock, 0, st>>>(n, pt.lo, n_own, w.tp, w.oc_t, w.ot_t, w.ow_t, w.ot_s, w.oc_s, w.ow_s, w.ocr_s, w.ocr_t, w.blk, fblk, w.fskip, w.result + 1, hubs_handled);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

static int db2_count(const char* who, int time_dtype, int64_t m, int64_t n, const Db2Part& pt, int delta_kind, int64_t delta_i, double delta_f,
                     const float* weight, int32_t* fo_bwd_ptr, int32_t* fo_bwd_idx, float* fo_w, int32_t* fo_fwd_ptr, int32_t* ho_fwd_ptr,
                     int32_t* ho_bwd_ptr, float* ho_deg, float* fo_deg, void* ws, size_t ws_bytes, const Db2HubSizes& hs, void* hub_ws,
                     size_t hub_ws_bytes, int64_t* host_result, hipStream_t st) {
    const bool part = pt.world > 1;
    const int64_t n_own = pt.n_own;
    PP_REQUIRE(m >= 0 && n >= 0, PP_ERR_ARG, "%s: negative size", who);
    PP_REQUIRE(delta_kind >= PP_DELTA_I64 && delta_kind <= PP_DELTA_F64, PP_ERR_ARG, "%s: bad delta kind", who);
    Db2Ws w = carve_db2(ws, m, n);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "%s: workspace too small", who);
    if (m == 0 || n == 0) {
        PP_HIP(hipMemsetAsync(fo_bwd_ptr, 0, (size_t)(n_own + 1) * sizeof(int32_t), st));
        PP_HIP(hipMemsetAsync(fo_fwd_ptr, 0, (size_t)(n_own + 1) * sizeof(int32_t), st));
        PP_HIP(hipMemsetAsync(ho_fwd_ptr, 0, (size_t)(m + 1) * sizeof(int32_t), st));
        PP_HIP(hipMemsetAsync(ho_bwd_ptr, 0, (size_t)(m + 1) * sizeof(int32_t), st));
        return PP_OK;
    }
    PP_REQUIRE(hs.hubs >= 0 && hs.hubs <= n && hs.out_hubs >= 0 && hs.out_hubs <= hs.hubs && hs.out_events >= 0 && hs.out_events <= m && hs.tasks >= hs.hubs &&
               hs.tasks <= db2_task_cap(m, n) && hs.parts >= hs.tasks && (hs.hubs == 0 || !part), PP_ERR_ARG, "%s: bad hub sizes", who);
    Db2HubWs hw = carve_db2_hub(hs.hubs > 0 ? hub_ws : nullptr, hs.out_events, hs.out_hubs, hs.parts);
    PP_REQUIRE(hs.hubs == 0 || (hub_ws != nullptr && hub_ws_bytes >= hw.total_bytes), PP_ERR_WORKSPACE, "%s: hub workspace too small", who);
    const unsigned egrid = (unsigned)ceil_div(m, kBlock), ngrid = (unsigned)ceil_div(n_own > 0 ? n_own : 1, kWavesPerBlock * kDb2Nodes);
    const int key_bits = bits_for((uint64_t)(n > 0 ? n - 1 : 0));
    int rc;
    if (hs.hubs > 0) {
        k_db2_hub_tasks<<<(unsigned)hs.hubs, kBlock, 0, st>>>(hs.hubs, w.hub_list, w.hp, w.tp, w.tbase, w.task_node);
        PP_LAUNCH_CHECK();
    }
    // 2'. out-hubs: their out-events in (successor, time) order by one sort over exactly those events
    if (hs.out_events > 0) {
        const unsigned hgrid = (unsigned)ceil_div(hs.out_events, kBlock);
        k_db2_hub_out_keys<<<egrid, kBlock, 0, st>>>(m, w.tkeys_s, w.tp, w.hub_flag, w.hoff, w.oc_t, key_bits, hw.keys, hw.pos);
        PP_LAUNCH_CHECK();
        rc = sort_pairs<uint64_t>(hw.keys, hw.pos, hw.keys_s, hw.pos_s, hs.out_events, 0, key_bits + bits_for((uint64_t)hs.out_events), hw.scratch,
                                  hw.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        k_db2_hub_out_heads<<<hgrid, kBlock, 0, st>>>(hs.out_events, hw.keys_s, hw.pos_s, w.tkeys_s, w.hoff, hw.head);
        PP_LAUNCH_CHECK();
        rc = exclusive_scan<uint32_t, uint32_t>(hw.head, hs.out_events, hw.head_scan, true, nullptr, hw.scratch, hw.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        if (weight) {
            k_db2_hub_out_write<true><<<hgrid, kBlock, 0, st>>>(hs.out_events, hw.keys_s, hw.pos_s, hw.head, hw.head_scan, w.tkeys_s, w.tp, w.hoff, w.oslot, key_bits,
                                                               w.ot_t, w.ow_t, w.ot_s, w.oc_s, w.ow_s, hw.rank_s, hw.rank_t, hw.run_start, w.blk,
                                                               w.result + kDb2HubStats);
            PP_LAUNCH_CHECK();
            k_db2_hub_out_runs<true><<<hgrid, kBlock, 0, st>>>(hs.out_events, hw.pos_s, hw.head, hw.rank_s, w.tkeys_s, w.tp, w.hoff, w.oslot, hw.run_start, w.ow_s);
        } else {
            k_db2_hub_out_write<false><<<hgrid, kBlock, 0, st>>>(hs.out_events, hw.keys_s, hw.pos_s, hw.head, hw.head_scan, w.tkeys_s, w.tp, w.hoff, w.oslot, key_bits,
                                                                w.ot_t, w.ow_t, w.ot_s, w.oc_s, w.ow_s, hw.rank_s, hw.rank_t, hw.run_start, w.blk,
                                                                w.result + kDb2HubStats);
            PP_LAUNCH_CHECK();
            k_db2_hub_out_runs<false><<<hgrid, kBlock, 0, st>>>(hs.out_events, hw.pos_s, hw.head, hw.rank_s, w.tkeys_s, w.tp, w.hoff, w.oslot, hw.run_start, w.ow_s);
        }
        PP_LAUNCH_CHECK();
    }
    const uint32_t* hoff = hs.out_events > 0 ? w.hoff : nullptr;
    const uint32_t *rank_s = hs.out_events > 0 ? hw.rank_s : nullptr, *rank_t = hs.out_events > 0 ? hw.rank_t : nullptr;
    if (part) {
        rc = exclusive_scan<int32_t, int32_t>(w.fblk, n, pt.fo2_bwd_ptr, true, nullptr, w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
    }
    rc = exclusive_scan<int32_t, int32_t>(w.blk, n_own, fo_bwd_ptr, true, w.result, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    const int32_t* perm = nullptr;
    if (!part) {
        k_db2_out_ids<<<egrid, kBlock, 0, st>>>(m, w.tp, w.tkeys_s, w.oc_s, w.ow_s, w.ocr_s, w.ot_t, w.ocr_t, fo_bwd_ptr, fo_bwd_idx, fo_w, w.src_t, hoff,
                                               rank_s, rank_t);
        PP_LAUNCH_CHECK();
    } else {
        // 2b. local row order = send order (who gathers from my rows), one node-id sort of the successors
        k_db2_out_heads<<<egrid, kBlock, 0, st>>>(m, pt.lo, n_own, w.tp, w.tkeys_s, w.oc_s, w.ow_s, w.ocr_s, fo_bwd_ptr, w.succ_old, w.w_old, nullptr, nullptr);
        PP_LAUNCH_CHECK();
        k_db2_send_keys<<<egrid, kBlock, 0, st>>>(m, w.result, w.succ_old, pt.lo, n_own, n, w.xkeys);
        PP_LAUNCH_CHECK();
        rc = sort_pairs<uint32_t>(w.xkeys, nullptr, w.xkeys_s, w.xorder, m, 0, bits_for((uint64_t)(n + n_own)), w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        PP_HIP(hipMemsetAsync(w.perm, 0, (size_t)m * sizeof(int32_t), st));      // (lanes beyond a node's block read perm[row0 + l]: any valid row will do)
        k_db2_apply_perm<<<egrid, kBlock, 0, st>>>(m, w.result, w.xkeys_s, w.xorder, n, w.succ_old, w.w_old, w.perm, fo_bwd_idx, fo_w, pt.send_slot, pt.row_of);
        PP_LAUNCH_CHECK();
        perm = w.perm;
        // 2b'. bipartite plan from the same order (no sort): count per padded destination (in the buffer of its self coefficients), scan, rotation
        const int64_t n_pad = (int64_t)pt.world * pt.cap_n;
        PP_REQUIRE(pt.cap_n >= 1 && n_pad < (int64_t)0x7ffffff0 && scan_ws_bytes(n_pad) <= w.scratch_bytes, PP_ERR_ARG, "%s: bad padded first-order layout", who);
        int32_t* bcnt = (int32_t*)pt.bip_self;
        PP_HIP(hipMemsetAsync(bcnt, 0, (size_t)n_pad * sizeof(int32_t), st));
        k_db2_bip_count<<<(unsigned)ceil_div(m + 1, kBlock), kBlock, 0, st>>>(m, w.result, fo_bwd_idx, pt.cuts, pt.world, pt.cap_n, pt.bip_bwd_idx, pt.bip_bwd_ptr, bcnt);
        PP_LAUNCH_CHECK();
        rc = exclusive_scan<int32_t, int32_t>(bcnt, n_pad, pt.bip_fwd_ptr, true, nullptr, w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        const int64_t total = n_pad > m ? n_pad : m;
        k_db2_bip_fill<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, st>>>(total, w.result, w.xkeys_s, m, pt.lo, n, n_pad, (void*)pt.bip_self, pt.bip_fwd_idx);
        PP_LAUNCH_CHECK();
    }
    if (part) {
        k_db2_out_fill<<<egrid, kBlock, 0, st>>>(m, pt.lo, n_own, w.tp, w.tkeys_s, w.ot_t, w.ocr_t, fo_bwd_ptr, perm, w.src_t, hoff, rank_t);
        PP_LAUNCH_CHECK();
    }
    if (part) {          // (the halo numbering below reads the in-events; on one GPU the count pass gathers them itself)
        k_db2_gather_in<<<egrid, kBlock, 0, st>>>(m, w.hl, w.src_t, weight ? w.ow_t : nullptr, w.is_t, w.is_a, w.is_u, w.is_w);
        PP_LAUNCH_CHECK();
    }
    if (part) {
        // 2c. halo numbering (whose rows I gather from): one 8-bit sort of the in-event positions by owner
        uint32_t* hkeys = (uint32_t*)w.da_s;             // (fill-pass scratch, unused until then)
        uint32_t* hkeys_sorted = (uint32_t*)w.du_s;
        k_db2_halo_keys<<<egrid, kBlock, 0, st>>>(m, pt.lo, n_own, w.hkeys_s, w.hp, w.is_a, w.is_u, pt.cuts, pt.world, hkeys);
        PP_LAUNCH_CHECK();
        rc = sort_pairs<uint32_t>(hkeys, nullptr, hkeys_sorted, w.xorder, m, 0, bits_for((uint64_t)pt.world), w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        k_db2_halo_assign<<<egrid, kBlock, 0, st>>>(m, hkeys_sorted, w.xorder, pt.world, w.result, w.is_u);
        PP_LAUNCH_CHECK();
        k_db2_publish<<<1, kDb2MaxWorld + 1, 0, st>>>(m, hkeys_sorted, w.xkeys_s, pt.cuts, pt.world, n, w.result);
        PP_LAUNCH_CHECK();
    }
    // 3. middle-node pass, counting
    PP_HIP(hipMemsetAsync(w.indeg2, 0, (size_t)m * sizeof(int32_t), st));
    PP_HIP(hipMemsetAsync(w.outdeg2, 0, (size_t)m * sizeof(int32_t), st));
    Db2Mid a{};
    mid_common(a, w, fo_bwd_ptr, weight != nullptr);
    a.lo = pt.lo; a.n_own = n_own; a.part = part ? 1 : 0; a.perm = perm; a.hubs_handled = part ? 0 : 1;
    a.indeg2 = w.indeg2; a.outdeg2 = w.outdeg2; a.ho_deg = ho_deg; a.fo_deg = fo_deg;
    a.nu = w.nu; a.pc = w.pc; a.status = w.result + 1;
    if (!part) {
        a.hl = w.hl; a.src_t = w.src_t; a.ow_t = weight ? w.ow_t : nullptr;
        a.is_t_out = w.is_t; a.is_a_out = w.is_a; a.is_u_out = w.is_u; a.is_w_out = w.is_w;
    }
    rc = launch_mid_any<false>(time_dtype, delta_kind, weight != nullptr, ngrid, st, n_own, delta_i, delta_f, a);
    if (rc != PP_OK) return rc;
    if (hs.hubs > 0) {
        // 3'. hub nodes: their in-events, one wave per chunk of them, the chunks' shares combined in task order
        k_db2_hub_gather_in<<<egrid, kBlock, 0, st>>>(m, w.hkeys_s, w.hub_flag, w.hl, w.src_t, weight ? w.ow_t : nullptr, w.is_t, w.is_a, w.is_u, w.is_w);
        PP_LAUNCH_CHECK();
        Db2Hub h{};
        hub_common(h, w, hw, n);
        a.hl = nullptr;
        rc = launch_hub_any<false>(time_dtype, delta_kind, weight != nullptr, st, hs.tasks, delta_i, delta_f, a, h);
        if (rc != PP_OK) return rc;
        if (hs.out_events > 0) {
            rc = launch_hubx_any<false>(time_dtype, delta_kind, weight != nullptr, st, hs.tasks, delta_i, delta_f, a, h, w.ot_t, hw.rank_t);
            if (rc != PP_OK) return rc;
        }
        k_db2_hub_combine<<<(unsigned)hs.hubs, kCombineWaves * kWave, 0, st>>>(hs.hubs, a, h);
        PP_LAUNCH_CHECK();
    }
    // first-order in-row pointers, both order-2 row pointers and E2 = sum of the per-node pair counts: four scans, one launch triple
    const int32_t* s_in[4] = {w.nu, w.indeg2, w.outdeg2, w.pc};
    const int64_t s_n[4] = {n_own, m, m, n_own};
    int32_t* s_out[4] = {fo_fwd_ptr, ho_fwd_ptr, ho_bwd_ptr, nullptr};
    int64_t* s_tot[4] = {w.result + 4, w.result + 2, nullptr, w.result + 3};
    PP_REQUIRE(scan_multi_ws_bytes(s_n, 4) <= w.scratch_bytes, PP_ERR_WORKSPACE, "%s: scan scratch too small", who);
    rc = exclusive_scan_multi(s_in, s_n, s_out, s_tot, 4, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    if (host_result != nullptr) {
        // the sizes travel to the caller's pinned buffer NOW; the first kernel of the fill pass that needs no size (degree^-1/2 + row start of every
        // order-2 row, packed) is queued behind the copy and runs while the host wakes up and sizes the plans
        PP_HIP(hipMemcpyAsync(host_result, w.result, kDb2Result * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        { const int erc = record_stats_event(st); if (erc != PP_OK) return erc; }
        if (!part) {
            k_db2_pack_rows<<<egrid, kBlock, 0, st>>>(m, ho_deg, ho_bwd_ptr, w.row_pack);
            PP_LAUNCH_CHECK();
        }
    }
    return PP_OK;
}

static int db2_fill(const char* who, int time_dtype, int64_t m, int64_t n, int64_t lo, int64_t n_own, bool part, int delta_kind, int64_t delta_i,
                    double delta_f, const float* weight, const int32_t* fo_bwd_ptr, const int32_t* fo_bwd_idx, const float* fo_w, const int32_t* fo_fwd_ptr,
                    const int32_t* ho_fwd_ptr, const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t num_ho_edges,
                    int32_t* ho_fwd_idx, float* ho_fwd_val, int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx,
                    float* fo_fwd_val, int32_t* fo_dst_order, const int32_t* fo2_bwd_ptr, int32_t* fo2_bwd_idx, float* fo_bwd_val, float* fo_self,
                    float* ho_fwd_w, void* pair_scratch, void* ws, size_t ws_bytes, const Db2HubSizes& hs, void* hub_ws, size_t hub_ws_bytes,
                    bool rows_packed, hipStream_t st) {
    PP_REQUIRE(m >= 0 && n >= 0, PP_ERR_ARG, "%s: negative size", who);
    Db2Ws w = carve_db2(ws, m, n);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "%s: workspace too small", who);
    if (n == 0 || n_own == 0) return PP_OK;
    PP_REQUIRE(m > 0, PP_ERR_ARG, "%s: an empty stream has no order-2 model to fill (use pp_gcn_plan on the empty graph)", who);
    Db2HubWs hw = carve_db2_hub(hs.hubs > 0 ? hub_ws : nullptr, hs.out_events, hs.out_hubs, hs.parts);
    PP_REQUIRE(hs.hubs == 0 || (!part && hub_ws != nullptr && hub_ws_bytes >= hw.total_bytes), PP_ERR_WORKSPACE, "%s: hub workspace too small", who);
    const unsigned egrid = (unsigned)ceil_div(m, kBlock), ngrid = (unsigned)ceil_div(n_own, kWavesPerBlock * kDb2Nodes);
    PP_REQUIRE(num_ho_edges >= 0 && (num_ho_edges == 0 || pair_scratch != nullptr), PP_ERR_ARG, "%s: pair_scratch (8 bytes per order-2 edge) missing", who);
    if (!rows_packed) {                          // (one GPU: the count call queued it behind its size copy)
        k_db2_pack_rows<<<egrid, kBlock, 0, st>>>(m, ho_deg, ho_bwd_ptr, w.row_pack);
        PP_LAUNCH_CHECK();
    }
    Db2Mid a{};
    mid_common(a, w, fo_bwd_ptr, weight != nullptr);
    a.lo = lo; a.n_own = n_own; a.part = part ? 1 : 0; a.perm = part ? w.perm : nullptr; a.hubs_handled = part ? 0 : 1;
    a.ho_deg = const_cast<float*>(ho_deg); a.fo_deg = const_cast<float*>(fo_deg);
    a.row_pack = w.row_pack;
    if (!part) { a.oc_s = w.oc_s; a.fo_w = fo_w; a.fo_bwd_val = fo_bwd_val; }
    a.ho_fwd_ptr = ho_fwd_ptr; a.fo_fwd_ptr = fo_fwd_ptr;
    a.in_idx2 = ho_fwd_idx; a.in_val2 = ho_fwd_val; a.in_w2 = ho_fwd_w; a.out_pack = (uint2*)pair_scratch; a.self2 = ho_self;
    a.fwd_idx1 = fo_fwd_idx; a.fwd_val1 = fo_fwd_val; a.dst_order = fo_dst_order; a.self1 = fo_self;
    int rc = launch_mid_any<true>(time_dtype, delta_kind, weight != nullptr, ngrid, st, n_own, delta_i, delta_f, a);
    if (rc != PP_OK) return rc;
    if (hs.hubs > 0) {
        Db2Hub h{};
        hub_common(h, w, hw, n);
        h.succ = fo_bwd_idx;
        rc = launch_hub_any<true>(time_dtype, delta_kind, weight != nullptr, st, hs.tasks, delta_i, delta_f, a, h);
        if (rc != PP_OK) return rc;
        if (hs.out_events > 0) {
            rc = launch_hubx_any<true>(time_dtype, delta_kind, weight != nullptr, st, hs.tasks, delta_i, delta_f, a, h, w.ot_t, hw.rank_t);
            if (rc != PP_OK) return rc;
        }
    }
    if (num_ho_edges > 0) {
        k_db2_unzip<<<(unsigned)ceil_div(num_ho_edges, kBlock), kBlock, 0, st>>>(num_ho_edges, (const uint2*)pair_scratch, ho_bwd_idx, ho_bwd_val);
        PP_LAUNCH_CHECK();
    }
    if (part) k_db2_fo_part_bwd<<<egrid, kBlock, 0, st>>>(m, lo, n_own, w.tp, w.tkeys_s, w.oc_s, w.ocr_s, w.ow_s, w.fskip, fo2_bwd_ptr, fo_deg, fo2_bwd_idx, fo_bwd_val);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp

extern "C" {

int pp_debruijn2_lists(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, const float* weight, void* ws,
                       size_t ws_bytes, int64_t* host_stats, pp_stream_t stream) {
    const Db2Part whole{0, num_nodes, nullptr, 1, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    return db2_lists("pp_debruijn2_lists", edge_index, time, time_dtype, m, num_nodes, whole, weight, ws, ws_bytes, host_stats, (hipStream_t)stream);
}

int pp_debruijn2_wait(void) {
    if (!tls_stats_event) return PP_OK;
    // (the copy is a few hundred microseconds away at most: poll first — a blocking wait adds the wake-up of the thread to the time the GPU idles)
    for (int spin = 0; spin < 20000; ++spin) {
        const hipError_t e = hipEventQuery(tls_stats_event);
        if (e == hipSuccess) return PP_OK;
        if (e != hipErrorNotReady) PP_HIP(e);
    }
    PP_HIP(hipEventSynchronize(tls_stats_event));
    return PP_OK;
}

int pp_debruijn2_count(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                       int32_t* fo_bwd_ptr, int32_t* fo_bwd_idx, float* fo_w, int32_t* fo_fwd_ptr, int32_t* ho_fwd_ptr, int32_t* ho_bwd_ptr,
                       float* ho_deg, float* fo_deg, void* ws, size_t ws_bytes, int64_t hub_nodes, int64_t out_hubs, int64_t hub_out_events,
                       int64_t hub_tasks, int64_t hub_parts, void* hub_ws, size_t hub_ws_bytes, int64_t* host_result, pp_stream_t stream) {
    const Db2Part whole{0, num_nodes, nullptr, 1, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    const Db2HubSizes hs{hub_nodes, out_hubs, hub_out_events, hub_tasks, hub_parts};
    return db2_count("pp_debruijn2_count", time_dtype, m, num_nodes, whole, delta_kind, delta_i, delta_f, weight, fo_bwd_ptr, fo_bwd_idx, fo_w, fo_fwd_ptr,
                     ho_fwd_ptr, ho_bwd_ptr, ho_deg, fo_deg, ws, ws_bytes, hs, hub_ws, hub_ws_bytes, host_result, (hipStream_t)stream);
}

int pp_debruijn2_fill(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                      const int32_t* fo_bwd_ptr, const int32_t* fo_bwd_idx, const float* fo_w, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                      const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t num_ho_edges, int32_t* ho_fwd_idx, float* ho_fwd_val,
                      int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val, int32_t* fo_dst_order,
                      float* fo_bwd_val, float* fo_self, float* ho_fwd_w, void* pair_scratch, void* ws, size_t ws_bytes, int64_t hub_nodes,
                      int64_t out_hubs, int64_t hub_out_events, int64_t hub_tasks, int64_t hub_parts, void* hub_ws, size_t hub_ws_bytes,
                      int rows_packed, pp_stream_t stream) {
    const Db2HubSizes hs{hub_nodes, out_hubs, hub_out_events, hub_tasks, hub_parts};
    return db2_fill("pp_debruijn2_fill", time_dtype, m, num_nodes, 0, num_nodes, false, delta_kind, delta_i, delta_f, weight, fo_bwd_ptr, fo_bwd_idx, fo_w,
                    fo_fwd_ptr, ho_fwd_ptr, ho_bwd_ptr, ho_deg, fo_deg, num_ho_edges, ho_fwd_idx, ho_fwd_val, ho_bwd_idx, ho_bwd_val, ho_self, fo_fwd_idx,
                    fo_fwd_val, fo_dst_order, nullptr, nullptr, fo_bwd_val, fo_self, ho_fwd_w, pair_scratch, ws, ws_bytes, hs, hub_ws, hub_ws_bytes,
                    rows_packed != 0, (hipStream_t)stream);
}

int pp_debruijn2_fill_ready(int time_dtype, int64_t m, int64_t num_nodes, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                            const int32_t* fo_bwd_ptr, const int32_t* fo_bwd_idx, const float* fo_w, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                            const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t ho_edge_capacity, int32_t* ho_fwd_idx,
                            float* ho_fwd_val, int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val,
                            int32_t* fo_dst_order, float* fo_bwd_val, float* fo_self, float* ho_fwd_w, void* pair_scratch, void* ws, size_t ws_bytes,
                            int64_t hub_nodes, int64_t out_hubs, int64_t hub_out_events, int64_t hub_tasks, int64_t hub_parts, void* hub_ws,
                            size_t hub_ws_bytes, int rows_packed, const int64_t* host_result, int64_t* launched, pp_stream_t stream) {
    PP_REQUIRE(host_result != nullptr && launched != nullptr && ho_edge_capacity >= 0, PP_ERR_ARG, "pp_debruijn2_fill_ready: host_result, launched");
    *launched = 0;
    const int rc = pp_debruijn2_wait();
    if (rc != PP_OK) return rc;
    const int64_t status = host_result[1], a2 = host_result[2];
    if (status != 0 || a2 > ho_edge_capacity || a2 >= ((int64_t)1 << 31) - 64) return PP_OK;      // (the caller reads the header and decides)
    *launched = 1;
    return pp_debruijn2_fill(time_dtype, m, num_nodes, delta_kind, delta_i, delta_f, weight, fo_bwd_ptr, fo_bwd_idx, fo_w, fo_fwd_ptr, ho_fwd_ptr, ho_bwd_ptr,
                             ho_deg, fo_deg, a2, ho_fwd_idx, ho_fwd_val, ho_bwd_idx, ho_bwd_val, ho_self, fo_fwd_idx, fo_fwd_val, fo_dst_order, fo_bwd_val,
                             fo_self, ho_fwd_w, pair_scratch, ws, ws_bytes, hub_nodes, out_hubs, hub_out_events, hub_tasks, hub_parts, hub_ws, hub_ws_bytes,
                             rows_packed, stream);
}

int pp_debruijn2_part_count(const int64_t* edge_index, const void* time, int time_dtype, int64_t m, int64_t num_nodes, int64_t node_lo, int64_t n_own,
                            const int64_t* cuts, int world, int rank, int delta_kind, int64_t delta_i, double delta_f, const float* weight,
                            int32_t* fo_bwd_ptr, int32_t* fo_bwd_idx, float* fo_w, int32_t* fo_fwd_ptr, int32_t* ho_fwd_ptr, int32_t* ho_bwd_ptr,
                            float* ho_deg, float* fo_deg, int32_t* send_slot, int32_t* row_of, int32_t* fo_shard_bwd_ptr, int64_t pad_rows,
                            int32_t* bip_fwd_ptr, int32_t* bip_fwd_idx, int32_t* bip_bwd_ptr, int32_t* bip_bwd_idx, float* bip_self, void* ws,
                            size_t ws_bytes, pp_stream_t stream) {
    PP_REQUIRE(world >= 2 && cuts != nullptr && send_slot != nullptr && row_of != nullptr && fo_shard_bwd_ptr != nullptr && bip_fwd_ptr != nullptr &&
               bip_fwd_idx != nullptr && bip_bwd_ptr != nullptr && bip_bwd_idx != nullptr && bip_self != nullptr, PP_ERR_ARG,
               "pp_debruijn2_part_count: world >= 2 with cuts and every output buffer");
    const Db2Part pt{node_lo, n_own, cuts, world, rank, send_slot, row_of, fo_shard_bwd_ptr, pad_rows, bip_fwd_ptr, bip_fwd_idx, bip_bwd_ptr, bip_bwd_idx, bip_self};
    const int rc = db2_lists("pp_debruijn2_part_count", edge_index, time, time_dtype, m, num_nodes, pt, weight, ws, ws_bytes, nullptr, (hipStream_t)stream);
    if (rc != PP_OK) return rc;
    const Db2HubSizes none{0, 0, 0, 0, 0};
    return db2_count("pp_debruijn2_part_count", time_dtype, m, num_nodes, pt, delta_kind, delta_i, delta_f, weight, fo_bwd_ptr, fo_bwd_idx, fo_w,
                     fo_fwd_ptr, ho_fwd_ptr, ho_bwd_ptr, ho_deg, fo_deg, ws, ws_bytes, none, nullptr, 0, nullptr, (hipStream_t)stream);
}

int pp_debruijn2_part_fill(int time_dtype, int64_t m, int64_t num_nodes, int64_t node_lo, int64_t n_own, int delta_kind, int64_t delta_i, double delta_f,
                           const float* weight, const int32_t* fo_bwd_ptr, const int32_t* fo_fwd_ptr, const int32_t* ho_fwd_ptr,
                           const int32_t* ho_bwd_ptr, const float* ho_deg, const float* fo_deg, int64_t num_ho_edges, int32_t* ho_fwd_idx,
                           float* ho_fwd_val, int32_t* ho_bwd_idx, float* ho_bwd_val, float* ho_self, int32_t* fo_fwd_idx, float* fo_fwd_val,
                           float* fo_self, const int32_t* fo_shard_bwd_ptr, int32_t* fo_shard_bwd_idx, float* fo_shard_bwd_val, void* pair_scratch,
                           void* ws, size_t ws_bytes, pp_stream_t stream) {
    const Db2HubSizes none{0, 0, 0, 0, 0};
    return db2_fill("pp_debruijn2_part_fill", time_dtype, m, num_nodes, node_lo, n_own, true, delta_kind, delta_i, delta_f, weight, fo_bwd_ptr, nullptr, nullptr,
                    fo_fwd_ptr, ho_fwd_ptr, ho_bwd_ptr, ho_deg, fo_deg, num_ho_edges, ho_fwd_idx, ho_fwd_val, ho_bwd_idx, ho_bwd_val, ho_self, fo_fwd_idx,
                    fo_fwd_val, nullptr, fo_shard_bwd_ptr, fo_shard_bwd_idx, fo_shard_bwd_val, fo_self, nullptr, pair_scratch, ws, ws_bytes, none, nullptr, 0,
                    false, (hipStream_t)stream);
}

}  // extern "C"
