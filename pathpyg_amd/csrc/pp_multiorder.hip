// pathpyg_amd — the De Bruijn layers of ALL orders of a temporal stream, level by level, without instance graphs (round 6).
//
// Reference path replaced (paths relative to the pathpyG repository root):
//   MultiOrderModel.from_temporal_graph, max_order >= 3     src/pathpyG/core/multi_order_model.py:124-192
//     = lift_order_temporal                                 src/pathpyG/algorithms/temporal.py:17-54
//     + per order: iterate_lift_order                       src/pathpyG/core/multi_order_model.py:83-122
//         lift_order_edge_index(_weighted, aggr="src")      src/pathpyG/algorithms/lift_order.py:48-106
//         node-sequence extension                           src/pathpyG/core/multi_order_model.py:114
//         aggregate_edge_index (unique rows + coalesce)     src/pathpyG/algorithms/lift_order.py:109-152
//
// The reference (and the generic kernels of this library) materialise the order-k INSTANCE graph [2, E_k], its [E_k, k] node
// sequences, sort them (torch.unique(dim=0)) and sort the remapped edges again (coalesce) — per order.  None of that is needed:
//
//   * An order-(k+1) instance is a time-respecting path of k events.  All that the NEXT order ever asks of it is (i) its TYPE — the
//     node sequence it realises, i.e. a node of layer k+1 = an edge of layer k (De Bruijn property) —, (ii) the continuations of its
//     LAST event, and (iii) the weight of its first event (lift_order_edge_index_weighted with aggr="src" hands it down unchanged).
//   * The continuations of an event are a contiguous window of the time-ordered out-list of its head node (pp_temporal_count).  `tab`
//     holds, per list position, {head node of that event, ITS window (first, count), event id}: reading the window of an instance's
//     last event yields, per child instance, the new last node AND the child's own window — one random 16..128-byte access per parent.
//   * Instances are kept GROUPED BY TYPE, types in lexicographic order of their node sequences, instances of one type in lexicographic
//     order of their event sequences (= the reference's instance numbering restricted to the type).  The children of all instances of
//     type s, stably sorted by their new last node d, are then exactly the instances of the types s ++ d, in the right order: the
//     global sorts of the reference shrink to one tiny sort per type (a lane's registers; a wave's or a workgroup's LDS for the larger
//     ones), the weight of a merged edge is the count (or the left-to-right sum, PyG's coalesce order) of a run, and everything is written
//     sequentially.  Level 1 is the same idea on the events: pp_temporal_count has them grouped by source in time order, every node's list
//     is sorted by target in LDS — the only global sort of the whole model is pp_temporal_count's sort by source.
//   * Layer k+1's edge (s -> c): c is the type suffix(s) ++ d.  suffix(s) is the column u of s in layer k, the candidates are u's
//     out-edges in layer k (one contiguous id block, last nodes ascending): a bisection in a handful of entries.
//
// Per level: k_mo_children (+ k_mo_children_wave, k_mo_children_big) -> scan of the row lengths -> k_mo_types (+ k_mo_types_wave,
// k_mo_types_big) -> scan of the children counts.  No read-back between them; the caller reads {types, status, children of the next
// level} once per level.  Algorithmic bytes per level (what the generic kernels move — 16 E_k + 16 E_{k+1} for the lift,
// 20 E_{k+1} + 20 A_{k+1} for the aggregation) are reported by bench.py beside the time (`multi_order`); the bytes this path moves are
// 16 I_k + 48 I_{k+1} + 20 A_{k+1} (top layer: 16 I_k + 24 I_{k+1} + 8 A_{k+1}) — and one 128-byte line of `tab` per parent instance plus
// the candidate blocks, which is what bounds it.
#include <stdlib.h>

#include "pp_internal.h"

namespace pp {

constexpr int64_t kMoBadIndex = 1, kMoUnsorted = 2, kMoOverflow = 4;
constexpr int kMoSmall = 8;              // children of a one-instance type a single lane sorts in registers
#ifndef PP_MO_BIG
#define PP_MO_BIG 4096
#endif
constexpr int kMoBigMax = PP_MO_BIG;     // children of one type a workgroup sorts in LDS; beyond: status bit kMoOverflow (caller falls back)
constexpr int kMoLongRun = 128;          // level 1: events of one node pair summed by a wave instead of a lane
constexpr uint32_t kHeadBit = 0x80000000u;

// An instance is 16 bytes: {first, count | head bit} of its last event's window in `tab`, last node, weight of its first event.
// Child records as the children pass leaves them for the types pass.  kFmt 0: the full 16-byte instance (a level below the top: the next
// level's parents).  The children of the TOP layer are nobody's parents — only their last node and the head flag are read again
// (kFmt 1: 4 bytes, last node | head bit; node ids stay below 2^31), with event weights also the weight (kFmt 2: 8 bytes).
template <int kFmt>
__device__ __forceinline__ void mo_store_child(void* __restrict__ out, int64_t i, uint32_t cf, uint32_t cc, bool head, uint32_t d, uint32_t w_bits) {
    if constexpr (kFmt == 0) ((uint4*)out)[i] = make_uint4(cf, cc | (head ? kHeadBit : 0u), d, w_bits);
    else if constexpr (kFmt == 1) ((uint32_t*)out)[i] = d | (head ? kHeadBit : 0u);
    else ((uint2*)out)[i] = make_uint2(d | (head ? kHeadBit : 0u), w_bits);
}
template <int kFmt>
__device__ __forceinline__ uint4 mo_load_child(const void* __restrict__ in, int64_t i) {
    if constexpr (kFmt == 0) return ((const uint4*)in)[i];
    else if constexpr (kFmt == 1) { const uint32_t x = ((const uint32_t*)in)[i]; return make_uint4(0u, x & kHeadBit, x & ~kHeadBit, 0x3f800000u); }
    else { const uint2 x = ((const uint2*)in)[i]; return make_uint4(0u, x.x & kHeadBit, x.x & ~kHeadBit, x.y); }
}
constexpr int mo_fmt(bool weighted, bool last) { return last ? (weighted ? 2 : 1) : 0; }

// ------------------------------------------------------------------ level 1: the events grouped by (source, target), time order inside
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_mo_key_pair(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t m, int64_t n, int bits,
                                                       KeyT* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    int64_t a = src[i], b = dst[i];
    if (a < 0 || a >= n) a = 0;            // (pp_temporal_count has set the status bit: the caller raises; nothing here may run off an array)
    if (b < 0 || b >= n) b = 0;
    keys[i] = ((KeyT)a << bits) | (KeyT)b;
}

// everything the later gathers want to know about an event, in one 16-byte record: head node, continuation window, weight
__global__ __launch_bounds__(kBlock) void k_mo_events(const int64_t* __restrict__ dst, const uint32_t* __restrict__ first_pos,
                                                     const int32_t* __restrict__ count, const float* __restrict__ weight, int64_t m, int64_t n,
                                                     uint4* __restrict__ ev) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= m) return;
    int64_t b = dst[e];
    if (b < 0 || b >= n) b = 0;
    ev[e] = make_uint4((uint32_t)b, first_pos[e], (uint32_t)count[e], __float_as_uint(weight ? weight[e] : 1.0f));
}

// the level-1 instance of position j of the (source, target, time) order + the flag "first event of its node pair"
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_mo_inst1(const KeyT* __restrict__ sorted, const uint32_t* __restrict__ perm, const uint4* __restrict__ ev,
                                                    int64_t m, uint4* __restrict__ inst, int32_t* __restrict__ head) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    const bool h = j == 0 || sorted[j - 1] != sorted[j];
    const uint4 r = ev[perm[j]];
    inst[j] = make_uint4(r.y, r.z | (h ? kHeadBit : 0u), r.x, r.w);
    head[j] = h ? 1 : 0;
}

// the types of level 1 = layer 1's edges: instance range, last node (= column), row pointers over the first-order nodes
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_mo_types1(const KeyT* __restrict__ sorted, int bits, const int32_t* __restrict__ head_before, int64_t m,
                                                     int64_t n, int32_t* __restrict__ tptr, int32_t* __restrict__ tlast, int32_t* __restrict__ rowptr) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    const int32_t t = head_before[j];
    const bool h = head_before[j + 1] != t;
    const KeyT me = sorted[j];
    const int64_t a = (int64_t)(me >> bits);
    if (h) {
        tptr[t] = (int32_t)j;
        tlast[t] = (int32_t)(me & (((KeyT)1 << bits) - 1));
        const int64_t a_prev = j == 0 ? -1 : (int64_t)(sorted[j - 1] >> bits);
        for (int64_t v = a_prev + 1; v <= a; ++v) rowptr[v] = t;
    }
    if (j == m - 1) {
        const int32_t total = head_before[m];
        tptr[total] = (int32_t)m;
        for (int64_t v = a + 1; v <= n; ++v) rowptr[v] = total;
    }
}

// ---- the same order WITHOUT a second global sort (the default): pp_temporal_count has the events grouped by source in time order (`ids`, `rowptr`);
// the (source, target, time) order is that sequence with every node's list stably sorted by target — lists of ~20 entries: ONE WAVE PER 64
// CONSECUTIVE NODES sorts (node, target, slot) keys in LDS (the bitonic network of k_mo_children_wave), a workgroup a list of up to kMoBigMax
// events; a longer list sets kMoLongList and the caller takes the radix sort above.  The kernels write the level-1 instances and the head flags
// in place: position p of node v's list range holds the event of rank p - rowptr[v] in its (target, time) order.
constexpr int64_t kMoLongList = 16;
constexpr int kMoListCap = 256;
struct MoListLds {
    uint64_t keys[kMoListCap];
    uint32_t cf[kMoListCap], cc[kMoListCap], w[kMoListCap];
    int32_t p0s[kWave], pre[kWave];
};

__global__ __launch_bounds__(kBlock) void k_mo_lists_wave(int64_t n_nodes, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                                                         const uint4* __restrict__ ev, uint4* __restrict__ inst, int32_t* __restrict__ head,
                                                         uint4* __restrict__ tab, int32_t* __restrict__ big_list, int32_t* __restrict__ counters,
                                                         int64_t* __restrict__ status) {
    __shared__ MoListLds lds[kWavesPerBlock];
    MoListLds& L = lds[wave_id()];
    const int lane = lane_id();
    for (int64_t first = ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * kWave; first < n_nodes; first += (int64_t)gridDim.x * kBlock) {
        const int64_t v = first + lane;
        int32_t p0 = 0, nv = 0;
        if (v < n_nodes) { p0 = (int32_t)rowptr[v]; nv = (int32_t)rowptr[v + 1] - p0; }
        const bool big = nv > kMoListCap;
        if (nv > kMoBigMax) atomicOr((unsigned long long*)status, (unsigned long long)kMoLongList);
        {   // lists for the workgroup kernel (one atomic per wave that has one)
            const bool take = big && nv <= kMoBigMax;
            const uint64_t mask = __ballot(take);
            if (mask) {
                const int leader = __ffsll((long long)mask) - 1;
                int32_t at = 0;
                if (lane == leader) at = atomicAdd(counters + 1, (int32_t)__popcll(mask));
                at = __shfl(at, leader, kWave);
                if (take) big_list[at + (int32_t)__popcll(mask & lanemask_lt())] = (int32_t)v;
            }
        }
        const int32_t ne = big ? 0 : nv;
        const int32_t incl = wave_inclusive_sum<int32_t>(ne);
        if (__shfl(incl, kWave - 1, kWave) == 0) continue;
        L.p0s[lane] = p0;
        L.pre[lane] = incl - ne;
        int l0 = 0;
        while (l0 < kWave) {
            const int32_t base = l0 ? __shfl(incl, l0 - 1, kWave) : 0;
            const uint64_t from = l0 ? ~((1ull << l0) - 1ull) : ~0ull;
            const uint64_t over = __ballot(incl - base > kMoListCap) & from;
            const int l1 = over ? __ffsll((long long)over) - 1 : kWave;
            const int32_t nb = __shfl(incl, l1 - 1, kWave) - base;
            if (nb > 0) {
                __builtin_amdgcn_wave_barrier();
                int np2 = 2;
                while (np2 < nb) np2 <<= 1;
                for (int slot = lane; slot < np2; slot += kWave) {
                    uint64_t key = ~0ull;
                    if (slot < nb) {
                        int lo = l0, hi = l1;              // last node of the batch whose entries start at or before the slot
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (L.pre[mid] - base <= slot) lo = mid; else hi = mid;
                        }
                        const int32_t at = L.p0s[lo] + (slot - (L.pre[lo] - base));
                        const uint32_t e = ids[at];
                        const uint4 r = ev[e];
                        tab[at] = make_uint4(r.x, r.y, r.z, e);          // (the window table's entry of this list position: k_mo_tab's gather, already here)
                        L.cf[slot] = r.y; L.cc[slot] = r.z; L.w[slot] = r.w;
                        key = ((uint64_t)lo << 40) | ((uint64_t)r.x << 8) | (uint64_t)slot;
                    }
                    L.keys[slot] = key;
                }
                __builtin_amdgcn_wave_barrier();
                for (int k = 2; k <= np2; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int q = lane; q < (np2 >> 1); q += kWave) {
                            const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), o = i | j;
                            const uint64_t a = L.keys[i], b = L.keys[o];
                            if ((a > b) == ((i & k) == 0)) { L.keys[i] = b; L.keys[o] = a; }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                for (int pos = lane; pos < nb; pos += kWave) {
                    const uint64_t key = L.keys[pos];
                    const uint64_t prev = pos ? L.keys[pos - 1] : ~0ull;
                    const bool h = (key >> 8) != (prev >> 8);                  // another node or another target
                    const int nl = (int)(key >> 40), slot = (int)(key & 0xffu);
                    const int32_t at = L.p0s[nl] + (pos - (L.pre[nl] - base));
                    inst[at] = make_uint4(L.cf[slot], L.cc[slot] | (h ? kHeadBit : 0u), (uint32_t)(key >> 8), L.w[slot]);
                    head[at] = h ? 1 : 0;
                }
                __builtin_amdgcn_wave_barrier();
            }
            l0 = l1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(kBlock) void k_mo_lists_big(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint4* __restrict__ ev,
                                                        uint4* __restrict__ inst, int32_t* __restrict__ head, uint4* __restrict__ tab,
                                                        const int32_t* __restrict__ big_list, const int32_t* __restrict__ counters) {
    __shared__ uint64_t s_key[kMoBigMax];
    const int n_big = counters[1];
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < n_big; g += gridDim.x) {
        const int32_t v = big_list[g];
        const int32_t p0 = (int32_t)rowptr[v], n = (int32_t)rowptr[v + 1] - p0;
        int np2 = 2;
        while (np2 < n) np2 <<= 1;
        for (int c = tid; c < np2; c += kBlock) {
            uint64_t key = ~0ull;
            if (c < n) {
                const uint32_t e = ids[p0 + c];
                const uint4 r = ev[e];
                tab[p0 + c] = make_uint4(r.x, r.y, r.z, e);
                key = ((uint64_t)r.x << 32) | (uint64_t)c;
            }
            s_key[c] = key;
        }
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += kBlock) {
                    const int o = i ^ j;
                    if (o > i) {
                        const uint64_t a = s_key[i], b = s_key[o];
                        if ((a > b) == ((i & k) == 0)) { s_key[i] = b; s_key[o] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int r = tid; r < n; r += kBlock) {
            const uint64_t key = s_key[r];
            const uint32_t dd = (uint32_t)(key >> 32);
            const bool h = r == 0 || (uint32_t)(s_key[r - 1] >> 32) != dd;
            const uint4 e = ev[ids[p0 + (int32_t)(uint32_t)key]];
            inst[p0 + r] = make_uint4(e.y, e.z | (h ? kHeadBit : 0u), dd, e.w);
            head[p0 + r] = h ? 1 : 0;
        }
        __syncthreads();
    }
}

// types of level 1 from the instances in place: instance range, last node; row pointers over the first-order nodes = heads before a node's list
__global__ __launch_bounds__(kBlock) void k_mo_types1_lists(const uint4* __restrict__ inst, const int32_t* __restrict__ head_before, const uint32_t* __restrict__ list_ptr,
                                                           int64_t m, int64_t n, int32_t* __restrict__ tptr, int32_t* __restrict__ tlast,
                                                           int32_t* __restrict__ rowptr) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j <= n) rowptr[j] = head_before[list_ptr[j]];
    if (j >= m) return;
    const int32_t t = head_before[j];
    if (head_before[j + 1] != t) { tptr[t] = (int32_t)j; tlast[t] = (int32_t)inst[j].z; }
    if (j == m - 1) tptr[head_before[m]] = (int32_t)m;
}

// merged weight (run length, or the left-to-right sum: PyG's coalesce order) and number of children of every level-1 type
template <bool kWeighted>
__global__ __launch_bounds__(kBlock) void k_mo_sums1(const int32_t* __restrict__ tptr, const int32_t* __restrict__ n_types, const uint4* __restrict__ inst,
                                                    float* __restrict__ w, int32_t* __restrict__ csum, int32_t* __restrict__ long_list,
                                                    int32_t* __restrict__ long_count, int64_t* __restrict__ status) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= *n_types) return;
    const int32_t x0 = tptr[t], x1 = tptr[t + 1];
    if (x1 - x0 > kMoLongRun) { long_list[atomicAdd(long_count, 1)] = (int32_t)t; return; }
    float acc = 0.f;
    int64_t c = 0;
    for (int32_t x = x0; x < x1; ++x) {
        const uint4 it = inst[x];
        acc += __uint_as_float(it.w);
        c += (int64_t)(it.y & ~kHeadBit);
    }
    w[t] = kWeighted ? acc : (float)(x1 - x0);
    if (c > 0x7fffffff) { atomicOr((unsigned long long*)status, (unsigned long long)kMoOverflow); c = 0x7fffffff; }     // (2^31 instances at level 2: the caller falls back)
    csum[t] = (int32_t)c;
}

// node pairs with very many events: one wave per pair, every lane a contiguous chunk left to right, chunks combined in lane order
template <bool kWeighted>
__global__ __launch_bounds__(kBlock) void k_mo_sums1_long(const int32_t* __restrict__ tptr, const uint4* __restrict__ inst, const int32_t* __restrict__ long_list,
                                                         const int32_t* __restrict__ long_count, float* __restrict__ w, int32_t* __restrict__ csum,
                                                         int64_t* __restrict__ status) {
    const int n_long = *long_count;
    for (int g = blockIdx.x * kWavesPerBlock + wave_id(); g < n_long; g += gridDim.x * kWavesPerBlock) {
        const int32_t t = long_list[g];
        const int32_t x0 = tptr[t], x1 = tptr[t + 1];
        const int32_t chunk = (x1 - x0 + kWave - 1) / kWave;
        const int32_t b = x0 + lane_id() * chunk, e = b + chunk < x1 ? b + chunk : x1;
        float acc = 0.f;
        long long c = 0;
        for (int32_t x = b; x < e; ++x) {
            const uint4 it = inst[x];
            acc += __uint_as_float(it.w);
            c += (long long)(it.y & ~kHeadBit);
        }
        float total = 0.f;
        for (int l = 0; l < kWave; ++l) total += __shfl(acc, l, kWave);        // fixed order
        c = wave_sum<long long>(c);
        if (lane_id() == 0) {
            w[t] = kWeighted ? total : (float)(x1 - x0);
            if (c > 0x7fffffff) { atomicOr((unsigned long long*)status, (unsigned long long)kMoOverflow); c = 0x7fffffff; }
            csum[t] = (int32_t)c;
        }
    }
}

// tab[p]: the event at position p of the per-node out-lists — its head node, its own continuation window, its id
__global__ __launch_bounds__(kBlock) void k_mo_tab(const uint32_t* __restrict__ ids, const uint4* __restrict__ ev, int64_t n_list, uint4* __restrict__ tab) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_list) return;
    const uint32_t l = ids[p];
    const uint4 r = ev[l];
    tab[p] = make_uint4(r.x, r.y, r.z, l);
}

// ------------------------------------------------------------------ one level: children of every type, sorted by their new last node
// stable odd-even transposition sort of 8 register slots by d (strict compare: equal last nodes keep their order)
#define PP_MO_CE(i, j)                                                                                  \
    do {                                                                                                \
        const bool sw = d[i] > d[j];                                                                    \
        const uint32_t td = sw ? d[j] : d[i], tf = sw ? cf[j] : cf[i], tc = sw ? cc[j] : cc[i], tw = sw ? w[j] : w[i]; \
        d[j] = sw ? d[i] : d[j]; cf[j] = sw ? cf[i] : cf[j]; cc[j] = sw ? cc[i] : cc[j]; w[j] = sw ? w[i] : w[j];      \
        d[i] = td; cf[i] = tf; cc[i] = tc; w[i] = tw;                                                   \
    } while (0)

// Which kernel takes a type: a LANE (k_mo_children) when it has one instance and at most kMoSmall children — the rule on sparse streams;
// a WAVE together with 63 other such types (k_mo_children_wave) up to kMoWaveCap children and kMoWaveParents instances; a WORKGROUP beyond.
constexpr int kMoWaveCap = 256;
constexpr int kMoWaveParents = 2048;
__device__ __forceinline__ bool mo_lane_type(int parents, int children) { return parents == 1 && children <= kMoSmall; }
__device__ __forceinline__ bool mo_big_type(int parents, int children) { return children > kMoWaveCap || parents > kMoWaveParents; }

// appends the flagged lanes' types to a list, one atomic per wave, lane order kept inside the wave's piece
__device__ __forceinline__ void mo_append(bool flag, int32_t value, int32_t* __restrict__ list, int32_t* __restrict__ count) {
    const uint64_t mask = __ballot(flag);
    if (mask == 0ull) return;
    const int leader = __ffsll((long long)mask) - 1;
    int32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(count, (int32_t)__popcll(mask));
    base = __shfl(base, leader, kWave);
    if (flag) list[base + (int32_t)__popcll(mask & lanemask_lt())] = value;
}

template <int kFmt>
__global__ __launch_bounds__(kBlock) void k_mo_children(int64_t n_types, const int32_t* __restrict__ tptr, const int32_t* __restrict__ ibase,
                                                       const uint4* __restrict__ inst, const uint4* __restrict__ tab, void* __restrict__ out,
                                                       int32_t* __restrict__ deg, uint8_t* __restrict__ cls, int32_t* __restrict__ big_list,
                                                       int32_t* __restrict__ counters) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = s < n_types;
    int32_t x0 = 0, parents = 0, c0 = 0, n = 0;
    if (valid) { x0 = tptr[s]; parents = tptr[s + 1] - x0; c0 = ibase[s]; n = ibase[s + 1] - c0; }
    const bool mine = valid && n > 0 && mo_lane_type(parents, n);
    const bool big = valid && n > 0 && mo_big_type(parents, n);
    if (valid) cls[s] = big ? 2 : ((n > 0 && !mine) ? 1 : 0);      // 1: left to the wave kernel, 2: to the workgroup kernel
    mo_append(big, (int32_t)s, big_list, counters);                 // (rare: one atomic per wave that has one)
    if (valid && n == 0) deg[s] = 0;
    if (!mine) return;
    uint32_t d[kMoSmall], cf[kMoSmall], cc[kMoSmall], w[kMoSmall];
    // one instance: its window's entries requested back to back
    const uint4 pi = inst[x0];
    const uint4* src = tab + pi.x;
#pragma unroll
    for (int q = 0; q < kMoSmall; ++q) {
        d[q] = 0xFFFFFFFFu; cf[q] = 0; cc[q] = 0; w[q] = pi.w;
        if (q < n) { const uint4 e = src[q]; d[q] = e.x; cf[q] = e.y; cc[q] = e.z; }
    }
    if (n > 1) {
#pragma unroll
        for (int round = 0; round < kMoSmall; ++round) {
            if ((round & 1) == 0) { PP_MO_CE(0, 1); PP_MO_CE(2, 3); PP_MO_CE(4, 5); PP_MO_CE(6, 7); }
            else { PP_MO_CE(1, 2); PP_MO_CE(3, 4); PP_MO_CE(5, 6); }
        }
    }
    int heads = 0;
#pragma unroll
    for (int q = 0; q < kMoSmall; ++q) {
        if (q < n) {
            const bool h = q == 0 || d[q] != d[q > 0 ? q - 1 : 0];
            heads += h ? 1 : 0;
            mo_store_child<kFmt>(out, c0 + q, cf[q], cc[q], h, d[q], w[q]);
        }
    }
    deg[s] = heads;
}

// The types in between (class 1), ONE WAVE PER 64 CONSECUTIVE TYPES: lanes are CHILD SLOTS, not types — a type with two instances and five children and a
// type with 30 instances and 200 children cost the wave the same per child.  The wave takes its types in batches of at most kMoWaveCap
// children: (1) the instances of the batch's types, type after type, in rounds of 64: wave prefix sum of their children counts; those that
// have children are compacted into LDS (window first, weight, type, first child slot) and mark that slot; a running maximum over the slots
// maps every slot to its instance; (2) every lane fetches the tab
// entries of its slots (the entries of one window are consecutive: coalesced) and leaves the key (type, new last node, slot) in LDS;
// (3) bitonic sort of the keys (unique, so it is the stable sort by last node inside every type); (4) the sorted children are written as
// consecutive 16-byte records, heads counted per type.
constexpr int kMoWaveTypes = 64;
struct MoWaveLds {
    uint64_t keys[kMoWaveCap];
    uint32_t nz_cf[kMoWaveCap], nz_w[kMoWaveCap], ccf[kMoWaveCap], ccc[kMoWaveCap];
    uint16_t marks[kMoWaveCap], nz_off[kMoWaveCap];
    uint8_t nz_type[kMoWaveCap];
    int32_t adj[kMoWaveTypes], deg[kMoWaveTypes], x0s[kMoWaveTypes], pin[kMoWaveTypes];
};

template <int kFmt>
__global__ __launch_bounds__(kBlock) void k_mo_children_wave(const int32_t* __restrict__ tptr, const int32_t* __restrict__ ibase,
                                                            const uint4* __restrict__ inst, const uint4* __restrict__ tab, void* __restrict__ out,
                                                            int32_t* __restrict__ deg, const uint8_t* __restrict__ cls, int64_t n_types) {
    __shared__ MoWaveLds lds[kWavesPerBlock];
    MoWaveLds& L = lds[wave_id()];
    const int lane = lane_id();
    // (a bounded grid, every wave strides over the 64-type pieces: on sparse streams almost every piece has nothing for this kernel, and
    //  a quarter of a million workgroups that read 256 bytes and leave cost 0.24 ms at 6.8 * 10^7 types)
    for (int64_t first = ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * kMoWaveTypes; first < n_types;
         first += (int64_t)gridDim.x * kWavesPerBlock * kMoWaveTypes) {
        const int64_t s = first + lane;
        const bool valid = s < n_types && cls[s] == 1;
        if (__ballot(valid) == 0ull) continue;          // (all of the piece's types were a lane's or a workgroup's: the rule on sparse streams)
        int32_t x0 = 0, x1 = 0, c0 = 0, n = 0;
        if (valid) { x0 = tptr[s]; x1 = tptr[s + 1]; c0 = ibase[s]; n = ibase[s + 1] - c0; }
        L.deg[lane] = 0;
        const int32_t incl = wave_inclusive_sum<int32_t>(n);
        const int32_t pin_incl = wave_inclusive_sum<int32_t>(x1 - x0);
        L.x0s[lane] = x0;
        L.pin[lane] = pin_incl - (x1 - x0);
        int l0 = 0;
        while (l0 < kWave) {
            const int32_t base = l0 ? __shfl(incl, l0 - 1, kWave) : 0;
            const uint64_t from = l0 ? ~((1ull << l0) - 1ull) : ~0ull;
            const uint64_t over = __ballot(incl - base > kMoWaveCap) & from;        // first type that no longer fits (every type alone fits)
            const int l1 = over ? __ffsll((long long)over) - 1 : kWave;
            const int32_t nb = __shfl(incl, l1 - 1, kWave) - base;
            if (nb > 0) {
                const bool in_batch = lane >= l0 && lane < l1;
                if (in_batch) L.adj[lane] = c0 - (incl - n - base);
                for (int i = lane; i < kMoWaveCap; i += kWave) L.marks[i] = 0;
                __builtin_amdgcn_wave_barrier();
                // the instances of the batch's types as ONE sequence (type after type), 64 per round
                const int32_t fa = __shfl(pin_incl - (x1 - x0), l0, kWave), fb = __shfl(pin_incl, l1 - 1, kWave);
                int32_t run_off = 0, run_k = 0;
                for (int32_t f0 = fa; f0 < fb; f0 += kWave) {
                    const int32_t f = f0 + lane;
                    const bool live = f < fb;
                    int lo = l0, hi = l1;                      // last type of the batch whose instances start at or before f
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (L.pin[mid] <= f) lo = mid; else hi = mid;
                    }
                    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
                    if (live) rec = inst[L.x0s[lo] + (f - L.pin[lo])];
                    const int32_t c = (int32_t)(rec.y & ~kHeadBit);
                    const int32_t ci = wave_inclusive_sum<int32_t>(c);
                    const uint64_t some = __ballot(c > 0);
                    if (c > 0) {
                        const int k = run_k + (int)__popcll(some & lanemask_lt());
                        const int32_t off = run_off + ci - c;
                        L.nz_cf[k] = rec.x; L.nz_off[k] = (uint16_t)off; L.nz_w[k] = rec.w; L.nz_type[k] = (uint8_t)lo;
                        L.marks[off] = (uint16_t)(k + 1);
                    }
                    run_off += __shfl(ci, kWave - 1, kWave);
                    run_k += (int)__popcll(some);
                }
                __builtin_amdgcn_wave_barrier();
                {   // slot -> instance: inclusive running maximum of the marks... of marks ordered by slot: an instance's mark is the largest at
                    // or before its slots only if marks grow with the slot — they need not (k is handed out in arrival order), so the value
                    // spread is the mark's SLOT, and the instance is read through it
                    uint32_t v[4], m = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = L.marks[4 * lane + q] ? (uint32_t)(4 * lane + q + 1) : 0u; m = v[q] > m ? v[q] : m; v[q] = m; }
                    uint32_t sc_m = m;
#pragma unroll
                    for (int dlt = 1; dlt < kWave; dlt <<= 1) {
                        const uint32_t o = __shfl_up(sc_m, dlt, kWave);
                        if (lane >= dlt) sc_m = o > sc_m ? o : sc_m;
                    }
                    uint32_t before = __shfl_up(sc_m, 1, kWave);
                    if (lane == 0) before = 0;
                    uint32_t owner[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) owner[q] = v[q] > before ? v[q] : before;     // 1 + first slot of the instance that owns this slot
                    __builtin_amdgcn_wave_barrier();
                    uint32_t kk[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) kk[q] = owner[q] ? L.marks[owner[q] - 1] : 0u;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 4; ++q) L.marks[4 * lane + q] = (uint16_t)kk[q];
                }
                __builtin_amdgcn_wave_barrier();
                int np2 = 2;
                while (np2 < nb) np2 <<= 1;
                for (int slot = lane; slot < np2; slot += kWave) {
                    uint64_t key = ~0ull;
                    if (slot < nb) {
                        const int k = (int)L.marks[slot] - 1;
                        const uint4 e = tab[L.nz_cf[k] + (uint32_t)(slot - (int)L.nz_off[k])];
                        L.ccf[slot] = e.y; L.ccc[slot] = e.z;
                        key = ((uint64_t)L.nz_type[k] << 40) | ((uint64_t)e.x << 8) | (uint64_t)slot;
                    }
                    L.keys[slot] = key;
                }
                __builtin_amdgcn_wave_barrier();
                for (int k = 2; k <= np2; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int p = lane; p < (np2 >> 1); p += kWave) {
                            const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), o = i | j;
                            const uint64_t a = L.keys[i], b = L.keys[o];
                            if ((a > b) == ((i & k) == 0)) { L.keys[i] = b; L.keys[o] = a; }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                for (int pos = lane; pos < nb; pos += kWave) {
                    const uint64_t key = L.keys[pos];
                    const uint64_t prev = pos ? L.keys[pos - 1] : ~0ull;
                    const bool h = (key >> 8) != (prev >> 8);
                    const int t_l = (int)(key >> 40), slot = (int)(key & 0xffu);
                    const int k = (int)L.marks[slot] - 1;
                    mo_store_child<kFmt>(out, L.adj[t_l] + pos, L.ccf[slot], L.ccc[slot], h, (uint32_t)(key >> 8), L.nz_w[k]);
                    if (h) atomicAdd(&L.deg[t_l], 1);
                }
                __builtin_amdgcn_wave_barrier();
            }
            l0 = l1;
        }
        if (valid) deg[s] = L.deg[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

// types with more children (or instances) than a lane takes: one workgroup each, (last node, slot) keys sorted in LDS
template <int kFmt>
__global__ __launch_bounds__(kBlock) void k_mo_children_big(const int32_t* __restrict__ tptr, const int32_t* __restrict__ ibase, const uint4* __restrict__ inst,
                                                           const uint4* __restrict__ tab, void* __restrict__ out, int32_t* __restrict__ deg,
                                                           const int32_t* __restrict__ big_list, const int32_t* __restrict__ big_count,
                                                           int64_t* __restrict__ status) {
    __shared__ uint32_t s_src[kMoBigMax];
    __shared__ uint32_t s_w[kMoBigMax];
    __shared__ uint64_t s_key[kMoBigMax];
    __shared__ uint32_t s_scratch[kWavesPerBlock + 1];
    const int n_big = *big_count;
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < n_big; g += gridDim.x) {
        const int32_t s = big_list[g];
        const int32_t x0 = tptr[s], x1 = tptr[s + 1];
        const int32_t c0 = ibase[s], n = ibase[s + 1] - c0;
        if (n > kMoBigMax) {                 // (uniform)
            if (tid == 0) { atomicOr((unsigned long long*)status, (unsigned long long)kMoOverflow); deg[s] = 0; }
            continue;
        }
        uint32_t running = 0;
        for (int32_t xb = x0; xb < x1; xb += kBlock) {
            const int32_t x = xb + tid;
            uint4 pi = make_uint4(0u, 0u, 0u, 0u);
            if (x < x1) pi = inst[x];
            const uint32_t c = pi.y & ~kHeadBit;
            uint32_t total;
            const uint32_t at = running + block_exclusive_sum<uint32_t>(c, s_scratch, &total);
            for (uint32_t j = 0; j < c; ++j) { s_src[at + j] = pi.x + j; s_w[at + j] = pi.w; }
            running += total;
        }
        __syncthreads();
        int np2 = 2;
        while (np2 < n) np2 <<= 1;
        for (int c = tid; c < np2; c += kBlock) s_key[c] = c < n ? (((uint64_t)tab[s_src[c]].x << 32) | (uint64_t)c) : ~0ull;
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += kBlock) {
                    const int o = i ^ j;
                    if (o > i) {
                        const uint64_t a = s_key[i], b = s_key[o];
                        if ((a > b) == ((i & k) == 0)) { s_key[i] = b; s_key[o] = a; }
                    }
                }
                __syncthreads();
            }
        }
        uint32_t heads = 0;
        for (int r = tid; r < n; r += kBlock) {
            const uint64_t key = s_key[r];
            const uint32_t c = (uint32_t)key, dd = (uint32_t)(key >> 32);
            const bool h = r == 0 || (uint32_t)(s_key[r - 1] >> 32) != dd;
            const uint4 e = tab[s_src[c]];
            mo_store_child<kFmt>(out, c0 + r, e.y, e.z, h, dd, s_w[c]);
            heads += h ? 1u : 0u;
        }
        uint32_t total;
        block_exclusive_sum<uint32_t>(heads, s_scratch, &total);
        if (tid == 0) deg[s] = (int32_t)total;
        __syncthreads();
    }
}

constexpr int kMoPre = 16;
// position of `key` among the ascending last nodes cand[0 .. bc) of the candidate block (it is there: the suffix of a path is a path)
__device__ __forceinline__ int32_t mo_find(const int32_t* __restrict__ cand, int32_t lo, int32_t bc, int32_t key) {
    int32_t hi = bc;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (cand[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the new types of every parent type: instance range, last node, column (= id of the suffix type), merged weight, number of children.
// One lane per parent type of class 0 (one instance, at most kMoSmall children: the types k_mo_children took); the others belong to
// k_mo_types_wave / k_mo_types_big.  kLast: the top layer — only columns and weights are wanted
template <bool kWeighted, bool kLast>
__global__ __launch_bounds__(kBlock) void k_mo_types(int64_t n_types, const int32_t* __restrict__ tptr, const int32_t* __restrict__ ibase,
                                                    const int32_t* __restrict__ col, const int32_t* __restrict__ cand_ptr,
                                                    const int32_t* __restrict__ cand_last, const void* __restrict__ child,
                                                    const int32_t* __restrict__ row_ptr, int32_t* __restrict__ tptr_out,
                                                    int32_t* __restrict__ tlast_out, int32_t* __restrict__ col_out, float* __restrict__ w_out,
                                                    int32_t* __restrict__ csum_out, int all_wave) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_types) return;
    const int32_t c0 = ibase[s], n = ibase[s + 1] - c0;
    if (!kLast && s == n_types - 1) tptr_out[row_ptr[n_types]] = ibase[n_types];
    if (all_wave || n == 0 || !mo_lane_type(tptr[s + 1] - tptr[s], n)) return;
    const int32_t u = col[s];
    const int32_t bf = cand_ptr[u], bc = cand_ptr[u + 1] - bf;
    const int32_t* cand = cand_last + bf;
    uint4 it[kMoSmall];
#pragma unroll
    for (int q = 0; q < kMoSmall; ++q) it[q] = q < n ? mo_load_child<mo_fmt(kWeighted, kLast)>(child, c0 + q) : make_uint4(0u, 0u, 0u, 0u);
    // the first kMoPre candidates in registers (one or two cache lines, requested back to back): the position of a last node among them is
    // the number of smaller ones — no dependent loads; longer blocks (first-order hubs) finish by bisection behind them
    int32_t pre[kMoPre];
#pragma unroll
    for (int q = 0; q < kMoPre; ++q) pre[q] = q < bc ? cand[q] : 0x7fffffff;
    int32_t t = row_ptr[s] - 1;
    int32_t at = 0;                      // candidates below `at` are smaller than the current last node
    float acc = 0.f;
    int32_t cnt = 0, cs = 0;
#pragma unroll
    for (int q = 0; q < kMoSmall; ++q) {
        if (q < n) {
            if (it[q].y & kHeadBit) {
                if (cnt) { w_out[t] = kWeighted ? acc : (float)cnt; if (!kLast) csum_out[t] = cs; }
                ++t;
                acc = 0.f; cnt = 0; cs = 0;
                const int32_t dd = (int32_t)it[q].z;
                int32_t pos = 0;
#pragma unroll
                for (int c = 0; c < kMoPre; ++c) pos += pre[c] < dd ? 1 : 0;
                if (pos == kMoPre) pos = mo_find(cand, at > kMoPre ? at : kMoPre, bc, dd);
                at = pos + 1;
                col_out[t] = bf + pos;
                if (!kLast) { tptr_out[t] = c0 + q; tlast_out[t] = dd; }
            }
            acc += __uint_as_float(it[q].w);
            ++cnt;
            cs += (int32_t)(it[q].y & ~kHeadBit);
        }
    }
    if (cnt) { w_out[t] = kWeighted ? acc : (float)cnt; if (!kLast) csum_out[t] = cs; }
}

// The same for the types of class 1 (k_mo_children_wave's), ONE WAVE PER 64 CONSECUTIVE TYPES, lanes = CHILD SLOTS in rounds of 64: the
// records arrive by one coalesced load per round, a slot's type by bisection over the 64 types' first children (LDS), the id of a new type
// = first id of its parent's row + the heads of that parent before it (ballots; `h_span` carries the count for a parent that spans rounds).
// A new type's run of children (up to the next head) is summed by its head lane LEFT TO RIGHT from LDS (PyG's coalesce order); a run that
// reaches the end of a round stays open (`o_*`) and is extended by lane 0 of the next round.
struct MoTypesLds {
    int32_t cb[kMoWaveTypes + 1], bf[kMoWaveTypes], bc[kMoWaveTypes], rp[kMoWaveTypes];
    uint32_t w[kWave], cc[kWave];
    uint8_t mid[kMoWaveTypes];
};

template <bool kWeighted, bool kLast>
__global__ __launch_bounds__(kBlock) void k_mo_types_wave(int64_t n_types, const int32_t* __restrict__ ibase, const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ cand_ptr, const int32_t* __restrict__ cand_last,
                                                         const void* __restrict__ child, const int32_t* __restrict__ row_ptr,
                                                         const uint8_t* __restrict__ cls, int32_t* __restrict__ tptr_out,
                                                         int32_t* __restrict__ tlast_out, int32_t* __restrict__ col_out, float* __restrict__ w_out,
                                                         int32_t* __restrict__ csum_out, int all_wave) {
    __shared__ MoTypesLds lds[kWavesPerBlock];
    MoTypesLds& L = lds[wave_id()];
    const int lane = lane_id();
    for (int64_t first = ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * kMoWaveTypes; first < n_types;
         first += (int64_t)gridDim.x * kWavesPerBlock * kMoWaveTypes) {
    const int64_t s = first + lane;
    const bool valid = s < n_types;
    const bool is_mid = valid && (cls[s] == 1 || (all_wave && cls[s] == 0 && ibase[s + 1] > ibase[s]));
    const uint64_t mids = __ballot(is_mid);
    if (mids == 0ull) continue;
    const int32_t c0 = ibase[valid ? s : n_types];
    const int32_t c1 = valid ? ibase[s + 1] : c0;
    L.cb[lane] = c0;
    if (lane == kWave - 1) L.cb[kWave] = c1;
    L.mid[lane] = is_mid ? 1 : 0;
    if (is_mid) {
        const int32_t u = col[s];
        const int32_t b0 = cand_ptr[u];
        L.bf[lane] = b0; L.bc[lane] = cand_ptr[u + 1] - b0; L.rp[lane] = row_ptr[s];
    }
    __builtin_amdgcn_wave_barrier();
    const int first_mid = __ffsll((long long)mids) - 1, last_mid = 63 - __clzll((long long)mids);
    const int32_t c_begin = L.cb[first_mid], c_end = L.cb[last_mid + 1];
    int32_t h_span = 0;                  // heads of the type that owns lane 0's slot, in earlier rounds
    int32_t o_t = -1, o_cnt = 0, o_cs = 0;    // the open run: its type id, children so far, their children
    float o_acc = 0.f;
    for (int32_t cr = c_begin; cr < c_end; cr += kWave) {
        const int32_t c = cr + lane;
        int lo = 0, hi = kWave;          // last type whose children start at or before c
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (L.cb[mid] <= c) lo = mid; else hi = mid;
        }
        const bool act = c < c_end && L.mid[lo];
        const uint64_t acts = __ballot(act);
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (act) rec = mo_load_child<mo_fmt(kWeighted, kLast)>(child, c);
        const bool head = act && (rec.y & kHeadBit);
        const uint64_t hb = __ballot(head);
        // close or extend the open run (lane 0 speaks for it)
        if (o_t >= 0 && !((acts & 1ull) && !(hb & 1ull))) {
            if (lane == 0) { w_out[o_t] = kWeighted ? o_acc : (float)o_cnt; if (!kLast) csum_out[o_t] = o_cs; }
            o_t = -1;
        }
        if (acts == 0ull) { h_span = 0; continue; }
        L.w[lane] = rec.w; L.cc[lane] = rec.y & ~kHeadBit;
        __builtin_amdgcn_wave_barrier();
        // my run ends before the next head, the first inactive slot, or the end of the round
        const uint64_t stops = (hb | ~acts) & ~((2ull << lane) - 1ull);
        const int run_end = stops ? __ffsll((long long)stops) - 1 : kWave;
        const int st = L.cb[lo] - cr;                                         // lane of my type's first slot (negative: an earlier round)
        const uint64_t before_type = st > 0 ? ((1ull << st) - 1ull) : 0ull;
        const int32_t t = act ? L.rp[lo] + (st < 0 ? h_span : 0) + (int32_t)__popcll(hb & lanemask_lt() & ~before_type) : 0;
        const bool owner = head || (lane == 0 && act);                        // (lane 0 without a head: the open run goes on)
        float acc = 0.f;
        int32_t cs = 0;
        if (owner) {
            if (!head) { acc = o_acc; cs = o_cs; }
            for (int q = lane; q < run_end; ++q) { acc += __uint_as_float(L.w[q]); cs += (int32_t)L.cc[q]; }
        }
        const int32_t cnt = (owner && !head ? o_cnt : 0) + (run_end - lane);
        const int32_t my_t = head ? t : o_t;
        if (head) {
            const int32_t dd = (int32_t)rec.z, b0 = L.bf[lo], bcount = L.bc[lo];
            const int32_t* cand = cand_last + b0;
            // up to 8 candidates (the rule above the first order): requested together, the position is the number of smaller ones; longer
            // blocks (a first-order node's out-edges: ~20) by bisection — 5 dependent loads instead of a 16-trip loop of them (same-box A/B,
            // headline stream: K = 3 4.4 -> 4.2 ms; configs[2] generator K = 3 11.67 -> 11.37 ms)
            int32_t pos = 0;
            if (bcount <= 2) {                   // (the common block above the first order: two load instructions, not eight)
                const int32_t c0v = bcount > 0 ? cand[0] : 0x7fffffff, c1v = bcount > 1 ? cand[1] : 0x7fffffff;
                pos = (c0v < dd ? 1 : 0) + (c1v < dd ? 1 : 0);
            } else if (bcount <= 8) {
                int32_t c8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) c8[q] = q < bcount ? cand[q] : 0x7fffffff;
#pragma unroll
                for (int q = 0; q < 8; ++q) pos += c8[q] < dd ? 1 : 0;
            } else {
                pos = mo_find(cand, 0, bcount, dd);
            }
            col_out[t] = b0 + pos;
            if (!kLast) { tptr_out[t] = c; tlast_out[t] = dd; }
        }
        // the run that reaches the end of the round stays open; the others are final
        const uint64_t owners = __ballot(owner);
        const int last_owner = 63 - __clzll((long long)owners);               // (owners != 0: the first active slot of a round is a head or lane 0)
        const bool reaches = __shfl(run_end, last_owner, kWave) == kWave;
        if (owner && !(lane == last_owner && reaches)) { w_out[my_t] = kWeighted ? acc : (float)cnt; if (!kLast) csum_out[my_t] = cs; }
        if (reaches) {
            o_t = __shfl(my_t, last_owner, kWave); o_acc = __shfl(acc, last_owner, kWave);
            o_cs = __shfl(cs, last_owner, kWave); o_cnt = __shfl(cnt, last_owner, kWave);
        } else {
            o_t = -1;
        }
        // heads of the type that continues into the next round
        const int st_last = __shfl(st, kWave - 1, kWave);
        const uint64_t before_last = st_last > 0 ? ((1ull << st_last) - 1ull) : 0ull;
        h_span = (st_last < 0 ? h_span : 0) + (int32_t)__popcll(hb & ~before_last);
        __builtin_amdgcn_wave_barrier();
    }
    if (o_t >= 0 && lane == 0) { w_out[o_t] = kWeighted ? o_acc : (float)o_cnt; if (!kLast) csum_out[o_t] = o_cs; }
    __builtin_amdgcn_wave_barrier();
    }
}

template <bool kWeighted, bool kLast>
__global__ __launch_bounds__(kBlock) void k_mo_types_big(const int32_t* __restrict__ ibase, const int32_t* __restrict__ col, const int32_t* __restrict__ cand_ptr,
                                                        const int32_t* __restrict__ cand_last, const void* __restrict__ child,
                                                        const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ big_list,
                                                        const int32_t* __restrict__ big_count, int32_t* __restrict__ tptr_out,
                                                        int32_t* __restrict__ tlast_out, int32_t* __restrict__ col_out, float* __restrict__ w_out,
                                                        int32_t* __restrict__ csum_out) {
    __shared__ uint32_t s_scratch[kWavesPerBlock + 1];
    const int n_big = *big_count;
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < n_big; g += gridDim.x) {
        const int32_t s = big_list[g];
        const int32_t c0 = ibase[s], n = ibase[s + 1] - c0;
        if (n > kMoBigMax) continue;
        const int32_t u = col[s];
        const int32_t bf = cand_ptr[u], bc = cand_ptr[u + 1] - bf;
        const int32_t t0 = row_ptr[s];
        uint32_t ranks = 0;
        for (int32_t rb = 0; rb < n; rb += kBlock) {
            const int32_t r = rb + tid;
            uint4 it = make_uint4(0u, 0u, 0u, 0u);
            if (r < n) it = mo_load_child<mo_fmt(kWeighted, kLast)>(child, c0 + r);
            const bool h = r < n && (it.y & kHeadBit);
            uint32_t total;
            const uint32_t ex = block_exclusive_sum<uint32_t>(h ? 1u : 0u, s_scratch, &total);
            if (h) {
                const int32_t t = t0 + (int32_t)(ranks + ex);
                const int32_t dd = (int32_t)it.z;
                col_out[t] = bf + mo_find(cand_last + bf, 0, bc, dd);
                if (!kLast) { tptr_out[t] = c0 + r; tlast_out[t] = dd; }
                float acc = __uint_as_float(it.w);
                int32_t cnt = 1, cs = (int32_t)(it.y & ~kHeadBit);
                for (int32_t q = r + 1; q < n; ++q) {
                    const uint4 nx = mo_load_child<mo_fmt(kWeighted, kLast)>(child, c0 + q);
                    if (nx.y & kHeadBit) break;
                    acc += __uint_as_float(nx.w);
                    ++cnt;
                    cs += (int32_t)(nx.y & ~kHeadBit);
                }
                w_out[t] = kWeighted ? acc : (float)cnt;
                if (!kLast) csum_out[t] = cs;
            }
            ranks += total;
        }
    }
}

// ------------------------------------------------------------------ continuation windows from a GIVEN event graph (from_temporal_graph(event_graph=...))
// the list entries (targets of the event graph's edges, in its order) as 32-bit ids; status bit 0: an event id outside [0, m)
__global__ __launch_bounds__(kBlock) void k_mo_graph_ids(const int64_t* __restrict__ event_graph, int64_t e2, int64_t m, uint32_t* __restrict__ ids,
                                                        int64_t* __restrict__ status) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= e2) return;
    const int64_t i = event_graph[p];
    int64_t j = event_graph[e2 + p];
    if (i < 0 || i >= m || j < 0 || j >= m) { atomicOr((unsigned long long*)status, (unsigned long long)kMoBadIndex); j = 0; }
    if (p + 1 < e2 && event_graph[p + 1] < i) atomicOr((unsigned long long*)status, (unsigned long long)kMoUnsorted);
    ids[p] = (uint32_t)j;
}
// window of event e = its out-edges in the (source-sorted) event graph: [row pointer, out-degree]
__global__ __launch_bounds__(kBlock) void k_mo_graph_windows(const int32_t* __restrict__ rowptr, int64_t m, uint32_t* __restrict__ first_pos,
                                                            int32_t* __restrict__ count) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= m) return;
    const int32_t a = rowptr[e], b = rowptr[e + 1];
    first_pos[e] = b > a ? (uint32_t)a : 0u;
    count[e] = b - a;
}

struct MoGraphWs {
    int64_t* result;          // {E2, status}
    uint32_t *ids, *first_pos;
    int32_t *count, *outdeg, *rowptr;
    void* scratch;
    size_t scratch_bytes, total_bytes;
};
static MoGraphWs carve_mo_graph(void* ws, int64_t m, int64_t e2) {
    Arena a(ws, (size_t)-1);
    MoGraphWs w;
    w.result = a.take<int64_t>(2);
    w.ids = a.take<uint32_t>(e2 + 4);
    w.first_pos = a.take<uint32_t>(m);
    w.count = a.take<int32_t>(m);
    w.outdeg = a.take<int32_t>(m);
    w.rowptr = a.take<int32_t>(m + 1);
    w.scratch_bytes = scan_ws_bytes(m + 1);
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

struct MoPrepWs {
    int64_t* result;          // {types of level 1, status, children of level 1 (= E2), node pairs with long runs}
    void *keys_a, *keys_b;    // (source, target) keys before / after the sort: uint32 or uint64 [m]
    uint32_t* perm;           // [m] the events in (source, target, time) order
    uint4* ev;                // [m] per event {head node, window first, window count, weight}
    int32_t *head, *head_before, *csum, *long_list, *counters;
    void* scratch;
    size_t scratch_bytes, total_bytes;
};

static MoPrepWs carve_mo_prep(void* ws, int64_t m) {
    Arena a(ws, (size_t)-1);
    MoPrepWs w;
    w.result = a.take<int64_t>(4);
    w.keys_a = a.take<uint64_t>(m);
    w.keys_b = a.take<uint64_t>(m);
    w.perm = a.take<uint32_t>(m);
    w.ev = a.take<uint4>(m);
    w.head = a.take<int32_t>(m);
    w.head_before = a.take<int32_t>(m + 1);
    w.csum = a.take<int32_t>(m);
    w.long_list = a.take<int32_t>(m);
    w.counters = a.take<int32_t>(4);
    const size_t s1 = sort_ws_bytes(m, 8), s2 = scan_ws_bytes(m + 1);
    w.scratch_bytes = s1 > s2 ? s1 : s2;
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

template <typename KeyT>
static int mo_level1(const MoPrepWs& p, const TemporalLists& tl, const int64_t* src, const int64_t* dst, const float* weight, int64_t m,
                     int64_t num_nodes, int bits, uint4* inst, int32_t* tptr, int32_t* tlast, int32_t* rowptr, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div(m, kBlock);
    KeyT* keys = (KeyT*)p.keys_a;
    KeyT* sorted = (KeyT*)p.keys_b;
    // the events in (source, target, time) order: ONE stable sort of the time-ordered stream by the (source, target) key
    k_mo_key_pair<KeyT><<<grid, kBlock, 0, st>>>(src, dst, m, num_nodes, bits, keys);
    PP_LAUNCH_CHECK();
    int rc = sort_pairs<KeyT>(keys, nullptr, sorted, p.perm, m, 0, 2 * bits, p.scratch, p.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_mo_events<<<grid, kBlock, 0, st>>>(dst, tl.first_pos, tl.count, weight, m, num_nodes, p.ev);
    PP_LAUNCH_CHECK();
    k_mo_inst1<KeyT><<<grid, kBlock, 0, st>>>(sorted, p.perm, p.ev, m, inst, p.head);
    PP_LAUNCH_CHECK();
    rc = exclusive_scan<int32_t, int32_t>(p.head, m, p.head_before, true, p.result, p.scratch, p.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_mo_types1<KeyT><<<grid, kBlock, 0, st>>>(sorted, bits, p.head_before, m, num_nodes, tptr, tlast, rowptr);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

struct MoStepWs {
    int64_t* result;          // {new types, status, children of the new level, types handled by workgroups}
    int32_t *deg, *csum, *big_list, *counters;
    uint8_t* cls;
    void* scratch;
    size_t scratch_bytes, total_bytes;
};

static MoStepWs carve_mo_step(void* ws, int64_t n_types, int64_t n_children) {
    Arena a(ws, (size_t)-1);
    MoStepWs w;
    w.result = a.take<int64_t>(4);
    w.deg = a.take<int32_t>(n_types);
    w.csum = a.take<int32_t>(n_children);
    w.cls = a.take<uint8_t>(n_types);
    w.big_list = a.take<int32_t>(n_types);
    w.counters = a.take<int32_t>(4);
    const int64_t longest = n_types > n_children ? n_types : n_children;
    w.scratch_bytes = scan_ws_bytes(longest + 1);
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

__global__ void k_mo_finish(const int32_t* __restrict__ counters, const int64_t* __restrict__ lift_result, int64_t* __restrict__ result) {
    result[3] = counters[0];
    if (lift_result) atomicOr((unsigned long long*)(result + 1), (unsigned long long)lift_result[1]);
}

}  // namespace pp

using namespace pp;

extern "C" {

size_t pp_multiorder_prepare_ws_bytes(int64_t m) { return carve_mo_prep(nullptr, m).total_bytes; }

static int mo_prepare(const int64_t* edge_index, int64_t m, int64_t num_nodes, const float* weight, const TemporalLists& tl, int64_t n_list,
                      bool sort_lists, void* tab, void* inst, int32_t* tptr, int32_t* ibase, int32_t* tlast, float* w, int32_t* rowptr, void* ws, size_t ws_bytes,
                      hipStream_t st) {
    PP_REQUIRE(m > 0 && num_nodes > 0, PP_ERR_ARG, "pp_multiorder_prepare: empty stream");
    PP_REQUIRE(m < (int64_t)0x7ffffff0 && num_nodes < ((int64_t)1 << 31), PP_ERR_TOO_LARGE, "pp_multiorder_prepare: m or num_nodes >= 2^31");
    MoPrepWs p = carve_mo_prep(ws, m);
    PP_REQUIRE(ws_bytes >= p.total_bytes, PP_ERR_WORKSPACE, "pp_multiorder_prepare: workspace too small");
    PP_HIP(hipMemsetAsync(p.result, 0, 4 * sizeof(int64_t), st));
    PP_HIP(hipMemsetAsync(p.counters, 0, 4 * sizeof(int32_t), st));
    PP_HIP(hipMemsetAsync(p.csum, 0, (size_t)m * sizeof(int32_t), st));
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + m;
    const unsigned grid = (unsigned)ceil_div(m, kBlock);
    const int bits = bits_for((uint64_t)(num_nodes - 1));
    int rc;
    bool tab_done = false;               // (the list kernels write the window table on their way)
    if (sort_lists && tl.rowptr != nullptr) {
        // the per-node lists of pp_temporal_count, every list sorted by target in LDS: no second global sort
        k_mo_events<<<grid, kBlock, 0, st>>>(dst, tl.first_pos, tl.count, weight, m, num_nodes, p.ev);
        PP_LAUNCH_CHECK();
        const int64_t pieces = ceil_div(num_nodes, kBlock);
        PP_HIP(hipMemsetAsync(p.head, 0, (size_t)m * sizeof(int32_t), st));       // (a list beyond the workgroup kernel stays unwritten: no heads there)
        k_mo_lists_wave<<<(unsigned)(pieces < 4096 ? pieces : 4096), kBlock, 0, st>>>(num_nodes, tl.rowptr, tl.ids, p.ev, (uint4*)inst, p.head, (uint4*)tab,
                                                                                      p.long_list, p.counters, p.result + 1);
        PP_LAUNCH_CHECK();
        k_mo_lists_big<<<1024, kBlock, 0, st>>>(tl.rowptr, tl.ids, p.ev, (uint4*)inst, p.head, (uint4*)tab, p.long_list, p.counters);
        tab_done = true;
        PP_LAUNCH_CHECK();
        rc = exclusive_scan<int32_t, int32_t>(p.head, m, p.head_before, true, p.result, p.scratch, p.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        const int64_t longest = m > num_nodes + 1 ? m : num_nodes + 1;
        k_mo_types1_lists<<<(unsigned)ceil_div(longest, kBlock), kBlock, 0, st>>>((const uint4*)inst, p.head_before, tl.rowptr, m, num_nodes, tptr, tlast, rowptr);
        PP_LAUNCH_CHECK();
        PP_HIP(hipMemsetAsync(p.counters, 0, 4 * sizeof(int32_t), st));          // (the list of long lists is done with; k_mo_sums1 fills its own)
    } else {
        rc = 2 * bits <= 32 ? mo_level1<uint32_t>(p, tl, src, dst, weight, m, num_nodes, bits, (uint4*)inst, tptr, tlast, rowptr, st)
                            : mo_level1<uint64_t>(p, tl, src, dst, weight, m, num_nodes, bits, (uint4*)inst, tptr, tlast, rowptr, st);
        if (rc != PP_OK) return rc;
    }
    if (weight) k_mo_sums1<true><<<grid, kBlock, 0, st>>>(tptr, p.head_before + m, (const uint4*)inst, w, p.csum, p.long_list, p.counters, p.result + 1);
    else k_mo_sums1<false><<<grid, kBlock, 0, st>>>(tptr, p.head_before + m, (const uint4*)inst, w, p.csum, p.long_list, p.counters, p.result + 1);
    PP_LAUNCH_CHECK();
    if (weight) k_mo_sums1_long<true><<<256, kBlock, 0, st>>>(tptr, (const uint4*)inst, p.long_list, p.counters, w, p.csum, p.result + 1);
    else k_mo_sums1_long<false><<<256, kBlock, 0, st>>>(tptr, (const uint4*)inst, p.long_list, p.counters, w, p.csum, p.result + 1);
    PP_LAUNCH_CHECK();
    rc = exclusive_scan<int32_t, int32_t>(p.csum, m, ibase, true, p.result + 2, p.scratch, p.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    if (n_list > 0 && !tab_done) k_mo_tab<<<(unsigned)ceil_div(n_list, kBlock), kBlock, 0, st>>>(tl.ids, p.ev, n_list, (uint4*)tab);
    PP_LAUNCH_CHECK();
    k_mo_finish<<<1, 1, 0, st>>>(p.counters, tl.result, p.result);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_multiorder_prepare(const int64_t* edge_index, int64_t m, int64_t num_nodes, const float* weight, void* lift_ws, size_t lift_ws_bytes,
                          int radix_sort, void* tab, void* inst, int32_t* tptr, int32_t* ibase, int32_t* tlast, float* w, int32_t* rowptr, void* ws,
                          size_t ws_bytes, pp_stream_t stream) {
    PP_REQUIRE(m > 0 && num_nodes > 0, PP_ERR_ARG, "pp_multiorder_prepare: empty stream");
    const TemporalLists tl = temporal_lists(lift_ws, m, num_nodes);
    PP_REQUIRE(lift_ws_bytes >= tl.total_bytes, PP_ERR_WORKSPACE, "pp_multiorder_prepare: not a pp_temporal_count workspace of this stream");
    return mo_prepare(edge_index, m, num_nodes, weight, tl, m, radix_sort == 0, tab, inst, tptr, ibase, tlast, w, rowptr, ws, ws_bytes, (hipStream_t)stream);
}

size_t pp_multiorder_graph_ws_bytes(int64_t m, int64_t num_event_edges) { return carve_mo_graph(nullptr, m, num_event_edges).total_bytes; }

int pp_multiorder_prepare_graph(const int64_t* edge_index, int64_t m, int64_t num_nodes, const float* weight, const int64_t* event_graph,
                                int64_t num_event_edges, void* graph_ws, size_t graph_ws_bytes, void* tab, void* inst, int32_t* tptr,
                                int32_t* ibase, int32_t* tlast, float* w, int32_t* rowptr, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(m > 0 && num_nodes > 0 && num_event_edges > 0, PP_ERR_ARG, "pp_multiorder_prepare_graph: empty stream or event graph");
    PP_REQUIRE(m < (int64_t)0x7ffffff0 && num_event_edges < (int64_t)0x7ffffff0, PP_ERR_TOO_LARGE, "pp_multiorder_prepare_graph: 2^31 events or event-graph edges");
    MoGraphWs gws = carve_mo_graph(graph_ws, m, num_event_edges);
    PP_REQUIRE(graph_ws_bytes >= gws.total_bytes, PP_ERR_WORKSPACE, "pp_multiorder_prepare_graph: event-graph workspace too small");
    PP_HIP(hipMemsetAsync(gws.result, 0, 2 * sizeof(int64_t), st));
    k_mo_graph_ids<<<(unsigned)ceil_div(num_event_edges, kBlock), kBlock, 0, st>>>(event_graph, num_event_edges, m, gws.ids, gws.result + 1);
    PP_LAUNCH_CHECK();
    int rc = histogram<int64_t>(event_graph, num_event_edges, m, gws.outdeg, st);
    if (rc != PP_OK) return rc;
    rc = exclusive_scan<int32_t, int32_t>(gws.outdeg, m, gws.rowptr, true, gws.result, gws.scratch, gws.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_mo_graph_windows<<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>(gws.rowptr, m, gws.first_pos, gws.count);
    PP_LAUNCH_CHECK();
    const TemporalLists tl{gws.ids, nullptr, gws.first_pos, gws.count, gws.result, gws.total_bytes};
    return mo_prepare(edge_index, m, num_nodes, weight, tl, num_event_edges, false, tab, inst, tptr, ibase, tlast, w, rowptr, ws, ws_bytes, st);
}

const int64_t* pp_multiorder_result_ptr(void* ws) { return (const int64_t*)ws; }

size_t pp_multiorder_step_ws_bytes(int64_t n_types, int64_t n_children) { return carve_mo_step(nullptr, n_types, n_children).total_bytes; }

int pp_multiorder_step(int64_t n_types, int64_t n_children, const int32_t* tptr, const int32_t* ibase, const int32_t* col, const void* inst,
                       const int32_t* cand_ptr, const int32_t* cand_last, const void* tab, int weighted, int last, void* child, int32_t* row_ptr,
                       int32_t* tptr_out, int32_t* ibase_out, int32_t* tlast_out, int32_t* col_out, float* w_out, void* ws, size_t ws_bytes,
                       pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_types > 0 && n_children > 0, PP_ERR_ARG, "pp_multiorder_step: empty level");
    PP_REQUIRE(n_types < (int64_t)0x7ffffff0 && n_children < (int64_t)0x7ffffff0, PP_ERR_TOO_LARGE, "pp_multiorder_step: level with 2^31 or more instances");
    MoStepWs p = carve_mo_step(ws, n_types, n_children);
    PP_REQUIRE(ws_bytes >= p.total_bytes, PP_ERR_WORKSPACE, "pp_multiorder_step: workspace too small");
    PP_HIP(hipMemsetAsync(p.result, 0, 4 * sizeof(int64_t), st));
    PP_HIP(hipMemsetAsync(p.counters, 0, 4 * sizeof(int32_t), st));
    const unsigned grid = (unsigned)ceil_div(n_types, kBlock);
    const int64_t pieces = ceil_div(n_types, kWavesPerBlock * kMoWaveTypes);
    const unsigned wave_grid = (unsigned)(pieces < 4096 ? pieces : 4096);
#define PP_MO_CHILDREN(F)                                                                                                                          \
    do {                                                                                                                                           \
        k_mo_children<F><<<grid, kBlock, 0, st>>>(n_types, tptr, ibase, (const uint4*)inst, (const uint4*)tab, child, p.deg, p.cls, p.big_list,    \
                                                  p.counters);                                                                                     \
        k_mo_children_wave<F><<<wave_grid, kBlock, 0, st>>>(tptr, ibase, (const uint4*)inst, (const uint4*)tab, child, p.deg, p.cls, n_types);     \
        k_mo_children_big<F><<<1024, kBlock, 0, st>>>(tptr, ibase, (const uint4*)inst, (const uint4*)tab, child, p.deg, p.big_list, p.counters,    \
                                                      p.result + 1);                                                                               \
    } while (0)
    if (!last) PP_MO_CHILDREN(0);
    else if (weighted) PP_MO_CHILDREN(2);
    else PP_MO_CHILDREN(1);
#undef PP_MO_CHILDREN
    PP_LAUNCH_CHECK();
    int rc = exclusive_scan<int32_t, int32_t>(p.deg, n_types, row_ptr, true, p.result, p.scratch, p.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    if (!last) PP_HIP(hipMemsetAsync(p.csum, 0, (size_t)n_children * sizeof(int32_t), st));
    // the lane types' share of the types pass: below the top layer (five outputs per new type) the wave kernel's coalesced stores win
    // (headline stream, 10^7 .. 3.6 * 10^7 parent types: 0.65 / 0.65 / 1.06 ms against 0.73 / 0.92 / 1.25 ms), at the top layer (two outputs)
    // one lane per type does (1.44 against 1.69 ms at 6.8 * 10^7 types).  PP_MO_TYPES_WAVE=0/1: measurement switch
    static const char* const force = getenv("PP_MO_TYPES_WAVE");
    const int all_wave = force ? (force[0] == '1' ? 1 : 0) : (last ? 0 : 1);
#define PP_MO_TYPES(W, L)                                                                                                                          \
    do {                                                                                                                                           \
        k_mo_types<W, L><<<grid, kBlock, 0, st>>>(n_types, tptr, ibase, col, cand_ptr, cand_last, child, row_ptr, tptr_out,                        \
                                                  tlast_out, col_out, w_out, p.csum, all_wave);                                                              \
        k_mo_types_wave<W, L><<<all_wave ? (unsigned)pieces : wave_grid, kBlock, 0, st>>>(                                                         \
            n_types, ibase, col, cand_ptr, cand_last, child, row_ptr, p.cls, tptr_out, tlast_out, col_out, w_out, p.csum, all_wave);               \
        k_mo_types_big<W, L><<<1024, kBlock, 0, st>>>(ibase, col, cand_ptr, cand_last, child, row_ptr, p.big_list, p.counters,                     \
                                                      tptr_out, tlast_out, col_out, w_out, p.csum);                                                \
    } while (0)
    if (weighted) { if (last) PP_MO_TYPES(true, true); else PP_MO_TYPES(true, false); }
    else { if (last) PP_MO_TYPES(false, true); else PP_MO_TYPES(false, false); }
#undef PP_MO_TYPES
    PP_LAUNCH_CHECK();
    if (!last) {
        rc = exclusive_scan<int32_t, int32_t>(p.csum, n_children, ibase_out, true, p.result + 2, p.scratch, p.scratch_bytes, st);
        if (rc != PP_OK) return rc;
    }
    k_mo_finish<<<1, 1, 0, st>>>(p.counters, nullptr, p.result);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
