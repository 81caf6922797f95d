// pathpyg_amd — De Bruijn aggregation on gfx950: lexicographic unique rows, edge coalescing, CSR/CSC.
//
// Reference functions replaced (paths relative to the pathpyG repository root):
//   aggregate_edge_index                 src/pathpyG/algorithms/lift_order.py:109-152
//     torch.unique(node_sequence, dim=0, return_inverse=True)      :133
//     inverse_idx[edge_index] / node_sequence.squeeze()[edge_index] :135-138
//     torch_geometric.utils.coalesce(..., reduce=aggr)             :139-144
//   Graph.__init__ row sort + CSR/CSC    src/pathpyG/core/graph.py:103-115
//
// Everything is sort -> flag run heads -> scan -> scatter, i.e. HBM streams around pp_sort.hip:
//   unique rows : LSD over the k columns (last column first), each column an LSD radix sort of
//                 (column value, row id) pairs restricted to the significant bits of max-min;
//                 heads = rows that differ from their predecessor; inverse[perm[i]] = #heads before i.
//   coalesce    : key = (row << b) | col with b = bits(U-1) (same order as PyG's row*U+col), radix sort
//                 of (key, edge id), heads -> segment starts, one thread per output edge reduces its
//                 run LEFT TO RIGHT in the stable sorted order (= the order PyG's CPU scatter adds in).
#include "pp_internal.h"

#include <type_traits>

namespace pp {

constexpr int64_t kBadIndex = 1;

// ------------------------------------------------------------------ unique rows
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_column_keys(const int64_t* __restrict__ rows, int64_t n_rows, int k, int col,
                                                       const uint32_t* __restrict__ perm, int64_t bias, KeyT* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t r = perm ? (int64_t)perm[i] : i;
    keys[i] = (KeyT)(uint64_t)(rows[r * k + col] - bias);
}

__global__ __launch_bounds__(kBlock) void k_row_heads(const int64_t* __restrict__ rows, int64_t n_rows, int k,
                                                     const uint32_t* __restrict__ perm, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    int32_t h = 1;
    if (i > 0) {
        const int64_t a = (int64_t)perm[i] * k, b = (int64_t)perm[i - 1] * k;
        h = 0;
        for (int c = 0; c < k; ++c) h |= rows[a + c] != rows[b + c];
    }
    head[i] = h;
}

__global__ __launch_bounds__(kBlock) void k_rank_rows(int64_t n_rows, const uint32_t* __restrict__ perm, const int32_t* __restrict__ head,
                                                     const int32_t* __restrict__ heads_before, int64_t* __restrict__ inverse,
                                                     uint32_t* __restrict__ first_row_of) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    const int32_t id = heads_before[i] + head[i] - 1;
    inverse[perm[i]] = id;
    if (head[i]) first_row_of[id] = perm[i];
}

__global__ __launch_bounds__(kBlock) void k_gather_rows(const int64_t* __restrict__ rows, int k, const uint32_t* __restrict__ which,
                                                       int64_t n_out, int64_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n_out * k) return;
    const int64_t u = idx / k;
    const int c = (int)(idx - u * k);
    out[idx] = rows[(int64_t)which[u] * k + c];
}

struct UniqueWs {
    int64_t* result;          // {U, status}
    uint32_t* perm_a;         // [M]
    uint32_t* perm_b;         // [M]
    void* keys_a;             // [M] u32 or u64
    void* keys_b;             // [M]
    int32_t* head;            // [M]
    int32_t* heads_before;    // [M+1]
    uint32_t* first_row_of;   // [M]
    void* scratch;
    size_t scratch_bytes;
    size_t total_bytes;
};

static UniqueWs carve_unique(void* ws, int64_t m) {
    Arena a(ws, (size_t)-1);
    UniqueWs w;
    w.result = a.take<int64_t>(2);
    w.perm_a = a.take<uint32_t>(m);
    w.perm_b = a.take<uint32_t>(m);
    w.keys_a = a.take<uint64_t>(m);
    w.keys_b = a.take<uint64_t>(m);
    w.head = a.take<int32_t>(m);
    w.heads_before = a.take<int32_t>(m + 1);
    w.first_row_of = a.take<uint32_t>(m);
    size_t s1 = sort_ws_bytes(m, 8), s2 = scan_ws_bytes(m);
    w.scratch_bytes = s1 > s2 ? s1 : s2;
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

template <typename KeyT>
static int lexsort_rows(const int64_t* rows, int64_t m, int k, int64_t bias, int bits, UniqueWs& w, hipStream_t st, uint32_t** perm_out) {
    const unsigned grid = (unsigned)ceil_div(m, kBlock);
    uint32_t* cur = nullptr;            // identity for the first column
    uint32_t* nxt = w.perm_a;
    for (int col = k - 1; col >= 0; --col) {
        k_column_keys<KeyT><<<grid, kBlock, 0, st>>>(rows, m, k, col, cur, bias, (KeyT*)w.keys_a);
        PP_LAUNCH_CHECK();
        int rc = sort_pairs<KeyT>((const KeyT*)w.keys_a, cur, (KeyT*)w.keys_b, nxt, m, 0, bits, w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
        cur = nxt;
        nxt = (cur == w.perm_a) ? w.perm_b : w.perm_a;
    }
    *perm_out = cur;
    return PP_OK;
}

// ------------------------------------------------------------------ coalesce
#ifndef PP_NT_KEYS
#define PP_NT_KEYS 1
#endif
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_edge_keys(const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                     const int64_t* __restrict__ remap, int64_t remap_len, int64_t num_nodes, int shift,
                                                     const int64_t* __restrict__ col_base, KeyT* __restrict__ keys,
                                                     int64_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_edges) return;
#if PP_NT_KEYS
    int64_t r = load_stream(edge_index + e), c = load_stream(edge_index + n_edges + e);      // (instance pairs: read once)
#else
    int64_t r = edge_index[e], c = edge_index[n_edges + e];
#endif
    bool bad = false;
    if (remap) {
        bad = r < 0 || r >= remap_len || c < 0 || c >= remap_len;
        r = bad ? 0 : remap[r];
        c = bad ? 0 : remap[c];
    }
    bad = bad || r < 0 || r >= num_nodes || c < 0 || c >= num_nodes;
    if (!bad && col_base) {                  // block-compressed column: every column of row r lies in [col_base[r], col_base[r] + 2^shift)
        c -= col_base[r];
        bad = c < 0 || c >= ((int64_t)1 << shift);
    }
    if (bad) { atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex); r = 0; c = 0; }
    keys[e] = (KeyT)(((uint64_t)r << shift) | (uint64_t)c);
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_key_heads(const KeyT* __restrict__ sorted_keys, int64_t n, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || sorted_keys[i] != sorted_keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_segment_starts(int64_t n, const int32_t* __restrict__ head, const int32_t* __restrict__ heads_before,
                                                          uint32_t* __restrict__ seg_start) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    if (i == n) { seg_start[heads_before[n]] = (uint32_t)n; return; }
    if (head[i]) seg_start[heads_before[i]] = (uint32_t)i;
}

template <typename T>
__device__ __forceinline__ T reduce_step(T acc, T v, int reduce) {
    switch (reduce) {
        case PP_REDUCE_MIN: return v < acc ? v : acc;
        case PP_REDUCE_MAX: return v > acc ? v : acc;
        default: return acc + v;
    }
}
template <typename T>
__device__ __forceinline__ T mean_of(T sum, uint32_t n) {
    if constexpr (std::is_floating_point<T>::value) return sum / (T)n;
    else {                                   // floor division like torch.div(rounding_mode="floor")
        T q = sum / (T)n, r = sum % (T)n;
        return (r != 0 && ((r < 0) != ((T)n < 0))) ? q - 1 : q;
    }
}

constexpr uint32_t kLongRun = 512;           // parallel-edge runs longer than this are reduced by a whole workgroup

template <typename KeyT, typename T>
__global__ __launch_bounds__(kBlock) void k_coalesce_fill(const KeyT* __restrict__ sorted_keys, const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ seg_start, int64_t n_out, int shift,
                                                         const int64_t* __restrict__ col_base, const T* __restrict__ weight, int reduce,
                                                         int64_t* __restrict__ out_index, T* __restrict__ out_weight,
                                                         uint32_t* __restrict__ long_runs) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = a < n_out;
    const uint32_t p0 = live ? seg_start[a] : 0u, p1 = live ? seg_start[a + 1] : 0u;
    if (live) {
        const uint64_t key = (uint64_t)sorted_keys[p0];
        const int64_t row = (int64_t)(key >> shift);
        out_index[a] = row;
        out_index[n_out + a] = (int64_t)(key & ((1ull << shift) - 1ull)) + (col_base ? col_base[row] : 0);
    }
    if (!weight) {          // unit weights (the reference's default, lift_order.py:130-131): sum = run length, mean / min / max = 1 - no gather
        if (out_weight != nullptr && live) out_weight[a] = reduce == PP_REDUCE_SUM ? (T)(p1 - p0) : (T)1;
        return;
    }
    // A run of thousands of parallel edges (one node pair carrying a large share of a contact stream) is not walked by its lane
    // alone (~1 us per entry): it is queued for k_coalesce_long_runs, a workgroup per run.
    const uint32_t len = p1 - p0;
    if (!live) return;
    if (len > kLongRun) {
        long_runs[1 + atomicAdd(&long_runs[0], 1u)] = (uint32_t)a;
        return;
    }
    T acc = weight[perm[p0]];
    for (uint32_t p = p0 + 1; p < p1; ++p) acc = reduce_step<T>(acc, weight[perm[p]], reduce);
    out_weight[a] = reduce == PP_REDUCE_MEAN ? mean_of<T>(acc, len) : acc;
}

// One workgroup per queued run: 256 threads stride over it with 8 gathers in flight each; partial results are folded in a fixed order
// (wave butterfly, then the waves in wave order) - sums of integer-valued weights stay exact, other float sums differ from the
// left-to-right order only by rounding.  The queue order is arbitrary, the result of each run is not.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_coalesce_long_runs(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ seg_start,
                                                              const T* __restrict__ weight, int reduce, const uint32_t* __restrict__ long_runs,
                                                              T* __restrict__ out_weight) {
    __shared__ T s_part[kWavesPerBlock];
    const uint32_t count = long_runs[0];
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        const uint32_t a = long_runs[1 + k];
        const uint32_t b = seg_start[a], e = seg_start[a + 1];
        T part = weight[perm[b + threadIdx.x]];                     // len > kLongRun >= kBlock: every thread starts from its own element
        constexpr int kUnroll = 8;
        uint32_t p = b + kBlock + threadIdx.x;
        for (; p + (kUnroll - 1) * kBlock < e; p += kUnroll * kBlock) {
            T ww[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) ww[u] = weight[perm[p + u * kBlock]];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) part = reduce_step<T>(part, ww[u], reduce);
        }
        for (; p < e; p += kBlock) part = reduce_step<T>(part, weight[perm[p]], reduce);
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) part = reduce_step<T>(part, __shfl_xor(part, d, kWave), reduce);
        if (lane_id() == 0) s_part[wave_id()] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            T acc = s_part[0];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) acc = reduce_step<T>(acc, s_part[w], reduce);
            out_weight[a] = reduce == PP_REDUCE_MEAN ? mean_of<T>(acc, e - b) : acc;
        }
        __syncthreads();
    }
}

// inverse[e] = index of the merged edge that input edge e ended up in
__global__ __launch_bounds__(kBlock) void k_coalesce_inverse(int64_t n, const uint32_t* __restrict__ perm, const int32_t* __restrict__ head,
                                                            const int32_t* __restrict__ heads_before, int64_t* __restrict__ inverse) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < n) inverse[perm[p]] = heads_before[p] + head[p] - 1;
}

struct CoalesceWs {
    int64_t* result;          // {A, status}
    void* keys_a;
    void* keys_b;
    uint32_t* perm;
    int32_t* head;
    int32_t* heads_before;    // [E+1]
    uint32_t* seg_start;      // [E+1]
    void* scratch;
    size_t scratch_bytes;
    size_t total_bytes;
};

static CoalesceWs carve_coalesce(void* ws, int64_t e) {
    Arena a(ws, (size_t)-1);
    CoalesceWs w;
    w.result = a.take<int64_t>(2);
    w.keys_a = a.take<uint64_t>(e);
    w.keys_b = a.take<uint64_t>(e);
    w.perm = a.take<uint32_t>(e);
    w.head = a.take<int32_t>(e);
    w.heads_before = a.take<int32_t>(e + 1);
    w.seg_start = a.take<uint32_t>(e + 1);
    size_t s1 = sort_ws_bytes(e, 8), s2 = scan_ws_bytes(e);
    w.scratch_bytes = s1 > s2 ? s1 : s2;
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

static inline int coalesce_shift(int64_t num_nodes) { return bits_for((uint64_t)(num_nodes > 1 ? num_nodes - 1 : 1)); }

template <typename KeyT>
static int coalesce_count_impl(const int64_t* edge_index, int64_t e, const int64_t* remap, int64_t remap_len, int64_t num_nodes,
                               const int64_t* col_base, int shift, CoalesceWs& w, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div(e, kBlock);
    k_edge_keys<KeyT><<<grid, kBlock, 0, st>>>(edge_index, e, remap, remap_len, num_nodes, shift, col_base, (KeyT*)w.keys_a, w.result + 1);
    PP_LAUNCH_CHECK();
    int rc = sort_pairs<KeyT>((const KeyT*)w.keys_a, nullptr, (KeyT*)w.keys_b, w.perm, e, 0, coalesce_shift(num_nodes) + shift, w.scratch,
                              w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_key_heads<KeyT><<<grid, kBlock, 0, st>>>((const KeyT*)w.keys_b, e, w.head);
    PP_LAUNCH_CHECK();
    rc = exclusive_scan<int32_t, int32_t>(w.head, e, w.heads_before, true, w.result, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_segment_starts<<<(unsigned)ceil_div(e + 1, kBlock), kBlock, 0, st>>>(e, w.head, w.heads_before, w.seg_start);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <typename KeyT>
static int coalesce_fill_impl(const void* weight, int dtype, int reduce, int64_t n_out, int shift, const int64_t* col_base,
                              int64_t* out_index, void* out_weight, CoalesceWs& w, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div(n_out, kBlock);
    const KeyT* keys = (const KeyT*)w.keys_b;
    uint32_t* long_runs = (uint32_t*)w.scratch;                 // {count, run ids...}: the sort scratch is free again (>= 12 bytes per edge)
    PP_HIP(hipMemsetAsync(long_runs, 0, sizeof(uint32_t), st));
#define PP_FILL(T)                                                                                                                 \
    do {                                                                                                                           \
        k_coalesce_fill<KeyT, T><<<grid, kBlock, 0, st>>>(keys, w.perm, w.seg_start, n_out, shift, col_base, (const T*)weight, reduce,  \
                                                           out_index, (T*)out_weight, long_runs);                                  \
        if (weight) k_coalesce_long_runs<T><<<256, kBlock, 0, st>>>(w.perm, w.seg_start, (const T*)weight, reduce, long_runs, (T*)out_weight); \
    } while (0)
    switch (weight ? dtype : PP_F32) {
        case PP_I32: PP_FILL(int32_t); break;
        case PP_I64: PP_FILL(int64_t); break;
        case PP_F32: PP_FILL(float); break;
        case PP_F64: PP_FILL(double); break;
        default: PP_REQUIRE(false, PP_ERR_ARG, "pp_coalesce_fill: unsupported weight dtype %d", dtype);
    }
#undef PP_FILL
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ------------------------------------------------------------------ sortedness, argsort, CSR pointers
__global__ __launch_bounds__(kBlock) void k_count_descents(const int64_t* __restrict__ a, int64_t n, int64_t* __restrict__ out) {
    // grid-stride: at most one atomic per wave of a bounded grid (an unsorted 10^7-element input used to queue 1.6*10^5 of them on one word)
    uint32_t mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * kBlock) mine += a[i] > a[i + 1] ? 1u : 0u;
    const uint32_t total = wave_sum<uint32_t>(mine);
    if (lane_id() == 0 && total) atomicAdd((unsigned long long*)out, (unsigned long long)total);
}

__global__ __launch_bounds__(kBlock) void k_count_descents_f64(const double* __restrict__ a, int64_t n, int64_t* __restrict__ out) {
    // grid-stride: at most one atomic per wave of a bounded grid (an unsorted 10^7-element input used to queue 1.6*10^5 of them on one word)
    uint32_t mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * kBlock) mine += a[i] > a[i + 1] ? 1u : 0u;
    const uint32_t total = wave_sum<uint32_t>(mine);
    if (lane_id() == 0 && total) atomicAdd((unsigned long long*)out, (unsigned long long)total);
}

// order-preserving map double -> uint64 (negative values: all bits flipped; others: sign bit set); -0.0 == +0.0
__global__ __launch_bounds__(kBlock) void k_keys_from_f64(const double* __restrict__ a, int64_t n, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double v = a[i];
    if (v == 0.0) v = 0.0;
    uint64_t b = (uint64_t)__double_as_longlong(v);
    keys[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_keys_from_i64(const int64_t* __restrict__ a, int64_t n, int64_t bias, KeyT* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) keys[i] = (KeyT)(uint64_t)(a[i] - bias);
}

// the time sort of a whole event stream (TemporalGraph.__init__): one pass reads the permutation once and gathers the three 8-byte
// columns of every event (source, destination, time bits) — instead of one indexing launch per column
__global__ __launch_bounds__(kBlock) void k_gather_events(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ time_bits,
                                                         const int64_t* __restrict__ perm, int64_t m, int64_t* __restrict__ edge_index_out,
                                                         int64_t* __restrict__ time_out, int64_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    int64_t p = perm[i];
    if (p < 0 || p >= m) { atomicOr((unsigned long long*)status, (unsigned long long)kBadIndex); p = 0; }
    edge_index_out[i] = edge_index[p];
    edge_index_out[m + i] = edge_index[m + p];
    time_out[i] = time_bits[p];
}

__global__ __launch_bounds__(kBlock) void k_widen_u32(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// ptr[v] = first position p with sorted[p] >= v, v in [0, num_rows]
__global__ __launch_bounds__(kBlock) void k_ptr_from_sorted_i64(const int64_t* __restrict__ sorted, int64_t n, int64_t num_rows,
                                                               int64_t* __restrict__ ptr) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    int64_t a = p == 0 ? -1 : sorted[p - 1];
    int64_t b = p == n ? num_rows : sorted[p];
    if (a < -1) a = -1;
    if (b > num_rows) b = num_rows;
    for (int64_t v = a + 1; v <= b; ++v) ptr[v] = p;
}

}  // namespace pp

using namespace pp;

extern "C" {

// ---------------------------------------------------------------- unique rows (torch.unique(dim=0, return_inverse=True))
size_t pp_unique_rows_ws_bytes(int64_t n_rows) { return carve_unique(nullptr, n_rows).total_bytes; }

int pp_unique_rows_count(const int64_t* rows, int64_t n_rows, int k, int64_t min_value, int64_t max_value, int64_t* inverse, void* ws,
                         size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && k >= 1, PP_ERR_ARG, "pp_unique_rows_count: bad shape [%lld,%d]", (long long)n_rows, k);
    PP_REQUIRE(n_rows < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_unique_rows_count: more than 2^31 rows");
    UniqueWs w = carve_unique(ws, n_rows);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_unique_rows_count: workspace too small");
    PP_HIP(hipMemsetAsync(w.result, 0, 2 * sizeof(int64_t), st));
    if (n_rows == 0) return PP_OK;
    PP_REQUIRE(max_value >= min_value, PP_ERR_ARG, "pp_unique_rows_count: max_value < min_value");
    const uint64_t span = (uint64_t)max_value - (uint64_t)min_value;
    const int bits = bits_for(span);
    uint32_t* perm = nullptr;
    int rc = bits <= 32 ? lexsort_rows<uint32_t>(rows, n_rows, k, min_value, bits, w, st, &perm)
                        : lexsort_rows<uint64_t>(rows, n_rows, k, min_value, bits, w, st, &perm);
    if (rc != PP_OK) return rc;
    const unsigned grid = (unsigned)ceil_div(n_rows, kBlock);
    k_row_heads<<<grid, kBlock, 0, st>>>(rows, n_rows, k, perm, w.head);
    PP_LAUNCH_CHECK();
    rc = exclusive_scan<int32_t, int32_t>(w.head, n_rows, w.heads_before, true, w.result, w.scratch, w.scratch_bytes, st);
    if (rc != PP_OK) return rc;
    k_rank_rows<<<grid, kBlock, 0, st>>>(n_rows, perm, w.head, w.heads_before, inverse, w.first_row_of);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_unique_rows_fill(const int64_t* rows, int64_t n_rows, int k, int64_t n_unique, int64_t* unique_rows, void* ws, size_t ws_bytes,
                        pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    UniqueWs w = carve_unique(ws, n_rows);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_unique_rows_fill: workspace too small");
    if (n_unique <= 0) return PP_OK;
    k_gather_rows<<<(unsigned)ceil_div(n_unique * k, kBlock), kBlock, 0, st>>>(rows, k, w.first_row_of, n_unique, unique_rows);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ---------------------------------------------------------------- coalesce (PyG coalesce on remapped edges)
size_t pp_coalesce_ws_bytes(int64_t n_edges) { return carve_coalesce(nullptr, n_edges).total_bytes; }

// key layout: (row << shift) | col with shift = bits(num_nodes - 1), or with a caller-supplied column block per row
// (col_base[row] <= col < col_base[row] + 2^col_bits): (row << col_bits) | (col - col_base[row]) - fewer radix passes
static inline int coalesce_col_shift(int64_t num_nodes, const int64_t* col_base, int col_bits) {
    return col_base ? col_bits : coalesce_shift(num_nodes);
}

int pp_coalesce_count(const int64_t* edge_index, int64_t n_edges, const int64_t* remap, int64_t remap_len, int64_t num_nodes,
                      const int64_t* col_base, int col_bits, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_edges >= 0 && num_nodes >= 0, PP_ERR_ARG, "pp_coalesce_count: negative size");
    PP_REQUIRE(n_edges < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_coalesce_count: more than 2^31 edges");
    PP_REQUIRE(num_nodes <= ((int64_t)1 << 32), PP_ERR_TOO_LARGE, "pp_coalesce_count: more than 2^32 nodes");
    CoalesceWs w = carve_coalesce(ws, n_edges);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_coalesce_count: workspace too small");
    PP_HIP(hipMemsetAsync(w.result, 0, 2 * sizeof(int64_t), st));
    if (n_edges == 0) return PP_OK;
    PP_REQUIRE(col_base == nullptr || (col_bits >= 0 && col_bits <= 32), PP_ERR_ARG, "pp_coalesce_count: col_bits must be in [0, 32]");
    const int shift = coalesce_col_shift(num_nodes, col_base, col_bits);
    return coalesce_shift(num_nodes) + shift <= 32
               ? coalesce_count_impl<uint32_t>(edge_index, n_edges, remap, remap_len, num_nodes, col_base, shift, w, st)
               : coalesce_count_impl<uint64_t>(edge_index, n_edges, remap, remap_len, num_nodes, col_base, shift, w, st);
}

int pp_coalesce_fill(const void* weight, int dtype, int reduce, int64_t n_edges, int64_t n_out, int64_t num_nodes, const int64_t* col_base,
                     int col_bits, int64_t* out_index, void* out_weight, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(reduce >= PP_REDUCE_SUM && reduce <= PP_REDUCE_MAX, PP_ERR_ARG, "pp_coalesce_fill: unknown reduce %d", reduce);
    CoalesceWs w = carve_coalesce(ws, n_edges);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_coalesce_fill: workspace too small");
    if (n_out <= 0) return PP_OK;
    const int shift = coalesce_col_shift(num_nodes, col_base, col_bits);
    return coalesce_shift(num_nodes) + shift <= 32
               ? coalesce_fill_impl<uint32_t>(weight, dtype, reduce, n_out, shift, col_base, out_index, out_weight, w, st)
               : coalesce_fill_impl<uint64_t>(weight, dtype, reduce, n_out, shift, col_base, out_index, out_weight, w, st);
}

// inverse[n_edges]: for every input edge the position of its merged edge in the coalesced output (after *_count).
// For a first-order edge list this IS torch.unique(stack(src,dst).T, dim=0, return_inverse=True)[1]: the sorted distinct
// (src,dst) pairs are the order-2 De Bruijn nodes, so layer 2's unique/inverse comes for free from layer 1's coalesce.
int pp_coalesce_inverse(int64_t n_edges, int64_t* inverse, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    CoalesceWs w = carve_coalesce(ws, n_edges);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_coalesce_inverse: workspace too small");
    if (n_edges <= 0) return PP_OK;
    k_coalesce_inverse<<<(unsigned)ceil_div(n_edges, kBlock), kBlock, 0, st>>>(n_edges, w.perm, w.head, w.heads_before, inverse);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// {size, status} of the last *_count on a unique/coalesce workspace
const int64_t* pp_aggregate_result_ptr(void* ws) { return (const int64_t*)ws; }

// ---------------------------------------------------------------- Graph.__init__ helpers
// descents[0] = #positions with a[i] > a[i+1]  (0 <=> already sorted by row, graph.py:103 becomes a no-op)
int pp_count_descents_i64(const int64_t* a, int64_t n, int64_t* descents, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_HIP(hipMemsetAsync(descents, 0, sizeof(int64_t), st));
    if (n < 2) return PP_OK;
    k_count_descents<<<(unsigned)(ceil_div(n, kBlock) < 2048 ? ceil_div(n, kBlock) : 2048), kBlock, 0, st>>>(a, n, descents);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_count_descents_f64(const double* a, int64_t n, int64_t* descents, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_HIP(hipMemsetAsync(descents, 0, sizeof(int64_t), st));
    if (n < 2) return PP_OK;
    k_count_descents_f64<<<(unsigned)(ceil_div(n, kBlock) < 2048 ? ceil_div(n, kBlock) : 2048), kBlock, 0, st>>>(a, n, descents);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// {descents, min, max} of a timestamp vector in one device buffer: ONE read-back decides whether TemporalGraph.__init__ has to sort
// and how many key bits the sort needs (float64: min / max are not used and left 0)
int pp_time_stats(const void* time, int time_dtype, int64_t n, int64_t* out3, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(time_dtype == PP_I64 || time_dtype == PP_F64, PP_ERR_ARG, "pp_time_stats: time must be int64 or float64");
    PP_REQUIRE(n >= 0, PP_ERR_ARG, "pp_time_stats: negative length");
    PP_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(int64_t), st));
    if (n == 0) return PP_OK;
    if (time_dtype == PP_F64) return pp_count_descents_f64((const double*)time, n, out3, stream);
    int rc = pp_count_descents_i64((const int64_t*)time, n, out3, stream);
    if (rc != PP_OK) return rc;
    return minmax_i64((const int64_t*)time, n, out3 + 1, st);
}

// edge_index_out[:, i] = edge_index[:, perm[i]], time_out[i] = time[perm[i]] (8-byte timestamps of either dtype); status bit 1: perm out of range
int pp_gather_events(const int64_t* edge_index, const void* time, const int64_t* perm, int64_t m, int64_t* edge_index_out, void* time_out,
                     int64_t* status, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(m >= 0, PP_ERR_ARG, "pp_gather_events: negative length");
    PP_HIP(hipMemsetAsync(status, 0, sizeof(int64_t), st));
    if (m == 0) return PP_OK;
    k_gather_events<<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>(edge_index, (const int64_t*)time, perm, m, edge_index_out, (int64_t*)time_out,
                                                                     status);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

size_t pp_argsort_ws_bytes(int64_t n) {
    Arena a(nullptr, (size_t)-1);
    a.take<uint64_t>(n); a.take<uint64_t>(n); a.take<uint32_t>(n);
    a.take<char>((int64_t)sort_ws_bytes(n, 8));
    return a.used;
}

// stable argsort of int64 keys in [min_value, max_value] -> int64 permutation (EdgeIndex.sort_by("row"), get_csc)
int pp_argsort_i64(const int64_t* keys, int64_t n, int64_t min_value, int64_t max_value, int64_t* perm_out, void* ws, size_t ws_bytes,
                   pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_argsort_i64: n outside [0, 2^31)");
    PP_REQUIRE(ws_bytes >= pp_argsort_ws_bytes(n), PP_ERR_WORKSPACE, "pp_argsort_i64: workspace too small");
    if (n == 0) return PP_OK;
    PP_REQUIRE(max_value >= min_value, PP_ERR_ARG, "pp_argsort_i64: max_value < min_value");
    Arena a(ws, ws_bytes);
    void* ka = a.take<uint64_t>(n);
    void* kb = a.take<uint64_t>(n);
    uint32_t* perm = a.take<uint32_t>(n);
    const size_t sb = sort_ws_bytes(n, 8);
    void* scratch = a.take<char>((int64_t)sb);
    const int bits = bits_for((uint64_t)max_value - (uint64_t)min_value);
    const unsigned grid = (unsigned)ceil_div(n, kBlock);
    int rc;
    if (bits <= 32) {
        k_keys_from_i64<uint32_t><<<grid, kBlock, 0, st>>>(keys, n, min_value, (uint32_t*)ka);
        PP_LAUNCH_CHECK();
        rc = sort_pairs<uint32_t>((const uint32_t*)ka, nullptr, (uint32_t*)kb, perm, n, 0, bits, scratch, sb, st);
    } else {
        k_keys_from_i64<uint64_t><<<grid, kBlock, 0, st>>>(keys, n, min_value, (uint64_t*)ka);
        PP_LAUNCH_CHECK();
        rc = sort_pairs<uint64_t>((const uint64_t*)ka, nullptr, (uint64_t*)kb, perm, n, 0, bits, scratch, sb, st);
    }
    if (rc != PP_OK) return rc;
    k_widen_u32<<<grid, kBlock, 0, st>>>(perm, n, perm_out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// stable argsort of float64 keys (torch.argsort(time), temporal_graph.py:58 - made stable here); NaNs sort last
int pp_argsort_f64(const double* keys, int64_t n, int64_t* perm_out, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_argsort_f64: n outside [0, 2^31)");
    PP_REQUIRE(ws_bytes >= pp_argsort_ws_bytes(n), PP_ERR_WORKSPACE, "pp_argsort_f64: workspace too small");
    if (n == 0) return PP_OK;
    Arena a(ws, ws_bytes);
    uint64_t* ka = a.take<uint64_t>(n);
    uint64_t* kb = a.take<uint64_t>(n);
    uint32_t* perm = a.take<uint32_t>(n);
    const size_t sb = sort_ws_bytes(n, 8);
    void* scratch = a.take<char>((int64_t)sb);
    const unsigned grid = (unsigned)ceil_div(n, kBlock);
    k_keys_from_f64<<<grid, kBlock, 0, st>>>(keys, n, ka);
    PP_LAUNCH_CHECK();
    int rc = sort_pairs<uint64_t>(ka, nullptr, kb, perm, n, 0, 64, scratch, sb, st);
    if (rc != PP_OK) return rc;
    k_widen_u32<<<grid, kBlock, 0, st>>>(perm, n, perm_out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// CSR/CSC pointer array of an index vector that is already sorted: ptr[v] = #entries < v, v in [0, num_rows]
int pp_ptr_from_sorted_i64(const int64_t* sorted, int64_t n, int64_t num_rows, int64_t* ptr, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n >= 0 && num_rows >= 0, PP_ERR_ARG, "pp_ptr_from_sorted_i64: negative size");
    k_ptr_from_sorted_i64<<<(unsigned)ceil_div(n + 1, kBlock), kBlock, 0, st>>>(sorted, n, num_rows, ptr);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
