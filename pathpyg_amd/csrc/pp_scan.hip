// pathpyg_amd — prefix sums, histograms and small reductions (gfx950).
//
// These replace the torch/PyG calls the reference's lifts are built from:
//   torch_geometric.utils.cumsum / degree   (reference src/pathpyG/algorithms/lift_order.py:65,74,77)
// All of them are pure HBM streams: 16-byte loads per lane, wave shuffles for the
// intra-wave step, LDS only for the 4 per-wave partials of a 256-thread workgroup.
#include "pp_internal.h"

#include <stdarg.h>

namespace pp {

static thread_local char g_err[512] = "";
static thread_local int g_launch_share = 1000;
int launch_share_permille() { return g_launch_share; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

size_t scan_ws_bytes(int64_t n);
constexpr int kScanItems = 8;                       // consecutive items per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 items per workgroup

// 8 consecutive items of one lane: 16-byte loads when the array is 16-byte aligned (always, for torch allocations) and the
// lane's chunk lies fully inside the array; scalar loads otherwise.
template <typename InT>
__device__ __forceinline__ void load_items(const InT* __restrict__ in, int64_t base, int64_t n, bool aligned, int64_t (&v)[8]) {
    if (aligned && base + 8 <= n) {
        constexpr int kVec = 16 / sizeof(InT);               // items per 16-byte load
        struct alignas(16) Chunk { InT x[kVec]; };
#pragma unroll
        for (int c = 0; c < 8 / kVec; ++c) {
            const Chunk ch = *reinterpret_cast<const Chunk*>(in + base + c * kVec);
#pragma unroll
            for (int e = 0; e < kVec; ++e) v[c * kVec + e] = (int64_t)ch.x[e];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = base + k < n ? (int64_t)in[base + k] : 0;
    }
}

// ---- pass 1: one partial sum per tile ---------------------------------------------------------
template <typename InT>
__global__ __launch_bounds__(kBlock) void k_tile_sums(const InT* __restrict__ in, int64_t n, int64_t* __restrict__ tile_sum) {
    __shared__ int64_t part[kWavesPerBlock];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    load_items<InT>(in, base, n, ((uintptr_t)in & 15) == 0, v);
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    s = wave_sum(s);
    if (lane_id() == 0) part[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) t += part[w];
        tile_sum[blockIdx.x] = t;
    }
}

// ---- pass 2: exclusive scan of the tile sums, in place, by ONE workgroup ------------------------
__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(int64_t* __restrict__ tile_sum, int64_t ntiles, int64_t* __restrict__ total_out) {
    __shared__ int64_t scratch[kWavesPerBlock + 1];
    // every thread owns one contiguous run of tile sums: two sequential sweeps over it around ONE workgroup scan (a loop of
    // workgroup scans over 256-entry slices costs two barriers per slice: 19 us for the 4883 tiles of a 10^7-item scan)
    const int64_t per_thread = (ntiles + kBlock - 1) / kBlock;
    const int64_t begin = (int64_t)threadIdx.x * per_thread;
    const int64_t end = begin + per_thread < ntiles ? begin + per_thread : ntiles;
    int64_t sum = 0;
    for (int64_t i = begin; i < end; ++i) sum += tile_sum[i];
    int64_t tot;
    int64_t run = block_exclusive_sum(sum, scratch, &tot);
    for (int64_t i = begin; i < end; ++i) {
        const int64_t c = tile_sum[i];
        tile_sum[i] = run;
        run += c;
    }
    if (threadIdx.x == 0 && total_out) *total_out = tot;
}

// ---- pass 3: rescan every tile with its base ------------------------------------------------------
// `slot_owner` (the lift fills): out[] are the output offsets of n sources; slot_owner[b] = the source that owns output slot
// b * slot_stride (the last source s with out[s] <= b * slot_stride), for every b <= slot_cap whose slot exists, and
// slot_owner[ceil(total / slot_stride)] = n - 1.  The expansion kernel reads its tile's first / last source from it: the scan has every
// (offset, count) pair in registers anyway, a separate kernel would bisect the finished offsets once per tile.
template <typename InT, typename OutT>
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const InT* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_base,
                                                      OutT* __restrict__ out, int write_total, int64_t* __restrict__ slot_owner,
                                                      int64_t slot_stride, int64_t slot_cap) {
    __shared__ int64_t scratch[kWavesPerBlock + 1];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    load_items<InT>(in, base, n, ((uintptr_t)in & 15) == 0, v);
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    int64_t tot;
    int64_t run = tile_base[blockIdx.x] + block_exclusive_sum(s, scratch, &tot);
    if (slot_owner) {
        int64_t at = run;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            const int64_t i = base + k;
            if (i < n) {
                const int64_t end = at + v[k];
                for (int64_t b = (at + slot_stride - 1) / slot_stride; b * slot_stride < end && b <= slot_cap; ++b) slot_owner[b] = i;
                if (i == n - 1) {
                    const int64_t b = (end + slot_stride - 1) / slot_stride;
                    if (b <= slot_cap) slot_owner[b] = i;
                }
                at = end;
            }
        }
    }
    if (base + kScanItems < n && ((uintptr_t)out & 15) == 0) {     // whole chunk inside the array (and not its last item): 16-byte stores
        constexpr int kVec = 16 / sizeof(OutT);
        struct alignas(16) Chunk { OutT x[kVec]; };
#pragma unroll
        for (int c = 0; c < kScanItems / kVec; ++c) {
            Chunk ch;
#pragma unroll
            for (int e = 0; e < kVec; ++e) {
                ch.x[e] = (OutT)run;
                run += v[c * kVec + e];
            }
            *reinterpret_cast<Chunk*>(out + base + c * kVec) = ch;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        int64_t i = base + k;
        if (i < n) out[i] = (OutT)run;
        run += v[k];
        if (write_total && i == n - 1) out[n] = (OutT)run;   // out has n+1 entries
    }
}

// ---- several int32 -> int32 scans in ONE launch triple (round 5: the order-2 builder ends its count pass with four scans back to back; at the
// sizes of a partition rank every launch costs its floor).  A job with `out == nullptr` is a reduction only (its total).
__global__ __launch_bounds__(kBlock) void k_tile_sums_multi(ScanJobs js) {
    __shared__ int64_t part[kWavesPerBlock];
    int j = 0;
    while (j + 1 < js.count && (int64_t)blockIdx.x >= js.job[j + 1].tile0) ++j;
    const ScanJob& job = js.job[j];
    const int64_t tile = (int64_t)blockIdx.x - job.tile0;
    const int64_t base = tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    load_items<int32_t>(job.in, base, job.n, ((uintptr_t)job.in & 15) == 0, v);
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    s = wave_sum(s);
    if (lane_id() == 0) part[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) t += part[w];
        job.tile_sum[tile] = t;
    }
}

__global__ __launch_bounds__(kBlock) void k_scan_tile_sums_multi(ScanJobs js) {
    __shared__ int64_t scratch[kWavesPerBlock + 1];
    const ScanJob& job = js.job[blockIdx.x];
    const int64_t ntiles = (job.n + kScanTile - 1) / kScanTile;
    const int64_t per_thread = (ntiles + kBlock - 1) / kBlock;
    const int64_t begin = (int64_t)threadIdx.x * per_thread;
    const int64_t end = begin + per_thread < ntiles ? begin + per_thread : ntiles;
    int64_t sum = 0;
    for (int64_t i = begin; i < end; ++i) sum += job.tile_sum[i];
    int64_t tot;
    int64_t run = block_exclusive_sum(sum, scratch, &tot);
    for (int64_t i = begin; i < end; ++i) {
        const int64_t c = job.tile_sum[i];
        job.tile_sum[i] = run;
        run += c;
    }
    if (threadIdx.x == 0 && job.total) *job.total = tot;
}

__global__ __launch_bounds__(kBlock) void k_scan_tiles_multi(ScanJobs js) {
    __shared__ int64_t scratch[kWavesPerBlock + 1];
    int j = 0;
    while (j + 1 < js.count && (int64_t)blockIdx.x >= js.job[j + 1].tile0) ++j;
    const ScanJob& job = js.job[j];
    if (job.out == nullptr) return;                        // (reduction only; uniform per workgroup)
    const int64_t tile = (int64_t)blockIdx.x - job.tile0;
    const int64_t n = job.n;
    const int64_t base = tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    load_items<int32_t>(job.in, base, n, ((uintptr_t)job.in & 15) == 0, v);
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    int64_t tot;
    int64_t run = job.tile_sum[tile] + block_exclusive_sum(s, scratch, &tot);
    int32_t* out = job.out;
    if (base + kScanItems < n && ((uintptr_t)out & 15) == 0) {
        struct alignas(16) Chunk { int32_t x[4]; };
#pragma unroll
        for (int c = 0; c < kScanItems / 4; ++c) {
            Chunk ch;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ch.x[e] = (int32_t)run;
                run += v[c * 4 + e];
            }
            *reinterpret_cast<Chunk*>(out + base + c * 4) = ch;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < n) out[i] = (int32_t)run;
        run += v[k];
        if (i == n - 1) out[n] = (int32_t)run;             // out has n + 1 entries
    }
}

size_t scan_multi_ws_bytes(const int64_t* n, int count) {
    size_t total = 0;
    for (int j = 0; j < count; ++j) total += scan_ws_bytes(n[j]);
    return total;
}

// out[j][i] = sum(in[j][0..i)), i in [0, n_j]; total_dev[j] optional (int64); out[j] == nullptr: only the total.  Empty jobs are written here.
int exclusive_scan_multi(const int32_t* const* in, const int64_t* n, int32_t* const* out, int64_t* const* total_dev, int count, void* ws,
                         size_t ws_bytes, hipStream_t st) {
    PP_REQUIRE(count >= 1 && count <= kScanMaxJobs, PP_ERR_ARG, "exclusive_scan_multi: 1..%d jobs", kScanMaxJobs);
    ScanJobs js{};
    Arena a(ws, ws_bytes);
    int64_t tiles = 0;
    for (int j = 0; j < count; ++j) {
        PP_REQUIRE(n[j] >= 0, PP_ERR_ARG, "exclusive_scan_multi: negative length");
        if (n[j] == 0) {
            if (out[j]) PP_HIP(hipMemsetAsync(out[j], 0, sizeof(int32_t), st));
            if (total_dev[j]) PP_HIP(hipMemsetAsync(total_dev[j], 0, sizeof(int64_t), st));
            continue;
        }
        ScanJob& job = js.job[js.count++];
        job.in = in[j]; job.out = out[j]; job.n = n[j]; job.total = total_dev[j];
        job.tile0 = tiles;
        const int64_t nt = ceil_div(n[j], kScanTile);
        job.tile_sum = a.take<int64_t>(nt + 1);
        tiles += nt;
    }
    PP_REQUIRE(a.ok(), PP_ERR_WORKSPACE, "exclusive_scan_multi: workspace too small");
    if (js.count == 0) return PP_OK;
    k_tile_sums_multi<<<(unsigned)tiles, kBlock, 0, st>>>(js);
    PP_LAUNCH_CHECK();
    k_scan_tile_sums_multi<<<(unsigned)js.count, kBlock, 0, st>>>(js);
    PP_LAUNCH_CHECK();
    k_scan_tiles_multi<<<(unsigned)tiles, kBlock, 0, st>>>(js);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

size_t scan_ws_bytes(int64_t n) { return align_up((size_t)(ceil_div(n > 0 ? n : 1, kScanTile) + 1) * sizeof(int64_t)); }

// out[i] = sum(in[0..i)), i in [0,n]; out[n] (= total) is written when `with_total`; *total_dev optional.
template <typename InT, typename OutT>
int exclusive_scan(const InT* in, int64_t n, OutT* out, bool with_total, int64_t* total_dev, void* ws, size_t ws_bytes,
                   hipStream_t st, int64_t* slot_owner, int64_t slot_stride, int64_t slot_cap) {
    PP_REQUIRE(n >= 0, PP_ERR_ARG, "exclusive_scan: negative length");
    PP_REQUIRE(ws_bytes >= scan_ws_bytes(n), PP_ERR_WORKSPACE, "exclusive_scan: workspace too small");
    int64_t* tile_sum = (int64_t*)ws;
    if (n == 0) {
        if (slot_owner) PP_HIP(hipMemsetAsync(slot_owner, 0, sizeof(int64_t), st));
        if (with_total) PP_HIP(hipMemsetAsync(out, 0, sizeof(OutT), st));
        if (total_dev) PP_HIP(hipMemsetAsync(total_dev, 0, sizeof(int64_t), st));
        return PP_OK;
    }
    const int64_t ntiles = ceil_div(n, kScanTile);
    k_tile_sums<InT><<<(unsigned)ntiles, kBlock, 0, st>>>(in, n, tile_sum);
    PP_LAUNCH_CHECK();
    k_scan_tile_sums<<<1, kBlock, 0, st>>>(tile_sum, ntiles, total_dev);
    PP_LAUNCH_CHECK();
    k_scan_tiles<InT, OutT><<<(unsigned)ntiles, kBlock, 0, st>>>(in, n, tile_sum, out, with_total ? 1 : 0, slot_owner, slot_stride, slot_cap);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template int exclusive_scan<int32_t, int64_t>(const int32_t*, int64_t, int64_t*, bool, int64_t*, void*, size_t, hipStream_t, int64_t*, int64_t, int64_t);
template int exclusive_scan<int32_t, int32_t>(const int32_t*, int64_t, int32_t*, bool, int64_t*, void*, size_t, hipStream_t, int64_t*, int64_t, int64_t);
template int exclusive_scan<int64_t, int64_t>(const int64_t*, int64_t, int64_t*, bool, int64_t*, void*, size_t, hipStream_t, int64_t*, int64_t, int64_t);
template int exclusive_scan<uint32_t, uint32_t>(const uint32_t*, int64_t, uint32_t*, bool, int64_t*, void*, size_t, hipStream_t, int64_t*, int64_t, int64_t);

// ---- histogram of an index vector (PyG degree) ---------------------------------------------------
// Runs of equal values inside a wave (the common case: the line-graph lift's input is source-sorted)
// are folded into ONE atomic by the run's first lane; unsorted input degrades to one atomic per lane.
// HOT bins (round 6): a scale-free stream sends a tenth of its 2 * 10^7 events to one node — 2 * 10^6 atomics on one address took 21 ms.
// Every workgroup keeps the first kHotSlots distinct values it meets in an LDS table (claimed by compare-and-swap, exact match: a value whose
// slot belongs to another one goes to memory as before) and adds the table to the bins at the end: a hot bin gets one atomic per workgroup.
constexpr int kHotSlots = 512;
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void k_histogram(const IdxT* __restrict__ idx, int64_t n, int64_t nbins, int32_t* __restrict__ bins) {
    __shared__ int s_hot[kHotSlots], s_count[kHotSlots];
    for (int e = threadIdx.x; e < kHotSlots; e += kBlock) { s_hot[e] = -1; s_count[e] = 0; }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += stride) {   // wave-uniform trip count
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        int64_t v = live ? (int64_t)idx[i] : -1;
        int64_t prev = __shfl_up(v, 1, kWave);
        bool head = live && (lane_id() == 0 || prev != v);
        uint64_t heads = __ballot(head);
        uint64_t lives = __ballot(live);
        if (head) {
            // run length = distance to the next head (or to the end of the live lanes)
            uint64_t later = heads & ~((2ull << lane_id()) - 1ull);
            int next = later ? __ffsll((long long)later) - 1 : (int)__popcll(lives);
            int len = next - lane_id();
            if (v >= 0 && v < nbins) {
                if (v < (int64_t)0x7fffffff) {
                    const int slot = (int)(((uint32_t)v * 2654435761u) >> 23);                 // 9 bits
                    const int owner = atomicCAS(&s_hot[slot], -1, (int)v);
                    if (owner == -1 || owner == (int)v) atomicAdd(&s_count[slot], len);
                    else atomicAdd(&bins[v], len);
                } else {
                    atomicAdd(&bins[v], len);
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kHotSlots; e += kBlock)
        if (s_count[e] > 0) atomicAdd(&bins[s_hot[e]], s_count[e]);
}

template <typename IdxT>
int histogram(const IdxT* idx, int64_t n, int64_t nbins, int32_t* bins, hipStream_t st) {
    PP_HIP(hipMemsetAsync(bins, 0, (size_t)(nbins > 0 ? nbins : 0) * sizeof(int32_t), st));
    if (n == 0 || nbins == 0) return PP_OK;
    int64_t g = ceil_div(n, kBlock);
    if (g > kMaxGrid * 4) g = kMaxGrid * 4;
    k_histogram<IdxT><<<(unsigned)g, kBlock, 0, st>>>(idx, n, nbins, bins);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
template int histogram<int64_t>(const int64_t*, int64_t, int64_t, int32_t*, hipStream_t);
template int histogram<uint32_t>(const uint32_t*, int64_t, int64_t, int32_t*, hipStream_t);

// ---- min / max of an int64 vector (index validation, radix pass count) ------------------------------
__global__ __launch_bounds__(kBlock) void k_minmax_i64(const int64_t* __restrict__ a, int64_t n, int64_t* __restrict__ out /*[2]: min,max*/) {
    __shared__ int64_t smin[kWavesPerBlock], smax[kWavesPerBlock];
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int64_t v = a[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    hi = wave_max(hi);
    lo = wave_min(lo);
    if (lane_id() == 0) { smin[wave_id()] = lo; smax[wave_id()] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
            lo = smin[w] < lo ? smin[w] : lo;
            hi = smax[w] > hi ? smax[w] : hi;
        }
        atomicMin((long long*)&out[0], (long long)lo);
        atomicMax((long long*)&out[1], (long long)hi);
    }
}

__global__ void k_init_minmax(int64_t* out) { out[0] = INT64_MAX; out[1] = INT64_MIN; }

int minmax_i64(const int64_t* a, int64_t n, int64_t* out2, hipStream_t st) {
    k_init_minmax<<<1, 1, 0, st>>>(out2);
    PP_LAUNCH_CHECK();
    if (n == 0) return PP_OK;
    int64_t g = ceil_div(n, kBlock * 4);
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    k_minmax_i64<<<(unsigned)g, kBlock, 0, st>>>(a, n, out2);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp

// =================================================================== C ABI
extern "C" {

int pp_version(void) { return PP_VERSION; }
const char* pp_last_error(void) { return pp::g_err; }
int pp_set_launch_share(int per_mille) {
    const int before = pp::g_launch_share;
    pp::g_launch_share = per_mille < 1 ? 1 : (per_mille > 1000 ? 1000 : per_mille);
    return before;
}

size_t pp_scan_ws_bytes(int64_t n) { return pp::scan_ws_bytes(n); }

int pp_exclusive_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp::exclusive_scan<int32_t, int64_t>(in, n, out, true, nullptr, ws, ws_bytes, (hipStream_t)stream, nullptr, 1, 0);
}
int pp_exclusive_scan_i64(const int64_t* in, int64_t n, int64_t* out, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp::exclusive_scan<int64_t, int64_t>(in, n, out, true, nullptr, ws, ws_bytes, (hipStream_t)stream, nullptr, 1, 0);
}
int pp_degree_i64(const int64_t* index, int64_t n, int64_t num_bins, int32_t* bins, pp_stream_t stream) {
    return pp::histogram<int64_t>(index, n, num_bins, bins, (hipStream_t)stream);
}
int pp_minmax_i64(const int64_t* a, int64_t n, int64_t* out2, pp_stream_t stream) {
    return pp::minmax_i64(a, n, out2, (hipStream_t)stream);
}

}  // extern "C"
