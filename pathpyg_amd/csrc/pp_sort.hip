// pathpyg_amd — stable LSD radix sort of (key, value) pairs for gfx950.
//
// Replaces, on the hot path of the reference,
//   torch.argsort(time)                         src/pathpyG/core/temporal_graph.py:58
//   torch.unique(node_sequence, dim=0)          src/pathpyG/algorithms/lift_order.py:133   (row sort)
//   torch_geometric.utils.coalesce (index_sort) src/pathpyG/algorithms/lift_order.py:139   (edge-key sort)
//   EdgeIndex.sort_by("row") / get_csc          src/pathpyG/core/graph.py:103,115
//
// One pass = 8 key bits, three launches, no inter-workgroup spinning:
//   k_digit_hist    each 256-thread workgroup counts the digits of its 4096-key tile in LDS
//   k_scan_digit_rows  one workgroup per digit scans its row of the digit-major table [256][tiles] (+ 256 row totals)
//   k_scatter       re-reads the tile; every wave ranks its 1024 consecutive keys with 8 __ballot
//                   rounds per key (wave-wide multi-split, no atomics), the tile is reordered by digit
//                   THROUGH LDS, and written out so that consecutive lanes store consecutive addresses
//                   of one digit run (coalesced 4/8-byte stores per run).
// Stability: order inside a tile is (wave, round, lane) == ascending index; tiles are ordered by the scan.
// HBM traffic per pass: read K (hist) + read K+4 + write K+4 bytes per pair (K = key bytes).
#include "pp_internal.h"

namespace pp {

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;             // 256 digits == kBlock threads
constexpr int kSortItems = 16;                      // keys per lane
constexpr int kSortTile = kBlock * kSortItems;      // 4096 keys per workgroup
constexpr int kWaveSpan = kWave * kSortItems;       // 1024 consecutive keys per wave
constexpr int kXcds = 8;                            // accelerator dies of an MI355X (workgroup i runs on die i % 8)

static_assert(kRadix == kBlock, "one thread per digit in the table steps");

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT k, int shift, unsigned mask) {
    return (unsigned)(k >> shift) & mask;
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_digit_hist(const KeyT* __restrict__ keys, int64_t n, int shift, unsigned mask,
                                                      uint32_t* __restrict__ table, int64_t ntiles) {
    // one private histogram per wave (LDS atomics of different waves never meet), 16-byte key loads; summed at the end
    __shared__ unsigned bins[kWavesPerBlock][kRadix];
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) bins[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t tile_base = (int64_t)blockIdx.x * kSortTile;
    unsigned* mine = bins[wave_id()];
    constexpr int kVec = 16 / sizeof(KeyT);                   // keys per 16-byte load
    struct alignas(16) Chunk { KeyT k[kVec]; };
    if (tile_base + kSortTile <= n && ((uintptr_t)keys & 15) == 0) {
#pragma unroll
        for (int r = 0; r < kSortItems / kVec; ++r) {
            const Chunk c = *reinterpret_cast<const Chunk*>(keys + tile_base + ((int64_t)r * kBlock + threadIdx.x) * kVec);
#pragma unroll
            for (int e = 0; e < kVec; ++e) atomicAdd(&mine[digit_of(c.k[e], shift, mask)], 1u);
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            int64_t i = tile_base + (int64_t)r * kBlock + threadIdx.x;
            if (i < n) atomicAdd(&mine[digit_of(keys[i], shift, mask)], 1u);
        }
    }
    __syncthreads();
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) total += bins[w][threadIdx.x];
    table[(int64_t)threadIdx.x * ntiles + blockIdx.x] = total;
}

// One workgroup per digit: exclusive scan of that digit's row of the table (its counts over the tiles, a few thousand entries) in
// place, row total to totals[digit].  k_scatter adds the exclusive scan of the 256 totals itself, so a pass needs ONE scan launch
// instead of the three of a generic scan over the flattened [256][tiles] table.
__global__ __launch_bounds__(kBlock) void k_scan_digit_rows(uint32_t* __restrict__ table, int64_t ntiles, uint32_t* __restrict__ totals) {
    __shared__ unsigned s_scratch[kWavesPerBlock + 1];
    uint32_t* row = table + (int64_t)blockIdx.x * ntiles;
    const int64_t per_thread = (ntiles + kBlock - 1) / kBlock;
    const int64_t begin = (int64_t)threadIdx.x * per_thread;
    const int64_t end = begin + per_thread < ntiles ? begin + per_thread : ntiles;
    unsigned sum = 0;
    for (int64_t i = begin; i < end; ++i) sum += row[i];
    unsigned total;
    unsigned run = block_exclusive_sum<unsigned>(sum, s_scratch, &total);
    for (int64_t i = begin; i < end; ++i) {
        const unsigned c = row[i];
        row[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

template <typename KeyT, bool kIota>
__global__ __launch_bounds__(kBlock) void k_scatter(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                   KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n,
                                                   int shift, unsigned mask, const uint32_t* __restrict__ table,
                                                   const uint32_t* __restrict__ totals, int64_t ntiles) {
    __shared__ KeyT s_keys[kSortTile];
    __shared__ uint32_t s_vals[kSortTile];
    __shared__ unsigned s_wcnt[kWavesPerBlock][kRadix];   // per-wave digit counts -> per-wave digit bases
    __shared__ unsigned s_dstart[kRadix];                 // start of digit d inside the reordered tile
    __shared__ unsigned s_gofs[kRadix];                   // global position of reordered slot j of digit d = s_gofs[d] + j
    __shared__ unsigned s_scratch[kWavesPerBlock + 1];

    const int lane = lane_id();
    const int wave = wave_id();
    // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2.  Tile t and tile t+1 extend the same 256 digit runs,
    // i.e. they write the two halves of the same cache lines: give every XCD a CONTIGUOUS range of tiles so that those partial-line
    // writes meet in one L2 instead of reaching the memory side as masked writes from two.
    const int64_t per_xcd = (ntiles + kXcds - 1) / kXcds;
    const int64_t tile = (int64_t)(blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
    if (tile >= ntiles) return;
    const int64_t tile_base = tile * kSortTile;
    const int64_t wave_base = tile_base + (int64_t)wave * kWaveSpan;

#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) s_wcnt[w][threadIdx.x] = 0;

    KeyT key[kSortItems];
    uint32_t val[kSortItems];
    unsigned local[kSortItems];        // rank among the wave's earlier keys with the same digit
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        int64_t i = wave_base + r * kWave + lane;
        bool live = i < n;
        key[r] = live ? keys_in[i] : (KeyT)0;
        val[r] = kIota ? (uint32_t)i : (live ? vals_in[i] : 0u);
    }
    __syncthreads();

    volatile unsigned* wcnt = s_wcnt[wave];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        int64_t i = wave_base + r * kWave + lane;
        bool live = i < n;
        unsigned d = digit_of(key[r], shift, mask);
        uint64_t same = __ballot(live);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            uint64_t vote = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? vote : ~vote;
        }
        unsigned rank = (unsigned)__popcll(same & lanemask_lt());
        unsigned before = live ? wcnt[d] : 0u;
        if (live && rank == 0) wcnt[d] = before + (unsigned)__popcll(same);
        local[r] = before + rank;
    }
    __syncthreads();

    // one thread per digit: per-wave bases, digit totals, tile-wide digit starts, global offsets
    {
        const int d = threadIdx.x;
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            unsigned c = s_wcnt[w][d];
            s_wcnt[w][d] = run;
            run += c;
        }
        unsigned total;
        unsigned start = block_exclusive_sum<unsigned>(run, s_scratch, &total);
        unsigned digit_base = block_exclusive_sum<unsigned>(totals[d], s_scratch, &total);   // keys with a smaller digit, all tiles
        s_dstart[d] = start;
        s_gofs[d] = digit_base + table[(int64_t)d * ntiles + tile] - start;   // modular arithmetic on purpose
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        int64_t i = wave_base + r * kWave + lane;
        if (i < n) {
            unsigned d = digit_of(key[r], shift, mask);
            unsigned slot = s_dstart[d] + s_wcnt[wave][d] + local[r];
            s_keys[slot] = key[r];
            s_vals[slot] = val[r];
        }
    }
    __syncthreads();

    const int64_t left = n - tile_base;
    const unsigned count = left < kSortTile ? (unsigned)left : (unsigned)kSortTile;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        unsigned j = r * kBlock + threadIdx.x;
        if (j < count) {
            KeyT k = s_keys[j];
            unsigned pos = s_gofs[digit_of(k, shift, mask)] + j;
            keys_out[pos] = k;
            vals_out[pos] = s_vals[j];
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_iota_u32(uint32_t* out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

static inline int64_t sort_tiles(int64_t n) { return ceil_div(n > 0 ? n : 1, kSortTile); }

size_t sort_ws_bytes(int64_t n, int key_bytes) {
    const int64_t nn = n > 0 ? n : 1;
    size_t table = (size_t)kRadix * (size_t)sort_tiles(nn);
    return align_up((size_t)nn * (size_t)key_bytes) + align_up((size_t)nn * 4) + align_up(table * 4) + scan_ws_bytes((int64_t)table) + 1024;
}

template <typename KeyT>
int sort_pairs(const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out, int64_t n, int begin_bit,
               int end_bit, void* ws, size_t ws_bytes, hipStream_t st) {
    PP_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "sort_pairs: n=%lld outside [0, 2^31)", (long long)n);
    PP_REQUIRE(begin_bit >= 0 && end_bit >= begin_bit && end_bit <= (int)sizeof(KeyT) * 8, PP_ERR_ARG, "sort_pairs: bad bit range [%d,%d)",
               begin_bit, end_bit);
    PP_REQUIRE(ws_bytes >= sort_ws_bytes(n, sizeof(KeyT)), PP_ERR_WORKSPACE, "sort_pairs: workspace too small");
    if (n == 0) return PP_OK;
    const int64_t ntiles = sort_tiles(n);
    Arena a(ws, ws_bytes);
    KeyT* tmp_keys = a.take<KeyT>(n);
    uint32_t* tmp_vals = a.take<uint32_t>(n);
    uint32_t* table = a.take<uint32_t>((int64_t)kRadix * ntiles);
    uint32_t* totals = a.take<uint32_t>(kRadix);       // sort_ws_bytes reserves scan_ws_bytes(...) + 1024 >= 1 KiB behind the table

    const int passes = (end_bit - begin_bit + kRadixBits - 1) / kRadixBits;
    if (passes == 0) {   // nothing to order on: identity permutation
        if (keys_out && keys_out != keys_in) PP_HIP(hipMemcpyAsync(keys_out, keys_in, (size_t)n * sizeof(KeyT), hipMemcpyDeviceToDevice, st));
        if (vals_in) {
            if (vals_out != vals_in) PP_HIP(hipMemcpyAsync(vals_out, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        } else {
            k_iota_u32<<<(unsigned)ceil_div(n, kBlock), kBlock, 0, st>>>(vals_out, n);
            PP_LAUNCH_CHECK();
        }
        return PP_OK;
    }
    const bool in_place = ((const void*)keys_in == (const void*)keys_out) || ((const void*)vals_in == (const void*)vals_out);
    PP_REQUIRE(!in_place, PP_ERR_ARG, "sort_pairs: input and output buffers must differ");
    PP_REQUIRE(keys_out != nullptr && vals_out != nullptr, PP_ERR_ARG, "sort_pairs: null output buffer");

    const KeyT* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    const unsigned scatter_grid = (unsigned)(ceil_div(ntiles, kXcds) * kXcds);
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + p * kRadixBits;
        const int nbits = (end_bit - shift) < kRadixBits ? (end_bit - shift) : kRadixBits;
        const unsigned mask = (1u << nbits) - 1u;
        const bool to_out = ((passes - 1 - p) % 2) == 0;      // ping-pong so that the last pass lands in `out`
        KeyT* dst_k = to_out ? keys_out : tmp_keys;
        uint32_t* dst_v = to_out ? vals_out : tmp_vals;

        k_digit_hist<KeyT><<<(unsigned)ntiles, kBlock, 0, st>>>(src_k, n, shift, mask, table, ntiles);
        PP_LAUNCH_CHECK();
        k_scan_digit_rows<<<kRadix, kBlock, 0, st>>>(table, ntiles, totals);
        PP_LAUNCH_CHECK();
        if (p == 0 && src_v == nullptr) {
            k_scatter<KeyT, true><<<scatter_grid, kBlock, 0, st>>>(src_k, nullptr, dst_k, dst_v, n, shift, mask, table, totals, ntiles);
        } else {
            k_scatter<KeyT, false><<<scatter_grid, kBlock, 0, st>>>(src_k, src_v, dst_k, dst_v, n, shift, mask, table, totals, ntiles);
        }
        PP_LAUNCH_CHECK();
        src_k = dst_k;
        src_v = dst_v;
    }
    return PP_OK;
}

template int sort_pairs<uint32_t>(const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, int64_t, int, int, void*, size_t, hipStream_t);
template int sort_pairs<uint64_t>(const uint64_t*, const uint32_t*, uint64_t*, uint32_t*, int64_t, int, int, void*, size_t, hipStream_t);

}  // namespace pp

extern "C" {

size_t pp_sort_ws_bytes(int64_t n, int key_bytes) { return pp::sort_ws_bytes(n, key_bytes); }

int pp_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, int64_t n,
                      int begin_bit, int end_bit, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp::sort_pairs<uint32_t>(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, ws, ws_bytes, (hipStream_t)stream);
}

int pp_sort_pairs_u64(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                      int begin_bit, int end_bit, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp::sort_pairs<uint64_t>(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
