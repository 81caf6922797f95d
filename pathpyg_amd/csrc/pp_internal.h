// pathpyg_amd — declarations shared between the .hip translation units (host side only).
#pragma once
#include "pp_common.h"

namespace pp {

// pp_scan.hip
// Share (per mille) of the resident workgroup slots a PERSISTENT grid of the calling thread may take (pp_set_launch_share): the kernels
// that size their grid to "exactly what is resident at once" leave the rest of the slots to kernels of another stream.
int launch_share_permille();
static inline int64_t shared_grid(int64_t resident) {
    const int64_t g = resident * launch_share_permille() / 1000;
    return g < 8 ? 8 : g / 8 * 8;
}
size_t scan_ws_bytes(int64_t n);
template <typename InT, typename OutT>
int exclusive_scan(const InT* in, int64_t n, OutT* out, bool with_total, int64_t* total_dev, void* ws, size_t ws_bytes,
                   hipStream_t st, int64_t* slot_owner = nullptr, int64_t slot_stride = 1, int64_t slot_cap = 0);
// several int32 -> int32 exclusive scans (out[j] has n[j] + 1 entries; nullptr: reduction only) in one launch triple
constexpr int kScanMaxJobs = 4;
struct ScanJob {
    const int32_t* in;
    int32_t* out;
    int64_t n, tile0;
    int64_t *total, *tile_sum;
};
struct ScanJobs {
    ScanJob job[kScanMaxJobs];
    int count;
};
size_t scan_multi_ws_bytes(const int64_t* n, int count);
int exclusive_scan_multi(const int32_t* const* in, const int64_t* n, int32_t* const* out, int64_t* const* total_dev, int count, void* ws,
                         size_t ws_bytes, hipStream_t st);
template <typename IdxT>
int histogram(const IdxT* idx, int64_t n, int64_t nbins, int32_t* bins, hipStream_t st);
int minmax_i64(const int64_t* a, int64_t n, int64_t* out2, hipStream_t st);

// pp_sort.hip
size_t sort_ws_bytes(int64_t n, int key_bytes);
// Stable LSD radix sort of (key, value) pairs on key bits [begin_bit, end_bit).
// vals_in == nullptr means "values are 0..n-1".
template <typename KeyT>
int sort_pairs(const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out, int64_t n, int begin_bit,
               int end_bit, void* ws, size_t ws_bytes, hipStream_t st);

// pp_lift.hip: views into the workspace of a finished pp_temporal_count (m events, num_nodes nodes)
struct TemporalLists {
    const uint32_t* ids;         // [m] event ids grouped by tail node, ascending (= time order) inside a node's list
    const uint32_t* rowptr;      // [num_nodes + 1] list bounds
    const uint32_t* first_pos;   // [m] per event: position in `ids` of its first continuation
    const int32_t* count;        // [m] per event: number of continuations (consecutive entries of `ids`)
    const int64_t* result;       // {E2, status}
    size_t total_bytes;
};
TemporalLists temporal_lists(void* ws, int64_t m, int64_t num_nodes);

// pp_dbgnn.hip
int weight_grad_reduce(const float* partial_w, const float* partial_b, int64_t n_parts, int M, int K, float* dW, float* db, hipStream_t st);

}  // namespace pp
