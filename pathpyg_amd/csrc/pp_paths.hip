// pathpyg_amd — consumers of the temporal event graph: all-pairs shortest time-respecting paths.
//
// Reference: temporal_shortest_paths, src/pathpyG/algorithms/temporal.py:57-107 — builds the event DAG with lift_order_temporal,
// adds a virtual source / sink per first-order node and runs scipy's Dijkstra (unit weights) from every source.  With unit
// weights that is a breadth-first search: here one workgroup per source node runs a level-synchronous frontier BFS straight on
// the CSR of the lifted event graph (events are the vertices, no augmented graph is materialised):
//   level 1 = the events that START at the source;   level L+1 = unvisited successors of level L;
//   dist[s, v] = first level with an event that ENDS in v;   pred[s, v] = source node of the LATEST such event (largest event id
//   among the tight ones: deterministic, reproduces the reference's known answer; scipy's heap order differs on <1 % of ties).
#include "pp_common.h"
#include "pp_internal.h"

namespace pp {

__global__ __launch_bounds__(kBlock) void k_temporal_bfs(const int64_t* __restrict__ edge_index, int64_t m, int64_t n,
                                                        const int64_t* __restrict__ succ_ptr, const int64_t* __restrict__ succ,
                                                        const int64_t* __restrict__ by_src_ptr, const int64_t* __restrict__ by_src,
                                                        int32_t* __restrict__ dist, int64_t* __restrict__ pred, int32_t* __restrict__ ws) {
    __shared__ int s_count[2];
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + m;
    int32_t* level = ws + (int64_t)blockIdx.x * 3 * m;
    int32_t* queue[2] = {level + m, level + 2 * m};
    for (int64_t s = blockIdx.x; s < n; s += gridDim.x) {
        int32_t* drow = dist + s * n;
        int64_t* prow = pred + s * n;
        for (int64_t e = threadIdx.x; e < m; e += kBlock) level[e] = -1;
        for (int64_t v = threadIdx.x; v < n; v += kBlock) { drow[v] = -1; prow[v] = 0; }        // pred holds (event id + 1) tags until the end
        if (threadIdx.x == 0) { s_count[0] = (int)(by_src_ptr[s + 1] - by_src_ptr[s]); s_count[1] = 0; }
        __syncthreads();
        const int64_t first = by_src_ptr[s];
        for (int k = threadIdx.x; k < s_count[0]; k += kBlock) {
            const int32_t e = (int32_t)by_src[first + k];
            queue[0][k] = e;
            level[e] = 1;
        }
        __syncthreads();
        int cur = 0;
        for (int depth = 1; s_count[cur] > 0; ++depth) {
            const int count = s_count[cur];
            const int32_t* q = queue[cur];
            int32_t* qn = queue[cur ^ 1];
            // (1) nodes first reached at this depth
            for (int k = threadIdx.x; k < count; k += kBlock) {
                const int64_t v = dst[q[k]];
                if (drow[v] < 0) drow[v] = depth;            // every writer of this pass stores the same value
            }
            __syncthreads();
            // (2) latest tight event per node, (3) expand
            for (int k = threadIdx.x; k < count; k += kBlock) {
                const int32_t e = q[k];
                const int64_t v = dst[e];
                if (drow[v] == depth) atomicMax((unsigned long long*)&prow[v], (unsigned long long)(e + 1));
                for (int64_t p = succ_ptr[e]; p < succ_ptr[e + 1]; ++p) {
                    const int32_t f = (int32_t)succ[p];
                    if (atomicCAS(&level[f], -1, depth + 1) == -1) qn[atomicAdd(&s_count[cur ^ 1], 1)] = f;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_count[cur] = 0;
            cur ^= 1;
            __syncthreads();
        }
        // event id -> source node of that event; diagonal as the reference sets it
        for (int64_t v = threadIdx.x; v < n; v += kBlock) {
            const int64_t tag = prow[v];                     // 0: none; e + 1 otherwise
            prow[v] = v == s ? s : (tag > 0 ? src[tag - 1] : -1);
            if (v == s) drow[v] = 0;
        }
        __syncthreads();
    }
}

}  // namespace pp

extern "C" {

static int64_t bfs_blocks(int64_t m, int64_t n) {
    const int64_t budget = (int64_t)1 << 30;                              // <= 1 GiB of per-workgroup BFS state
    int64_t blocks = budget / (12 * (m > 0 ? m : 1));
    if (blocks > n) blocks = n;
    if (blocks > 2048) blocks = 2048;
    return blocks < 1 ? 1 : blocks;
}

size_t pp_temporal_bfs_ws_bytes(int64_t m, int64_t n) { return pp::align_up((size_t)bfs_blocks(m, n) * 3 * (size_t)(m > 0 ? m : 1) * sizeof(int32_t)); }

int pp_temporal_bfs(const int64_t* edge_index, int64_t m, int64_t n, const int64_t* succ_ptr, const int64_t* succ, const int64_t* by_src_ptr,
                    const int64_t* by_src, int32_t* dist, int64_t* pred, void* ws, size_t ws_bytes, pp_stream_t stream) {
    PP_REQUIRE(m >= 0 && n >= 0, PP_ERR_ARG, "pp_temporal_bfs: negative size");
    PP_REQUIRE(m < (int64_t)0x7fffffff && n < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_temporal_bfs: size >= 2^31");
    PP_REQUIRE(ws_bytes >= pp_temporal_bfs_ws_bytes(m, n), PP_ERR_WORKSPACE, "pp_temporal_bfs: workspace too small");
    if (n == 0) return PP_OK;
    pp::k_temporal_bfs<<<(unsigned)bfs_blocks(m, n), pp::kBlock, 0, (hipStream_t)stream>>>(edge_index, m, n, succ_ptr, succ, by_src_ptr, by_src, dist,
                                                                                           pred, (int32_t*)ws);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
