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

// ---------------------------------------------------------------------------------------------------------------------
// Temporal betweenness centrality (reference temporal_betweenness_centrality, src/pathpyG/algorithms/centrality.py:164-297:
// Brandes' algorithm on the event DAG with a virtual source per first-order node, pure-Python dict/deque loops).
// Level-synchronous form, one workgroup per source node:
//   forward : BFS levels over the events; sigma[e] = number of shortest event paths from the source (integer-valued doubles:
//             the atomic adds are exact in any order); dist_fo / sigma_fo per first-order node from the TIGHT events into it;
//   backward: levels in reverse; every event pulls x * dep[w] from its successors on the next level (x = sigma[v] / sigma[w]) in
//             CSR order, adds its own target term, and is credited dep[w] * x towards its head node;
//   per node: credits summed over the node's in-events in event order (no floating-point atomics anywhere);
//   source  : + sum of dep over its first-level events - (number of reached nodes) + 1, as centrality.py:274-278,295 nets out.
struct BetweennessWs {
    int32_t* level;
    int32_t* order;
    int32_t* level_start;
    double* sigma;
    double* dep;
    double* credit;
    int32_t* dist_fo;
    double* sigma_fo;
};

__host__ __device__ static inline size_t betweenness_block_bytes(int64_t m, int64_t n) {
    const size_t me = (size_t)(m > 0 ? m : 1), ne = (size_t)(n > 0 ? n : 1);
    return align_up(me * 4) + align_up(me * 4) + align_up((me + 2) * 4) + 3 * align_up(me * 8) + align_up(ne * 4) + align_up(ne * 8);
}

__device__ static inline BetweennessWs carve_betweenness(char* base, int64_t m, int64_t n) {
    const size_t me = (size_t)(m > 0 ? m : 1), ne = (size_t)(n > 0 ? n : 1);
    BetweennessWs w;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    w.level = (int32_t*)base; base += up(me * 4);
    w.order = (int32_t*)base; base += up(me * 4);
    w.level_start = (int32_t*)base; base += up((me + 2) * 4);
    w.sigma = (double*)base; base += up(me * 8);
    w.dep = (double*)base; base += up(me * 8);
    w.credit = (double*)base; base += up(me * 8);
    w.dist_fo = (int32_t*)base; base += up(ne * 4);
    w.sigma_fo = (double*)base;
    return w;
}

__global__ __launch_bounds__(kBlock) void k_temporal_betweenness(const int64_t* __restrict__ edge_index, int64_t m, int64_t n,
                                                                const int64_t* __restrict__ succ_ptr, const int64_t* __restrict__ succ,
                                                                const int64_t* __restrict__ by_src_ptr, const int64_t* __restrict__ by_src,
                                                                const int64_t* __restrict__ by_dst_ptr, const int64_t* __restrict__ by_dst,
                                                                double* __restrict__ bw_blocks, char* __restrict__ ws, size_t block_bytes) {
    __shared__ int s_tail;
    __shared__ int s_reached;
    const int64_t* dst = edge_index + m;
    const BetweennessWs w = carve_betweenness(ws + (size_t)blockIdx.x * block_bytes, m, n);
    double* bw = bw_blocks + (int64_t)blockIdx.x * n;
    for (int64_t v = threadIdx.x; v < n; v += kBlock) bw[v] = 0.0;
    for (int64_t s = blockIdx.x; s < n; s += gridDim.x) {
        const int64_t first = by_src_ptr[s];
        const int n_first = (int)(by_src_ptr[s + 1] - first);
        if (n_first == 0) continue;                                   // centrality.py:210: only nodes with out-going events are sources
        for (int64_t e = threadIdx.x; e < m; e += kBlock) { w.level[e] = -1; w.sigma[e] = 0.0; w.credit[e] = 0.0; }
        for (int64_t v = threadIdx.x; v < n; v += kBlock) { w.dist_fo[v] = -1; w.sigma_fo[v] = 0.0; }
        if (threadIdx.x == 0) { s_tail = n_first; s_reached = 0; }
        __syncthreads();
        if (threadIdx.x == 0) { w.dist_fo[s] = 0; w.sigma_fo[s] = 1.0; w.level_start[1] = 0; }
        for (int k = threadIdx.x; k < n_first; k += kBlock) {
            const int32_t e = (int32_t)by_src[first + k];
            w.order[k] = e;
            w.level[e] = 1;
            w.sigma[e] = 1.0;
        }
        __syncthreads();
        int begin = 0, depth = 1;
        while (true) {
            const int end = s_tail;
            if (begin == end) break;
            if (threadIdx.x == 0) w.level_start[depth + 1] = end;
            for (int k = begin + threadIdx.x; k < end; k += kBlock) {                       // nodes first reached at this depth
                const int64_t v = dst[w.order[k]];
                if (w.dist_fo[v] < 0) w.dist_fo[v] = depth;
            }
            __syncthreads();
            for (int k = begin + threadIdx.x; k < end; k += kBlock) {                       // tight events; claim the next level
                const int32_t e = w.order[k];
                const int64_t v = dst[e];
                if (w.dist_fo[v] == depth) atomicAdd(&w.sigma_fo[v], w.sigma[e]);
                for (int64_t p = succ_ptr[e]; p < succ_ptr[e + 1]; ++p) {
                    const int32_t f = (int32_t)succ[p];
                    if (atomicCAS(&w.level[f], -1, depth + 1) == -1) w.order[atomicAdd(&s_tail, 1)] = f;
                }
            }
            __syncthreads();
            for (int k = begin + threadIdx.x; k < end; k += kBlock) {                       // path counts of the next level
                const int32_t e = w.order[k];
                const double se = w.sigma[e];
                for (int64_t p = succ_ptr[e]; p < succ_ptr[e + 1]; ++p) {
                    const int32_t f = (int32_t)succ[p];
                    if (w.level[f] == depth + 1) atomicAdd(&w.sigma[f], se);
                }
            }
            __syncthreads();
            begin = end;
            ++depth;
        }
        for (int d = depth - 1; d >= 1; --d) {                                             // dependencies, deepest level first
            for (int k = w.level_start[d] + threadIdx.x; k < w.level_start[d + 1]; k += kBlock) {
                const int32_t v = w.order[k];
                const double sv = w.sigma[v];
                double acc = 0.0, cr = 0.0;
                for (int64_t p = succ_ptr[v]; p < succ_ptr[v + 1]; ++p) {
                    const int32_t f = (int32_t)succ[p];
                    if (w.level[f] == d + 1) {
                        const double x = sv / w.sigma[f];
                        acc += x * w.dep[f];
                        cr += w.dep[f] * x;
                    }
                }
                const int64_t head = dst[v];
                if (w.dist_fo[head] == d) acc += sv / w.sigma_fo[head];
                w.dep[v] = acc;
                w.credit[v] = cr;
            }
            __syncthreads();
        }
        int reached = 0;
        for (int64_t v = threadIdx.x; v < n; v += kBlock) {                                 // credits per node, in event order
            double t = 0.0;
            for (int64_t p = by_dst_ptr[v]; p < by_dst_ptr[v + 1]; ++p) t += w.credit[by_dst[p]];
            bw[v] += t;
            reached += w.dist_fo[v] >= 0;
        }
        atomicAdd(&s_reached, reached);
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < n_first; ++k) t += w.dep[by_src[first + k]];
            bw[s] += t - (double)s_reached + 1.0;
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

static int64_t betweenness_blocks(int64_t m, int64_t n) {
    const int64_t budget = (int64_t)1 << 30;
    int64_t blocks = budget / (int64_t)pp::betweenness_block_bytes(m, n);
    if (blocks > n) blocks = n;
    if (blocks > 1024) blocks = 1024;
    return blocks < 1 ? 1 : blocks;
}

int64_t pp_temporal_betweenness_parts(int64_t m, int64_t n) { return betweenness_blocks(m, n); }

size_t pp_temporal_betweenness_ws_bytes(int64_t m, int64_t n) { return (size_t)betweenness_blocks(m, n) * pp::betweenness_block_bytes(m, n); }

int pp_temporal_betweenness(const int64_t* edge_index, int64_t m, int64_t n, const int64_t* succ_ptr, const int64_t* succ,
                            const int64_t* by_src_ptr, const int64_t* by_src, const int64_t* by_dst_ptr, const int64_t* by_dst,
                            double* partial, void* ws, size_t ws_bytes, pp_stream_t stream) {
    PP_REQUIRE(m >= 0 && n >= 0, PP_ERR_ARG, "pp_temporal_betweenness: negative size");
    PP_REQUIRE(m < (int64_t)0x7fffffff && n < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_temporal_betweenness: size >= 2^31");
    PP_REQUIRE(ws_bytes >= pp_temporal_betweenness_ws_bytes(m, n), PP_ERR_WORKSPACE, "pp_temporal_betweenness: workspace too small");
    if (n == 0) return PP_OK;
    pp::k_temporal_betweenness<<<(unsigned)betweenness_blocks(m, n), pp::kBlock, 0, (hipStream_t)stream>>>(
        edge_index, m, n, succ_ptr, succ, by_src_ptr, by_src, by_dst_ptr, by_dst, partial, (char*)ws, pp::betweenness_block_bytes(m, n));
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
