// pathpyg_amd — DBGNN message passing on gfx950: GCN normalisation plans, CSR segment-reduce SpMM
// (forward and transposed), bipartite projection, fused bias/ELU epilogue and ELU backward.
//
// Reference code replaced (paths relative to the pathpyG repository root):
//   DBGNN.forward                       src/pathpyG/nn/dbgnn.py:121-151
//   BipartiteGraphOperator.forward      src/pathpyG/nn/dbgnn.py:50-69
// and the torch_geometric 2.7.0 pieces those call (not in the reference tree):
//   gcn_norm / add_remaining_self_loops, GCNConv.propagate (aggr="add"), MessagePassing("add").
//
// Design: every propagation is   Y[r,:] = act( sum_p val[p] * X[idx[p],:] + self[r] * S[r,:] + bias )
// over a CSR whose rows are the DESTINATIONS (forward) or the SOURCES (backward = transposed graph), so
// no atomics are needed and the accumulation order is fixed.  A row is owned by LPR = F/4 lanes, each
// holding one float4 of the feature row (F=64: 16 lanes/row, 4 rows per wave; F=256: a whole wave), the
// gather X[idx[p],:] is one 16-byte load per lane = full 64..1024-byte row segments per edge; bias add and
// ELU are fused into the store.  The normalisation (self loops, weighted in-degree, d^-1/2, per-edge
// coefficient) is computed ONCE per graph into a "plan" (the reference recomputes it every forward).
// The dense X*W^T products stay plain library GEMMs (MFMA through rocBLAS) on the Python side.
#include "pp_internal.h"

namespace pp {

// ------------------------------------------------------------------ plan construction
__global__ __launch_bounds__(kBlock) void k_index_key(const int64_t* __restrict__ index, int64_t n, int64_t limit, uint32_t* __restrict__ keys,
                                                     int64_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int64_t v = index[i];
    if (v < 0 || v >= limit) { atomicOr((unsigned long long*)status, 1ull); v = 0; }
    keys[i] = (uint32_t)v;
}

__global__ __launch_bounds__(kBlock) void k_gcn_coefficients(const int64_t* __restrict__ edge_index, int64_t n_edges, const float* __restrict__ w,
                                                            const uint32_t* __restrict__ order, const float* __restrict__ dinv,
                                                            int by_dst, int32_t* __restrict__ idx_out, float* __restrict__ val_out) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_edges) return;
    const int64_t e = order ? (int64_t)order[p] : p;
    const int64_t r = edge_index[e], c = edge_index[n_edges + e];
    idx_out[p] = (int32_t)(by_dst ? r : c);
    val_out[p] = r == c ? 0.0f : dinv[r] * (w ? w[e] : 1.0f) * dinv[c];
}

// pp_gcn_plan's one pass over the edge list: both endpoints validated, (source, weight) packed, the destination written as the sort key
// of the forward grouping and the last self loop of every node recorded (one read of the edge list instead of three)
__global__ __launch_bounds__(kBlock) void k_plan_edges(const int64_t* __restrict__ edge_index, int64_t n_edges, int64_t n_nodes, int64_t n_dst,
                                                      const float* __restrict__ w, uint2* __restrict__ packed, uint32_t* __restrict__ dst_keys,
                                                      int32_t* __restrict__ last_loop, int32_t* __restrict__ src_ptr,
                                                      int64_t* __restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_edges) return;
    int64_t r = edge_index[e], c = edge_index[n_edges + e];
    if (src_ptr) {          // row-sorted edge list: the source-major row pointer falls out of the same read (rows between two sources)
        int64_t a = e == 0 ? -1 : edge_index[e - 1], b = r;
        if (a < -1) a = -1;
        if (b > n_nodes) b = n_nodes;
        for (int64_t v = a + 1; v <= b; ++v) src_ptr[v] = (int32_t)e;
        if (e == n_edges - 1) {
            a = r < -1 ? -1 : r;
            for (int64_t v = a + 1; v <= n_nodes; ++v) src_ptr[v] = (int32_t)n_edges;
        }
    }
    const bool bad_r = r < 0 || r >= n_nodes, bad_c = c < 0 || c >= n_dst;       // n_nodes = source rows (owned + halo), n_dst = owned rows
    if (bad_r || bad_c) atomicOr((unsigned long long*)status, 1ull);
    if (!bad_r && r == c) atomicMax(&last_loop[r], (int32_t)e);
    if (bad_r) r = 0;
    if (bad_c) c = 0;
    packed[e] = make_uint2((uint32_t)r, __float_as_uint(w ? w[e] : 1.0f));
    dst_keys[e] = (uint32_t)c;
}

// destination grouping in ONE pass over the sorted keys: in_ptr (gap fill between neighbouring keys), in_idx[p] = source of the p-th incoming
// edge, w_by_dst[p] = its weight, dst_order[p] = its edge id (three launches and a second read of the sorted keys before)
__global__ __launch_bounds__(kBlock) void k_group_by_dst(const uint2* __restrict__ packed, int64_t n_edges, const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ sorted_keys, int64_t n_dst, int32_t* __restrict__ in_ptr,
                                                        int32_t* __restrict__ in_idx, float* __restrict__ w_by_dst, int32_t* __restrict__ dst_order) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n_edges) return;
    {
        const int64_t a = p == 0 ? -1 : (int64_t)sorted_keys[p - 1];
        int64_t b = p == n_edges ? n_dst : (int64_t)sorted_keys[p];
        if (b > n_dst) b = n_dst;
        for (int64_t v = a + 1; v <= b; ++v) in_ptr[v] = (int32_t)p;
    }
    if (p == n_edges) return;
    const uint32_t e = order[p];
    const uint2 sw = packed[e];
    in_idx[p] = (int32_t)sw.x;
    w_by_dst[p] = __uint_as_float(sw.y);
    if (dst_order) dst_order[p] = (int32_t)e;
}

// weighted in-degree from the destination-grouped (contiguous) weights; existing self loops are replaced by ONE loop
__global__ __launch_bounds__(kBlock) void k_gcn_degree_grouped(const int32_t* __restrict__ in_idx, const float* __restrict__ w_by_dst,
                                                              const uint32_t* __restrict__ dst_ptr, const int32_t* __restrict__ last_loop,
                                                              const float* __restrict__ w, int64_t n_nodes, float* __restrict__ dinv,
                                                              float* __restrict__ self_coef) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n_nodes;
    const uint32_t p0 = live ? dst_ptr[i] : 0u, p1 = live ? dst_ptr[i + 1] : 0u;
    const bool is_long = p1 - p0 > 256u;                         // hub: summed by the whole wave below, not by this lane alone
    float deg = 0.0f;
    if (!is_long) {
        // eight entries per round trip (an entry-by-entry walk pays one dependent load latency per entry: the longest row of the wave
        // sets the pace); the sum stays strictly left to right
        constexpr int kChunk = 8;
        for (uint32_t p = p0; p < p1; p += kChunk) {
            int32_t jj[kChunk];
            float ww[kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const bool in = p + k < p1;
                jj[k] = in ? in_idx[p + k] : (int32_t)i;
                ww[k] = in ? w_by_dst[p + k] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                if (jj[k] != (int32_t)i) deg += ww[k];
        }
    }
    // hub rows: the wave strides over the row with 8 independent loads in flight per lane (a lone lane needs ~1 us per entry: 3 ms for
    // a 10^5-entry row); waves work on different hubs in parallel, partial sums folded by a fixed-order butterfly
    for (uint64_t todo = __ballot(is_long); todo != 0; todo &= todo - 1) {
        const int owner = __ffsll((long long)todo) - 1;
        const uint32_t b = __shfl(p0, owner, kWave), e = __shfl(p1, owner, kWave);
        const int32_t node = (int32_t)__shfl((int)i, owner, kWave);
        float part = 0.0f;
        constexpr int kUnroll = 8;
        uint32_t p = b + lane_id();
        for (; p + (kUnroll - 1) * kWave < e; p += kUnroll * kWave) {
            int32_t jj[kUnroll];
            float ww[kUnroll];
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) { jj[k] = in_idx[p + k * kWave]; ww[k] = w_by_dst[p + k * kWave]; }
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) part += jj[k] != node ? ww[k] : 0.0f;
        }
        for (; p < e; p += kWave)
            if (in_idx[p] != node) part += w_by_dst[p];
        part = wave_sum(part);                                   // butterfly: fixed order, the same total in every lane
        if (lane_id() == owner) deg = part;
    }
    if (!live) return;
    const float lw = last_loop[i] >= 0 ? (w ? w[last_loop[i]] : 1.0f) : 1.0f;
    deg += lw;
    float d = 1.0f / sqrtf(deg);                                  // deg^-1/2 ; inf -> 0 like masked_fill_(== inf, 0)
    if (isinf(d)) d = 0.0f;
    dinv[i] = d;
    self_coef[i] = d * lw * d;
}

// in place: weight of the p-th incoming edge -> its normalised coefficient
__global__ __launch_bounds__(kBlock) void k_in_coefficients(const int32_t* __restrict__ in_idx, const uint32_t* __restrict__ dst_of,
                                                           const float* __restrict__ dinv, int64_t n_edges, float* __restrict__ val) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_edges) return;
    const int32_t r = in_idx[p];
    const uint32_t c = dst_of[p];
    val[p] = (uint32_t)r == c ? 0.0f : dinv[r] * val[p] * dinv[c];
}

__global__ __launch_bounds__(kBlock) void k_ptr_from_sorted_i64_i32(const int64_t* __restrict__ sorted, int64_t n, int64_t num_rows,
                                                                   int32_t* __restrict__ ptr) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    int64_t a = p == 0 ? -1 : sorted[p - 1];
    int64_t b = p == n ? num_rows : sorted[p];
    if (a < -1) a = -1;
    if (b > num_rows) b = num_rows;
    for (int64_t v = a + 1; v <= b; ++v) ptr[v] = (int32_t)p;
}

__global__ __launch_bounds__(kBlock) void k_u32_to_i32_ptr(const uint32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

__global__ __launch_bounds__(kBlock) void k_gather_index(const int64_t* __restrict__ values, const uint32_t* __restrict__ order, int64_t n,
                                                        int32_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < n) out[p] = (int32_t)values[order ? order[p] : p];
}

__global__ __launch_bounds__(kBlock) void k_gather_f32(const float* __restrict__ values, const uint32_t* __restrict__ order, int64_t n,
                                                      float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < n) out[p] = values[order ? order[p] : p];
}

__global__ __launch_bounds__(kBlock) void k_ptr_diff_f32(const int32_t* __restrict__ ptr, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (float)(ptr[i + 1] - ptr[i]);
}

// rowptr[v] = first position p with sorted_keys[p] >= v (same gap-fill as pp_lift.hip, int32 output)
__global__ __launch_bounds__(kBlock) void k_ptr_from_sorted_u32(const uint32_t* __restrict__ sorted_keys, int64_t n, int64_t num_rows,
                                                               uint32_t* __restrict__ ptr) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    int64_t a = p == 0 ? -1 : (int64_t)sorted_keys[p - 1];
    int64_t b = p == n ? num_rows : (int64_t)sorted_keys[p];
    if (b > num_rows) b = num_rows;
    for (int64_t v = a + 1; v <= b; ++v) ptr[v] = (uint32_t)p;
}

// ------------------------------------------------------------------ SpMM (segment reduce over CSR rows)

// kLanes lanes own one row (each lane one float4 of a column block of kLanes*4 columns); every lane group walks kRows
// consecutive rows TOGETHER so that their index loads, and then their row gathers, are in flight at the same time.
// The kLanes lanes of a row fetch up to kLanes (index, value) pairs with ONE coalesced load each and hand them round by
// shuffle: no row gather ever waits for an index load of its own row (measured: 2.12 -> 1.77 ms on the 10^7-row graph).
template <int kLanes, int kRows, bool kFull>       // kFull: F == kLanes * 4, every lane owns a live column block
__global__ __launch_bounds__(kBlock) void k_spmm_v4(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
                                                   int64_t n_rows, const float* __restrict__ X, int F, const float* __restrict__ self_coef,
                                                   const float* __restrict__ S, const float* __restrict__ bias, int act, HeavyRows heavy,
                                                   float* __restrict__ Y) {
    constexpr int kGroups = kBlock / kLanes;
    const int64_t r0 = ((int64_t)blockIdx.x * kGroups + threadIdx.x / kLanes) * kRows;
    const int lane = threadIdx.x % kLanes;
    if (r0 >= n_rows) return;
    int p0[kRows], p1[kRows];
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const bool live = r0 + q < n_rows;
        p0[q] = live ? ptr[r0 + q] : 0;
        p1[q] = live ? ptr[r0 + q + 1] : 0;
    }
    // kLanes * 4 >= F: one float4 column block per lane, the loop below runs exactly once.  Lanes beyond the feature width (F / 4
    // not a power of two) still take part in the index chunk loads and shuffles - only their row loads and stores are switched off.
    for (int c0 = lane * 4; c0 < kLanes * 4; c0 += kLanes * 4) {
        const bool col_live = kFull || c0 < F;
        float4 acc[kRows];
        int my_j[kRows];
        float my_v[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) {                  // first chunk of every row: issued back to back
            acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (heavy.slot != nullptr && r0 + q < n_rows) {               // a hub row: its neighbour sum is already in heavy.sum
                const int hs = heavy.slot[r0 + q];
                if (hs >= 0) {
                    p1[q] = p0[q];
                    if (col_live) acc[q] = *(const float4*)(heavy.sum + (int64_t)hs * F + c0);
                }
            }
            const int mine = p0[q] + lane;
            my_j[q] = mine < p1[q] ? idx[mine] : 0;
            my_v[q] = mine < p1[q] ? (val ? val[mine] : 1.f) : 0.f;
        }
        float4 self_row[kRows];
        float self_c[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const bool live = self_coef != nullptr && r0 + q < n_rows;
            self_c[q] = live ? self_coef[r0 + q] : 0.f;
            self_row[q] = (live && col_live) ? *(const float4*)(S + (r0 + q) * F + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            for (int base = p0[q]; base < p1[q]; base += kLanes) {
                if (base != p0[q]) {                       // rows with more than kLanes entries: next chunk
                    const int mine = base + lane;
                    my_j[q] = mine < p1[q] ? idx[mine] : 0;
                    my_v[q] = mine < p1[q] ? (val ? val[mine] : 1.f) : 0.f;
                }
                const int cnt = p1[q] - base < kLanes ? p1[q] - base : kLanes;
                for (int e = 0; e < cnt; e += 4) {
                    float4 x[4];
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src_lane = (e + u) < cnt ? e + u : e;          // clamp: harmless re-read with weight 0
                        const int j = __shfl(my_j[q], src_lane, kLanes);
                        v[u] = (e + u) < cnt ? __shfl(my_v[q], src_lane, kLanes) : 0.f;
                        x[u] = col_live ? *(const float4*)(X + (int64_t)j * F + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[q].x += v[u] * x[u].x; acc[q].y += v[u] * x[u].y; acc[q].z += v[u] * x[u].z; acc[q].w += v[u] * x[u].w;
                    }
                }
            }
        }
        if (!col_live) continue;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b = *(const float4*)(bias + c0);
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            if (r0 + q >= n_rows) break;
            float4 o = acc[q];
            o.x += self_c[q] * self_row[q].x + b.x; o.y += self_c[q] * self_row[q].y + b.y;
            o.z += self_c[q] * self_row[q].z + b.z; o.w += self_c[q] * self_row[q].w + b.w;
            if (act) o = elu_fast4(o);
            *(float4*)(Y + (r0 + q) * F + c0) = o;
        }
    }
}

template <int kLanes>   // scalar-column variant for feature widths that are not multiples of 4
__global__ __launch_bounds__(kBlock) void k_spmm_s1(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
                                                   int64_t n_rows, const float* __restrict__ X, int F, const float* __restrict__ self_coef,
                                                   const float* __restrict__ S, const float* __restrict__ bias, int act, HeavyRows heavy,
                                                   float* __restrict__ Y) {
    constexpr int kRowsPerBlock = kBlock / kLanes;
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x / kLanes;
    const int lane = threadIdx.x % kLanes;
    if (r >= n_rows) return;
    const int hs = heavy.slot != nullptr ? heavy.slot[r] : -1;
    const int p0 = ptr[r], p1 = hs >= 0 ? p0 : ptr[r + 1];
    for (int c = lane; c < F; c += kLanes) {
        float acc = hs >= 0 ? heavy.sum[(int64_t)hs * F + c] : 0.f;
        for (int p = p0; p < p1; ++p) acc += (val ? val[p] : 1.f) * X[(int64_t)idx[p] * F + c];
        if (self_coef) acc += self_coef[r] * S[r * F + c];
        if (bias) acc += bias[c];
        Y[r * F + c] = act ? elu_fast(acc) : acc;
    }
}

// ------------------------------------------------------------------ ELU backward (+ column sums for the bias gradient)
// dpre = dY * elu'(pre) with elu'(pre) = 1 for y > 0 and y + 1 otherwise (y = elu(pre)); act == 0: dpre = dY.
// kFixedColumn: the grid stride is a multiple of F, so a thread always meets the same column and keeps
// its partial bias gradient in a register (one LDS atomic per thread at the end instead of one per element).
template <bool kFixedColumn>
__global__ __launch_bounds__(kBlock) void k_act_backward(const float* __restrict__ dY, const float* __restrict__ Y, int64_t n_rows, int F, int act,
                                                        float* __restrict__ dpre, float* __restrict__ dbias) {
    extern __shared__ float s_col[];                      // [F] partial column sums of this workgroup
    for (int c = threadIdx.x; c < F; c += kBlock) s_col[c] = 0.f;
    __syncthreads();
    const int64_t total = n_rows * (int64_t)F;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float mine = 0.f;
    for (int64_t i = first; i < total; i += stride) {
        float g = dY[i];
        if (act) { const float y = Y[i]; g *= (y > 0.f ? 1.f : y + 1.f); }
        if (dpre) dpre[i] = g;
        if (dbias) {
            if (kFixedColumn) mine += g;
            else atomicAdd(&s_col[(int)(i % F)], g);
        }
    }
    if (dbias && kFixedColumn && first < total) atomicAdd(&s_col[(int)(first % F)], mine);
    __syncthreads();
    if (dbias)
        for (int c = threadIdx.x; c < F; c += kBlock) atomicAdd(&dbias[c], s_col[c]);
}

// ------------------------------------------------------------------ bipartite combine (reference nn/dbgnn.py:66-69,143-144)
// y = ELU(a + deg[r] * (p + b)) on the first-order rows: a = lin1 applied to the summed higher-order rows, p = lin2(x) per first-order row,
// b = lin1's bias (it enters once per incoming pair: deg[r] times), deg = in-degree of the bipartite graph.  One pass instead of the
// add / addcmul / elu chain of element-wise library kernels; the backward pass below replaces elu_backward / mul / column-sum kernels.
__global__ __launch_bounds__(kBlock) void k_bip_combine(const float* __restrict__ A, const float* __restrict__ P, const float* __restrict__ deg,
                                                       const float* __restrict__ bias, int64_t n_rows, int F, float* __restrict__ Y) {
    const int64_t total = n_rows * (int64_t)F;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        Y[i] = elu_fast(A[i] + deg[r] * (P[i] + (bias ? bias[c] : 0.f)));
    }
}

// dpre = dY * ELU'(y); dA = dpre; dP = deg[r] * dpre; dbias[c] = sum_r dP[r][c]
template <bool kFixedColumn>
__global__ __launch_bounds__(kBlock) void k_bip_combine_backward(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ deg,
                                                                int64_t n_rows, int F, float* __restrict__ dA, float* __restrict__ dP,
                                                                float* __restrict__ dbias) {
    extern __shared__ float s_col[];
    for (int c = threadIdx.x; c < F; c += kBlock) s_col[c] = 0.f;
    __syncthreads();
    const int64_t total = n_rows * (int64_t)F;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float mine = 0.f;
    for (int64_t i = first; i < total; i += stride) {
        const int64_t r = i / F;
        const float y = Y[i];
        const float g = dY[i] * (y > 0.f ? 1.f : y + 1.f);
        const float gp = deg[r] * g;
        dA[i] = g;
        dP[i] = gp;
        if (dbias) {
            if (kFixedColumn) mine += gp;
            else atomicAdd(&s_col[(int)(i - r * F)], gp);
        }
    }
    if (dbias && kFixedColumn && first < total) atomicAdd(&s_col[(int)(first % F)], mine);
    __syncthreads();
    if (dbias)
        for (int c = threadIdx.x; c < F; c += kBlock) atomicAdd(&dbias[c], s_col[c]);
}

// ------------------------------------------------------------------ dropout with counter-based masks
// out = x * keep / (1 - p)      (x may alias out)
__global__ __launch_bounds__(kBlock) void k_dropout(const float* __restrict__ X, int64_t n_rows, int F, uint32_t key, uint32_t threshold, float scale,
                                                   int64_t row0, const int64_t* __restrict__ rows, float* __restrict__ out) {
    const int64_t total = n_rows * (int64_t)F;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        const int64_t gr = rows ? rows[r] : row0 + r;
        out[i] = dropout_keep(gr, c, F, key, threshold) ? X[i] * scale : 0.f;
    }
}

// dpre = dY * keep / (1 - p) * (act ? ELU'(y) : 1) with y = Ydrop * (1 - p) where kept; column sums of dpre (optional)
template <bool kFixedColumn>
__global__ __launch_bounds__(kBlock) void k_dropout_act_backward(const float* __restrict__ dY, const float* __restrict__ Ydrop, int64_t n_rows, int F,
                                                                uint32_t key, uint32_t threshold, float scale, float keep_prob, int64_t row0,
                                                                const int64_t* __restrict__ rows, int act, float* __restrict__ dpre,
                                                                float* __restrict__ dbias) {
    extern __shared__ float s_col[];
    for (int c = threadIdx.x; c < F; c += kBlock) s_col[c] = 0.f;
    __syncthreads();
    const int64_t total = n_rows * (int64_t)F;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float mine = 0.f;
    for (int64_t i = first; i < total; i += stride) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        const int64_t gr = rows ? rows[r] : row0 + r;
        float g = 0.f;
        if (dropout_keep(gr, c, F, key, threshold)) {
            g = dY[i] * scale;
            if (act) { const float y = Ydrop[i] * keep_prob; g *= (y > 0.f ? 1.f : y + 1.f); }
        }
        dpre[i] = g;
        if (dbias) {
            if (kFixedColumn) mine += g;
            else atomicAdd(&s_col[c], g);
        }
    }
    if (dbias && kFixedColumn && first < total) atomicAdd(&s_col[(int)(first % F)], mine);
    __syncthreads();
    if (dbias)
        for (int c = threadIdx.x; c < F; c += kBlock) atomicAdd(&dbias[c], s_col[c]);
}

__global__ __launch_bounds__(kBlock) void k_scale_rows(const float* __restrict__ X, const float* __restrict__ coef, int64_t n_rows, int F,
                                                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_rows * (int64_t)F) out[i] = X[i] * coef[i / F];
}

// out[r] = own[r] + recv[slot[r]] (slot[r] >= 0) + extra[r] + self_coef[r] * dpre[r]   — every addend optional (NULL).
// The partitioned backward pass (pathpyg_amd/nn/sharded.py): the owner of a row folds the halo contributions its consumers sent back
// (`recv`, one row per sent row, `slot` = the inverse of the send list where every owned row has at most one consumer; `extra` = their CSR sum
// otherwise) and the self-loop term of the transposed aggregation into its own partial sum in ONE pass (index_add_ + addcmul otherwise).
// One float4 per thread; out may alias own.
__global__ __launch_bounds__(kBlock) void k_halo_fold(const float* own, const float* __restrict__ recv, const int32_t* __restrict__ slot,
                                                     const float* __restrict__ extra, const float* __restrict__ self_coef,
                                                     const float* __restrict__ dpre, int64_t n_rows, int q, float* out) {
    const int64_t total = n_rows * (int64_t)q;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t r = e / q;
        const int c = (int)(e - r * q);
        float4 v = *(const float4*)(own + 4 * e);
        if (slot != nullptr) {
            const int sl = slot[r];
            if (sl >= 0) {
                const float4 g = *(const float4*)(recv + ((int64_t)sl * q + c) * 4);
                v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
            }
        }
        if (extra != nullptr) {
            const float4 g = *(const float4*)(extra + 4 * e);
            v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
        }
        if (self_coef != nullptr) {
            const float sc = self_coef[r];
            const float4 g = *(const float4*)(dpre + 4 * e);
            v.x += sc * g.x; v.y += sc * g.y; v.z += sc * g.z; v.w += sc * g.w;
        }
        *(float4*)(out + 4 * e) = v;
    }
}

// dX[r] = (sum_e val[e] D[idx[e]]) (*) ELU'(Z[r]),  colsum[c] += dX[r][c]      (ELU' from the stored activation Z = ELU(pre))
// The transposed aggregation of a layer whose input was an activation, fused with that activation's backward and the bias
// gradient of the layer that produced it: one pass (gather D, read Z, write dX) instead of SpMM + a three-pass ELU backward.
// Persistent lane groups (kLanes = F/4 lanes per row, one float4 each) keep their column sums in registers; one LDS fold and
// F atomics per workgroup at the end.
// kDrop: Z is the DROPPED activation (dropout site `drop`): the mask, 1 / (1 - p) and ELU' at Z * (1 - p) go into dX.
#ifndef PP_NT_BIP
#define PP_NT_BIP 1
#endif
#ifndef PP_NT_WG
#define PP_NT_WG 0
#endif
template <int kLanes, bool kDrop = false>
__global__ __launch_bounds__(kBlock) void k_spmm_act_backward(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ val, int64_t n_rows, const float* __restrict__ D,
                                                             int F, const float* __restrict__ Z, float* __restrict__ colsum,
                                                             float* __restrict__ dX, DropSite drop) {
    constexpr int kGroups = kBlock / kLanes;
    __shared__ float s_col[kGroups][kLanes * 4 + 4];
    const int g = threadIdx.x / kLanes, l = threadIdx.x % kLanes;
    const bool col_live = 4 * l < F;
    const bool stream_rows = PP_NT_BIP && n_rows * (int64_t)F * 4 >= kStreamFromBytes;      // Z and dX: touched once, far larger than the caches
    float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = (int64_t)blockIdx.x * kGroups + g; r < n_rows; r += (int64_t)gridDim.x * kGroups) {
        if (!col_live) continue;
        const int p0 = ptr[r], p1 = ptr[r + 1];
        const float4 z = load_row_f4(Z + r * F + 4 * l, stream_rows);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = p0; e < p1; e += 2) {                 // two neighbour rows in flight (the mapping plans have one or two)
            const bool two = e + 1 < p1;
            const int j0 = idx[e], j1 = two ? idx[e + 1] : j0;
            const float v0 = val ? val[e] : 1.f, v1 = two ? (val ? val[e + 1] : 1.f) : 0.f;
            const float4 x0 = *(const float4*)(D + (int64_t)j0 * F + 4 * l);
            const float4 x1 = *(const float4*)(D + (int64_t)j1 * F + 4 * l);
            acc.x += v0 * x0.x + v1 * x1.x; acc.y += v0 * x0.y + v1 * x1.y;
            acc.z += v0 * x0.z + v1 * x1.z; acc.w += v0 * x0.w + v1 * x1.w;
        }
        float4 zz = z;
        if constexpr (kDrop) {
            const int64_t gr = drop.row0 + r;
            acc.x = dropout_keep(gr, 4 * l, F, drop.key, drop.thr) ? acc.x * drop.scale : 0.f;
            acc.y = dropout_keep(gr, 4 * l + 1, F, drop.key, drop.thr) ? acc.y * drop.scale : 0.f;
            acc.z = dropout_keep(gr, 4 * l + 2, F, drop.key, drop.thr) ? acc.z * drop.scale : 0.f;
            acc.w = dropout_keep(gr, 4 * l + 3, F, drop.key, drop.thr) ? acc.w * drop.scale : 0.f;
            zz = make_float4(z.x * drop.keep, z.y * drop.keep, z.z * drop.keep, z.w * drop.keep);
        }
        acc.x *= zz.x > 0.f ? 1.f : zz.x + 1.f; acc.y *= zz.y > 0.f ? 1.f : zz.y + 1.f;
        acc.z *= zz.z > 0.f ? 1.f : zz.z + 1.f; acc.w *= zz.w > 0.f ? 1.f : zz.w + 1.f;
        part.x += acc.x; part.y += acc.y; part.z += acc.z; part.w += acc.w;
        store_row_f4(dX + r * F + 4 * l, acc, stream_rows);
    }
    if (colsum == nullptr) return;
    *(float4*)&s_col[g][4 * l] = part;
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += kBlock) {
        float t = 0.f;
#pragma unroll 4
        for (int q = 0; q < kGroups; ++q) t += s_col[q][c];
        atomicAdd(&colsum[c], t);
    }
}

struct PlanWs {
    int64_t* status;       // [4]: {unused, status bits, longest destination row, longest source row}
    uint32_t* keys;        // [E]
    uint32_t* sorted;      // [E]
    uint32_t* order;       // [E]
    int32_t* last_loop;    // [N]
    float* dinv;           // [N]
    uint2* packed;         // [E] (source, weight bits) per edge
    void* scratch;
    size_t scratch_bytes;
    size_t total_bytes;
};

static PlanWs carve_plan(void* ws, int64_t e, int64_t n) {
    Arena a(ws, (size_t)-1);
    PlanWs w;
    w.status = a.take<int64_t>(4);
    w.keys = a.take<uint32_t>(e);
    w.sorted = a.take<uint32_t>(e);
    w.order = a.take<uint32_t>(e);
    w.last_loop = a.take<int32_t>(n);
    w.dinv = a.take<float>(n);
    w.packed = a.take<uint2>(e);
    w.scratch_bytes = sort_ws_bytes(e, 4);
    w.scratch = a.take<char>((int64_t)w.scratch_bytes);
    w.total_bytes = a.used;
    return w;
}

// group edge ids by `index` (stable): order[p] = edge id, ptr = CSR pointer over [0, n_groups]
static int group_by(const int64_t* index, int64_t e, int64_t n_groups, PlanWs& w, int32_t* ptr_out, hipStream_t st, bool keys_ready = false) {
    const unsigned grid = (unsigned)ceil_div(e > 0 ? e : 1, kBlock);
    if (e > 0) {
        if (!keys_ready) {
            k_index_key<<<grid, kBlock, 0, st>>>(index, e, n_groups, w.keys, w.status + 1);
            PP_LAUNCH_CHECK();
        }
        int rc = sort_pairs<uint32_t>(w.keys, nullptr, w.sorted, w.order, e, 0, bits_for((uint64_t)(n_groups > 0 ? n_groups - 1 : 0)),
                                      w.scratch, w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
    }
    k_ptr_from_sorted_u32<<<(unsigned)ceil_div(e + 1, kBlock), kBlock, 0, st>>>(w.sorted, e, n_groups, (uint32_t*)ptr_out);   // < 2^31: same bits
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ------------------------------------------------------------------ hub rows: chunked, bit-reproducible neighbour sums
// One workgroup per chunk of <= kHeavyChunk CSR entries of a heavy row: the kLanes-lane groups stride through the chunk, their
// partial rows are folded through LDS in group order; k_heavy_combine adds a row's chunks in chunk order.
constexpr int kHeavyChunk = 2048;

template <int kLanes>
__global__ __launch_bounds__(kBlock) void k_heavy_partial(const int32_t* __restrict__ idx, const float* __restrict__ val, const float* __restrict__ X,
                                                         int F, const int32_t* __restrict__ chunk_begin, const int32_t* __restrict__ chunk_end,
                                                         float* __restrict__ partial) {
    constexpr int kGroups = kBlock / kLanes;
    __shared__ __attribute__((aligned(16))) float s_part[kGroups][kLanes * 4];
    const int g = threadIdx.x / kLanes, l = threadIdx.x % kLanes;
    const int begin = chunk_begin[blockIdx.x], end = chunk_end[blockIdx.x];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * l < F) {
        for (int p = begin + g; p < end; p += 2 * kGroups) {            // two gathers in flight per lane group
            const int q = p + kGroups;
            const bool two = q < end;
            const int j0 = idx[p], j1 = two ? idx[q] : j0;
            const float v0 = val ? val[p] : 1.f, v1 = two ? (val ? val[q] : 1.f) : 0.f;
            const float4 x0 = *(const float4*)(X + (int64_t)j0 * F + 4 * l);
            const float4 x1 = *(const float4*)(X + (int64_t)j1 * F + 4 * l);
            acc.x += v0 * x0.x; acc.y += v0 * x0.y; acc.z += v0 * x0.z; acc.w += v0 * x0.w;
            acc.x += v1 * x1.x; acc.y += v1 * x1.y; acc.z += v1 * x1.z; acc.w += v1 * x1.w;
        }
    }
    *(float4*)&s_part[g][4 * l] = acc;
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += kBlock) {
        float t = 0.f;
        for (int q = 0; q < kGroups; ++q) t += s_part[q][c];
        partial[(int64_t)blockIdx.x * F + c] = t;
    }
}

__global__ __launch_bounds__(kBlock) void k_heavy_combine(const float* __restrict__ partial, const int32_t* __restrict__ heavy_chunk_ptr, int F,
                                                         float* __restrict__ heavy_sum) {
    const int c0 = heavy_chunk_ptr[blockIdx.x], c1 = heavy_chunk_ptr[blockIdx.x + 1];
    for (int c = threadIdx.x; c < F; c += kBlock) {
        float t = 0.f;
        for (int k = c0; k < c1; ++k) t += partial[(int64_t)k * F + c];
        heavy_sum[(int64_t)blockIdx.x * F + c] = t;
    }
}

__global__ __launch_bounds__(kBlock) void k_max_row_length(const int32_t* __restrict__ ptr, int64_t n_rows, int64_t* __restrict__ out) {
    __shared__ int s_max[kWavesPerBlock];
    int best = 0;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kBlock) {
        const int len = ptr[r + 1] - ptr[r];
        best = len > best ? len : best;
    }
    best = wave_max(best);
    if (lane_id() == 0) s_max[wave_id()] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) best = s_max[w] > best ? s_max[w] : best;
        atomicMax((unsigned long long*)out, (unsigned long long)best);
    }
}

static int launch_spmm(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* X, int F, const float* self_coef,
                       const float* S, const float* bias, int act, HeavyRows heavy, float* Y, hipStream_t st) {
    if (n_rows == 0 || F == 0) return PP_OK;
#define PP_SPMM_V4(L)                                                                                                              \
    do {                                                                                                                           \
        const unsigned grid = (unsigned)ceil_div(n_rows, (kBlock / L) * 2);                                                        \
        if (F == L * 4) k_spmm_v4<L, 2, true><<<grid, kBlock, 0, st>>>(ptr, idx, val, n_rows, X, F, self_coef, S, bias, act, heavy, Y);  \
        else k_spmm_v4<L, 2, false><<<grid, kBlock, 0, st>>>(ptr, idx, val, n_rows, X, F, self_coef, S, bias, act, heavy, Y);       \
    } while (0)
#define PP_SPMM_S1(L) k_spmm_s1<L><<<(unsigned)ceil_div(n_rows, kBlock / L), kBlock, 0, st>>>(ptr, idx, val, n_rows, X, F, self_coef, S, bias, act, heavy, Y)
    const bool vec = (F % 4 == 0) && (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)S | (uintptr_t)bias) % 16 == 0);
    if (vec) {
        const int q = F / 4;
        if (q <= 1) PP_SPMM_V4(1);
        else if (q <= 2) PP_SPMM_V4(2);
        else if (q <= 4) PP_SPMM_V4(4);
        else if (q <= 8) PP_SPMM_V4(8);
        else if (q <= 16) PP_SPMM_V4(16);
        else if (q <= 32) PP_SPMM_V4(32);
        else PP_SPMM_V4(64);
    } else {
        if (F <= 4) PP_SPMM_S1(4);
        else if (F <= 16) PP_SPMM_S1(16);
        else PP_SPMM_S1(64);
    }
#undef PP_SPMM_V4
#undef PP_SPMM_S1
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// longest[0] / longest[1] = longest row of the destination-major / source-major CSR of a finished plan (zeroed by the caller)
static int plan_longest_rows(const int32_t* in_ptr, int64_t n_dst, const int32_t* out_ptr, int64_t n_src, int64_t* longest, hipStream_t st) {
    if (n_dst > 0) {
        k_max_row_length<<<(unsigned)(ceil_div(n_dst, kBlock) < 1024 ? ceil_div(n_dst, kBlock) : 1024), kBlock, 0, st>>>(in_ptr, n_dst, longest);
        PP_LAUNCH_CHECK();
    }
    if (n_src > 0) {
        k_max_row_length<<<(unsigned)(ceil_div(n_src, kBlock) < 1024 ? ceil_div(n_src, kBlock) : 1024), kBlock, 0, st>>>(out_ptr, n_src, longest + 1);
        PP_LAUNCH_CHECK();
    }
    return PP_OK;
}

}  // namespace pp

using namespace pp;

extern "C" {

// ---------------------------------------------------------------- GCN plan
size_t pp_gcn_plan_ws_bytes(int64_t n_edges, int64_t n_nodes) { return carve_plan(nullptr, n_edges, n_nodes).total_bytes; }

// Phase 1 (see the header): everything that needs no foreign data.  n_src >= n_dst; the first n_dst source rows ARE the destinations.
int pp_gcn_plan_begin(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_src, int64_t n_dst, int row_sorted,
                      int32_t* in_ptr, int32_t* in_idx, float* in_val, int32_t* out_ptr, float* self_coef, float* dinv, int32_t* dst_order,
                      void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_edges >= 0 && n_dst >= 0 && n_src >= n_dst, PP_ERR_ARG, "pp_gcn_plan: bad sizes (need n_src >= n_dst >= 0)");
    PP_REQUIRE(n_edges < (int64_t)0x7fffffff && n_src < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_gcn_plan: E or N >= 2^31");
    PlanWs w = carve_plan(ws, n_edges, n_src);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_gcn_plan: workspace too small");
    PP_HIP(hipMemsetAsync(w.status, 0, 4 * sizeof(int64_t), st));
    if (n_src == 0) return PP_OK;
    const unsigned egrid = (unsigned)ceil_div(n_edges > 0 ? n_edges : 1, kBlock);
    if (n_dst > 0) PP_HIP(hipMemsetAsync(w.last_loop, 0xff, (size_t)n_dst * sizeof(int32_t), st));     // -1
    if (n_edges > 0) {
        k_plan_edges<<<egrid, kBlock, 0, st>>>(edge_index, n_edges, n_src, n_dst, edge_weight, w.packed, w.keys, w.last_loop,
                                               row_sorted ? out_ptr : nullptr, w.status + 1);
        PP_LAUNCH_CHECK();
    } else if (row_sorted) {
        k_ptr_from_sorted_i64_i32<<<1, kBlock, 0, st>>>(edge_index, 0, n_src, out_ptr);
        PP_LAUNCH_CHECK();
    }
    // edges grouped by destination (forward aggregation): order[p] = edge id, sorted[p] = its destination; one pass over the sorted keys
    // writes the row pointers, the (source, weight) copies and — for an order-2 De Bruijn model — the edge ids themselves, which ARE the
    // bipartite "last" plan
    if (n_edges > 0) {
        int rc = sort_pairs<uint32_t>(w.keys, nullptr, w.sorted, w.order, n_edges, 0, bits_for((uint64_t)(n_dst > 0 ? n_dst - 1 : 0)), w.scratch,
                                      w.scratch_bytes, st);
        if (rc != PP_OK) return rc;
    }
    k_group_by_dst<<<(unsigned)ceil_div(n_edges + 1, kBlock), kBlock, 0, st>>>(w.packed, n_edges, w.order, w.sorted, n_dst, in_ptr, in_idx, in_val,
                                                                              dst_order);
    PP_LAUNCH_CHECK();
    if (n_dst > 0) {
        k_gcn_degree_grouped<<<(unsigned)ceil_div(n_dst, kBlock), kBlock, 0, st>>>(in_idx, in_val, (const uint32_t*)in_ptr, w.last_loop, edge_weight,
                                                                                  n_dst, dinv, self_coef);
        PP_LAUNCH_CHECK();
    }
    // longest rows of both groupings (hub rows): bounded-grid reductions with one atomic per workgroup (folding them into the degree kernel —
    // one guarded atomic per wave on one word — doubled that kernel's time: 133 -> 251 us at 10^7 rows); the source grouping of an unsorted
    // edge list only exists after pp_gcn_plan_finish, which reports it then
    return plan_longest_rows(in_ptr, n_dst, row_sorted ? out_ptr : nullptr, row_sorted ? n_src : 0, w.status + 2, st);
}

// Phase 2: dinv[0..n_src) is complete (the halo part came from the owners) -> normalised coefficients in both groupings.
int pp_gcn_plan_finish(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_src, int64_t n_dst, int row_sorted,
                       const float* dinv, const int32_t* in_idx, float* in_val, int32_t* out_ptr, int32_t* out_idx, float* out_val, void* ws,
                       size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_edges >= 0 && n_dst >= 0 && n_src >= n_dst, PP_ERR_ARG, "pp_gcn_plan: bad sizes (need n_src >= n_dst >= 0)");
    PlanWs w = carve_plan(ws, n_edges, n_src);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_gcn_plan: workspace too small");
    if (n_src == 0) return PP_OK;
    const unsigned egrid = (unsigned)ceil_div(n_edges > 0 ? n_edges : 1, kBlock);
    if (n_edges > 0) {
        k_in_coefficients<<<egrid, kBlock, 0, st>>>(in_idx, w.sorted, dinv, n_edges, in_val);
        PP_LAUNCH_CHECK();
    }
    // edges grouped by source (backward = transposed aggregation)
    if (row_sorted) {       // De Bruijn layers come out of coalesce (row, col)-sorted: the edge order already is the grouping
        if (n_edges > 0) {  // (k_plan_edges wrote the row pointer in phase 1)
            k_gcn_coefficients<<<egrid, kBlock, 0, st>>>(edge_index, n_edges, edge_weight, nullptr, dinv, 0, out_idx, out_val);
            PP_LAUNCH_CHECK();
        }
        return PP_OK;
    }
    int rc = group_by(edge_index, n_edges, n_src, w, out_ptr, st);
    if (rc != PP_OK) return rc;
    if (n_edges > 0) {
        k_gcn_coefficients<<<egrid, kBlock, 0, st>>>(edge_index, n_edges, edge_weight, w.order, dinv, 0, out_idx, out_val);
        PP_LAUNCH_CHECK();
    }
    return plan_longest_rows(nullptr, 0, out_ptr, n_src, w.status + 2, st);
}

int pp_gcn_plan(const int64_t* edge_index, const float* edge_weight, int64_t n_edges, int64_t n_nodes, int row_sorted, int32_t* in_ptr,
                int32_t* in_idx, float* in_val, int32_t* out_ptr, int32_t* out_idx, float* out_val, float* self_coef, int32_t* dst_order,
                void* ws, size_t ws_bytes, pp_stream_t stream) {
    PP_REQUIRE(n_edges >= 0 && n_nodes >= 0, PP_ERR_ARG, "pp_gcn_plan: negative size");
    PlanWs w = carve_plan(ws, n_edges, n_nodes);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_gcn_plan: workspace too small");
    int rc = pp_gcn_plan_begin(edge_index, edge_weight, n_edges, n_nodes, n_nodes, row_sorted, in_ptr, in_idx, in_val, out_ptr, self_coef, w.dinv,
                               dst_order, ws, ws_bytes, stream);
    if (rc != PP_OK) return rc;
    return pp_gcn_plan_finish(edge_index, edge_weight, n_edges, n_nodes, n_nodes, row_sorted, w.dinv, in_idx, in_val, out_ptr, out_idx, out_val, ws,
                              ws_bytes, stream);
}

// bipartite higher-order -> first-order projection plan: forward rows = first-order nodes, backward rows = higher-order nodes
int pp_bipartite_plan(const int64_t* bipartite_index, int64_t n_pairs, int64_t n_ho, int64_t n_fo, int src_sorted, const float* pair_value,
                      int32_t* in_ptr, int32_t* in_idx, float* in_val, float* in_degree, int32_t* out_ptr, int32_t* out_idx, float* out_val,
                      void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_pairs >= 0 && n_ho >= 0 && n_fo >= 0, PP_ERR_ARG, "pp_bipartite_plan: negative size");
    const int64_t nmax = n_ho > n_fo ? n_ho : n_fo;
    PP_REQUIRE(n_pairs < (int64_t)0x7fffffff && nmax < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_bipartite_plan: size >= 2^31");
    PlanWs w = carve_plan(ws, n_pairs, nmax);
    PP_REQUIRE(ws_bytes >= w.total_bytes, PP_ERR_WORKSPACE, "pp_bipartite_plan: workspace too small");
    PP_HIP(hipMemsetAsync(w.status, 0, 4 * sizeof(int64_t), st));
    const unsigned egrid = (unsigned)ceil_div(n_pairs > 0 ? n_pairs : 1, kBlock);
    if (n_pairs > 0) {      // validate the sources (the destinations are validated by group_by)
        k_index_key<<<egrid, kBlock, 0, st>>>(bipartite_index, n_pairs, n_ho, w.keys, w.status + 1);
        PP_LAUNCH_CHECK();
    }
    int rc = group_by(bipartite_index + n_pairs, n_pairs, n_fo, w, in_ptr, st);   // by first-order destination
    if (rc != PP_OK) return rc;
    if (n_fo > 0) {
        k_ptr_diff_f32<<<(unsigned)ceil_div(n_fo, kBlock), kBlock, 0, st>>>(in_ptr, n_fo, in_degree);
        PP_LAUNCH_CHECK();
    }
    if (n_pairs > 0) {
        k_gather_index<<<egrid, kBlock, 0, st>>>(bipartite_index, w.order, n_pairs, in_idx);
        PP_LAUNCH_CHECK();
        if (pair_value && in_val) {
            k_gather_f32<<<egrid, kBlock, 0, st>>>(pair_value, w.order, n_pairs, in_val);
            PP_LAUNCH_CHECK();
        }
    }
    if (src_sorted) {       // bipartite_edge_index[0] = arange(U) for the "last"/"first" mappings: the pair order IS the grouping
        k_ptr_from_sorted_i64_i32<<<(unsigned)ceil_div(n_pairs + 1, kBlock), kBlock, 0, st>>>(bipartite_index, n_pairs, n_ho, out_ptr);
        PP_LAUNCH_CHECK();
        if (n_pairs > 0) {
            k_gather_index<<<egrid, kBlock, 0, st>>>(bipartite_index + n_pairs, nullptr, n_pairs, out_idx);
            PP_LAUNCH_CHECK();
            if (pair_value && out_val) PP_HIP(hipMemcpyAsync(out_val, pair_value, (size_t)n_pairs * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        return plan_longest_rows(in_ptr, n_fo, out_ptr, n_ho, w.status + 2, st);
    }
    rc = group_by(bipartite_index, n_pairs, n_ho, w, out_ptr, st);               // by higher-order source
    if (rc != PP_OK) return rc;
    if (n_pairs > 0) {
        k_gather_index<<<egrid, kBlock, 0, st>>>(bipartite_index + n_pairs, w.order, n_pairs, out_idx);
        PP_LAUNCH_CHECK();
        if (pair_value && out_val) {
            k_gather_f32<<<egrid, kBlock, 0, st>>>(pair_value, w.order, n_pairs, out_val);
            PP_LAUNCH_CHECK();
        }
    }
    return plan_longest_rows(in_ptr, n_fo, out_ptr, n_ho, w.status + 2, st);
}

const int64_t* pp_plan_result_ptr(void* ws) { return (const int64_t*)ws; }

// ---------------------------------------------------------------- propagation
int pp_spmm_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* X, int F, const float* self_coef,
                const float* S, const float* bias, int act, const int32_t* heavy_slot, const float* heavy_sum, float* Y, pp_stream_t stream) {
    PP_REQUIRE(n_rows >= 0 && F >= 0, PP_ERR_ARG, "pp_spmm_f32: negative size");
    PP_REQUIRE(act == 0 || act == 1, PP_ERR_ARG, "pp_spmm_f32: act must be 0 (none) or 1 (elu)");
    PP_REQUIRE(heavy_slot == nullptr || heavy_sum != nullptr, PP_ERR_ARG, "pp_spmm_f32: heavy_slot without heavy_sum");
    return launch_spmm(ptr, idx, val, n_rows, X, F, self_coef, self_coef ? (S ? S : X) : nullptr, bias, act, HeavyRows{heavy_slot, heavy_sum}, Y,
                       (hipStream_t)stream);
}

// out_max[0] = max_r (ptr[r+1] - ptr[r]) of a CSR row-pointer array (device int64): does the plan have hub rows?
int pp_max_row_length_i32(const int32_t* ptr, int64_t n_rows, int64_t* out_max, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_max_row_length_i32: negative size");
    PP_HIP(hipMemsetAsync(out_max, 0, sizeof(int64_t), st));
    if (n_rows == 0) return PP_OK;
    int64_t blocks = ceil_div(n_rows, kBlock);
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    k_max_row_length<<<(unsigned)blocks, kBlock, 0, st>>>(ptr, n_rows, out_max);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_heavy_chunk_entries(void) { return kHeavyChunk; }

size_t pp_spmm_heavy_ws_bytes(int64_t n_chunks, int F) { return align_up((size_t)(n_chunks > 0 ? n_chunks : 1) * (size_t)(F > 0 ? F : 1) * sizeof(float)); }

// heavy_sum[h, :] = sum over the CSR entries of heavy row h of val[p] * X[idx[p], :], h < n_heavy.  The entries of row h are covered
// by the chunks heavy_chunk_ptr[h] .. heavy_chunk_ptr[h+1], chunk k = entries [chunk_begin[k], chunk_end[k]) (<= pp_heavy_chunk_entries()).
int pp_spmm_heavy_f32(const int32_t* idx, const float* val, const float* X, int F, int64_t n_chunks, const int32_t* chunk_begin,
                      const int32_t* chunk_end, int64_t n_heavy, const int32_t* heavy_chunk_ptr, float* heavy_sum, void* ws, size_t ws_bytes,
                      pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_chunks >= 0 && n_heavy >= 0, PP_ERR_ARG, "pp_spmm_heavy_f32: negative size");
    PP_REQUIRE(F >= 4 && F % 4 == 0 && F <= 256, PP_ERR_ARG, "pp_spmm_heavy_f32: F must be a multiple of 4 in [4, 256]");
    PP_REQUIRE(((uintptr_t)X) % 16 == 0, PP_ERR_ARG, "pp_spmm_heavy_f32: X must be 16-byte aligned");
    PP_REQUIRE(ws_bytes >= pp_spmm_heavy_ws_bytes(n_chunks, F), PP_ERR_WORKSPACE, "pp_spmm_heavy_f32: workspace too small");
    if (n_heavy == 0 || n_chunks == 0) return PP_OK;
    float* partial = (float*)ws;
    const int q = F / 4;
#define PP_HEAVY(L) k_heavy_partial<L><<<(unsigned)n_chunks, kBlock, 0, st>>>(idx, val, X, F, chunk_begin, chunk_end, partial)
    if (q <= 1) PP_HEAVY(1);
    else if (q <= 2) PP_HEAVY(2);
    else if (q <= 4) PP_HEAVY(4);
    else if (q <= 8) PP_HEAVY(8);
    else if (q <= 16) PP_HEAVY(16);
    else if (q <= 32) PP_HEAVY(32);
    else PP_HEAVY(64);
#undef PP_HEAVY
    PP_LAUNCH_CHECK();
    k_heavy_combine<<<(unsigned)n_heavy, kBlock, 0, st>>>(partial, heavy_chunk_ptr, F, heavy_sum);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_act_backward_f32(const float* dY, const float* Y, int64_t n_rows, int F, int act, float* dpre, float* dbias, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 0, PP_ERR_ARG, "pp_act_backward_f32: negative size");
    if (dbias) PP_HIP(hipMemsetAsync(dbias, 0, (size_t)F * sizeof(float), st));
    const int64_t total = n_rows * (int64_t)F;
    if (total == 0) return PP_OK;
    int64_t g = ceil_div(total, kBlock * 8);
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    if ((g * kBlock) % F == 0)
        k_act_backward<true><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Y, n_rows, F, act, dpre, dbias);
    else
        k_act_backward<false><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Y, n_rows, F, act, dpre, dbias);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_bip_combine_f32(const float* A, const float* P, const float* deg, const float* bias, int64_t n_rows, int F, float* Y, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 0, PP_ERR_ARG, "pp_bip_combine_f32: negative size");
    const int64_t total = n_rows * (int64_t)F;
    if (total == 0) return PP_OK;
    int64_t g = ceil_div(total, kBlock * 4);
    if (g > kMaxGrid) g = kMaxGrid;
    k_bip_combine<<<(unsigned)g, kBlock, 0, st>>>(A, P, deg, bias, n_rows, F, Y);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_bip_combine_backward_f32(const float* dY, const float* Y, const float* deg, int64_t n_rows, int F, float* dA, float* dP, float* dbias,
                                pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 0, PP_ERR_ARG, "pp_bip_combine_backward_f32: negative size");
    if (dbias) PP_HIP(hipMemsetAsync(dbias, 0, (size_t)F * sizeof(float), st));
    const int64_t total = n_rows * (int64_t)F;
    if (total == 0) return PP_OK;
    int64_t g = ceil_div(total, kBlock * 8);
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    if ((g * kBlock) % F == 0)
        k_bip_combine_backward<true><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Y, deg, n_rows, F, dA, dP, dbias);
    else
        k_bip_combine_backward<false><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Y, deg, n_rows, F, dA, dP, dbias);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_spmm_act_backward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* D, int F, const float* Z,
                             float* colsum, float* dX, pp_stream_t stream) {
    return pp_spmm_act_backward_drop_f32(ptr, idx, val, n_rows, D, F, Z, colsum, dX, 0.0, 0, 0, 0, stream);
}

int pp_spmm_act_backward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, const float* D, int F, const float* Z,
                                  float* colsum, float* dX, double drop_p, int64_t drop_seed, int64_t drop_tag, int64_t drop_row0,
                                  pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_spmm_act_backward_f32: negative size");
    PP_REQUIRE(F >= 4 && F % 4 == 0 && F <= 256, PP_ERR_ARG, "pp_spmm_act_backward_f32: F must be a multiple of 4 in [4, 256]");
    PP_REQUIRE(((uintptr_t)D | (uintptr_t)Z | (uintptr_t)dX) % 16 == 0, PP_ERR_ARG, "pp_spmm_act_backward_f32: 16-byte alignment");
    PP_REQUIRE(drop_p >= 0.0 && drop_p < 1.0, PP_ERR_ARG, "pp_spmm_act_backward_drop_f32: p must lie in [0, 1)");
    const DropSite drop = drop_site(drop_p, drop_seed, drop_tag, drop_row0);
    if (colsum) PP_HIP(hipMemsetAsync(colsum, 0, (size_t)F * sizeof(float), st));
    if (n_rows == 0) return PP_OK;
    const int q = F / 4;
#define PP_SAB(L)                                                                                                       \
    do {                                                                                                                \
        int64_t blocks = ceil_div(n_rows, kBlock / L);                                                                  \
        if (blocks > kMaxGrid) blocks = kMaxGrid;                                                                       \
        if (drop.thr != 0u) k_spmm_act_backward<L, true><<<(unsigned)blocks, kBlock, 0, st>>>(ptr, idx, val, n_rows, D, F, Z, colsum, dX, drop);  \
        else k_spmm_act_backward<L, false><<<(unsigned)blocks, kBlock, 0, st>>>(ptr, idx, val, n_rows, D, F, Z, colsum, dX, drop);                \
    } while (0)
    if (q <= 1) PP_SAB(1);
    else if (q <= 2) PP_SAB(2);
    else if (q <= 4) PP_SAB(4);
    else if (q <= 8) PP_SAB(8);
    else if (q <= 16) PP_SAB(16);
    else if (q <= 32) PP_SAB(32);
    else PP_SAB(64);
#undef PP_SAB
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_scale_rows_f32(const float* X, const float* coef, int64_t n_rows, int F, float* out, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n_rows * (int64_t)F;
    if (total <= 0) return PP_OK;
    k_scale_rows<<<(unsigned)ceil_div(total, kBlock), kBlock, 0, st>>>(X, coef, n_rows, F, out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_halo_fold_f32(const float* own, const float* recv, const int32_t* slot, const float* extra, const float* self_coef, const float* dpre,
                     int64_t n_rows, int F, float* out, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 4 && F % 4 == 0, PP_ERR_ARG, "pp_halo_fold_f32: F must be a positive multiple of 4");
    PP_REQUIRE(own && out, PP_ERR_ARG, "pp_halo_fold_f32: null matrix");
    PP_REQUIRE((slot == nullptr) == (recv == nullptr) || n_rows == 0, PP_ERR_ARG, "pp_halo_fold_f32: recv and slot come together");
    PP_REQUIRE((self_coef == nullptr) == (dpre == nullptr), PP_ERR_ARG, "pp_halo_fold_f32: self_coef and dpre come together");
    PP_REQUIRE(((uintptr_t)own | (uintptr_t)recv | (uintptr_t)extra | (uintptr_t)dpre | (uintptr_t)out) % 16 == 0, PP_ERR_ARG,
               "pp_halo_fold_f32: 16-byte alignment");
    const int64_t total = n_rows * (int64_t)(F / 4);
    if (total == 0) return PP_OK;
    int64_t blocks = ceil_div(total, kBlock);
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    k_halo_fold<<<(unsigned)blocks, kBlock, 0, st>>>(own, recv, slot, extra, self_coef, dpre, n_rows, F / 4, out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_dropout_f32(const float* X, int64_t n_rows, int F, double p, int64_t seed, int64_t tag, int64_t row0, const int64_t* rows, float* out,
                   pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 1, PP_ERR_ARG, "pp_dropout_f32: bad shape");
    PP_REQUIRE(p >= 0.0 && p < 1.0, PP_ERR_ARG, "pp_dropout_f32: p must lie in [0, 1)");
    const int64_t total = n_rows * (int64_t)F;
    if (total == 0) return PP_OK;
    int64_t g = ceil_div(total, kBlock * 8);
    if (g > kMaxGrid) g = kMaxGrid;
    k_dropout<<<(unsigned)g, kBlock, 0, st>>>(X, n_rows, F, dropout_key(seed, tag), (uint32_t)(p * 4294967296.0), (float)(1.0 / (1.0 - p)), row0, rows, out);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_dropout_act_backward_f32(const float* dY, const float* Ydrop, int64_t n_rows, int F, double p, int64_t seed, int64_t tag, int64_t row0,
                                const int64_t* rows, int act, float* dpre, float* dbias, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && F >= 1, PP_ERR_ARG, "pp_dropout_act_backward_f32: bad shape");
    PP_REQUIRE(p >= 0.0 && p < 1.0, PP_ERR_ARG, "pp_dropout_act_backward_f32: p must lie in [0, 1)");
    PP_REQUIRE(n_rows == 0 || (dpre != nullptr && (!act || Ydrop != nullptr)), PP_ERR_ARG, "pp_dropout_act_backward_f32: dpre (and Ydrop with act) required");
    if (dbias) PP_HIP(hipMemsetAsync(dbias, 0, (size_t)F * sizeof(float), st));
    const int64_t total = n_rows * (int64_t)F;
    if (total == 0) return PP_OK;
    int64_t g = ceil_div(total, kBlock * 8);
    if (g > kMaxGrid) g = kMaxGrid;
    const uint32_t key = dropout_key(seed, tag), thr = (uint32_t)(p * 4294967296.0);
    const float scale = (float)(1.0 / (1.0 - p)), keep = (float)(1.0 - p);
    if ((g * kBlock) % F == 0)
        k_dropout_act_backward<true><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Ydrop, n_rows, F, key, thr, scale, keep, row0, rows, act, dpre, dbias);
    else
        k_dropout_act_backward<false><<<(unsigned)g, kBlock, (size_t)F * sizeof(float), st>>>(dY, Ydrop, n_rows, F, key, thr, scale, keep, row0, rows, act, dpre, dbias);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"

// =====================================================================================================
// Weight gradient of a dense layer on the matrix cores:  dW[M,K] = dH^T X  (+ db[M] = column sums of dH)
//   dH : [N, M] row-major (gradient of the layer output),  X : [N, K] row-major (layer input),  N >> M, K.
// This is the one GEMM of the train step a vendor library handles badly (reduction length N ~ 10^7, output
// 64x64): rocBLAS needs 4.7 ms for it at N = 10^7, the HBM floor for reading dH and X once is ~0.9 ms.
// v_mfma_f32_32x32x2_f32 takes its A operand as A[i = lane&31][k = lane>>5] and B as B[k = lane>>5][j = lane&31];
// with the reduction index k running over ROWS n of dH and X, both operands are plain coalesced row reads
// (lanes 0-31: 128 contiguous bytes of row n, lanes 32-63: of row n+1) - no LDS staging, no transposes.
// Each wave owns a contiguous range of rows and a TMxTK block of 32x32 output tiles in accumulators; the
// waves of all workgroups write partial tiles that a second tiny kernel sums in a fixed order (deterministic).
namespace pp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWgTile = 2;                 // 2x2 tiles of 32x32 per wave = 64x64 outputs, 64 accumulator registers
constexpr int kWgUnroll = 8;               // row pairs in flight per iteration (16 rows of dH and X)

__global__ __launch_bounds__(kBlock) void k_weight_grad(const float* __restrict__ dH, const float* __restrict__ X, int64_t n_rows, int M, int K,
                                                       int64_t rows_per_wave, float* __restrict__ partial,
                                                       float* __restrict__ partial_bias) {
    // blockIdx.y selects the 64x64 output block (i-block major), blockIdx.x the row range
    const int k_blocks = (K + 63) / 64;
    const int i_base = (blockIdx.y / k_blocks) * 64;
    const int j_base = (blockIdx.y % k_blocks) * 64;
    const int lane = lane_id();
    const int c = lane & 31, h = lane >> 5;
    const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    const int64_t n_begin = wave_global * rows_per_wave;
    int64_t n_end = n_begin + rows_per_wave;
    if (n_end > n_rows) n_end = n_rows;

    f32x16 acc[kWgTile][kWgTile];
#pragma unroll
    for (int a = 0; a < kWgTile; ++a)
#pragma unroll
        for (int b = 0; b < kWgTile; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bias_acc[kWgTile] = {0.f, 0.f};
    const bool ok_i[kWgTile] = {i_base + c < M, i_base + 32 + c < M};
    const bool ok_j[kWgTile] = {j_base + c < K, j_base + 32 + c < K};

    for (int64_t n0 = n_begin; n0 < n_end; n0 += 2 * kWgUnroll) {
        float a[kWgUnroll][kWgTile], b[kWgUnroll][kWgTile];
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u) {
            const int64_t n = n0 + 2 * u + h;
            const bool live = n < n_end;
#pragma unroll
            for (int t = 0; t < kWgTile; ++t) {
                a[u][t] = (live && ok_i[t]) ? dH[n * M + i_base + 32 * t + c] : 0.f;
                b[u][t] = (live && ok_j[t]) ? X[n * K + j_base + 32 * t + c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u) {
#pragma unroll
            for (int ti = 0; ti < kWgTile; ++ti) {
                bias_acc[ti] += a[u][ti];
#pragma unroll
                for (int tj = 0; tj < kWgTile; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][ti], b[u][tj], acc[ti][tj], 0, 0, 0);
            }
        }
    }
    // the 4 waves of the workgroup fold their 64x64 tiles through LDS in wave order, then ONE partial tile per
    // workgroup goes to HBM: partial[blockIdx.x * gridDim.y + blockIdx.y][64][64].
    // C/D layout of the 32x32 MFMA: row = (r&3) + 8*(r>>2) + 4*h, col = c
    __shared__ float s_tile[64 * 64];
    __shared__ float s_bias[64];
    for (int w = 0; w < kWavesPerBlock; ++w) {
        if (wave_id() == w) {
#pragma unroll
            for (int ti = 0; ti < kWgTile; ++ti)
#pragma unroll
                for (int tj = 0; tj < kWgTile; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int at = (32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * tj + c;
                        s_tile[at] = (w == 0 ? 0.f : s_tile[at]) + acc[ti][tj][r];
                    }
#pragma unroll
            for (int t = 0; t < kWgTile; ++t) {
                const float v = bias_acc[t] + __shfl_xor(bias_acc[t], 32, kWave);   // both row parities of column c
                if (h == 0) s_bias[32 * t + c] = (w == 0 ? 0.f : s_bias[32 * t + c]) + v;
            }
        }
        __syncthreads();
    }
    float* out = partial + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) << 12);
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) out[e] = s_tile[e];
    if (partial_bias && (blockIdx.y % k_blocks) == 0 && threadIdx.x < 64)
        partial_bias[((int64_t)blockIdx.x * ((M + 63) / 64) + blockIdx.y / k_blocks) * 64 + threadIdx.x] = s_bias[threadIdx.x];
}

// Sums the per-workgroup partial tiles: 16 outputs x 16 interleaved slices of partials per workgroup, every slice
// accumulated sequentially and the 16 slice sums folded in a fixed order => bitwise reproducible results.
constexpr int kWgSlices = 16;
// 64 x 64 layers: the same contraction on v_mfma_f32_16x16x4_f32 with 16-byte operand loads.  Lane (i, kq) reads the four
// consecutive columns 4i .. 4i+3 of row n + kq of dH and of X (one dwordx4 each: a load instruction covers four whole rows), and the
// 16 MFMAs of a step pair every dH component c with every X component c': tile (c, c') accumulates dW[4*row + c][4*col + c'].
#ifndef PP_WG64_STEPS
#define PP_WG64_STEPS 6
#endif
#ifndef PP_WG64_WAVES
#define PP_WG64_WAVES 3
#endif
__global__ __launch_bounds__(kBlock, PP_WG64_WAVES) void k_weight_grad64(const float* __restrict__ dH, const float* __restrict__ X, int64_t n_rows,
                                                         int64_t rows_per_wave, float* __restrict__ partial,
                                                         float* __restrict__ partial_bias) {
    constexpr int kSteps = PP_WG64_STEPS;                  // 4-row steps per iteration, fetched in two half-batches
    const bool stream_rows = PP_NT_WG && n_rows * (int64_t)256 >= kStreamFromBytes;      // both operands are read exactly once
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    const int64_t n_begin = wave_global * rows_per_wave;
    int64_t n_end = n_begin + rows_per_wave;
    if (n_end > n_rows) n_end = n_rows;
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    f32x4v acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bias_acc[4] = {0.f, 0.f, 0.f, 0.f};
    // two half-batches of kSteps/2 steps: while the MFMAs of one half run, the loads of the other are in flight
    constexpr int kHalf = kSteps / 2;
    float4 a[kSteps], b[kSteps];
    auto fetch = [&](int h, int64_t n0) {
#pragma unroll
        for (int u = h * kHalf; u < (h + 1) * kHalf; ++u) {
            const int64_t n = n0 + 4 * u + kq;
            const bool live = n < n_end;
            a[u] = live ? load_row_f4(dH + n * 64 + 4 * i, stream_rows) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[u] = live ? load_row_f4(X + n * 64 + 4 * i, stream_rows) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto contract = [&](int h) {
#pragma unroll
        for (int u = h * kHalf; u < (h + 1) * kHalf; ++u) {
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bias_acc[c] += av[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[d], acc[c][d], 0, 0, 0);
            }
        }
    };
    fetch(0, n_begin);
    for (int64_t n0 = n_begin; n0 < n_end; n0 += 4 * kSteps) {
        fetch(1, n0);
        contract(0);
        fetch(0, n0 + 4 * kSteps);
        contract(1);
    }
    // fold the 4 waves through LDS in wave order; C/D layout of the 16x16 MFMA: row = 4*(lane>>4) + reg, col = lane&15
    __shared__ float s_tile[64 * 64];
    __shared__ float s_bias[64];
    for (int w = 0; w < kWavesPerBlock; ++w) {
        if (wave_id() == w) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int at = (4 * (4 * kq + reg) + c) * 64 + 4 * i + d;
                        s_tile[at] = (w == 0 ? 0.f : s_tile[at]) + acc[c][d][reg];
                    }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = bias_acc[c];
                v += __shfl_xor(v, 16, kWave);
                v += __shfl_xor(v, 32, kWave);
                if (kq == 0) s_bias[4 * i + c] = (w == 0 ? 0.f : s_bias[4 * i + c]) + v;
            }
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)blockIdx.x << 12);
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) out[e] = s_tile[e];
    if (partial_bias && threadIdx.x < 64) partial_bias[(int64_t)blockIdx.x * 64 + threadIdx.x] = s_bias[threadIdx.x];
}

// Wider layers (M = 64*MB, K = 64*KB, MB*KB <= 16 waves): the workgroup walks ONE row range and wave (bi, bj) owns output block (bi, bj),
// contracted exactly like k_weight_grad64 from the 256-byte column slices bi of dH and bj of X.  The waves of a workgroup read the same
// rows at the same time, so every slice comes from HBM once (the per-block launch of k_weight_grad re-read dH KB times and X MB times:
// 9.4 ms for 2*10^7 rows of 128x128 - 2 x 20 GB - against the 20 GB a single pass needs).
template <int MB, int KB>
__global__ __launch_bounds__(64 * MB * KB, (MB * KB > 4 ? 4 : 3)) void k_weight_grad_blocks(const float* __restrict__ dH, const float* __restrict__ X, int64_t n_rows,
                                                                     int64_t rows_per_group, float* __restrict__ partial,
                                                                     float* __restrict__ partial_bias) {
    constexpr int kSteps = MB * KB > 4 ? 2 : 6;            // 16 waves per workgroup leave 128 registers per lane
    constexpr int M = 64 * MB, K = 64 * KB;
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    const int wave = threadIdx.x >> 6, bi = wave / KB, bj = wave % KB;
    const int64_t n_begin = (int64_t)blockIdx.x * rows_per_group;
    int64_t n_end = n_begin + rows_per_group;
    if (n_end > n_rows) n_end = n_rows;
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    f32x4v acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bias_acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* ap = dH + 64 * bi + 4 * i;
    const float* bp = X + 64 * bj + 4 * i;
    constexpr int kHalf = kSteps / 2;                       // half-batches: loads of one in flight under the MFMAs of the other
    float4 a[kSteps], b[kSteps];
    auto fetch = [&](int h, int64_t n0) {
#pragma unroll
        for (int u = h * kHalf; u < (h + 1) * kHalf; ++u) {
            const int64_t n = n0 + 4 * u + kq;
            const bool live = n < n_end;
            a[u] = live ? *(const float4*)(ap + n * M) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[u] = live ? *(const float4*)(bp + n * K) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto contract = [&](int h) {
#pragma unroll
        for (int u = h * kHalf; u < (h + 1) * kHalf; ++u) {
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bias_acc[c] += av[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[d], acc[c][d], 0, 0, 0);
            }
        }
    };
    fetch(0, n_begin);
    for (int64_t n0 = n_begin; n0 < n_end; n0 += 4 * kSteps) {
        fetch(1, n0);
        contract(0);
        fetch(0, n0 + 4 * kSteps);
        contract(1);
    }
    // every wave owns its block: straight to partial[group][bi*KB + bj][64][64]; tile (c, d), register reg of lane (i, kq) is element
    // [4*(4*kq + reg) + c][4*i + d] of the block (as in k_weight_grad64)
    float* out = partial + (((int64_t)blockIdx.x * (MB * KB) + wave) << 12);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            *(float4*)(out + (4 * (4 * kq + reg) + c) * 64 + 4 * i) = make_float4(acc[c][0][reg], acc[c][1][reg], acc[c][2][reg], acc[c][3][reg]);
    if (partial_bias && bj == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = bias_acc[c];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) partial_bias[((int64_t)blockIdx.x * MB + bi) * 64 + 4 * i + c] = v;
        }
    }
}

// The same contraction with the operands STAGED THROUGH LDS (round 3).  In k_weight_grad_blocks every wave fetches its two 256-byte column
// slices itself: a 4-row step is 2 KB of L1 traffic per wave for 16 MFMAs, and with 4 waves per SIMD that is 64 bytes per clock and CU —
// the whole L1 rate — for data of which only a quarter is distinct (slice bi is read by KB waves, slice bj by MB): 54 % of the fp32 matrix
// peak at 256 x 256.  Here the workgroup copies each row stage (kWgRows rows of dH and X, distinct bytes only) into LDS once, double
// buffered, one barrier per stage; the waves take their slices with ds_read_b128 (row stride padded by 16 bytes so that the four k-rows of
// an MFMA step fall into different banks).
#ifndef PP_WG_ROWS
#define PP_WG_ROWS 8
#endif
constexpr int kWgRows = PP_WG_ROWS;                   // rows per LDS stage (a multiple of 4)

template <int MB, int KB>
__global__ __launch_bounds__(64 * MB * KB, (MB * KB > 4 ? 4 : 3)) void k_weight_grad_lds(const float* __restrict__ dH, const float* __restrict__ X, int64_t n_rows,
                                                                  int64_t rows_per_group, float* __restrict__ partial,
                                                                  float* __restrict__ partial_bias) {
    constexpr int M = 64 * MB, K = 64 * KB, T = 64 * MB * KB, R = kWgRows;
    constexpr int SA = M + 4, SB = K + 4;                  // padded row strides (floats)
    constexpr int NA = R * M / 4, NB = R * K / 4;          // float4 per stage and matrix
    constexpr int LA = (NA + T - 1) / T, LB = (NB + T - 1) / T;
    __shared__ __attribute__((aligned(16))) float s_a[2][R * SA];
    __shared__ __attribute__((aligned(16))) float s_b[2][R * SB];
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    const int wave = threadIdx.x >> 6, bi = wave / KB, bj = wave % KB;
    const int64_t n_begin = (int64_t)blockIdx.x * rows_per_group;
    int64_t n_end = n_begin + rows_per_group;
    if (n_end > n_rows) n_end = n_rows;
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    f32x4v acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bias_acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 ra[LA], rb[LB];
    auto gload = [&](int64_t n0) {
#pragma unroll
        for (int l = 0; l < LA; ++l) {
            const int e = threadIdx.x + l * T;
            const int row = e / (M / 4), c4 = e % (M / 4);
            const int64_t n = n0 + row;
            ra[l] = (e < NA && n < n_end) ? *(const float4*)(dH + n * M + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int l = 0; l < LB; ++l) {
            const int e = threadIdx.x + l * T;
            const int row = e / (K / 4), c4 = e % (K / 4);
            const int64_t n = n0 + row;
            rb[l] = (e < NB && n < n_end) ? *(const float4*)(X + n * K + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int l = 0; l < LA; ++l) {
            const int e = threadIdx.x + l * T;
            if (e < NA) *(float4*)(&s_a[buf][(e / (M / 4)) * SA + 4 * (e % (M / 4))]) = ra[l];
        }
#pragma unroll
        for (int l = 0; l < LB; ++l) {
            const int e = threadIdx.x + l * T;
            if (e < NB) *(float4*)(&s_b[buf][(e / (K / 4)) * SB + 4 * (e % (K / 4))]) = rb[l];
        }
    };
    gload(n_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int64_t n0 = n_begin; n0 < n_end; n0 += R) {
        const bool more = n0 + R < n_end;
        if (more) gload(n0 + R);                           // (in flight under the MFMAs of this stage)
#pragma unroll
        for (int u = 0; u < R / 4; ++u) {
            const float4 a = *(const float4*)(&s_a[buf][(4 * u + kq) * SA + 64 * bi + 4 * i]);
            const float4 b = *(const float4*)(&s_b[buf][(4 * u + kq) * SB + 64 * bj + 4 * i]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bias_acc[c] += av[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[d], acc[c][d], 0, 0, 0);
            }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = partial + (((int64_t)blockIdx.x * (MB * KB) + wave) << 12);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            *(float4*)(out + (4 * (4 * kq + reg) + c) * 64 + 4 * i) = make_float4(acc[c][0][reg], acc[c][1][reg], acc[c][2][reg], acc[c][3][reg]);
    if (partial_bias && bj == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = bias_acc[c];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) partial_bias[((int64_t)blockIdx.x * MB + bi) * 64 + 4 * i + c] = v;
        }
    }
}

#ifndef PP_WG_LDS_FROM
#define PP_WG_LDS_FROM 4
#endif
constexpr int kWgLdsFrom = PP_WG_LDS_FROM;             // shapes with at least this many 64 x 64 blocks stage their operands through LDS

// groups = workgroups resident at once (asked from the runtime once per shape), never more than the workspace formula provides
template <int MB, int KB>
static int launch_weight_grad_blocks(hipStream_t st, const float* dH, const float* X, int64_t n_rows, int64_t max_groups, float* partial,
                                     float* partial_bias, int64_t* groups_out) {
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (MB * KB >= kWgLdsFrom) PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_weight_grad_lds<MB, KB>, 64 * MB * KB, 0));
        else PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_weight_grad_blocks<MB, KB>, 64 * MB * KB, 0));
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
    }
    int64_t groups = ceil_div(n_rows > 0 ? n_rows : 1, 128);
    if (groups > resident) groups = resident;
    if (groups > max_groups) groups = max_groups;
    const int64_t rows_per_group = ceil_div(ceil_div(n_rows > 0 ? n_rows : 1, groups), 4) * 4;
    groups = ceil_div(n_rows > 0 ? n_rows : 1, rows_per_group);
    if (MB * KB >= kWgLdsFrom) k_weight_grad_lds<MB, KB><<<(unsigned)groups, 64 * MB * KB, 0, st>>>(dH, X, n_rows, rows_per_group, partial, partial_bias);
    else k_weight_grad_blocks<MB, KB><<<(unsigned)groups, 64 * MB * KB, 0, st>>>(dH, X, n_rows, rows_per_group, partial, partial_bias);
    *groups_out = groups;
    return PP_OK;
}

static int launch_weight_grad_blocks_any(int mb, int kb, hipStream_t st, const float* dH, const float* X, int64_t n_rows, int64_t max_groups,
                                         float* partial, float* partial_bias, int64_t* groups_out) {
#define PP_WGB(A, B) if (mb == A && kb == B) return launch_weight_grad_blocks<A, B>(st, dH, X, n_rows, max_groups, partial, partial_bias, groups_out)
    PP_WGB(1, 2); PP_WGB(2, 1); PP_WGB(2, 2); PP_WGB(1, 4); PP_WGB(4, 1); PP_WGB(2, 4); PP_WGB(4, 2); PP_WGB(4, 4);
#undef PP_WGB
    return PP_ERR_ARG;
}

static inline bool weight_grad_blocks_shape(int M, int K) {
    const int mb = M / 64, kb = K / 64;
    return M % 64 == 0 && K % 64 == 0 && mb * kb > 1 && (mb == 1 || mb == 2 || mb == 4) && (kb == 1 || kb == 2 || kb == 4);
}

__global__ __launch_bounds__(kBlock) void k_weight_grad_reduce(const float* __restrict__ partial, const float* __restrict__ partial_bias,
                                                              int64_t n_parts, int M, int K, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float s_sum[kWgSlices][kBlock / kWgSlices];
    const int k_blocks = (K + 63) / 64, i_blocks = (M + 63) / 64;
    const int n_blocks = k_blocks * i_blocks;
    const int o = threadIdx.x % (kBlock / kWgSlices), g = threadIdx.x / (kBlock / kWgSlices);
    const int idx = blockIdx.x * (kBlock / kWgSlices) + o;            // output element (weights first, then bias)
    const int n_w = M * K;
    float s = 0.f;
    if (idx < n_w) {
        const int i = idx / K, j = idx - i * K;
        const int blk = (i / 64) * k_blocks + (j / 64);
        const float* p = partial + ((int64_t)blk << 12) + (i % 64) * 64 + (j % 64);
        for (int64_t w = g; w < n_parts; w += kWgSlices) s += p[(w * n_blocks) << 12];
    } else if (db && idx < n_w + M) {
        const int i = idx - n_w;
        const float* p = partial_bias + (i / 64) * 64 + (i % 64);
        for (int64_t w = g; w < n_parts; w += kWgSlices) s += p[w * i_blocks * 64];
    }
    s_sum[g][o] = s;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < kWgSlices; ++q) t += s_sum[q][o];
        if (idx < n_w) dW[idx] = t;
        else if (db && idx < n_w + M) db[idx - n_w] = t;
    }
}

static inline int64_t weight_grad_waves(int64_t n_rows) {
    int64_t waves = ceil_div(n_rows, 2 * kWgUnroll * 8);        // at least 128 rows per wave
#ifndef PP_WG_CAP
#define PP_WG_CAP 3
#endif
    // (k_weight_grad64 is resident at 3 workgroups per CU: a cap of 4 left a second, quarter-filled round — 1.11 -> 0.975 ms at 10^7 x 64 x 64)
    const int64_t cap = 256 * PP_WG_CAP * kWavesPerBlock;       // workgroups per CU
    if (waves > cap) waves = cap;
    if (waves < 1) waves = 1;
    return ceil_div(waves, kWavesPerBlock) * kWavesPerBlock;
}

}  // namespace pp

extern "C" {

size_t pp_weight_grad_ws_bytes(int64_t n_rows, int M, int K) {
    const int64_t blocks = (int64_t)((M + 63) / 64) * ((K + 63) / 64);
    const int64_t parts = pp::weight_grad_waves(n_rows) / pp::kWavesPerBlock;
    return pp::align_up((size_t)parts * blocks * 4096 * sizeof(float)) + pp::align_up((size_t)parts * ((M + 63) / 64) * 64 * sizeof(float));
}

// dW[M,K] = dH[N,M]^T X[N,K], db[M] = column sums of dH (db may be NULL); fp32 on v_mfma_f32_32x32x2_f32
int pp_weight_grad_f32(const float* dH, const float* X, int64_t n_rows, int M, int K, float* dW, float* db, void* ws, size_t ws_bytes,
                       pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && M >= 1 && K >= 1, PP_ERR_ARG, "pp_weight_grad_f32: bad shape");
    PP_REQUIRE(ws_bytes >= pp_weight_grad_ws_bytes(n_rows, M, K), PP_ERR_WORKSPACE, "pp_weight_grad_f32: workspace too small");
    const int i_blocks = (M + 63) / 64, k_blocks = (K + 63) / 64;
    const int64_t waves = pp::weight_grad_waves(n_rows);
    const int64_t rows_per_wave = pp::ceil_div(pp::ceil_div(n_rows > 0 ? n_rows : 1, waves), 2) * 2;
    float* partial = (float*)ws;
    const int64_t parts = waves / pp::kWavesPerBlock;
    float* partial_bias = (float*)((char*)ws + pp::align_up((size_t)parts * i_blocks * k_blocks * 4096 * sizeof(float)));
    dim3 grid((unsigned)(waves / pp::kWavesPerBlock), (unsigned)(i_blocks * k_blocks));
    if (M == 64 && K == 64 && (((uintptr_t)dH | (uintptr_t)X) % 16 == 0)) {
        const int64_t rows4 = pp::ceil_div(rows_per_wave, 4) * 4;          // a wave's range starts on a 4-row step
        const int64_t waves4 = pp::ceil_div(pp::ceil_div(n_rows > 0 ? n_rows : 1, rows4), pp::kWavesPerBlock) * pp::kWavesPerBlock;
        pp::k_weight_grad64<<<(unsigned)(waves4 / pp::kWavesPerBlock), pp::kBlock, 0, st>>>(dH, X, n_rows, rows4, partial, db ? partial_bias : nullptr);
        PP_LAUNCH_CHECK();
        const int outs64 = M * K + (db ? M : 0);
        pp::k_weight_grad_reduce<<<(unsigned)pp::ceil_div(outs64, pp::kBlock / pp::kWgSlices), pp::kBlock, 0, st>>>(
            partial, db ? partial_bias : nullptr, waves4 / pp::kWavesPerBlock, M, K, dW, db);
        PP_LAUNCH_CHECK();
        return PP_OK;
    }
    if (pp::weight_grad_blocks_shape(M, K) && (((uintptr_t)dH | (uintptr_t)X) % 16 == 0)) {
        int64_t groups = 0;
        const int rc = pp::launch_weight_grad_blocks_any(M / 64, K / 64, st, dH, X, n_rows, parts, partial, db ? partial_bias : nullptr, &groups);
        if (rc != PP_OK) return rc;
        PP_LAUNCH_CHECK();
        const int outs_b = M * K + (db ? M : 0);
        pp::k_weight_grad_reduce<<<(unsigned)pp::ceil_div(outs_b, pp::kBlock / pp::kWgSlices), pp::kBlock, 0, st>>>(
            partial, db ? partial_bias : nullptr, groups, M, K, dW, db);
        PP_LAUNCH_CHECK();
        return PP_OK;
    }
    // the general kernel is resident at 2 workgroups per CU (171 registers): its own, smaller cap keeps the launch to one full round
#ifndef PP_WG_GENERIC_CAP
#define PP_WG_GENERIC_CAP 2
#endif
    int64_t parts_g = parts;
    const int64_t round_g = 256 * PP_WG_GENERIC_CAP / (i_blocks * k_blocks) > 0 ? 256 * PP_WG_GENERIC_CAP / (i_blocks * k_blocks) : 1;
    if (parts_g > round_g) parts_g = round_g;
    const int64_t rows_g = pp::ceil_div(pp::ceil_div(n_rows > 0 ? n_rows : 1, parts_g * pp::kWavesPerBlock), 2) * 2;
    grid.x = (unsigned)parts_g;
    pp::k_weight_grad<<<grid, pp::kBlock, 0, st>>>(dH, X, n_rows, M, K, rows_g, partial, db ? partial_bias : nullptr);
    PP_LAUNCH_CHECK();
    const int outs = M * K + (db ? M : 0);
    pp::k_weight_grad_reduce<<<(unsigned)pp::ceil_div(outs, pp::kBlock / pp::kWgSlices), pp::kBlock, 0, st>>>(
        partial, db ? partial_bias : nullptr, parts_g, M, K, dW, db);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"

// =====================================================================================================
// Dense layer on the matrix cores, optionally with the ELU backward of the layer BELOW fused into its epilogue:
//     out[N,Q] = ( A[N,P] . B[P,Q] + bias ) (*) g'               B = W^T (W is [Q,P], forward) or W (W is [P,Q], input gradient)
//   grad_act : y = ELU(pre) of the layer below, stored by its forward; g' = ELU'(pre) = (y > 0 ? 1 : y + 1) is multiplied into
//              the result and colsum[q] accumulates it: the input-gradient GEMM then directly yields the gradient w.r.t. the
//              lower layer's PRE-activation and that layer's bias gradient - no separate 3-pass ELU-backward kernel.
// v_mfma_f32_16x16x4_f32: lane (i = lane&15, kq = lane>>4) owns row i of a 16-row tile and the k-range [kq*P/4, (kq+1)*P/4):
// the reduction order inside a dot product is free, so each lane reads P/16 float4 of its row (contiguous quarter row) and
// the matching rows of B live in registers for the whole kernel (P*Q/64 VGPRs: 64 for 64x64).  Waves are persistent and
// prefetch the next tile's A registers while the MFMAs of the current tile run.
namespace pp {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int P, int Q>
__global__ __launch_bounds__(kBlock) void k_dense(const float* __restrict__ A, const float* __restrict__ W, int w_transposed, int64_t n_rows,
                                                 const float* __restrict__ bias, const float* __restrict__ grad_act,
                                                 float* __restrict__ colsum, float* __restrict__ out) {
    constexpr int KQ = P / 4;            // k values per lane
    constexpr int CT = Q / 16;           // 16-column output tiles
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    float b[KQ][CT];
#pragma unroll
    for (int t = 0; t < KQ; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int k = kq * KQ + t, j = ct * 16 + i;
            b[t][ct] = w_transposed ? W[j * P + k] : W[k * Q + j];
        }
    float bias_c[CT], col_acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        bias_c[ct] = bias ? bias[ct * 16 + i] : 0.f;
        col_acc[ct] = 0.f;
    }
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    float4 a_cur[KQ / 4], a_nxt[KQ / 4];
    auto load_tile = [&](int64_t t, float4 (&dst)[KQ / 4]) {
        const int64_t r = t * 16 + i;
        const bool live = t < n_tiles && r < n_rows;
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c)
            dst[c] = live ? *(const float4*)(A + r * P + kq * KQ + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_tile(tile, a_cur);
    for (; tile < n_tiles; tile += n_waves) {
        load_tile(tile + n_waves, a_nxt);
        // the activation values of the gradient epilogue are fetched BEFORE the MFMAs so that their latency hides behind them
        float gp[CT][4];
        if (grad_act) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int64_t r = tile * 16 + 4 * kq + reg;
                    gp[ct][reg] = r < n_rows ? grad_act[r * Q + ct * 16 + i] : 0.f;
                }
        }
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) {
            float av[4] = {a_cur[c].x, a_cur[c].y, a_cur[c].z, a_cur[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b[4 * c + e][ct], acc[ct], 0, 0, 0);
            }
        }
        // C/D layout of the 16x16 MFMA: row = 4*(lane>>4) + reg, col = lane&15
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = tile * 16 + 4 * kq + reg;
                if (r < n_rows) {
                    float v = acc[ct][reg] + bias_c[ct];
                    if (grad_act) {
                        const float y = gp[ct][reg];
                        v *= y > 0.f ? 1.f : y + 1.f;            // ELU'(pre) from the stored activation y = ELU(pre)
                        col_acc[ct] += v;
                    }
                    out[r * Q + ct * 16 + i] = v;
                }
            }
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) a_cur[c] = a_nxt[c];
    }
    if (colsum) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = col_acc[ct];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) atomicAdd(&colsum[ct * 16 + i], v);
        }
    }
}

template <int P>
static int launch_dense_q(int Q, unsigned grid, hipStream_t st, const float* A, const float* W, int wt, int64_t n,
                          const float* bias, const float* grad_act, float* colsum, float* out) {
    switch (Q) {
        case 16: k_dense<P, 16><<<grid, kBlock, 0, st>>>(A, W, wt, n, bias, grad_act, colsum, out); break;
        case 32: k_dense<P, 32><<<grid, kBlock, 0, st>>>(A, W, wt, n, bias, grad_act, colsum, out); break;
        case 64: k_dense<P, 64><<<grid, kBlock, 0, st>>>(A, W, wt, n, bias, grad_act, colsum, out); break;
        default: return PP_ERR_ARG;
    }
    return PP_OK;
}

}  // namespace pp

extern "C" {

static inline bool dense_exact(int P, int Q) { return (P == 16 || P == 32 || P == 64) && (Q == 16 || Q == 32 || Q == 64); }

// 1: both widths in {16,32,64} (B in registers, also pp_dense_backward_f32); 2: other widths up to 64 (zero-padded, pp_dense_narrow_f32);
// 3: 64/128/256 with a side > 64 (weights streamed through LDS, pp_wide_layer_f32); 0: library GEMM territory
int pp_dense_supported(int P, int Q) {
    if (dense_exact(P, Q)) return 1;
    if (pp_dense_narrow_supported(P, Q)) return 2;
    return pp_wide_layer_supported(P, Q) ? 3 : 0;
}

int pp_dense_f32(const float* A, const float* W, int w_transposed, int64_t n_rows, int P, int Q, const float* bias,
                 const float* grad_act, float* colsum, float* out, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_dense_f32: negative size");
    const int kind = pp_dense_supported(P, Q);
    PP_REQUIRE(kind != 0, PP_ERR_ARG, "pp_dense_f32: unsupported layer shape %dx%d (supported: widths up to 64, and 64/128/256)", P, Q);
    if (kind == 2) return pp_dense_narrow_f32(A, W, w_transposed, n_rows, P, Q, bias, grad_act, colsum, out, stream);
    if (kind == 3) {
        // forward (W [Q,P], one row per output column: as the kernel wants it) or input gradient (W [P,Q]: transposed into ws first)
        if (grad_act == nullptr && colsum == nullptr)
            return pp_wide_layer_f32(nullptr, nullptr, nullptr, n_rows, n_rows, n_rows, A, P, nullptr, W, w_transposed ? 0 : 1, Q, bias, 0, 0, nullptr, nullptr,
                                     nullptr, nullptr, out, nullptr, ws, ws_bytes, stream);
        PP_REQUIRE(bias == nullptr, PP_ERR_ARG, "pp_dense_f32: the gradient epilogue of a wide layer takes no bias");
        return pp_wide_layer_f32(nullptr, nullptr, nullptr, n_rows, n_rows, n_rows, A, P, nullptr, W, w_transposed ? 0 : 1, Q, nullptr, grad_act ? 1 : 0, 1,
                                 grad_act, nullptr, nullptr, nullptr, out, colsum, ws, ws_bytes, stream);
    }
    PP_REQUIRE(((uintptr_t)A | (uintptr_t)out) % 16 == 0, PP_ERR_ARG, "pp_dense_f32: A and out must be 16-byte aligned");
    if (colsum) PP_HIP(hipMemsetAsync(colsum, 0, (size_t)Q * sizeof(float), st));
    if (n_rows == 0) return PP_OK;
    int64_t blocks = pp::ceil_div(pp::ceil_div(n_rows, 16), pp::kWavesPerBlock);
    const int64_t cap = 256 * 3;                       // 3 workgroups (12 waves) per CU: the kernel is register-heavy
    if (blocks > cap) blocks = cap;
    int rc;
    switch (P) {
        case 16: rc = pp::launch_dense_q<16>(Q, (unsigned)blocks, st, A, W, w_transposed, n_rows, bias, grad_act, colsum, out); break;
        case 32: rc = pp::launch_dense_q<32>(Q, (unsigned)blocks, st, A, W, w_transposed, n_rows, bias, grad_act, colsum, out); break;
        default: rc = pp::launch_dense_q<64>(Q, (unsigned)blocks, st, A, W, w_transposed, n_rows, bias, grad_act, colsum, out); break;
    }
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"

// =====================================================================================================
// Softmax cross-entropy over node logits, forward AND gradient in one pass (the train step's loss; the reference has no
// training loop of its own).  One lane per row (C <= 64 classes kept in registers), mean reduction:
//   loss = mean_i( logsumexp(z_i) - z_i[y_i] ),   dz[i][c] = (softmax(z_i)[c] - [c == y_i]) / N
// torch's generic nll_loss kernels need 0.75 ms for 5*10^5 x 8 logits; this streams them once (~20 MB).
namespace pp {

template <int kMaxC>
__global__ __launch_bounds__(kBlock) void k_cross_entropy(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t n, int C,
                                                         float inv_n, float* __restrict__ partial, float* __restrict__ dlogits) {
    __shared__ float s_part[kWavesPerBlock];
    float local = 0.f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float z[kMaxC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            z[c] = c < C ? logits[i * C + c] : -INFINITY;
            m = z[c] > m ? z[c] : m;
        }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            z[c] = c < C ? expf(z[c] - m) : 0.f;
            sum += z[c];
        }
        const int64_t y = target[i];
        const float inv = 1.f / sum;
        float zy = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            if (c < C) {
                const float p = z[c] * inv;
                if (c == y) zy = p;
                if (dlogits) dlogits[i * C + c] = (p - (c == y ? 1.f : 0.f)) * inv_n;
            }
        }
        local += -logf(zy > 0.f ? zy : 1e-45f);
    }
    local = wave_sum(local);
    if (lane_id() == 0) s_part[wave_id()] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) t += s_part[w];
        partial[blockIdx.x] = t * inv_n;                      // (summed in a fixed order by k_sum_partials: the loss is bitwise reproducible)
    }
}

// out[0] = sum of partial[0..n) in a fixed order (thread j takes j, j + 256, ..; fixed-shape tree over the threads)
__global__ __launch_bounds__(kBlock) void k_sum_partials(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float s_part[kWavesPerBlock];
    float t = 0.f;
    for (int j = threadIdx.x; j < n; j += kBlock) t += partial[j];
    t = wave_sum(t);
    if (lane_id() == 0) s_part[wave_id()] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) v += s_part[w];
        out[0] = v;
    }
}

}  // namespace pp

extern "C" {

size_t pp_cross_entropy_ws_bytes(void) { return pp::align_up((size_t)pp::kMaxGrid * sizeof(float)); }

// loss[0] = mean cross-entropy of logits [n, C] against int64 targets [n]; dlogits [n, C] (may be NULL) = its gradient.  C <= 64.
// ws (pp_cross_entropy_ws_bytes()): per-workgroup partial sums, added in a fixed order — the loss is bitwise reproducible.
int pp_cross_entropy_f32(const float* logits, const int64_t* target, int64_t n, int C, float* loss, float* dlogits, void* ws, size_t ws_bytes,
                         pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n >= 0 && C >= 1 && C <= 64, PP_ERR_ARG, "pp_cross_entropy_f32: needs 1 <= C <= 64 (got %d)", C);
    PP_REQUIRE(ws != nullptr && ws_bytes >= pp_cross_entropy_ws_bytes(), PP_ERR_WORKSPACE, "pp_cross_entropy_f32: workspace too small");
    if (n == 0) {
        PP_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return PP_OK;
    }
    int64_t g = pp::ceil_div(n, pp::kBlock);
    if (g > pp::kMaxGrid) g = pp::kMaxGrid;
    const float inv_n = 1.0f / (float)n;
    float* partial = (float*)ws;
    if (C <= 8) pp::k_cross_entropy<8><<<(unsigned)g, pp::kBlock, 0, st>>>(logits, target, n, C, inv_n, partial, dlogits);
    else if (C <= 16) pp::k_cross_entropy<16><<<(unsigned)g, pp::kBlock, 0, st>>>(logits, target, n, C, inv_n, partial, dlogits);
    else pp::k_cross_entropy<64><<<(unsigned)g, pp::kBlock, 0, st>>>(logits, target, n, C, inv_n, partial, dlogits);
    PP_LAUNCH_CHECK();
    pp::k_sum_partials<<<1, pp::kBlock, 0, st>>>(partial, (int)g, loss);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"

// =====================================================================================================
// Adam over ALL parameter tensors of a model in one launch (what torch.optim.Adam does with ~10 multi-tensor launches and, per step, a
// few dozen host-side device queries: on a small graph or a per-rank share the optimizer's host time was 1 ms of a 3.3 ms step).
// Same update as torch.optim.Adam (amsgrad off, maximize off):  g' = g + wd*p;  m += (1-b1)(g'-m);  v = b2 v + (1-b2) g'^2;
// p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps).   blockIdx.y = tensor, blockIdx.x strides over its elements.
namespace pp {
constexpr int kAdamTensors = 24;                                  // tensors per launch (the table travels as a kernel argument)
struct AdamTable {
    float* p[kAdamTensors];
    const float* g[kAdamTensors];
    float* m[kAdamTensors];
    float* v[kAdamTensors];
    int64_t n[kAdamTensors];
};
__global__ __launch_bounds__(kBlock) void k_adam(AdamTable t, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float eps, float wd) {
    const int k = blockIdx.y;
    const int64_t n = t.n[k];
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
        const float pe = p[e];
        const float ge = g[e] + wd * pe;
        const float me = m[e] + (1.f - b1) * (ge - m[e]);
        const float ve = b2 * v[e] + (1.f - b2) * ge * ge;
        m[e] = me;
        v[e] = ve;
        p[e] = pe - lr_over_bc1 * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
    }
}
}  // namespace pp

extern "C" {

// One Adam step (step = 1, 2, ...) over n_tensors fp32 tensors: HOST arrays of DEVICE pointers params / grads / exp_avg / exp_avg_sq and of
// element counts.  Stands in for torch.optim.Adam.step() of a user's training loop around the reference's DBGNN (nn/dbgnn.py:72-151; the
// reference itself ships no training loop).
int pp_adam_f32(int n_tensors, void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_tensors >= 0 && step >= 1, PP_ERR_ARG, "pp_adam_f32: needs n_tensors >= 0 and step >= 1");
    PP_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, PP_ERR_ARG, "pp_adam_f32: betas in [0, 1), eps >= 0");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    for (int k = 0; k < n_tensors;) {
        pp::AdamTable t;
        int cnt = 0;
        int64_t longest = 0;
        for (; k < n_tensors && cnt < pp::kAdamTensors; ++k) {
            PP_REQUIRE(numel[k] >= 0, PP_ERR_ARG, "pp_adam_f32: negative element count");
            if (numel[k] == 0) continue;
            PP_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k], PP_ERR_ARG, "pp_adam_f32: null tensor %d", k);
            t.p[cnt] = (float*)params[k];
            t.g[cnt] = (const float*)grads[k];
            t.m[cnt] = (float*)exp_avg[k];
            t.v[cnt] = (float*)exp_avg_sq[k];
            t.n[cnt] = numel[k];
            if (numel[k] > longest) longest = numel[k];
            ++cnt;
        }
        if (cnt == 0) continue;
        for (int q = cnt; q < pp::kAdamTensors; ++q) { t.p[q] = nullptr; t.g[q] = nullptr; t.m[q] = nullptr; t.v[q] = nullptr; t.n[q] = 0; }
        int64_t gx = pp::ceil_div(longest, pp::kBlock);
        if (gx > 1024) gx = 1024;
        pp::k_adam<<<dim3((unsigned)gx, (unsigned)cnt), pp::kBlock, 0, st>>>(t, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2,
                                                                             (float)eps, (float)weight_decay);
        PP_LAUNCH_CHECK();
    }
    return PP_OK;
}

}  // extern "C"

// =====================================================================================================
// Whole backward of a dense layer y = x W^T (+ b) in ONE pass over its two big operands:
//     d_in[N,K] = (dH . W) (*) ELU'(x)      colsum_in[K] = column sums of d_in        (gradient for the layer below, as in k_dense)
//     dW[M,K]   = dH^T x                    db[M]        = column sums of dH           (this layer's parameter gradients)
// Separately (k_dense + k_weight_grad) dH and x are each read twice: 5 feature-matrix passes; fused: 3 (read dH, read x, write d_in).
// Per 16-row tile a wave runs two MFMA streams on v_mfma_f32_16x16x4_f32:
//   rows x W    : A = dH quarter rows (float4 loads, lane = (row, k-quarter)), B = W in registers            -> d_in tile
//   dH^T x rows : the contraction runs over the tile's 16 ROWS; both operands are natural coalesced row reads: A[i'][k] =
//                 dH[row 4k+reg][col], B[k][j] = x[row 4k+reg][col] - the SAME registers that feed the ELU' epilogue.
// dW lives in (M/16)(K/16) accumulators for the whole persistent wave, is folded through LDS per workgroup and summed in a fixed
// order by k_weight_grad_reduce (partial layout [workgroup][64][64]); M, K <= 64.
namespace pp {

template <int M, int K>
__global__ __launch_bounds__(kBlock) void k_dense_backward(const float* __restrict__ dH, const float* __restrict__ X, const float* __restrict__ W,
                                                          int64_t n_rows, int fuse_act, float* __restrict__ d_in,
                                                          float* __restrict__ colsum_in, float* __restrict__ partial_w,
                                                          float* __restrict__ partial_b) {
    constexpr int KQ = M / 4;            // dH columns per lane in the quarter-row layout
    constexpr int MT = M / 16, CT = K / 16;
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    float b[KQ][CT];                     // W[k][j]: d_in = dH . W, W is [M, K]
#pragma unroll
    for (int t = 0; t < KQ; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[t][ct] = W[(kq * KQ + t) * K + ct * 16 + i];
    f32x4 acc_w[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc_w[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float col_in[CT], col_b[MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) col_in[ct] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) col_b[mt] = 0.f;

    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
    float4 a_cur[KQ / 4], a_nxt[KQ / 4];
    auto load_tile = [&](int64_t t, float4 (&dst)[KQ / 4]) {
        const int64_t r = t * 16 + i;
        const bool live = t < n_tiles && r < n_rows;
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c)
            dst[c] = live ? *(const float4*)(dH + r * M + kq * KQ + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_tile(tile, a_cur);
    for (; tile < n_tiles; tile += n_waves) {
        load_tile(tile + n_waves, a_nxt);
        // natural-layout rows of this tile: x (ELU' epilogue AND B operand of dW) and dH (A operand of dW)
        float xr[CT][4], hr[MT][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t r = tile * 16 + 4 * kq + reg;
            const bool live = r < n_rows;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) xr[ct][reg] = live ? X[r * K + ct * 16 + i] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hr[mt][reg] = live ? dH[r * M + mt * 16 + i] : 0.f;
        }
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) {
            const float av[4] = {a_cur[c].x, a_cur[c].y, a_cur[c].z, a_cur[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], b[4 * c + e][ct], acc[ct], 0, 0, 0);
        }
        // dW += dH_tile^T x_tile : step `reg` contracts the rows {reg, 4+reg, 8+reg, 12+reg} (k = lane>>4)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                col_b[mt] += hr[mt][reg];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc_w[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[mt][reg], xr[ct][reg], acc_w[mt][ct], 0, 0, 0);
            }
        if (d_in) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int64_t r = tile * 16 + 4 * kq + reg;
                    if (r < n_rows) {
                        float v = acc[ct][reg];
                        if (fuse_act) {
                            const float y = xr[ct][reg];
                            v *= y > 0.f ? 1.f : y + 1.f;
                        }
                        col_in[ct] += v;
                        d_in[r * K + ct * 16 + i] = v;
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) a_cur[c] = a_nxt[c];
    }
    if (colsum_in) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = col_in[ct];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) atomicAdd(&colsum_in[ct * 16 + i], v);
        }
    }
    // fold the 4 waves' dW (and db) through LDS in wave order, one partial [64][64] tile per workgroup (zero padded)
    __shared__ float s_tile[64 * 64];
    __shared__ float s_bias[64];
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) s_tile[e] = 0.f;
    if (threadIdx.x < 64) s_bias[threadIdx.x] = 0.f;
    __syncthreads();
    for (int w = 0; w < kWavesPerBlock; ++w) {
        if (wave_id() == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)          // C/D layout: row = 4*(lane>>4) + reg, col = lane&15
                        s_tile[(mt * 16 + 4 * kq + reg) * 64 + ct * 16 + i] += acc_w[mt][ct][reg];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v = col_b[mt];
                v += __shfl_xor(v, 16, kWave);
                v += __shfl_xor(v, 32, kWave);
                if (kq == 0) s_bias[mt * 16 + i] += v;
            }
        }
        __syncthreads();
    }
    float* out = partial_w + ((int64_t)blockIdx.x << 12);
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) out[e] = s_tile[e];
    if (partial_b && threadIdx.x < 64) partial_b[(int64_t)blockIdx.x * 64 + threadIdx.x] = s_bias[threadIdx.x];
}

template <int M>
static int launch_dense_backward_k(int K, unsigned grid, hipStream_t st, const float* dH, const float* X, const float* W, int64_t n, int fuse,
                                   float* d_in, float* colsum_in, float* pw, float* pb) {
    switch (K) {
        case 16: k_dense_backward<M, 16><<<grid, kBlock, 0, st>>>(dH, X, W, n, fuse, d_in, colsum_in, pw, pb); break;
        case 32: k_dense_backward<M, 32><<<grid, kBlock, 0, st>>>(dH, X, W, n, fuse, d_in, colsum_in, pw, pb); break;
        case 64: k_dense_backward<M, 64><<<grid, kBlock, 0, st>>>(dH, X, W, n, fuse, d_in, colsum_in, pw, pb); break;
        default: return PP_ERR_ARG;
    }
    return PP_OK;
}

// shared with pp_gcn_fused.hip: fixed-order sum of per-workgroup [64][64] partial weight gradients
int weight_grad_reduce(const float* partial_w, const float* partial_b, int64_t n_parts, int M, int K, float* dW, float* db, hipStream_t st) {
    const int outs = M * K + (db ? M : 0);
    k_weight_grad_reduce<<<(unsigned)ceil_div(outs, kBlock / kWgSlices), kBlock, 0, st>>>(partial_w, db ? partial_b : nullptr, n_parts, M, K, dW, db);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

static inline int64_t dense_backward_blocks(int64_t n_rows) {
    int64_t blocks = ceil_div(ceil_div(n_rows > 0 ? n_rows : 1, 16), kWavesPerBlock);
    const int64_t cap = 256 * 2;                       // 2 workgroups (8 waves) per CU: ~230 registers per lane
    return blocks > cap ? cap : blocks;
}

}  // namespace pp

extern "C" {

size_t pp_dense_backward_ws_bytes(int64_t n_rows) {
    const int64_t blocks = pp::dense_backward_blocks(n_rows);
    return pp::align_up((size_t)blocks * 4096 * sizeof(float)) + pp::align_up((size_t)blocks * 64 * sizeof(float));
}

// dH [N,M], X [N,K] (the layer's input), W [M,K].  d_in [N,K] / colsum_in [K] may be NULL (input gradient not needed),
// db [M] may be NULL.  fuse_act: multiply d_in by ELU'(.) recovered from X as a stored activation.  M, K in {16, 32, 64}.
int pp_dense_backward_f32(const float* dH, const float* X, const float* W, int64_t n_rows, int M, int K, int fuse_act, float* d_in,
                          float* colsum_in, float* dW, float* db, void* ws, size_t ws_bytes, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_dense_backward_f32: negative size");
    PP_REQUIRE(pp_dense_supported(M, K) == 1, PP_ERR_ARG, "pp_dense_backward_f32: unsupported layer shape %dx%d (16/32/64 only)", M, K);
    PP_REQUIRE(ws_bytes >= pp_dense_backward_ws_bytes(n_rows), PP_ERR_WORKSPACE, "pp_dense_backward_f32: workspace too small");
    PP_REQUIRE(((uintptr_t)dH) % 16 == 0, PP_ERR_ARG, "pp_dense_backward_f32: dH must be 16-byte aligned");
    if (colsum_in) PP_HIP(hipMemsetAsync(colsum_in, 0, (size_t)K * sizeof(float), st));
    const int64_t blocks = pp::dense_backward_blocks(n_rows);
    float* pw = (float*)ws;
    float* pb = (float*)((char*)ws + pp::align_up((size_t)blocks * 4096 * sizeof(float)));
    int rc;
    switch (M) {
        case 16: rc = pp::launch_dense_backward_k<16>(K, (unsigned)blocks, st, dH, X, W, n_rows, fuse_act, d_in, colsum_in, pw, db ? pb : nullptr); break;
        case 32: rc = pp::launch_dense_backward_k<32>(K, (unsigned)blocks, st, dH, X, W, n_rows, fuse_act, d_in, colsum_in, pw, db ? pb : nullptr); break;
        default: rc = pp::launch_dense_backward_k<64>(K, (unsigned)blocks, st, dH, X, W, n_rows, fuse_act, d_in, colsum_in, pw, db ? pb : nullptr); break;
    }
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    const int outs = M * K + (db ? M : 0);
    pp::k_weight_grad_reduce<<<(unsigned)pp::ceil_div(outs, pp::kBlock / pp::kWgSlices), pp::kBlock, 0, st>>>(pw, db ? pb : nullptr, blocks, M, K, dW, db);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
