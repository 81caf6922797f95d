// pathpyg_amd — a whole GCN layer per kernel: CSR aggregation fused with the dense product on the matrix cores.
//
//   forward :  Y[r] = act( (sum_e val[e] X[idx[e]] + self[r] X[r]) . W^T + bias )            (reference nn/dbgnn.py:131-140, GCNConv)
//
// The reference (and pp_dense_f32 + pp_spmm_f32) evaluate A_hat (X W^T): the transformed matrix H = X W^T makes a round trip
// through HBM (write N*Q, gather it back).  (A_hat X) W^T is the same product re-associated: a wave aggregates a tile of 16
// destination rows straight from X into LDS and multiplies the tile by W on v_mfma_f32_16x16x4_f32, so the only N-sized
// traffic left is the gather itself and the store of Y.  W (<= 64x64 fp32) sits in LDS for the whole persistent workgroup.
#include <stdlib.h>

#include <type_traits>

#include "pp_common.h"
#include "pp_internal.h"

namespace pp {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kGcnThreads = 256;                 // 4 waves share one copy of W
constexpr int kFirst = 4;                        // neighbours per row fetched in the first, fully overlapped, batch
constexpr int kGcnWaves = kGcnThreads / kWave;

// ---------------------------------------------------------------------------------------------------------------------------------
// Gather stage shared by the layer kernels (the form for matrices below 4 GiB): aggregates one 16-row tile
//     tile[r] = self[r] * X[r] + sum_e val[e] * X[idx[e]]
// into a wave-private LDS buffer (row stride P + 4 floats).  kLanes = P/4 lanes own a row (one float4 each), a lane group walks its kRows
// rows kBatch at a time.  Everything is read with buffer loads (32-bit offsets; an offset >= kBufOob reads as zeros without touching
// memory, so absent neighbours and rows past the end need no branch around a load and no select behind it — on this chip the VALU and
// the matrix pipe of a SIMD do not overlap across waves, tools/probes/mfma/overlap.hip, so the gather's VALU instructions are paid in
// full).  Per tile ONE coalesced load each brings the 17 row pointers, the 16 self coefficients (both one tile ahead) and the first 4 and
// the next 4 (index, value) pairs of all 16 rows (lane 4*row + slot); the lane groups take them with ds_bpermute.  Rows with more than 8
// neighbours continue chunk-wise.
template <int P, bool kHeavy, int kBatchOverride = 0>
struct TileGather {
    static constexpr int kLanes = P / 4, kGroups = kWave / kLanes, kRows = 16 / kGroups, TS = P + 4;
    static constexpr int kBatch = kBatchOverride > 0 ? kBatchOverride : (kRows < 2 ? kRows : (kRows >= 8 ? 4 : 2));      // 128-wide rows: registers to spare for a deeper gather
    static constexpr int kShift = P == 16 ? 6 : (P == 32 ? 7 : (P == 64 ? 8 : 9));      // log2(row bytes)
    buf_t rs_x, rs_ptr, rs_self, rs_slot;
    const int32_t* idx;
    const float* val;
    const float* self_coef;
    HeavyRows heavy;
    int64_t n_rows, n_self, n_tiles;
    int lane, g, l, a_rr, a_row, a_pair;
    uint32_t l16;
    int pv_next, hv_next;
    float scv_next;

    __device__ __forceinline__ TileGather(const float* X, const int32_t* ptr, const int32_t* idx_, const float* val_, const float* self_coef_,
                                          HeavyRows heavy_, int64_t n_rows_, int64_t n_self_)
        : rs_x(buf_of(X)), rs_ptr(buf_of(ptr)), rs_self(buf_of(self_coef_)), rs_slot(buf_of(heavy_.slot)), idx(idx_), val(val_),
          self_coef(self_coef_), heavy(heavy_), n_rows(n_rows_), n_self(n_self_), n_tiles((n_rows_ + 15) / 16), lane(lane_id()),
          g(lane_id() / kLanes), l(lane_id() % kLanes), pv_next(0), hv_next(-1), scv_next(0.f) {
        a_rr = (lane >> 2) * 4;            // ds_bpermute byte addresses: lane = row of this lane's (row, slot) pair ..
        a_row = g * kRows * 4;             // .. lane = first row of this lane group ..
        a_pair = g * kRows * 16;           // .. lane = 4 * (first row of this lane group)
        l16 = 16u * l;
    }

    // row pointers, self coefficients and hub slots of `tile` (one round trip off the critical path when called a tile ahead)
    __device__ __forceinline__ void prefetch(int64_t tile) {
        const int64_t rl = tile * 16 + lane;
        const bool live = tile < n_tiles;
        pv_next = (int)buf_load_u32(rs_ptr, (live && lane <= 16) ? (uint32_t)(rl < n_rows ? rl : n_rows) * 4u : kBufOob);
        scv_next = self_coef != nullptr ? buf_load_f32(rs_self, (live && lane < 16 && rl < n_self) ? (uint32_t)rl * 4u : kBufOob) : 0.f;
        if constexpr (kHeavy) hv_next = (live && lane < 16 && rl < n_rows) ? (int)buf_load_u32(rs_slot, (uint32_t)rl * 4u) : -1;
    }

    // aggregates tile t (whose pointers the last prefetch() fetched) and prefetches tile t_next
    __device__ __forceinline__ void run(int64_t t, int64_t t_next, float* __restrict__ tile, float* __restrict__ agg_out) {
        const int pv = pv_next, hv = hv_next;
        const float scv = scv_next;
        prefetch(t_next);
        // (entry offsets are taken relative to the tile's first entry: the index / value arrays may exceed 4 GiB, one tile's share cannot)
        const int p_first = __builtin_amdgcn_readfirstlane(pv);
        const buf_t rs_idx = buf_of(idx + p_first), rs_val = buf_of(val != nullptr ? val + p_first : nullptr);
        uint32_t jo_a, jo_b;               // byte offsets of the source rows of pairs 0..3 / 4..7 of row lane >> 2 (slot lane & 3)
        float cv_a, cv_b;
        bool more;                         // some row of the tile has more than 4 neighbours (wave-uniform)
        {
            const int pr = lane_read_i(a_rr, pv), pn = lane_read_i(a_rr + 4, pv);
            const bool hub = kHeavy && lane_read_i(a_rr, hv) >= 0;
            const int e = pr + (lane & 3);
            const bool in_a = !hub && e < pn, in_b = !hub && e + 4 < pn;
            more = __ballot(in_b) != 0ull;
            const uint32_t eo_a = in_a ? (uint32_t)(e - p_first) * 4u : kBufOob, eo_b = in_b ? (uint32_t)(e + 4 - p_first) * 4u : kBufOob;
            const uint32_t j_a = buf_load_u32(rs_idx, eo_a);
            cv_a = val != nullptr ? buf_load_f32(rs_val, eo_a) : (in_a ? 1.f : 0.f);
            uint32_t j_b = 0u;
            cv_b = 0.f;
            if (more) {
                j_b = buf_load_u32(rs_idx, eo_b);
                cv_b = val != nullptr ? buf_load_f32(rs_val, eo_b) : (in_b ? 1.f : 0.f);
            }
            jo_a = in_a ? (j_a << kShift) : kBufOob;
            jo_b = in_b ? (j_b << kShift) : kBufOob;
        }
        const uint32_t row_base = ((uint32_t)(t * 16) + (uint32_t)(g * kRows)) << kShift;        // (below 4 GiB by the caller's choice of this form)
#pragma unroll
        for (int b0 = 0; b0 < kRows; b0 += kBatch) {
            float4 x[kBatch][kFirst], sr[kBatch];
            float v[kBatch][kFirst], sc[kBatch];
#pragma unroll
            for (int qq = 0; qq < kBatch; ++qq) {
                const int q = b0 + qq;
                sc[qq] = lane_read_f(a_row + 4 * q, scv);
                const bool self_here = t * 16 + g * kRows + q < n_self;
                sr[qq] = buf_load_f4(rs_x, self_here ? row_base + ((uint32_t)q << kShift) + l16 : kBufOob);
#pragma unroll
                for (int u = 0; u < kFirst; ++u) {
                    const uint32_t jo = (uint32_t)lane_read_i(a_pair + 16 * q + 4 * u, (int)jo_a);
                    v[qq][u] = lane_read_f(a_pair + 16 * q + 4 * u, cv_a);
                    x[qq][u] = buf_load_f4(rs_x, jo + l16);
                }
            }
            float4 acc[kBatch];
#pragma unroll
            for (int qq = 0; qq < kBatch; ++qq) {
                acc[qq] = make_float4(sc[qq] * sr[qq].x, sc[qq] * sr[qq].y, sc[qq] * sr[qq].z, sc[qq] * sr[qq].w);
#pragma unroll
                for (int u = 0; u < kFirst; ++u) {
                    acc[qq].x += v[qq][u] * x[qq][u].x; acc[qq].y += v[qq][u] * x[qq][u].y;
                    acc[qq].z += v[qq][u] * x[qq][u].z; acc[qq].w += v[qq][u] * x[qq][u].w;
                }
            }
            if (more) {                    // pairs 4..7 (every lane takes part: ds_bpermute reads active lanes only; absent pairs read zeros)
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    const int q = b0 + qq;
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        const uint32_t jo = (uint32_t)lane_read_i(a_pair + 16 * q + 4 * u, (int)jo_b);
                        v[qq][u] = lane_read_f(a_pair + 16 * q + 4 * u, cv_b);
                        x[qq][u] = buf_load_f4(rs_x, jo + l16);
                    }
                }
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq)
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        acc[qq].x += v[qq][u] * x[qq][u].x; acc[qq].y += v[qq][u] * x[qq][u].y;
                        acc[qq].z += v[qq][u] * x[qq][u].z; acc[qq].w += v[qq][u] * x[qq][u].w;
                    }
            }
#pragma unroll
            for (int qq = 0; qq < kBatch; ++qq) {
                const int q = b0 + qq;
                const int64_t r = t * 16 + g * kRows + q;
                const int p0 = lane_read_i(a_row + 4 * q, pv);
                const int hs = kHeavy ? lane_read_i(a_row + 4 * q, hv) : -1;
                const int p_end = lane_read_i(a_row + 4 * q + 4, pv);                  // (every lane takes part, see above)
                const int p1 = hs >= 0 ? p0 : p_end;                                  // a hub row: its neighbour sum is already in heavy.sum
                for (int base = p0 + 2 * kFirst; base < p1; base += kLanes) {         // rows with more than 8 neighbours
                    const int mine = base + l;
                    const uint32_t eo = mine < p1 ? (uint32_t)(mine - p_first) * 4u : kBufOob;
                    const uint32_t my_jo = mine < p1 ? (buf_load_u32(rs_idx, eo) << kShift) : kBufOob;
                    const float my_v = val != nullptr ? buf_load_f32(rs_val, eo) : (mine < p1 ? 1.f : 0.f);
                    const int cnt = p1 - base < kLanes ? p1 - base : kLanes;
                    for (int e = 0; e < cnt; e += 4) {
                        float4 y[4];
                        float w4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t jo = (uint32_t)__shfl((int)my_jo, e + u, kLanes);
                            w4[u] = __shfl(my_v, e + u, kLanes);
                            y[u] = buf_load_f4(rs_x, jo + l16);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc[qq].x += w4[u] * y[u].x; acc[qq].y += w4[u] * y[u].y; acc[qq].z += w4[u] * y[u].z; acc[qq].w += w4[u] * y[u].w;
                        }
                    }
                }
                if (kHeavy && hs >= 0) {
                    const float4 h = *(const float4*)(heavy.sum + (int64_t)hs * P + 4 * l);
                    acc[qq].x += h.x; acc[qq].y += h.y; acc[qq].z += h.z; acc[qq].w += h.w;
                }
                *(float4*)(tile + (g * kRows + q) * TS + 4 * l) = acc[qq];
                if (agg_out != nullptr && r < n_rows) store_row_f4(agg_out + r * P + 4 * l, acc[qq], n_rows * (int64_t)(P * 4) >= kStreamFromBytes);
            }
        }
    }
};

// One wave per 16-row tile, persistent workgroups of 4 waves.  Gather stage: kLanes = P/4 lanes own a row (one float4 each), a lane
// group walks its kRows rows kBatch at a time and issues the first kFirst neighbour rows of the batch plus the rows themselves
// back to back (index / weight chunks come by one coalesced load per row and are handed round by shuffle); longer rows continue
// four gathers at a time.  The aggregated tile goes through a wave-private LDS buffer into the A layout of the 16x16x4 MFMA
// (lane (i, kq) = quarter row kq of tile row i), B is read from LDS [k][i][ct] with one ds_read_b128 per k, bias + ELU in the
// epilogue.  Measured alternatives (MI355X, 10^7 x 64 De Bruijn graph): a three-tile-deep register pipeline (prefetching the next
// tiles' pointers, chunks and rows; 2 waves/SIMD) and a producer/consumer split (12 gather waves + 4 MFMA waves through an LDS
// ring) both land on the same 2.0 ms: the kernel moves ~10 GB per launch through L2 at the ~5 TB/s this chip sustains for
// 256-byte random rows, as do pp_dense_f32 + pp_spmm_f32 with their 12.6 GB in 2.4 ms.
// kHeavy: the plan has hub rows; kWide: X is 4 GiB or larger (64-bit row offsets); kThreads: 256, or 512 for the 128-wide shapes (W alone
// takes 64 KB of LDS there: one workgroup of 8 waves per CU).  kEpi 0: Y = act(tile . W^T + bias), W [Q,P] (the layer, forward).
// kEpi 1: Y = (tile . W) (*) ELU'(act_in), W [P,Q], + column sums: the INPUT GRADIENT of a layer over the transposed graph (X = dpre) for
// the widths whose weight gradient does not fit beside it in registers (pp_gcn_input_grad_f32).
#ifndef PP_FWD_WAVES
#define PP_FWD_WAVES 4
#endif
#ifndef PP_XCD_TILES
#define PP_XCD_TILES 1          // every XCD sweeps a contiguous eighth of the tiles (0: the plain round-robin order, kept for A/B measurements)
#endif
template <int P, int Q, bool kHeavy, bool kWide, int kThreads = kGcnThreads, int kEpi = 0, bool kDrop = false>
// (64 x 64: capped at 128 registers = 4 waves per SIMD — 126 VGPRs, no accumulator AGPRs, no spills; at the 146 registers the compiler
// takes when left alone the kernel runs 3 waves per SIMD and the layer is 6 % slower: 1.77 -> 1.66 ms at 10^7 rows)
__global__ __launch_bounds__(kThreads, (kThreads == 256 && P == 64 && Q == 64) ? PP_FWD_WAVES : 1) void k_gcn_forward(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ val, int64_t n_rows, const float* __restrict__ X,
                                                              const float* __restrict__ self_coef, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int act, HeavyRows heavy,
                                                              float* __restrict__ agg_out, float* __restrict__ Y,
                                                              const float* __restrict__ act_in, float* __restrict__ colsum, int64_t n_self,
                                                              DropSite drop) {
    constexpr int kLanes = P / 4, kGroups = kWave / kLanes, kRows = 16 / kGroups, KQ = P / 4, CT = Q / 16, TS = P + 4;
    constexpr int kBatch = kRows < 2 ? kRows : (kRows >= 8 ? 4 : 2);      // 128-wide rows: 2 waves/SIMD, registers to spare for a deeper gather
    constexpr int kWaves = kThreads / kWave;
    using off_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;     // byte offset of a gathered row
    __shared__ __attribute__((aligned(16))) float s_b[P * 16 * CT];
    __shared__ __attribute__((aligned(16))) float s_tile[kWaves][16 * TS];
    // B[k][j] = W[j][k] (forward) or W[k][j] (input gradient), stored [k][j]: lane i of output tile ct works on column CT*i + ct, so that
    // a lane ends up with CT CONSECUTIVE output columns per row (16-byte stores instead of 4-byte ones) and reads its CT B values at once
    for (int e = threadIdx.x; e < P * Q; e += kThreads) {
        if (kEpi == 0) {
            const int j = e / P, k = e - j * P;
            s_b[k * Q + j] = W[e];
        } else {
            s_b[e] = W[e];
        }
    }
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int g = lane / kLanes, l = lane % kLanes;
    const int i = lane & 15, kq = lane >> 4;
    float* tile = s_tile[wave];
    float bias_c[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bias_c[ct] = bias ? bias[CT * i + ct] : 0.f;
    const char* xb = (const char*)X;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const bool stream_out = n_rows * (int64_t)(Q * 4) >= kStreamFromBytes;
    // Tile order.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on die b % 8) and every die has its own L2: with the
    // plain order (tile = b * kWaves + wave, + grid) two neighbouring 64-row groups never share an L2, although on a De Bruijn graph the rows
    // of one destination block (a, .) gather from the SAME source set (., a).  Every die therefore sweeps a CONTIGUOUS eighth of the tiles.
    const int64_t n_groups = (n_tiles + kWaves - 1) / kWaves;
    const bool by_die = PP_XCD_TILES && gridDim.x % 8 == 0;
    const int64_t per_die = (n_groups + 7) / 8, die_stride = gridDim.x / 8;
    auto tile_at = [&](int64_t it) -> int64_t {
        if (by_die) {
            const int64_t q = (int64_t)(blockIdx.x >> 3) + it * die_stride;
            const int64_t t = ((int64_t)(blockIdx.x & 7) * per_die + q) * kWaves + wave;
            return (q < per_die && t < n_tiles) ? t : n_tiles;
        }
        const int64_t t = ((int64_t)blockIdx.x + it * gridDim.x) * kWaves + wave;
        return t < n_tiles ? t : n_tiles;
    };
    float col_in[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) col_in[ct] = 0.f;
    // ---- gather stage, two forms.  kWide (X of 4 GiB or more): 64-bit addresses, per-lane row pointers (the original form).  Otherwise:
    // buffer loads with 32-bit offsets; the tile's 17 row pointers, 16 self coefficients and the first 4 (index, value) pairs of all 16 rows
    // come by ONE coalesced load each (lane 4*row + slot) and are handed round with ds_bpermute; absent neighbours / rows past the end carry
    // the out-of-range offset and read as zeros.  The pointers are fetched one tile ahead.
    [[maybe_unused]] int p_next[kRows + 1];
    [[maybe_unused]] TileGather<P, kHeavy> gather(X, ptr, idx, val, self_coef, heavy, n_rows, n_self);
    if constexpr (kWide) {
        {
            const int64_t t0 = tile_at(0);
#pragma unroll
            for (int q = 0; q <= kRows; ++q) {
                const int64_t r = t0 * 16 + g * kRows + q;
                p_next[q] = t0 < n_tiles ? ptr[r < n_rows ? r : n_rows] : 0;
            }
        }

    } else {
        gather.prefetch(tile_at(0));
    }
    int64_t t_after = n_tiles;
    for (int64_t it = 0, t = tile_at(0); t < n_tiles; ++it, t = t_after) {
        t_after = tile_at(it + 1);
        if constexpr (kWide) {
            const int64_t r0 = t * 16 + g * kRows;
            int p[kRows + 1];
#pragma unroll
            for (int q = 0; q <= kRows; ++q) p[q] = p_next[q];
            {
                const int64_t tn = t_after;
#pragma unroll
                for (int q = 0; q <= kRows; ++q) {
                    const int64_t r = tn * 16 + g * kRows + q;
                    p_next[q] = tn < n_tiles ? ptr[r < n_rows ? r : n_rows] : 0;
                }
            }
            int cj[kRows], pe[kRows], hs[kRows];
            float cv[kRows], sc[kRows];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                hs[q] = (kHeavy && r0 + q < n_rows) ? heavy.slot[r0 + q] : -1;
                pe[q] = (kHeavy && hs[q] >= 0) ? p[q] : p[q + 1];                   // a hub row: its neighbour sum is already in heavy.sum
                const int mine = p[q] + l;
                const bool in = mine < pe[q];
                cj[q] = in ? idx[mine] : 0;
                cv[q] = in ? (val ? val[mine] : 1.f) : 0.f;
                sc[q] = (self_coef != nullptr && r0 + q < n_self) ? self_coef[r0 + q] : 0.f;
            }
#pragma unroll
            for (int b0 = 0; b0 < kRows; b0 += kBatch) {
                off_t off[kBatch][kFirst], self_off[kBatch];
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    const int q = b0 + qq;
                    const bool self_here = self_coef != nullptr && r0 + q < n_self;
                    const int first = __shfl(cj[q], 0, kLanes);
                    const int dummy = p[q] < pe[q] ? first : (self_here ? (int)(r0 + q) : 0);
                    self_off[qq] = (off_t)(uint32_t)(self_here ? (int)(r0 + q) : dummy) * (off_t)(P * 4) + (off_t)(16 * l);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        const int j = u == 0 ? first : __shfl(cj[q], u, kLanes);
                        off[qq][u] = (off_t)(uint32_t)(p[q] + u < pe[q] ? j : dummy) * (off_t)(P * 4) + (off_t)(16 * l);
                    }
                }
                float4 x[kBatch][kFirst], sr[kBatch];
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    sr[qq] = *(const float4*)(xb + self_off[qq]);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) x[qq][u] = *(const float4*)(xb + off[qq][u]);
                }
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    const int q = b0 + qq;
                    float4 acc = make_float4(sc[q] * sr[qq].x, sc[q] * sr[qq].y, sc[q] * sr[qq].z, sc[q] * sr[qq].w);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        const float v = __shfl(cv[q], u, kLanes);
                        acc.x += v * x[qq][u].x; acc.y += v * x[qq][u].y; acc.z += v * x[qq][u].z; acc.w += v * x[qq][u].w;
                    }
                    int my_j = cj[q];
                    float my_v = cv[q];
                    const int p0 = p[q], p1 = pe[q];
                    for (int base = p0; base < p1; base += kLanes) {      // rows with more than kFirst neighbours
                        if (base != p0) {
                            const int mine = base + l;
                            my_j = mine < p1 ? idx[mine] : 0;
                            my_v = mine < p1 ? (val ? val[mine] : 1.f) : 0.f;
                        }
                        const int cnt = p1 - base < kLanes ? p1 - base : kLanes;
                        for (int e = base == p0 ? kFirst : 0; e < cnt; e += 4) {
                            float4 y[4];
                            float v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int src_lane = (e + u) < cnt ? e + u : e;
                                const int j = __shfl(my_j, src_lane, kLanes);
                                v[u] = (e + u) < cnt ? __shfl(my_v, src_lane, kLanes) : 0.f;
                                y[u] = *(const float4*)(xb + (off_t)(uint32_t)j * (off_t)(P * 4) + (off_t)(16 * l));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc.x += v[u] * y[u].x; acc.y += v[u] * y[u].y; acc.z += v[u] * y[u].z; acc.w += v[u] * y[u].w;
                            }
                        }
                    }
                    if (p0 == p1 && sc[q] == 0.f) acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kHeavy && hs[q] >= 0) {
                        const float4 h = *(const float4*)(heavy.sum + (int64_t)hs[q] * P + 4 * l);
                        acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
                    }
                    *(float4*)(tile + (g * kRows + q) * TS + 4 * l) = acc;
                    if (agg_out != nullptr && r0 + q < n_rows) *(float4*)(agg_out + (r0 + q) * P + 4 * l) = acc;
                }
            }

        } else {
            gather.run(t, t_after, tile, agg_out);
        }
        __builtin_amdgcn_wave_barrier();
        // input gradient: the activation rows of the epilogue (lane (i, kq): rows 4*kq .., columns CT*i ..) fly during the MFMAs
        float xr[kEpi == 1 ? CT : 1][4];
        if constexpr (kEpi == 1) if (act) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = t * 16 + 4 * kq + reg;
                const float* xp = act_in + r * Q + CT * i;
#pragma unroll
                for (int c4 = 0; c4 < CT; c4 += 4) {
                    if constexpr (CT >= 4) {
                        const float4 v = r < n_rows ? *(const float4*)(xp + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        xr[c4][reg] = v.x; xr[c4 + 1][reg] = v.y; xr[c4 + 2][reg] = v.z; xr[c4 + 3][reg] = v.w;
                    } else {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) xr[ct][reg] = r < n_rows ? xp[ct] : 0.f;
                    }
                }
            }
        }
        float4 a[KQ / 4];
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) a[c] = *(const float4*)(tile + i * TS + kq * KQ + 4 * c);
        f32x4 out[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) out[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(3);                    // a wave in its MFMA burst goes first: it frees the matrix pipe sooner (-2 %)
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) {
            const float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* bp = s_b + ((kq * KQ + 4 * c + e) * 16 + i) * CT;
                float bv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bv[ct] = bp[ct];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) out[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[ct], out[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_wave_barrier();
        // C/D layout: out[ct][reg] = row 4*kq + reg of the tile, column CT*i + ct: CT consecutive columns per lane and row
        float* yp = Y + (t * 16 + 4 * kq) * Q + CT * i;
        const int rows_here = n_rows - (t * 16 + 4 * kq) < 4 ? (int)(n_rows - (t * 16 + 4 * kq)) : 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float y[CT];
            if constexpr (kEpi == 1) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    y[ct] = out[ct][reg];
                    if (act) {
                        float s = xr[ct][reg];
                        if constexpr (kDrop) {                         // act_in is the DROPPED activation: y = act_in * (1 - p) where kept
                            const bool kept = dropout_keep(drop.row0 + t * 16 + 4 * kq + reg, CT * i + ct, Q, drop.key, drop.thr);
                            s *= drop.keep;
                            y[ct] *= kept ? drop.scale : 0.f;
                        }
                        y[ct] *= s > 0.f ? 1.f : s + 1.f;              // ELU'(pre) from the stored activation
                    }
                    col_in[ct] += y[ct];                               // rows past the end aggregate nothing: 0 there
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < CT; ct += 2) {
                    if (ct + 1 < CT) {
                        pp_f32x2 v = {out[ct][reg] + bias_c[ct], out[ct + 1][reg] + bias_c[ct + 1]};
                        if (act) v = elu_fast2(v);
                        y[ct] = v[0];
                        y[ct + 1] = v[1];
                    } else {
                        const float v = out[ct][reg] + bias_c[ct];
                        y[ct] = act ? elu_fast(v) : v;
                    }
                }
            }
            if constexpr (kEpi == 0 && kDrop) {
                {                                                      // dropout of the layer OUTPUT (the next layer's input dropout), fused
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        y[ct] = dropout_keep(drop.row0 + t * 16 + 4 * kq + reg, CT * i + ct, Q, drop.key, drop.thr) ? y[ct] * drop.scale : 0.f;
                }
            }
            if (reg < rows_here) {
                if constexpr (CT >= 4) {
#pragma unroll
                    for (int c4 = 0; c4 < CT; c4 += 4) store_row_f4(yp + reg * Q + c4, make_float4(y[c4], y[c4 + 1], y[c4 + 2], y[c4 + 3]), stream_out);
                } else if constexpr (CT == 2) *(float2*)(yp + reg * Q) = make_float2(y[0], y[1]);
                else yp[reg * Q] = y[0];
            }
        }
    }
    if (kEpi == 1) if (colsum != nullptr) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = col_in[ct];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) atomicAdd(&colsum[CT * i + ct], v);
        }
    }
}

// Persistent grid = exactly the workgroups that are resident at once (registers and LDS decide; asked from the runtime once).
struct GcnArgs {
    const int32_t *ptr, *idx;
    const float* val;
    int64_t n;
    const float *X, *self_coef, *W, *bias;
    int act;
    HeavyRows heavy;
    bool wide;
    float *agg_out, *Y;
    const float* act_in;
    float* colsum;
    int64_t n_self;          // rows with a self term (rectangular partition plans: the owned rows come first, halo rows have none)
    DropSite drop;           // kEpi 0: dropout of Y; kEpi 1: the dropout that produced act_in (thr == 0: none)
};

template <int P, int Q, int kEpi>
static int launch_gcn_forward(int64_t n_tiles, hipStream_t st, const GcnArgs& a) {
    constexpr int kThreads = P * Q > 64 * 64 ? 512 : kGcnThreads;       // W of a 128-wide layer fills 32-64 KB of LDS: one workgroup per CU
    static int resident_of[2] = {0, 0};
    const int hv = a.heavy.slot != nullptr ? 1 : 0;
    if (resident_of[hv] == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hv) PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gcn_forward<P, Q, true, false, kThreads, kEpi>, kThreads, 0));
        else PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gcn_forward<P, Q, false, false, kThreads, kEpi>, kThreads, 0));
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident_of[hv] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
    }
    const int64_t resident = shared_grid(resident_of[hv]);
    int64_t blocks = ceil_div(n_tiles, kThreads / kWave);
    if (blocks > resident) blocks = resident;
    else blocks = (blocks + 7) / 8 * 8;                                 // (a multiple of the 8 XCDs: the kernel's tile order wants whole dies)
#define PP_FWD(H, WIDE)                                                                                                                  \
    do {                                                                                                                                  \
        if (a.drop.thr != 0u)                                                                                                             \
            k_gcn_forward<P, Q, H, WIDE, kThreads, kEpi, true><<<(unsigned)blocks, kThreads, 0, st>>>(a.ptr, a.idx, a.val, a.n, a.X, a.self_coef, a.W, \
                a.bias, a.act, a.heavy, a.agg_out, a.Y, a.act_in, a.colsum, a.n_self, a.drop);                                           \
        else                                                                                                                              \
            k_gcn_forward<P, Q, H, WIDE, kThreads, kEpi, false><<<(unsigned)blocks, kThreads, 0, st>>>(a.ptr, a.idx, a.val, a.n, a.X, a.self_coef, a.W, \
                a.bias, a.act, a.heavy, a.agg_out, a.Y, a.act_in, a.colsum, a.n_self, a.drop);                                           \
    } while (0)
    if (a.heavy.slot != nullptr) { if (a.wide) PP_FWD(true, true); else PP_FWD(true, false); }
    else { if (a.wide) PP_FWD(false, true); else PP_FWD(false, false); }
#undef PP_FWD
    return PP_OK;
}

template <int P>
static int launch_gcn_forward_q(int Q, int64_t n_tiles, hipStream_t st, const GcnArgs& a) {
    switch (Q) {
        case 16: return launch_gcn_forward<P, 16, 0>(n_tiles, st, a);
        case 32: return launch_gcn_forward<P, 32, 0>(n_tiles, st, a);
        case 64: return launch_gcn_forward<P, 64, 0>(n_tiles, st, a);
        default: return PP_ERR_ARG;
    }
}

// 128-wide shapes (64x128, 128x64, 128x128), forward (kEpi 0) and input gradient (kEpi 1)
template <int kEpi>
static int launch_gcn_wide(int P, int Q, int64_t n_tiles, hipStream_t st, const GcnArgs& a) {
    if (P == 128 && Q == 128) return launch_gcn_forward<128, 128, kEpi>(n_tiles, st, a);
    if (P == 64 && Q == 128) return launch_gcn_forward<64, 128, kEpi>(n_tiles, st, a);
    if (P == 128 && Q == 64) return launch_gcn_forward<128, 64, kEpi>(n_tiles, st, a);
    return PP_ERR_ARG;
}

static inline bool gcn_wide_shape(int P, int Q) {
    static const bool stream_all = getenv("PP_WIDE_STREAMED") != nullptr;      // measurement switch: 128-wide shapes on pp_gcn_wide.hip too
    return !stream_all && ((P == 128 && (Q == 64 || Q == 128)) || (P == 64 && Q == 128));
}

// =====================================================================================================
// Backward of such a layer, again in one kernel (the math of pp_spmm_f32 on the transposed CSR + pp_dense_backward_f32):
//     G      = A^T dpre + diag(self) dpre                       gathered per 16-row tile into LDS, never written to HBM
//     d_in   = (G . W) (*) ELU'(x),  colsum_in = column sums    gradient w.r.t. the PRE-activation of the layer below + its bias gradient
//     dW     = G^T x                                             contraction over the tile's rows on a second MFMA stream
// Saves the write and the re-read of G (2 of the 7 N x 64 matrix passes of the two-kernel form).
// natural-layout rows of the layer input: B operand of the dW stream and source of the ELU' epilogue.  Lane (i, kq) owns rows 4*kq .. 4*kq+3 of
// the tile and the CT consecutive columns CT*i ..: one vector load per row.
__device__ __forceinline__ void buf_store_f4_nt(buf_t r, uint32_t off, float4 v) {
    decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)off, 0, 2);                          // aux bit 1 = nt on gfx94x / gfx950
}
__device__ __forceinline__ void buf_store_f4(buf_t r, uint32_t off, float4 v) {
    decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)off, 0, 0);
}

#ifndef PP_NT_XR
#define PP_NT_XR 0
#endif
#define PP_LOAD_XR \
        _Pragma("unroll") \
        for (int reg = 0; reg < 4; ++reg) { \
            const int64_t r = t * 16 + 4 * kq + reg; \
            const float* xp = X + r * K + CT * i; \
            if constexpr (CT == 4 && !kWide) {   /* (32-bit offsets: no 64-bit lane addresses to keep across the loop) */ \
                const float4 v = buf_load_f4(rs_xin, r < n_rows ? (uint32_t)r * (uint32_t)(K * 4) + 16u * i : kBufOob); \
                xr[0][reg] = v.x; xr[1][reg] = v.y; xr[2][reg] = v.z; xr[3][reg] = v.w; \
            } else if constexpr (CT == 4) { \
                const float4 v = r < n_rows ? load_row_f4(xp, PP_NT_XR && stream_out) : make_float4(0.f, 0.f, 0.f, 0.f); \
                xr[0][reg] = v.x; xr[1][reg] = v.y; xr[2][reg] = v.z; xr[3][reg] = v.w; \
            } else { \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) xr[ct][reg] = r < n_rows ? xp[ct] : 0.f; \
            } \
        }

// 64 x 64 (round 3): capped at 170 registers = 3 waves per SIMD, with the natural-layout X rows fetched AFTER the gather (16 registers less
// while the gather's loads are in flight; 19 spilled registers remain): 2.75 -> 2.64 ms per 10^7-row layer.  Left alone the kernel takes
// 158 VGPRs + 80 AGPRs (2 waves); the cap without the late fetch spills 41 registers (3.25 ms); one gather row per batch spills none and
// gains nothing (2.75 ms).
#ifndef PP_BWD_WAVES
#define PP_BWD_WAVES 3
#endif
#ifndef PP_BWD_LATE_X
#define PP_BWD_LATE_X 1
#endif
#ifndef PP_BWD_BATCH
#define PP_BWD_BATCH 0
#endif
template <int M, int K, bool kHeavy, bool kWide, bool kDrop = false, bool kCap = false>
__global__ __launch_bounds__(kGcnThreads, (M == 64 && K == 64 && kCap) ? PP_BWD_WAVES : 1) void k_gcn_backward(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ val, int64_t n_rows, const float* __restrict__ D,
                                                             const float* __restrict__ self_coef, const float* __restrict__ X,
                                                             const float* __restrict__ W, int fuse_act, HeavyRows heavy, float* __restrict__ d_in,
                                                             float* __restrict__ colsum_in, float* __restrict__ partial_w, int64_t n_self, DropSite drop) {
    constexpr int kLanes = M / 4, kGroups = kWave / kLanes, kRows = 16 / kGroups, KQ = M / 4, MT = M / 16, CT = K / 16, TS = M + 4;
    constexpr int kBatch = kRows < 2 ? kRows : 2;
    using off_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;     // byte offset of a gathered row
    __shared__ __attribute__((aligned(16))) float s_b[M * 16 * CT];          // [k][i][ct] = W[k][ct*16 + i]
    __shared__ __attribute__((aligned(16))) float s_tile[kGcnWaves][16 * TS];
    __shared__ float s_fold[64 * 64];
    for (int e = threadIdx.x; e < M * K; e += kGcnThreads) s_b[e] = W[e];       // [k][j]: lane i of tile ct works on column CT*i + ct
    for (int e = threadIdx.x; e < 64 * 64; e += kGcnThreads) s_fold[e] = 0.f;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int g = lane / kLanes, l = lane % kLanes;
    const int i = lane & 15, kq = lane >> 4;
    float* tile = s_tile[wave];
    const char* db = (const char*)D;
    [[maybe_unused]] const buf_t rs_xin = buf_of(X), rs_din = buf_of(d_in);
    f32x4 acc_w[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc_w[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float col_in[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) col_in[ct] = 0.f;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const bool stream_out = n_rows * (int64_t)(K * 4) >= kStreamFromBytes;
    const int64_t step = (int64_t)gridDim.x * kGcnWaves;
    // gather stage as in k_gcn_forward: buffer loads + ds_bpermute (TileGather) unless D is 4 GiB or larger (kWide)
    [[maybe_unused]] TileGather<M, kHeavy, (M == 64 && K == 64 && kCap) ? PP_BWD_BATCH : 0> gather(D, ptr, idx, val, self_coef, heavy, n_rows, n_self);
    if constexpr (!kWide) gather.prefetch((int64_t)blockIdx.x * kGcnWaves + wave);
    for (int64_t t = (int64_t)blockIdx.x * kGcnWaves + wave; t < n_tiles; t += step) {
        float xr[CT][4];
        if constexpr (!(kCap && PP_BWD_LATE_X)) { PP_LOAD_XR }        // (in flight during the gather)
        if constexpr (kWide) {
            const int64_t r0 = t * 16 + g * kRows;
            int p[kRows + 1];
#pragma unroll
            for (int q = 0; q <= kRows; ++q) {
                const int64_t r = r0 + q < n_rows ? r0 + q : n_rows;
                p[q] = ptr[r];
            }
            int cj[kRows], pe[kRows], hs[kRows];
            float cv[kRows], sc[kRows];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                hs[q] = (kHeavy && r0 + q < n_rows) ? heavy.slot[r0 + q] : -1;
                pe[q] = (kHeavy && hs[q] >= 0) ? p[q] : p[q + 1];                   // a hub row: its neighbour sum is already in heavy.sum
                const int mine = p[q] + l;
                const bool in = mine < pe[q];
                cj[q] = in ? idx[mine] : 0;
                cv[q] = in ? (val ? val[mine] : 1.f) : 0.f;
                sc[q] = (self_coef != nullptr && r0 + q < n_self) ? self_coef[r0 + q] : 0.f;
            }
#pragma unroll
            for (int b0 = 0; b0 < kRows; b0 += kBatch) {
                off_t off[kBatch][kFirst], self_off[kBatch];
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    const int q = b0 + qq;
                    const bool self_here = self_coef != nullptr && r0 + q < n_self;
                    const int first = __shfl(cj[q], 0, kLanes);
                    const int dummy = p[q] < pe[q] ? first : (self_here ? (int)(r0 + q) : 0);
                    self_off[qq] = (off_t)(uint32_t)(self_here ? (int)(r0 + q) : dummy) * (off_t)(M * 4) + (off_t)(16 * l);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        const int j = u == 0 ? first : __shfl(cj[q], u, kLanes);
                        off[qq][u] = (off_t)(uint32_t)(p[q] + u < pe[q] ? j : dummy) * (off_t)(M * 4) + (off_t)(16 * l);
                    }
                }
                float4 x[kBatch][kFirst], sr[kBatch];
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    sr[qq] = *(const float4*)(db + self_off[qq]);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) x[qq][u] = *(const float4*)(db + off[qq][u]);
                }
#pragma unroll
                for (int qq = 0; qq < kBatch; ++qq) {
                    const int q = b0 + qq;
                    float4 acc = make_float4(sc[q] * sr[qq].x, sc[q] * sr[qq].y, sc[q] * sr[qq].z, sc[q] * sr[qq].w);
#pragma unroll
                    for (int u = 0; u < kFirst; ++u) {
                        const float v = __shfl(cv[q], u, kLanes);
                        acc.x += v * x[qq][u].x; acc.y += v * x[qq][u].y; acc.z += v * x[qq][u].z; acc.w += v * x[qq][u].w;
                    }
                    int my_j = cj[q];
                    float my_v = cv[q];
                    const int p0 = p[q], p1 = pe[q];
                    for (int base = p0; base < p1; base += kLanes) {      // rows with more than kFirst neighbours
                        if (base != p0) {
                            const int mine = base + l;
                            my_j = mine < p1 ? idx[mine] : 0;
                            my_v = mine < p1 ? (val ? val[mine] : 1.f) : 0.f;
                        }
                        const int cnt = p1 - base < kLanes ? p1 - base : kLanes;
                        for (int e = base == p0 ? kFirst : 0; e < cnt; e += 4) {
                            float4 y[4];
                            float v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int src_lane = (e + u) < cnt ? e + u : e;
                                const int j = __shfl(my_j, src_lane, kLanes);
                                v[u] = (e + u) < cnt ? __shfl(my_v, src_lane, kLanes) : 0.f;
                                y[u] = *(const float4*)(db + (off_t)(uint32_t)j * (off_t)(M * 4) + (off_t)(16 * l));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc.x += v[u] * y[u].x; acc.y += v[u] * y[u].y; acc.z += v[u] * y[u].z; acc.w += v[u] * y[u].w;
                            }
                        }
                    }
                    if (p0 == p1 && sc[q] == 0.f) acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kHeavy && hs[q] >= 0) {
                        const float4 h = *(const float4*)(heavy.sum + (int64_t)hs[q] * M + 4 * l);
                        acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
                    }
                    *(float4*)(tile + (g * kRows + q) * TS + 4 * l) = acc;
                }
            }
        } else {
            gather.run(t, t + step, tile, nullptr);
        }
        if constexpr (kCap && PP_BWD_LATE_X) { PP_LOAD_XR }         // (capped variant: fetched AFTER the gather, 16 registers less while its loads are in flight)
        __builtin_amdgcn_wave_barrier();
        // ---------------------------------------------------------------- G . W  ->  d_in tile
        float4 a[KQ / 4];
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) a[c] = *(const float4*)(tile + i * TS + kq * KQ + 4 * c);
        float hr[MT][4];                                              // G rows 4*kq + s, columns MT*i ..: A operand of the dW stream
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float* gp = tile + (4 * kq + reg) * TS + MT * i;
            if constexpr (MT == 4) {
                const float4 v = *(const float4*)gp;
                hr[0][reg] = v.x; hr[1][reg] = v.y; hr[2][reg] = v.z; hr[3][reg] = v.w;
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hr[mt][reg] = gp[mt];
            }
        }
        f32x4 out[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) out[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) {
            const float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* bp = s_b + (kq * KQ + 4 * c + e) * K + CT * i;
                float bv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bv[ct] = bp[ct];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) out[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[ct], out[ct], 0, 0, 0);
            }
        }
        // ---------------------------------------------------------------- dW += G_tile^T x_tile: step `reg` contracts the rows 4*kq + reg
        // (kq = the MFMA's k index); tile (mt, ct) holds dW[MT*row + mt][CT*col + ct]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc_w[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[mt][reg], xr[ct][reg], acc_w[mt][ct], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        [[maybe_unused]] float* yp = d_in + (t * 16 + 4 * kq) * K + CT * i;
        const int rows_here = n_rows - (t * 16 + 4 * kq) < 4 ? (int)(n_rows - (t * 16 + 4 * kq)) : 4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float v[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                v[ct] = out[ct][reg];
                if (fuse_act) {
                    float y = xr[ct][reg];
                    if constexpr (kDrop) {                             // X is the DROPPED activation of the layer below: y = X * (1 - p) where kept
                        const bool kept = dropout_keep(drop.row0 + t * 16 + 4 * kq + reg, CT * i + ct, K, drop.key, drop.thr);
                        y *= drop.keep;
                        v[ct] *= kept ? drop.scale : 0.f;
                    }
                    v[ct] *= y > 0.f ? 1.f : y + 1.f;
                }
                col_in[ct] += v[ct];                                  // rows past the end aggregate nothing: v == 0 there
            }
            if constexpr (CT == 4 && !kWide) {
                const uint32_t off = reg < rows_here ? (uint32_t)(t * 16 + 4 * kq + reg) * (uint32_t)(K * 4) + 16u * i : kBufOob;
                if (stream_out) buf_store_f4_nt(rs_din, off, make_float4(v[0], v[1], v[2], v[3]));
                else buf_store_f4(rs_din, off, make_float4(v[0], v[1], v[2], v[3]));
            } else if (reg < rows_here) {
                if constexpr (CT == 4) store_row_f4(yp + reg * K, make_float4(v[0], v[1], v[2], v[3]), stream_out);
                else if constexpr (CT == 2) *(float2*)(yp + reg * K) = make_float2(v[0], v[1]);
                else yp[reg * K] = v[0];
            }
        }
    }
    if (colsum_in) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = col_in[ct];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0) atomicAdd(&colsum_in[CT * i + ct], v);
        }
    }
    // fold the waves' dW through LDS in wave order: one partial [64][64] tile per workgroup (zero padded), summed by weight_grad_reduce
    for (int w = 0; w < kGcnWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) s_fold[(MT * (4 * kq + reg) + mt) * 64 + CT * i + ct] += acc_w[mt][ct][reg];
        }
        __syncthreads();
    }
    float* pw = partial_w + ((int64_t)blockIdx.x << 12);
    for (int e = threadIdx.x; e < 64 * 64; e += kGcnThreads) pw[e] = s_fold[e];
}

constexpr int64_t kGcnBackwardMaxBlocks = 256 * 4;

template <int M, int K>
static int launch_gcn_backward(int64_t n_tiles, hipStream_t st, const int32_t* ptr, const int32_t* idx, const float* val, int64_t n,
                               const float* D, const float* self_coef, const float* X, const float* W, int fuse_act, HeavyRows heavy, bool wide,
                               float* d_in, float* colsum_in, float* partial_w, int64_t* blocks_out, int64_t n_self, DropSite drop, bool cap) {
    // cap: the register-capped variant (3 waves per SIMD, 64 x 64 only) — faster where rows are short (a De Bruijn layer: ~2 neighbours per
    // row, 2.72 -> 2.60 ms at 10^7 rows), slower on long rows (the 20-neighbour first-order graph: 0.61 -> 0.70 ms); the caller decides
    // from the average row length
    cap = cap && M == 64 && K == 64 && !wide && heavy.slot == nullptr && drop.thr == 0u;
    static int resident_of[3] = {0, 0, 0};
    const int hv = cap ? 2 : (heavy.slot != nullptr ? 1 : 0);
    if (resident_of[hv] == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hv == 2) PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gcn_backward<M, K, false, false, false, true>, kGcnThreads, 0));
        else if (hv) PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gcn_backward<M, K, true, false>, kGcnThreads, 0));
        else PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gcn_backward<M, K, false, false>, kGcnThreads, 0));
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident_of[hv] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
        if (resident_of[hv] > kGcnBackwardMaxBlocks) resident_of[hv] = (int)kGcnBackwardMaxBlocks;
    }
    const int64_t resident = shared_grid(resident_of[hv]);
    int64_t blocks = ceil_div(n_tiles, kGcnWaves);
    if (blocks > resident) blocks = resident;
    *blocks_out = blocks;
#define PP_BWD(H, WIDE)                                                                                                                  \
    do {                                                                                                                                  \
        if (drop.thr != 0u)                                                                                                               \
            k_gcn_backward<M, K, H, WIDE, true><<<(unsigned)blocks, kGcnThreads, 0, st>>>(ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, d_in, \
                                                                                          colsum_in, partial_w, n_self, drop);          \
        else                                                                                                                              \
            k_gcn_backward<M, K, H, WIDE, false><<<(unsigned)blocks, kGcnThreads, 0, st>>>(ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, d_in, \
                                                                                           colsum_in, partial_w, n_self, drop);         \
    } while (0)
    if (cap)
        k_gcn_backward<M, K, false, false, false, true><<<(unsigned)blocks, kGcnThreads, 0, st>>>(ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, d_in,
                                                                                                 colsum_in, partial_w, n_self, drop);
    else if (heavy.slot != nullptr) { if (wide) PP_BWD(true, true); else PP_BWD(true, false); }
    else { if (wide) PP_BWD(false, true); else PP_BWD(false, false); }
#undef PP_BWD
    return PP_OK;
}

template <int M>
static int launch_gcn_backward_k(int K, int64_t n_tiles, hipStream_t st, const int32_t* ptr, const int32_t* idx, const float* val, int64_t n,
                                 const float* D, const float* self_coef, const float* X, const float* W, int fuse_act, HeavyRows heavy, bool wide,
                                 float* d_in, float* colsum_in, float* partial_w, int64_t* blocks_out, int64_t n_self, DropSite drop, bool cap) {
    switch (K) {
        case 16: return launch_gcn_backward<M, 16>(n_tiles, st, ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, partial_w, blocks_out, n_self, drop, cap);
        case 32: return launch_gcn_backward<M, 32>(n_tiles, st, ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, partial_w, blocks_out, n_self, drop, cap);
        case 64: return launch_gcn_backward<M, 64>(n_tiles, st, ptr, idx, val, n, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, partial_w, blocks_out, n_self, drop, cap);
        default: return PP_ERR_ARG;
    }
}


}  // namespace pp

extern "C" {

static inline bool dense_exact(int P, int Q) { return (P == 16 || P == 32 || P == 64) && (Q == 16 || Q == 32 || Q == 64); }

int pp_gcn_fused_supported(int P, int Q) { return dense_exact(P, Q) ? 1 : ((pp::gcn_wide_shape(P, Q) || pp_wide_layer_supported(P, Q)) ? 2 : 0); }

int pp_gcn_drop_supported(int P, int Q) { return (dense_exact(P, Q) || pp::gcn_wide_shape(P, Q)) ? 1 : 0; }

int pp_gcn_forward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_src, const float* X, int P,
                       const float* self_coef, const float* W, int Q, const float* bias, int act, const int32_t* heavy_slot, const float* heavy_sum,
                       float* agg_out, float* Y, pp_stream_t stream) {
    return pp_gcn_forward_drop_f32(ptr, idx, val, n_rows, n_src, X, P, self_coef, W, Q, bias, act, heavy_slot, heavy_sum, agg_out, Y, 0.0, 0, 0, 0, stream);
}

int pp_gcn_forward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_src, const float* X, int P,
                            const float* self_coef, const float* W, int Q, const float* bias, int act, const int32_t* heavy_slot,
                            const float* heavy_sum, float* agg_out, float* Y, double drop_p, int64_t drop_seed, int64_t drop_tag, int64_t drop_row0,
                            pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_gcn_forward_f32: negative size");
    PP_REQUIRE(drop_p >= 0.0 && drop_p < 1.0, PP_ERR_ARG, "pp_gcn_forward_drop_f32: p must lie in [0, 1)");
    PP_REQUIRE(drop_p == 0.0 || pp_gcn_drop_supported(P, Q), PP_ERR_ARG, "pp_gcn_forward_drop_f32: no fused dropout for layer shape %dx%d", P, Q);
    PP_REQUIRE(pp_gcn_fused_supported(P, Q), PP_ERR_ARG, "pp_gcn_forward_f32: unsupported layer shape %dx%d (supported: 16/32/64 and 64/128/256)", P, Q);
    if (!dense_exact(P, Q) && !pp::gcn_wide_shape(P, Q))          // a side of 256: weights streamed through LDS (pp_gcn_wide.hip)
        return pp_wide_layer_f32(ptr, idx, val, n_rows, n_rows, n_src, X, P, self_coef, W, 0, Q, bias, act, 0, nullptr, heavy_slot, heavy_sum, agg_out, Y,
                                 nullptr, nullptr, 0, stream);
    PP_REQUIRE(act == 0 || act == 1, PP_ERR_ARG, "pp_gcn_forward_f32: act must be 0 (none) or 1 (elu)");
    PP_REQUIRE(((uintptr_t)X | (uintptr_t)agg_out | (uintptr_t)Y) % 16 == 0, PP_ERR_ARG, "pp_gcn_forward_f32: X, Y and agg_out must be 16-byte aligned");
    PP_REQUIRE(n_src >= 0 && n_src < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_gcn_forward_f32: more than 2^31 source rows");
    // 64-bit row addresses once X (or a per-row array) does not fit the 32-bit offsets of the buffer loads
    const bool wide = (uint64_t)n_src * (uint64_t)P * 4 >= (uint64_t)pp::kBufOob || n_rows >= ((int64_t)1 << 30) - 64;
    if (n_rows == 0) return PP_OK;
    const int64_t n_tiles = pp::ceil_div(n_rows, 16);
    const pp::GcnArgs a{ptr, idx, val, n_rows, X, self_coef, W, bias, act, pp::HeavyRows{heavy_slot, heavy_sum}, wide, agg_out, Y, nullptr, nullptr, n_rows,
                        pp::drop_site(drop_p, drop_seed, drop_tag, drop_row0)};
    int rc;
    if (pp::gcn_wide_shape(P, Q)) rc = pp::launch_gcn_wide<0>(P, Q, n_tiles, st, a);
    else switch (P) {
        case 16: rc = pp::launch_gcn_forward_q<16>(Q, n_tiles, st, a); break;
        case 32: rc = pp::launch_gcn_forward_q<32>(Q, n_tiles, st, a); break;
        default: rc = pp::launch_gcn_forward_q<64>(Q, n_tiles, st, a); break;
    }
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int pp_gcn_input_grad_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                          const float* self_coef, const float* W, int K, const float* X_act, int fuse_act, const int32_t* heavy_slot,
                          const float* heavy_sum, float* d_in, float* colsum_in, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp_gcn_input_grad_drop_f32(ptr, idx, val, n_rows, n_self, D, M, self_coef, W, K, X_act, fuse_act, heavy_slot, heavy_sum, d_in, colsum_in, ws,
                                      ws_bytes, 0.0, 0, 0, 0, stream);
}

int pp_gcn_input_grad_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                               const float* self_coef, const float* W, int K, const float* X_act, int fuse_act, const int32_t* heavy_slot,
                               const float* heavy_sum, float* d_in, float* colsum_in, void* ws, size_t ws_bytes, double drop_p, int64_t drop_seed,
                               int64_t drop_tag, int64_t drop_row0, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(drop_p >= 0.0 && drop_p < 1.0, PP_ERR_ARG, "pp_gcn_input_grad_drop_f32: p must lie in [0, 1)");
    PP_REQUIRE(drop_p == 0.0 || (fuse_act && pp::gcn_wide_shape(M, K)), PP_ERR_ARG, "pp_gcn_input_grad_drop_f32: fused dropout needs fuse_act and a 128-wide shape");
    PP_REQUIRE(n_rows >= 0 && n_self >= 0 && n_self <= n_rows, PP_ERR_ARG, "pp_gcn_input_grad_f32: bad sizes");
    PP_REQUIRE(pp_gcn_fused_supported(M, K) == 2, PP_ERR_ARG, "pp_gcn_input_grad_f32: unsupported layer shape %dx%d (64/128/256 with a side > 64)", M, K);
    if (!pp::gcn_wide_shape(M, K))
        return pp_wide_layer_f32(ptr, idx, val, n_rows, n_self, n_self, D, M, self_coef, W, 1, K, nullptr, fuse_act ? 1 : 0, 1, X_act, heavy_slot, heavy_sum,
                                 nullptr, d_in, colsum_in, ws, ws_bytes, stream);
    PP_REQUIRE(n_rows == 0 || (d_in != nullptr && (!fuse_act || X_act != nullptr)), PP_ERR_ARG, "pp_gcn_input_grad_f32: d_in (and X_act with fuse_act) required");
    PP_REQUIRE(((uintptr_t)D | (uintptr_t)X_act | (uintptr_t)d_in) % 16 == 0, PP_ERR_ARG, "pp_gcn_input_grad_f32: D, X_act and d_in must be 16-byte aligned");
    PP_REQUIRE(n_rows < (int64_t)0x7fffffff, PP_ERR_TOO_LARGE, "pp_gcn_input_grad_f32: more than 2^31 rows");
    const bool wide = (uint64_t)n_rows * (uint64_t)M * 4 >= (uint64_t)pp::kBufOob || n_rows >= ((int64_t)1 << 30) - 64;
    if (colsum_in) PP_HIP(hipMemsetAsync(colsum_in, 0, (size_t)K * sizeof(float), st));
    if (n_rows == 0) return PP_OK;
    const int64_t n_tiles = pp::ceil_div(n_rows, 16);
    const pp::GcnArgs a{ptr, idx, val, n_rows, D, self_coef, W, nullptr, fuse_act ? 1 : 0, pp::HeavyRows{heavy_slot, heavy_sum}, wide, nullptr, d_in,
                        X_act, colsum_in, n_self, pp::drop_site(drop_p, drop_seed, drop_tag, drop_row0)};
    const int rc = pp::launch_gcn_wide<1>(M, K, n_tiles, st, a);
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    return PP_OK;
}

size_t pp_gcn_backward_ws_bytes(int64_t n_rows) {
    int64_t blocks = pp::ceil_div(pp::ceil_div(n_rows > 0 ? n_rows : 1, 16), pp::kGcnWaves);
    if (blocks > pp::kGcnBackwardMaxBlocks) blocks = pp::kGcnBackwardMaxBlocks;
    return pp::align_up((size_t)blocks * 4096 * sizeof(float));
}

int pp_gcn_backward_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                        const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                        const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, pp_stream_t stream) {
    return pp_gcn_backward_drop_f32(ptr, idx, val, n_rows, n_self, D, M, self_coef, X, K, W, fuse_act, heavy_slot, heavy_sum, d_in, colsum_in, dW, ws,
                                    ws_bytes, 0.0, 0, 0, 0, stream);
}

int pp_gcn_backward_drop_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, const float* D, int M,
                             const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                             const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, double drop_p,
                             int64_t drop_seed, int64_t drop_tag, int64_t drop_row0, pp_stream_t stream) {
    return pp_gcn_backward_nnz_f32(ptr, idx, val, n_rows, n_self, -1, D, M, self_coef, X, K, W, fuse_act, heavy_slot, heavy_sum, d_in, colsum_in, dW, ws,
                                   ws_bytes, drop_p, drop_seed, drop_tag, drop_row0, stream);
}

int pp_gcn_backward_nnz_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, int64_t nnz, const float* D, int M,
                            const float* self_coef, const float* X, int K, const float* W, int fuse_act, const int32_t* heavy_slot,
                            const float* heavy_sum, float* d_in, float* colsum_in, float* dW, void* ws, size_t ws_bytes, double drop_p,
                            int64_t drop_seed, int64_t drop_tag, int64_t drop_row0, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool cap = nnz >= 0 && nnz <= 8 * n_rows;            // short rows (every neighbour among the 8 prefetched pairs on average)
    PP_REQUIRE(drop_p >= 0.0 && drop_p < 1.0, PP_ERR_ARG, "pp_gcn_backward_drop_f32: p must lie in [0, 1)");
    PP_REQUIRE(drop_p == 0.0 || fuse_act, PP_ERR_ARG, "pp_gcn_backward_drop_f32: the fused dropout belongs to the activation below (fuse_act)");
    const pp::DropSite drop = pp::drop_site(drop_p, drop_seed, drop_tag, drop_row0);
    PP_REQUIRE(n_rows >= 0 && n_self >= 0 && n_self <= n_rows, PP_ERR_ARG, "pp_gcn_backward_f32: bad sizes");
    PP_REQUIRE(pp_dense_supported(M, K) == 1, PP_ERR_ARG, "pp_gcn_backward_f32: unsupported layer shape %dx%d (supported: 16/32/64)", M, K);
    PP_REQUIRE((n_rows == 0 || d_in != nullptr) && dW != nullptr, PP_ERR_ARG, "pp_gcn_backward_f32: d_in and dW are required");
    PP_REQUIRE(((uintptr_t)D) % 16 == 0, PP_ERR_ARG, "pp_gcn_backward_f32: D must be 16-byte aligned");
    // 64-bit row addresses as soon as ANY of the addressed matrices reaches 4 GiB: D is [n_rows, M], X and d_in are [n_rows, K] (with
    // M < K the gradient fitted 32-bit offsets while X / d_in did not, and their offsets wrapped)
    const bool wide = (uint64_t)n_rows * (uint64_t)(M > K ? M : K) * 4 >= (uint64_t)pp::kBufOob || n_rows >= ((int64_t)1 << 30) - 64;
    PP_REQUIRE(ws_bytes >= pp_gcn_backward_ws_bytes(n_rows), PP_ERR_WORKSPACE, "pp_gcn_backward_f32: workspace too small");
    if (colsum_in) PP_HIP(hipMemsetAsync(colsum_in, 0, (size_t)K * sizeof(float), st));
    if (n_rows == 0) {
        PP_HIP(hipMemsetAsync(dW, 0, (size_t)M * K * sizeof(float), st));
        return PP_OK;
    }
    const int64_t n_tiles = pp::ceil_div(n_rows, 16);
    const pp::HeavyRows heavy{heavy_slot, heavy_sum};
    int64_t blocks = 0;
    int rc;
    switch (M) {
        case 16: rc = pp::launch_gcn_backward_k<16>(K, n_tiles, st, ptr, idx, val, n_rows, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, (float*)ws, &blocks, n_self, drop, cap); break;
        case 32: rc = pp::launch_gcn_backward_k<32>(K, n_tiles, st, ptr, idx, val, n_rows, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, (float*)ws, &blocks, n_self, drop, cap); break;
        case 64: rc = pp::launch_gcn_backward_k<64>(K, n_tiles, st, ptr, idx, val, n_rows, D, self_coef, X, W, fuse_act, heavy, wide, d_in, colsum_in, (float*)ws, &blocks, n_self, drop, cap); break;
        default: rc = PP_ERR_ARG; break;
    }
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    return pp::weight_grad_reduce((const float*)ws, nullptr, blocks, M, K, dW, nullptr, st);
}

}  // extern "C"
