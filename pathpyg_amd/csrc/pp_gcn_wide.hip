// pathpyg_amd — GCN / dense layers wider than the 160 KB LDS can hold a weight matrix for (64/128/256-wide, any combination with a
// side > 64; BASELINE configs[4]: 256-dim features "MFMA feature GEMM"), still ONE kernel per layer:
//
//   out[n, :Q] = epi( tile[n, :P] . Wr[:Q, :P]^T )          Wr = the weight matrix with one ROW per output column
//   tile[n]    = sum_e val[e] X[idx[e]] + self[n] X[n]       (CSR aggregation, as pp_gcn_fused.hip)   or   X[n]   (dense layer, ptr == NULL)
//
// Reference code replaced: GCNConv.lin + propagate + bias + F.elu and torch.nn.Linear of DBGNN.forward (nn/dbgnn.py:104-119,131-150) and
// their autograd backward, for hidden widths up to 256.
//
// Structure (gfx950): a 256-thread workgroup = 4 waves, each wave owns a 16-row tile.  Phase 1 — every wave aggregates its tile
// (gather stage of pp_gcn_fused.hip: kLanes = P/4 lanes per row, index/value chunks by one coalesced load + shuffles, 4 neighbour rows
// + the row itself in flight per row) in batches of 4 rows per lane group, through a small wave-private LDS stage, into the A
// operand layout of v_mfma_f32_16x16x4_f32 held in P/4 VGPRs per lane (the whole 16 x P tile lives in registers).  Phase 2 — the
// weight matrix streams through LDS in chunks of 16 OUTPUT COLUMNS (16 rows of Wr = 16 x P floats, contiguous and coalesced in
// HBM/L2; double buffered, one __syncthreads per chunk): per chunk a wave runs P/4 MFMAs into ONE accumulator tile, applies the
// epilogue and stores 16 columns x 16 rows.  Chunking over output columns (not over k) keeps the accumulator at 4 registers, so the
// register budget goes to the tile (64 VGPRs at P = 256) and the gathers; 2 workgroups per CU (50 KB LDS each) overlap one
// workgroup's gather phase with the other's MFMA phase.  LDS rows are padded to P + 4 floats: the ds_read_b128 of lane (i, kq)
// at row i, column kq*P/4 + 4c then hits 16 distinct bank quads inside each of the instruction's 16-lane groups.
// The input gradient of such a layer is the same kernel over the transposed CSR with Wr = W^T (transposed once per call into the
// caller's workspace, 256 KB at most) and the ELU' / column-sum epilogue; its weight gradient is dW = dpre^T (A_hat X) on
// k_weight_grad_blocks (pp_dbgnn.hip) from the aggregated input the forward call stores.
#include "pp_common.h"
#include "pp_internal.h"

namespace pp {

using f32x4w = __attribute__((ext_vector_type(4))) float;

#ifndef PP_WIDE_OCC
#define PP_WIDE_OCC 2          // workgroups per CU of k_wide_layer (register budget 256 / 168 per lane at 2 / 3)
#endif
#ifndef PP_WIDE_STATIONARY
#define PP_WIDE_STATIONARY 1   // 256 x 256 layers on k_wide_ws (weights in registers); 0 = k_wide_layer for every shape
#endif
constexpr int kWideThreads = 256;
constexpr int kWideWaves = kWideThreads / kWave;
constexpr int kWideFirst = 4;                  // neighbours per row fetched in the first, fully overlapped, batch

__global__ __launch_bounds__(kBlock) void k_transpose_f32(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= rows * cols) return;
    const int r = e / cols, c = e - r * cols;
    out[c * rows + r] = in[e];
}

// The epilogue of ONE ROW GROUP (reg) of a 64-column block.  The block's four chunks hold the weight rows 64*blk + 4*i' + c' (i' = MFMA column
// index, c' = chunk), so after the four chunks lane (i, kq) owns, for each of its rows 4*kq + reg, the FOUR CONSECUTIVE output columns
// 64*blk + 4*i .. + 3 (res[c'][reg]): 16-byte stores instead of four 4-byte ones.
template <int Q, int kEpi>
__device__ __forceinline__ void wide_epilogue_row(const f32x4w (&res)[4], int reg, const float4 gp, int blk, int64_t t, int i, int kq, int64_t n_rows,
                                                  int act, const float* __restrict__ bias, float* __restrict__ Y, float4& csum) {
    const int64_t r = t * 16 + 4 * kq + reg;
    float v[4] = {res[0][reg], res[1][reg], res[2][reg], res[3][reg]};
    if constexpr (kEpi == 0) {
        const float4 b = *(const float4*)(bias + 64 * blk + 4 * i);          // (the kernel's LDS copy; zeros when the layer has no bias)
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            pp_f32x2 p = {v[e] + bb[e], v[e + 1] + bb[e + 1]};
            const pp_f32x2 q = elu_fast2(p);
            v[e] = act ? q[0] : p[0];
            v[e + 1] = act ? q[1] : p[1];
        }
    } else {
        const float g[4] = {gp.x, gp.y, gp.z, gp.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= (act && !(g[e] > 0.f)) ? g[e] + 1.f : 1.f;        // ELU'(pre) from the stored activation
        if (r < n_rows) { csum.x += v[0]; csum.y += v[1]; csum.z += v[2]; csum.w += v[3]; }
    }
    if (r < n_rows) *(float4*)(Y + r * Q + 64 * blk + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
}

// kEpi 0: Y = act(tile . Wr^T + bias)            (forward; optional copy of the aggregated tile to agg_out)
// kEpi 1: Y = (tile . Wr^T) (*) ELU'(act_in)     (input gradient; act_in NULL = no factor) + column sums
template <int P, int Q, int kEpi>
__global__ __launch_bounds__(kWideThreads, PP_WIDE_OCC) void k_wide_layer(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                                const float* __restrict__ val, int64_t n_rows, int64_t n_self,
                                                                const float* __restrict__ X, const float* __restrict__ self_coef,
                                                                const float* __restrict__ Wr, const float* __restrict__ bias, int act,
                                                                HeavyRows heavy, float* __restrict__ agg_out, float* __restrict__ Y,
                                                                const float* __restrict__ act_in, float* __restrict__ colsum) {
    constexpr int kLanes = P / 4, kGroups = kWave / kLanes, kRows = 16 / kGroups, KQ = P / 4, TS = P + 4, NC = Q / 16;
    constexpr int kBatch = kRows < 4 ? kRows : 4;                        // rows per lane group gathered at a time
    constexpr int kStageRows = kGroups * kBatch;
    constexpr int kChunkVec = 16 * P / 4 / kWideThreads;          // float4 per thread and weight chunk
    // two separate LDS objects for the two weight buffers: the compiler orders every LDS read behind a pending LDS-DMA into the SAME
    // object (s_waitcnt vmcnt(0) in the middle of the MFMA stream), but keeps distinct objects apart
    __shared__ __attribute__((aligned(16))) float s_w0[16 * TS];
    __shared__ __attribute__((aligned(16))) float s_w1[16 * TS];
    __shared__ __attribute__((aligned(16))) float s_stage[kWideWaves][kStageRows * TS];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int g = lane / kLanes, l = lane % kLanes;
    const int i = lane & 15, kq = lane >> 4;
    float* stage = s_stage[wave];
    const char* xb = (const char*)X;
    const bool dense = ptr == nullptr;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_groups = (n_tiles + kWideWaves - 1) / kWideWaves;
    __shared__ float s_col[kEpi == 1 ? kWideWaves : 1][kEpi == 1 ? Q : 1];          // per-wave column sums of the gradient epilogue
    __shared__ float s_bias[kEpi == 0 ? Q : 1];                                     // LDS copy: no global load inside the chunk pipeline
    if constexpr (kEpi == 0) {
        for (int e = threadIdx.x; e < Q; e += kWideThreads) s_bias[e] = bias != nullptr ? bias[e] : 0.f;
    }

    // weight chunk loader: chunk c = rows 16c .. 16c+15 of Wr.
    //  P == 256: one row = 1 KiB = one wave-wide global_load_lds_dwordx4 straight into the (padded) LDS row — no staging registers, no
    //            ds_write, and nothing the compiler can sink into the MFMA stream; wave w moves rows w, w+4, w+8, w+12.  The DMA of
    //            chunk c+1 is issued right after the barrier that publishes chunk c and lands while chunk c's MFMAs run.
    //  P < 256:  thread e handles float4 number e, e + 256, ... of the chunk through registers (written to LDS after the MFMAs).
    constexpr bool kDma = P == 256;
    float4 wnext[kDma ? 1 : kChunkVec];
#define PP_FETCH_CHUNK(BLK, CP, BUF)   /* chunk CP of block BLK: LDS row r <- weight row 64*BLK + 4*r + CP */                       \
    if constexpr (kDma) {                                                                                                          \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                                            \
            const int row_ = wave + 4 * v;                                                                                         \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wr + (size_t)(64 * (BLK) + 4 * row_ + (CP)) * P + 4 * lane), \
                                             (__attribute__((address_space(3))) void*)(((BUF) == 0 ? s_w0 : s_w1) + row_ * TS), 16, 0, 0); \
        }                                                                                                                          \
    } else {                                                                                                                       \
        _Pragma("unroll") for (int v = 0; v < kChunkVec; ++v) {                                                                    \
            const int e_ = threadIdx.x + v * kWideThreads;                                                                         \
            const int r_ = e_ / (P / 4), q4_ = e_ - r_ * (P / 4);                                                                  \
            wnext[v] = *(const float4*)(Wr + (size_t)(64 * (BLK) + 4 * r_ + (CP)) * P + 4 * q4_);                                  \
        }                                                                                                                          \
    }
#define PP_STORE_CHUNK(BUF)                                                                                                        \
    if constexpr (!kDma) {                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < kChunkVec; ++v) {                                                                    \
            const int e_ = threadIdx.x + v * kWideThreads;                                                                         \
            const int r_ = e_ / (P / 4), q4_ = e_ - r_ * (P / 4);                                                                  \
            *(float4*)(((BUF) == 0 ? s_w0 : s_w1) + r_ * TS + 4 * q4_) = wnext[v];                                                 \
        }                                                                                                                          \
    }
    static_assert(NC % 4 == 0, "output columns come in blocks of 64 = four 16-column chunks");
    constexpr int NB = NC / 4;
    PP_FETCH_CHUNK(0, 0, 0)
    PP_STORE_CHUNK(0)
    if constexpr (kEpi == 1) {
        for (int e = lane; e < Q; e += kWave) s_col[wave][e] = 0.f;
    }

    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t t = grp * kWideWaves + wave;
        const bool have_tile = t < n_tiles;
        // ------------------------------------------------------------------ phase 1: aggregate the tile into the A operand registers
        float4 a[KQ / 4];
#pragma unroll
        for (int c = 0; c < KQ / 4; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have_tile) {
            // one coalesced load each for the tile's 17 row pointers, its hub slots and self coefficients, then ONE load for the first 4
            // (index, value) pairs of all 16 rows (lane 4*row + slot): the batches below take everything by shuffle and issue their
            // row gathers without a dependent index load in between (rows with more than 4 neighbours continue chunk-wise)
            const int64_t tr = t * 16;
            const int64_t rl = tr + lane;
            const int pv = (!dense && lane <= 16) ? ptr[rl < n_rows ? rl : n_rows] : 0;
            const int hv = (heavy.slot != nullptr && lane < 16 && rl < n_rows) ? heavy.slot[rl] : -1;
            const float scv = (lane < 16 && rl < n_self) ? (dense ? 1.f : (self_coef != nullptr ? self_coef[rl] : 0.f)) : 0.f;
            int cj4;
            float cv4;
            {
                const int rr = lane >> 2, slot = lane & 3;
                const int pr = __shfl(pv, rr, kWave), pn = __shfl(pv, rr + 1, kWave);
                const bool hub = __shfl(hv, rr, kWave) >= 0;
                const int e = pr + slot;
                const bool in = !hub && e < pn;
                cj4 = in ? idx[e] : 0;
                cv4 = in ? (val ? val[e] : 1.f) : 0.f;
            }
#pragma unroll 1
            for (int b0 = 0; b0 < kRows; b0 += kBatch) {
                int p0[kBatch], p1[kBatch], hs[kBatch], jrow[kBatch][kWideFirst], srow[kBatch];
                float sc[kBatch];
#pragma unroll
                for (int q = 0; q < kBatch; ++q) {
                    const int rt = g * kRows + b0 + q;                     // row of the tile this lane group works on
                    const int64_t r = tr + rt;
                    p0[q] = __shfl(pv, rt, kWave);
                    hs[q] = __shfl(hv, rt, kWave);
                    p1[q] = hs[q] >= 0 ? p0[q] : __shfl(pv, rt + 1, kWave);      // a hub row: its neighbour sum is already in heavy.sum
                    sc[q] = __shfl(scv, rt, kWave);
                    const bool self_here = r < n_self && (dense || self_coef != nullptr);
                    const int first = __shfl(cj4, 4 * rt, kWave);
                    const int dummy = p0[q] < p1[q] ? first : (self_here ? (int)r : 0);
                    srow[q] = self_here ? (int)r : dummy;
#pragma unroll
                    for (int u = 0; u < kWideFirst; ++u) {
                        const int j = __shfl(cj4, 4 * rt + u, kWave);
                        jrow[q][u] = p0[q] + u < p1[q] ? j : dummy;
                    }
                }
                float4 x[kBatch][kWideFirst], sr[kBatch];
#pragma unroll
                for (int q = 0; q < kBatch; ++q) {
                    sr[q] = *(const float4*)(xb + (uint64_t)(uint32_t)srow[q] * (uint64_t)(P * 4) + (uint64_t)(16 * l));
#pragma unroll
                    for (int u = 0; u < kWideFirst; ++u)
                        x[q][u] = *(const float4*)(xb + (uint64_t)(uint32_t)jrow[q][u] * (uint64_t)(P * 4) + (uint64_t)(16 * l));
                }
#pragma unroll
                for (int q = 0; q < kBatch; ++q) {
                    const int rt = g * kRows + b0 + q;
                    const int64_t r = tr + rt;
                    float4 acc = make_float4(sc[q] * sr[q].x, sc[q] * sr[q].y, sc[q] * sr[q].z, sc[q] * sr[q].w);
#pragma unroll
                    for (int u = 0; u < kWideFirst; ++u) {
                        const float v = p0[q] + u < p1[q] ? __shfl(cv4, 4 * rt + u, kWave) : 0.f;
                        acc.x += v * x[q][u].x; acc.y += v * x[q][u].y; acc.z += v * x[q][u].z; acc.w += v * x[q][u].w;
                    }
                    for (int base = p0[q] + kWideFirst; base < p1[q]; base += kLanes) {      // rows with more than kWideFirst neighbours
                        const int mine = base + l;
                        const int my_j = mine < p1[q] ? idx[mine] : 0;
                        const float my_v = mine < p1[q] ? (val ? val[mine] : 1.f) : 0.f;
                        const int cnt = p1[q] - base < kLanes ? p1[q] - base : kLanes;
                        for (int e = 0; e < cnt; e += 4) {
                            float4 y[4];
                            float v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int src_lane = (e + u) < cnt ? e + u : e;
                                const int j = __shfl(my_j, src_lane, kLanes);
                                v[u] = (e + u) < cnt ? __shfl(my_v, src_lane, kLanes) : 0.f;
                                y[u] = *(const float4*)(xb + (uint64_t)(uint32_t)j * (uint64_t)(P * 4) + (uint64_t)(16 * l));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc.x += v[u] * y[u].x; acc.y += v[u] * y[u].y; acc.z += v[u] * y[u].z; acc.w += v[u] * y[u].w;
                            }
                        }
                    }
                    if (p0[q] == p1[q] && sc[q] == 0.f) acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (hs[q] >= 0) {
                        const float4 h = *(const float4*)(heavy.sum + (int64_t)hs[q] * P + 4 * l);
                        acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
                    }
                    if (r >= n_rows) acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(stage + (g * kBatch + q) * TS + 4 * l) = acc;
                    if (kEpi == 0 && agg_out != nullptr && r < n_rows) *(float4*)(agg_out + r * P + 4 * l) = acc;
                }
                __builtin_amdgcn_wave_barrier();
                // the lanes whose tile row i belongs to this batch pick up their quarter row (A layout: lane (i, kq) = row i, k-range kq)
                const int within = (i % kRows) - b0;
                if (within >= 0 && within < kBatch) {
                    const float* sp = stage + ((i / kRows) * kBatch + within) * TS + kq * KQ;
#pragma unroll
                    for (int c = 0; c < KQ / 4; ++c) a[c] = *(const float4*)(sp + 4 * c);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ------------------------------------------------------------------ phase 2: stream the weight chunks, 16 output columns at a time
        // Software pipeline over the chunks: [wait own DMA + older stores] barrier | DMA of the next chunk | one row group of the PREVIOUS
        // block's epilogue (a 16-byte store per lane) | LDS reads + 64 MFMAs of this chunk.  Stores and DMA of one iteration are covered
        // by the MFMAs of the same iteration; the results of a block wait in `res` (two sets: the block being accumulated and the one
        // whose epilogue is being drained), so no MFMA ever waits for a pending store to read its registers.
        f32x4w res[2][4];
        float4 gpv[2][4];
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int sset = 0; sset < 2; ++sset)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                res[sset][q] = f32x4w{0.f, 0.f, 0.f, 0.f};
                gpv[sset][q] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
#pragma unroll 1
        for (int blk2 = 0; blk2 < NB; blk2 += 2) {
#pragma unroll
            for (int sset = 0; sset < 2; ++sset) {
                const int blk = blk2 + sset;
                if (blk < NB) {
                    if constexpr (kEpi == 1) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int64_t r = t * 16 + 4 * kq + reg;
                            gpv[sset][reg] = (act && have_tile && r < n_rows) ? *(const float4*)(act_in + r * Q + 64 * blk + 4 * i)
                                                                              : make_float4(1.f, 1.f, 1.f, 1.f);
                        }
                    }
#pragma unroll
                    for (int cp = 0; cp < 4; ++cp) {
                        const int buf = cp & 1;
                        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): this wave's DMA rows of the chunk have landed (and its older stores are out)
                        __syncthreads();                              // the chunk is complete in buffer `buf`; the other buffer is free again
                        {
                            const int nblk = cp < 3 ? blk : (blk + 1 < NB ? blk + 1 : 0);      // (after the last chunk: chunk 0 of the next tile group)
                            PP_FETCH_CHUNK(nblk, (cp + 1) & 3, buf ^ 1)
                        }
                        if (blk > 0 && have_tile) {                   // drain one row group of the previous block
                            wide_epilogue_row<Q, kEpi>(res[sset ^ 1], cp, gpv[sset ^ 1][cp], blk - 1, t, i, kq, n_rows, act, s_bias, Y, csum);
                            if (kEpi == 1 && cp == 3) {
                                float cs[4] = {csum.x, csum.y, csum.z, csum.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    cs[e] += __shfl_xor(cs[e], 16, kWave);
                                    cs[e] += __shfl_xor(cs[e], 32, kWave);
                                    if (kq == 0) s_col[kEpi == 1 ? wave : 0][(kEpi == 1 ? 64 * (blk - 1) + 4 * i + e : 0)] += cs[e];
                                }
                                csum = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                        // two accumulators per chunk, alternating: a 16x16x4 fp32 MFMA issues every 32 cycles but its result is ready after 40 —
                        // a single dependent chain would idle the matrix pipe a fifth of the time
                        f32x4w acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        const float* wp = (buf == 0 ? s_w0 : s_w1) + i * TS + kq * KQ;
                        // register double buffer for the B operands: the LDS reads of the next 4 k-quads are issued BEFORE the 16 MFMAs of the
                        // current ones (the scheduling barriers pin that order; left alone the compiler reads two quads, waits out the LDS
                        // latency with an idle matrix pipe, and repeats)
                        constexpr int kQuads = KQ / 4, kStep = kQuads < 4 ? kQuads : 4;
                        float4 bc[kStep], bn[kStep];
#pragma unroll
                        for (int u = 0; u < kStep; ++u) bc[u] = *(const float4*)(wp + 4 * u);
#pragma unroll
                        for (int cc = 0; cc < kQuads; cc += kStep) {
                            if (cc + kStep < kQuads) {
#pragma unroll
                                for (int u = 0; u < kStep; ++u) bn[u] = *(const float4*)(wp + 4 * (cc + kStep + u));
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < kStep; ++u) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cc + u].x, bc[u].x, acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cc + u].y, bc[u].y, acc1, 0, 0, 0);
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cc + u].z, bc[u].z, acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cc + u].w, bc[u].w, acc1, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < kStep; ++u) bc[u] = bn[u];
                        }
                        PP_STORE_CHUNK(buf ^ 1)
                        res[sset][cp] = acc0 + acc1;
                    }
                }
            }
        }
        if (have_tile) {                                              // the last block's epilogue
            constexpr int kLast = (NB - 1) & 1;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                wide_epilogue_row<Q, kEpi>(res[kLast], reg, gpv[kLast][reg], NB - 1, t, i, kq, n_rows, act, s_bias, Y, csum);
            if constexpr (kEpi == 1) {
                float cs[4] = {csum.x, csum.y, csum.z, csum.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[e] += __shfl_xor(cs[e], 16, kWave);
                    cs[e] += __shfl_xor(cs[e], 32, kWave);
                    if (kq == 0) s_col[wave][64 * (NB - 1) + 4 * i + e] += cs[e];
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // (the DMA of the chunk nobody will use must not outlive the workgroup's LDS)
#undef PP_FETCH_CHUNK
#undef PP_STORE_CHUNK
    if constexpr (kEpi == 1) {
        if (colsum != nullptr) {
            __syncthreads();
            for (int e = threadIdx.x; e < Q; e += kWideThreads) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < kWideWaves; ++w) v += s_col[w][e];
                atomicAdd(&colsum[e], v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 256 x 256 layers, weights STATIONARY in registers (round 4).  k_wide_layer streams the 256 KB weight matrix through LDS once per 64 rows
// (one workgroup barrier per 64 MFMAs of a wave) and runs the gathers and the matrix stage of a workgroup one after the other.  Here a
// 512-thread workgroup (one per CU, 8 waves, 256 registers per lane) keeps the matrix in registers for the whole launch — wave w holds the
// 32 weight rows of its output columns [32 w, 32 w + 32) as B operands, 128 VGPRs — and the aggregated 64 x 256 tile moves through a
// double-buffered LDS tile: while the waves multiply tile n out of one buffer (every wave reads all 64 rows; 512 MFMAs per wave and tile),
// each wave gathers 8 rows of tile n + 1 into the other: the rows themselves (self term) by LDS-DMA straight into their slots at the start
// of the tile step, the first four neighbours of a row by register loads issued 64 MFMAs before their use.  One workgroup barrier per
// tile, no weight traffic after the prologue.  The (index, value) pairs of a tile are fetched one tile further ahead (row pointers at the
// start of a tile step, pairs in its middle), so no gather waits for an index load.
// MFMA column i of column tile ct is output column 32 w + 2 i + ct: a lane ends up with two adjacent columns of a row (8-byte stores,
// 128 contiguous bytes per row and wave).  k order inside a dot product: lane (i, kq) contracts k = 16 j + 4 kq + c at step (j, c).
constexpr int kWsThreads = 512, kWsWaves = kWsThreads / kWave, kWsTile = 64, kWsRowsPerWave = kWsTile / kWsWaves;
// MEASUREMENT SWITCHES (never set in the product build; tools/probes/wide_ws_counters.sh builds the variants and collects the SQ counters of
// DESIGN §5 from THIS source): bit 0 = no gather at all (no index loads, no neighbour / self rows: the matrix stream runs on whatever the LDS
// tiles hold), bit 1 = no stores of Y / agg_out.  Results are meaningless with either bit set.
#ifndef PP_WS_DBG
#define PP_WS_DBG 0
#endif
#ifndef PP_WS_REAL_BRANCHES
#define PP_WS_REAL_BRANCHES 0   // 1: real branches around the per-neighbour FMAs instead of the compiler's if-conversion (6 VALU per neighbour SLOT whether the
//                                 neighbour exists or not).  Measured (round 5, same box, 10^7 rows): 272 VALU instructions fewer per tile step and SLOWER — forward
//                                 13.9 -> 15.6 ms: the branches cut the MFMA stream's schedule; kept as a switch for A/B runs only
#endif
#ifndef PP_WS_INIT_ROWS
#define PP_WS_INIT_ROWS (-1)    // every neighbour slot of a row in flight zeroed at issue (20 v_mov per row that nothing reads): 1 always, 0 never, -1 (default) in the
//                                 forward form only — measured (round 5, same box): without them the input gradient runs 15.28 -> 14.57 ms, the forward layer
//                                 13.9 -> 15.1 ms (what the zeroing buys the forward form is its schedule, not its values)
#endif

struct WsRowIndex {          // per-lane index state of ONE tile for this wave's 8 rows
    int pv;                  // lanes 0..8: row pointers
    int hv;                  // lanes 0..7: hub slot or -1
    float scv;               // lanes 0..7: self coefficient (0 = no self term)
    int cj4;                 // lanes 4 row + slot: the row's first four neighbours
    float cv4;
};

template <int kEpi>
__global__ __launch_bounds__(kWsThreads, 1) void k_wide_ws(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ val, int64_t n_rows, int64_t n_self,
                                                            const float* __restrict__ X, const float* __restrict__ self_coef,
                                                            const float* __restrict__ Wr, const float* __restrict__ bias, int act,
                                                            HeavyRows heavy, float* __restrict__ agg_out, float* __restrict__ Y,
                                                            const float* __restrict__ act_in, float* __restrict__ colsum) {
    constexpr int P = 256, Q = 256, TS = P + 4;
    __shared__ __attribute__((aligned(16))) float s_tile0[kWsTile * TS];
    __shared__ __attribute__((aligned(16))) float s_tile1[kWsTile * TS];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const bool dense = ptr == nullptr;
    const bool self_any = dense || self_coef != nullptr;
    const char* xb = (const char*)X;
    const int64_t n_tiles = (n_rows + kWsTile - 1) / kWsTile;
    const int c0 = 32 * wave + 2 * i;                                     // this lane's two output columns: c0, c0 + 1
    const uint32_t out_off = (uint32_t)(4 * kq * 256 + c0) * 4u;         // byte offset of (row 4 kq, column c0) inside a 16-row block of Y / act_in
    float bias0 = 0.f, bias1 = 0.f;
    if (kEpi == 0 && bias != nullptr) { bias0 = bias[c0]; bias1 = bias[c0 + 1]; }
    float4 wreg[2][16];                                                  // B operands: [column tile][16-float block of k]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) wreg[ct][j] = *(const float4*)(Wr + (size_t)(c0 + ct) * P + 16 * j + 4 * kq);
    float cs0 = 0.f, cs1 = 0.f;                                          // column sums of the gradient epilogue (this lane's rows)

    auto load_pointers = [&](int64_t t, WsRowIndex& s) {
        const int64_t rl = t * kWsTile + kWsRowsPerWave * wave + lane;
        const bool live = t < n_tiles;
        s.pv = (live && !dense && lane <= kWsRowsPerWave) ? ptr[rl < n_rows ? rl : n_rows] : 0;
        s.hv = (live && heavy.slot != nullptr && lane < kWsRowsPerWave && rl < n_rows) ? heavy.slot[rl] : -1;
        s.scv = (live && lane < kWsRowsPerWave && rl < n_self && rl < n_rows) ? (dense ? 1.f : (self_coef != nullptr ? self_coef[rl] : 0.f)) : 0.f;
    };
    auto load_pairs = [&](WsRowIndex& s) {
        const int rr = (lane >> 2) & (kWsRowsPerWave - 1), slot = lane & 3;
        const int pr = __shfl(s.pv, rr, kWave), pn = __shfl(s.pv, rr + 1, kWave);
        const bool hub = __shfl(s.hv, rr, kWave) >= 0;
        const int e = pr + slot;
        const bool in = lane < 4 * kWsRowsPerWave && !hub && e < pn;
        s.cj4 = in ? idx[e] : 0;
        s.cv4 = in ? (val ? val[e] : 1.f) : 0.f;
    };

    struct RowFlight {       // one row between the issue of its neighbour loads and their use
        float4 x[4];
        float v[4], sc;
        int p0, p1, hs;
        int64_t r;
        bool self_here;
    };
    // the rows THEMSELVES (self term) go straight into their slots of the tile being filled, by LDS-DMA (no registers): all 8 of a wave at the
    // start of a tile step — 8 KB per wave in flight behind the first MFMAs
    auto fetch_self_rows = [&](int64_t t, float* fill) {
#pragma unroll
        for (int q = 0; q < kWsRowsPerWave; ++q) {
            const int64_t r = t * kWsTile + kWsRowsPerWave * wave + q;
            if (self_any && r < n_self && r < n_rows)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + (uint64_t)r * (uint64_t)(P * 4) + (uint64_t)(16 * lane)),
                                                 (__attribute__((address_space(3))) void*)(fill + (kWsRowsPerWave * wave + q) * TS), 16, 0, 0);
        }
    };
    auto issue_row = [&](int q, int64_t t, const WsRowIndex& s, RowFlight& f) {
        f.r = t * kWsTile + kWsRowsPerWave * wave + q;
        f.p0 = __builtin_amdgcn_readlane(s.pv, q);
        f.hs = __builtin_amdgcn_readlane(s.hv, q);
        f.p1 = f.hs >= 0 ? f.p0 : __builtin_amdgcn_readlane(s.pv, q + 1);       // a hub row: its neighbour sum is already in heavy.sum
        f.sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.scv), q));
        f.self_here = self_any && f.r < n_self && f.r < n_rows;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (PP_WS_INIT_ROWS > 0 || (PP_WS_INIT_ROWS < 0 && kEpi == 0)) {
                f.x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                f.v[u] = 0.f;
            }
            if (f.p0 + u < f.p1) {                                          // (wave-uniform; finish_row reads x[u] / v[u] under the same test only)
                const int j = __builtin_amdgcn_readlane(s.cj4, 4 * q + u);
                f.v[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.cv4), 4 * q + u));
                f.x[u] = *(const float4*)(xb + (uint64_t)(uint32_t)j * (uint64_t)(P * 4) + (uint64_t)(16 * lane));
            }
        }
    };
    auto finish_row = [&](int q, const RowFlight& f, float* fill) {
        float* slot = fill + (kWsRowsPerWave * wave + q) * TS + 4 * lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f.self_here) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the row's DMA has landed (the compiler does not always order this read behind it)
            const float4 sr = *(const float4*)slot;
            acc = make_float4(f.sc * sr.x, f.sc * sr.y, f.sc * sr.z, f.sc * sr.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (f.p0 + u < f.p1) {                                          // (wave-uniform: an absent neighbour costs no VALU slot — they are paid in matrix time)
#if PP_WS_REAL_BRANCHES
                asm volatile("" ::: "memory");                              // (keeps the compiler from turning the branch into 2 FMAs + 4 selects that always run)
#endif
                acc.x += f.v[u] * f.x[u].x; acc.y += f.v[u] * f.x[u].y; acc.z += f.v[u] * f.x[u].z; acc.w += f.v[u] * f.x[u].w;
            }
        for (int base = f.p0 + 4; base < f.p1; base += kWave) {             // rows with more than four neighbours
            const int mine = base + lane;
            const int my_j = mine < f.p1 ? idx[mine] : 0;
            const float my_v = mine < f.p1 ? (val ? val[mine] : 1.f) : 0.f;
            const int cnt = f.p1 - base < kWave ? f.p1 - base : kWave;
            for (int e = 0; e < cnt; e += 4) {
                float4 y[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int src_lane = (e + u) < cnt ? e + u : e;
                    const int j = __builtin_amdgcn_readlane(my_j, src_lane);
                    v[u] = (e + u) < cnt ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), src_lane)) : 0.f;
                    y[u] = *(const float4*)(xb + (uint64_t)(uint32_t)j * (uint64_t)(P * 4) + (uint64_t)(16 * lane));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += v[u] * y[u].x; acc.y += v[u] * y[u].y; acc.z += v[u] * y[u].z; acc.w += v[u] * y[u].w;
                }
            }
        }
        if (f.hs >= 0) {
            const float4 h = *(const float4*)(heavy.sum + (int64_t)f.hs * P + 4 * lane);
            acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
        }
        if (f.r >= n_rows) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)slot = acc;
        if (kEpi == 0 && agg_out != nullptr && f.r < n_rows) *(float4*)(agg_out + f.r * P + 4 * lane) = acc;
    };

    // prologue: the first tile, gathered without anything to hide behind; the index state of the second
    int64_t t = blockIdx.x;
    WsRowIndex cur{}, nxt{};
    if (!(PP_WS_DBG & 1) && t < n_tiles) {
        load_pointers(t, cur);
        load_pairs(cur);
        fetch_self_rows(t, s_tile0);
#pragma unroll 1
        for (int q = 0; q < kWsRowsPerWave; ++q) {
            RowFlight f;
            issue_row(q, t, cur, f);
            finish_row(q, f, s_tile0);
        }
    }
    if (!(PP_WS_DBG & 1)) {
        load_pointers(t + gridDim.x, cur);
        load_pairs(cur);
    }
    __syncthreads();

    // one tile step: multiply tile t out of `tile`, gather tile t + grid into `fill` (called with the two LDS buffers in both roles: each call
    // site sees which object its DMA writes, so that the LDS reads of the MFMA stream never wait for it)
    auto tile_step = [&](const float* tile, float* fill) {
        const int64_t tn = t + gridDim.x;                                 // the tile being gathered (index state: cur), tn + grid: being indexed (nxt)
        const bool gather = !(PP_WS_DBG & 1) && tn < n_tiles;
        if (!(PP_WS_DBG & 1)) load_pointers(tn + gridDim.x, nxt);
        if (gather) fetch_self_rows(tn, fill);
        // 4 row tiles x 8 steps of 16 MFMAs (two 16-float k blocks x 4 x two column tiles; two accumulator chains).  The A operands of step
        // s + 1 are read from LDS before the MFMAs of step s.  Gather rows 2 rt and 2 rt + 1 of the next tile ride on row tile rt: loads
        // issued before steps 0 / 4, consumed behind steps 3 / 7 (64 MFMAs later).
        float4 a[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) a[0][jj] = *(const float4*)(tile + i * TS + 16 * jj + 4 * kq);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            f32x4w acc[2] = {f32x4w{0.f, 0.f, 0.f, 0.f}, f32x4w{0.f, 0.f, 0.f, 0.f}};
            pp_f32x2 gp[4];
            RowFlight f;
            const int64_t row0 = t * kWsTile + 16 * rt;                   // first row of this row tile
            const bool full = row0 + 16 <= n_rows;                        // (wave-uniform: the gradient epilogue tests no bound per lane on the common path)
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (gather && (st & 3) == 0) issue_row(2 * rt + (st >> 2), tn, cur, f);
                if (!(PP_WS_DBG & 1) && rt == 2 && st == 0) load_pairs(nxt);
                if (kEpi == 1 && st == 6) {                               // the activations the row tile's epilogue multiplies by, 32 MFMAs ahead
                    const char* ab = (const char*)(act_in + row0 * Q);    // (scalar row-tile base + one constant lane offset: no per-row address VALU)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        gp[reg] = pp_f32x2{1.f, 1.f};
                        if (act && (full || row0 + 4 * kq + reg < n_rows)) gp[reg] = *(const pp_f32x2*)(ab + out_off + reg * Q * 4);
                    }
                }
                const int cb = st & 1;
                if (!(rt == 3 && st == 7)) {
                    const int nrt = st == 7 ? rt + 1 : rt, nst = st == 7 ? 0 : st + 1;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) a[cb ^ 1][jj] = *(const float4*)(tile + (16 * nrt + i) * TS + 16 * (2 * nst + jj) + 4 * kq);
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * st + jj;
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].x, wreg[0][j].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].x, wreg[1][j].x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].y, wreg[0][j].y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].y, wreg[1][j].y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].z, wreg[0][j].z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].z, wreg[1][j].z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].w, wreg[0][j].w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][jj].w, wreg[1][j].w, acc[1], 0, 0, 0);
                }
                if (gather && (st & 3) == 3) finish_row(2 * rt + (st >> 2), f, fill);
            }
            // the row tile's epilogue: 4 rows x 2 adjacent columns per lane
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = t * kWsTile + 16 * rt + 4 * kq + reg;
                pp_f32x2 v = {acc[0][reg], acc[1][reg]};
                if constexpr (kEpi == 0) {
                    v += pp_f32x2{bias0, bias1};
                    if (act) v = elu_fast2(v);
                } else {
                    const pp_f32x2 g = gp[reg];
                    v[0] *= (act && !(g[0] > 0.f)) ? g[0] + 1.f : 1.f;          // ELU'(pre) from the stored activation
                    v[1] *= (act && !(g[1] > 0.f)) ? g[1] + 1.f : 1.f;
                    if (full || r < n_rows) {
                        cs0 += v[0];
                        cs1 += v[1];
                        if (!(PP_WS_DBG & 2)) *(pp_f32x2*)((char*)(Y + row0 * Q) + out_off + reg * Q * 4) = v;
                    }
                }
                if (kEpi == 0 && r < n_rows) {
                    if (!(PP_WS_DBG & 2)) *(pp_f32x2*)(Y + r * Q + c0) = v;
                    else if (v[0] == 1.2345e-30f) Y[r * Q + c0] = v[1];      // (keeps the value alive without a store that ever happens)
                }
            }
        }
        cur = nxt;
        __syncthreads();
        t += gridDim.x;
    };
    while (t < n_tiles) {
        tile_step(s_tile0, s_tile1);
        if (t >= n_tiles) break;
        tile_step(s_tile1, s_tile0);
    }
    if constexpr (kEpi == 1) {
        if (colsum != nullptr) {
            cs0 += __shfl_xor(cs0, 16, kWave); cs0 += __shfl_xor(cs0, 32, kWave);
            cs1 += __shfl_xor(cs1, 16, kWave); cs1 += __shfl_xor(cs1, 32, kWave);
            if (kq == 0) { atomicAdd(&colsum[c0], cs0); atomicAdd(&colsum[c0 + 1], cs1); }
        }
    }
}

struct WideArgs {
    const int32_t *ptr, *idx;
    const float* val;
    int64_t n_rows, n_self;
    const float *X, *self_coef, *Wr, *bias;
    int act;
    HeavyRows heavy;
    float *agg_out, *Y;
    const float* act_in;
    float* colsum;
};

template <int P, int Q, int kEpi>
static int launch_wide(hipStream_t st, const WideArgs& a) {
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        PP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wide_layer<P, Q, kEpi>, kWideThreads, 0));
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
    }
    int64_t blocks = ceil_div(ceil_div(a.n_rows, 16), kWideWaves);
    if (blocks > shared_grid(resident)) blocks = shared_grid(resident);
    k_wide_layer<P, Q, kEpi><<<(unsigned)blocks, kWideThreads, 0, st>>>(a.ptr, a.idx, a.val, a.n_rows, a.n_self, a.X, a.self_coef, a.Wr, a.bias, a.act,
                                                                         a.heavy, a.agg_out, a.Y, a.act_in, a.colsum);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <int P, int kEpi>
static int launch_wide_q(int Q, hipStream_t st, const WideArgs& a) {
    switch (Q) {
        case 64: return launch_wide<P, 64, kEpi>(st, a);
        case 128: return launch_wide<P, 128, kEpi>(st, a);
        case 256: return launch_wide<P, 256, kEpi>(st, a);
        default: return PP_ERR_ARG;
    }
}

// 256 x 256: the weight-stationary kernel, one workgroup per CU
template <int kEpi>
static int launch_wide_ws(hipStream_t st, const WideArgs& a) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus = n > 0 ? n : 256;
    }
    int64_t blocks = ceil_div(a.n_rows, kWsTile);
    if (blocks > shared_grid(cus)) blocks = shared_grid(cus);
    k_wide_ws<kEpi><<<(unsigned)blocks, kWsThreads, 0, st>>>(a.ptr, a.idx, a.val, a.n_rows, a.n_self, a.X, a.self_coef, a.Wr, a.bias, a.act, a.heavy,
                                                             a.agg_out, a.Y, a.act_in, a.colsum);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <int kEpi>
static int launch_wide_pq(int P, int Q, hipStream_t st, const WideArgs& a) {
    if (P == 256 && Q == 256 && PP_WIDE_STATIONARY && ((uintptr_t)a.Y | (uintptr_t)a.act_in) % 8 == 0) return launch_wide_ws<kEpi>(st, a);
    switch (P) {
        case 64: return launch_wide_q<64, kEpi>(Q, st, a);
        case 128: return launch_wide_q<128, kEpi>(Q, st, a);
        case 256: return launch_wide_q<256, kEpi>(Q, st, a);
        default: return PP_ERR_ARG;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Dense mode of the wide layers as a plain LDS-staged GEMM (round 3):  Y[n, :Q] = epi( X[n, :P] . Wr[:Q, :P]^T ).
// k_wide_layer keeps a wave's 16 x P tile in registers and streams W 16 output columns at a time: one workgroup barrier per 64 MFMAs of a
// wave, an epilogue per chunk.  Here a 1024-thread workgroup owns 16 / (Q/64) row blocks x Q/64 column blocks (wave (rb, cb) = 16 or 32
// rows x 64 columns, four accumulator tiles per row tile), the contraction runs in chunks of 32 (64 for the 128-column shapes): both operand
// chunks are copied into LDS once per workgroup (weights from a chunk-major copy, rows permuted against bank conflicts; fetched two chunk
// times ahead into two register sets, parked in the other LDS buffer one chunk time ahead, across row groups too), every wave takes its
// operands with ds_read_b128 (lane (i, kq) contracts its own consecutive k of the chunk: the order inside a dot product is free), one barrier
// per chunk, ONE epilogue per row group with 16-byte stores (column tile ct of lane i = column 4*i + ct).
// Measured at 10^7 rows (tools/probes/dense_wide.py, ms before -> after): 256 x 256 13.84 -> 12.7 (103 TFLOP/s), 128 x 128 5.10 -> 3.31,
// 256 -> 128 9.16 -> 6.08, 128 -> 256 8.42 -> 7.1, 64 -> 256 4.89 -> 4.15, 256 -> 64 5.61 -> 3.52.  What did NOT move it further (each built and
// measured): conflict-free B rows, two row tiles per wave at 128 columns, the second prefetch stage, the chunk-major weights — the ISA of the
// chunk loop is 32 back-to-back MFMAs, what is left is the workgroup barrier per chunk (64-wide chunks: +10 %).
// chunk-major copy of the weights for k_dense_lds: Wc[c][q][kk] = (row q of Wr)[32*c + kk].  A chunk read straight from Wr touches 256 lines
// that sit 1 KiB apart — a few L2 channels serve every CU at once; chunk-major it is one contiguous 32 KiB block.
__global__ __launch_bounds__(kBlock) void k_chunk_major_w(const float* __restrict__ W, int P, int Q, int w_is_kq, int kc, float* __restrict__ out) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= P * Q) return;
    const int c = e / (Q * kc), rem = e - c * (Q * kc), q = rem / kc, kk = rem - q * kc;
    const int k = kc * c + kk;
    out[e] = w_is_kq ? W[(size_t)k * Q + q] : W[(size_t)q * P + k];
}

constexpr int kGemmThreads = 1024;
#ifndef PP_GEMM_KC
#define PP_GEMM_KC 64
#endif
constexpr int kGemmChunk = 32;                       // contraction chunk of the 256-column shapes
constexpr int kGemmChunkNarrow = PP_GEMM_KC;          // .. and of the 128-column shapes with P >= 128 (their LDS leaves room for 64: one barrier per
                                                      // 64 MFMAs of a wave instead of 32 — 128 x 128 at 10^7 rows 3.70 -> 3.31 ms)

#ifndef PP_GEMM_RT
#define PP_GEMM_RT 2
#endif
constexpr int kGemmRowTiles = PP_GEMM_RT;             // 16-row tiles per wave: every B operand read from LDS feeds this many MFMAs

template <int P, int Q, int kEpi>
__global__ __launch_bounds__(kGemmThreads, 4) void k_dense_lds(const float* __restrict__ X, int64_t n_rows, const float* __restrict__ Wr,
                                                             const float* __restrict__ bias, int act, float* __restrict__ Y,
                                                             const float* __restrict__ act_in, float* __restrict__ colsum) {
    constexpr int RT = Q == 256 ? kGemmRowTiles : 1, CB = Q / 64, RB = 16 / CB, ROWS = 16 * RT * RB, KC = (Q == 128 && P >= 128) ? kGemmChunkNarrow : kGemmChunk, NCH = P / KC,
                  XS = KC + 4, WS = KC + 4, H = KC / 16;
    constexpr int NX = ROWS * KC / 4, NW = Q * KC / 4;                  // float4 per chunk
    constexpr int LX = (NX + kGemmThreads - 1) / kGemmThreads, LW = (NW + kGemmThreads - 1) / kGemmThreads;
    __shared__ __attribute__((aligned(16))) float s_x[2][ROWS * XS];
    __shared__ __attribute__((aligned(16))) float s_w[2][Q * WS];
    const int lane = lane_id(), wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    const int rb = wave / CB, cb = wave % CB;
    const int64_t n_groups = (n_rows + ROWS - 1) / ROWS;
    // two register sets for the chunks in flight: a chunk is fetched TWO chunk times before its MFMAs (its X rows come from HBM, ~2 us
    // away; one chunk time is 1.7 us) and parked in LDS one chunk time before
    float4 rx0[LX], rw0[LW], rx1[LX], rw1[LW];
    auto gload = [&](float4 (&rx)[LX], float4 (&rw)[LW], int64_t grp, int c) {
        const int64_t r0 = grp * ROWS;
#pragma unroll
        for (int l = 0; l < LX; ++l) {
            const int e = threadIdx.x + l * kGemmThreads;
            const int row = e / (KC / 4), k4 = e % (KC / 4);
            const int64_t r = r0 + row;
            rx[l] = (e < NX && r < n_rows) ? *(const float4*)(X + r * P + c * KC + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int l = 0; l < LW; ++l) {
            const int e = threadIdx.x + l * kGemmThreads;
            const int q = e / (KC / 4), k4 = e % (KC / 4);
            rw[l] = e < NW ? *(const float4*)(Wr + (size_t)c * (Q * KC) + (size_t)q * KC + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);   // (Wr: chunk-major)
        }
    };
    auto sstore = [&](const float4 (&rx)[LX], const float4 (&rw)[LW], int buf) {
#pragma unroll
        for (int l = 0; l < LX; ++l) {
            const int e = threadIdx.x + l * kGemmThreads;
            if (e < NX) *(float4*)(&s_x[buf][(e / (KC / 4)) * XS + 4 * (e % (KC / 4))]) = rx[l];
        }
#pragma unroll
        for (int l = 0; l < LW; ++l) {
            const int e = threadIdx.x + l * kGemmThreads;
            // weight row q = 64*blk + 4*i' + ct goes to LDS row 64*blk + 16*ct + i': lane i of column tile ct reads rows a lane apart (no bank
            // conflicts) and still owns 4 consecutive output columns
            if (e < NW) {
                const int q = e / (KC / 4);
                const int row = (q & ~63) + 16 * (q & 3) + ((q & 63) >> 2);
                *(float4*)(&s_w[buf][row * WS + 4 * (e % (KC / 4))]) = rw[l];
            }
        }
    };
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kEpi == 0 && bias != nullptr) bias4 = *(const float4*)(bias + 64 * cb + 4 * i);
    static_assert(NCH % 2 == 0, "chunks are walked in pairs (two register sets)");
    if ((int64_t)blockIdx.x >= n_groups) return;
    gload(rx0, rw0, blockIdx.x, 0);
    sstore(rx0, rw0, 0);
    gload(rx1, rw1, blockIdx.x, 1);
    __syncthreads();
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        f32x4w acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4w{0.f, 0.f, 0.f, 0.f};
        auto compute = [&](int buf) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 a4[RT], b4[4];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a4[rt] = *(const float4*)(&s_x[buf][(16 * (RT * rb + rt) + i) * XS + (KC / 4) * kq + 4 * h]);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) b4[ct] = *(const float4*)(&s_w[buf][(64 * cb + 16 * ct + i) * WS + (KC / 4) * kq + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float av = e == 0 ? a4[rt].x : (e == 1 ? a4[rt].y : (e == 2 ? a4[rt].z : a4[rt].w));
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            const float bv = e == 0 ? b4[ct].x : (e == 1 ? b4[ct].y : (e == 2 ? b4[ct].z : b4[ct].w));
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[rt][ct], 0, 0, 0);
                        }
                    }
            }
        };
        // chunk c (even) sits in LDS buffer 0, chunk c + 1 in registers set 1; (grp2, c2) = the chunk two ahead of the one being multiplied
#pragma unroll 1
        for (int c = 0; c < NCH; c += 2) {
            {
                const bool wrap = c + 2 >= NCH;
                const int64_t g2 = wrap ? grp + gridDim.x : grp;
                if (g2 < n_groups) gload(rx0, rw0, g2, wrap ? c + 2 - NCH : c + 2);
                compute(0);
                sstore(rx1, rw1, 1);                       // (chunk c + 1 always exists: NCH is even)
                __syncthreads();
            }
            {
                const bool wrap = c + 3 >= NCH;
                const int64_t g3 = wrap ? grp + gridDim.x : grp;
                const bool more = (wrap ? grp + gridDim.x : grp) < n_groups;
                if (g3 < n_groups) gload(rx1, rw1, g3, wrap ? c + 3 - NCH : c + 3);
                compute(1);
                if (c + 2 < NCH || more) sstore(rx0, rw0, 0);
                __syncthreads();
            }
        }
        // epilogue of the row group: lane (i, kq) owns rows 16*(RT*rb + rt) + 4*kq + reg and the four consecutive columns 64*cb + 4*i ..
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = grp * ROWS + 16 * (RT * rb + rt) + 4 * kq + reg;
                if (r >= n_rows) continue;
                float v[4] = {acc[rt][0][reg], acc[rt][1][reg], acc[rt][2][reg], acc[rt][3][reg]};
                if constexpr (kEpi == 0) {
                    const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        pp_f32x2 p = {v[e] + bb[e], v[e + 1] + bb[e + 1]};
                        const pp_f32x2 q = elu_fast2(p);
                        v[e] = act ? q[0] : p[0];
                        v[e + 1] = act ? q[1] : p[1];
                    }
                } else {
                    if (act) {
                        const float4 g4 = *(const float4*)(act_in + r * Q + 64 * cb + 4 * i);
                        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= g[e] > 0.f ? 1.f : g[e] + 1.f;       // ELU'(pre) from the stored activation
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += v[e];
                }
                *(float4*)(Y + r * Q + 64 * cb + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
    if constexpr (kEpi == 1) {
        if (colsum != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = csum[e];
                v += __shfl_xor(v, 16, kWave);
                v += __shfl_xor(v, 32, kWave);
                if (kq == 0) atomicAdd(&colsum[64 * cb + 4 * i + e], v);
            }
        }
    }
}

template <int P, int Q, int kEpi>
static int launch_gemm(hipStream_t st, const WideArgs& a) {
    constexpr int ROWS = 16 * (Q == 256 ? kGemmRowTiles : 1) * (16 / (Q / 64));
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        PP_HIP(hipGetDevice(&dev));
        PP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus <= 0) cus = 256;
    }
    int64_t blocks = ceil_div(a.n_rows, ROWS);
    if (blocks > cus) blocks = cus;                       // one 1024-thread workgroup (74 - 92 KB of LDS) per CU
    k_dense_lds<P, Q, kEpi><<<(unsigned)blocks, kGemmThreads, 0, st>>>(a.X, a.n_rows, a.Wr, a.bias, a.act, a.Y, a.act_in, a.colsum);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

template <int kEpi>
static int launch_gemm_pq(int P, int Q, hipStream_t st, const WideArgs& a) {
#define PP_GEMM(PP_, QQ_) if (P == PP_ && Q == QQ_) return launch_gemm<PP_, QQ_, kEpi>(st, a)
    PP_GEMM(64, 128); PP_GEMM(64, 256); PP_GEMM(128, 64); PP_GEMM(128, 128); PP_GEMM(128, 256); PP_GEMM(256, 64); PP_GEMM(256, 128); PP_GEMM(256, 256);
#undef PP_GEMM
    return PP_ERR_ARG;
}

#ifndef PP_GEMM_LDS
#define PP_GEMM_LDS 1
#endif

// ---------------------------------------------------------------------------------------------------------------------------------
// Narrow dense layers: one side small (a classifier head: 64 -> 8 or 256 -> 8 classes and its input gradient 8 -> 256; odd hidden widths):
// the k_dense scheme (pp_dbgnn.hip) with both sides zero-padded to the next of 16/32/64/128/256 (padded P*Q <= 4096: the weight block
// lives in P*Q/64 registers per lane) and guarded scalar I/O on the true widths.
template <int P, int Q>
__global__ __launch_bounds__(kBlock) void k_dense_narrow(const float* __restrict__ A, const float* __restrict__ W, int w_transposed, int64_t n_rows,
                                                        int p_true, int q_true, const float* __restrict__ bias, const float* __restrict__ grad_act,
                                                        float* __restrict__ colsum, float* __restrict__ out) {
    constexpr int KQ = P / 4, CT = Q / 16;
    const int lane = lane_id(), i = lane & 15, kq = lane >> 4;
    float b[KQ][CT];
#pragma unroll
    for (int t = 0; t < KQ; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int k = kq * KQ + t, j = ct * 16 + i;
            b[t][ct] = (k < p_true && j < q_true) ? (w_transposed ? W[j * p_true + k] : W[k * q_true + j]) : 0.f;
        }
    float bias_c[CT], col_acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        bias_c[ct] = (bias && ct * 16 + i < q_true) ? bias[ct * 16 + i] : 0.f;
        col_acc[ct] = 0.f;
    }
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave_id(); tile < n_tiles; tile += n_waves) {
        const int64_t ra = tile * 16 + i;
        float av[KQ];
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            const int k = kq * KQ + t;
            av[t] = (ra < n_rows && k < p_true) ? A[ra * p_true + k] : 0.f;
        }
        f32x4w acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KQ; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b[t][ct], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = tile * 16 + 4 * kq + reg;
                const int j = ct * 16 + i;
                if (r < n_rows && j < q_true) {
                    float v = acc[ct][reg] + bias_c[ct];
                    if (grad_act) {
                        const float y = grad_act[r * q_true + j];
                        v *= y > 0.f ? 1.f : y + 1.f;
                        col_acc[ct] += v;
                    }
                    out[r * q_true + j] = v;
                }
            }
    }
    if (colsum) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = col_acc[ct];
            v += __shfl_xor(v, 16, kWave);
            v += __shfl_xor(v, 32, kWave);
            if (kq == 0 && ct * 16 + i < q_true) atomicAdd(&colsum[ct * 16 + i], v);
        }
    }
}

static inline int pad_width(int w) { return w <= 16 ? 16 : (w <= 32 ? 32 : (w <= 64 ? 64 : (w <= 128 ? 128 : 256))); }

// padded shapes whose weight block (P*Q/64 registers per lane) and accumulators (Q/4 registers) fit beside each other: P*Q <= 4096
static inline bool narrow_shape(int P, int Q) { return P >= 1 && Q >= 1 && P <= 256 && Q <= 256 && pad_width(P) * pad_width(Q) <= 4096; }

#define PP_NARROW(PP_, QQ_) k_dense_narrow<PP_, QQ_><<<grid, kBlock, 0, st>>>(A, W, wt, n, p, q, bias, grad_act, colsum, out)
static int launch_narrow(int Pp, int Qp, unsigned grid, hipStream_t st, const float* A, const float* W, int wt, int64_t n, int p, int q,
                         const float* bias, const float* grad_act, float* colsum, float* out) {
    switch (Pp * 1000 + Qp) {
        case 16016: PP_NARROW(16, 16); break;
        case 16032: PP_NARROW(16, 32); break;
        case 16064: PP_NARROW(16, 64); break;
        case 16128: PP_NARROW(16, 128); break;
        case 16256: PP_NARROW(16, 256); break;
        case 32016: PP_NARROW(32, 16); break;
        case 32032: PP_NARROW(32, 32); break;
        case 32064: PP_NARROW(32, 64); break;
        case 32128: PP_NARROW(32, 128); break;
        case 64016: PP_NARROW(64, 16); break;
        case 64032: PP_NARROW(64, 32); break;
        case 64064: PP_NARROW(64, 64); break;
        case 128016: PP_NARROW(128, 16); break;
        case 128032: PP_NARROW(128, 32); break;
        case 256016: PP_NARROW(256, 16); break;
        default: return PP_ERR_ARG;
    }
    return PP_OK;
}
#undef PP_NARROW

}  // namespace pp

extern "C" {

int pp_wide_layer_supported(int P, int Q) {
    const bool p_ok = P == 64 || P == 128 || P == 256, q_ok = Q == 64 || Q == 128 || Q == 256;
    return p_ok && q_ok && (P > 64 || Q > 64);
}

size_t pp_wide_layer_ws_bytes(int P, int Q) { return pp::align_up((size_t)(P > 0 ? P : 1) * (size_t)(Q > 0 ? Q : 1) * sizeof(float)); }

int pp_wide_layer_f32(const int32_t* ptr, const int32_t* idx, const float* val, int64_t n_rows, int64_t n_self, int64_t n_src, const float* X, int P,
                      const float* self_coef, const float* W, int w_is_kq, int Q, const float* bias, int act, int epilogue, const float* act_in,
                      const int32_t* heavy_slot, const float* heavy_sum, float* agg_out, float* Y, float* colsum, void* ws, size_t ws_bytes,
                      pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0 && n_self >= 0 && n_self <= n_rows && n_src >= 0, PP_ERR_ARG, "pp_wide_layer_f32: bad sizes");
    PP_REQUIRE(n_rows == 0 || n_src >= 1, PP_ERR_ARG, "pp_wide_layer_f32: rows without any source row");
    PP_REQUIRE(pp_wide_layer_supported(P, Q), PP_ERR_ARG, "pp_wide_layer_f32: unsupported layer shape %dx%d (64/128/256 with a side > 64)", P, Q);
    PP_REQUIRE(epilogue == 0 || epilogue == 1, PP_ERR_ARG, "pp_wide_layer_f32: epilogue must be 0 (forward) or 1 (input gradient)");
    PP_REQUIRE(act == 0 || act == 1, PP_ERR_ARG, "pp_wide_layer_f32: act must be 0 or 1");
    PP_REQUIRE(epilogue == 0 || act == 0 || act_in != nullptr, PP_ERR_ARG, "pp_wide_layer_f32: act_in required for the ELU' epilogue");
    PP_REQUIRE(((uintptr_t)X | (uintptr_t)agg_out | (uintptr_t)W) % 16 == 0, PP_ERR_ARG, "pp_wide_layer_f32: X, W and agg_out must be 16-byte aligned");
    PP_REQUIRE(ptr != nullptr || (idx == nullptr && n_src >= n_rows), PP_ERR_ARG, "pp_wide_layer_f32: dense mode takes no CSR and n_src >= n_rows");
    PP_REQUIRE(heavy_slot == nullptr || heavy_sum != nullptr, PP_ERR_ARG, "pp_wide_layer_f32: heavy_slot without heavy_sum");
    if (colsum) PP_HIP(hipMemsetAsync(colsum, 0, (size_t)Q * sizeof(float), st));
    if (n_rows == 0) return PP_OK;
    if (PP_GEMM_LDS && ptr == nullptr && agg_out == nullptr && n_self == n_rows && ws != nullptr && ws_bytes >= pp_wide_layer_ws_bytes(P, Q) &&
        ((uintptr_t)Y | (uintptr_t)act_in | (uintptr_t)bias | (uintptr_t)ws) % 16 == 0) {
        // dense mode: the LDS-staged GEMM on a chunk-major copy of the weights (either orientation of W)
        pp::k_chunk_major_w<<<(unsigned)pp::ceil_div((int64_t)P * Q, pp::kBlock), pp::kBlock, 0, st>>>(W, P, Q, w_is_kq,
                                                                                                      (Q == 128 && P >= 128) ? pp::kGemmChunkNarrow : pp::kGemmChunk, (float*)ws);
        PP_LAUNCH_CHECK();
        const pp::WideArgs a{nullptr, nullptr, nullptr, n_rows, n_self, X, nullptr, (const float*)ws, bias, act, pp::HeavyRows{nullptr, nullptr}, nullptr, Y,
                             act_in, colsum};
        return epilogue == 0 ? pp::launch_gemm_pq<0>(P, Q, st, a) : pp::launch_gemm_pq<1>(P, Q, st, a);
    }
    const float* wr = W;
    if (w_is_kq) {              // W is [P, Q] (k-major, the input-gradient case): the kernel wants one row per output column
        PP_REQUIRE(ws != nullptr && ws_bytes >= pp_wide_layer_ws_bytes(P, Q), PP_ERR_WORKSPACE, "pp_wide_layer_f32: workspace too small");
        pp::k_transpose_f32<<<(unsigned)pp::ceil_div((int64_t)P * Q, pp::kBlock), pp::kBlock, 0, st>>>(W, P, Q, (float*)ws);
        PP_LAUNCH_CHECK();
        wr = (const float*)ws;
    }
    const pp::WideArgs a{ptr, idx, val, n_rows, n_self, X, self_coef, wr, bias, act, pp::HeavyRows{heavy_slot, heavy_sum}, agg_out, Y, act_in, colsum};
    return epilogue == 0 ? pp::launch_wide_pq<0>(P, Q, st, a) : pp::launch_wide_pq<1>(P, Q, st, a);
}

int pp_dense_narrow_supported(int P, int Q) { return pp::narrow_shape(P, Q) ? 1 : 0; }

int pp_dense_narrow_f32(const float* A, const float* W, int w_transposed, int64_t n_rows, int P, int Q, const float* bias, const float* grad_act,
                        float* colsum, float* out, pp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    PP_REQUIRE(n_rows >= 0, PP_ERR_ARG, "pp_dense_narrow_f32: negative size");
    PP_REQUIRE(pp_dense_narrow_supported(P, Q), PP_ERR_ARG, "pp_dense_narrow_f32: padded widths must satisfy P*Q <= 4096 (each <= 256), got %dx%d", P, Q);
    if (colsum) PP_HIP(hipMemsetAsync(colsum, 0, (size_t)Q * sizeof(float), st));
    if (n_rows == 0) return PP_OK;
    int64_t blocks = pp::ceil_div(pp::ceil_div(n_rows, 16), pp::kWavesPerBlock);
    if (blocks > 256 * 3) blocks = 256 * 3;
    const int rc = pp::launch_narrow(pp::pad_width(P), pp::pad_width(Q), (unsigned)blocks, st, A, W, w_transposed, n_rows, P, Q, bias, grad_act, colsum, out);
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // extern "C"
