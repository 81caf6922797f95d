// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/probes/expand_probe tools/probes/expand_probe.hip   (stand-alone; run on the GPU box)
// Stand-alone probe: where does the fill kernel's time go?  hipcc --offload-arch=gfx950 -O3 -o /tmp/expand_probe expand_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kBlock = 256, kItems = 8, kTile = kBlock * kItems;
struct alignas(16) I64x2 { int64_t a, b; };

// V0: stores only
template <int kItemsT>
__global__ __launch_bounds__(kBlock) void v0_store(int64_t total, int64_t* out) {
    const int64_t p0 = (int64_t)blockIdx.x * (kBlock * kItemsT);
#pragma unroll
    for (int u = 0; u < kItemsT / 2; ++u) {
        const int64_t p = p0 + 2 * (u * kBlock + threadIdx.x);
        if (p + 1 < total) {
            *(I64x2*)(out + p) = I64x2{p, p + 1};
            *(I64x2*)(out + total + p) = I64x2{p + 7, p + 8};
        }
    }
}
// V1: + one dependent load chain (tile_src -> offsets staged to LDS) feeding the stores
__global__ __launch_bounds__(kBlock) void v1_stage(const int64_t* offset, const int64_t* tile_src, int64_t total, int64_t* out) {
    __shared__ int32_t s_rel[kTile + 2];
    const int64_t p0 = (int64_t)blockIdx.x * kTile;
    const int64_t s_first = tile_src[blockIdx.x], s_beyond = tile_src[blockIdx.x + 1];
    const int n_bound = (int)(s_beyond - s_first + 1);
    int64_t g[9];
#pragma unroll
    for (int it = 0; it < 9; ++it) { int k = it * kBlock + threadIdx.x; g[it] = (k <= n_bound && k <= kTile) ? offset[s_first + k] : 0; }
#pragma unroll
    for (int it = 0; it < 9; ++it) { int k = it * kBlock + threadIdx.x; if (k <= n_bound && k <= kTile) s_rel[k] = (int32_t)(g[it] - p0); }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kItems / 2; ++u) {
        const int q = 2 * (u * kBlock + threadIdx.x);
        const int64_t p = p0 + q;
        const int64_t a = s_rel[(q >> 1) % (n_bound + 1)];
        if (p + 1 < total) {
            *(I64x2*)(out + p) = I64x2{s_first + a, s_first + a};
            *(I64x2*)(out + total + p) = I64x2{a, a + 1};
        }
    }
}
// V2: V0 with 8-byte stores (the first version of the kernel)
__global__ __launch_bounds__(kBlock) void v2_store8(int64_t total, int64_t* out) {
    const int64_t p0 = (int64_t)blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t p = p0 + k * kBlock + threadIdx.x;
        if (p < total) { out[p] = p; out[total + p] = p + 7; }
    }
}
__global__ void k_tile_src(const int64_t* offset, int64_t n_src, int64_t total, int64_t n_tiles, int64_t* tile_src) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_tiles) return;
    int64_t p = b * kTile, lo = 0, hi = n_src + 1;
    if (p >= total) { tile_src[b] = n_src - 1; return; }
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (offset[mid] > p) hi = mid; else lo = mid + 1; }
    tile_src[b] = lo - 1;
}

// ---------------------------------------------------------------------------------------------------------
// V4: wave-private tiles: every wave resolves and stores its own 512 slots, no workgroup barriers at all
constexpr int kWaveTile = 512, kWCap = 512 + 1;
__device__ __forceinline__ int lane() { return threadIdx.x & 63; }
template <bool kList>
__global__ __launch_bounds__(kBlock) void v4_wave(const int64_t* __restrict__ offset, const uint32_t* __restrict__ first_pos,
                                                  const uint32_t* __restrict__ list, int64_t n_src, int64_t total,
                                                  const int64_t* __restrict__ tile_src, int64_t* __restrict__ out) {
    __shared__ int32_t s_rel_all[4][kWCap + 1];
    __shared__ uint32_t s_pos_all[4][kWCap + 1];
    __shared__ __attribute__((aligned(16))) int32_t s_src_all[4][kWaveTile];
    __shared__ __attribute__((aligned(16))) uint32_t s_at_all[4][kWaveTile];
    const int w = threadIdx.x >> 6, l = lane();
    int32_t* s_rel = s_rel_all[w]; uint32_t* s_pos = s_pos_all[w]; int32_t* s_src = s_src_all[w]; uint32_t* s_at = s_at_all[w];
    const int64_t tile = (int64_t)blockIdx.x * 4 + w;
    const int64_t p0 = tile * kWaveTile;
    if (p0 >= total) return;
    const int64_t p1 = p0 + kWaveTile < total ? p0 + kWaveTile : total;
    const int64_t s_first = tile_src[tile], s_beyond = tile_src[tile + 1];
    const bool staged = (s_beyond - s_first + 1) <= kWCap;
    const int n_bound = staged ? (int)(s_beyond - s_first + 1) : 0;
    const int64_t rel0 = offset[s_first] - p0;
    constexpr int kIt = (kWCap + 1 + 63) / 64;   // 9
    int64_t g_off[kIt]; uint32_t g_pos[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) { const int k = it * 64 + l; g_off[it] = k <= n_bound ? offset[s_first + k] : 0; g_pos[it] = k < n_bound ? first_pos[s_first + k] : 0u; }
#pragma unroll
    for (int it = 0; it < kIt; ++it) { const int k = it * 64 + l; if (k <= n_bound) { const int64_t rel = g_off[it] - p0; s_rel[k] = rel < 0 ? -1 : (rel > kWaveTile ? kWaveTile + 1 : (int32_t)rel); s_pos[k] = g_pos[it]; } }
    *(int4*)(s_src + 8 * l) = make_int4(0, 0, 0, 0);
    *(int4*)(s_src + 8 * l + 4) = make_int4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < kIt; ++it) { const int k = it * 64 + l; if (k >= 1 && k < n_bound) { const int rel = s_rel[k]; if (rel < kWaveTile && s_rel[k + 1] > rel) s_src[rel] = k; } }
    __builtin_amdgcn_wave_barrier();
    int4 a4 = *(int4*)(s_src + 8 * l), b4 = *(int4*)(s_src + 8 * l + 4);
    int kk[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int j = 1; j < 8; ++j) kk[j] = kk[j] > kk[j - 1] ? kk[j] : kk[j - 1];
    int v = kk[7];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d, 64); if (l >= d) v = o > v ? o : v; }
    int before = __shfl_up(v, 1, 64); if (l == 0) before = 0;
    uint32_t at[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int k = kk[j] > before ? kk[j] : before; const int q = 8 * l + j; kk[j] = k; at[j] = s_pos[k] + (uint32_t)(k == 0 ? (int64_t)q - rel0 : (int64_t)(q - s_rel[k])); }
    *(int4*)(s_src + 8 * l) = make_int4(kk[0], kk[1], kk[2], kk[3]);
    *(int4*)(s_src + 8 * l + 4) = make_int4(kk[4], kk[5], kk[6], kk[7]);
    *(int4*)(s_at + 8 * l) = make_int4(at[0], at[1], at[2], at[3]);
    *(int4*)(s_at + 8 * l + 4) = make_int4(at[4], at[5], at[6], at[7]);
    __builtin_amdgcn_wave_barrier();
    int64_t src[8], dst[8]; uint32_t at2[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int q = 2 * (u * 64 + l); const int2 k2 = *(const int2*)(s_src + q); const uint2 a2 = *(const uint2*)(s_at + q); src[2*u] = s_first + k2.x; src[2*u+1] = s_first + k2.y; at2[2*u] = a2.x; at2[2*u+1] = a2.y; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int64_t p = p0 + 2 * ((j >> 1) * 64 + l) + (j & 1); dst[j] = kList ? (p < p1 ? (int64_t)list[at2[j]] : 0) : (int64_t)at2[j]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t p = p0 + 2 * (u * 64 + l); if (p + 1 < p1) { *(I64x2*)(out + p) = I64x2{src[2*u], src[2*u+1]}; *(I64x2*)(out + total + p) = I64x2{dst[2*u], dst[2*u+1]}; } }
}
__global__ void k_tile_src_w(const int64_t* offset, int64_t n_src, int64_t total, int64_t n_tiles, int64_t* tile_src) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_tiles) return;
    int64_t p = b * kWaveTile, lo = 0, hi = n_src + 1;
    if (p >= total) { tile_src[b] = n_src - 1; return; }
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (offset[mid] > p) hi = mid; else lo = mid + 1; }
    tile_src[b] = lo - 1;
}

int main() {
    const int64_t n_src = 19000000;
    std::vector<int64_t> off(n_src + 1);
    uint64_t x = 88172645463325252ull;
    off[0] = 0;
    for (int64_t i = 0; i < n_src; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; off[i + 1] = off[i] + (x % 5 == 0 ? 0 : (x >> 20) % 4); }
    const int64_t total = off[n_src] & ~1ll;
    const int64_t n_tiles = (total + kTile - 1) / kTile;
    int64_t *d_off, *d_tile, *d_out;
    CK(hipMalloc(&d_off, (n_src + 1) * 8)); CK(hipMalloc(&d_tile, (n_tiles + 2) * 8)); CK(hipMalloc(&d_out, total * 16));
    CK(hipMemcpy(d_off, off.data(), (n_src + 1) * 8, hipMemcpyHostToDevice));
    k_tile_src<<<(n_tiles + 256) / 256, 256>>>(d_off, n_src, total, n_tiles, d_tile);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto fn) {
        fn(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 10; ++r) { hipEventRecord(a); fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-44s %8.3f ms  %8.1f GB/s written\n", name, best, total * 16.0 / best / 1e6);
    };
    printf("total slots %lld (%.1f MB out), tiles %lld\n", (long long)total, total * 16.0 / 1e6, (long long)n_tiles);
    run("v0 16B stores, 2048 slots/block", [&] { v0_store<8><<<n_tiles, kBlock>>>(total, d_out); });
    run("v0 16B stores, 4096 slots/block", [&] { v0_store<16><<<(n_tiles + 1) / 2, kBlock>>>(total, d_out); });
    run("v0 16B stores, 8192 slots/block", [&] { v0_store<32><<<(n_tiles + 3) / 4, kBlock>>>(total, d_out); });
    run("v2 8B stores, 2048 slots/block", [&] { v2_store8<<<n_tiles, kBlock>>>(total, d_out); });
    run("v1 tile_src + staged offsets + LDS + stores", [&] { v1_stage<<<n_tiles, kBlock>>>(d_off, d_tile, total, d_out); });
    {
        const int64_t n_wt = (total + kWaveTile - 1) / kWaveTile;
        int64_t* d_wt; uint32_t *d_pos, *d_list;
        CK(hipMalloc(&d_wt, (n_wt + 6) * 8)); CK(hipMalloc(&d_pos, n_src * 4)); CK(hipMalloc(&d_list, n_src * 4));
        std::vector<uint32_t> pos(n_src), lst(n_src);
        for (int64_t i = 0; i < n_src; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; pos[i] = (uint32_t)(x % (n_src - 8)); lst[i] = (uint32_t)i; }
        CK(hipMemcpy(d_pos, pos.data(), n_src * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_list, lst.data(), n_src * 4, hipMemcpyHostToDevice));
        k_tile_src_w<<<(n_wt + 256) / 256, 256>>>(d_off, n_src, total, n_wt + 4, d_wt);
        run("v4 wave-private tiles, no list", [&] { v4_wave<false><<<(n_wt + 3) / 4, kBlock>>>(d_off, d_pos, nullptr, n_src, total, d_wt, d_out); });
        run("v4 wave-private tiles, random list gather", [&] { v4_wave<true><<<(n_wt + 3) / 4, kBlock>>>(d_off, d_pos, d_list, n_src, total, d_wt, d_out); });
    }
    CK(hipMemset(d_out, 0, total * 16));
    run("hipMemset same bytes", [&] { hipMemsetAsync(d_out, 1, total * 16); });
    return 0;
}
