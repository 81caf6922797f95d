// Can a SIMD overlap one wave's MFMA burst with another wave's memory streaming?  512-thread workgroups: waves 0-3 (one per SIMD) run MFMAs with
// B from LDS, waves 4-7 copy a private slice of a large buffer (16-byte loads / stores).  Times: MFMA only, copy only, both.
//   hipcc -O3 --offload-arch=gfx950 overlap.hip -o overlap && ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int P = 128, Q = 128, TS = P + 4, KQ = P / 4;

__global__ __launch_bounds__(512) void k_both(const float* __restrict__ W, int tiles, const float4* __restrict__ src, float4* __restrict__ dst,
                                               long per_wave_vec, int do_mfma, int do_copy, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_w[Q * TS];
    for (int e = threadIdx.x; e < Q * P; e += 512) s_w[(e / P) * TS + e % P] = W[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    if (wave < 4) {
        if (!do_mfma) return;
        float4 a[KQ / 4];
        for (int c = 0; c < KQ / 4; ++c) a[c] = make_float4(lane * 0.001f + c, 1.f, 2.f, 3.f);
        f32x4 total = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
            for (int chunk = 0; chunk < Q / 16; ++chunk) {
                const float* wp = s_w + (16 * chunk + i) * TS + kq * KQ;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                float4 b[KQ / 4];
#pragma unroll
                for (int c = 0; c < KQ / 4; ++c) b[c] = *(const float4*)(wp + 4 * c);
#pragma unroll
                for (int c = 0; c < KQ / 4; ++c) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, b[c].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, b[c].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, b[c].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, b[c].w, acc1, 0, 0, 0);
                }
                total += acc0 + acc1;
            }
        }
        if (total[0] == 12345.678f) out[threadIdx.x] = total[1];
    } else {
        if (!do_copy) return;
        if (do_copy == 2) {                                // pure VALU work: dependent-free FMAs, no memory
            float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = 1.f, x5 = 2.f, x6 = 3.f, x7 = 4.f;
            for (int it = 0; it < tiles * 136; ++it) {     // 8 FMAs per iteration: ~1090 VALU per "tile"
                x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f; x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f;
                x4 = x4 * 1.0001f + 0.5f; x5 = x5 * 1.0001f + 0.5f; x6 = x6 * 1.0001f + 0.5f; x7 = x7 * 1.0001f + 0.5f;
            }
            if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 1.2345f) out[threadIdx.x] = x0;
            return;
        }
        const long w = (long)blockIdx.x * 4 + (wave - 4);
        const float4* s = src + w * per_wave_vec;
        float4* d = dst + w * per_wave_vec;
        for (long v = lane; v < per_wave_vec; v += 64 * 4) {
            float4 x0 = s[v], x1 = v + 64 < per_wave_vec ? s[v + 64] : x0, x2 = v + 128 < per_wave_vec ? s[v + 128] : x0,
                   x3 = v + 192 < per_wave_vec ? s[v + 192] : x0;
            d[v] = x0;
            if (v + 64 < per_wave_vec) d[v + 64] = x1;
            if (v + 128 < per_wave_vec) d[v + 128] = x2;
            if (v + 192 < per_wave_vec) d[v + 192] = x3;
        }
    }
}

int main() {
    const int blocks = 256, tiles = 600;
    const long per_wave_vec = 160000;                     // 2.56 MB per wave, 2.6 GB per direction in total
    float *w, *out; float4 *src, *dst;
    hipMalloc(&w, P * Q * 4); hipMalloc(&out, 4096);
    hipMalloc(&src, per_wave_vec * 16 * blocks * 4); hipMalloc(&dst, per_wave_vec * 16 * blocks * 4);
    hipMemset(src, 0, per_wave_vec * 16 * blocks * 4);
    std::vector<float> h(P * Q, 0.5f);
    hipMemcpy(w, h.data(), P * Q * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"MFMA only", "copy only", "MFMA + copy", "VALU only", "MFMA + VALU"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            const int m = mode == 0 || mode == 2 || mode == 4, c = mode == 1 || mode == 2 ? 1 : (mode >= 3 ? 2 : 0);
            hipEventRecord(e0);
            k_both<<<blocks, 512>>>(w, tiles, src, dst, per_wave_vec, m, c, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-12s %7.3f ms\n", names[mode], ms);
        }
    return 0;
}
