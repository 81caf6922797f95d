// Matrix-pipe rate probe for the resident-weight layer scheme: every wave runs `tiles` x (P/4 * Q/16) v_mfma_f32_16x16x4_f32 with the A tile in
// registers and B read from LDS (or from registers), nothing else.  Prints the fraction of the fp32 MFMA peak per variant and waves/SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int P = 128, Q = 128, TS = P + 4, KQ = P / 4;

template <int kVariant, int kThreads>
__global__ __launch_bounds__(kThreads) void k_probe(const float* __restrict__ W, int tiles, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_w[Q * TS];
    for (int e = threadIdx.x; e < Q * P; e += kThreads) s_w[(e / P) * TS + e % P] = W[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    float4 a[KQ / 4];
    for (int c = 0; c < KQ / 4; ++c) a[c] = make_float4(lane * 0.001f + c, 1.f, 2.f, 3.f);
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
        for (int chunk = 0; chunk < Q / 16; ++chunk) {
            const float* wp = s_w + (16 * chunk + i) * TS + kq * KQ;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (kVariant == 0) {                 // the k_wide_layer loop: register double buffer + scheduling barriers
                constexpr int kQuads = KQ / 4, kStep = 4;
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
            } else if constexpr (kVariant == 1) {          // all B of the chunk loaded first (8 x b128), then 32 MFMAs
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
            } else {                                       // B from registers: the pipe alone
#pragma unroll
                for (int c = 0; c < KQ / 4; ++c) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, a[c].y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, a[c].z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, a[c].w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, a[c].x, acc1, 0, 0, 0);
                }
            }
            total += acc0 + acc1;
        }
    }
    if (total[0] == 12345.678f) out[threadIdx.x] = total[1];          // keep the work alive
}

template <int kVariant, int kThreads>
static void run(const char* name, const float* w, float* out) {
    const int tiles = 600, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<kVariant, kThreads><<<blocks, kThreads>>>(w, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<kVariant, kThreads><<<blocks, kThreads>>>(w, tiles, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * P * Q * (double)tiles * (kThreads / 64) * blocks;
    printf("%-40s %4d threads (%d waves/SIMD): %7.3f ms  %6.1f TFLOP/s = %.2f of 157\n", name, kThreads, kThreads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.0);
}

int main() {
    float *w, *out;
    hipMalloc(&w, P * Q * 4); hipMalloc(&out, 4096);
    std::vector<float> h(P * Q, 0.5f);
    hipMemcpy(w, h.data(), P * Q * 4, hipMemcpyHostToDevice);
    run<2, 256>("B from registers", w, out);
    run<2, 512>("B from registers", w, out);
    run<2, 768>("B from registers", w, out);
    run<0, 256>("LDS B, double buffer + sched_barrier", w, out);
    run<0, 512>("LDS B, double buffer + sched_barrier", w, out);
    run<0, 768>("LDS B, double buffer + sched_barrier", w, out);
    run<0, 1024>("LDS B, double buffer + sched_barrier", w, out);
    run<1, 256>("LDS B, chunk loaded first", w, out);
    run<1, 512>("LDS B, chunk loaded first", w, out);
    run<1, 768>("LDS B, chunk loaded first", w, out);
    run<1, 1024>("LDS B, chunk loaded first", w, out);
    return 0;
}
