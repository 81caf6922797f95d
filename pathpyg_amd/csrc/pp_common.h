// pathpyg_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave width is 64 everywhere; nothing here is meant to build for another target.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pathpyg_amd.h"

namespace pp {

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxGrid = 256 * 8;      // grid-stride cap: 256 CUs x 8 resident 256-thread blocks

void set_error(const char* fmt, ...);

#define PP_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            pp::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return PP_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define PP_LAUNCH_CHECK() PP_HIP(hipGetLastError())

#define PP_REQUIRE(cond, code, ...)                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            pp::set_error(__VA_ARGS__);                                               \
            return (code);                                                            \
        }                                                                             \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int grid_for(int64_t n, int per_block) {
    int64_t g = ceil_div(n, per_block);
    return (int)(g < 1 ? 1 : g);
}

// Carves 256-byte aligned pieces out of a caller-provided workspace (torch owns the memory).
struct Arena {
    char* base;
    size_t cap;
    size_t used = 0;
    Arena(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
    template <typename T>
    T* take(int64_t count) {
        size_t bytes = align_up((size_t)(count < 1 ? 1 : count) * sizeof(T));
        char* p = base ? base + used : nullptr;
        used += bytes;
        return (T*)p;
    }
    bool ok() const { return used <= cap; }
};

static inline int bits_for(uint64_t max_value) {       // number of significant bits of max_value
    int b = 0;
    while (max_value) { ++b; max_value >>= 1; }
    return b < 1 ? 1 : b;
}

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (lane_id() >= d) v += o;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        T o = __shfl_xor(v, d, kWave);
        v = o > v ? o : v;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        T o = __shfl_xor(v, d, kWave);
        v = o < v ? o : v;
    }
    return v;
}

// Block-wide exclusive sum of one value per thread (kBlock threads). `scratch` holds >= kWavesPerBlock+1 Ts.
// Returns the exclusive prefix; *total receives the block sum (same in every thread).
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T v, T* scratch, T* total) {
    T inc = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) scratch[wave_id()] = inc;
    __syncthreads();
    T wave_base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        T s = scratch[w];
        if (w < wave_id()) wave_base += s;
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return wave_base + inc - v;
}

// Streaming accesses (data touched once per kernel): the `nt` cache policy keeps them from evicting the small tables the same kernel
// reads at random (row pointers, degree^-1/2) out of the 4 MB L2 of its XCD.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T load_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void store_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ uint4 load_stream_u4(const uint32_t* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream_u4(uint32_t* p, uint4 v) {
    u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
}

typedef float f32x4s_t __attribute__((ext_vector_type(4)));
// 16-byte store of a matrix row piece the kernel never reads again.  `stream` (wave-uniform): the matrix is far larger than the 256 MB
// Infinity Cache (layer outputs of the higher-order graph: consumed by a LATER kernel, gigabytes away) — the `nt` policy keeps it from
// washing the gather sources out of the caches (round 3: 2 % on each fused layer kernel); small matrices (the first-order graph's 128 MB)
// are kept cacheable for the kernel that gathers from them next.
constexpr int64_t kStreamFromBytes = (int64_t)512 << 20;
__device__ __forceinline__ void store_row_f4(float* p, float4 v, bool stream) {
    if (stream) {
        f32x4s_t t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4s_t*>(p));
    } else {
        *(float4*)p = v;
    }
}

// 16-byte load of a matrix row piece that is read exactly once (same policy and threshold as store_row_f4)
__device__ __forceinline__ float4 load_row_f4(const float* p, bool stream) {
    if (stream) {
        const f32x4s_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4s_t*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *(const float4*)p;
}

// Raw buffer access (buffer_load_* with a 32-bit byte offset): the address arithmetic of a gather is ONE 32-bit VALU add instead of a 64-bit
// multiply-add, and an offset at or above kBufOob reads as zeros WITHOUT touching memory — absent neighbours, rows past the end and masked
// lanes need neither a branch around the load nor a select behind it.  (On this chip the VALU and the matrix pipe of a SIMD do not
// overlap across waves — tools/probes/mfma/overlap.hip — so every VALU instruction of the gather stage is paid in full.)
// Arrays addressed this way must be smaller than kBufOob bytes.
constexpr uint32_t kBufOob = 0xFFFFF000u;
using buf_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ buf_t buf_of(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)kBufOob, 0x00020000); }
__device__ __forceinline__ float4 buf_load_f4(buf_t r, uint32_t off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);      // (a 16-byte GCC vector: copy it out, do not convert it)
    static_assert(sizeof(v) == 16, "b128");
    float4 out;
    __builtin_memcpy(&out, &v, 16);
    return out;
}
__device__ __forceinline__ uint32_t buf_load_u32(buf_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0); }
__device__ __forceinline__ float buf_load_f32(buf_t r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
// value of `v` in the lane whose byte address (4 * lane) is `addr4` — constant parts of the address fold into the instruction's offset field
__device__ __forceinline__ int lane_read_i(int addr4, int v) { return __builtin_amdgcn_ds_bpermute(addr4, v); }
__device__ __forceinline__ float lane_read_f(int addr4, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr4, __builtin_bit_cast(int, v))); }

// Counter-based dropout (pp_dropout_f32 and the fused layer epilogues): keep(row, col) is a pure function of (seed, call site, GLOBAL row id,
// column) — two multiply-xorshift rounds on 32 bits, the same arithmetic as pathpyg_amd.nn.sharded.dropout_mask, bit for bit.  No mask
// tensor exists: the backward pass regenerates the decision, and every rank of a partitioned run derives the same one for the same row.
struct DropSite {
    uint32_t key, thr;       // thr == 0: no dropout at this site
    float scale, keep;       // 1 / (1 - p), 1 - p
    int64_t row0;            // global id of the matrix's first row
};
__device__ __forceinline__ bool dropout_keep(int64_t row, int col, int width, uint32_t key, uint32_t threshold) {
    const uint64_t idx = (uint64_t)row * (uint64_t)width + (uint64_t)col;
    uint32_t x = (uint32_t)idx * 2654435761u + (uint32_t)(idx >> 32) * 40503u + key;
    x = ((x >> 16) ^ x) * 0x45D9F3Bu;
    x = ((x >> 16) ^ x) * 0x45D9F3Bu;
    x = (x >> 16) ^ x;
    return x >= threshold;
}
static inline uint32_t dropout_key(int64_t seed, int64_t tag) {
    return (uint32_t)(((uint64_t)seed * 0x9E3779B1ull + (uint64_t)tag * 0x85EBCA6Bull + 0x27D4EB2Full) & 0xFFFFFFFFull);
}
static inline DropSite drop_site(double p, int64_t seed, int64_t tag, int64_t row0) {
    DropSite d{0u, 0u, 1.f, 1.f, row0};
    if (p > 0.0) { d.key = dropout_key(seed, tag); d.thr = (uint32_t)(p * 4294967296.0); d.scale = (float)(1.0 / (1.0 - p)); d.keep = (float)(1.0 - p); }
    return d;
}

// Rows of a CSR with very many entries (hubs of a scale-free graph) are not walked by the lane group that owns the row: a pre-pass
// (pp_spmm_heavy_f32) sums them chunk-wise with whole workgroups; the row kernels read the finished sum instead.
struct HeavyRows {
    const int32_t* slot;     // [n_rows]: -1 for ordinary rows, else the row of `sum`; nullptr = no heavy rows at all
    const float* sum;        // [n_heavy, F]
};

// ELU(x) = x > 0 ? x : exp(x) - 1, branch-free and cheap (libm's expm1f is ~40 instructions and branches): on (-0.25, 0] the
// degree-6 Taylor polynomial of expm1 (truncation < 5e-8 relative), below that v_exp_f32 - 1 (the result is <= -0.22, so the 1-ulp
// error of the exponential stays < 3e-7 relative).  The two-lane form runs the polynomial on packed fp32 (v_pk_fma_f32).
using pp_f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ pp_f32x2 elu_fast2(pp_f32x2 x) {
    pp_f32x2 p = {1.f / 720.f, 1.f / 720.f};
    p = __builtin_elementwise_fma(p, x, pp_f32x2{1.f / 120.f, 1.f / 120.f});
    p = __builtin_elementwise_fma(p, x, pp_f32x2{1.f / 24.f, 1.f / 24.f});
    p = __builtin_elementwise_fma(p, x, pp_f32x2{1.f / 6.f, 1.f / 6.f});
    p = __builtin_elementwise_fma(p, x, pp_f32x2{0.5f, 0.5f});
    p = __builtin_elementwise_fma(p, x, pp_f32x2{1.f, 1.f});
    p = p * x;
    pp_f32x2 r;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float e = __expf(x[k]) - 1.f;
        const float neg = x[k] > -0.25f ? p[k] : e;
        r[k] = x[k] > 0.f ? x[k] : neg;
    }
    return r;
}

__device__ __forceinline__ float elu_fast(float x) { return elu_fast2(pp_f32x2{x, x})[0]; }

__device__ __forceinline__ float4 elu_fast4(float4 v) {
    const pp_f32x2 a = elu_fast2(pp_f32x2{v.x, v.y}), b = elu_fast2(pp_f32x2{v.z, v.w});
    return make_float4(a[0], a[1], b[0], b[1]);
}

// first index in [lo, hi) with a[idx] > key (upper bound), a ascending
template <typename T, typename I>
__device__ __forceinline__ I upper_bound_dev(const T* __restrict__ a, I lo, I hi, T key) {
    while (lo < hi) {
        I mid = lo + ((hi - lo) >> 1);
        if (a[mid] > key) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// first index in [lo, hi) with a[idx] >= key (lower bound), a ascending
template <typename T, typename I>
__device__ __forceinline__ I lower_bound_dev(const T* __restrict__ a, I lo, I hi, T key) {
    while (lo < hi) {
        I mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

}  // namespace pp
