// Shared helpers for the weclip_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

// native vector types (arrays of HIP's struct uint4 are not promoted to registers by SROA)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WC_OK 0
#define WC_ERR_ARG 1
#define WC_ERR_HIP 2

extern "C" void wc_set_error(const char* fmt, ...);

#define WC_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            wc_set_error(__VA_ARGS__);          \
            return WC_ERR_ARG;                  \
        }                                       \
    } while (0)

#define WC_LAUNCH_CHECK(name)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            wc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return WC_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

static inline int wc_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// optional HIP-event timing of a launch (core.hip; no-ops unless wc_prof_enable(1))
int wc_prof_begin(void* stream);
void wc_prof_end(int idx, const char* name, double work, void* stream);
void wc_prof_end2(int idx, const char* name, double work, double bytes, void* stream);   // + algorithmic HBM bytes
int wc_prof_begin_always(void* stream);            // records whenever profiling is on (brackets around many launches)
void wc_prof_end_n(int idx, const char* name, double work, void* stream, int n);   // one event pair around n launches

// 64-lane wavefront reductions (DPP/shuffle based).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Phi(u) = 0.5 (1 + erf(u / sqrt 2)) and e = exp(-u^2 / 2) together: Abramowitz & Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. two
// float ulps of a value near 1) -- one reciprocal, one exponential and five FMAs, where the library erff is a ~50-instruction
// branchy polynomial that made every exact-GELU epilogue VALU-bound (row GEMM with GELU: 2.7 TB/s against 5.6 TB/s without).
// The exponential is the one the derivative's phi(u) needs anyway.
__device__ __forceinline__ void wc_gelu_parts(float u, float& cdf, float& e) {
    const float ax = fabsf(u) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    e = __expf(-ax * ax);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float h = 0.5f * p * t * e;                 // 0.5 * erfc(|u| / sqrt 2)
    cdf = u >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float wc_gelu(float u) {               // u * Phi(u)
    float c, e;
    wc_gelu_parts(u, c, e);
    return u * c;
}
__device__ __forceinline__ float wc_gelu_grad(float u) {          // d/du [u * Phi(u)] = Phi(u) + u * phi(u)
    float c, e;
    wc_gelu_parts(u, c, e);
    return fmaf(u * 0.3989422804014327f, e, c);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide reductions for blockDim.x <= 1024 (<=16 waves); `red` is >=16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}
__device__ __forceinline__ float block_min(float v, float* red) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fminf(r, red[i]);
    return r;
}
