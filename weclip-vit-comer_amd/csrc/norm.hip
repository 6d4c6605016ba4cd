// Row LayerNorm (fp32 statistics) for gfx950.  Replaces `LayerNorm.forward`
// (reference clip/model.py:177-183, WeCLIP_model/Decoder/TransDecoder.py:50-56): fp32 in,
// fp32 math, and the result written in the formats the next consumer wants -- fp32 and/or an
// fp16 hi(+lo) pair that feeds the MFMA GEMM without a separate cast pass.
// One 64-lane wave per row (wavefront shuffle reductions), 4 rows per 256-thread block.
#include "common.h"

template <int MAXV>   // MAXV float4 per lane => D <= 256*MAXV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ b, float eps,
                                                         float* __restrict__ y32,
                                                         __half* __restrict__ y16,
                                                         __half* __restrict__ y16lo, long rows, int D,
                                                         long ldx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;   // float4 per row
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    const float4* wr = reinterpret_cast<const float4*>(w);
    const float4* br = reinterpret_cast<const float4*>(b);
    // Every load is issued UNCONDITIONALLY from a clamped index and masked afterwards: behind an `if (j < nv)` hipcc waits for each
    // load before it issues the next (round 4, found in the ISA: 3 dependent HBM latencies per row at D = 768).  For rows of up to
    // 1024 values the affine parameters are requested together with the row as well.
    constexpr bool HOIST = MAXV <= 4;
    float4 v[MAXV], ww[HOIST ? MAXV : 1], bb[HOIST ? MAXV : 1];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i, jc = j < nv ? j : nv - 1;
        v[i] = xr[jc];
        if constexpr (HOIST) {
            ww[i] = wr[jc];
            bb[i] = br[jc];
        }
    }
    if constexpr (HOIST) {
        // (pinned behind a scheduling barrier: without them hipcc sinks the gamma / beta loads behind the two reductions, where
        //  their L2 latency sits between the statistics and the stores of every row)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            asm volatile("" : "+v"(ww[i].x), "+v"(ww[i].y), "+v"(ww[i].z), "+v"(ww[i].w), "+v"(bb[i].x), "+v"(bb[i].y), "+v"(bb[i].z),
                         "+v"(bb[i].w));
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + 64 * i >= nv) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        const float t = (a * a + b2 * b2) + (c * c + d * d);
        q += lane + 64 * i < nv ? t : 0.f;
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float4 w4 = HOIST ? ww[HOIST ? i : 0] : wr[j], b4 = HOIST ? bb[HOIST ? i : 0] : br[j];
            float o[4] = {(v[i].x - mean) * rstd * w4.x + b4.x, (v[i].y - mean) * rstd * w4.y + b4.y,
                          (v[i].z - mean) * rstd * w4.z + b4.z, (v[i].w - mean) * rstd * w4.w + b4.w};
            const long off = row * D + 4L * j;
            if (y32) *reinterpret_cast<float4*>(y32 + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (y16) {
                __half h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    h[k] = __float2half(o[k]);
                    l[k] = __float2half(o[k] - __half2float(h[k]));
                }
                *reinterpret_cast<uint2*>(y16 + off) = *reinterpret_cast<uint2*>(h);
                if (y16lo) *reinterpret_cast<uint2*>(y16lo + off) = *reinterpret_cast<uint2*>(l);
            }
        }
    }
}

extern "C" int wc_layernorm(const float* x, long ldx, const float* w, const float* b, float eps,
                            float* y32, void* y16, void* y16lo, long rows, int D, void* stream) {
    WC_CHECK_ARG(x && w && b && rows > 0 && D > 0 && D % 4 == 0 && D <= 4096 && ldx >= D && ldx % 4 == 0,
                 "wc_layernorm: need D %% 4 == 0, D <= 4096, ldx %% 4 == 0");
    WC_CHECK_ARG(y32 || y16, "wc_layernorm: no output requested");
    dim3 grid(wc_cdiv(rows, 4));
    hipStream_t st = (hipStream_t)stream;
    if (D <= 256)
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, x, w, b, eps, y32, (__half*)y16, (__half*)y16lo, rows, D, ldx);
    else if (D <= 1024)
        hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, x, w, b, eps, y32, (__half*)y16, (__half*)y16lo, rows, D, ldx);
    else
        hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, st, x, w, b, eps, y32, (__half*)y16, (__half*)y16lo, rows, D, ldx);
    WC_LAUNCH_CHECK("layernorm_kernel");
    return WC_OK;
}
