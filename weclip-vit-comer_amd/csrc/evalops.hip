// Multi-scale + flip inference arithmetic and the confusion histogram, on the device (HBM-bound, fp32 / int64).
//
// Replaces the per-image torch / numpy glue of the reference evaluation loop
// (test_msc_flip_coco.py:61-96 `validate`, test_msc_flip_voc.py; utils/evaluate.py:10-36):
//   scale_flip_pair_kernel : inputs -> [F.interpolate(inputs, scale / size), its horizontal flip]   (:61-62, 78-80)
//   flip_avg_kernel        : msc (+)= w * (resize(seg[0]) + flip(resize(seg[1]))) / 2               (:67-68, 84-86, 88)
//   resize_argmax_kernel   : argmax_c F.interpolate(seg, size=labels)[c]   without the (nc, H, W) tensor (:90-94)
//   confusion_hist_kernel  : hist[t * nc + p] += 1 over pixels with 0 <= t < nc                     (evaluate.py:10-16)
// Bilinear index arithmetic = ATen's area_pixel_compute_source_index with align_corners=False: src =
// max(scale * (dst + 0.5) - 0.5, 0) where `scale` is in/out for size= calls and 1/scale_factor for scale_factor=
// calls (F.interpolate keeps the user's factor when recompute_scale_factor is unset) -- the caller passes it.
#include "common.h"

__device__ __forceinline__ void ev_src(int d, int in, float scale, int& i0, int& i1, float& l1) {
    const float s = fmaxf(scale * (d + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - i0;
}

__device__ __forceinline__ float ev_bilerp(const float* __restrict__ S, int Ws, int y0, int y1, int x0, int x1, float ly,
                                           float lx) {
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * S[(long)y0 * Ws + x0] + lx * S[(long)y0 * Ws + x1]) +
           ly * (hx * S[(long)y1 * Ws + x0] + lx * S[(long)y1 * Ws + x1]);
}

// dst (2, C, Hd, Wd): [0] = bilinear(src (C, Hs, Ws)), [1] = [0] flipped along x.  identity != 0: plain copy + flip.
__global__ __launch_bounds__(256) void scale_flip_pair_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                               int Hs, int Ws, int Hd, int Wd, float sy, float sx,
                                                               int identity) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.z;
    if (x >= Wd || y >= Hd) return;
    const float* S = src + (long)c * Hs * Ws;
    float v;
    if (identity) {
        v = S[(long)y * Ws + x];
    } else {
        int y0, y1, x0, x1;
        float ly, lx;
        ev_src(y, Hs, sy, y0, y1, ly);
        ev_src(x, Ws, sx, x0, x1, lx);
        v = ev_bilerp(S, Ws, y0, y1, x0, x1, ly, lx);
    }
    const long plane = (long)Hd * Wd;
    dst[(long)c * plane + (long)y * Wd + x] = v;
    dst[((long)C + c) * plane + (long)y * Wd + (Wd - 1 - x)] = v;
}

// out (C, Hd, Wd) = (accumulate ? out : 0) + wgt * 0.5 * (R(seg[0])[y, x] + R(seg[1])[y, Wd-1-x]),
// R = bilinear resize (Hs, Ws) -> (Hd, Wd) with size-derived scales (identity when the sizes match)
__global__ __launch_bounds__(256) void flip_avg_kernel(const float* __restrict__ segs, float* __restrict__ out, int C, int Hs,
                                                        int Ws, int Hd, int Wd, float sy, float sx, float wgt,
                                                        int accumulate) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.z;
    if (x >= Wd || y >= Hd) return;
    const float* S0 = segs + (long)c * Hs * Ws;
    const float* S1 = segs + ((long)C + c) * Hs * Ws;
    float a, b;
    if (Hs == Hd && Ws == Wd) {
        a = S0[(long)y * Ws + x];
        b = S1[(long)y * Ws + (Wd - 1 - x)];
    } else {
        int y0, y1, x0, x1, xf0, xf1;
        float ly, lx, lxf;
        ev_src(y, Hs, sy, y0, y1, ly);
        ev_src(x, Ws, sx, x0, x1, lx);
        ev_src(Wd - 1 - x, Ws, sx, xf0, xf1, lxf);
        a = ev_bilerp(S0, Ws, y0, y1, x0, x1, ly, lx);
        b = ev_bilerp(S1, Ws, y0, y1, xf0, xf1, ly, lxf);
    }
    const long o = ((long)c * Hd + y) * Wd + x;
    const float v = wgt * ((a + b) / 2.f);
    out[o] = accumulate ? out[o] + v : v;
}

// pred[y, x] = argmax_c bilinear(seg (C, Hs, Ws))[c, y, x]  (first maximum wins, like torch.argmax on the CPU)
__global__ __launch_bounds__(256) void resize_argmax_kernel(const float* __restrict__ seg, long* __restrict__ pred, int C,
                                                             int Hs, int Ws, int Hd, int Wd, float sy, float sx) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || y >= Hd) return;
    int y0, y1, x0, x1;
    float ly, lx;
    ev_src(y, Hs, sy, y0, y1, ly);
    ev_src(x, Ws, sx, x0, x1, lx);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
        const float v = ev_bilerp(seg + (long)c * Hs * Ws, Ws, y0, y1, x0, x1, ly, lx);
        if (v > best) { best = v; arg = c; }
    }
    pred[(long)y * Wd + x] = arg;
}

// hist[t * nc + p] += #pixels with true label t in [0, nc) and prediction p.  Integer atomics: the result does not
// depend on their order.  Per-workgroup LDS histogram when nc*nc fits, else straight global atomics.
// flag[0] is set when a prediction lies outside [0, nc) (np.bincount would silently widen the histogram).
__global__ __launch_bounds__(256) void confusion_hist_kernel(const long* __restrict__ lt, const long* __restrict__ lp,
                                                              unsigned long long* __restrict__ hist, int* __restrict__ flag,
                                                              long n, int nc, int use_lds) {
    extern __shared__ unsigned int sh[];
    const int cells = nc * nc;
    if (use_lds) {
        for (int i = threadIdx.x; i < cells; i += 256) sh[i] = 0;
        __syncthreads();
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long t = lt[i], p = lp[i];
        if (t < 0 || t >= nc) continue;
        if (p < 0 || p >= nc) { *flag = 1; continue; }
        if (use_lds) atomicAdd(&sh[t * nc + p], 1u);
        else atomicAdd(&hist[t * nc + p], 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256)
            if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int wc_scale_flip_pair(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, float scale_y,
                                  float scale_x, void* stream) {
    WC_CHECK_ARG(src && dst && C > 0 && C <= 65535 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && scale_y > 0 && scale_x > 0,
                 "wc_scale_flip_pair: bad argument");
    const int identity = (Hs == Hd && Ws == Wd && scale_y == 1.0f && scale_x == 1.0f) ? 1 : 0;
    hipLaunchKernelGGL(scale_flip_pair_kernel, dim3(wc_cdiv(Wd, 64), wc_cdiv(Hd, 4), C), dim3(256), 0, (hipStream_t)stream,
                       src, dst, C, Hs, Ws, Hd, Wd, scale_y, scale_x, identity);
    WC_LAUNCH_CHECK("scale_flip_pair_kernel");
    return WC_OK;
}

extern "C" int wc_flip_avg(const float* segs, float* out, int C, int Hs, int Ws, int Hd, int Wd, float weight,
                           int accumulate, void* stream) {
    WC_CHECK_ARG(segs && out && C > 0 && C <= 65535 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "wc_flip_avg: bad argument");
    hipLaunchKernelGGL(flip_avg_kernel, dim3(wc_cdiv(Wd, 64), wc_cdiv(Hd, 4), C), dim3(256), 0, (hipStream_t)stream, segs, out,
                       C, Hs, Ws, Hd, Wd, (float)Hs / Hd, (float)Ws / Wd, weight, accumulate);
    WC_LAUNCH_CHECK("flip_avg_kernel");
    return WC_OK;
}

extern "C" int wc_resize_argmax(const float* seg, long* pred, int C, int Hs, int Ws, int Hd, int Wd, void* stream) {
    WC_CHECK_ARG(seg && pred && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "wc_resize_argmax: bad argument");
    hipLaunchKernelGGL(resize_argmax_kernel, dim3(wc_cdiv(Wd, 64), wc_cdiv(Hd, 4)), dim3(256), 0, (hipStream_t)stream, seg, pred,
                       C, Hs, Ws, Hd, Wd, (float)Hs / Hd, (float)Ws / Wd);
    WC_LAUNCH_CHECK("resize_argmax_kernel");
    return WC_OK;
}

extern "C" int wc_confusion_hist(const long* label_true, const long* label_pred, long* hist, int* flag, long n, int nc,
                                 void* stream) {
    WC_CHECK_ARG(label_true && label_pred && hist && flag && n >= 0 && nc > 0 && nc <= 4096, "wc_confusion_hist: bad argument");
    if (n == 0) return WC_OK;
    const size_t lds = (size_t)nc * nc * sizeof(unsigned int);
    const int use_lds = lds <= 64 * 1024;
    long blocks = (n + 256 * 16 - 1) / (256 * 16);         // ~16 pixels per thread, so the LDS histogram is worth its flush
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(confusion_hist_kernel, dim3((unsigned)blocks), dim3(256), use_lds ? lds : 0, (hipStream_t)stream, label_true,
                       label_pred, (unsigned long long*)hist, flag, n, nc, use_lds);
    WC_LAUNCH_CHECK("confusion_hist_kernel");
    return WC_OK;
}
