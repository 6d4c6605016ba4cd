// Bilinear plane resize (fp32, NCHW planes).  Replaces
//   cv2.resize(INTER_LINEAR) of the refined CAMs  (clip/clip_tool.py:202-216 via
//     pytorch_grad_cam/utils/image.py:57)                      -> align_corners = 0
//   F.interpolate(imgs, bilinear, align_corners=True)          (WeCLIP_model/PAR.py:67)
//   F.interpolate(segs, bilinear, align_corners=False)         (scripts/dist_clip_voc.py:250)
// Index arithmetic follows ATen's area_pixel_compute_source_index (half-pixel centres,
// negative source clamped to 0), which equals OpenCV's INTER_LINEAR for float data.
#include "common.h"

__device__ __forceinline__ void src_index(int d, int in, float scale, int align, int& i0, int& i1,
                                          float& l1) {
    float s = align ? scale * d : fmaxf(scale * (d + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ src,
                                                        float* __restrict__ dst, int Hs, int Ws,
                                                        int Hd, int Wd, float sy, float sx, int align) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || y >= Hd) return;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(y, Hs, sy, align, y0, y1, ly);
    src_index(x, Ws, sx, align, x0, x1, lx);
    const float* S = src + (long)blockIdx.z * Hs * Ws;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v = hy * (hx * S[(long)y0 * Ws + x0] + lx * S[(long)y0 * Ws + x1]) +
                    ly * (hx * S[(long)y1 * Ws + x0] + lx * S[(long)y1 * Ws + x1]);
    dst[((long)blockIdx.z * Hd + y) * Wd + x] = v;
}

// Backward of the bilinear up-sampling, separable and deterministic (ATen's backward scatters with
// atomics): bilinear interpolation is R_y (x) R_x with two non-zeros per destination row, so
//   pass Y: tmp[p, ys, x]   = sum_y wy(y -> ys) * gdst[p, y, x]          (Hd x Wd -> Hs x Wd)
//   pass X: gsrc[p, ys, xs] = sum_x wx(x -> xs) * tmp[p, ys, x]          (Hs x Wd -> Hs x Ws)
// each output sums the ~2*scale destinations that can touch it.
// Pass Y first: it runs on the big (Hd x Wd) gradient with lanes along x, so every read is a coalesced row
// segment; the strided pass X then only sees the (Hs x Wd) intermediate.
__global__ __launch_bounds__(256) void bilinear_bwd_y_kernel(const float* __restrict__ gdst, float* __restrict__ tmp,
                                                              int Hs, int Hd, int Wd, float sy, int align) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ys = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || ys >= Hs) return;
    const float* G = gdst + (long)blockIdx.z * Hd * Wd + x;
    const float iy = 1.0f / sy;
    // destinations whose source coordinate can fall in (ys-1, ys+1), either index convention
    int y_lo = (int)floorf((ys - 1.5f) * iy) - 1, y_hi = (int)ceilf((ys + 1.5f) * iy) + 1;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > Hd - 1) y_hi = Hd - 1;
    float acc = 0.f;
    int y = y_lo;
    for (; y + 7 <= y_hi; y += 8) {            // 8 rows per trip: the loads are issued before the FMAs
        float g[8], wy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int y0, y1;
            float ly;
            src_index(y + u, Hs, sy, align, y0, y1, ly);
            wy[u] = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);   // wave-uniform
            g[u] = G[(long)(y + u) * Wd];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(wy[u], g[u], acc);
    }
    for (; y <= y_hi; ++y) {
        int y0, y1;
        float ly;
        src_index(y, Hs, sy, align, y0, y1, ly);
        const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
        if (wy != 0.f) acc = fmaf(wy, G[(long)y * Wd], acc);
    }
    tmp[((long)blockIdx.z * Hs + ys) * Wd + x] = acc;
}

__global__ __launch_bounds__(256) void bilinear_bwd_x_kernel(const float* __restrict__ tmp, float* __restrict__ gsrc,
                                                              int Hs, int Ws, int Wd, float sx, int align) {
    const int xs = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ys = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xs >= Ws || ys >= Hs) return;
    const float* T = tmp + ((long)blockIdx.z * Hs + ys) * Wd;
    const float ix = 1.0f / sx;
    int x_lo = (int)floorf((xs - 1.5f) * ix) - 1, x_hi = (int)ceilf((xs + 1.5f) * ix) + 1;
    if (x_lo < 0) x_lo = 0;
    if (x_hi > Wd - 1) x_hi = Wd - 1;
    float acc = 0.f;
    for (int x = x_lo; x <= x_hi; ++x) {
        int x0, x1;
        float lx;
        src_index(x, Ws, sx, align, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        acc = fmaf(wx, T[x], acc);
    }
    gsrc[((long)blockIdx.z * Hs + ys) * Ws + xs] = acc;
}

// tmp: workspace planes*Hs*Wd floats.
extern "C" int wc_bilinear_resize_bwd(const float* gdst, float* gsrc, float* tmp, int planes, int Hs, int Ws, int Hd,
                                      int Wd, int align_corners, void* stream) {
    WC_CHECK_ARG(gdst && gsrc && tmp && planes > 0 && planes <= 65535 && Hs > 0 && Ws > 0 && Hd >= Hs && Wd >= Ws,
                 "wc_bilinear_resize_bwd: bad argument (up-sampling only)");
    float sy, sx;
    if (align_corners) {
        sy = Hd > 1 ? (float)(Hs - 1) / (Hd - 1) : 0.f;
        sx = Wd > 1 ? (float)(Ws - 1) / (Wd - 1) : 0.f;
        WC_CHECK_ARG(sy > 0.f && sx > 0.f, "wc_bilinear_resize_bwd: degenerate size");
    } else {
        sy = (float)Hs / Hd;
        sx = (float)Ws / Wd;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bilinear_bwd_y_kernel, dim3(wc_cdiv(Wd, 64), wc_cdiv(Hs, 4), planes), dim3(256), 0, st, gdst, tmp, Hs,
                       Hd, Wd, sy, align_corners ? 1 : 0);
    WC_LAUNCH_CHECK("bilinear_bwd_y_kernel");
    hipLaunchKernelGGL(bilinear_bwd_x_kernel, dim3(wc_cdiv(Ws, 64), wc_cdiv(Hs, 4), planes), dim3(256), 0, st, tmp, gsrc, Hs,
                       Ws, Wd, sx, align_corners ? 1 : 0);
    WC_LAUNCH_CHECK("bilinear_bwd_x_kernel");
    return WC_OK;
}

extern "C" int wc_bilinear_resize(const float* src, float* dst, int planes, int Hs, int Ws, int Hd,
                                  int Wd, int align_corners, void* stream) {
    WC_CHECK_ARG(src && dst && planes > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0,
                 "wc_bilinear_resize: bad argument");
    WC_CHECK_ARG(planes <= 65535, "wc_bilinear_resize: at most 65535 planes per call");
    float sy, sx;
    if (align_corners) {
        sy = Hd > 1 ? (float)(Hs - 1) / (Hd - 1) : 0.f;
        sx = Wd > 1 ? (float)(Ws - 1) / (Wd - 1) : 0.f;
    } else {
        sy = (float)Hs / Hd;
        sx = (float)Ws / Wd;
    }
    dim3 grid(wc_cdiv(Wd, 64), wc_cdiv(Hd, 4), planes);
    hipLaunchKernelGGL(bilinear_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, Hs, Ws, Hd,
                       Wd, sy, sx, align_corners ? 1 : 0);
    WC_LAUNCH_CHECK("bilinear_kernel");
    return WC_OK;
}
