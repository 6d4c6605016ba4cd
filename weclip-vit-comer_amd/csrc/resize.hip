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
