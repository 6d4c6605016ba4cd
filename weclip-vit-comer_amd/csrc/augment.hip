// Device-side input pipeline of the training step: uint8 HWC image -> random rescale ->
// horizontal flip -> zero pad + random crop -> mean/std normalise -> float32 CHW, one gather kernel per batch.
//
// Replaces the host-side numpy / PIL chain of the reference loader, per image:
//   datasets/transforms.py:26-49  random_scaling (PIL BILINEAR resize to (int(s*w), int(s*h)), result rounded to uint8)
//   datasets/transforms.py:70-84  random_fliplr
//   datasets/transforms.py:119-176 random_crop (pad to >= crop with mean_rgb = [0,0,0] at a random offset, crop window)
//   datasets/transforms.py:8-15   normalize_img ((x - mean) / std per channel), then HWC -> CHW (datasets/voc.py:137-143)
// The random draws stay on the host (a few scalars per image, data.DeviceAugment); this kernel applies them.
// Resampling is half-pixel bilinear (what PIL's BILINEAR is for s >= 1); for s < 1 PIL widens the filter support
// (area averaging), which this kernel does not reproduce: "parity unpinned vs PIL" for down-scaling.
#include "common.h"

struct AugParams {      // one per image, 8 ints / floats = 32 B
    float scale;        // s of random_scaling
    int flip;           // 1: np.fliplr
    int rh, rw;         // rescaled size (int(s*h), int(s*w))
    int pad_y, pad_x;   // where the rescaled image sits in the padded canvas (H_pad, W_pad)
    int crop_y, crop_x; // crop window origin in the canvas (H_start, W_start)
};

__global__ __launch_bounds__(256) void augment_normalize_kernel(const unsigned char* __restrict__ src,
                                                                 const AugParams* __restrict__ params,
                                                                 float* __restrict__ dst, int Hs, int Ws, int crop,
                                                                 float m0, float m1, float m2, float s0, float s1, float s2) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= crop || y >= crop) return;
    const AugParams p = params[b];
    const int iy = y + p.crop_y - p.pad_y;
    int ix = x + p.crop_x - p.pad_x;
    float v[3] = {0.f, 0.f, 0.f};                       // canvas padding (mean_rgb = [0, 0, 0])
    if (iy >= 0 && iy < p.rh && ix >= 0 && ix < p.rw) {
        if (p.flip) ix = p.rw - 1 - ix;
        const float fy = fmaxf((float)Hs / p.rh * (iy + 0.5f) - 0.5f, 0.f);
        const float fx = fmaxf((float)Ws / p.rw * (ix + 0.5f) - 0.5f, 0.f);
        int y0 = (int)fy, x0 = (int)fx;
        if (y0 > Hs - 1) y0 = Hs - 1;
        if (x0 > Ws - 1) x0 = Ws - 1;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const unsigned char* S = src + (long)b * Hs * Ws * 3;
        const unsigned char* p00 = S + ((long)y0 * Ws + x0) * 3;
        const unsigned char* p01 = S + ((long)y0 * Ws + x1) * 3;
        const unsigned char* p10 = S + ((long)y1 * Ws + x0) * 3;
        const unsigned char* p11 = S + ((long)y1 * Ws + x1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            v[c] = fminf(fmaxf(floorf(t + 0.5f), 0.f), 255.f);     // the reference's resize returns uint8
        }
    }
    const long plane = (long)crop * crop;
    float* D = dst + (long)b * 3 * plane + (long)y * crop + x;
    D[0] = (v[0] - m0) / s0;
    D[plane] = (v[1] - m1) / s1;
    D[2 * plane] = (v[2] - m2) / s2;
}

extern "C" int wc_augment_normalize(const void* src_u8, const void* params, float* dst, int B, int Hs, int Ws, int crop,
                                    const float* mean3, const float* std3, void* stream) {
    WC_CHECK_ARG(src_u8 && params && dst && mean3 && std3 && B > 0 && B <= 65535 && Hs > 0 && Ws > 0 && crop > 0,
                 "wc_augment_normalize: bad argument");
    WC_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "wc_augment_normalize: zero std");
    hipLaunchKernelGGL(augment_normalize_kernel, dim3(wc_cdiv(crop, 64), wc_cdiv(crop, 4), B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src_u8, (const AugParams*)params, dst, Hs, Ws, crop, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2]);
    WC_LAUNCH_CHECK("augment_normalize_kernel");
    return WC_OK;
}
