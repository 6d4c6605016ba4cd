// Device-side input pipeline of the training step: uint8 HWC image -> random rescale ->
// horizontal flip -> zero pad + random crop -> mean/std normalise -> float32 CHW.
//
// Replaces the host-side numpy / PIL chain of the reference loader, per image:
//   datasets/transforms.py:26-49  random_scaling (PIL Image.BILINEAR resize to (int(s*w), int(s*h)), uint8 result)
//   datasets/transforms.py:70-84  random_fliplr
//   datasets/transforms.py:119-176 random_crop (pad to >= crop with mean_rgb = [0,0,0] at a random offset, crop window)
//   datasets/transforms.py:8-15   normalize_img ((x - mean) / std per channel), then HWC -> CHW (datasets/voc.py:137-143)
// The random draws stay on the host (a few scalars per image, data.DeviceAugment); the kernels apply them.
//
// PIL's BILINEAR (Pillow, ImagingResample 8 bits per channel) is NOT plain half-pixel bilinear: it is a separable
// triangle filter whose support grows with the down-scaling ratio (support = max(in / out, 1), up to
// 2 * ceil(support) + 1 taps), evaluated with coefficients normalised in double precision and rounded to 22-bit fixed
// point, a horizontal pass whose result is rounded to uint8, then a vertical pass over those uint8 values.  The two
// kernels below follow that arithmetic step by step (same double-precision operation order, same integer rounding), so
// the output equals PIL's on every pixel, up- and down-scaling (tests/golden/augment_ref.npz, made by the reference's
// own transforms with real PIL).
//   aug_coeff_kernel: per image and axis, for the `crop` output coordinates only: first tap, tap count and the
//                     fixed-point coefficients (or "outside the rescaled image": zero padding);
//   augment_normalize_kernel: one thread per output pixel, ny x nx taps (3 x 3 when up-scaling, 5 x 5 at scale 0.5).
#include "common.h"

#define AUG_KMAX 9          // 2 * ceil(support) + 1 with support <= 4, i.e. down-scaling by at most 4
#define AUG_ENT 12          // ints per table entry: first tap, count, AUG_KMAX coefficients, pad (48 B)
#define AUG_PREC 22         // Pillow: PRECISION_BITS = 32 - 8 - 2

struct AugParams {      // one per image, 8 ints / floats = 32 B
    float scale;        // s of random_scaling
    int flip;           // 1: np.fliplr
    int rh, rw;         // rescaled size (int(s*h), int(s*w))
    int pad_y, pad_x;   // where the rescaled image sits in the padded canvas (H_pad, W_pad)
    int crop_y, crop_x; // crop window origin in the canvas (H_start, W_start)
};

// grid (cdiv(crop, 256), 2, B); axis 0 = rows, 1 = columns
__global__ __launch_bounds__(256) void aug_coeff_kernel(const AugParams* __restrict__ params, int* __restrict__ tab, int Hs, int Ws,
                                                         int crop) {
    const int o = blockIdx.x * 256 + threadIdx.x, axis = blockIdx.y, b = blockIdx.z;
    if (o >= crop) return;
    const AugParams p = params[b];
    const int in_size = axis ? Ws : Hs, out_size = axis ? p.rw : p.rh;
    int r = o + (axis ? p.crop_x - p.pad_x : p.crop_y - p.pad_y);          // coordinate in the rescaled image
    int* e = tab + (((long)b * 2 + axis) * crop + o) * AUG_ENT;
    if (r < 0 || r >= out_size) {
        e[0] = 0;
        e[1] = 0;                                                          // canvas padding
        return;
    }
    if (axis && p.flip) r = out_size - 1 - r;
    // Pillow precompute_coeffs(inSize, in0 = 0, in1 = inSize, outSize, BILINEAR) for output coordinate r
    const double scale = (double)((float)in_size - 0.f) / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fscale;
    const double center = 0.0 + (r + 0.5) * scale;
    const double ss = 1.0 / fscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    int n = xmax - xmin;
    if (n > AUG_KMAX) {                                                    // down-scaling beyond 4x: the table cannot hold Pillow's
        e[0] = 0;                                                          // filter -> the pixel is POISONED (NaN), never a
        e[1] = -1;                                                         // silently different filter (checked precondition)
        return;
    }
    double k[AUG_KMAX], ww = 0.0;
#pragma unroll
    for (int x = 0; x < AUG_KMAX; ++x) {
        double a = ((double)(x + xmin) - center + 0.5) * ss;
        a = a < 0.0 ? -a : a;
        const double w = (x < n && a < 1.0) ? 1.0 - a : 0.0;
        k[x] = w;
        ww += w;
    }
    e[0] = xmin;
    e[1] = n;
#pragma unroll
    for (int x = 0; x < AUG_KMAX; ++x) {
        const double v = ww != 0.0 ? k[x] / ww : k[x];
        e[2 + x] = (int)(0.5 + v * (double)(1 << AUG_PREC));               // normalize_coeffs_8bpc (all coefficients >= 0)
    }
    e[2 + AUG_KMAX] = 0;
}

__device__ __forceinline__ int aug_clip8(int v) {
    v >>= AUG_PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void augment_normalize_kernel(const unsigned char* __restrict__ src, const int* __restrict__ tab,
                                                                 float* __restrict__ dst, int Hs, int Ws, int crop,
                                                                 float m0, float m1, float m2, float s0, float s1, float s2) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= crop || y >= crop) return;
    const int* ey = tab + (((long)b * 2 + 0) * crop + y) * AUG_ENT;       // one entry per wave
    const int* ex = tab + (((long)b * 2 + 1) * crop + x) * AUG_ENT;
    const int ymin = ey[0], ny = ey[1], xmin = ex[0], nx = ex[1];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;                                    // canvas padding (mean_rgb = [0, 0, 0])
    if (ny == 0 || nx == 0) {
        // outside the rescaled image on either axis: canvas padding, whatever the other axis' table says
    } else if (ny < 0 || nx < 0) {                                         // precondition in/out <= 4 violated (aug_coeff_kernel)
        v0 = v1 = v2 = __builtin_nanf("");
    } else {
        const unsigned char* S = src + ((long)b * Hs * Ws + (long)ymin * Ws + xmin) * 3;
        int a0 = 1 << (AUG_PREC - 1), a1 = a0, a2 = a0;
        for (int j = 0; j < ny; ++j) {
            const unsigned char* row = S + (long)j * Ws * 3;
            int h0 = 1 << (AUG_PREC - 1), h1 = h0, h2 = h0;                // horizontal pass of source row ymin + j
            for (int i = 0; i < nx; ++i) {
                const int kx = ex[2 + i];
                h0 += kx * row[3 * i];
                h1 += kx * row[3 * i + 1];
                h2 += kx * row[3 * i + 2];
            }
            const int ky = ey[2 + j];                                      // vertical pass over the uint8-rounded rows
            a0 += ky * aug_clip8(h0);
            a1 += ky * aug_clip8(h1);
            a2 += ky * aug_clip8(h2);
        }
        v0 = (float)aug_clip8(a0);
        v1 = (float)aug_clip8(a1);
        v2 = (float)aug_clip8(a2);
    }
    const long plane = (long)crop * crop;
    float* D = dst + (long)b * 3 * plane + (long)y * crop + x;
    D[0] = (v0 - m0) / s0;
    D[plane] = (v1 - m1) / s1;
    D[2 * plane] = (v2 - m2) / s2;
}

extern "C" int wc_augment_workspace_ints(int B, int crop, long* n_ints) {
    WC_CHECK_ARG(n_ints && B > 0 && crop > 0, "wc_augment_workspace_ints: bad argument");
    *n_ints = (long)B * 2 * crop * AUG_ENT;
    return WC_OK;
}

extern "C" int wc_augment_normalize(const void* src_u8, const void* params, float* dst, int* coeff_ws, int B, int Hs, int Ws,
                                    int crop, const float* mean3, const float* std3, void* stream) {
    WC_CHECK_ARG(src_u8 && params && dst && coeff_ws && mean3 && std3 && B > 0 && B <= 65535 && Hs > 0 && Ws > 0 && crop > 0,
                 "wc_augment_normalize: bad argument");
    WC_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "wc_augment_normalize: zero std");
    hipLaunchKernelGGL(aug_coeff_kernel, dim3(wc_cdiv(crop, 256), 2, B), dim3(256), 0, (hipStream_t)stream,
                       (const AugParams*)params, coeff_ws, Hs, Ws, crop);
    WC_LAUNCH_CHECK("aug_coeff_kernel");
    hipLaunchKernelGGL(augment_normalize_kernel, dim3(wc_cdiv(crop, 64), wc_cdiv(crop, 4), B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src_u8, (const int*)coeff_ws, dst, Hs, Ws, crop, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2]);
    WC_LAUNCH_CHECK("augment_normalize_kernel");
    return WC_OK;
}
