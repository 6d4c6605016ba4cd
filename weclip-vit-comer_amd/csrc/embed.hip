// Patch-embedding front end of the ViT (reference clip/model.py:264-272).
// conv2d(3->D, k=P, s=P, no bias) on non-overlapping patches is a GEMM over an im2col matrix
// that is a pure re-indexing of the image: patchify_kernel writes it as fp16 hi(+lo) rows
// (b, py, px) x (c, ky, kx) -- the weight's own (D, 3, P, P) flattening -- and wc_gemm_f16 does
// the rest (epilogue adds the resized position embedding, rows 1.. of each image).
// cls_rows_kernel writes token 0 of every image: class_embedding + pos[0].
#include "common.h"

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                        __half* __restrict__ hi, __half* __restrict__ lo,
                                                        int H, int W, int P, long total4) {
    // one thread = 4 consecutive kx of one (row, c, ky)
    const int h = H / P, w = W / P, K = 3 * P * P, q = P / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int kx4 = i % q;
        long r = i / q;
        const int ky = r % P; r /= P;
        const int c = r % 3; r /= 3;           // r = patch row index (b*h*w + py*w + px)
        const int px = r % w;
        const long r2 = r / w;
        const int py = r2 % h;
        const long b = r2 / h;
        const float* src = img + ((b * 3 + c) * H + (long)py * P + ky) * W + (long)px * P + kx4 * 4;
        float f[4];
        if (W % 4 == 0) {        // rows 16-byte aligned
            const float4 v = *reinterpret_cast<const float4*>(src);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
            f[0] = src[0]; f[1] = src[1]; f[2] = src[2]; f[3] = src[3];
        }
        __half hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hh[k] = __float2half(f[k]);
            ll[k] = __float2half(f[k] - __half2float(hh[k]));
        }
        const long o = r * K + ((long)c * P + ky) * P + kx4 * 4;
        *reinterpret_cast<uint2*>(hi + o) = *reinterpret_cast<uint2*>(hh);
        if (lo) *reinterpret_cast<uint2*>(lo + o) = *reinterpret_cast<uint2*>(ll);
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                        const float* __restrict__ pos0, long ldb, int E) {
    for (int e = threadIdx.x; e < E; e += 256) x[(long)blockIdx.x * ldb + e] = cls[e] + pos0[e];
}

extern "C" int wc_patchify(const float* img, void* hi, void* lo, int B, int H, int W, int P, void* stream) {
    // a stride-P convolution ignores the H % P bottom rows / W % P right columns (multi-scale inference feeds such sizes)
    WC_CHECK_ARG(img && hi && B > 0 && P > 0 && P % 4 == 0 && H >= P && W >= P,
                 "wc_patchify: need H, W >= patch size and patch %% 4 == 0");
    const long total4 = (long)B * (H / P) * (W / P) * 3 * P * (P / 4);
    int blocks = wc_cdiv(total4, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (__half*)hi,
                       (__half*)lo, H, W, P, total4);
    WC_LAUNCH_CHECK("patchify_kernel");
    return WC_OK;
}

extern "C" int wc_cls_rows(float* x, const float* cls, const float* pos0, int B, int L, int E, void* stream) {
    WC_CHECK_ARG(x && cls && pos0 && B > 0 && L > 0 && E > 0, "wc_cls_rows: bad argument");
    hipLaunchKernelGGL(cls_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, cls, pos0,
                       (long)L * E, E);
    WC_LAUNCH_CHECK("cls_rows_kernel");
    return WC_OK;
}
