// Segmentation loss of the training step, fused with the bilinear up-sampling of the logits.
// reference scripts/dist_clip_voc.py:250 (F.interpolate(segs, size=(H,W), bilinear, align_corners=False))
// + get_seg_loss :105-113:  0.5 * (CE(pred, label with fg->ignore) + CE(pred, label with bg->ignore)),
// each CE a mean over its non-ignored pixels.  The reference materialises the (B, nc, H, W) up-sampled
// logits (352 MB at 16x21x512x512) for log_softmax forward and backward; here every high-res pixel
// interpolates its nc logits from the low-res map on the fly.
//   seg_loss_fwd_kernel : per pixel nll; block sums of (nll_bg, n_bg, nll_fg, n_fg) -> partials
//   seg_loss_bwd_kernel : per pixel g[c] = w_pix * (softmax_c - [c == label]) written as (B, nc, H, W);
//                         the low-res gradient is then the separable bilinear backward (resize.hip).
#include "common.h"

#define SEG_MAX_C 96

__device__ __forceinline__ void bil_index(int d, int in, float scale, int& i0, int& i1, float& l1) {
    float s = fmaxf(scale * (d + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - i0;
}

template <bool BWD>
__global__ __launch_bounds__(256) void seg_loss_kernel(const float* __restrict__ seg, const long* __restrict__ label,
                                                        float* __restrict__ part, const float* __restrict__ wts,
                                                        float* __restrict__ ghr, int nc, int h, int w, int H, int W,
                                                        float sy, float sx, int ignore) {
    __shared__ float red[16];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    float nll = 0.f;
    int cls = -1;   // -1: outside / ignored, 0: background pixel, 1: foreground pixel
    if (x < W && y < H) {
        const long lab = label[((long)b * H + y) * W + x];
        int y0, y1, x0, x1;
        float ly, lx;
        bil_index(y, h, sy, y0, y1, ly);
        bil_index(x, w, sx, x0, x1, lx);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        const float* S = seg + (long)b * nc * h * w;
        const long o00 = (long)y0 * w + x0, o01 = (long)y0 * w + x1, o10 = (long)y1 * w + x0, o11 = (long)y1 * w + x1;
        float mx = -INFINITY, sum = 0.f, zl = 0.f;
        for (int c = 0; c < nc; ++c) {
            const float* Sc = S + (long)c * h * w;
            const float z = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11];
            if (c == lab) zl = z;
            const float nm = fmaxf(mx, z);
            sum = sum * __expf(mx - nm) + __expf(z - nm);
            mx = nm;
        }
        const bool valid = lab != ignore && lab >= 0 && lab < nc;
        if (valid) {
            cls = lab == 0 ? 0 : 1;
            nll = mx + __logf(sum) - zl;
        }
        if (BWD) {
            const float wp = valid ? wts[cls] : 0.f;     // 0.5 / n_bg or 0.5 / n_fg (times upstream grad)
            float* G = ghr + (long)b * nc * H * W + (long)y * W + x;
            const float lse = mx + __logf(sum);
            for (int c = 0; c < nc; ++c) {
                const float* Sc = S + (long)c * h * w;
                const float z = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11];
                G[(long)c * H * W] = wp * (__expf(z - lse) - (c == lab ? 1.f : 0.f));
            }
        }
    }
    if (!BWD) {
        const float a0 = block_sum(cls == 0 ? nll : 0.f, red), a1 = block_sum(cls == 0 ? 1.f : 0.f, red);
        const float a2 = block_sum(cls == 1 ? nll : 0.f, red), a3 = block_sum(cls == 1 ? 1.f : 0.f, red);
        if (threadIdx.x == 0) {
            const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            part[blk * 4] = a0; part[blk * 4 + 1] = a1; part[blk * 4 + 2] = a2; part[blk * 4 + 3] = a3;
        }
    }
}

// sums[k] = sum over blocks of part[blk*4 + k]  (deterministic second stage)
__global__ __launch_bounds__(256) void seg_loss_reduce_kernel(const float* __restrict__ part, float* __restrict__ sums, long nblk) {
    __shared__ float red[16];
    for (int k = 0; k < 4; ++k) {
        float s = 0.f;
        for (long i = threadIdx.x; i < nblk; i += 256) s += part[i * 4 + k];
        s = block_sum(s, red);
        if (threadIdx.x == 0) sums[k] = s;
        __syncthreads();
    }
}

extern "C" int wc_seg_loss_fwd(const float* seg, const int64_t* label, float* part, float* sums, int B, int nc, int h,
                               int w, int H, int W, int ignore, void* stream) {
    WC_CHECK_ARG(seg && label && part && sums && B > 0 && nc > 0 && nc <= SEG_MAX_C && h > 0 && w > 0 && H >= h && W >= w,
                 "wc_seg_loss_fwd: bad argument");
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(H, 4), B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_loss_kernel<false>, grid, dim3(256), 0, st, seg, (const long*)label, part, nullptr, nullptr, nc,
                       h, w, H, W, (float)h / H, (float)w / W, ignore);
    WC_LAUNCH_CHECK("seg_loss_kernel<fwd>");
    hipLaunchKernelGGL(seg_loss_reduce_kernel, dim3(1), dim3(256), 0, st, part, sums, (long)grid.x * grid.y * grid.z);
    WC_LAUNCH_CHECK("seg_loss_reduce_kernel");
    return WC_OK;
}

// wts (2) device floats: gradient weight of a background / foreground pixel.  ghr: (B, nc, H, W) workspace.
extern "C" int wc_seg_loss_bwd(const float* seg, const int64_t* label, const float* wts, float* ghr, int B, int nc, int h,
                               int w, int H, int W, int ignore, void* stream) {
    WC_CHECK_ARG(seg && label && wts && ghr && B > 0 && nc > 0 && nc <= SEG_MAX_C && h > 0 && w > 0 && H >= h && W >= w,
                 "wc_seg_loss_bwd: bad argument");
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(H, 4), B);
    hipLaunchKernelGGL(seg_loss_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, seg, (const long*)label, nullptr, wts,
                       ghr, nc, h, w, H, W, (float)h / H, (float)w / W, ignore);
    WC_LAUNCH_CHECK("seg_loss_kernel<bwd>");
    return WC_OK;
}

// Fused backward: d loss / d seg (B, nc, h, w) in ONE pass, without the (B, nc, H, W) high-res gradient
// (352 MB written and read back at 16x21x512x512) or the separable resize-backward passes.
// Block = one low-res row (b, ys): it walks the high-res rows Y that interpolate from row ys; per Y,
//   phase 1: thread per X -> log-sum-exp of the nc interpolated logits, pixel weight, label  (LDS)
//   phase 2: thread per (class, xs) -> sum over the X that interpolate from column xs of
//            wy * wx * w_pix * (softmax_c - [c == label]), accumulated in registers over all Y.
// The three low-res rows ys-1..ys+1 of every class sit in LDS; a high-res row is visited by the (at most
// two) blocks whose low-res rows it reads.
#define SEG_BWD_ITEMS 8     // (class, xs) outputs per thread: nc * w <= 256 * 8
__global__ __launch_bounds__(256) void seg_loss_bwd_fused_kernel(const float* __restrict__ seg, const long* __restrict__ label,
                                                                  const float* __restrict__ wts, float* __restrict__ dseg,
                                                                  int nc, int h, int w, int H, int W, float sy, float sx,
                                                                  int ignore) {
    extern __shared__ float sm[];
    float* rows = sm;                          // [3][nc][w]
    float* lse_s = rows + 3 * nc * w;          // [W]
    float* wp_s = lse_s + W;                   // [W]
    int* lab_s = reinterpret_cast<int*>(wp_s + W);
    const int tid = threadIdx.x, ys = blockIdx.x, b = blockIdx.y;
    const float* S = seg + (long)b * nc * h * w;
    const int ncw = nc * w;
    for (int i = tid; i < 3 * ncw; i += 256) {
        const int r = i / ncw, rem = i - r * ncw, c = rem / w, x = rem - c * w;
        int yy = ys - 1 + r;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        rows[i] = S[((long)c * h + yy) * w + x];
    }
    const float w_bg = wts[0], w_fg = wts[1];
    float acc[SEG_BWD_ITEMS];
#pragma unroll
    for (int k = 0; k < SEG_BWD_ITEMS; ++k) acc[k] = 0.f;
    int Ylo = (int)floorf((ys - 0.5f) / sy - 0.5f) - 1, Yhi = (int)ceilf((ys + 1.5f) / sy - 0.5f) + 1;
    Ylo = Ylo < 0 ? 0 : Ylo;
    Yhi = Yhi > H - 1 ? H - 1 : Yhi;
    __syncthreads();
    for (int Y = Ylo; Y <= Yhi; ++Y) {
        int y0, y1;
        float ly;
        bil_index(Y, h, sy, y0, y1, ly);
        const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
        if (wy == 0.f) continue;               // block-uniform
        const float* R0 = rows + (y0 - ys + 1) * ncw;
        const float* R1 = rows + (y1 - ys + 1) * ncw;
        for (int X = tid; X < W; X += 256) {
            int x0, x1;
            float lx;
            bil_index(X, w, sx, x0, x1, lx);
            const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
            float mx = -INFINITY, sum = 0.f;
            for (int c = 0; c < nc; ++c) {
                const float z = w00 * R0[c * w + x0] + w01 * R0[c * w + x1] + w10 * R1[c * w + x0] + w11 * R1[c * w + x1];
                const float nm = fmaxf(mx, z);
                sum = sum * __expf(mx - nm) + __expf(z - nm);
                mx = nm;
            }
            const long lab = label[((long)b * H + Y) * W + X];
            const bool valid = lab != ignore && lab >= 0 && lab < nc;
            lse_s[X] = mx + __logf(sum);
            wp_s[X] = valid ? (lab == 0 ? w_bg : w_fg) : 0.f;
            lab_s[X] = valid ? (int)lab : -1;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SEG_BWD_ITEMS; ++k) {
            const int i = tid + 256 * k;
            if (i < ncw) {
                const int c = i / w, xs = i - c * w;
                int Xlo = (int)floorf((xs - 0.5f) / sx - 0.5f) - 1, Xhi = (int)ceilf((xs + 1.5f) / sx - 0.5f) + 1;
                Xlo = Xlo < 0 ? 0 : Xlo;
                Xhi = Xhi > W - 1 ? W - 1 : Xhi;
                float a = 0.f;
                for (int X = Xlo; X <= Xhi; ++X) {
                    int x0, x1;
                    float lx;
                    bil_index(X, w, sx, x0, x1, lx);
                    const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
                    if (wx == 0.f) continue;
                    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
                    const float z = w00 * R0[c * w + x0] + w01 * R0[c * w + x1] + w10 * R1[c * w + x0] + w11 * R1[c * w + x1];
                    a += wx * wp_s[X] * (__expf(z - lse_s[X]) - (c == lab_s[X] ? 1.f : 0.f));
                }
                acc[k] += wy * a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < SEG_BWD_ITEMS; ++k) {
        const int i = tid + 256 * k;
        if (i < ncw) {
            const int c = i / w, xs = i - c * w;
            dseg[(((long)b * nc + c) * h + ys) * w + xs] = acc[k];
        }
    }
}

// dseg (B, nc, h, w) = gradient of the fused loss w.r.t. the low-res logits.  wts as in wc_seg_loss_bwd.
extern "C" int wc_seg_loss_bwd_fused(const float* seg, const int64_t* label, const float* wts, float* dseg, int B, int nc,
                                     int h, int w, int H, int W, int ignore, void* stream) {
    WC_CHECK_ARG(seg && label && wts && dseg && B > 0 && nc > 0 && nc <= SEG_MAX_C && h > 0 && w > 0 && H >= h && W >= w,
                 "wc_seg_loss_bwd_fused: bad argument");
    WC_CHECK_ARG(nc * w <= 256 * SEG_BWD_ITEMS && B <= 65535, "wc_seg_loss_bwd_fused: nc * w must be <= %d", 256 * SEG_BWD_ITEMS);
    const size_t lds = ((size_t)3 * nc * w + 3 * (size_t)W) * 4;
    WC_CHECK_ARG(lds <= 64 * 1024, "wc_seg_loss_bwd_fused: rows do not fit LDS (nc=%d w=%d W=%d)", nc, w, W);
    hipLaunchKernelGGL(seg_loss_bwd_fused_kernel, dim3(h, B), dim3(256), lds, (hipStream_t)stream, seg, (const long*)label, wts,
                       dseg, nc, h, w, H, W, (float)h / H, (float)w / W, ignore);
    WC_LAUNCH_CHECK("seg_loss_bwd_fused_kernel");
    return WC_OK;
}
