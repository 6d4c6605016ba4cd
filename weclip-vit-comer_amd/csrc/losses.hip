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

// sums[k] = sum over blocks of part[blk*4 + k]  (deterministic second stage), then the scalar tail of the loss so
// that no elementwise torch launches follow: sums[4] = the loss, sums[5..6] = d loss / d (per-pixel term) for the two
// classes of terms (the backward kernels' weights for an upstream gradient of 1).
//   mode 0 (segmentation, get_seg_loss):  0.5 * (s0 / s1 + s2 / s3);                 0.5 / s1,  0.5 / s3
//   mode 1 (affinity, get_aff_loss):      0.5 * s0 / (s1 + 1) + 0.5 * s2 / (s3 + 1); -0.5 / (s1 + 1), 0.5 / (s3 + 1)
__global__ __launch_bounds__(1024) void seg_loss_reduce_kernel(const float* __restrict__ part, float* __restrict__ sums, long nblk,
                                                               int mode) {
    __shared__ float red[16];
    __shared__ float tot[4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i = threadIdx.x; i < nblk; i += 1024) {          // one 16-byte load per partial block
        const float4 v = *reinterpret_cast<const float4*>(part + i * 4);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    for (int k = 0; k < 4; ++k) {
        const float t = block_sum(s[k], red);
        if (threadIdx.x == 0) { sums[k] = t; tot[k] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (mode == 0) {
            sums[4] = 0.5f * (tot[0] / tot[1] + tot[2] / tot[3]);
            sums[5] = 0.5f / tot[1];
            sums[6] = 0.5f / tot[3];
        } else {
            sums[4] = 0.5f * tot[0] / (tot[1] + 1.0f) + 0.5f * tot[2] / (tot[3] + 1.0f);
            sums[5] = -0.5f / (tot[1] + 1.0f);
            sums[6] = 0.5f / (tot[3] + 1.0f);
        }
        sums[7] = 0.f;
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
    hipLaunchKernelGGL(seg_loss_reduce_kernel, dim3(1), dim3(1024), 0, st, part, sums, (long)grid.x * grid.y * grid.z, 0);
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


// ---------------------------------------------------------------------------------------------
// Affinity loss of the training step, fused with the label -> affinity-label construction.
// reference utils/camutils.py:226-247 (cams_to_affinity_label: nearest down-sampling of the pseudo labels by 16,
// pairwise equality, ignore where either token is ignored or the pair is outside the radius mask of
// scripts/dist_clip_voc.py:116-133) + utils/losses.py:11-22 (get_aff_loss):
//   loss = 0.5 * sum_pos(1 - p) / (n_pos + 1) + 0.5 * sum_neg(p) / (n_neg + 1)
// The reference materialises the (B, hw, hw) affinity label, the two masks and their products (six passes over
// 67 MB at hw = 1024); here one pass reads attn_pred and the hw low-res labels of the image (LDS).
__device__ __forceinline__ int aff_lowres_label(const long* __restrict__ cam, int b, int t, int h, int w, int H, int W) {
    const int y = t / w, x = t - y * w;
    // F.interpolate(mode="nearest"): src = min(floor(dst * in / out), in - 1)
    int sy = (int)floorf(y * ((float)H / h)), sx = (int)floorf(x * ((float)W / w));
    sy = sy > H - 1 ? H - 1 : sy;
    sx = sx > W - 1 ? W - 1 : sx;
    return (int)cam[((long)b * H + sy) * W + sx];
}

template <bool BWD>
__global__ __launch_bounds__(256) void aff_loss_kernel(const float* __restrict__ ap, const long* __restrict__ cam,
                                                        float* __restrict__ part, const float* __restrict__ coef,
                                                        float* __restrict__ dap, int h, int w, int H, int W, int radius,
                                                        int ignore) {
    extern __shared__ int lab[];          // [hw] low-res labels of image b
    __shared__ float red[16];
    const int hw = h * w, b = blockIdx.y;
    for (int t = threadIdx.x; t < hw; t += 256) lab[t] = aff_lowres_label(cam, b, t, h, w, H, W);
    __syncthreads();
    // block = ROWS rows i of the (hw, hw) matrix, threads sweep j
    constexpr int ROWS = 8;
    float ps = 0.f, pc = 0.f, ns = 0.f, nc = 0.f;
    float cp = 0.f, cn = 0.f;
    if (BWD) { cp = coef[0]; cn = coef[1]; }
    for (int r = 0; r < ROWS; ++r) {
        const int i = blockIdx.x * ROWS + r;
        if (i >= hw) break;
        const int li = lab[i], yi = i / w, xi = i - yi * w;
        const float* row = ap + ((long)b * hw + i) * hw;
        for (int j = threadIdx.x; j < hw; j += 256) {
            const int lj = lab[j], yj = j / w, xj = j - yj * w;
            const int dy = yi - yj, dx = xi - xj;
            const bool ok = li != ignore && lj != ignore && dy <= radius && -dy <= radius && dx <= radius && -dx <= radius;
            const bool pos = ok && li == lj, neg = ok && li != lj;
            if (BWD) {
                dap[((long)b * hw + i) * hw + j] = pos ? cp : (neg ? cn : 0.f);
            } else {
                const float p = row[j];
                if (pos) { ps += 1.f - p; pc += 1.f; }
                if (neg) { ns += p; nc += 1.f; }
            }
        }
    }
    if (!BWD) {
        const float a0 = block_sum(ps, red), a1 = block_sum(pc, red), a2 = block_sum(ns, red), a3 = block_sum(nc, red);
        if (threadIdx.x == 0) {
            const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
            part[blk * 4] = a0; part[blk * 4 + 1] = a1; part[blk * 4 + 2] = a2; part[blk * 4 + 3] = a3;
        }
    }
}

// sums (4) = [sum_pos(1-p), n_pos, sum_neg(p), n_neg]; part: workspace 4 * B * ceil(hw/8) floats.
extern "C" int wc_aff_loss_fwd(const float* attn_pred, const int64_t* cam_label, float* part, float* sums, int B, int h,
                               int w, int H, int W, int radius, int ignore, void* stream) {
    WC_CHECK_ARG(attn_pred && cam_label && part && sums && B > 0 && h > 0 && w > 0 && H >= h && W >= w && radius >= 0,
                 "wc_aff_loss_fwd: bad argument");
    WC_CHECK_ARG((size_t)h * w * 4 <= 64 * 1024 && B <= 65535, "wc_aff_loss_fwd: h*w <= 16384");
    dim3 grid(wc_cdiv(h * w, 8), B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aff_loss_kernel<false>, grid, dim3(256), (size_t)h * w * 4, st, attn_pred, (const long*)cam_label, part,
                       nullptr, nullptr, h, w, H, W, radius, ignore);
    WC_LAUNCH_CHECK("aff_loss_kernel<fwd>");
    hipLaunchKernelGGL(seg_loss_reduce_kernel, dim3(1), dim3(1024), 0, st, part, sums, (long)grid.x * grid.y, 1);
    WC_LAUNCH_CHECK("seg_loss_reduce_kernel");
    return WC_OK;
}

// dap (B, hw, hw) = coef[0] on positive pairs, coef[1] on negative pairs, 0 elsewhere (coef: 2 device floats,
// = upstream gradient * [-0.5 / (n_pos + 1), 0.5 / (n_neg + 1)]).
extern "C" int wc_aff_loss_bwd(const int64_t* cam_label, const float* coef, float* dap, int B, int h, int w, int H, int W,
                               int radius, int ignore, void* stream) {
    WC_CHECK_ARG(cam_label && coef && dap && B > 0 && h > 0 && w > 0 && H >= h && W >= w && radius >= 0,
                 "wc_aff_loss_bwd: bad argument");
    WC_CHECK_ARG((size_t)h * w * 4 <= 64 * 1024 && B <= 65535, "wc_aff_loss_bwd: h*w <= 16384");
    hipLaunchKernelGGL(aff_loss_kernel<true>, dim3(wc_cdiv(h * w, 8), B), dim3(256), (size_t)h * w * 4, (hipStream_t)stream,
                       nullptr, (const long*)cam_label, nullptr, coef, dap, h, w, H, W, radius, ignore);
    WC_LAUNCH_CHECK("aff_loss_kernel<bwd>");
    return WC_OK;
}
