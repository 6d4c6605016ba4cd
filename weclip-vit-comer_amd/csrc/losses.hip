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


// Backward of the fused up-sampling + cross-entropy WITHOUT the (B, nc, H, W) high-resolution gradient (352 MB written and
// read back at 16 x 21 x 512 x 512): bilinear interpolation is separable, so the low-resolution gradient is
//   tmp[b,c,ys,X] = sum_Y wy(Y -> ys) * g[b,c,Y,X]      (this kernel: g is formed on the fly, per pixel, from the logits)
//   out[b,c,ys,xs] = sum_X wx(X -> xs) * tmp[b,c,ys,X]  (seg_bwd_x_kernel)
// A thread owns one high-resolution column X of one low-resolution row ys and walks the ~2/sy rows Y that touch ys in
// ascending order (fixed summation order: deterministic; for nc > 24 bit-identical to the two-kernel path it replaces, for
// nc <= 24 the log-sum-exp is formed max-first instead of online: same value to rounding); every pixel's soft-max is
// evaluated for both low-resolution rows it feeds (2 x the exponentials, none of the HBM traffic).
template <int NCT>
__global__ __launch_bounds__(256) void seg_loss_bwd_y_kernel(const float* __restrict__ seg, const long* __restrict__ label,
                                                              const float* __restrict__ wts, float* __restrict__ tmp, int nc,
                                                              int h, int w, int H, int W, float sy, float sx, int ignore) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), ys = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || ys >= h) return;
    int x0, x1;
    float lx;
    bil_index(x, w, sx, x0, x1, lx);
    const float iy = 1.0f / sy;
    int y_lo = (int)floorf((ys - 1.5f) * iy) - 1, y_hi = (int)ceilf((ys + 1.5f) * iy) + 1;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > H - 1) y_hi = H - 1;
    float acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = 0.f;
    const float* S = seg + (long)b * nc * h * w;
    const float wbg = wts[0], wfg = wts[1];
    // CACHE (small class counts): the four low-resolution neighbours of this column for every class live in registers and
    // are re-read only when the source row pair (y0, y1) changes (every ~1/sy rows): a row then costs FMAs and
    // exponentials, not 2 x 4 x nc loads
    constexpr bool CACHE = NCT <= 24;
    float s00[CACHE ? NCT : 1], s01[CACHE ? NCT : 1], s10[CACHE ? NCT : 1], s11[CACHE ? NCT : 1];
    int cy0 = -1, cy1 = -1;
    long labs[8];
    for (int y = y_lo; y <= y_hi; ++y) {
        if (((y - y_lo) & 7) == 0) {               // the labels of the next 8 rows in one go: 8 independent loads in flight
#pragma unroll
            for (int u = 0; u < 8; ++u) labs[u] = label[((long)b * H + min(y + u, H - 1)) * W + x];
        }
        int y0, y1;
        float ly;
        bil_index(y, h, sy, y0, y1, ly);
        const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);       // wave-uniform
        long lab = labs[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) lab = ((y - y_lo) & 7) == u ? labs[u] : lab;
        if (wy == 0.f) continue;
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        const long o00 = (long)y0 * w + x0, o01 = (long)y0 * w + x1, o10 = (long)y1 * w + x0, o11 = (long)y1 * w + x1;
        if constexpr (CACHE) {
            if (y0 != cy0 || y1 != cy1) {          // wave-uniform
                cy0 = y0; cy1 = y1;
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    if (c < nc) {
                        const float* Sc = S + (long)c * h * w;
                        s00[c] = Sc[o00]; s01[c] = Sc[o01]; s10[c] = Sc[o10]; s11[c] = Sc[o11];
                    }
                }
            }
        }
        float mx = -INFINITY, sum = 0.f;
        float zc[CACHE ? NCT : 1];
        if constexpr (CACHE) {
            // maximum first, then independent exponentials (the online form's max -> rescale -> add chain is serial over
            // the classes); the interpolated logits stay in registers for the gradient pass
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                zc[c] = c < nc ? w00 * s00[c] + w01 * s01[c] + w10 * s10[c] + w11 * s11[c] : -INFINITY;
                mx = fmaxf(mx, zc[c]);
            }
#pragma unroll
            for (int c = 0; c < NCT; ++c) sum += c < nc ? __expf(zc[c] - mx) : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                if (c < nc) {
                    const float* Sc = S + (long)c * h * w;
                    const float z = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11];
                    const float nm = fmaxf(mx, z);
                    sum = sum * __expf(mx - nm) + __expf(z - nm);
                    mx = nm;
                }
            }
        }
        const bool valid = lab != ignore && lab >= 0 && lab < nc;
        const float wp = valid ? (lab == 0 ? wbg : wfg) : 0.f;
        const float lse = mx + __logf(sum);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            if (c < nc) {
                float z;
                if constexpr (CACHE) z = zc[c];
                else { const float* Sc = S + (long)c * h * w; z = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11]; }
                const float gval = wp * (__expf(z - lse) - (c == lab ? 1.f : 0.f));
                acc[c] = fmaf(wy, gval, acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
        if (c < nc) tmp[(((long)b * nc + c) * h + ys) * W + x] = acc[c];
}

__global__ __launch_bounds__(256) void seg_bwd_x_kernel(const float* __restrict__ tmp, float* __restrict__ gsrc, int Hs, int Ws,
                                                         int Wd, float sx) {
    const int xs = blockIdx.x * 64 + (threadIdx.x & 63), ys = blockIdx.y * 4 + (threadIdx.x >> 6);
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
        bil_index(x, Ws, sx, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        acc = fmaf(wx, T[x], acc);
    }
    gsrc[((long)blockIdx.z * Hs + ys) * Ws + xs] = acc;
}

// d loss / d seg (B, nc, h, w) of get_seg_loss(F.interpolate(seg, (H, W)), label) in two launches.
// wts (2) device floats: gradient weight of a background / foreground pixel.  tmp: workspace B*nc*h*W floats.
extern "C" int wc_seg_loss_bwd_fused(const float* seg, const int64_t* label, const float* wts, float* tmp, float* out, int B,
                                     int nc, int h, int w, int H, int W, int ignore, void* stream) {
    WC_CHECK_ARG(seg && label && wts && tmp && out && B > 0 && B <= 65535 && nc > 0 && nc <= SEG_MAX_C && (long)B * nc <= 65535 &&
                 h > 0 && w > 0 && H >= h && W >= w, "wc_seg_loss_bwd_fused: bad argument");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(h, 4), B);
    if (nc <= 24)
        hipLaunchKernelGGL(seg_loss_bwd_y_kernel<24>, grid, dim3(256), 0, st, seg, (const long*)label, wts, tmp, nc, h, w, H, W,
                           (float)h / H, (float)w / W, ignore);
    else
        hipLaunchKernelGGL(seg_loss_bwd_y_kernel<SEG_MAX_C>, grid, dim3(256), 0, st, seg, (const long*)label, wts, tmp, nc, h, w, H,
                           W, (float)h / H, (float)w / W, ignore);
    WC_LAUNCH_CHECK("seg_loss_bwd_y_kernel");
    hipLaunchKernelGGL(seg_bwd_x_kernel, dim3(wc_cdiv(w, 64), wc_cdiv(h, 4), B * nc), dim3(256), 0, st, tmp, out, h, w, W,
                       (float)w / W);
    WC_LAUNCH_CHECK("seg_bwd_x_kernel");
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
