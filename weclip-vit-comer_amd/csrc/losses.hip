// Segmentation loss of the training step, fused with the bilinear up-sampling of the logits.
// reference scripts/dist_clip_voc.py:250 (F.interpolate(segs, size=(H,W), bilinear, align_corners=False))
// + get_seg_loss :105-113:  0.5 * (CE(pred, label with fg->ignore) + CE(pred, label with bg->ignore)),
// each CE a mean over its non-ignored pixels.  The reference materialises the (B, nc, H, W) up-sampled
// logits (352 MB at 16x21x512x512) for log_softmax forward and backward; here every high-res pixel
// interpolates its nc logits from the low-res map on the fly.
//   seg_loss_kernel<false>  : per pixel nll; block sums of (nll_bg, n_bg, nll_fg, n_fg) -> partials (forward only)
//   seg_loss_kernel<true>   : per pixel g[c] = w_pix * (softmax_c - [c == label]) written as (B, nc, H, W);
//                             the low-res gradient is then the separable bilinear backward (resize.hip).
//   seg_loss_bwd_y_kernel   : the same gradient formed inside the Y pass (no (B, nc, H, W) tensor), + seg_bwd_x_kernel
//   seg_loss_fused_kernel   : TRAINING: loss and gradient in one pixel pass (nc <= 24), + seg_count / seg_bwd_x2 kernels
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
                for (int c = 0; c < NCT; ++c) {          // (classes past nc re-read class nc - 1: no branch around the loads)
                    const float* Sc = S + (long)min(c, nc - 1) * h * w;
                    s00[c] = Sc[o00]; s01[c] = Sc[o01]; s10[c] = Sc[o10]; s11[c] = Sc[o11];
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
            // (the logits of eight classes are requested together from clamped class indices; a class past nc enters the online
            //  log-sum-exp as -inf: exp(-inf - max) = 0, the maximum is unchanged -- behind `if (c < nc)` every class's four
            //  loads were waited for before the next class's were issued: 81 dependent latencies per pixel row at COCO's nc)
#pragma unroll
            for (int c0 = 0; c0 < NCT; c0 += 8) {
                float zz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    const float* Sc = S + (long)min(c, nc - 1) * h * w;
                    zz[u] = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c0 + u < nc) {
                        const float nm = fmaxf(mx, zz[u]);
                        sum = sum * __expf(mx - nm) + __expf(zz[u] - nm);
                        mx = nm;
                    }
                }
            }
        }
        const bool valid = lab != ignore && lab >= 0 && lab < nc;
        const float wp = valid ? (lab == 0 ? wbg : wfg) : 0.f;
        const float lse = mx + __logf(sum);
        if constexpr (CACHE) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                if (c < nc) {
                    const float gval = wp * (__expf(zc[c] - lse) - (c == lab ? 1.f : 0.f));
                    acc[c] = fmaf(wy, gval, acc[c]);
                }
            }
        } else {
#pragma unroll
            for (int c0 = 0; c0 < NCT; c0 += 8) {
                float zz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* Sc = S + (long)min(c0 + u, nc - 1) * h * w;
                    zz[u] = w00 * Sc[o00] + w01 * Sc[o01] + w10 * Sc[o10] + w11 * Sc[o11];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    if (c < nc) {
                        const float gval = wp * (__expf(zz[u] - lse) - (c == lab ? 1.f : 0.f));
                        acc[c] = fmaf(wy, gval, acc[c]);
                    }
                }
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
// Training form: loss AND its gradient in one pass over the pixels (the soft-max of a pixel is formed once; the
// separate forward + gradient passes formed it three times: once for the loss and once for each of the two low-resolution
// rows a pixel contributes to).  The gradient needs the class weights 0.5 / n_bg, 0.5 / n_fg before the pass: a label
// count runs first (SEG_CNT_BLOCKS block partials, summed in a fixed order by every workgroup of the main pass).
// Thread (x, ys): the pixels of column x whose upper source row is ys.  Their logits are a_c + ly * (b_c - a_c) with
// a_c, b_c = the x-interpolated logits of rows ys, ys + 1 (registers); their gradient goes to row ys with weight 1 - ly
// (tmpA) and to row ys + 1 with ly (tmpB); the X pass adds the two in a fixed order.
#define SEG_CNT_BLOCKS 1024
__global__ __launch_bounds__(256) void seg_count_kernel(const long* __restrict__ label, float* __restrict__ cnt, long n, int nc,
                                                         int ignore) {
    __shared__ float red[16];
    float nb = 0.f, nf = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long lab = label[i];
        const bool valid = lab != ignore && lab >= 0 && lab < nc;
        nb += (valid && lab == 0) ? 1.f : 0.f;
        nf += (valid && lab != 0) ? 1.f : 0.f;
    }
    const float a = block_sum(nb, red), b = block_sum(nf, red);      // integers < 2^24: exact
    if (threadIdx.x == 0) { cnt[2 * blockIdx.x] = a; cnt[2 * blockIdx.x + 1] = b; }
}

template <int NCT>
__global__ __launch_bounds__(256) void seg_loss_fused_kernel(const float* __restrict__ seg, const long* __restrict__ label,
                                                              const float* __restrict__ cnt, float* __restrict__ part,
                                                              float* __restrict__ tmpA, float* __restrict__ tmpB, int nc, int h,
                                                              int w, int H, int W, float sy, float sx, int ignore) {
    __shared__ float red[16];
    float pb = 0.f, pf = 0.f;
    for (int i = threadIdx.x; i < SEG_CNT_BLOCKS; i += 256) { pb += cnt[2 * i]; pf += cnt[2 * i + 1]; }
    const float nbg = block_sum(pb, red), nfg = block_sum(pf, red);
    const float wbg = nbg > 0.f ? 0.5f / nbg : 0.f, wfg = nfg > 0.f ? 0.5f / nfg : 0.f;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), ys = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    float l_bg = 0.f, c_bg = 0.f, l_fg = 0.f, c_fg = 0.f;
    if (x < W && ys < h) {
        int x0, x1;
        float lx;
        bil_index(x, w, sx, x0, x1, lx);
        const int y1s = ys + (ys < h - 1 ? 1 : 0);
        const float* S = seg + (long)b * nc * h * w;
        float av[NCT], dv[NCT], accT[NCT], accB[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            accT[c] = accB[c] = 0.f;
            // (classes past nc read class nc - 1 and are masked: behind an `if (c < nc)` hipcc waited for the four loads of a
            //  class before issuing the next class's -- 24 dependent latencies at the head of every thread)
            const float* Sc = S + (long)min(c, nc - 1) * h * w;
            const float a = (1.f - lx) * Sc[ys * w + x0] + lx * Sc[ys * w + x1];
            const float bb = (1.f - lx) * Sc[y1s * w + x0] + lx * Sc[y1s * w + x1];
            av[c] = c < nc ? a : 0.f;
            dv[c] = c < nc ? bb - a : 0.f;
        }
        const float iy = 1.0f / sy;
        int y_lo = ys == 0 ? 0 : (int)floorf((ys + 0.5f) * iy - 0.5f) - 1;
        int y_hi = ys == h - 1 ? H - 1 : (int)ceilf((ys + 1.5f) * iy - 0.5f) + 1;
        if (y_lo < 0) y_lo = 0;
        if (y_hi > H - 1) y_hi = H - 1;
        long labs[8];
        for (int y = y_lo; y <= y_hi; ++y) {
            if (((y - y_lo) & 7) == 0) {               // the labels of the next 8 rows: 8 independent loads in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) labs[u] = label[((long)b * H + min(y + u, H - 1)) * W + x];
            }
            int y0, y1;
            float ly;
            bil_index(y, h, sy, y0, y1, ly);
            long lab = labs[0];
#pragma unroll
            for (int u = 1; u < 8; ++u) lab = ((y - y_lo) & 7) == u ? labs[u] : lab;
            if (y0 != ys) continue;                      // wave-uniform
            float zc[NCT];
            float mx = -INFINITY, zl = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                zc[c] = c < nc ? fmaf(ly, dv[c], av[c]) : -INFINITY;
                mx = fmaxf(mx, zc[c]);
                zl = c == lab ? zc[c] : zl;
            }
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                zc[c] = c < nc ? __expf(zc[c] - mx) : 0.f;
                sum += zc[c];
            }
            const bool valid = lab != ignore && lab >= 0 && lab < nc;
            const float nll = mx + __logf(sum) - zl;
            if (valid && lab == 0) { l_bg += nll; c_bg += 1.f; }
            if (valid && lab != 0) { l_fg += nll; c_fg += 1.f; }
            const float wp = valid ? (lab == 0 ? wbg : wfg) : 0.f;
            const float u = wp / sum;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float gval = fmaf(u, zc[c], c == lab ? -wp : 0.f);       // wp * (softmax_c - [c == label])
                accT[c] += gval;
                accB[c] = fmaf(ly, gval, accB[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c)
            if (c < nc) {
                const long o = (((long)b * nc + c) * h) * W + x;
                if (y1s == ys) tmpA[o + (long)ys * W] = accT[c];               // last row: both weights land on it
                else {
                    tmpA[o + (long)ys * W] = accT[c] - accB[c];
                    tmpB[o + (long)y1s * W] = accB[c];
                }
            }
    }
    const float a0 = block_sum(l_bg, red), a1 = block_sum(c_bg, red), a2 = block_sum(l_fg, red), a3 = block_sum(c_fg, red);
    if (threadIdx.x == 0) {
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[blk * 4] = a0; part[blk * 4 + 1] = a1; part[blk * 4 + 2] = a2; part[blk * 4 + 3] = a3;
    }
}

// X pass over tmpA (+ tmpB for rows > 0: row 0 receives nothing from above).  One wave per (image, class, row): the Wd
// values are read coalesced and parked in LDS (index x + x/16: the 32 output lanes then read their windows, 16 apart, from
// different banks); each output lane sums its window in ascending x, the order of the per-thread form (which read global
// memory at a 64-byte lane stride: 51 us for 22 MB).
__global__ __launch_bounds__(256) void seg_bwd_x2_kernel(const float* __restrict__ tmpA, const float* __restrict__ tmpB,
                                                          float* __restrict__ gsrc, long rows, int Hs, int Ws, int Wd, float sx) {
    extern __shared__ float xrow[];                       // 4 waves x (Wd + Wd/16 + 1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;         // (image * nc + class) * Hs + ys
    if (row >= rows) return;                              // wave-uniform; no block-wide barrier below
    const int ys = (int)(row % Hs);
    float* v = xrow + wave * (Wd + (Wd >> 4) + 1);
    const float* TA = tmpA + row * Wd;
    const float* TB = tmpB + row * Wd;
    for (int x = lane; x < Wd; x += 64) v[x + (x >> 4)] = ys > 0 ? TA[x] + TB[x] : TA[x];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes of this wave are done before its lanes read them
    __builtin_amdgcn_wave_barrier();
    const float ix = 1.0f / sx;
    for (int xs = lane; xs < Ws; xs += 64) {
        int x_lo = (int)floorf((xs - 1.5f) * ix) - 1, x_hi = (int)ceilf((xs + 1.5f) * ix) + 1;
        if (x_lo < 0) x_lo = 0;
        if (x_hi > Wd - 1) x_hi = Wd - 1;
        float acc = 0.f;
        for (int x = x_lo; x <= x_hi; ++x) {
            int x0, x1;
            float lx;
            bil_index(x, Ws, sx, x0, x1, lx);
            const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
            acc = fmaf(wx, v[x + (x >> 4)], acc);
        }
        gsrc[row * Ws + xs] = acc;
    }
}

// Loss (sums, as wc_seg_loss_fwd) and d loss / d seg for an upstream gradient of 1 (grad, (B,nc,h,w)) in one pixel pass.
// cnt: 2 * 1024 floats, part: 4 floats per 64 x 4 block of (W, h) per image, tmp: 2 * B*nc*h*W floats.  nc <= 24.
extern "C" int wc_seg_loss_fwd_bwd(const float* seg, const int64_t* label, float* cnt, float* part, float* sums, float* tmp,
                                   float* grad, int B, int nc, int h, int w, int H, int W, int ignore, void* stream) {
    WC_CHECK_ARG(seg && label && cnt && part && sums && tmp && grad && B > 0 && B <= 65535 && nc > 0 && nc <= 24 &&
                 (long)B * nc <= 65535 && h > 0 && w > 0 && H >= h && W >= w, "wc_seg_loss_fwd_bwd: bad argument (nc <= 24)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_count_kernel, dim3(SEG_CNT_BLOCKS), dim3(256), 0, st, (const long*)label, cnt, (long)B * H * W, nc, ignore);
    WC_LAUNCH_CHECK("seg_count_kernel");
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(h, 4), B);
    float* tmpB = tmp + (long)B * nc * h * W;
    hipLaunchKernelGGL(seg_loss_fused_kernel<24>, grid, dim3(256), 0, st, seg, (const long*)label, cnt, part, tmp, tmpB, nc, h, w, H,
                       W, (float)h / H, (float)w / W, ignore);
    WC_LAUNCH_CHECK("seg_loss_fused_kernel");
    hipLaunchKernelGGL(seg_loss_reduce_kernel, dim3(1), dim3(1024), 0, st, part, sums, (long)grid.x * grid.y * grid.z, 0);
    WC_LAUNCH_CHECK("seg_loss_reduce_kernel");
    const long rows = (long)B * nc * h;
    WC_CHECK_ARG(W <= 8192, "wc_seg_loss_fwd_bwd: W too large for the row buffer");
    hipLaunchKernelGGL(seg_bwd_x2_kernel, dim3((unsigned)wc_cdiv(rows, 4)), dim3(256), 4 * (W + (W >> 4) + 1) * sizeof(float), st,
                       tmp, tmpB, grad, rows, h, w, W, (float)w / W);
    WC_LAUNCH_CHECK("seg_bwd_x2_kernel");
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

#define AFF_ROWS 32      // rows of the (hw, hw) matrix per workgroup
template <bool BWD>
__global__ __launch_bounds__(256) void aff_loss_kernel(const float* __restrict__ ap, const long* __restrict__ cam,
                                                        float* __restrict__ part, const float* __restrict__ coef,
                                                        float* __restrict__ dap, int h, int w, int H, int W, int radius,
                                                        int ignore) {
    extern __shared__ int lab[];          // [hw] low-res labels of image b | [hw] (y << 16 | x) of every token
    __shared__ float red[16];
    const int hw = h * w, b = blockIdx.y;
    int* yx = lab + hw;                   // token coordinates once per block: the sweep below has no integer division
    for (int t = threadIdx.x; t < hw; t += 256) {
        lab[t] = aff_lowres_label(cam, b, t, h, w, H, W);
        const int ty = t / w;
        yx[t] = (ty << 16) | (t - ty * w);
    }
    __syncthreads();
    // block = ROWS rows i of the (hw, hw) matrix, threads sweep j (four consecutive j per thread and 16-byte access when
    // hw % 4 == 0; the low-resolution label table above is rebuilt per block, so the rows per block amortise it)
    constexpr int ROWS = AFF_ROWS;
    float ps = 0.f, pc = 0.f, ns = 0.f, nc = 0.f;
    float cp = 0.f, cn = 0.f;
    if (BWD) { cp = coef[0]; cn = coef[1]; }
    const bool vec = (hw & 3) == 0;
    if (!BWD && vec && (blockIdx.x + 1) * ROWS <= hw) {
        // forward, full block: the 16-byte loads of 8 rows are issued before any of them is consumed (the row loop
        // below keeps one load per thread in flight: 32 dependent round trips per thread)
        for (int r0 = 0; r0 < ROWS; r0 += 8) {
            for (int j0 = threadIdx.x * 4; j0 < hw; j0 += 1024) {
                float4 pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    pv[u] = *reinterpret_cast<const float4*>(ap + ((long)b * hw + blockIdx.x * ROWS + r0 + u) * hw + j0);
                int lj[4], yj[4], xj[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { lj[k] = lab[j0 + k]; yj[k] = yx[j0 + k] >> 16; xj[k] = yx[j0 + k] & 0xffff; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = blockIdx.x * ROWS + r0 + u;
                    const int li = lab[i], yi = yx[i] >> 16, xi = yx[i] & 0xffff;
                    const float p4[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int dy = yi - yj[k], dx = xi - xj[k];
                        const bool ok = li != ignore && lj[k] != ignore && dy <= radius && -dy <= radius && dx <= radius && -dx <= radius;
                        const bool pos = ok && li == lj[k], neg = ok && li != lj[k];
                        if (pos) { ps += 1.f - p4[k]; pc += 1.f; }
                        if (neg) { ns += p4[k]; nc += 1.f; }
                    }
                }
            }
        }
    } else
    for (int r = 0; r < ROWS; ++r) {
        const int i = blockIdx.x * ROWS + r;
        if (i >= hw) break;
        const int li = lab[i], yi = i / w, xi = i - yi * w;
        const float* row = ap + ((long)b * hw + i) * hw;
        if (vec) {
            for (int j0 = threadIdx.x * 4; j0 < hw; j0 += 1024) {
                float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!BWD) pv = *reinterpret_cast<const float4*>(row + j0);
                const float p4[4] = {pv.x, pv.y, pv.z, pv.w};
                float o4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + k;
                    const int lj = lab[j], yj = yx[j] >> 16, xj = yx[j] & 0xffff;
                    const int dy = yi - yj, dx = xi - xj;
                    const bool ok = li != ignore && lj != ignore && dy <= radius && -dy <= radius && dx <= radius && -dx <= radius;
                    const bool pos = ok && li == lj, neg = ok && li != lj;
                    o4[k] = pos ? cp : (neg ? cn : 0.f);
                    if (!BWD) {
                        if (pos) { ps += 1.f - p4[k]; pc += 1.f; }
                        if (neg) { ns += p4[k]; nc += 1.f; }
                    }
                }
                if (BWD) *reinterpret_cast<float4*>(dap + ((long)b * hw + i) * hw + j0) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
            continue;
        }
        for (int j = threadIdx.x; j < hw; j += 256) {
            const int lj = lab[j], yj = yx[j] >> 16, xj = yx[j] & 0xffff;
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
    WC_CHECK_ARG((size_t)h * w * 8 <= 64 * 1024 && B <= 65535 && h < 32768 && w < 65536, "wc_aff_loss_fwd: h*w <= 8192");
    dim3 grid(wc_cdiv(h * w, AFF_ROWS), B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aff_loss_kernel<false>, grid, dim3(256), (size_t)h * w * 8, st, attn_pred, (const long*)cam_label, part,
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
    WC_CHECK_ARG((size_t)h * w * 8 <= 64 * 1024 && B <= 65535 && h < 32768 && w < 65536, "wc_aff_loss_bwd: h*w <= 8192");
    hipLaunchKernelGGL(aff_loss_kernel<true>, dim3(wc_cdiv(h * w, AFF_ROWS), B), dim3(256), (size_t)h * w * 8, (hipStream_t)stream,
                       nullptr, (const long*)cam_label, nullptr, coef, dap, h, w, H, W, radius, ignore);
    WC_LAUNCH_CHECK("aff_loss_kernel<bwd>");
    return WC_OK;
}
