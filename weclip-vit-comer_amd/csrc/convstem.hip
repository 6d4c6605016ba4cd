// Conv stem of the ViT-CoMer spatial-prior branch on token rows (NHWC): 3x3 / stride-s / pad-1 convolutions as an
// im2col gather + the MFMA GEMM (csrc/gemm.hip), and GroupNorm + ReLU, forward and backward.
// (No reference code exists for the CoMer inserts; these replace the nn.Conv2d / nn.GroupNorm /
// nn.ReLU modules of WeCLIP_model/comer.py's SpatialPrior, which otherwise run MIOpen's fp32 kernels.)
//
// Activations are rows x[(n*H + y)*W + x][c] (the layout the 1x1 projections and the GEMM want).
//   im2col_kernel      : cols[(n*Ho + oy)*Wo + ox][(ky*3 + kx)*C + c] = x[n, oy*s - 1 + ky, ox*s - 1 + kx, c] (0 outside),
//                        written as fp16 hi (+ lo), zero padded to Kp columns (a multiple of 64: the GEMM's K).
//                        The weight matrix is laid out to match: Wmat[o][(ky*3 + kx)*C + c] = w[o][c][ky][kx].
//   col2im_kernel      : dx[n, iy, ix, c] = sum over the (ky, kx, oy, ox) with oy*s - 1 + ky == iy, ox*s - 1 + kx == ix
//                        of dcols[...]: a gather per input pixel (at most ceil(3/s)^2 terms), deterministic.
//   gn_stats_kernel    : per (n, group) mean and rstd over H*W*(C/G) values, fixed-order two-stage reduction.
//   gn_relu_fwd_kernel : y = relu((x - mean) * rstd * gamma + beta)
//   gn_relu_bwd_*      : GroupNorm backward through the ReLU mask: per-(n, g) sums of g and g*xhat (two-stage,
//                        fixed order), then dx; per-channel dgamma / dbeta partials per (n, row block), summed in order.
#include "common.h"

// One thread = 8 consecutive columns of one row (Kp % 8 == 0): the row is decoded once, the eight values leave as ONE 16-byte
// store per output (round 2 wrote 2 bytes per thread: 1.2 TB/s on the 134 MB of the first layer); with C % 8 == 0 the eight
// columns share a tap and come from two 16-byte loads.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, __half* __restrict__ hi,
                                                      __half* __restrict__ lo, int N, int H, int W, int C, int Ho, int Wo,
                                                      int stride, int Kp, long total) {
    const int K = 9 * C, K8 = Kp >> 3;
    const bool vec = (C & 7) == 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (total >> 3); i += (long)gridDim.x * 256) {
        const int col0 = (int)(i % K8) * 8;
        const long row = i / K8;
        const int ox = (int)(row % Wo);
        const long r2 = row / Wo;
        const int oy = (int)(r2 % Ho);
        const long n = r2 / Ho;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (vec) {
            if (col0 < K) {
                const int tap = col0 / C, c = col0 - tap * C;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const float4* p = reinterpret_cast<const float4*>(x + ((n * H + iy) * W + ix) * C + c);
                    const float4 a4 = p[0], b4 = p[1];
                    v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = col0 + j;
                if (col < K) {
                    const int tap = col / C, c = col - tap * C;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = x[((n * H + iy) * W + ix) * C + c];
                }
            }
        }
        __half h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = __float2half(v[j]);
            l[j] = __float2half(v[j] - __half2float(h[j]));
        }
        *reinterpret_cast<u32x4*>(hi + i * 8) = *reinterpret_cast<const u32x4*>(h);
        if (lo) *reinterpret_cast<u32x4*>(lo + i * 8) = *reinterpret_cast<const u32x4*>(l);
    }
}

// one thread = 4 consecutive channels of one input pixel (C % 4 == 0: 16-byte loads / store), else one channel
template <int V>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int N, int H,
                                                      int W, int C, int Ho, int Wo, int stride, int Kp, long total) {
    const int CV = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total / V; i += (long)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        const long p = i / CV;
        const int ix = (int)(p % W);
        const long p2 = p / W;
        const int iy = (int)(p2 % H);
        const long n = p2 / H;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        // all nine taps are LOADED (a tap that does not exist for this pixel reads the pixel-independent address dcols + c: one
        // cached line) and masked by a multiplication: behind the `continue`s hipcc waited for every load before the next one
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky, oy = ty / stride;
            const bool oky = ty >= 0 && ty % stride == 0 && oy < Ho;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx, ox = tx / stride;
                const bool ok = oky && tx >= 0 && tx % stride == 0 && ox < Wo;
                const float* src = dcols + (ok ? ((n * Ho + oy) * Wo + ox) * Kp + (ky * 3 + kx) * C : 0) + c;
                const float msk = ok ? 1.f : 0.f;
                if constexpr (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    acc[0] += t.x * msk; acc[1] += t.y * msk; acc[2] += t.z * msk; acc[3] += t.w * msk;
                } else {
                    acc[0] += src[0] * msk;
                }
            }
        }
        if constexpr (V == 4) *reinterpret_cast<float4*>(dx + i * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else dx[i] = acc[0];
    }
}

// ---- GroupNorm + ReLU on rows (n, HW, C), G groups of Cg = C / G channels --------------------------------------
// Round 3: a thread owns ONE channel and a row lane (consecutive lanes = consecutive channels: coalesced NHWC rows, no
// per-element division), the row lanes meet in LDS in a fixed order, group sums are formed from the channel sums.  The
// round-2 kernels walked the block once per group with two block reductions each and the backward a second time with one
// thread per channel (126 us per launch at 16 x 256 x 256 x 32).  Needs 256 % C == 0 (C = 32, 64, 128 in the stem).
#define GN_ROWS 256      // rows of one partial block (the callers size their buffers for 64: more than enough)

// part[(n*nblk + blk)*G + g] = (sum, sumsq) of rows [blk*GN_ROWS, ...) of image n, group g       (grid: nblk, N)
// Round 4: a thread owns FOUR consecutive channels (16-byte loads, four rows in flight per thread) -- one 4-byte load per
// thread and iteration left the kernels at a third of the HBM rate (27 / 80 us per launch forward / backward partials).
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, float2* __restrict__ part, int HW,
                                                          int C, int G) {
    __shared__ float2 red[4][256];
    const int blk = blockIdx.x, n = blockIdx.y, nblk = gridDim.x;
    const int Cq = C >> 2, Cg = C / G, cq = threadIdx.x % Cq, rl = threadIdx.x / Cq, nrl = 256 / Cq;
    const int r0 = blk * GN_ROWS, r1 = min(r0 + GN_ROWS, HW);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xp = x + (long)n * HW * C + 4 * cq;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += nrl) {
        const float4 v = *reinterpret_cast<const float4*>(xp + (long)r * C);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = make_float2(s[k], q[k]);
    __syncthreads();
    float2 t = make_float2(0.f, 0.f);
    if (threadIdx.x < C) {                    // channel sums: the row lanes in a fixed order
        const int k = threadIdx.x & 3, cq2 = threadIdx.x >> 2;
        t = red[k][cq2];
        for (int j = 1; j < nrl; ++j) { t.x += red[k][j * Cq + cq2].x; t.y += red[k][j * Cq + cq2].y; }
    }
    __syncthreads();
    if (threadIdx.x < C) red[0][threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < G) {
        float2 u = make_float2(0.f, 0.f);
        for (int k = 0; k < Cg; ++k) { u.x += red[0][threadIdx.x * Cg + k].x; u.y += red[0][threadIdx.x * Cg + k].y; }
        part[((long)n * nblk + blk) * G + threadIdx.x] = u;
    }
}

// stats[n*G + g] = (mean, rstd): fixed-order sum of the partials, one 64-thread workgroup per (n, g)      (grid: N*G)
__global__ __launch_bounds__(64) void gn_finish_kernel(const float2* __restrict__ part, float2* __restrict__ stats, int nblk, int G,
                                                        int N, float count, float eps) {
    __shared__ double sh[2][64];
    const int i = blockIdx.x, n = i / G, g = i - n * G;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        const float2 p = part[((long)n * nblk + b) * G + g];
        s += p.x;
        q += p.y;
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0; q = 0.0;
        for (int k = 0; k < 64; ++k) { s += sh[0][k]; q += sh[1][k]; }
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

// four consecutive channels per thread (C % 4 == 0 and Cg % 4 == 0 in the stem: 32 / 8 = 4 channels per group at least)
__global__ __launch_bounds__(256) void gn_relu_fwd_kernel(const float* __restrict__ x, const float2* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, int HW, int C, int G, long total) {
    const int Cg = C / G;
    if ((C & 3) == 0 && (Cg & 3) == 0) {
        const long per = (long)HW * C;
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024) {
            const int c = (int)(i % C);
            const long n = i / per;
            const float2 st = stats[n * G + c / Cg];
            const float4 xv = *reinterpret_cast<const float4*>(x + i), gm = *reinterpret_cast<const float4*>(gamma + c),
                         bt = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = fmaxf((xv.x - st.x) * st.y * gm.x + bt.x, 0.f);
            o.y = fmaxf((xv.y - st.x) * st.y * gm.y + bt.y, 0.f);
            o.z = fmaxf((xv.z - st.x) * st.y * gm.z + bt.z, 0.f);
            o.w = fmaxf((xv.w - st.x) * st.y * gm.w + bt.w, 0.f);
            *reinterpret_cast<float4*>(y + i) = o;
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long n = i / ((long)HW * C);
        const float2 st = stats[n * G + c / Cg];
        const float v = (x[i] - st.x) * st.y * gamma[c] + beta[c];
        y[i] = v > 0.f ? v : 0.f;
    }
}

// backward partials: per (n, blk): per channel (sum d*xhat, sum d) with d = dy*mask, and from them the group sums
// (sum g, sum g*xhat) with g = d*gamma: sum_rows g = gamma_c * sum d, sum_rows g*xhat = gamma_c * sum d*xhat     (grid: nblk, N)
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy, const float2* __restrict__ stats,
                                                              const float* __restrict__ gamma, float2* __restrict__ gpart,
                                                              float2* __restrict__ cpart, int HW, int C, int G) {
    __shared__ float2 red[4][256];
    const int blk = blockIdx.x, n = blockIdx.y, nblk = gridDim.x;
    const int Cq = C >> 2, Cg = C / G, cq = threadIdx.x % Cq, rl = threadIdx.x / Cq, nrl = 256 / Cq;
    const int r0 = blk * GN_ROWS, r1 = min(r0 + GN_ROWS, HW);
    const float2 st = stats[n * G + (4 * cq) / Cg];      // (Cg % 4 == 0: the four channels share a group)
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    const long base = (long)n * HW * C + 4 * cq;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += nrl) {
        const long o = base + (long)r * C;
        const float4 yv = *reinterpret_cast<const float4*>(y + o), dv = *reinterpret_cast<const float4*>(dy + o),
                     xv = *reinterpret_cast<const float4*>(x + o);
        const float d0 = yv.x > 0.f ? dv.x : 0.f, d1 = yv.y > 0.f ? dv.y : 0.f, d2 = yv.z > 0.f ? dv.z : 0.f, d3 = yv.w > 0.f ? dv.w : 0.f;
        a[0] += d0 * (xv.x - st.x) * st.y; a[1] += d1 * (xv.y - st.x) * st.y; a[2] += d2 * (xv.z - st.x) * st.y; a[3] += d3 * (xv.w - st.x) * st.y;
        b[0] += d0; b[1] += d1; b[2] += d2; b[3] += d3;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = make_float2(a[k], b[k]);
    __syncthreads();
    float2 t = make_float2(0.f, 0.f);
    if (threadIdx.x < C) {
        const int k = threadIdx.x & 3, cq2 = threadIdx.x >> 2;
        t = red[k][cq2];
        for (int j = 1; j < nrl; ++j) { t.x += red[k][j * Cq + cq2].x; t.y += red[k][j * Cq + cq2].y; }
        cpart[((long)n * nblk + blk) * C + threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const float gm = gamma[threadIdx.x];
        red[0][threadIdx.x] = make_float2(gm * t.y, gm * t.x);          // (sum g, sum g*xhat) of this channel
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float2 u = make_float2(0.f, 0.f);
        for (int k = 0; k < Cg; ++k) { u.x += red[0][threadIdx.x * Cg + k].x; u.y += red[0][threadIdx.x * Cg + k].y; }
        gpart[((long)n * nblk + blk) * G + threadIdx.x] = u;
    }
}

// gsum[n*G + g] = sum of the group partials; dgamma/dbeta[c] = sum over (n, blk) of the channel partials.
// One workgroup per output; thread t sums partials t, t+256, ... and block_sum combines them: a fixed order.
// grid: N*G + C workgroups
__global__ __launch_bounds__(256) void gn_bwd_finish_kernel(const float2* __restrict__ gpart, const float2* __restrict__ cpart,
                                                             float2* __restrict__ gsum, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int nblk, int G, int N, int C) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    float a = 0.f, b = 0.f;
    if (i < N * G) {
        const int n = i / G, g = i - n * G;
        for (int k = threadIdx.x; k < nblk; k += 256) {
            const float2 p = gpart[((long)n * nblk + k) * G + g];
            a += p.x;
            b += p.y;
        }
    } else {
        const int c = i - N * G;
        for (long k = threadIdx.x; k < (long)N * nblk; k += 256) {
            const float2 p = cpart[k * C + c];
            a += p.x;
            b += p.y;
        }
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        if (i < N * G) gsum[i] = make_float2(a, b);
        else { dgamma[i - N * G] = a; dbeta[i - N * G] = b; }
    }
}

// dx = rstd * (g - (sum_g + xhat * sum_gx) / count)
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dy, const float2* __restrict__ stats,
                                                         const float2* __restrict__ gsum, const float* __restrict__ gamma,
                                                         float* __restrict__ dx, int HW, int C, int G, float count,
                                                         long total) {
    const int Cg = C / G;
    if ((C & 3) == 0 && (Cg & 3) == 0) {
        const long per = (long)HW * C;
        const float ic = 1.f / count;
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024) {
            const int c = (int)(i % C);
            const long n = i / per;
            const int g = c / Cg;
            const float2 st = stats[n * G + g], gs = gsum[n * G + g];
            const float4 xv = *reinterpret_cast<const float4*>(x + i), yv = *reinterpret_cast<const float4*>(y + i),
                         dv = *reinterpret_cast<const float4*>(dy + i), gm = *reinterpret_cast<const float4*>(gamma + c);
            float4 o;
            o.x = st.y * ((yv.x > 0.f ? dv.x : 0.f) * gm.x - (gs.x + (xv.x - st.x) * st.y * gs.y) / count);
            o.y = st.y * ((yv.y > 0.f ? dv.y : 0.f) * gm.y - (gs.x + (xv.y - st.x) * st.y * gs.y) / count);
            o.z = st.y * ((yv.z > 0.f ? dv.z : 0.f) * gm.z - (gs.x + (xv.z - st.x) * st.y * gs.y) / count);
            o.w = st.y * ((yv.w > 0.f ? dv.w : 0.f) * gm.w - (gs.x + (xv.w - st.x) * st.y * gs.y) / count);
            (void)ic;
            *reinterpret_cast<float4*>(dx + i) = o;
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long n = i / ((long)HW * C);
        const int g = c / Cg;
        const float2 st = stats[n * G + g], gs = gsum[n * G + g];
        const float xh = (x[i] - st.x) * st.y;
        const float gg = (y[i] > 0.f ? dy[i] : 0.f) * gamma[c];
        dx[i] = st.y * (gg - (gs.x + xh * gs.y) / count);
    }
}

static int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

extern "C" int wc_im2col3x3(const float* x, void* cols_hi, void* cols_lo, int N, int H, int W, int C, int stride, int Kp,
                            void* stream) {
    WC_CHECK_ARG(x && cols_hi && N > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && Kp >= 9 * C && Kp % 64 == 0,
                 "wc_im2col3x3: bad argument (stride 1 or 2, Kp >= 9 C and a multiple of 64)");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)N * Ho * Wo * Kp;
    WC_CHECK_ARG(((uintptr_t)cols_hi | (uintptr_t)cols_lo | (uintptr_t)x) % 16 == 0, "wc_im2col3x3: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total >> 3)), dim3(256), 0, (hipStream_t)stream, x, (__half*)cols_hi,
                       (__half*)cols_lo, N, H, W, C, Ho, Wo, stride, Kp, total);
    WC_LAUNCH_CHECK("im2col_kernel");
    return WC_OK;
}

extern "C" int wc_col2im3x3(const float* dcols, float* dx, int N, int H, int W, int C, int stride, int Kp, void* stream) {
    WC_CHECK_ARG(dcols && dx && N > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && Kp >= 9 * C,
                 "wc_col2im3x3: bad argument");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)N * H * W * C;
    if (C % 4 == 0 && Kp % 4 == 0 && ((uintptr_t)dcols | (uintptr_t)dx) % 16 == 0)
        hipLaunchKernelGGL(col2im_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, dcols, dx, N, H, W, C, Ho, Wo,
                           stride, Kp, total);
    else
        hipLaunchKernelGGL(col2im_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dcols, dx, N, H, W, C, Ho, Wo,
                           stride, Kp, total);
    WC_LAUNCH_CHECK("col2im_kernel");
    return WC_OK;
}

extern "C" int wc_groupnorm_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                     float* part, int N, int HW, int C, int G, float eps, void* stream) {
    WC_CHECK_ARG(x && gamma && beta && y && stats && part && N > 0 && N <= 65535 && HW > 0 && C > 0 && G > 0 && C % G == 0 &&
                 C <= 256 && 256 % C == 0 && (C / G) % 4 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0,
                 "wc_groupnorm_relu_fwd: bad argument (C must divide 256, 4 | C / G, 16-byte aligned rows)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = wc_cdiv(HW, GN_ROWS);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, N), dim3(256), 0, st, x, (float2*)part, HW, C, G);
    WC_LAUNCH_CHECK("gn_partial_kernel");
    hipLaunchKernelGGL(gn_finish_kernel, dim3(N * G), dim3(64), 0, st, (const float2*)part, (float2*)stats, nblk, G, N,
                       (float)((double)HW * (C / G)), eps);
    WC_LAUNCH_CHECK("gn_finish_kernel");
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, (const float2*)stats, gamma, beta, y, HW, C,
                       G, total);
    WC_LAUNCH_CHECK("gn_relu_fwd_kernel");
    return WC_OK;
}

extern "C" int wc_groupnorm_relu_bwd(const float* x, const float* y, const float* dy, const float* stats, const float* gamma,
                                     float* dx, float* dgamma, float* dbeta, float* gpart, float* cpart, float* gsum, int N,
                                     int HW, int C, int G, void* stream) {
    WC_CHECK_ARG(x && y && dy && stats && gamma && dx && dgamma && dbeta && gpart && cpart && gsum && N > 0 && N <= 65535 &&
                 HW > 0 && C > 0 && G > 0 && C % G == 0 && C <= 256 && 256 % C == 0 && (C / G) % 4 == 0 &&
                 ((uintptr_t)x | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0,
                 "wc_groupnorm_relu_bwd: bad argument (C must divide 256, 4 | C / G, 16-byte aligned rows)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = wc_cdiv(HW, GN_ROWS);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nblk, N), dim3(256), 0, st, x, y, dy, (const float2*)stats, gamma,
                       (float2*)gpart, (float2*)cpart, HW, C, G);
    WC_LAUNCH_CHECK("gn_bwd_partial_kernel");
    hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(N * G + C), dim3(256), 0, st, (const float2*)gpart, (const float2*)cpart,
                       (float2*)gsum, dgamma, dbeta, nblk, G, N, C);
    WC_LAUNCH_CHECK("gn_bwd_finish_kernel");
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(gn_bwd_dx_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, y, dy, (const float2*)stats,
                       (const float2*)gsum, gamma, dx, HW, C, G, (float)((double)HW * (C / G)), total);
    WC_LAUNCH_CHECK("gn_bwd_dx_kernel");
    return WC_OK;
}
