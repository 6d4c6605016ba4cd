// Depth-wise 2-D convolution (stride 1, "same" zero padding, odd kernel) forward and backward for gfx950:
// the multi-receptive-field perception block of the ViT-CoMer inserts (no reference code exists,
// this is nn.Conv2d(C, C, k, padding=k/2, groups=C) as the paper's MRFP uses it).  The feature maps are small
// ((B, 64, 64, 64) at most for 512x512 inputs) but MIOpen's grouped backward-weight path takes 3.7 ms per call on
// them; these are plain one-pass kernels.
//   fwd : y[n,c,y,x]  = b[c] + sum_{ky,kx} w[c,ky,kx] * x[n,c,y+ky-p,x+kx-p]
//   bwdx: dx[n,c,y,x] =        sum_{ky,kx} w[c,ky,kx] * dy[n,c,y-ky+p,x-kx+p]
//   bwdw: dw[c,ky,kx] = sum_{n,y,x} dy[n,c,y,x] * x[n,c,y+ky-p,x+kx-p];  db[c] = sum dy[n,c,:,:]
#include "common.h"

#define DW_MAXK 7

template <bool BWD>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int C, int H,
                                                      int W, int k) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int nc = blockIdx.z, c = nc % C;
    if (x >= W || y >= H) return;
    const int p = k >> 1;
    const float* I = in + (long)nc * H * W;
    const float* wc = w + (long)c * k * k;
    float acc = (!BWD && bias) ? bias[c] : 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int yy = BWD ? y - ky + p : y + ky - p;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int xx = BWD ? x - kx + p : x + kx - p;
            if (xx >= 0 && xx < W) acc = fmaf(wc[ky * k + kx], I[(long)yy * W + xx], acc);
        }
    }
    out[(long)nc * H * W + (long)y * W + x] = acc;
}

// one block per (channel, image): every thread accumulates all k*k taps (+ bias) over its share of (y, x);
// part[n][c][k*k + 1] is summed over the images by dwconv_bwdw_final_kernel
__global__ __launch_bounds__(256) void dwconv_bwdw_kernel(const float* __restrict__ in, const float* __restrict__ dy,
                                                           float* __restrict__ part, int C, int H, int W, int k) {
    __shared__ float red[16];
    const int c = blockIdx.x, n = blockIdx.y, p = k >> 1;
    float acc[DW_MAXK * DW_MAXK + 1];
#pragma unroll
    for (int t = 0; t < DW_MAXK * DW_MAXK + 1; ++t) acc[t] = 0.f;
    const int HW = H * W;
    const float* G = dy + ((long)n * C + c) * HW;
    const float* I = in + ((long)n * C + c) * HW;
    for (int r = threadIdx.x; r < HW; r += 256) {
        const int y = r / W, x = r - y * W;
        const float g = G[r];
        acc[DW_MAXK * DW_MAXK] += g;
#pragma unroll
        for (int ky = 0; ky < DW_MAXK; ++ky)
#pragma unroll
            for (int kx = 0; kx < DW_MAXK; ++kx) {
                if (ky < k && kx < k) {
                    const int yy = y + ky - p, xx = x + kx - p;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[ky * DW_MAXK + kx] = fmaf(g, I[yy * W + xx], acc[ky * DW_MAXK + kx]);
                }
            }
    }
    float* out = part + ((long)n * C + c) * (k * k + 1);
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const float s = block_sum(acc[ky * DW_MAXK + kx], red);
            if (threadIdx.x == 0) out[ky * k + kx] = s;
            __syncthreads();
        }
    const float sb = block_sum(acc[DW_MAXK * DW_MAXK], red);
    if (threadIdx.x == 0) out[k * k] = sb;
}

__global__ __launch_bounds__(256) void dwconv_bwdw_final_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 float* __restrict__ db, int N, int C, int kk) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // over C * (kk + 1)
    if (i >= C * (kk + 1)) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += part[(long)n * C * (kk + 1) + i];
    const int c = i / (kk + 1), t = i - c * (kk + 1);
    if (t < kk) dw[(long)c * kk + t] = s;
    else if (db) db[c] = s;
}

static int dw_check(const char* who, const void* a, const void* b, const void* c, int N, int C, int H, int W, int k) {
    WC_CHECK_ARG(a && b && c && N > 0 && C > 0 && H > 0 && W > 0 && (k & 1) && k >= 1 && k <= DW_MAXK && (long)N * C <= 65535,
                 "%s: bad argument (odd k <= 7, N*C <= 65535)", who);
    return WC_OK;
}

extern "C" int wc_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int k,
                             void* stream) {
    if (dw_check("wc_dwconv_fwd", x, w, y, N, C, H, W, k)) return WC_ERR_ARG;
    hipLaunchKernelGGL(dwconv_kernel<false>, dim3(wc_cdiv(W, 64), wc_cdiv(H, 4), N * C), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, y, C, H, W, k);
    WC_LAUNCH_CHECK("dwconv_kernel<fwd>");
    return WC_OK;
}

// part: workspace N * C * (k*k + 1) floats.
extern "C" int wc_dwconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, float* part,
                             int N, int C, int H, int W, int k, void* stream) {
    if (dw_check("wc_dwconv_bwd", x, w, dy, N, C, H, W, k)) return WC_ERR_ARG;
    WC_CHECK_ARG(dx && dw && part && N <= 65535, "wc_dwconv_bwd: dx, dw and part are required");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dwconv_kernel<true>, dim3(wc_cdiv(W, 64), wc_cdiv(H, 4), N * C), dim3(256), 0, st, dy, w,
                       (const float*)nullptr, dx, C, H, W, k);
    WC_LAUNCH_CHECK("dwconv_kernel<bwd>");
    hipLaunchKernelGGL(dwconv_bwdw_kernel, dim3(C, N), dim3(256), 0, st, x, dy, part, C, H, W, k);
    WC_LAUNCH_CHECK("dwconv_bwdw_kernel");
    hipLaunchKernelGGL(dwconv_bwdw_final_kernel, dim3(wc_cdiv((long)C * (k * k + 1), 256)), dim3(256), 0, st, part, dw, db, N, C,
                       k * k);
    WC_LAUNCH_CHECK("dwconv_bwdw_final_kernel");
    return WC_OK;
}
