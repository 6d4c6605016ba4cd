// Fused glue kernels of the ViT-CoMer insert engine (comer_engine.py).  There is NO CoMer code in the
// reference repository (only ViT_CoMer.pdf §3.2-3.3 and the brief): these kernels implement this package's own
// CoMerInteraction (WeCLIP_model/comer.py) on token rows (N, S, C) without the NCHW transposes, concatenations and
// element-wise launches of the module-by-module form; parity is pinned against that module only ("parity unpinned" w.r.t.
// the reference).
//   mrfp_dwconv_*   MRFP's multi-receptive-field depth-wise convolutions (3x3 on channels [0, C/2), 5x5 on [C/2, C)) of all
//                   pyramid levels in ONE launch on NHWC rows, + bias, + exact GELU for the following FC; backward w.r.t. the
//                   input and (fixed-order two-stage reduction) the filters / biases
//   msda_prep_*     sampling_offsets | attention_weights rows of one fused Linear -> sampling locations (reference point +
//                   offset / level size) and soft-maxed attention weights, and the backward of both
//   rows_copy / rows_add: the concatenation of the 8 CTI outputs in front of the fusion conv and its backward
#include "common.h"

#define CM_MAX_LEVELS 8
struct CmLevels {
    int n;
    int H[CM_MAX_LEVELS], W[CM_MAX_LEVELS], start[CM_MAX_LEVELS];
};

__device__ __forceinline__ float cm_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// x (N, S, C) f32 rows; w3 (C/2, 9), b3 (C/2), w5 (C/2, 25), b5 (C/2).  y = conv + bias (f32, optional), g16 = gelu(y) (f16).
// thread = (pixel, channel); a workgroup = 256 / C pixels x C channels (C = 64, 128 or 256).
__global__ __launch_bounds__(256) void mrfp_dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                               const float* __restrict__ b3, const float* __restrict__ w5,
                                                               const float* __restrict__ b5, float* __restrict__ y,
                                                               __half* __restrict__ g16, CmLevels lv, int S, int C) {
    const int c = threadIdx.x % C, s = blockIdx.x * (256 / C) + threadIdx.x / C, n = blockIdx.y;
    if (s >= S) return;
    int l = 0;
    while (l + 1 < lv.n && s >= lv.start[l + 1]) ++l;
    const int H = lv.H[l], W = lv.W[l], p = s - lv.start[l], py = p / W, px = p - py * W;
    const int half = C >> 1, big = c >= half, k = big ? 5 : 3, r = k >> 1;
    const float* wc = big ? w5 + (long)(c - half) * 25 : w3 + (long)c * 9;
    float acc = big ? b5[c - half] : b3[c];
    const float* xb = x + ((long)n * S + lv.start[l]) * C + c;
    for (int dy = -r; dy <= r; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            acc = fmaf(wc[(dy + r) * k + dx + r], xb[((long)yy * W + xx) * C], acc);
        }
    }
    const long o = ((long)n * S + s) * C + c;
    if (y) y[o] = acc;
    if (g16) g16[o] = __float2half(cm_gelu(acc));
}

// dx (N, S, C) = conv^T(dy): dx[p] = sum_taps w[tap] * dy[p - tap]; written as f32 and / or f16
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w3,
                                                                    const float* __restrict__ w5, float* __restrict__ dx32,
                                                                    __half* __restrict__ dx16, CmLevels lv, int S, int C) {
    const int c = threadIdx.x % C, s = blockIdx.x * (256 / C) + threadIdx.x / C, n = blockIdx.y;
    if (s >= S) return;
    int l = 0;
    while (l + 1 < lv.n && s >= lv.start[l + 1]) ++l;
    const int H = lv.H[l], W = lv.W[l], p = s - lv.start[l], py = p / W, px = p - py * W;
    const int half = C >> 1, big = c >= half, k = big ? 5 : 3, r = k >> 1;
    const float* wc = big ? w5 + (long)(c - half) * 25 : w3 + (long)c * 9;
    float acc = 0.f;
    const float* db = dy + ((long)n * S + lv.start[l]) * C + c;
    for (int dyy = -r; dyy <= r; ++dyy) {
        const int yy = py - dyy;                      // output pixel that read this input pixel through tap (dyy, dxx)
        if (yy < 0 || yy >= H) continue;
        for (int dxx = -r; dxx <= r; ++dxx) {
            const int xx = px - dxx;
            if (xx < 0 || xx >= W) continue;
            acc = fmaf(wc[(dyy + r) * k + dxx + r], db[((long)yy * W + xx) * C], acc);
        }
    }
    const long o = ((long)n * S + s) * C + c;
    if (dx32) dx32[o] = acc;
    if (dx16) dx16[o] = __float2half(acc);
}

// filter / bias gradients, stage 1: workgroup (chunk of CM_WCH pixels, image) -> part[(n * nchunk + chunk)][c][26]
// (taps 0..24 [3x3 filters use 0..8] and the bias sum at 25).  256 threads = 256 / C pixel lanes x C channels; the pixel
// lanes are combined through LDS in a fixed order.
#define CM_WCH 128
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ part, CmLevels lv, int S, int C) {
    __shared__ float red[256 * 26];
    const int c = threadIdx.x % C, pl = threadIdx.x / C, npl = 256 / C, n = blockIdx.y;
    const int half = C >> 1, big = c >= half, k = big ? 5 : 3, r = k >> 1;
    float acc[26];
#pragma unroll
    for (int t = 0; t < 26; ++t) acc[t] = 0.f;
    const int s0 = blockIdx.x * CM_WCH;
    for (int s = s0 + pl; s < s0 + CM_WCH && s < S; s += npl) {
        int l = 0;
        while (l + 1 < lv.n && s >= lv.start[l + 1]) ++l;
        const int H = lv.H[l], W = lv.W[l], p = s - lv.start[l], py = p / W, px = p - py * W;
        const float g = dy[((long)n * S + s) * C + c];
        const float* xb = x + ((long)n * S + lv.start[l]) * C + c;
        acc[25] += g;
        if (big) {
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                const int yy = py + t / 5 - 2, xx = px + t % 5 - 2;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[t] = fmaf(g, xb[((long)yy * W + xx) * C], acc[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[t] = fmaf(g, xb[((long)yy * W + xx) * C], acc[t]);
            }
        }
    }
    (void)k; (void)r;
#pragma unroll
    for (int t = 0; t < 26; ++t) red[t * 256 + threadIdx.x] = acc[t];
    __syncthreads();
    if (pl == 0) {
        float* out = part + (((long)n * gridDim.x + blockIdx.x) * C + c) * 26;
#pragma unroll
        for (int t = 0; t < 26; ++t) {
            float sum = 0.f;
            for (int q = 0; q < npl; ++q) sum += red[t * 256 + q * C + c];
            out[t] = sum;
        }
    }
}

// stage 2: out[c][t] = alpha * sum over the nparts partial rows (fixed order); dw3 (C/2, 9), db3, dw5 (C/2, 25), db5
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_w_final_kernel(const float* __restrict__ part, float* __restrict__ dw3,
                                                                       float* __restrict__ db3, float* __restrict__ dw5,
                                                                       float* __restrict__ db5, int nparts, int C, float alpha) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // (c, t)
    if (i >= C * 26) return;
    const int c = i / 26, t = i - c * 26, half = C >> 1;
    if (c < half && t >= 9 && t < 25) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = 0;
    for (; q + 4 <= nparts; q += 4) {
        s0 += part[((long)q * C + c) * 26 + t];
        s1 += part[((long)(q + 1) * C + c) * 26 + t];
        s2 += part[((long)(q + 2) * C + c) * 26 + t];
        s3 += part[((long)(q + 3) * C + c) * 26 + t];
    }
    for (; q < nparts; ++q) s0 += part[((long)q * C + c) * 26 + t];
    const float v = ((s0 + s1) + (s2 + s3)) * alpha;
    if (c < half) {
        if (t == 25) db3[c] = v; else dw3[c * 9 + t] = v;
    } else {
        if (t == 25) db5[c - half] = v; else dw5[(c - half) * 25 + t] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ow (N*Lq, ld): per query row [offsets: M*nL*P*2 | logits: M*nL*P] (the outputs of MSDeformAttn's sampling_offsets and
// attention_weights Linears, computed by ONE GEMM on the concatenated weights).  ref (Lq, nL_ref, 2) reference points in
// [0,1] (x, y) (nL_ref = 1: the same point for every level).  -> loc (N,Lq,M,nL,P,2), attn (N,Lq,M,nL,P) = softmax over nL*P.
// thread = (query, head).
__global__ __launch_bounds__(256) void msda_prep_fwd_kernel(const float* __restrict__ ow, const float* __restrict__ boff,
                                                             const float* __restrict__ baw, const float* __restrict__ ref,
                                                             float* __restrict__ loc, float* __restrict__ attn, CmLevels lv,
                                                             long NQ, int Lq, int M, int P, int ld, int nl_ref) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // nq * M + m
    if (i >= NQ * M) return;
    const long nq = i / M;
    const int m = (int)(i - nq * M), q = (int)(nq % Lq);
    const int T = lv.n * P;
    const float* row = ow + nq * ld;
    const float* off = row + (long)m * T * 2;
    const float* lg = row + (long)M * T * 2 + (long)m * T;
    const float* bo = boff ? boff + (long)m * T * 2 : nullptr;       // the two Linears' biases (the fused GEMM carries none)
    const float* ba = baw ? baw + (long)m * T : nullptr;
    float mx = -3.4e38f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, lg[t] + (ba ? ba[t] : 0.f));
    float sum = 0.f;
    for (int t = 0; t < T; ++t) sum += __expf(lg[t] + (ba ? ba[t] : 0.f) - mx);
    const float inv = 1.f / sum;
    for (int l = 0; l < lv.n; ++l) {
        const float* rp = ref + ((long)q * nl_ref + (nl_ref > 1 ? l : 0)) * 2;
        const float rx = rp[0], ry = rp[1], iw = 1.f / lv.W[l], ih = 1.f / lv.H[l];
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            loc[(i * T + t) * 2] = rx + (off[t * 2] + (bo ? bo[t * 2] : 0.f)) * iw;
            loc[(i * T + t) * 2 + 1] = ry + (off[t * 2 + 1] + (bo ? bo[t * 2 + 1] : 0.f)) * ih;
            attn[i * T + t] = __expf(lg[t] + (ba ? ba[t] : 0.f) - mx) * inv;
        }
    }
}

// dow (N*Lq, ld) f32 [+ f16 copy]: d_off = gloc / level size, d_logit = attn * (gattn - sum_t attn_t gattn_t); columns
// beyond 3*M*nL*P are zeroed (they are the K padding of the input-gradient GEMM).  One workgroup row per query: thread = column.
__global__ __launch_bounds__(256) void msda_prep_bwd_kernel(const float* __restrict__ gloc, const float* __restrict__ gattn,
                                                             const float* __restrict__ attn, float* __restrict__ dow32,
                                                             __half* __restrict__ dow16, CmLevels lv, long NQ, int M, int P, int ld) {
    const int T = lv.n * P, ncol = 3 * M * T;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < NQ * ld; idx += (long)gridDim.x * 256) {
        const long nq = idx / ld;
        const int col = (int)(idx - nq * ld);
        float v = 0.f;
        if (col < 2 * M * T) {
            const int m = col / (2 * T), rem = col - m * 2 * T, t = rem >> 1, xy = rem & 1, l = t / P;
            v = gloc[((nq * M + m) * T + t) * 2 + xy] / (float)(xy ? lv.H[l] : lv.W[l]);
        } else if (col < ncol) {
            const int cc = col - 2 * M * T, m = cc / T, t = cc - m * T;
            const float* a = attn + (nq * M + m) * T;
            const float* g = gattn + (nq * M + m) * T;
            float dot = 0.f;
            for (int j = 0; j < T; ++j) dot = fmaf(a[j], g[j], dot);
            v = a[t] * (g[t] - dot);
        }
        if (dow32) dow32[idx] = v;
        if (dow16) dow16[idx] = __float2half(v);
    }
}

// dst[b][r][0:C] (f16, row stride ldd, batch stride sd) = (f16) src[b][r][0:C] (row stride lds, batch stride ss); src f32 or f16
template <typename T>
__global__ __launch_bounds__(256) void rows_copy_f16_kernel(const T* __restrict__ src, __half* __restrict__ dst, int R, int C,
                                                             long lds_, long ss, long ldd, long sd) {
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)R * C; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        dst[b * sd + r * ldd + c] = __float2half((float)src[b * ss + r * lds_ + c]);
    }
}

// dst[b][r][0:C] (f32, dense rows of C, batch stride sd) += alpha * src[b][r][0:C] (f32, row stride lds, batch stride ss)
__global__ __launch_bounds__(256) void rows_add_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C,
                                                            long lds_, long ss, long sd, float alpha) {
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)R * C; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        dst[b * sd + r * C + c] += alpha * src[b * ss + r * lds_ + c];
    }
}

static int cm_fill_levels(CmLevels* lv, const int* h_shapes, int n_levels, int* S) {
    if (n_levels < 1 || n_levels > CM_MAX_LEVELS) return 1;
    lv->n = n_levels;
    int s = 0;
    for (int l = 0; l < n_levels; ++l) {
        lv->H[l] = h_shapes[2 * l];
        lv->W[l] = h_shapes[2 * l + 1];
        if (lv->H[l] <= 0 || lv->W[l] <= 0) return 1;
        lv->start[l] = s;
        s += lv->H[l] * lv->W[l];
    }
    *S = s;
    return 0;
}

extern "C" int wc_mrfp_dwconv_fwd(const float* x, const float* w3, const float* b3, const float* w5, const float* b5, float* y,
                                  void* g16, const int* h_shapes, int n_levels, int N, int C, void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(x && w3 && b3 && w5 && b5 && (y || g16) && N > 0 && N <= 65535 && (C == 64 || C == 128 || C == 256),
                 "wc_mrfp_dwconv_fwd: bad argument (C = 64, 128 or 256)");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0, "wc_mrfp_dwconv_fwd: 1..8 levels with positive sizes");
    hipLaunchKernelGGL(mrfp_dwconv_fwd_kernel, dim3(wc_cdiv(S, 256 / C), N), dim3(256), 0, (hipStream_t)stream, x, w3, b3, w5, b5, y,
                       (__half*)g16, lv, S, C);
    WC_LAUNCH_CHECK("mrfp_dwconv_fwd_kernel");
    return WC_OK;
}

/* dx32 / dx16 (either may be null): gradient w.r.t. x; dw3 / db3 / dw5 / db5 = alpha * filter / bias gradients;
 * part: workspace of N * ceil(S / 128) * C * 26 floats. */
extern "C" int wc_mrfp_dwconv_bwd(const float* dy, const float* x, const float* w3, const float* w5, float* dx32, void* dx16,
                                  float* dw3, float* db3, float* dw5, float* db5, float* part, float alpha, const int* h_shapes,
                                  int n_levels, int N, int C, void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(dy && x && w3 && w5 && (dx32 || dx16) && dw3 && db3 && dw5 && db5 && part && N > 0 && N <= 65535 &&
                 (C == 64 || C == 128 || C == 256), "wc_mrfp_dwconv_bwd: bad argument");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0, "wc_mrfp_dwconv_bwd: 1..8 levels with positive sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mrfp_dwconv_bwd_data_kernel, dim3(wc_cdiv(S, 256 / C), N), dim3(256), 0, st, dy, w3, w5, dx32, (__half*)dx16,
                       lv, S, C);
    WC_LAUNCH_CHECK("mrfp_dwconv_bwd_data_kernel");
    const int nchunk = wc_cdiv(S, CM_WCH);
    hipLaunchKernelGGL(mrfp_dwconv_bwd_w_kernel, dim3(nchunk, N), dim3(256), 0, st, dy, x, part, lv, S, C);
    WC_LAUNCH_CHECK("mrfp_dwconv_bwd_w_kernel");
    hipLaunchKernelGGL(mrfp_dwconv_bwd_w_final_kernel, dim3(wc_cdiv(C * 26, 256)), dim3(256), 0, st, part, dw3, db3, dw5, db5,
                       N * nchunk, C, alpha);
    WC_LAUNCH_CHECK("mrfp_dwconv_bwd_w_final_kernel");
    return WC_OK;
}

extern "C" int wc_msda_prep_fwd(const float* ow, const float* bias_off, const float* bias_aw, const float* ref, float* loc,
                                float* attn, const int* h_shapes, int n_levels, int N, int Lq, int M, int P, int ld, int nl_ref,
                                void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(ow && ref && loc && attn && N > 0 && Lq > 0 && M > 0 && P > 0 && (nl_ref == 1 || nl_ref == n_levels),
                 "wc_msda_prep_fwd: bad argument");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0 && ld >= 3 * M * n_levels * P,
                 "wc_msda_prep_fwd: 1..8 levels, ld >= 3 * heads * levels * points");
    const long NQ = (long)N * Lq;
    hipLaunchKernelGGL(msda_prep_fwd_kernel, dim3((unsigned)wc_cdiv(NQ * M, 256)), dim3(256), 0, (hipStream_t)stream, ow, bias_off,
                       bias_aw, ref, loc, attn, lv, NQ, Lq, M, P, ld, nl_ref);
    WC_LAUNCH_CHECK("msda_prep_fwd_kernel");
    return WC_OK;
}

extern "C" int wc_msda_prep_bwd(const float* gloc, const float* gattn, const float* attn, float* dow32, void* dow16,
                                const int* h_shapes, int n_levels, int N, int Lq, int M, int P, int ld, void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(gloc && gattn && attn && (dow32 || dow16) && N > 0 && Lq > 0 && M > 0 && P > 0, "wc_msda_prep_bwd: bad argument");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0 && ld >= 3 * M * n_levels * P,
                 "wc_msda_prep_bwd: 1..8 levels, ld >= 3 * heads * levels * points");
    const long NQ = (long)N * Lq;
    long blocks = wc_cdiv(NQ * ld, 256 * 4);
    if (blocks > 65535 * 16) blocks = 65535 * 16;
    hipLaunchKernelGGL(msda_prep_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gloc, gattn, attn, dow32,
                       (__half*)dow16, lv, NQ, M, P, ld);
    WC_LAUNCH_CHECK("msda_prep_bwd_kernel");
    return WC_OK;
}

extern "C" int wc_rows_copy_f16(const void* src, int src_is_f32, void* dst, int B, int R, int C, long ld_src, long s_src,
                                long ld_dst, long s_dst, void* stream) {
    WC_CHECK_ARG(src && dst && B > 0 && B <= 65535 && R > 0 && C > 0 && ld_src >= C && ld_dst >= C, "wc_rows_copy_f16: bad argument");
    long blocks = wc_cdiv((long)R * C, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    if (src_is_f32)
        hipLaunchKernelGGL(rows_copy_f16_kernel<float>, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, (__half*)dst, R, C, ld_src, s_src, ld_dst, s_dst);
    else
        hipLaunchKernelGGL(rows_copy_f16_kernel<__half>, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream,
                           (const __half*)src, (__half*)dst, R, C, ld_src, s_src, ld_dst, s_dst);
    WC_LAUNCH_CHECK("rows_copy_f16_kernel");
    return WC_OK;
}

extern "C" int wc_rows_add_f32(const float* src, float* dst, int B, int R, int C, long ld_src, long s_src, long s_dst, float alpha,
                               void* stream) {
    WC_CHECK_ARG(src && dst && B > 0 && B <= 65535 && R > 0 && C > 0 && ld_src >= C, "wc_rows_add_f32: bad argument");
    long blocks = wc_cdiv((long)R * C, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rows_add_f32_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, src, dst, R, C, ld_src,
                       s_src, s_dst, alpha);
    WC_LAUNCH_CHECK("rows_add_f32_kernel");
    return WC_OK;
}
