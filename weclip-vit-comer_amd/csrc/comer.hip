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
    int sstart[CM_MAX_LEVELS], nstrips;      // horizontal strips of 8 pixels (depth-wise conv kernels): first strip per level, total
};

__device__ __forceinline__ float cm_gelu(float x) { return wc_gelu(x); }

// Depth-wise kernels: a thread owns one channel of a horizontal STRIP of CM_SW consecutive pixels of one map row, so a row
// segment of CM_SW + k - 1 inputs loaded once serves all k taps of the strip's pixels (3.75 / 7.5 loads per output for the
// 3x3 / 5x5 filters instead of 9 / 25); consecutive lanes are consecutive channels (coalesced NHWC rows), and the channel
// halves (3x3 | 5x5) fall on whole waves (C/2 a multiple of 64) or the branch is per lane (C = 64).
#define CM_SW 8

__device__ __forceinline__ void cm_strip(const CmLevels& lv, int t, int& l, int& py, int& x0) {
    l = 0;
    while (l + 1 < lv.n && t >= lv.sstart[l + 1]) ++l;
    const int local = t - lv.sstart[l], spr = (lv.W[l] + CM_SW - 1) / CM_SW;
    py = local / spr;
    x0 = (local - py * spr) * CM_SW;
}

__device__ __forceinline__ float cm_ld(const float* p) { return *p; }
__device__ __forceinline__ float cm_ld(const __half* p) { return __half2float(*p); }

// out[o] += sum_{dy,dx} w[(dy+R)*K + dx+R] * in(py+dy, x0+o+dx)   (FLIP: w index mirrored = transposed convolution)
template <int K, bool FLIP, typename T>
__device__ __forceinline__ void cm_conv_strip(const T* __restrict__ xb, const float* __restrict__ wc, int H, int W, int C, int py,
                                              int x0, float (&out)[CM_SW]) {
    constexpr int R = K / 2;
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = wc[FLIP ? K * K - 1 - i : i];
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
        // (every load is issued unconditionally from a clamped address and masked afterwards: a bounds BRANCH per load made hipcc
        //  wait for each load before the next one -- up to 60 dependent-latency loads per thread, 57 us for a 44-MB pass)
        const int yy = py + dy;
        const bool rok = yy >= 0 && yy < H;
        const long rb = (long)min(max(yy, 0), H - 1) * W;
        float xs[CM_SW + K - 1];
#pragma unroll
        for (int j = 0; j < CM_SW + K - 1; ++j) {
            const int xx = x0 + j - R;
            const float v = cm_ld(xb + (rb + min(max(xx, 0), W - 1)) * C);
            xs[j] = (rok && xx >= 0 && xx < W) ? v : 0.f;
        }
#pragma unroll
        for (int dx = 0; dx < K; ++dx)
#pragma unroll
            for (int o = 0; o < CM_SW; ++o) out[o] = fmaf(wr[(dy + R) * K + dx], xs[o + dx], out[o]);
    }
}

// x (N, S, C) f32 rows; w3 (C/2, 9), b3 (C/2), w5 (C/2, 25), b5 (C/2).  y = conv + bias (f32, optional), g16 = gelu(y) (f16).
// grid (ceil(nstrips / (256 / C)), N); thread = (strip lane, channel)
__global__ __launch_bounds__(256) void mrfp_dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                               const float* __restrict__ b3, const float* __restrict__ w5,
                                                               const float* __restrict__ b5, float* __restrict__ y,
                                                               __half* __restrict__ g16, CmLevels lv, int S, int C) {
    const int c = threadIdx.x % C, t = blockIdx.x * (256 / C) + threadIdx.x / C, n = blockIdx.y;
    if (t >= lv.nstrips) return;
    int l, py, x0;
    cm_strip(lv, t, l, py, x0);
    const int H = lv.H[l], W = lv.W[l], half = C >> 1, big = c >= half;
    const float* xb = x + ((long)n * S + lv.start[l]) * C + c;
    float out[CM_SW];
    const float bias = big ? b5[c - half] : b3[c];
#pragma unroll
    for (int o = 0; o < CM_SW; ++o) out[o] = bias;
    if (big) cm_conv_strip<5, false>(xb, w5 + (long)(c - half) * 25, H, W, C, py, x0, out);
    else cm_conv_strip<3, false>(xb, w3 + (long)c * 9, H, W, C, py, x0, out);
    const long o0 = ((long)n * S + lv.start[l] + (long)py * W + x0) * C + c;
#pragma unroll
    for (int o = 0; o < CM_SW; ++o) {
        if (x0 + o >= W) break;
        if (y) y[o0 + (long)o * C] = out[o];
        if (g16) g16[o0 + (long)o * C] = __float2half(cm_gelu(out[o]));
    }
}

// dx (N, S, C) = conv^T(dy): dx[p] = sum_taps w[tap] * dy[p - tap]; dy f16 or f32; written as f32 and / or f16
template <typename T>
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w3,
                                                                    const float* __restrict__ w5, float* __restrict__ dx32,
                                                                    __half* __restrict__ dx16, CmLevels lv, int S, int C) {
    const int c = threadIdx.x % C, t = blockIdx.x * (256 / C) + threadIdx.x / C, n = blockIdx.y;
    if (t >= lv.nstrips) return;
    int l, py, x0;
    cm_strip(lv, t, l, py, x0);
    const int H = lv.H[l], W = lv.W[l], half = C >> 1, big = c >= half;
    const T* db = dy + ((long)n * S + lv.start[l]) * C + c;
    float out[CM_SW];
#pragma unroll
    for (int o = 0; o < CM_SW; ++o) out[o] = 0.f;
    if (big) cm_conv_strip<5, true>(db, w5 + (long)(c - half) * 25, H, W, C, py, x0, out);
    else cm_conv_strip<3, true>(db, w3 + (long)c * 9, H, W, C, py, x0, out);
    const long o0 = ((long)n * S + lv.start[l] + (long)py * W + x0) * C + c;
#pragma unroll
    for (int o = 0; o < CM_SW; ++o) {
        if (x0 + o >= W) break;
        if (dx32) dx32[o0 + (long)o * C] = out[o];
        if (dx16) dx16[o0 + (long)o * C] = __float2half(out[o]);
    }
}

// acc[(dy+R)*K + dx+R] += sum_o g[o] * x(py+dy, x0+o+dx)
template <int K, typename T>
__device__ __forceinline__ void cm_wgrad_strip(const float* __restrict__ xb, const T* __restrict__ gb, int H, int W, int C, int py,
                                               int x0, float (&acc)[26]) {
    constexpr int R = K / 2;
    float g[CM_SW];
#pragma unroll
    for (int o = 0; o < CM_SW; ++o) {
        // (masked by a MULTIPLICATION: a select lets hipcc sink the load back into a branch, and behind a branch it waits for
        //  every load before issuing the next one)
        g[o] = cm_ld(gb + ((long)py * W + min(x0 + o, W - 1)) * C) * (x0 + o < W ? 1.f : 0.f);
        acc[25] += g[o];
    }
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
        const int yy = py + dy;
        const bool rok = yy >= 0 && yy < H;
        const long rb = (long)min(max(yy, 0), H - 1) * W;
        float xs[CM_SW + K - 1];
#pragma unroll
        for (int j = 0; j < CM_SW + K - 1; ++j) {
            const int xx = x0 + j - R;
            xs[j] = xb[(rb + min(max(xx, 0), W - 1)) * C] * ((rok && xx >= 0 && xx < W) ? 1.f : 0.f);
        }
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            float sum = 0.f;
#pragma unroll
            for (int o = 0; o < CM_SW; ++o) sum = fmaf(g[o], xs[o + dx], sum);
            acc[(dy + R) * K + dx] += sum;
        }
    }
}

// filter / bias gradients, stage 1: workgroup (chunk of CM_WST strips per strip lane, image) -> part[(n * nchunk + chunk)][26][c]
// (taps 0..24 [3x3 filters use 0..8] and the bias sum at 25).  256 threads = 256 / C strip lanes x C channels; the lanes are
// combined through LDS in a fixed order.
#define CM_WST 4
template <typename T>
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_w_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ part, CmLevels lv, int S, int C) {
    __shared__ float red[256 * 26];
    const int c = threadIdx.x % C, pl = threadIdx.x / C, npl = 256 / C, n = blockIdx.y;
    const int half = C >> 1, big = c >= half;
    float acc[26];
#pragma unroll
    for (int t = 0; t < 26; ++t) acc[t] = 0.f;
    const int t0 = blockIdx.x * (CM_WST * npl);
    for (int i = 0; i < CM_WST; ++i) {
        const int t = t0 + i * npl + pl;
        if (t >= lv.nstrips) break;
        int l, py, x0;
        cm_strip(lv, t, l, py, x0);
        const int H = lv.H[l], W = lv.W[l];
        const float* xb = x + ((long)n * S + lv.start[l]) * C + c;
        const T* gb = dy + ((long)n * S + lv.start[l]) * C + c;
        if (big) cm_wgrad_strip<5>(xb, gb, H, W, C, py, x0, acc);
        else cm_wgrad_strip<3>(xb, gb, H, W, C, py, x0, acc);
    }
#pragma unroll
    for (int t = 0; t < 26; ++t) red[t * 256 + threadIdx.x] = acc[t];
    __syncthreads();
    if (pl == 0) {
        float* out = part + ((long)n * gridDim.x + blockIdx.x) * C * 26 + c;      // [part][26][C]: 4*C-byte rows per store
#pragma unroll
        for (int t = 0; t < 26; ++t) {
            float sum = 0.f;
            for (int q = 0; q < npl; ++q) sum += red[t * 256 + q * C + c];
            out[(long)t * C] = sum;
        }
    }
}

// stage 2: out[c][t] = alpha * sum over the nparts partial rows (fixed order); dw3 (C/2, 9), db3, dw5 (C/2, 25), db5.
// A workgroup owns 16 consecutive (t, c) entries (64 contiguous bytes of every partial row); its 16 thread groups walk the partial
// rows interleaved, eight loads in flight each, and meet in LDS -- a thread per entry walking all rows alone was one long chain of
// dependent-latency loads (54 us for 336 rows).
__global__ __launch_bounds__(256) void mrfp_dwconv_bwd_w_final_kernel(const float* __restrict__ part, float* __restrict__ dw3,
                                                                       float* __restrict__ db3, float* __restrict__ dw5,
                                                                       float* __restrict__ db5, int nparts, int C, float alpha) {
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + j;                   // t * C + c
    const long stride = (long)C * 26;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < C * 26) {
        int q = pg;
        for (; q + 7 * 16 < nparts; q += 8 * 16)
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(long)(q + u * 16) * stride + i];
        for (; q < nparts; q += 16) s[0] += part[(long)q * stride + i];
    }
    red[pg][j] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (pg != 0 || i >= C * 26) return;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][j];
    v *= alpha;
    const int t = i / C, c = i - t * C, half = C >> 1;
    if (c < half && t >= 9 && t < 25) return;
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
    int t = 0;
    for (int l = 0; l < n_levels; ++l) {
        lv->sstart[l] = t;
        t += lv->H[l] * ((lv->W[l] + 7) / 8);
    }
    lv->nstrips = t;
    return 0;
}

extern "C" int wc_mrfp_dwconv_fwd(const float* x, const float* w3, const float* b3, const float* w5, const float* b5, float* y,
                                  void* g16, const int* h_shapes, int n_levels, int N, int C, void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(x && w3 && b3 && w5 && b5 && (y || g16) && N > 0 && N <= 65535 && (C == 64 || C == 128 || C == 256),
                 "wc_mrfp_dwconv_fwd: bad argument (C = 64, 128 or 256)");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0, "wc_mrfp_dwconv_fwd: 1..8 levels with positive sizes");
    hipLaunchKernelGGL(mrfp_dwconv_fwd_kernel, dim3(wc_cdiv(lv.nstrips, 256 / C), N), dim3(256), 0, (hipStream_t)stream, x, w3, b3,
                       w5, b5, y, (__half*)g16, lv, S, C);
    WC_LAUNCH_CHECK("mrfp_dwconv_fwd_kernel");
    return WC_OK;
}

/* dy: f16 (dy_is_f16) or f32.  dx32 / dx16 (either may be null): gradient w.r.t. x; dw3 / db3 / dw5 / db5 = alpha * filter /
 * bias gradients; part: workspace of wc_mrfp_dwconv_parts(...) * C * 26 floats. */
static int cm_wparts(const CmLevels& lv, int C) { return wc_cdiv(lv.nstrips, CM_WST * (256 / C)); }

extern "C" int wc_mrfp_dwconv_parts(const int* h_shapes, int n_levels, int N, int C, long* n_parts) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(n_parts && N > 0 && (C == 64 || C == 128 || C == 256) && cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0,
                 "wc_mrfp_dwconv_parts: bad argument");
    *n_parts = (long)N * cm_wparts(lv, C);
    return WC_OK;
}

extern "C" int wc_mrfp_dwconv_bwd(const void* dy, int dy_is_f16, const float* x, const float* w3, const float* w5, float* dx32,
                                  void* dx16, float* dw3, float* db3, float* dw5, float* db5, float* part, float alpha,
                                  const int* h_shapes, int n_levels, int N, int C, void* stream) {
    CmLevels lv;
    int S = 0;
    WC_CHECK_ARG(dy && x && w3 && w5 && (dx32 || dx16) && dw3 && db3 && dw5 && db5 && part && N > 0 && N <= 65535 &&
                 (C == 64 || C == 128 || C == 256), "wc_mrfp_dwconv_bwd: bad argument");
    WC_CHECK_ARG(cm_fill_levels(&lv, h_shapes, n_levels, &S) == 0, "wc_mrfp_dwconv_bwd: 1..8 levels with positive sizes");
    hipStream_t st = (hipStream_t)stream;
    const dim3 gd(wc_cdiv(lv.nstrips, 256 / C), N);
    const int nchunk = cm_wparts(lv, C);
    if (dy_is_f16) {
        hipLaunchKernelGGL(mrfp_dwconv_bwd_data_kernel<__half>, gd, dim3(256), 0, st, (const __half*)dy, w3, w5, dx32, (__half*)dx16, lv, S, C);
        hipLaunchKernelGGL(mrfp_dwconv_bwd_w_kernel<__half>, dim3(nchunk, N), dim3(256), 0, st, (const __half*)dy, x, part, lv, S, C);
    } else {
        hipLaunchKernelGGL(mrfp_dwconv_bwd_data_kernel<float>, gd, dim3(256), 0, st, (const float*)dy, w3, w5, dx32, (__half*)dx16, lv, S, C);
        hipLaunchKernelGGL(mrfp_dwconv_bwd_w_kernel<float>, dim3(nchunk, N), dim3(256), 0, st, (const float*)dy, x, part, lv, S, C);
    }
    WC_LAUNCH_CHECK("mrfp_dwconv_bwd kernels");
    hipLaunchKernelGGL(mrfp_dwconv_bwd_w_final_kernel, dim3(wc_cdiv(C * 26, 16)), dim3(256), 0, st, part, dw3, db3, dw5, db5,
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

// ---------------------------------------------------------------------------------------------
// Gradients of the gated output projection of CTI-toV, v1 = v + gamma * (o1 Wop^T + bop), from the UN-GATED products the
// weight-gradient GEMM delivers (G = dv1^T o1 (C x K), s = dv1^T 1 (C)):
//   dWop = diag(gamma) G,   dbop = gamma * s,   dgamma = rowsum(Wop * G) + bop * s.
// One workgroup per output channel (row); replaces ~9 element-wise / reduction launches of torch glue per stage.
__global__ __launch_bounds__(256) void cti_gate_grads_kernel(const float* __restrict__ G, const float* __restrict__ gs,
                                                              const float* __restrict__ gam, const float* __restrict__ Wop,
                                                              const float* __restrict__ bop, float* __restrict__ dWop,
                                                              float* __restrict__ dbop, float* __restrict__ dgam, int K) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const float gr = gam[r];
    float acc = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) {
        const float g = G[(long)r * K + c];
        dWop[(long)r * K + c] = gr * g;
        acc += Wop[(long)r * K + c] * g;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        dbop[r] = gr * gs[r];
        dgam[r] = acc + bop[r] * gs[r];
    }
}

extern "C" int wc_cti_gate_grads(const float* G, const float* gs, const float* gamma, const float* Wop, const float* bop,
                                 float* dWop, float* dbop, float* dgamma, int C, int K, void* stream) {
    WC_CHECK_ARG(G && gs && gamma && Wop && bop && dWop && dbop && dgamma && C > 0 && K > 0, "wc_cti_gate_grads: bad argument");
    hipLaunchKernelGGL(cti_gate_grads_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, G, gs, gamma, Wop, bop, dWop, dbop, dgamma, K);
    WC_LAUNCH_CHECK("cti_gate_grads_kernel");
    return WC_OK;
}
