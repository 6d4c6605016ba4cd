// GradCAM on the last CLIP block without autograd, for all (image, class) pairs of a batch.
//
// Replaces reference pytorch_grad_cam/base_cam.py:62-154 + grad_cam.py:16-23 +
// activations_and_gradients.py:19-47 driving autograd through CLIP.forward_last_layer
// (clip/model.py:407-429): one block-12 forward per IMAGE (the reference repeats it per class)
// and one analytic backward per (image, class) PAIR, batched as P right-hand sides.  The backward
// is linear in the seed; GradCAM only needs w_c = mean over patch tokens of dp_j/dA[token, c]
// (A = ln_1 output), and A feeds nothing but the in-projection, so w = colsum(dqkv) W_in / hw.
// Column sums of dq, dk, dv over the patch tokens reduce to (per head, S = natural scores):
//   u_l  = sum_q dS[q,l]                 dS = P * (dP - delta),  dP = dO V^T,  delta = rowsum(dO*O)
//   cq   = sum_l u_l k_l / sqrt(dh)
//   ck   = - sum_q dS[q,0] q_q / sqrt(dh)            (rows of dS sum to zero)
//   cv   = sum_q (1 - P[q,0]) dO_q                   (rows of P sum to one)
// (dO of the CLS query is exactly 0: the class logit pools patch tokens only.)
// Gradients are carried multiplied by a power of two `gs` wherever they are stored as fp16 hi+lo
// MFMA operands, and unscaled where the reference itself rounds them to fp16 (backward of the
// forced-fp16 out-projection, clip/myAtt.py:321).
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// (a) partial[b, chunk, :] = sum over tokens l in chunk, l >= 1, of ln_post(x2[b, l, :])
//     chunk = LNP_ROWS token rows (one per wave iteration, the row held in registers: one read of x2)
#define LNP_ROWS 16
template <int NV>   // E <= 64 * NV
__global__ __launch_bounds__(256) void lnpost_pool_kernel(const float* __restrict__ x2,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bb, float eps,
                                                           float* __restrict__ partial, int L, int E,
                                                           int nchunk) {
    extern __shared__ float sm[];   // [4][E]
    const int b = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[NV], wv_[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        acc[i] = 0.f;
        wv_[i] = e < E ? w[e] : 0.f;
        bv[i] = e < E ? bb[e] : 0.f;
    }
    const int l0 = chunk * LNP_ROWS;
    for (int l = l0 + wv; l < l0 + LNP_ROWS && l < L; l += 4) {
        if (l == 0) continue;
        const float* xr = x2 + ((long)b * L + l) * E;
        float xv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = lane + 64 * i;
            xv[i] = e < E ? xr[e] : 0.f;
            s += xv[i];
        }
        const float mean = wave_sum(s) / E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float d = (lane + 64 * i < E) ? xv[i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) / E + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < E) acc[i] += (xv[i] - mean) * rstd * wv_[i] + bv[i];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < E) sm[wv * E + lane + 64 * i] = acc[i];
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256)
        partial[((long)b * nchunk + chunk) * E + e] = sm[e] + sm[E + e] + sm[2 * E + e] + sm[3 * E + e];
}

// (b) per pair: f = mean token feature, y = f proj, probs = softmax(s * yhat . that), seed -> df
// text rows are pre-normalised; text_idx[p, t] indexes rows of `text` (T_p = n_text[p] rows in use).
__global__ __launch_bounds__(512) void cam_head_kernel(const float* __restrict__ partial,
                                                        const float* __restrict__ proj,
                                                        const float* __restrict__ text,
                                                        const int* __restrict__ text_idx,
                                                        const int* __restrict__ n_text,
                                                        const int* __restrict__ pair_img,
                                                        const int* __restrict__ pair_cls, float logit_scale,
                                                        float* __restrict__ probs, float* __restrict__ df,
                                                        int L, int E, int Ed, int nchunk, int Tmax) {
    extern __shared__ float sm[];   // f[E] | y[Ed] | dy[Ed] | logit[Tmax] | red[16]
    float* f = sm;
    float* y = f + E;
    float* dy = y + Ed;
    float* lg = dy + Ed;
    float* red = lg + Tmax;
    const int p = blockIdx.x, img = pair_img[p], cls = pair_cls[p], T = n_text[p], tid = threadIdx.x;
    for (int e = tid; e < E; e += 512) {
        const float* pp = partial + (long)img * nchunk * E + e;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // 8 independent loads in flight
        int c = 0;
        for (; c + 8 <= nchunk; c += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += pp[(long)(c + u) * E];
        for (; c < nchunk; ++c) s[0] += pp[(long)c * E];
        f[e] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) / (L - 1);
    }
    __syncthreads();
    for (int t = tid; t < Ed; t += 512) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int e = 0;
        for (; e + 16 <= E; e += 16) {           // 16 coalesced row loads in flight per thread
            float pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pv[u] = proj[(long)(e + u) * Ed + t];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                s0 = fmaf(f[e + u], pv[u], s0);
                s1 = fmaf(f[e + u + 1], pv[u + 1], s1);
                s2 = fmaf(f[e + u + 2], pv[u + 2], s2);
                s3 = fmaf(f[e + u + 3], pv[u + 3], s3);
            }
        }
        for (; e < E; ++e) s0 = fmaf(f[e], proj[(long)e * Ed + t], s0);
        y[t] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    float q = 0.f;
    for (int t = tid; t < Ed; t += 512) q += y[t] * y[t];
    const float ynorm = sqrtf(block_sum(q, red));
    __syncthreads();
    for (int t = tid; t < Ed; t += 512) y[t] /= ynorm;     // yhat
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    for (int t = wv; t < T; t += 8) {
        const float* tr = text + (long)text_idx[(long)p * Tmax + t] * Ed;
        float s = 0.f;
        for (int k = lane; k < Ed; k += 64) s += y[k] * tr[k];
        s = wave_sum(s);
        if (lane == 0) lg[t] = logit_scale * s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, lg[t]);
    float den = 0.f;
    for (int t = 0; t < T; ++t) den += expf(lg[t] - mx);
    __syncthreads();
    if (tid < T) {
        const float pr = expf(lg[tid] - mx) / den;
        probs[(long)p * Tmax + tid] = pr;
        lg[tid] = pr;
    }
    __syncthreads();
    const float pj = lg[cls];
    // dyhat = s * sum_t dlogit_t that_t,  dlogit_t = pj (delta_jt - p_t)
    for (int k = tid; k < Ed; k += 512) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const float dl = pj * ((t == cls ? 1.f : 0.f) - lg[t]);
            s = fmaf(dl, text[(long)text_idx[(long)p * Tmax + t] * Ed + k], s);
        }
        dy[k] = logit_scale * s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int k = tid; k < Ed; k += 512) dot += y[k] * dy[k];
    dot = block_sum(dot, red);
    __syncthreads();
    for (int k = tid; k < Ed; k += 512) dy[k] = (dy[k] - y[k] * dot) / ynorm;   // d/dy of y/|y|
    __syncthreads();
    // df = proj dy: 8 rows of proj per wave at a time, 8 lanes per row reading 16-B pieces (coalesced 128 B per
    // row and step, all loads of a row independent), 3-step shuffle reduction over the 8 lanes
    if ((Ed & 3) == 0) {
        const int rsub = lane >> 3, j = lane & 7;
        for (int e0 = wv * 8; e0 < E; e0 += 64) {
            const int e = e0 + rsub;
            float s = 0.f;
            if (e < E) {
                const float* pr = proj + (long)e * Ed;
#pragma unroll 4
                for (int k = 4 * j; k < Ed; k += 32) {
                    const float4 pv = *reinterpret_cast<const float4*>(pr + k);
                    s = fmaf(pv.x, dy[k], s);
                    s = fmaf(pv.y, dy[k + 1], s);
                    s = fmaf(pv.z, dy[k + 2], s);
                    s = fmaf(pv.w, dy[k + 3], s);
                }
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (j == 0 && e < E) df[(long)p * E + e] = s;
        }
    } else {
        for (int e = tid; e < E; e += 512) {
            const float* pr = proj + (long)e * Ed;
            float s = 0.f;
            for (int k = 0; k < Ed; ++k) s = fmaf(pr[k], dy[k], s);
            df[(long)p * E + e] = s;
        }
    }
}

// (c) dx2[p,l,:] = gs * LNpost_bwd(df[p]/(L-1); x2[img,l,:]) for l >= 1, 0 for l = 0.
//     outputs: fp32 and fp16 hi/lo (GEMM operand).  One wave per row.
template <int NV>   // NV > 0: E <= 64 * NV, register form; 0: any E
__global__ __launch_bounds__(256) void lnpost_bwd_kernel(const float* __restrict__ df,
                                                          const float* __restrict__ x2,
                                                          const float* __restrict__ w, float eps, float gs,
                                                          const int* __restrict__ pair_img,
                                                          float* __restrict__ d32, __half* __restrict__ dhi,
                                                          __half* __restrict__ dlo, int L, int E, long rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int p = row / L, l = row % L;
    const long o = row * E;
    if (l == 0) {
        for (int e = lane; e < E; e += 64) {
            d32[o + e] = 0.f;
            dhi[o + e] = __float2half(0.f);
            if (dlo) dlo[o + e] = __float2half(0.f);
        }
        return;
    }
    const float* xr = x2 + ((long)pair_img[p] * L + l) * E;
    const float* g0 = df + (long)p * E;
    const float invn = 1.0f / (L - 1);
    if constexpr (NV > 0) {
        // E <= 64 * NV: the row and its upstream gradient live in registers -- one read of each, all loads of the row in flight
        // together (clamped addresses, multiplicative mask: see ln2_bwd_add_kernel), the same summation order as the loop form
        float xv[NV], gv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = lane + 64 * i, ec = e < E ? e : 0;
            const float ok = e < E ? 1.f : 0.f;
            xv[i] = xr[ec] * ok;
            gv[i] = g0[ec] * invn * w[ec] * ok;
            s += xv[i];
        }
        const float mean = wave_sum(s) / E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float d = (lane + 64 * i < E) ? xv[i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) / E + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            sg += gv[i];
            sgx += gv[i] * (xv[i] - mean) * rstd;
        }
        sg = wave_sum(sg) / E;
        sgx = wave_sum(sgx) / E;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = lane + 64 * i;
            if (e < E) {
                const float v = gs * rstd * (gv[i] - sg - (xv[i] - mean) * rstd * sgx);
                d32[o + e] = v;
                const __half h = __float2half(v);
                dhi[o + e] = h;
                if (dlo) dlo[o + e] = __float2half(v - __half2float(h));
            }
        }
    } else {
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s += xr[e];
        const float mean = wave_sum(s) / E;
        float q = 0.f;
        for (int e = lane; e < E; e += 64) { const float d = xr[e] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / E + eps);
        float sg = 0.f, sgx = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float g = g0[e] * invn * w[e];
            sg += g;
            sgx += g * (xr[e] - mean) * rstd;
        }
        sg = wave_sum(sg) / E;
        sgx = wave_sum(sgx) / E;
        for (int e = lane; e < E; e += 64) {
            const float g = g0[e] * invn * w[e];
            const float v = gs * rstd * (g - sg - (xr[e] - mean) * rstd * sgx);
            d32[o + e] = v;
            const __half h = __float2half(v);
            dhi[o + e] = h;
            if (dlo) dlo[o + e] = __float2half(v - __half2float(h));
        }
    }
}

// (f) g16[p,l,:] = fp16( (dx2s + LN2_bwd(da2s; x1[img,l,:])) / gs )  -- gradient reaching the fp16
//     out-projection output, rounded like autograd does for the fp16 tensor (myAtt.py:321).
template <int NV>   // E <= 64 * NV; the row of x1 / da2 is held in registers (one read of each)
__global__ __launch_bounds__(256) void ln2_bwd_add_kernel(const float* __restrict__ da2,
                                                           const float* __restrict__ dx2,
                                                           const float* __restrict__ x1,
                                                           const float* __restrict__ w, float eps,
                                                           float inv_gs, const int* __restrict__ pair_img,
                                                           __half* __restrict__ g16, int L, int E, long rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int p = row / L, l = row % L;
    const long o = row * E;
    const float* xr = x1 + ((long)pair_img[p] * L + l) * E;
    float xv[NV], gv[NV], dv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // (clamped addresses and a multiplicative mask: `ok ? load : 0` is compiled into a branch around the load, and behind a
        //  branch hipcc waits for every load before it issues the next one -- 16 dependent latencies per row)
        const int e = lane + 64 * i, ec = e < E ? e : 0;
        const float ok = e < E ? 1.f : 0.f;
        xv[i] = xr[ec] * ok;
        gv[i] = da2[o + ec] * w[ec] * ok;
        dv[i] = dx2[o + ec] * ok;
        s += xv[i];
    }
    const float mean = wave_sum(s) / E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d = (lane + 64 * i < E) ? xv[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / E + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        sg += gv[i];
        sgx += gv[i] * (xv[i] - mean) * rstd;
    }
    sg = wave_sum(sg) / E;
    sgx = wave_sum(sgx) / E;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        if (e < E) {
            const float v = (dv[i] + rstd * (gv[i] - sg - (xv[i] - mean) * rstd * sgx)) * inv_gs;
            g16[o + e] = __float2half(v);
        }
    }
}

// (h) delta[p,h,q] = sum_d dO[p,q,h,d] * O[img,q,h,d].  One wave per token row (coalesced reads of the
//     whole E-wide row), segmented shuffle reduction per head (DH = 32 or 64).
__global__ __launch_bounds__(256) void attn_delta_kernel(const __half* __restrict__ dO,
                                                          const float* __restrict__ o32,
                                                          const int* __restrict__ pair_img,
                                                          float* __restrict__ delta, int L, int H, int DH,
                                                          long total) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // over (p, q)
    if (row >= total) return;
    const int lane = threadIdx.x & 63;
    const int q = row % L;
    const long p = row / L;
    const int E = H * DH;
    const __half* a = dO + row * E;
    const float* b = o32 + ((long)pair_img[p] * L + q) * E;
    // 8 consecutive elements per lane (16-byte dO load, two 16-byte O loads): a head of DH elements lives in DH / 8
    // adjacent lanes, summed by 2-3 shuffles (the element-per-lane form: 2- and 4-byte loads, 5-6 shuffles per 64 elements)
    const int nch = E >> 3, lph = DH >> 3;             // 16-byte chunks per row, lanes per head
    for (int c0 = 0; c0 < nch; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < nch) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(a + c * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8), b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
            const __half* ah = reinterpret_cast<const __half*>(&av);
            s = __half2float(ah[0]) * b0.x + __half2float(ah[1]) * b0.y + __half2float(ah[2]) * b0.z + __half2float(ah[3]) * b0.w +
                __half2float(ah[4]) * b1.x + __half2float(ah[5]) * b1.y + __half2float(ah[6]) * b1.z + __half2float(ah[7]) * b1.w;
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        if (lph == 8) s += __shfl_xor(s, 4, 64);
        if (c < nch && (c & (lph - 1)) == 0) delta[(p * H + c / lph) * L + q] = s;
    }
}

// (i) u[p,h,l] = sum_q dS[q,l];  column 0:  dS0[p,h,q] = dS[q,0], P0[p,h,q] = P[q,0].
// grid (key tiles of 128, H, P); 4 waves as 2(q) x 2(key), 128-row q tiles streamed through LDS.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_colsum_kernel(const __half* __restrict__ qkv,
                                                               const __half* __restrict__ dO,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               const int* __restrict__ pair_img,
                                                               float* __restrict__ u, float* __restrict__ dS0,
                                                               float* __restrict__ P0, int L, int H, int E, int origin,
                                                               int nkt, int PH) {
    constexpr int KS = DH / 16;
    constexpr int ROW = DH * 2 + 16;
    constexpr int TB = 128 * ROW;
    constexpr int CH = DH / 8;
    constexpr int NC = 128 * CH / 256;
    constexpr int QBUF = 2 * TB + 1024;   // Q tile | dO tile | lse[128] | delta[128]
    // K | V | QBUF | red[2][128]: ONE q-tile buffer (76 KiB in all) so that two workgroups share a CU; the next
    // tile waits in registers during the MFMAs and is stored between two barriers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l31 = lane & 31, wr = wave >> 1, wc = wave & 1;
    // XCD-aware order: the key tiles of one (pair, head) run back to back on one XCD, which then streams that
    // head's Q / dO tiles from its own L2 (they were re-fetched from HBM by every key tile before: 5x traffic).
    // Tiles cover keys and queries [origin, L); attn_bwd_colsum_edge_kernel adds the first `origin` of both.
    const int slot = blockIdx.x >> 3;
    const int ph = (slot / nkt) * 8 + (blockIdx.x & 7);
    if (ph >= PH) return;
    const int k0 = origin + (slot - (slot / nkt) * nkt) * 128, h = ph % H, p = ph / H, img = pair_img[p];
    const long ldq = 3L * E;
    const __half* qb = qkv + (long)img * L * ldq + (long)h * DH;
    const __half* dob = dO + (long)p * L * E + (long)h * DH;
    const float* lseb = lse + ((long)img * H + h) * L;
    const float* delb = delta + ((long)p * H + h) * L;
    char* Ks = smem;
    char* Vs = smem + TB;
    char* Qs = smem + 2 * TB;
    float* red = reinterpret_cast<float*>(smem + 2 * TB + QBUF);

    // K, V tiles of this key block (rows clamped; padded keys are masked by zero dS below)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + 256 * i;
        int kr = k0 + c / CH;
        if (kr > L - 1) kr = L - 1;
        *reinterpret_cast<u32x4*>(Ks + (c / CH) * ROW + (c % CH) * 16) =
            *reinterpret_cast<const u32x4*>(qb + (long)kr * ldq + E + (c % CH) * 8);
        *reinterpret_cast<u32x4*>(Vs + (c / CH) * ROW + (c % CH) * 16) =
            *reinterpret_cast<const u32x4*>(qb + (long)kr * ldq + 2 * E + (c % CH) * 8);
    }
    u32x4 rq[NC], rd[NC];
    float rl = 0.f, rdl = 0.f;
#undef GLOAD
#define GLOAD(t_) \
    { \
        const int t__ = (t_); \
        _Pragma("unroll") \
        for (int i = 0; i < NC; ++i) { \
            const int c = tid + 256 * i; \
            int qr = origin + t__ * 128 + c / CH; \
            if (qr > L - 1) qr = L - 1; \
            rq[i] = *reinterpret_cast<const u32x4*>(qb + (long)qr * ldq + (c % CH) * 8); \
            rd[i] = *reinterpret_cast<const u32x4*>(dob + (long)qr * E + (c % CH) * 8); \
        } \
        if (tid < 128) { \
            const int qr = origin + t__ * 128 + tid; \
            rl = qr < L ? lseb[qr] : 1.0e30f; \
            rdl = qr < L ? delb[qr] : 0.f; \
        } \
    }
#undef LSTORE
#define LSTORE(buf_) \
    { \
        const int buf__ = (buf_); \
        char* base = Qs + buf__ * QBUF; \
        _Pragma("unroll") \
        for (int i = 0; i < NC; ++i) { \
            const int c = tid + 256 * i; \
            *reinterpret_cast<u32x4*>(base + (c / CH) * ROW + (c % CH) * 16) = rq[i]; \
            *reinterpret_cast<u32x4*>(base + TB + (c / CH) * ROW + (c % CH) * 16) = rd[i]; \
        } \
        if (tid < 128) { \
            reinterpret_cast<float*>(base + 2 * TB)[tid] = rl; \
            reinterpret_cast<float*>(base + 2 * TB + 512)[tid] = rdl; \
        } \
    }
    float usum[2] = {0.f, 0.f};
    const int nt = (L - origin + 127) / 128;
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    const bool col0 = (k0 == 0 && wc == 0 && l31 == 0);
    for (int t = 0; t < nt; ++t) {
        const int buf = 0;
        if (t + 1 < nt) GLOAD(t + 1);
        const char* base = Qs + buf * QBUF;
        const float* ls = reinterpret_cast<const float*>(base + 2 * TB);
        const float* dl = reinterpret_cast<const float*>(base + 2 * TB + 512);
        f32x16 s[2][2], dp[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                s[mi][0][r] = s[mi][1][r] = -ls[rr];
                dp[mi][0][r] = dp[mi][1][r] = -dl[rr];
            }
        const char* Aq = base + (wr * 64 + l31) * ROW + hh * 16;
        const char* Ad = base + TB + (wr * 64 + l31) * ROW + hh * 16;
        const char* Bk = Ks + (wc * 64 + l31) * ROW + hh * 16;
        const char* Bv = Vs + (wc * 64 + l31) * ROW + hh * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 q0 = *reinterpret_cast<const f16x8*>(Aq + ks * 32);
            const f16x8 q1 = *reinterpret_cast<const f16x8*>(Aq + 32 * ROW + ks * 32);
            const f16x8 kk0 = *reinterpret_cast<const f16x8*>(Bk + ks * 32);
            const f16x8 kk1 = *reinterpret_cast<const f16x8*>(Bk + 32 * ROW + ks * 32);
            s[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, kk0, s[0][0], 0, 0, 0);
            s[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, kk1, s[0][1], 0, 0, 0);
            s[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, kk0, s[1][0], 0, 0, 0);
            s[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, kk1, s[1][1], 0, 0, 0);
            const f16x8 d0 = *reinterpret_cast<const f16x8*>(Ad + ks * 32);
            const f16x8 d1 = *reinterpret_cast<const f16x8*>(Ad + 32 * ROW + ks * 32);
            const f16x8 v0 = *reinterpret_cast<const f16x8*>(Bv + ks * 32);
            const f16x8 v1 = *reinterpret_cast<const f16x8*>(Bv + 32 * ROW + ks * 32);
            dp[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, v0, dp[0][0], 0, 0, 0);
            dp[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, v1, dp[0][1], 0, 0, 0);
            dp[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, v0, dp[1][0], 0, 0, 0);
            dp[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, v1, dp[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(s[mi][ni][r]);
                    const float ds = pr * dp[mi][ni][r];
                    usum[ni] += ds;
                    if (ni == 0 && col0) {
                        const int q = origin + t * 128 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (q < L) {
                            dS0[((long)p * H + h) * L + q] = ds;
                            P0[((long)p * H + h) * L + q] = pr;
                        }
                    }
                }
        __syncthreads();                       // every wave is done reading the q-tile buffer
        if (t + 1 < nt) LSTORE(0);
        __syncthreads();
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        usum[ni] += __shfl_xor(usum[ni], 32, 64);
        if (hh == 0) red[wr * 128 + wc * 64 + ni * 32 + l31] = usum[ni];
    }
    __syncthreads();
    if (tid < 128) {
        const int key = k0 + tid;
        if (key < L) u[((long)p * H + h) * L + key] = red[tid] + red[128 + tid];
    }
}

// Edge part of (i) when the tiles start at `r` = L % 128 (the CLS token): one workgroup per (pair, head).
//   keys l < r, every query q:   dS[q,l] -> dS0 / P0 (l == 0) and u[l] = sum_q dS[q,l]
//   queries q < r, keys l >= r:  u[l] += dS[q,l]        (runs after the tiled kernel on the same stream)
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_colsum_edge_kernel(const __half* __restrict__ qkv, const __half* __restrict__ dO,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    const int* __restrict__ pair_img, float* __restrict__ u,
                                                                    float* __restrict__ dS0, float* __restrict__ P0, int L, int H,
                                                                    int E, int r) {
    __shared__ float red[16];
    const int tid = threadIdx.x, h = blockIdx.x, p = blockIdx.y, img = pair_img[p];
    const long ldq = 3L * E;
    const __half* qb = qkv + (long)img * L * ldq + (long)h * DH;
    const __half* dob = dO + (long)p * L * E + (long)h * DH;
    const float* lseb = lse + ((long)img * H + h) * L;
    const float* delb = delta + ((long)p * H + h) * L;
    const long ob = ((long)p * H + h) * L;
    for (int l = 0; l < r; ++l) {                      // key l, thread per query
        float usum = 0.f;
        for (int q = tid; q < L; q += 256) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                const f16x8 qv = *reinterpret_cast<const f16x8*>(qb + (long)q * ldq + c * 8);
                const f16x8 kv = *reinterpret_cast<const f16x8*>(qb + (long)l * ldq + E + c * 8);
                const f16x8 dv = *reinterpret_cast<const f16x8*>(dob + (long)q * E + c * 8);
                const f16x8 vv = *reinterpret_cast<const f16x8*>(qb + (long)l * ldq + 2 * E + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s = fmaf((float)qv[j], (float)kv[j], s);
                    dp = fmaf((float)dv[j], (float)vv[j], dp);
                }
            }
            const float pr = __builtin_amdgcn_exp2f(s - lseb[q]);
            const float ds = pr * (dp - delb[q]);
            usum += ds;
            if (l == 0) { dS0[ob + q] = ds; P0[ob + q] = pr; }
        }
        usum = block_sum(usum, red);
        if (tid == 0) u[ob + l] = usum;
    }
    for (int l = r + tid; l < L; l += 256) {           // thread per key, the first r queries
        float add = 0.f;
        for (int q = 0; q < r; ++q) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                const f16x8 qv = *reinterpret_cast<const f16x8*>(qb + (long)q * ldq + c * 8);
                const f16x8 kv = *reinterpret_cast<const f16x8*>(qb + (long)l * ldq + E + c * 8);
                const f16x8 dv = *reinterpret_cast<const f16x8*>(dob + (long)q * E + c * 8);
                const f16x8 vv = *reinterpret_cast<const f16x8*>(qb + (long)l * ldq + 2 * E + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s = fmaf((float)qv[j], (float)kv[j], s);
                    dp = fmaf((float)dv[j], (float)vv[j], dp);
                }
            }
            add += __builtin_amdgcn_exp2f(s - lseb[q]) * (dp - delb[q]);
        }
        u[ob + l] += add;
    }
}

// (j) column sums of dq, dk, dv for one (pair, head): c[p, {0,E,2E} + h*DH + d]
// thread = (16-byte chunk of the head's DH halves, token slice): 256 / (DH/8) slices, 16-B row loads.
__global__ __launch_bounds__(256) void qkv_colsum_kernel(const __half* __restrict__ qkv,
                                                          const __half* __restrict__ dO,
                                                          const float* __restrict__ u,
                                                          const float* __restrict__ dS0,
                                                          const float* __restrict__ P0,
                                                          const int* __restrict__ pair_img,
                                                          float* __restrict__ c, int L, int H, int DH,
                                                          float inv_sqrt_dh, float inv_qscale) {
    __shared__ float red[3][64][65];          // [q|k|v][slice][d]
    const int h = blockIdx.x, p = blockIdx.y, img = pair_img[p];
    const int E = H * DH, nch = DH >> 3, nsl = 256 / nch;          // 8 chunks x 32 slices (DH 64), 4 x 64 (DH 32)
    const int ch = threadIdx.x % nch, sl = threadIdx.x / nch;
    const __half* qb = qkv + (long)img * L * 3 * E + (long)h * DH + ch * 8;
    const __half* dob = dO + (long)p * L * E + (long)h * DH + ch * 8;
    const float* ub = u + ((long)p * H + h) * L;
    const float* sb = dS0 + ((long)p * H + h) * L;
    const float* pb = P0 + ((long)p * H + h) * L;
    float cq[8], ck[8], cv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cq[j] = ck[j] = cv[j] = 0.f;
    for (int l = sl; l < L; l += nsl) {
        const __half* row = qb + (long)l * 3 * E;
        const f16x8 kv = *reinterpret_cast<const f16x8*>(row + E);
        const f16x8 qv = *reinterpret_cast<const f16x8*>(row);
        const f16x8 dv = *reinterpret_cast<const f16x8*>(dob + (long)l * E);
        const float a = ub[l], s0 = sb[l], om = 1.0f - pb[l];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cq[j] = fmaf(a, (float)kv[j], cq[j]);
            ck[j] = fmaf(s0, (float)qv[j], ck[j]);
            cv[j] = fmaf(om, (float)dv[j], cv[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][sl][ch * 8 + j] = cq[j];
        red[1][sl][ch * 8 + j] = ck[j];
        red[2][sl][ch * 8 + j] = cv[j];
    }
    __syncthreads();
    const int d = threadIdx.x & 63, w3 = threadIdx.x >> 6;      // waves 0..2 finish q, k, v
    if (w3 < 3 && d < DH) {
        float t = 0.f;
        for (int k = 0; k < nsl; ++k) t += red[w3][k][d];
        float* out = c + (long)p * 3 * E + h * DH + d;
        if (w3 == 0) out[0] = t * inv_sqrt_dh;
        else if (w3 == 1) out[E] = -t * inv_qscale;
        else out[2 * E] = t;
    }
}

// (l) cam[p, l] = norm(norm(relu(sum_c w[p,c] A[img, l+1, c]))), norm(z) = (z - min z)/(1e-7 + max(z - min z))
//     base_cam.py:56-60,144-154 + utils/image.py:51-61.  cam_raw_kernel: one wave per (pair, token), 16 tokens
//     per block; cam_norm_kernel: one block per pair does the two min-max passes in LDS.
__global__ __launch_bounds__(1024) void cam_raw_kernel(const float* __restrict__ a32, const float* __restrict__ w,
                                                        const int* __restrict__ pair_img, float* __restrict__ cam, int L,
                                                        int E) {
    const int p = blockIdx.y, img = pair_img[p], lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l = blockIdx.x * 16 + wv, hw = L - 1;
    if (l >= hw) return;
    const float* ar = a32 + ((long)img * L + l + 1) * E;
    const float* wr = w + (long)p * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(wr[e], ar[e], s);
    s = wave_sum(s);
    if (lane == 0) cam[(long)p * hw + l] = fmaxf(s, 0.f);
}

__global__ __launch_bounds__(1024) void cam_norm_kernel(float* __restrict__ cam, int hw) {
    extern __shared__ float sm[];   // cam[hw] | red[16]
    float* cs = sm;
    float* red = cs + hw;
    const int p = blockIdx.x, tid = threadIdx.x;
    for (int l = tid; l < hw; l += 1024) cs[l] = cam[(long)p * hw + l];
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        float mn = INFINITY;
        for (int l = tid; l < hw; l += 1024) mn = fminf(mn, cs[l]);
        mn = block_min(mn, red);
        float mx = -INFINITY;
        for (int l = tid; l < hw; l += 1024) mx = fmaxf(mx, cs[l] - mn);
        mx = block_max(mx, red);
        __syncthreads();
        for (int l = tid; l < hw; l += 1024) cs[l] = fmaxf((cs[l] - mn) / (1e-7f + mx), 0.f);
        __syncthreads();
    }
    for (int l = tid; l < hw; l += 1024) cam[(long)p * hw + l] = cs[l];
}

// ---------------------------------------------------------------------------------------------
extern "C" int wc_cam_head(const float* x2, const float* lnw, const float* lnb, const float* proj,
                           const float* text, const int* text_idx, const int* n_text, const int* pair_img,
                           const int* pair_cls, float logit_scale, float* partial, float* probs, float* df,
                           int B, int P, int L, int E, int Ed, int Tmax, void* stream) {
    WC_CHECK_ARG(x2 && lnw && lnb && proj && text && text_idx && n_text && pair_img && pair_cls && partial &&
                     probs && df && B > 0 && P > 0 && L > 1 && E > 0 && Ed > 0,
                 "wc_cam_head: bad argument");
    WC_CHECK_ARG(Tmax > 0 && Tmax <= 512 && E <= 4096 && Ed <= 4096, "wc_cam_head: Tmax <= 512, E, Ed <= 4096");
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = wc_cdiv(L, LNP_ROWS);
    WC_CHECK_ARG(E <= 1024, "wc_cam_head: E <= 1024");
    if (E <= 256)
        hipLaunchKernelGGL(lnpost_pool_kernel<4>, dim3(nchunk, B), dim3(256), 4 * E * sizeof(float), st, x2, lnw, lnb,
                           1e-5f, partial, L, E, nchunk);
    else
        hipLaunchKernelGGL(lnpost_pool_kernel<16>, dim3(nchunk, B), dim3(256), 4 * E * sizeof(float), st, x2, lnw, lnb,
                           1e-5f, partial, L, E, nchunk);
    WC_LAUNCH_CHECK("lnpost_pool_kernel");
    const size_t sm = (E + 2 * Ed + Tmax + 16) * sizeof(float);
    hipLaunchKernelGGL(cam_head_kernel, dim3(P), dim3(512), sm, st, partial, proj, text, text_idx, n_text,
                       pair_img, pair_cls, logit_scale, probs, df, L, E, Ed, nchunk, Tmax);
    WC_LAUNCH_CHECK("cam_head_kernel");
    return WC_OK;
}

extern "C" int wc_lnpost_bwd(const float* df, const float* x2, const float* lnw, float gs,
                             const int* pair_img, float* d32, void* dhi, void* dlo, int P, int L, int E,
                             void* stream) {
    WC_CHECK_ARG(df && x2 && lnw && pair_img && d32 && dhi && P > 0 && L > 1 && E > 0,
                 "wc_lnpost_bwd: bad argument");
    const long rows = (long)P * L;
#define LNPOST_BWD(NV_)                                                                                            \
    hipLaunchKernelGGL(lnpost_bwd_kernel<NV_>, dim3(wc_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, df, x2, \
                       lnw, 1e-5f, gs, pair_img, d32, (__half*)dhi, (__half*)dlo, L, E, rows)
    if (E <= 256) LNPOST_BWD(4);
    else if (E <= 768) LNPOST_BWD(12);
    else if (E <= 1024) LNPOST_BWD(16);
    else LNPOST_BWD(0);
#undef LNPOST_BWD
    WC_LAUNCH_CHECK("lnpost_bwd_kernel");
    return WC_OK;
}

extern "C" int wc_ln2_bwd_add(const float* da2, const float* dx2, const float* x1, const float* lnw, float gs,
                              const int* pair_img, void* g16, int P, int L, int E, void* stream) {
    WC_CHECK_ARG(da2 && dx2 && x1 && lnw && pair_img && g16 && P > 0 && L > 0 && E > 0 && gs > 0,
                 "wc_ln2_bwd_add: bad argument");
    const long rows = (long)P * L;
    WC_CHECK_ARG(E <= 1024, "wc_ln2_bwd_add: E <= 1024");
    if (E <= 256)
        hipLaunchKernelGGL(ln2_bwd_add_kernel<4>, dim3(wc_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, da2,
                           dx2, x1, lnw, 1e-5f, 1.0f / gs, pair_img, (__half*)g16, L, E, rows);
    else if (E <= 768)
        hipLaunchKernelGGL(ln2_bwd_add_kernel<12>, dim3(wc_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, da2,
                           dx2, x1, lnw, 1e-5f, 1.0f / gs, pair_img, (__half*)g16, L, E, rows);
    else
        hipLaunchKernelGGL(ln2_bwd_add_kernel<16>, dim3(wc_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, da2,
                           dx2, x1, lnw, 1e-5f, 1.0f / gs, pair_img, (__half*)g16, L, E, rows);
    WC_LAUNCH_CHECK("ln2_bwd_add_kernel");
    return WC_OK;
}

extern "C" int wc_attn_bwd_colsum(const void* qkv, const void* dO, const float* o32, const float* lse,
                                  const int* pair_img, float* delta, float* u, float* dS0, float* P0,
                                  float* c, int P, int L, int H, int DH, void* stream) {
    WC_CHECK_ARG(qkv && dO && o32 && lse && pair_img && delta && u && dS0 && P0 && c && P > 0 && L > 0 && H > 0,
                 "wc_attn_bwd_colsum: bad argument");
    WC_CHECK_ARG(DH == 64 || DH == 32, "wc_attn_bwd_colsum: head dim must be 32 or 64");
    WC_CHECK_ARG(P <= 65535 && H <= 65535, "wc_attn_bwd_colsum: too many pairs/heads for one launch");
    hipStream_t st = (hipStream_t)stream;
    const int E = H * DH;
    const long total = (long)P * L;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(wc_cdiv(total, 4)), dim3(256), 0, st, (const __half*)dO, o32,
                       pair_img, delta, L, H, DH, total);
    WC_LAUNCH_CHECK("attn_delta_kernel");
    const int rr = L % 128;
    const int origin = (L >= 128 && rr > 0 && rr <= 8) ? rr : 0;   // tiny remainder (CLS row): tiles start behind it
    const int nkt = wc_cdiv(L - origin, 128), PH = P * H;
    dim3 grid((unsigned)(nkt * ((PH + 7) / 8 * 8)));
    if (DH == 64) {
        const size_t lds = 2 * 128 * (64 * 2 + 16) + (2 * 128 * (64 * 2 + 16) + 1024) + 1024;
        hipLaunchKernelGGL(attn_bwd_colsum_kernel<64>, grid, dim3(256), lds, st, (const __half*)qkv,
                           (const __half*)dO, lse, delta, pair_img, u, dS0, P0, L, H, E, origin, nkt, PH);
        if (origin)
            hipLaunchKernelGGL(attn_bwd_colsum_edge_kernel<64>, dim3(H, P), dim3(256), 0, st, (const __half*)qkv,
                               (const __half*)dO, lse, delta, pair_img, u, dS0, P0, L, H, E, origin);
    } else {
        const size_t lds = 2 * 128 * (32 * 2 + 16) + (2 * 128 * (32 * 2 + 16) + 1024) + 1024;
        hipLaunchKernelGGL(attn_bwd_colsum_kernel<32>, grid, dim3(256), lds, st, (const __half*)qkv,
                           (const __half*)dO, lse, delta, pair_img, u, dS0, P0, L, H, E, origin, nkt, PH);
        if (origin)
            hipLaunchKernelGGL(attn_bwd_colsum_edge_kernel<32>, dim3(H, P), dim3(256), 0, st, (const __half*)qkv,
                               (const __half*)dO, lse, delta, pair_img, u, dS0, P0, L, H, E, origin);
    }
    WC_LAUNCH_CHECK("attn_bwd_colsum_kernel");
    hipLaunchKernelGGL(qkv_colsum_kernel, dim3(H, P), dim3(256), 0, st, (const __half*)qkv, (const __half*)dO, u,
                       dS0, P0, pair_img, c, L, H, DH, 1.0f / sqrtf((float)DH),
                       1.0f / 1.4426950408889634f);
    WC_LAUNCH_CHECK("qkv_colsum_kernel");
    return WC_OK;
}

// out[p, e] = scale * sum_n c[p, n] * W[n, e]   (plain fp32 FMA: P is tiny, values span ~1e-9..1e-3)
// block = 64 columns x 4 row-slices of n (LDS reduce); all pairs of a PB-chunk share each W read; the n
// range is additionally split over blockIdx.z into `ns` slices (partials summed by a second kernel) so the
// launch has enough workgroups to fill the chip.
#define RVM_PB 8
__global__ __launch_bounds__(256) void rowvec_matmul_kernel(const float* __restrict__ c,
                                                             const float* __restrict__ W, float* __restrict__ part,
                                                             int P, int N, int E, int nper) {
    __shared__ float red[4][RVM_PB][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane, p0 = blockIdx.y * RVM_PB;
    const int nbeg = blockIdx.z * nper;
    int nend = nbeg + nper;
    if (nend > N) nend = N;
    float acc[RVM_PB];
#pragma unroll
    for (int k = 0; k < RVM_PB; ++k) acc[k] = 0.f;
    if (e < E) {
        // rows past P read row P - 1 and are dropped at the end: a bounds branch per c load made hipcc wait for each of the 32
        // loads of an iteration before issuing the next one
        long crow[RVM_PB];
#pragma unroll
        for (int k = 0; k < RVM_PB; ++k) crow[k] = (long)min(p0 + k, P - 1) * N;
        int n = nbeg + wv;
        for (; n + 12 < nend; n += 16) {              // four W rows in flight per wave (one per iteration: latency bound)
            float w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = W[(long)(n + 4 * u) * E + e];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < RVM_PB; ++k) acc[k] = fmaf(c[crow[k] + n + 4 * u], w4[u], acc[k]);
        }
        for (; n < nend; n += 4) {
            const float w = W[(long)n * E + e];
#pragma unroll
            for (int k = 0; k < RVM_PB; ++k) acc[k] = fmaf(c[crow[k] + n], w, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < RVM_PB; ++k) red[wv][k][lane] = acc[k];
    __syncthreads();
    if (wv == 0 && e < E)
        for (int k = 0; k < RVM_PB && p0 + k < P; ++k)
            part[((long)blockIdx.z * P + p0 + k) * E + e] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
}

__global__ __launch_bounds__(256) void rowvec_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int ns,
                                                          long n, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < ns; ++k) s += part[(long)k * n + i];
    out[i] = s * scale;
}

// ws: workspace of 16 * P * E floats.
extern "C" int wc_rowvec_matmul(const float* c, const float* W, float* out, float* ws, int P, int N, int E, float scale,
                                void* stream) {
    WC_CHECK_ARG(c && W && out && ws && P > 0 && N > 0 && E > 0 && P <= 65535 * RVM_PB, "wc_rowvec_matmul: bad argument");
    const int ns = 16, nper = wc_cdiv(N, ns);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowvec_matmul_kernel, dim3(wc_cdiv(E, 64), wc_cdiv(P, RVM_PB), ns), dim3(256), 0, st, c, W, ws, P, N, E,
                       nper);
    WC_LAUNCH_CHECK("rowvec_matmul_kernel");
    hipLaunchKernelGGL(rowvec_sum_kernel, dim3(wc_cdiv((long)P * E, 256)), dim3(256), 0, st, ws, out, ns, (long)P * E, scale);
    WC_LAUNCH_CHECK("rowvec_sum_kernel");
    return WC_OK;
}

extern "C" int wc_cam_map(const float* a32, const float* w, const int* pair_img, float* cam, int P, int L,
                          int E, void* stream) {
    WC_CHECK_ARG(a32 && w && pair_img && cam && P > 0 && L > 1 && E > 0, "wc_cam_map: bad argument");
    const size_t sm = ((L - 1) + 16) * sizeof(float);
    WC_CHECK_ARG(sm <= 160 * 1024 && P <= 65535, "wc_cam_map: token grid too large for LDS");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cam_raw_kernel, dim3(wc_cdiv(L - 1, 16), P), dim3(1024), 0, st, a32, w, pair_img, cam, L, E);
    WC_LAUNCH_CHECK("cam_raw_kernel");
    hipLaunchKernelGGL(cam_norm_kernel, dim3(P), dim3(1024), sm, st, cam, L - 1);
    WC_LAUNCH_CHECK("cam_norm_kernel");
    return WC_OK;
}
