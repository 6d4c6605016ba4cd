// Backward-pass helpers for the trainable adapters / decoder (reference autograd through
// WeCLIP_model/segformer_head.py:69-80, WeCLIP_model/Decoder/TransDecoder.py:63-125 and
// model_attn_aff_voc.py:134-137): operand transposes for weight-gradient GEMMs, bias-gradient
// column sums, LayerNorm backward with parameter gradients, and the sigmoid-Gram backward.
#include "common.h"

// dst[b? .. ] : out[c, coff_b + r] = fp16(scale * src[b, r, c]);  src rows r < R of batch b start at
// src + b*sSrc with row stride ld; out row stride ldo (>= batch*R, zero padded by the caller).
template <typename T>
__global__ __launch_bounds__(256) void transpose_f16_kernel(const T* __restrict__ src, long ld, long sSrc,
                                                             __half* __restrict__ hi, __half* __restrict__ lo,
                                                             long ldo, long oR, int R, int C, float scale) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const T* s = src + (long)b * sSrc;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? (float)s[(long)(r0 + r) * ld + c0 + c] * scale : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < C && r0 + r < R) {
            const float v = tile[r][c];
            const __half h = __float2half(v);
            const long o = (long)(c0 + c) * ldo + (long)b * oR + r0 + r;
            hi[o] = h;
            if (lo) lo[o] = __float2half(v - __half2float(h));
        }
    }
}

// out[c] = alpha * sum_r src[r, c]  (optionally rounded through fp16); two-stage via partial sums.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ src, long ld, float* __restrict__ part,
                                                              long R, int C, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > R) r1 = R;
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += (float)src[r * ld + c];
    part[(long)blockIdx.y * C + c] = s;
}
// out2 (optional): columns [C / 2, C) go to out2[0 .. C / 2) instead of out (two destinations of one partial matrix)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             int nblk, int C, float alpha, int round16, float* __restrict__ out2 = nullptr) {
    // 64 columns per block, the partial rows dealt over 16 waves (256-B coalesced reads, 4 loads in flight each): the grid
    // is only C/64 blocks, so the block is as wide as it gets (4 waves: 22 us for 1024 partial rows of 512 columns)
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int b = rg;
        for (; b + 48 < nblk; b += 64) {
            s0 += part[(long)b * C + c];
            s1 += part[(long)(b + 16) * C + c];
            s2 += part[(long)(b + 32) * C + c];
            s3 += part[(long)(b + 48) * C + c];
        }
        for (; b < nblk; b += 16) s0 += part[(long)b * C + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        s *= alpha;
        float* o = (out2 && c >= C / 2) ? out2 + (c - C / 2) : out + c;
        *o = round16 ? __half2float(__float2half(s)) : s;
    }
}

// LayerNorm backward.  dx = add + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*w.
// Outputs: dx32 (optional), dx16 = fp16(dx * out_scale) (optional); per-block partial sums of
// dgamma = sum dy*xhat and dbeta = sum dy in part (nblk, 2, D).  One wave per row, LNB_ROWS rows per wave
// (few rows per wave = many waves: the three dependent wave reductions per row are latency, hidden by occupancy).
#define LNB_ROWS 4
// TWO: two LayerNorms of the SAME input x (different affine parameters: in CTI, LN(c1) feeds the values of one attention and the
// queries of the other) back-propagated in one pass: with g = dy_a * gamma_a + dy_b * gamma_b the input gradient is the ordinary
// formula in g (both share xhat), the four parameter gradients are separate column sums (part: (nblk, 4, D)).  One read of x,
// one write of dx instead of two passes with the first one's result re-read as `add` by the second.
template <int NV, bool TWO = false>   // D <= 64*NV
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const __half* __restrict__ dy16,
                                                      const float* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ add,
                                                      float eps, float* __restrict__ dx32, __half* __restrict__ dx16,
                                                      float out_scale, float* __restrict__ part, long rows, int D, long ngroups,
                                                      const __half* __restrict__ dy16b = nullptr, const float* __restrict__ wb = nullptr) {
    extern __shared__ float sm[];   // [4][2 or 4][D]
    constexpr int NP = TWO ? 4 : 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float ag[NV], ab[NV], ag2[TWO ? NV : 1], ab2[TWO ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) ag[i] = ab[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (TWO ? NV : 1); ++i) ag2[i] = ab2[i] = 0.f;
    // slot i of a lane = element ((i >> 2) * 64 + lane) * 4 + (i & 3): four consecutive elements per lane, 16-byte IO
    // (D % 4 == 0, so a group of four slots is inside the row or outside it as a whole)
#define LN_E(i_) ((((i_) >> 2) * 64 + lane) * 4 + ((i_) & 3))
    // A block walks its groups of 16 rows (grid-stride): the dgamma / dbeta partial sums stay in registers over all of them, so
    // the partial-sum matrix the final column reduction reads has gridDim.x rows instead of rows / 16 (86 016 rows: 11 MB ->
    // 2 MB; colsum_final 19.8 -> ~6 us).  dy may arrive as fp16 (dy16: the producing GEMM then writes and this kernel reads
    // half the bytes).
    float wv_[NV];              // gamma: the same for every group of the block
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const int e = LN_E(i);
        const float4 wa = *reinterpret_cast<const float4*>(w + (e < D ? e : 0));
        wv_[i] = e < D ? wa.x : 0.f; wv_[i + 1] = e < D ? wa.y : 0.f; wv_[i + 2] = e < D ? wa.z : 0.f; wv_[i + 3] = e < D ? wa.w : 0.f;
    }
    float wvb_[TWO ? NV : 1];
    if constexpr (TWO) {
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const int e = LN_E(i);
            const float4 wa = *reinterpret_cast<const float4*>(wb + (e < D ? e : 0));
            wvb_[i] = e < D ? wa.x : 0.f; wvb_[i + 1] = e < D ? wa.y : 0.f; wvb_[i + 2] = e < D ? wa.z : 0.f; wvb_[i + 3] = e < D ? wa.w : 0.f;
        }
    }
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    // the LNB_ROWS rows of a wave are processed TOGETHER: all their loads in flight at once and their four reduction
    // chains interleaved (the row-after-row loop exposed one load latency and four shuffle chains per row: 29 us for
    // 16 384 x 256 at two waves per SIMD)
    const long r0 = (grp * 4 + wv) * LNB_ROWS;
    float xv[LNB_ROWS][NV], dv[LNB_ROWS][NV], s[LNB_ROWS];
    bool live[LNB_ROWS];
    // Every load of the group is issued UNCONDITIONALLY from a clamped address and masked afterwards (behind a bounds branch hipcc
    // waits for each load before it issues the next one); the fp16 / fp32 form of dy is ONE uniform branch around all of its loads;
    // for rows of up to 256 values the residual gradient `add` is requested here as well instead of after the reduction chains.
    constexpr bool HOIST = NV <= 4;
    float4 av[LNB_ROWS][HOIST ? NV / 4 : 1];
    long rowc[LNB_ROWS];
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        live[r] = r0 + r < rows;
        rowc[r] = live[r] ? r0 + r : rows - 1;
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const int e = LN_E(i), ec = e < D ? e : 0;
            const float4 xa = *reinterpret_cast<const float4*>(x + rowc[r] * D + ec);
            xv[r][i] = xa.x; xv[r][i + 1] = xa.y; xv[r][i + 2] = xa.z; xv[r][i + 3] = xa.w;
            if constexpr (HOIST) {
                if (add) av[r][i >> 2] = *reinterpret_cast<const float4*>(add + rowc[r] * D + ec);
            }
        }
    }
    if (dy16) {
#pragma unroll
        for (int r = 0; r < LNB_ROWS; ++r)
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                const int e = LN_E(i), ec = e < D ? e : 0;
                const uint2 hq = *reinterpret_cast<const uint2*>(dy16 + rowc[r] * D + ec);
                const __half* hp = reinterpret_cast<const __half*>(&hq);
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[r][i + k] = __half2float(hp[k]);
            }
    } else {
#pragma unroll
        for (int r = 0; r < LNB_ROWS; ++r)
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                const int e = LN_E(i), ec = e < D ? e : 0;
                const float4 da = *reinterpret_cast<const float4*>(dy + rowc[r] * D + ec);
                dv[r][i] = da.x; dv[r][i + 1] = da.y; dv[r][i + 2] = da.z; dv[r][i + 3] = da.w;
            }
    }
    float dvb[TWO ? LNB_ROWS : 1][TWO ? NV : 1];
    if constexpr (TWO) {
#pragma unroll
        for (int r = 0; r < LNB_ROWS; ++r)
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                const int e = LN_E(i), ec = e < D ? e : 0;
                const uint2 hq = *reinterpret_cast<const uint2*>(dy16b + rowc[r] * D + ec);
                const __half* hp = reinterpret_cast<const __half*>(&hq);
#pragma unroll
                for (int k = 0; k < 4; ++k) dvb[r][i + k] = __half2float(hp[k]);      // (masked below, with dv: no select around the load)
            }
    }
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const bool in = LN_E(i) < D;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xv[r][i + k] = in ? xv[r][i + k] : 0.f;
                dv[r][i + k] = (in && live[r]) ? dv[r][i + k] : 0.f;
                if constexpr (TWO) dvb[r][i + k] = (in && live[r]) ? dvb[r][i + k] : 0.f;
            }
            s[r] += (xv[r][i] + xv[r][i + 1]) + (xv[r][i + 2] + xv[r][i + 3]);
        }
    }
    float mean[LNB_ROWS], rstd[LNB_ROWS], sg[LNB_ROWS], sgx[LNB_ROWS];
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) mean[r] = wave_sum(s[r]) / D;
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float d = LN_E(i) < D ? xv[r][i] - mean[r] : 0.f;
            q += d * d;
        }
        s[r] = q;
    }
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) rstd[r] = rsqrtf(wave_sum(s[r]) / D + eps);
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (LN_E(i) < D) {
                const float xh = (xv[r][i] - mean[r]) * rstd[r];
                float g = dv[r][i] * wv_[i];
                ag[i] += dv[r][i] * xh;           // a dead row has dv = 0
                ab[i] += dv[r][i];
                if constexpr (TWO) {
                    g = fmaf(dvb[r][i], wvb_[i], g);
                    ag2[i] += dvb[r][i] * xh;
                    ab2[i] += dvb[r][i];
                    dv[r][i] = g;                 // (from here on dv holds g = sum of dy * gamma)
                }
                a += g;
                bsum += g * xh;
            }
        }
        sg[r] = a;
        sgx[r] = bsum;
    }
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        sg[r] = wave_sum(sg[r]) / D;
        sgx[r] = wave_sum(sgx[r]) / D;
    }
    // the finished values of all rows first, then nothing but stores: a load (or anything the compiler must wait for with vmcnt)
    // between the stores of two rows would wait for the earlier row's stores as well -- loads and stores share the counter
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r)
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            float4 aa = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add) {
                if constexpr (HOIST) aa = av[r][i >> 2];
                else aa = *reinterpret_cast<const float4*>(add + rowc[r] * D + (LN_E(i) < D ? LN_E(i) : 0));
            }
            const float a4[4] = {aa.x, aa.y, aa.z, aa.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xv[r][i + k] - mean[r]) * rstd[r];
                dv[r][i + k] = rstd[r] * ((TWO ? dv[r][i + k] : dv[r][i + k] * wv_[i + k]) - sg[r] - xh * sgx[r]) + a4[k];
            }
        }
#pragma unroll
    for (int r = 0; r < LNB_ROWS; ++r) {
        if (!live[r]) continue;
        const long row = r0 + r;
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const int e = LN_E(i);
            if (e < D) {
                if (dx32) *reinterpret_cast<float4*>(dx32 + row * D + e) = make_float4(dv[r][i], dv[r][i + 1], dv[r][i + 2], dv[r][i + 3]);
                if (dx16) {
                    __half hv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hv[k] = __float2half(dv[r][i + k] * out_scale);
                    *reinterpret_cast<uint2*>(dx16 + row * D + e) = *reinterpret_cast<const uint2*>(hv);
                }
            }
        }
    }
    }      // groups of this block
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = LN_E(i);
        if (e < D) {
            sm[(wv * NP + 0) * D + e] = ag[i];
            sm[(wv * NP + 1) * D + e] = ab[i];
            if constexpr (TWO) {
                sm[(wv * NP + 2) * D + e] = ag2[i];
                sm[(wv * NP + 3) * D + e] = ab2[i];
            }
        }
    }
#undef LN_E
    __syncthreads();
    for (int e = threadIdx.x; e < NP * D; e += 256) {
        const int which = e / D, k = e - which * D;
        part[((long)blockIdx.x * NP + which) * D + k] = sm[(0 * NP + which) * D + k] + sm[(1 * NP + which) * D + k] +
                                                         sm[(2 * NP + which) * D + k] + sm[(3 * NP + which) * D + k];
    }
}

// S[b,i,j] = z(i,j) + z(j,i),  z = dAP * AP * (1 - AP)   (backward of sigmoid(F^T F) w.r.t. the Gram matrix,
// symmetrised so that dF = S F); fp16 hi/lo operands for the GEMM.
// One block per unordered pair of 64x64 tiles (ti <= tj): z is computed once per element, both tiles are read
// coalesced (rows), and the transposed partner comes from LDS.
__global__ __launch_bounds__(256) void sigmoid_gram_bwd_kernel(const float* __restrict__ dAP, const float* __restrict__ AP,
                                                                __half* __restrict__ hi, __half* __restrict__ lo, int n,
                                                                int ldo, float scale, int nt) {
    __shared__ float za[64][65], zb[64][65];      // z of tile (ti, tj) and of tile (tj, ti)
    // blockIdx.x enumerates pairs ti <= tj
    int ti = 0, rem = blockIdx.x;
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem, b = blockIdx.y;
    const long base = (long)b * n * n;
    const int i0 = ti * 64, j0 = tj * 64;
    const bool vec = (n & 3) == 0 && (ldo & 3) == 0 && (((uintptr_t)dAP | (uintptr_t)AP) & 15) == 0 && ((uintptr_t)hi & 7) == 0 &&
                     (!lo || ((uintptr_t)lo & 7) == 0);
    if (vec) {            // 16-byte loads, 8-byte stores: four columns per thread
        // the sixteen 16-byte loads of a thread (4 passes x {AP, dAP} x {tile, transposed tile}) are issued together from clamped
        // addresses and masked afterwards: behind their bounds branches each pair was waited for before the next was requested
        float4 a4[4], d4[4], at4[4], dt4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * 256 + threadIdx.x, r = e >> 4, c = (e & 15) * 4;
            const long o = base + (long)min(i0 + r, n - 1) * n + min(j0 + c, n - 4);
            const long ot = base + (long)min(j0 + r, n - 1) * n + min(i0 + c, n - 4);
            a4[q] = *reinterpret_cast<const float4*>(AP + o);
            d4[q] = *reinterpret_cast<const float4*>(dAP + o);
            at4[q] = *reinterpret_cast<const float4*>(AP + ot);
            dt4[q] = *reinterpret_cast<const float4*>(dAP + ot);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * 256 + threadIdx.x, r = e >> 4, c = (e & 15) * 4;
            const float m = (i0 + r < n && j0 + c < n) ? 1.f : 0.f, mt = (ti != tj && j0 + r < n && i0 + c < n) ? 1.f : 0.f;
            const float4 a = a4[q], d = d4[q], at = at4[q], dt = dt4[q];
            za[r][c] = m * (d.x * a.x * (1.f - a.x)); za[r][c + 1] = m * (d.y * a.y * (1.f - a.y));
            za[r][c + 2] = m * (d.z * a.z * (1.f - a.z)); za[r][c + 3] = m * (d.w * a.w * (1.f - a.w));
            zb[r][c] = mt * (dt.x * at.x * (1.f - at.x)); zb[r][c + 1] = mt * (dt.y * at.y * (1.f - at.y));
            zb[r][c + 2] = mt * (dt.z * at.z * (1.f - at.z)); zb[r][c + 3] = mt * (dt.w * at.w * (1.f - at.w));
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, c = (e & 15) * 4;
            if (i0 + r < n && j0 + c < n) {
                __half h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = scale * (za[r][c + k] + (ti == tj ? za[c + k][r] : zb[c + k][r]));
                    h[k] = __float2half(v);
                    l[k] = __float2half(v - __half2float(h[k]));
                }
                const long oo = (long)b * n * ldo + (long)(i0 + r) * ldo + j0 + c;
                *reinterpret_cast<uint2*>(hi + oo) = *reinterpret_cast<const uint2*>(h);
                if (lo) *reinterpret_cast<uint2*>(lo + oo) = *reinterpret_cast<const uint2*>(l);
            }
            if (ti != tj && j0 + r < n && i0 + c < n) {
                __half h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = scale * (zb[r][c + k] + za[c + k][r]);
                    h[k] = __float2half(v);
                    l[k] = __float2half(v - __half2float(h[k]));
                }
                const long oo = (long)b * n * ldo + (long)(j0 + r) * ldo + i0 + c;
                *reinterpret_cast<uint2*>(hi + oo) = *reinterpret_cast<const uint2*>(h);
                if (lo) *reinterpret_cast<uint2*>(lo + oo) = *reinterpret_cast<const uint2*>(l);
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        float v = 0.f, vt = 0.f;
        if (i0 + r < n && j0 + c < n) {
            const long o = base + (long)(i0 + r) * n + j0 + c;
            const float a = AP[o];
            v = dAP[o] * a * (1.f - a);
        }
        if (ti != tj && j0 + r < n && i0 + c < n) {
            const long o = base + (long)(j0 + r) * n + i0 + c;
            const float a = AP[o];
            vt = dAP[o] * a * (1.f - a);
        }
        za[r][c] = v;
        zb[r][c] = vt;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        // S(i0+r, j0+c) = z(i0+r, j0+c) + z(j0+c, i0+r)
        if (i0 + r < n && j0 + c < n) {
            const float v = scale * (za[r][c] + (ti == tj ? za[c][r] : zb[c][r]));
            const __half h = __float2half(v);
            const long oo = (long)b * n * ldo + (long)(i0 + r) * ldo + j0 + c;
            hi[oo] = h;
            if (lo) lo[oo] = __float2half(v - __half2float(h));
        }
        if (ti != tj && j0 + r < n && i0 + c < n) {
            const float v = scale * (zb[r][c] + za[c][r]);
            const __half h = __float2half(v);
            const long oo = (long)b * n * ldo + (long)(j0 + r) * ldo + i0 + c;
            hi[oo] = h;
            if (lo) lo[oo] = __float2half(v - __half2float(h));
        }
    }
}

// out32[r,c] = x[r,c] * cs[(r / rpb), c];  hi/lo = fp16 split of it   (dropout-mask backward + operand split)
// V = 4: four consecutive values per thread (C % 4 == 0, 16-byte aligned buffers): 16-byte loads, 8-byte fp16 stores -- the
// one-value-per-thread form issued 22 M threads for an 86 016 x 256 operand and ran at a third of the HBM rate.
template <int V>
__global__ __launch_bounds__(256) void colscale_split_kernel(const float* __restrict__ x, const float* __restrict__ cs,
                                                              float* __restrict__ out32, __half* __restrict__ hi,
                                                              __half* __restrict__ lo, long rows, int C, int rpb,
                                                              float alpha) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= rows * C) return;
    const long r = i / C;
    const int c = i - r * C;
    float v[V];
    if constexpr (V == 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + i);
        const float4 sv = cs ? *reinterpret_cast<const float4*>(cs + (r / rpb) * C + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        v[0] = alpha * xv.x * sv.x; v[1] = alpha * xv.y * sv.y; v[2] = alpha * xv.z * sv.z; v[3] = alpha * xv.w * sv.w;
        if (out32) *reinterpret_cast<float4*>(out32 + i) = make_float4(v[0], v[1], v[2], v[3]);
        __half h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = __float2half(v[k]);
            l[k] = __float2half(v[k] - __half2float(h[k]));
        }
        *reinterpret_cast<uint2*>(hi + i) = *reinterpret_cast<const uint2*>(h);
        if (lo) *reinterpret_cast<uint2*>(lo + i) = *reinterpret_cast<const uint2*>(l);
    } else {
        v[0] = alpha * x[i] * (cs ? cs[(r / rpb) * C + c] : 1.f);
        if (out32) out32[i] = v[0];
        const __half h = __float2half(v[0]);
        hi[i] = h;
        if (lo) lo[i] = __float2half(v[0] - __half2float(h));
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int wc_transpose_f16(const void* src, int src_f32, long ld, long sSrc, void* hi, void* lo, long ldo,
                                long oR, int batch, int R, int C, float scale, void* stream) {
    WC_CHECK_ARG(src && hi && batch > 0 && R > 0 && C > 0 && ld >= C && oR >= R && ldo >= (long)(batch - 1) * oR + R &&
                     batch <= 65535,
                 "wc_transpose_f16: bad argument");
    dim3 grid(wc_cdiv(R, 64), wc_cdiv(C, 64), batch);
    if (src_f32)
        hipLaunchKernelGGL(transpose_f16_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, ld,
                           sSrc, (__half*)hi, (__half*)lo, ldo, oR, R, C, scale);
    else
        hipLaunchKernelGGL(transpose_f16_kernel<__half>, grid, dim3(256), 0, (hipStream_t)stream, (const __half*)src,
                           ld, sSrc, (__half*)hi, (__half*)lo, ldo, oR, R, C, scale);
    WC_LAUNCH_CHECK("transpose_f16_kernel");
    return WC_OK;
}

extern "C" int wc_colsum(const void* src, int src_f32, long ld, float* part, float* out, long R, int C, float alpha,
                         int round16, void* stream) {
    WC_CHECK_ARG(src && part && out && R > 0 && C > 0 && ld >= C, "wc_colsum: bad argument");
    const int rpb = 256;
    const int nblk = wc_cdiv(R, rpb);
    WC_CHECK_ARG(nblk <= 65535, "wc_colsum: too many rows");
    dim3 grid(wc_cdiv(C, 256), nblk);
    hipStream_t st = (hipStream_t)stream;
    if (src_f32)
        hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)src, ld, part, R, C, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<__half>, grid, dim3(256), 0, st, (const __half*)src, ld, part, R, C, rpb);
    WC_LAUNCH_CHECK("colsum_partial_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(wc_cdiv(C, 64)), dim3(1024), 0, st, part, out, nblk, C, alpha, round16);
    WC_LAUNCH_CHECK("colsum_final_kernel");
    return WC_OK;
}

// part: workspace ceil(rows/16)*2*D floats; dgb (2,D) = alpha * [dgamma ; dbeta].
static int layernorm_bwd_impl(const float* dy, const void* dy16, const float* x, const float* w, const float* add, float eps,
                              float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                              long rows, int D, void* stream);

extern "C" int wc_layernorm_bwd(const float* dy, const float* x, const float* w, const float* add, float eps,
                                float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                                long rows, int D, void* stream) {
    WC_CHECK_ARG(dy, "wc_layernorm_bwd: bad argument (dy)");
    return layernorm_bwd_impl(dy, nullptr, x, w, add, eps, dx32, dx16, out_scale, part, dgb, alpha, rows, D, stream);
}

// the same with the incoming gradient as fp16 rows (dy16 (rows, D)): the producing GEMM writes, and this kernel reads, half the bytes
extern "C" int wc_layernorm_bwd_h(const void* dy16, const float* x, const float* w, const float* add, float eps,
                                  float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                                  long rows, int D, void* stream) {
    WC_CHECK_ARG(dy16 && (uintptr_t)dy16 % 8 == 0, "wc_layernorm_bwd_h: bad argument (dy16)");
    return layernorm_bwd_impl(nullptr, dy16, x, w, add, eps, dx32, dx16, out_scale, part, dgb, alpha, rows, D, stream);
}

// two LayerNorms of one input (fp16 gradients dya16 / dyb16, gammas wa / wb): dx = LN_bwd_a(dya) + LN_bwd_b(dyb) [+ add];
// part: >= min(ceil(rows / 16), 2048) * 4 * D floats; dgba / dgbb: (2, D) each = alpha * [dgamma; dbeta].  D <= 256.
extern "C" int wc_layernorm_bwd2_h(const void* dya16, const float* wa, const void* dyb16, const float* wb, const float* x,
                                   const float* add, float eps, float* dx32, void* dx16, float out_scale, float* part, float* dgba,
                                   float* dgbb, float alpha, long rows, int D, void* stream) {
    WC_CHECK_ARG(dya16 && dyb16 && wa && wb && x && part && dgba && dgbb && rows > 0 && D > 0 && D <= 256 && D % 4 == 0 && (dx32 || dx16),
                 "wc_layernorm_bwd2_h: bad argument (D <= 256, D %% 4 == 0)");
    WC_CHECK_ARG(((uintptr_t)x | (uintptr_t)wa | (uintptr_t)wb | (uintptr_t)add | (uintptr_t)dx32) % 16 == 0 &&
                     ((uintptr_t)dx16 | (uintptr_t)dya16 | (uintptr_t)dyb16) % 8 == 0,
                 "wc_layernorm_bwd2_h: operands must be 16-byte (fp16: 8-byte) aligned");
    const long ngroups = (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS);
    const int nblk = ngroups < 768 ? (int)ngroups : 768;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((ln_bwd_kernel<4, true>), dim3(nblk), dim3(256), 16 * (size_t)D * sizeof(float), st, (const float*)nullptr,
                       (const __half*)dya16, x, wa, add, eps, dx32, (__half*)dx16, out_scale, part, rows, D, ngroups,
                       (const __half*)dyb16, wb);
    WC_LAUNCH_CHECK("ln_bwd_kernel<two>");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(wc_cdiv(4 * D, 64)), dim3(1024), 0, st, part, dgba, nblk, 4 * D, alpha, 0, dgbb);
    WC_LAUNCH_CHECK("colsum_final_kernel");
    return WC_OK;
}

static int layernorm_bwd_impl(const float* dy, const void* dy16, const float* x, const float* w, const float* add, float eps,
                              float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                              long rows, int D, void* stream) {
    WC_CHECK_ARG(x && w && part && dgb && rows > 0 && D > 0 && D <= 1024 && D % 4 == 0 && (dx32 || dx16),
                 "wc_layernorm_bwd: bad argument (D <= 1024, D %% 4 == 0)");
    WC_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)w | (uintptr_t)add | (uintptr_t)dx32) % 16 == 0 && (uintptr_t)dx16 % 8 == 0,
                 "wc_layernorm_bwd: operands must be 16-byte aligned");
    const long ngroups = (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS);
    const int nblk = ngroups < 768 ? (int)ngroups : 768;        // (three resident blocks per CU: the partial matrix the column
                                                                //  reduction reads is 768 rows, not 2048 -- 9.9 us per launch for 8 workgroups)
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = 8 * (size_t)D * sizeof(float);
    if (D <= 256)
        hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nblk), dim3(256), sm, st, dy, (const __half*)dy16, x, w, add, eps, dx32, (__half*)dx16,
                           out_scale, part, rows, D, ngroups);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<16>, dim3(nblk), dim3(256), sm, st, dy, (const __half*)dy16, x, w, add, eps, dx32, (__half*)dx16,
                           out_scale, part, rows, D, ngroups);
    WC_LAUNCH_CHECK("ln_bwd_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(wc_cdiv(2 * D, 64)), dim3(1024), 0, st, part, dgb, nblk, 2 * D, alpha, 0);
    WC_LAUNCH_CHECK("colsum_final_kernel");
    return WC_OK;
}

extern "C" int wc_sigmoid_gram_bwd(const float* dAP, const float* AP, void* hi, void* lo, int B, int n, int ldo,
                                   float scale, void* stream) {
    WC_CHECK_ARG(dAP && AP && hi && B > 0 && n > 0 && n <= 65535 && ldo >= n, "wc_sigmoid_gram_bwd: bad argument");
    const int nt = wc_cdiv(n, 64);
    hipLaunchKernelGGL(sigmoid_gram_bwd_kernel, dim3(nt * (nt + 1) / 2, B), dim3(256), 0, (hipStream_t)stream, dAP, AP,
                       (__half*)hi, (__half*)lo, n, ldo, scale, nt);
    WC_LAUNCH_CHECK("sigmoid_gram_bwd_kernel");
    return WC_OK;
}

extern "C" int wc_colscale_split(const float* x, const float* cs, float* out32, void* hi, void* lo, long rows, int C,
                                 int rows_per_batch, float alpha, void* stream) {
    WC_CHECK_ARG(x && hi && rows > 0 && C > 0 && rows_per_batch > 0, "wc_colscale_split: bad argument");
    const bool vec = C % 4 == 0 && ((uintptr_t)x | (uintptr_t)cs | (uintptr_t)out32) % 16 == 0 && ((uintptr_t)hi | (uintptr_t)lo) % 8 == 0;
    if (vec)
        hipLaunchKernelGGL(colscale_split_kernel<4>, dim3(wc_cdiv(rows * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, cs,
                           out32, (__half*)hi, (__half*)lo, rows, C, rows_per_batch, alpha);
    else
        hipLaunchKernelGGL(colscale_split_kernel<1>, dim3(wc_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, cs,
                           out32, (__half*)hi, (__half*)lo, rows, C, rows_per_batch, alpha);
    WC_LAUNCH_CHECK("colscale_split_kernel");
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// All trainable weight matrices of the head -> fp16 MFMA operands in ONE launch per step: row-major (the
// forward's W, K-contiguous) and transposed (the backward's W^T for dX = dY W).  Replaces the per-layer
// `.half()` / `.t().contiguous()` conversions (reference: implicit in F.linear and its autograd).
// table: count rows of 8 int64 = {src f32 (R,C), hi (R,C), lo or 0, hiT (C,ldT), loT or 0, R, C, ldT}.
__global__ __launch_bounds__(256) void convert_weights_kernel(const long* __restrict__ table, int count) {
    __shared__ float tile[64][65];
    const long* e = table + (long)blockIdx.y * 8;
    const float* src = reinterpret_cast<const float*>(e[0]);
    __half* hi = reinterpret_cast<__half*>(e[1]);
    __half* lo = reinterpret_cast<__half*>(e[2]);
    __half* hiT = reinterpret_cast<__half*>(e[3]);
    __half* loT = reinterpret_cast<__half*>(e[4]);
    const int R = (int)e[5], C = (int)e[6];
    const long ldT = e[7];
    const int tc = (C + 63) / 64, tr = (R + 63) / 64;
    // 16-byte loads / 8-byte stores when the rows allow it (C % 4 == 0 for the row-major outputs, ldT % 4 == 0 for the
    // transposed ones; every 2-D weight of the heads does), element-wise otherwise (bias vectors, odd shapes)
    const bool vrow = (C & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)hi & 7) == 0 && (!lo || ((uintptr_t)lo & 7) == 0);
    const bool vcol = hiT && (ldT & 3) == 0 && ((uintptr_t)hiT & 7) == 0 && (!loT || ((uintptr_t)loT & 7) == 0);
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
        if (vrow) {
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int r = i >> 4, c = (i & 15) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 + r < R && c0 + c < C) {
                    const long o = (long)(r0 + r) * C + c0 + c;
                    v = *reinterpret_cast<const float4*>(src + o);
                    const float f[4] = {v.x, v.y, v.z, v.w};
                    __half h[4], l[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        h[k] = __float2half(f[k]);
                        l[k] = __float2half(f[k] - __half2float(h[k]));
                    }
                    *reinterpret_cast<uint2*>(hi + o) = *reinterpret_cast<const uint2*>(h);
                    if (lo) *reinterpret_cast<uint2*>(lo + o) = *reinterpret_cast<const uint2*>(l);
                }
                tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
            }
        } else {
            for (int i = threadIdx.x; i < 64 * 64; i += 256) {
                const int r = i >> 6, c = i & 63;
                float v = 0.f;
                if (r0 + r < R && c0 + c < C) {
                    v = src[(long)(r0 + r) * C + c0 + c];
                    const __half h = __float2half(v);
                    hi[(long)(r0 + r) * C + c0 + c] = h;
                    if (lo) lo[(long)(r0 + r) * C + c0 + c] = __float2half(v - __half2float(h));
                }
                tile[r][c] = v;
            }
        }
        __syncthreads();
        if (hiT) {
            if (vcol && r0 + 64 <= R) {
                for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                    const int c = i >> 4, r = (i & 15) * 4;
                    if (c0 + c < C) {
                        __half h[4], l[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float v = tile[r + k][c];
                            h[k] = __float2half(v);
                            l[k] = __float2half(v - __half2float(h[k]));
                        }
                        const long o = (long)(c0 + c) * ldT + r0 + r;
                        *reinterpret_cast<uint2*>(hiT + o) = *reinterpret_cast<const uint2*>(h);
                        if (loT) *reinterpret_cast<uint2*>(loT + o) = *reinterpret_cast<const uint2*>(l);
                    }
                }
            } else {
                for (int i = threadIdx.x; i < 64 * 64; i += 256) {
                    const int c = i >> 6, r = i & 63;
                    if (c0 + c < C && r0 + r < R) {
                        const float v = tile[r][c];
                        const __half h = __float2half(v);
                        hiT[(long)(c0 + c) * ldT + r0 + r] = h;
                        if (loT) loT[(long)(c0 + c) * ldT + r0 + r] = __float2half(v - __half2float(h));
                    }
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int wc_convert_weights(const int64_t* table, int count, int blocks_per_tensor, void* stream) {
    WC_CHECK_ARG(table && count > 0 && count <= 65535 && blocks_per_tensor > 0, "wc_convert_weights: bad argument");
    hipLaunchKernelGGL(convert_weights_kernel, dim3(blocks_per_tensor, count), dim3(256), 0, (hipStream_t)stream,
                       (const long*)table, count);
    WC_LAUNCH_CHECK("convert_weights_kernel");
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// AdamW step over many parameter tensors in ONE launch (reference utils/optimizer.py:3-33 = torch.optim.AdamW with a
// scheduled lr; torch runs ~10 multi-tensor kernels per group).  table: count rows of 8 int64
// {param, grad, exp_avg, exp_avg_sq, n, 0, 0, 0}.  Same update order as torch._multi_tensor_adam:
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = v*b2 + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_multi_kernel(const long* __restrict__ table, float decay, float b2, float w1,
                                                           float w2, float step_size, float bc2_sqrt, float eps) {
    const long* e = table + (long)blockIdx.y * 8;
    float* p = reinterpret_cast<float*>(e[0]);
    const float* g = reinterpret_cast<const float*>(e[1]);
    float* m = reinterpret_cast<float*>(e[2]);
    float* v = reinterpret_cast<float*>(e[3]);
    const long n = e[4];
    if (((n & 3) | (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) == 0) {
        // four elements per thread and access (16-byte IO); element-wise arithmetic unchanged
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + i), p4 = *reinterpret_cast<const float4*>(p + i);
            const float4 m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
            float po[4], mo[4], vo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pi = pp[k] * decay;
                mo[k] = mm[k] + w1 * (gg[k] - mm[k]);
                vo[k] = vv[k] * b2 + (w2 * gg[k]) * gg[k];
                const float denom = sqrtf(vo[k]) / bc2_sqrt + eps;
                po[k] = pi - step_size * (mo[k] / denom);
            }
            *reinterpret_cast<float4*>(p + i) = make_float4(po[0], po[1], po[2], po[3]);
            *reinterpret_cast<float4*>(m + i) = make_float4(mo[0], mo[1], mo[2], mo[3]);
            *reinterpret_cast<float4*>(v + i) = make_float4(vo[0], vo[1], vo[2], vo[3]);
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * decay;
        const float mi = m[i] + w1 * (gi - m[i]);
        const float vi = v[i] * b2 + (w2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// Hyper-parameters arrive as doubles and the derived scalars (1 - lr*wd, 1 - beta, lr / bc1, sqrt(bc2)) are formed in
// double like torch's Python-side arithmetic, then rounded once to fp32.
extern "C" int wc_adamw_multi(const int64_t* table, int count, double lr, double beta1, double beta2, double eps,
                              double weight_decay, double bias_correction1, double bias_correction2, int blocks_per_tensor,
                              void* stream) {
    WC_CHECK_ARG(table && count > 0 && count <= 65535 && blocks_per_tensor > 0 && bias_correction1 > 0 && bias_correction2 > 0,
                 "wc_adamw_multi: bad argument");
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks_per_tensor, count), dim3(256), 0, (hipStream_t)stream, (const long*)table,
                       (float)(1.0 - lr * weight_decay), (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                       (float)(lr / bias_correction1), (float)sqrt(bias_correction2), (float)eps);
    WC_LAUNCH_CHECK("adamw_multi_kernel");
    return WC_OK;
}
