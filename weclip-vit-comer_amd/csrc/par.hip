// Pixel-Adaptive Refinement (PAR) for gfx950.
// Replaces the stock-op composition of reference WeCLIP_model/PAR.py:39-92
// (replicate F.pad + one-hot dilated conv2d per dilation, std/softmax over the 48 taps,
// 20 x gather-multiply-sum) with two HBM-streaming stencil kernels:
//   par_affinity_kernel : img (B,3,H,W) -> aff (B,T,H,W), T = 8 * n_dilations taps, once
//   par_iter_kernel     : masks <- sum_t aff_t * masks(nbr_t), one launch per iteration
// Neighbour fetch is by index clamping (== replicate padding).  Everything is fp32.
// Layout: aff is tap-major planes so a wave reads 64 consecutive x of one plane (256 B).
#include "common.h"
#include <stdlib.h>

#define PAR_MAX_TAPS 64

struct ParTaps {
    int n;
    int dy[PAR_MAX_TAPS];
    int dx[PAR_MAX_TAPS];
    float pi[PAR_MAX_TAPS];   // w2 * softmax_t(-(pos_t/(std(pos)+1e-8)/w1)^2)
};

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// One thread per pixel; 4 passes over the taps (mean, variance, softmax stats, write) re-reading
// the image through L1/L2 (3 planes of H*W floats are cache resident); runs once per image.
__global__ __launch_bounds__(256) void par_affinity_kernel(const float* __restrict__ img,
                                                            float* __restrict__ aff, int H, int W,
                                                            float w1, ParTaps taps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const long HW = (long)H * W;
    const float* I = img + (long)blockIdx.z * 3 * HW;
    const long p = (long)y * W + x;
    const float c0 = I[p], c1 = I[HW + p], c2 = I[2 * HW + p];
    const int T = taps.n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < T; ++t) {
        const long o = (long)clampi(y + taps.dy[t], H - 1) * W + clampi(x + taps.dx[t], W - 1);
        s0 += I[o]; s1 += I[HW + o]; s2 += I[2 * HW + o];
    }
    const float m0 = s0 / T, m1 = s1 / T, m2 = s2 / T;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    for (int t = 0; t < T; ++t) {
        const long o = (long)clampi(y + taps.dy[t], H - 1) * W + clampi(x + taps.dx[t], W - 1);
        const float a = I[o] - m0, b = I[HW + o] - m1, c = I[2 * HW + o] - m2;
        q0 += a * a; q1 += b * b; q2 += c * c;
    }
    // unbiased std (torch.std default), PAR.py:77
    const float d0 = sqrtf(q0 / (T - 1)) + 1e-8f, d1 = sqrtf(q1 / (T - 1)) + 1e-8f,
                d2 = sqrtf(q2 / (T - 1)) + 1e-8f;
    float mx = -INFINITY, sum = 0.f;
    for (int t = 0; t < T; ++t) {
        const long o = (long)clampi(y + taps.dy[t], H - 1) * W + clampi(x + taps.dx[t], W - 1);
        const float a = fabsf(I[o] - c0) / d0 / w1, b = fabsf(I[HW + o] - c1) / d1 / w1,
                    c = fabsf(I[2 * HW + o] - c2) / d2 / w1;
        const float e = -(a * a + b * b + c * c) / 3.0f;
        const float nm = fmaxf(mx, e);
        sum = sum * __expf(mx - nm) + __expf(e - nm);
        mx = nm;
    }
    const float inv = 1.0f / sum;
    float* A = aff + (long)blockIdx.z * T * HW + p;
    for (int t = 0; t < T; ++t) {
        const long o = (long)clampi(y + taps.dy[t], H - 1) * W + clampi(x + taps.dx[t], W - 1);
        const float a = fabsf(I[o] - c0) / d0 / w1, b = fabsf(I[HW + o] - c1) / d1 / w1,
                    c = fabsf(I[2 * HW + o] - c2) / d2 / w1;
        const float e = -(a * a + b * b + c * c) / 3.0f;
        A[(long)t * HW] = __expf(e - mx) * inv + taps.pi[t];
    }
}

// Register-resident variant for a compile-time tap count (ND dilations, T = 8*ND taps): the 3*T
// neighbour values are fetched once and kept in VGPRs (one pass over the image instead of four).
// TILED aff layout (internal to wc_par_forward): [image][y][x/64][tap][64] -- the T tap values of a
// 64-pixel strip are contiguous (T*256 B), so one wave streams one contiguous block per sweep instead of
// T streams 1 MiB apart (DRAM/MALL friendlier); plane-major (B,T,H,W) stays the public wc_par_affinity layout.
__device__ __forceinline__ long aff_base(bool tiled, long img, int T, int H, int W, int y, int x, long* tstride) {
    if (tiled) {
        const int XT = (W + 63) >> 6;
        *tstride = 64;
        return ((img * H + y) * XT + (x >> 6)) * (long)T * 64 + (x & 63);
    }
    *tstride = (long)H * W;
    return img * T * (long)H * W + (long)y * W + x;
}

// H16 (only with TILED): the affinities are stored as 16-bit fixed-point PAIRS with one fp32 scale per pixel,
// [image][y][x/64][tap/2 (+1 slot: the scale)][64] dwords -- one 4-byte load per lane fetches two taps, a sweep reads
// half the bytes.  q_t = round(a_t / scale), scale = max_t a_t / 65535: |error| <= max_t a_t * 7.7e-6 per weight (a
// flat region has max a ~ 0.02: 1.6e-7; fp16 storage would be 7.6e-6 there and measured 3e-3 on the masks after 20
// sweeps).  Used by the "fast" precision mode only; the exact mode keeps fp32.
#define PAR_Q_SLOTS(T_) ((T_) / 2 + 1)
template <int ND, bool TILED, bool H16 = false>
__global__ __launch_bounds__(256) void par_affinity_reg_kernel(const float* __restrict__ img,
                                                                float* __restrict__ aff, int H, int W,
                                                                float w1, ParTaps taps) {
    constexpr int T = 8 * ND;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const long HW = (long)H * W;
    const float* I = img + (long)blockIdx.z * 3 * HW;
    const long p = (long)y * W + x;
    const float c0 = I[p], c1 = I[HW + p], c2 = I[2 * HW + p];
    float v0[T], v1[T], v2[T];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    // the 8 neighbours of a dilation d (build_taps order: rows y-d, y, y+d x columns x-d, x, x+d without the centre) from
    // three clamped row offsets and three clamped columns: 6 clamps + 8 adds per dilation instead of 16 clamps + 8 multiplies
    constexpr int RY[8] = {0, 0, 0, 1, 1, 2, 2, 2}, CX[8] = {0, 1, 2, 0, 2, 0, 1, 2};
#pragma unroll
    for (int dl = 0; dl < ND; ++dl) {
        const int d = taps.dx[8 * dl + 2];
        const int rw[3] = {clampi(y - d, H - 1) * W, y * W, clampi(y + d, H - 1) * W};
        const int cl[3] = {clampi(x - d, W - 1), x, clampi(x + d, W - 1)};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = 8 * dl + u;
            const int o = rw[RY[u]] + cl[CX[u]];
            v0[t] = I[o]; v1[t] = I[HW + o]; v2[t] = I[2 * HW + o];
            s0 += v0[t]; s1 += v1[t]; s2 += v2[t];
        }
    }
    const float m0 = s0 / T, m1 = s1 / T, m2 = s2 / T;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float a = v0[t] - m0, b = v1[t] - m1, c = v2[t] - m2;
        q0 += a * a; q1 += b * b; q2 += c * c;
    }
    const float d0 = sqrtf(q0 / (T - 1)) + 1e-8f, d1 = sqrtf(q1 / (T - 1)) + 1e-8f,
                d2 = sqrtf(q2 / (T - 1)) + 1e-8f;
    // |v - c| / d / w1 as one multiplication by 1 / (d * w1) (one exact division per channel and pixel instead
    // of 6 per tap: the kernel was division bound); differs from the two divisions by <= 2 ulp
    const float r0 = 1.0f / (d0 * w1), r1 = 1.0f / (d1 * w1), r2 = 1.0f / (d2 * w1);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float a = fabsf(v0[t] - c0) * r0, b = fabsf(v1[t] - c1) * r1, c = fabsf(v2[t] - c2) * r2;
        v0[t] = -(a * a + b * b + c * c) * (1.0f / 3.0f);
        mx = fmaxf(mx, v0[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        v0[t] = __expf(v0[t] - mx);
        sum += v0[t];
    }
    const float inv = 1.0f / sum;
    if constexpr (H16) {
        const int XT = (W + 63) >> 6;
        unsigned* A2 = reinterpret_cast<unsigned*>(aff) + (((long)blockIdx.z * H + y) * XT + (x >> 6)) * (long)PAR_Q_SLOTS(T) * 64 + (x & 63);
        float amax = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            v0[t] = v0[t] * inv + taps.pi[t];
            amax = fmaxf(amax, v0[t]);
        }
        const float scale = amax * (1.0f / 65535.0f), rs = 65535.0f / amax;
        // error-diffused rounding: each weight takes over the rounding remainder of the previous one, so the 48 stored
        // weights of a pixel sum to the fp32 row sum within ONE quantum (plain rounding is biased where the weights are
        // nearly equal: flat regions round all 48 the same way and the bias compounds over the 20 sweeps)
        float carry = 0.f;
#pragma unroll
        for (int j = 0; j < T / 2; ++j) {
            const float a0 = v0[2 * j] + carry;
            const float q0 = fminf(fmaxf(rintf(a0 * rs), 0.f), 65535.f);
            carry = a0 - q0 * scale;
            const float a1 = v0[2 * j + 1] + carry;
            const float q1 = fminf(fmaxf(rintf(a1 * rs), 0.f), 65535.f);
            carry = a1 - q1 * scale;
            A2[j * 64] = (unsigned)q0 | ((unsigned)q1 << 16);
        }
        A2[(T / 2) * 64] = __float_as_uint(scale);
        return;
    }
    long ts;
    float* A = aff + aff_base(TILED, blockIdx.z, T, H, W, y, x, &ts);
#pragma unroll
    for (int t = 0; t < T; ++t) A[t * ts] = v0[t] * inv + taps.pi[t];
}

static void launch_affinity(const float* img, float* aff, int nb, int H, int W, const ParTaps& tp,
                            hipStream_t st, bool tiled, bool h16 = false) {
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(H, 4), nb);
    const int pr = wc_prof_begin(st);
    if (tp.n == 48 && tiled && h16) {
        // (an LDS-tile variant of this set-up was measured slower in round 3: tools/probes/rejected_r03/)
        hipLaunchKernelGGL((par_affinity_reg_kernel<6, true, true>), grid, dim3(256), 0, st, img, aff, H, W, 0.3f, tp);
        wc_prof_end(pr, "par_affinity_reg_kernel<6, true, true>", 4.0 * nb * (double)H * W * 3 + nb * (double)H * W * (2.0 * tp.n + 4.0), st);
        return;
    }
    if (tp.n == 48 && tiled)
        hipLaunchKernelGGL((par_affinity_reg_kernel<6, true>), grid, dim3(256), 0, st, img, aff, H, W, 0.3f, tp);
    else if (tp.n == 48)
        hipLaunchKernelGGL((par_affinity_reg_kernel<6, false>), grid, dim3(256), 0, st, img, aff, H, W, 0.3f, tp);
    else
        hipLaunchKernelGGL(par_affinity_kernel, grid, dim3(256), 0, st, img, aff, H, W, 0.3f, tp);
    // algorithmic bytes: 3 image planes read + T affinity planes written
    wc_prof_end(pr, tp.n == 48 ? (tiled ? "par_affinity_reg_kernel<6, true>" : "par_affinity_reg_kernel<6, false>") : "par_affinity_kernel",
                4.0 * nb * (double)H * W * (3 + tp.n), st);
}

// One PAR iteration.  Thread = one pixel, CG channels at a time (aff value reused across the
// channel group).  HBM traffic per launch: T*H*W*4 (aff) + 2*C*H*W*4 (masks in/out); the
// neighbour gathers of `min` hit L1/L2 (C planes of H*W floats).
#ifndef PAR_TAP_BATCH
#define PAR_TAP_BATCH 8
#endif
template <int CG, bool TILED, bool H16 = false, int ROWS = 4, int PAR_HALO = 4>
__global__ __launch_bounds__(64 * ROWS) void par_iter_kernel(const float* __restrict__ aff,
                                                        const float* __restrict__ min,
                                                        float* __restrict__ mout, int C, int H,
                                                        int W, ParTaps taps) {
    // H16: the mask values of the taps with dilation <= PAR_HALO come from an LDS tile of the block's 64 x ROWS pixels
    // + halo, filled with already clamped (replicate-padded) values: with 16-bit affinities the sweep is bound by the
    // L1 / address path (168 four-byte loads per pixel), not by HBM; with 8 rows and a halo of 12 the tile replaces
    // 120 of the 144 gathers by ~17 fill loads + LDS reads.
    constexpr int TW = 64 + 2 * PAR_HALO, TH = ROWS + 2 * PAR_HALO;
    __shared__ float tile[H16 ? CG * TH * TW : 1];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    int x = blockIdx.x * 64 + lx;
    int y = blockIdx.y * ROWS + ly;
    const bool live = x < W && y < H;
    if constexpr (H16) {          // every thread reaches the barriers: out-of-image threads compute on a clamped pixel
        x = x < W ? x : W - 1;
        y = y < H ? y : H - 1;
    } else {
        if (!live) return;
    }
    const long HW = (long)H * W;
    const long p = (long)y * W + x;
    const int T = taps.n;
    long ts;
    const float* A = aff + aff_base(TILED, blockIdx.z, T, H, W, y, x, &ts);
    const unsigned* A2 = reinterpret_cast<const unsigned*>(aff) +
                         (((long)blockIdx.z * H + y) * ((W + 63) >> 6) + (x >> 6)) * (long)PAR_Q_SLOTS(T) * 64 + (x & 63);   // H16 layout
    float qscale = 1.f;
    if constexpr (H16) qscale = __uint_as_float(A2[(T / 2) * 64]);
    const float* M = min + (long)blockIdx.z * C * HW;
    float* O = mout + (long)blockIdx.z * C * HW + p;
    for (int cb = 0; cb < C; cb += CG) {
        float acc[CG];
        const float* Mc[CG];
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            acc[k] = 0.f;
            Mc[k] = M + (long)(cb + k < C ? cb + k : C - 1) * HW;
        }
        if constexpr (H16) {
            if (cb) __syncthreads();          // the previous channel group is done with the tile
            // thread (lx, ly) fills columns lx, lx + 64 of rows ly, ly + ROWS, ..: no division, one row multiply per row
            for (int ty = ly; ty < TH; ty += ROWS) {
                const int ro = clampi(blockIdx.y * ROWS + ty - PAR_HALO, H - 1) * W;
                for (int tx = lx; tx < TW; tx += 64) {
                    // uniform plane pointer + unsigned 32-bit byte offset: the scalar-base addressing form, no 64-bit
                    // vector address arithmetic per load
                    const unsigned ob = (unsigned)(ro + clampi(blockIdx.x * 64 + tx - PAR_HALO, W - 1)) * 4u;
#pragma unroll
                    for (int k = 0; k < CG; ++k)
                        tile[(k * TH + ty) * TW + tx] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Mc[k]) + ob);
                }
            }
            __syncthreads();
        }
        constexpr int TB_ = PAR_TAP_BATCH;
        // TB_ taps per batch: their aff loads and TB_*CG gathers are all issued before the first FMA
        // (the rolled loop kept only 1+CG loads in flight per wave and was latency bound).
        int t = 0;
        for (; t + TB_ <= T; t += TB_) {
            float a[TB_], m[TB_][CG];
            unsigned a2[TB_ / 2];
            if constexpr (H16) {
#pragma unroll
                for (int u = 0; u < TB_ / 2; ++u) a2[u] = A2[((t >> 1) + u) * 64];
            }
            bool near = false;                 // one batch = the 8 taps of one dilation (get_kernel order)
            if constexpr (H16) near = (taps.dy[t] < 0 ? -taps.dy[t] : taps.dy[t]) <= PAR_HALO && (taps.dx[t] < 0 ? -taps.dx[t] : taps.dx[t]) <= PAR_HALO;
            if (near) {
                // tile index = the thread's own + a wave-uniform tap offset (scalar arithmetic): one vector add per tap
                const int tb = (ly + PAR_HALO) * TW + lx + PAR_HALO;
#pragma unroll
                for (int u = 0; u < TB_; ++u) {
                    const int o = tb + (taps.dy[t + u] * TW + taps.dx[t + u]);
#pragma unroll
                    for (int k = 0; k < CG; ++k) m[u][k] = tile[k * TH * TW + o];
                }
            } else if constexpr (H16) {
                // the 8 neighbours of one dilation d (build_taps: rows y-d, y, y+d x columns x-d, x, x+d without the
                // centre): three clamped row offsets and three clamped columns instead of 8 x (2 clamps + multiply)
                const int d = taps.dx[t + 2];
                const int rw[3] = {clampi(y - d, H - 1) * W, y * W, clampi(y + d, H - 1) * W};
                const int cl[3] = {clampi(x - d, W - 1), x, clampi(x + d, W - 1)};
                constexpr int RY[8] = {0, 0, 0, 1, 1, 2, 2, 2}, CX[8] = {0, 1, 2, 0, 2, 0, 1, 2};
                static_assert(TB_ == 8, "one batch = one dilation");
#pragma unroll
                for (int u = 0; u < TB_; ++u) {
                    const unsigned ob = (unsigned)(rw[RY[u]] + cl[CX[u]]) * 4u;
#pragma unroll
                    for (int k = 0; k < CG; ++k) m[u][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Mc[k]) + ob);
                }
            } else {
#pragma unroll
                for (int u = 0; u < TB_; ++u) {
                    a[u] = A[(t + u) * ts];
                    const int o = clampi(y + taps.dy[t + u], H - 1) * W + clampi(x + taps.dx[t + u], W - 1);
#pragma unroll
                    for (int k = 0; k < CG; ++k) m[u][k] = Mc[k][o];
                }
            }
            if constexpr (H16) {
#pragma unroll
                for (int u = 0; u < TB_ / 2; ++u) {
                    a[2 * u] = (float)(a2[u] & 0xffffu);
                    a[2 * u + 1] = (float)(a2[u] >> 16);
                }
            }
#pragma unroll
            for (int u = 0; u < TB_; ++u)
#pragma unroll
                for (int k = 0; k < CG; ++k) acc[k] = fmaf(a[u], m[u][k], acc[k]);
        }
        for (; t < T; ++t) {
            const float a = A[t * ts];
            const int o = clampi(y + taps.dy[t], H - 1) * W + clampi(x + taps.dx[t], W - 1);
#pragma unroll
            for (int k = 0; k < CG; ++k) acc[k] = fmaf(a, Mc[k][o], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < CG; ++k)
            if (cb + k < C && live) O[(long)(cb + k) * HW] = H16 ? acc[k] * qscale : acc[k];
    }
}

// labels[b,y,x] = valid_key[b, argmax_c masks[b,c,y,x]] over the first nch[b] channels
// (first maximum wins like torch.argmax).  model_attn_aff_voc.py:49-57.
__global__ __launch_bounds__(256) void par_labels_kernel(const float* __restrict__ masks,
                                                          const long* __restrict__ valid_key,
                                                          const int* __restrict__ nch,
                                                          long* __restrict__ labels, int C, long HW) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int b = blockIdx.y;
    const int n = nch ? nch[b] : C;
    const float* M = masks + (long)b * C * HW + p;
    float best = M[0];
    int bi = 0;
    for (int c = 1; c < n; ++c) {
        const float v = M[(long)c * HW];
        if (v > best) { best = v; bi = c; }
    }
    labels[(long)b * HW + p] = valid_key[(long)b * C + bi];
}

static int build_taps(ParTaps* tp, const int* dilations, int n_dil, float w1, float w2) {
    if (n_dil < 1 || n_dil * 8 > PAR_MAX_TAPS) return 1;
    static const int DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};   // get_kernel() order, PAR.py:10-24
    static const int DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    const int T = n_dil * 8;
    float pos[PAR_MAX_TAPS];
    for (int d = 0; d < n_dil; ++d)
        for (int k = 0; k < 8; ++k) {
            const int t = d * 8 + k;
            tp->dy[t] = DY[k] * dilations[d];
            tp->dx[t] = DX[k] * dilations[d];
            pos[t] = (float)dilations[d] * ((DY[k] != 0 && DX[k] != 0) ? sqrtf(2.0f) : 1.0f);
        }
    float mean = 0.f;
    for (int t = 0; t < T; ++t) mean += pos[t];
    mean /= T;
    float var = 0.f;
    for (int t = 0; t < T; ++t) var += (pos[t] - mean) * (pos[t] - mean);
    const float sd = sqrtf(var / (T - 1)) + 1e-8f;
    float e[PAR_MAX_TAPS], mx = -INFINITY, sum = 0.f;
    for (int t = 0; t < T; ++t) {
        const float v = pos[t] / sd / w1;
        e[t] = -(v * v);
        mx = fmaxf(mx, e[t]);
    }
    for (int t = 0; t < T; ++t) sum += expf(e[t] - mx);
    for (int t = 0; t < T; ++t) tp->pi[t] = w2 * expf(e[t] - mx) / sum;
    tp->n = T;
    return 0;
}

extern "C" int wc_par_affinity(const float* img, float* aff, int B, int H, int W,
                               const int* dilations, int n_dil, void* stream) {
    WC_CHECK_ARG(img && aff && B > 0 && H > 0 && W > 0, "wc_par_affinity: bad argument");
    ParTaps tp;
    WC_CHECK_ARG(build_taps(&tp, dilations, n_dil, 0.3f, 0.01f) == 0, "wc_par_affinity: 1..8 dilations");
    launch_affinity(img, aff, B, H, W, tp, (hipStream_t)stream, false);
    WC_LAUNCH_CHECK("par_affinity_kernel");
    return WC_OK;
}

static const char* par_iter_name(int C, bool tiled, bool h16) {
    static const char* hn[3] = {"par_iter_kernel<2, true, true, 8, 12>", "par_iter_kernel<3, true, true, 8, 12>", "par_iter_kernel<4, true, true, 8, 12>"};
    static const char* names[2][3] = {{"par_iter_kernel<2, false>", "par_iter_kernel<3, false>", "par_iter_kernel<4, false>"},
                                      {"par_iter_kernel<2, true>", "par_iter_kernel<3, true>", "par_iter_kernel<4, true>"}};
    const int ci = C <= 2 ? 0 : (C == 3 ? 1 : 2);
    return h16 ? hn[ci] : names[tiled ? 1 : 0][ci];
}
// algorithmic bytes of one sweep: the affinities (T 16-bit values + a scale, or T floats) + C mask planes read, C written
static double par_iter_bytes(int B, int C, int H, int W, const ParTaps& tp, bool h16) {
    return h16 ? B * (double)H * W * (2.0 * tp.n + 4.0 + 8.0 * C) : 4.0 * B * (double)H * W * (tp.n + 2 * C);
}

// prof: bracket this launch with its own event pair (a stand-alone sweep); the sweeps of a whole PAR.forward share one pair
static int launch_iter(const float* aff, const float* src, float* dst, int B, int C, int H, int W,
                       const ParTaps& tp, hipStream_t st, bool tiled, bool h16 = false, bool prof = true) {
    dim3 grid(wc_cdiv(W, 64), wc_cdiv(H, 4), B);
    if (h16) {       // fp16-pair affinities (tiled layout, 48 taps)
        const int pr = prof ? wc_prof_begin(st) : -1;
        // blocks of 64 x 8 pixels with a 12-pixel halo: the tile serves dilations 1..12 (40 of the 48 taps); measured
        // 2.44 ms (4 rows, halo 4) -> 2.29 ms per 16-image PAR.forward
        dim3 grid8(wc_cdiv(W, 64), wc_cdiv(H, 8), B);
        if (C <= 2) hipLaunchKernelGGL((par_iter_kernel<2, true, true, 8, 12>), grid8, dim3(512), 0, st, aff, src, dst, C, H, W, tp);
        else if (C == 3) hipLaunchKernelGGL((par_iter_kernel<3, true, true, 8, 12>), grid8, dim3(512), 0, st, aff, src, dst, C, H, W, tp);
        else hipLaunchKernelGGL((par_iter_kernel<4, true, true, 8, 12>), grid8, dim3(512), 0, st, aff, src, dst, C, H, W, tp);
        wc_prof_end(pr, par_iter_name(C, tiled, true), par_iter_bytes(B, C, H, W, tp, true), st);
        WC_LAUNCH_CHECK("par_iter_kernel");
        return WC_OK;
    }
#define PAR_ITER_LAUNCH(CG_) \
    if (tiled) hipLaunchKernelGGL((par_iter_kernel<CG_, true>), grid, dim3(256), 0, st, aff, src, dst, C, H, W, tp); \
    else hipLaunchKernelGGL((par_iter_kernel<CG_, false>), grid, dim3(256), 0, st, aff, src, dst, C, H, W, tp);
    const int pr = prof ? wc_prof_begin(st) : -1;
    if (C <= 2) { PAR_ITER_LAUNCH(2) }
    else if (C == 3) { PAR_ITER_LAUNCH(3) }
    else { PAR_ITER_LAUNCH(4) }
    wc_prof_end(pr, par_iter_name(C, tiled, false), par_iter_bytes(B, C, H, W, tp, false), st);
    WC_LAUNCH_CHECK("par_iter_kernel");
    return WC_OK;
}

extern "C" int wc_par_iterate(const float* aff, const float* masks_in, float* masks_out, int B, int C,
                              int H, int W, const int* dilations, int n_dil, void* stream) {
    WC_CHECK_ARG(aff && masks_in && masks_out && masks_in != masks_out && B > 0 && C > 0,
                 "wc_par_iterate: bad argument (in-place not allowed)");
    ParTaps tp;
    WC_CHECK_ARG(build_taps(&tp, dilations, n_dil, 0.3f, 0.01f) == 0, "wc_par_iterate: 1..8 dilations");
    return launch_iter(aff, masks_in, masks_out, B, C, H, W, tp, (hipStream_t)stream, false);
}

// Whole PAR.forward for images already at mask resolution (PAR.py:64-92).
// Images are processed in groups of `group` so that a group's aff planes (T*H*W*4 B each)
// stay resident in the 256 MiB Infinity Cache across the num_iter sweeps.
// Workspaces: aff_ws >= min(group,B)*T*H*ceil64(W) floats, tmp >= B*C*H*W floats.
static int par_forward_impl(const float* img, const float* masks, float* out, float* tmp, float* aff_ws, int B, int C,
                            int H, int W, const int* dilations, int n_dil, int num_iter, int group, void* stream, bool h16);

extern "C" int wc_par_forward(const float* img, const float* masks, float* out, float* tmp,
                              float* aff_ws, int B, int C, int H, int W, const int* dilations,
                              int n_dil, int num_iter, int group, void* stream) {
    return par_forward_impl(img, masks, out, tmp, aff_ws, B, C, H, W, dilations, n_dil, num_iter, group, stream, false);
}

// Same, with the affinities kept as 16-bit fixed-point pairs + one scale per pixel between the sweeps (half the bytes
// per sweep; |error| <= max_t a_t * 7.7e-6 per weight): the "fast" precision mode.  Needs the 48-tap configuration
// (6 dilations); aff_ws: (T/2 + 1) * H * ceil64(W) dwords per image of a group.
extern "C" int wc_par_forward_h(const float* img, const float* masks, float* out, float* tmp,
                                float* aff_ws, int B, int C, int H, int W, const int* dilations,
                                int n_dil, int num_iter, int group, void* stream) {
    WC_CHECK_ARG(n_dil == 6, "wc_par_forward_h: needs 6 dilations (48 taps)");
    return par_forward_impl(img, masks, out, tmp, aff_ws, B, C, H, W, dilations, n_dil, num_iter, group, stream, true);
}

static int par_forward_impl(const float* img, const float* masks, float* out, float* tmp, float* aff_ws, int B, int C,
                            int H, int W, const int* dilations, int n_dil, int num_iter, int group, void* stream, bool h16) {
    WC_CHECK_ARG(img && masks && out && tmp && aff_ws && B > 0 && C > 0 && H > 0 && W > 0 &&
                     num_iter >= 1 && group >= 1,
                 "wc_par_forward: bad argument");
    WC_CHECK_ARG(out != masks && tmp != masks && out != tmp, "wc_par_forward: buffers must not alias");
    ParTaps tp;
    WC_CHECK_ARG(build_taps(&tp, dilations, n_dil, 0.3f, 0.01f) == 0, "wc_par_forward: 1..8 dilations");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const bool tiled = (tp.n == 48);   // strip-interleaved aff (needs aff_ws >= group*T*H*ceil64(W) floats)
    for (int b0 = 0; b0 < B; b0 += group) {
        const int nb = (B - b0 < group) ? B - b0 : group;
        launch_affinity(img + (long)b0 * 3 * HW, aff_ws, nb, H, W, tp, st, tiled, h16 && tiled);
        WC_LAUNCH_CHECK("par_affinity_kernel");
        const float* src = masks + (long)b0 * C * HW;
        const int pr = wc_prof_begin_always(st);   // ONE event pair around the group's num_iter dependent sweeps
        for (int i = 0; i < num_iter; ++i) {
            float* dst = (((num_iter - i) & 1) ? out : tmp) + (long)b0 * C * HW;
            int rc = launch_iter(aff_ws, src, dst, nb, C, H, W, tp, st, tiled, h16 && tiled, false);
            if (rc) return rc;
            src = dst;
        }
        wc_prof_end_n(pr, par_iter_name(C, tiled, h16 && tiled), num_iter * par_iter_bytes(nb, C, H, W, tp, h16 && tiled), st, num_iter);
    }
    return WC_OK;
}

extern "C" int wc_par_labels(const float* masks, const int64_t* valid_key, const int* nch,
                             int64_t* labels, int B, int C, int H, int W, void* stream) {
    WC_CHECK_ARG(masks && valid_key && labels && B > 0 && C > 0, "wc_par_labels: bad argument");
    const long HW = (long)H * W;
    dim3 grid(wc_cdiv(HW, 256), B);
    hipLaunchKernelGGL(par_labels_kernel, grid, dim3(256), 0, (hipStream_t)stream, masks,
                       (const long*)valid_key, nch, (long*)labels, C, HW);
    WC_LAUNCH_CHECK("par_labels_kernel");
    return WC_OK;
}
