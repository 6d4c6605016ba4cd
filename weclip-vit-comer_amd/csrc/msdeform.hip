// Multi-scale deformable attention core (ViT-CoMer CTI blocks) for gfx950.
//
// There is NO reference implementation of the CoMer inserts in dayae1204/WeCLIP-ViT-CoMer (only the
// paper ViT_CoMer.pdf §3.3 and the task brief); this follows the published MSDeformAttn definition
// (Deformable-DETR, which ViT-CoMer's CTI uses):
//   out[n,q,m,:] = sum_{l<nL} sum_{p<nP} A[n,q,m,l,p] * bilinear(value_l[n,:,m,:], loc[n,q,m,l,p])
// with loc in [0,1]^2 (x,y), pixel coordinate = loc * size - 0.5, zero outside the map
// (== F.grid_sample(align_corners=False, padding_mode='zeros')).  Parity is pinned only against the
// CPU restatement oracle/comer_oracle.py ("parity unpinned" w.r.t. the reference).
//
// Layout: value (N, S, M, D) with S = sum_l H_l*W_l; one 64*k-thread block per query, thread = (head m,
// channel d): the D channels of a sampled corner are D consecutive floats, so every gather is a
// coalesced D*4-byte read.  Backward scatters grad_value with float atomics (one dword per lane,
// contiguous per head) and reduces the location / weight gradients across the D lanes of a head.
#include "common.h"

#define MSDA_MAX_LEVELS 8

struct MsdaShapes {
    int n_levels;
    int H[MSDA_MAX_LEVELS], W[MSDA_MAX_LEVELS], start[MSDA_MAX_LEVELS];
};

__global__ void msda_fwd_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, float* __restrict__ out, MsdaShapes sh, int S, int Lq,
                                int M, int D, int P) {
    const long nq = blockIdx.x;                  // n * Lq + q
    const int n = nq / Lq;
    const int m = threadIdx.x / D, d = threadIdx.x - m * D;
    if (m >= M) return;
    const float* vb = value + (long)n * S * M * D + (long)m * D + d;
    const float* lb = loc + ((nq * M + m) * sh.n_levels) * P * 2;
    const float* ab = attn + ((nq * M + m) * sh.n_levels) * P;
    float acc = 0.f;
    for (int l = 0; l < sh.n_levels; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const float* vl = vb + (long)sh.start[l] * M * D;
        for (int p = 0; p < P; ++p) {
            const float x = lb[(l * P + p) * 2] * W - 0.5f, y = lb[(l * P + p) * 2 + 1] * H - 0.5f;
            const float w = ab[l * P + p];
            if (y > -1.f && x > -1.f && y < H && x < W) {
                const int y0 = (int)floorf(y), x0 = (int)floorf(x);
                const float ly = y - y0, lx = x - x0, hy = 1.f - ly, hx = 1.f - lx;
                float v = 0.f;
                if (y0 >= 0 && x0 >= 0) v += hy * hx * vl[((long)y0 * W + x0) * M * D];
                if (y0 >= 0 && x0 + 1 < W) v += hy * lx * vl[((long)y0 * W + x0 + 1) * M * D];
                if (y0 + 1 < H && x0 >= 0) v += ly * hx * vl[((long)(y0 + 1) * W + x0) * M * D];
                if (y0 + 1 < H && x0 + 1 < W) v += ly * lx * vl[((long)(y0 + 1) * W + x0 + 1) * M * D];
                acc = fmaf(w, v, acc);
            }
        }
    }
    out[nq * M * D + (long)m * D + d] = acc;
}

__device__ __forceinline__ float head_sum(float v, int D) {   // sum over the D consecutive lanes of a head
    for (int o = D >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void msda_bwd_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, const float* __restrict__ gout,
                                float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gattn,
                                MsdaShapes sh, int S, int Lq, int M, int D, int P) {
    const long nq = blockIdx.x;
    const int n = nq / Lq;
    const int m = threadIdx.x / D, d = threadIdx.x - m * D;
    if (m >= M) return;
    const long vo = (long)n * S * M * D + (long)m * D + d;
    const float* lb = loc + ((nq * M + m) * sh.n_levels) * P * 2;
    const float* ab = attn + ((nq * M + m) * sh.n_levels) * P;
    const float go = gout[nq * M * D + (long)m * D + d];
    for (int l = 0; l < sh.n_levels; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const long lvl = vo + (long)sh.start[l] * M * D;
        for (int p = 0; p < P; ++p) {
            const float x = lb[(l * P + p) * 2] * W - 0.5f, y = lb[(l * P + p) * 2 + 1] * H - 0.5f;
            const float w = ab[l * P + p];
            float gx = 0.f, gy = 0.f, ga = 0.f;
            if (y > -1.f && x > -1.f && y < H && x < W) {
                const int y0 = (int)floorf(y), x0 = (int)floorf(x);
                const float ly = y - y0, lx = x - x0, hy = 1.f - ly, hx = 1.f - lx;
                const float gw = go * w;
                float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
                if (y0 >= 0 && x0 >= 0) {
                    const long o = lvl + ((long)y0 * W + x0) * M * D;
                    v00 = value[o];
                    atomicAdd(gvalue + o, gw * hy * hx);
                }
                if (y0 >= 0 && x0 + 1 < W) {
                    const long o = lvl + ((long)y0 * W + x0 + 1) * M * D;
                    v01 = value[o];
                    atomicAdd(gvalue + o, gw * hy * lx);
                }
                if (y0 + 1 < H && x0 >= 0) {
                    const long o = lvl + ((long)(y0 + 1) * W + x0) * M * D;
                    v10 = value[o];
                    atomicAdd(gvalue + o, gw * ly * hx);
                }
                if (y0 + 1 < H && x0 + 1 < W) {
                    const long o = lvl + ((long)(y0 + 1) * W + x0 + 1) * M * D;
                    v11 = value[o];
                    atomicAdd(gvalue + o, gw * ly * lx);
                }
                ga = go * (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
                gx = gw * W * (hy * (v01 - v00) + ly * (v11 - v10));     // d/dloc_x (pixel x = loc_x*W - 0.5)
                gy = gw * H * (hx * (v10 - v00) + lx * (v11 - v01));
            }
            gx = head_sum(gx, D); gy = head_sum(gy, D); ga = head_sum(ga, D);
            if (d == 0) {
                gloc[((nq * M + m) * sh.n_levels * P + l * P + p) * 2] = gx;
                gloc[((nq * M + m) * sh.n_levels * P + l * P + p) * 2 + 1] = gy;
                gattn[(nq * M + m) * sh.n_levels * P + l * P + p] = ga;
            }
        }
    }
}

static int fill_shapes(MsdaShapes* sh, const int* h_shapes, int n_levels, int* S) {
    if (n_levels < 1 || n_levels > MSDA_MAX_LEVELS) return 1;
    sh->n_levels = n_levels;
    int s = 0;
    for (int l = 0; l < n_levels; ++l) {
        sh->H[l] = h_shapes[2 * l];
        sh->W[l] = h_shapes[2 * l + 1];
        if (sh->H[l] <= 0 || sh->W[l] <= 0) return 1;
        sh->start[l] = s;
        s += sh->H[l] * sh->W[l];
    }
    *S = s;
    return 0;
}

extern "C" int wc_msda_fwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                           float* out, int N, int Lq, int M, int D, int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && loc && attn && out && N > 0 && Lq > 0 && M > 0 && P > 0, "wc_msda_fwd: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0, "wc_msda_fwd: 1..8 levels with positive sizes");
    WC_CHECK_ARG((D == 16 || D == 32 || D == 64) && M * D <= 1024 && (M * D) % 64 == 0,
                 "wc_msda_fwd: head dim 16/32/64, heads*dim a multiple of 64 and <= 1024");
    hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)((long)N * Lq)), dim3(M * D), 0, (hipStream_t)stream, value, loc, attn,
                       out, sh, S, Lq, M, D, P);
    WC_LAUNCH_CHECK("msda_fwd_kernel");
    return WC_OK;
}

// gvalue must be zero-initialised by the caller (it is accumulated with atomics).
extern "C" int wc_msda_bwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                           const float* gout, float* gvalue, float* gloc, float* gattn, int N, int Lq, int M, int D,
                           int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && loc && attn && gout && gvalue && gloc && gattn && N > 0 && Lq > 0 && M > 0 && P > 0,
                 "wc_msda_bwd: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0, "wc_msda_bwd: 1..8 levels with positive sizes");
    WC_CHECK_ARG((D == 16 || D == 32 || D == 64) && M * D <= 1024 && (M * D) % 64 == 0,
                 "wc_msda_bwd: head dim 16/32/64, heads*dim a multiple of 64 and <= 1024");
    hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)((long)N * Lq)), dim3(M * D), 0, (hipStream_t)stream, value, loc, attn,
                       gout, gvalue, gloc, gattn, sh, S, Lq, M, D, P);
    WC_LAUNCH_CHECK("msda_bwd_kernel");
    return WC_OK;
}
