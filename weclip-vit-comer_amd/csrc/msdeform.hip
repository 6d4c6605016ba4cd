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
// coalesced D*4-byte read.  Backward: the location / weight gradients are reduced across the D lanes of a head
// (msda_bwd_kernel); grad_value is a scatter (many samples land on one pixel), done WITHOUT float atomics as a bucketed
// gather (msda_bwd_value_kernel: count -> scan -> fill -> per-pixel gather in 64-bit fixed point), bit-reproducible and
// with every element of grad_value written exactly once.
#include "common.h"

#define MSDA_MAX_LEVELS 8

struct MsdaShapes {
    int n_levels;
    int H[MSDA_MAX_LEVELS], W[MSDA_MAX_LEVELS], start[MSDA_MAX_LEVELS];
};

__global__ __launch_bounds__(1024) void msda_fwd_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, float* __restrict__ out, __half* __restrict__ out16,
                                MsdaShapes sh, int S, int Lq, int M, int D, int P) {
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
    if (out) out[nq * M * D + (long)m * D + d] = acc;
    if (out16) out16[nq * M * D + (long)m * D + d] = __float2half(acc);       // the output projection's MFMA operand
}

__device__ __forceinline__ float head_sum(float v, int D) {   // sum over the D consecutive lanes of a head
    for (int o = D >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(1024) void msda_bwd_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, const float* __restrict__ gout,
                                float* __restrict__ gloc, float* __restrict__ gattn,
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
                }
                if (y0 >= 0 && x0 + 1 < W) {
                    const long o = lvl + ((long)y0 * W + x0 + 1) * M * D;
                    v01 = value[o];
                }
                if (y0 + 1 < H && x0 >= 0) {
                    const long o = lvl + ((long)(y0 + 1) * W + x0) * M * D;
                    v10 = value[o];
                }
                if (y0 + 1 < H && x0 + 1 < W) {
                    const long o = lvl + ((long)(y0 + 1) * W + x0 + 1) * M * D;
                    v11 = value[o];
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

// ---- channel-quad forms (round 3): thread = (query, head, 4 consecutive channels).  A corner is then ONE 16-byte load per
// lane and D/4 lanes cover a head's D channels (a 128-byte line for D = 32), four queries share a 256-thread workgroup
// (M*D/4 = 64 lanes per query), and the per-sample reductions of the location / weight gradients run over D/4 lanes (3
// shuffle steps for D = 32) instead of D.  The one-thread-per-channel kernels above issued four times the instructions for
// the same bytes (forward 237 -> see profiles/r03_comer_*; they remain for D % 4 != 0).
__device__ __forceinline__ float4 msda_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 msda_ld4(const __half* p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// The four corner rows of a bilinear sample, loaded UNCONDITIONALLY from clamped addresses and zeroed afterwards where the corner
// (or, `in` false, the whole sample) lies outside the map.  A bounds BRANCH around each load made hipcc put `s_waitcnt vmcnt(0)`
// behind every one of them: 4 x nL x P dependent-latency loads per thread (round 4: found in the ISA, not in a profile).
template <typename VT>
__device__ __forceinline__ void msda_corners(const VT* __restrict__ vl, int y0, int x0, int H, int W, long MD, bool in, float4& v00,
                                             float4& v01, float4& v10, float4& v11) {
    const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
    const float4 a = msda_ld4(vl + ((long)ya * W + xa) * MD), b = msda_ld4(vl + ((long)ya * W + xb) * MD);
    const float4 c = msda_ld4(vl + ((long)yb * W + xa) * MD), d = msda_ld4(vl + ((long)yb * W + xb) * MD);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool top = in && y0 >= 0, bot = in && y0 + 1 < H, lef = x0 >= 0, rig = x0 + 1 < W;
    v00 = (top && lef) ? a : z;
    v01 = (top && rig) ? b : z;
    v10 = (bot && lef) ? c : z;
    v11 = (bot && rig) ? d : z;
}

// VT: value as f32 or f16 (f16 halves the gather traffic that bounds these kernels: 4 corners x nL*P samples per query and head)
template <typename VT>
__global__ __launch_bounds__(256) void msda_fwd4_kernel(const VT* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, float* __restrict__ out, __half* __restrict__ out16,
                                MsdaShapes sh, int S, long NQ, int Lq, int M, int D, int P) {
    const int tpq = (M * D) >> 2, j = threadIdx.x % tpq;
    const long nq = (long)blockIdx.x * (256 / tpq) + threadIdx.x / tpq;
    if (nq >= NQ) return;
    const int n = (int)(nq / Lq), dq = D >> 2, m = j / dq, d4 = (j - m * dq) * 4;
    const VT* vb = value + (long)n * S * M * D + (long)m * D + d4;
    const float* lb = loc + ((nq * M + m) * sh.n_levels) * P * 2;
    const float* ab = attn + ((nq * M + m) * sh.n_levels) * P;
    const long MD = (long)M * D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < sh.n_levels; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const VT* vl = vb + (long)sh.start[l] * MD;
#pragma unroll 4
        for (int p = 0; p < P; ++p) {
            const float x = lb[(l * P + p) * 2] * W - 0.5f, y = lb[(l * P + p) * 2 + 1] * H - 0.5f;
            const float w = ab[l * P + p];
            {
                const bool in = y > -1.f && x > -1.f && y < H && x < W;
                const float yc = in ? y : 0.f, xc = in ? x : 0.f;       // (an outside sample contributes exact zeros)
                const int y0 = (int)floorf(yc), x0 = (int)floorf(xc);
                const float ly = yc - y0, lx = xc - x0, hy = 1.f - ly, hx = 1.f - lx;
                float4 v00, v01, v10, v11;
                msda_corners(vl, y0, x0, H, W, MD, in, v00, v01, v10, v11);
                const float w00 = w * hy * hx, w01 = w * hy * lx, w10 = w * ly * hx, w11 = w * ly * lx;
                acc.x += w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
                acc.y += w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
                acc.z += w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
                acc.w += w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
            }
        }
    }
    const long o = nq * MD + (long)m * D + d4;
    if (out) *reinterpret_cast<float4*>(out + o) = acc;
    if (out16) {
        __half h[4] = {__float2half(acc.x), __float2half(acc.y), __float2half(acc.z), __float2half(acc.w)};
        *reinterpret_cast<uint2*>(out16 + o) = *reinterpret_cast<const uint2*>(h);
    }
}

template <typename VT, typename GT>
__global__ __launch_bounds__(256) void msda_bwd4_kernel(const VT* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, const GT* __restrict__ gout,
                                float* __restrict__ gloc, float* __restrict__ gattn,
                                MsdaShapes sh, int S, long NQ, int Lq, int M, int D, int P) {
    const int tpq = (M * D) >> 2, j = threadIdx.x % tpq;
    const long nq = (long)blockIdx.x * (256 / tpq) + threadIdx.x / tpq;
    if (nq >= NQ) return;                           // (whole groups of D/4 lanes leave together: the shuffles below stay inside one)
    const int n = (int)(nq / Lq), dq = D >> 2, m = j / dq, d4 = (j - m * dq) * 4;
    const long MD = (long)M * D;
    const VT* vb = value + (long)n * S * MD + (long)m * D + d4;
    const float* lb = loc + ((nq * M + m) * sh.n_levels) * P * 2;
    const float* ab = attn + ((nq * M + m) * sh.n_levels) * P;
    const float4 go = msda_ld4(gout + nq * MD + (long)m * D + d4);
    for (int l = 0; l < sh.n_levels; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const VT* vl = vb + (long)sh.start[l] * MD;
#pragma unroll 4
        for (int p = 0; p < P; ++p) {
            const float x = lb[(l * P + p) * 2] * W - 0.5f, y = lb[(l * P + p) * 2 + 1] * H - 0.5f;
            const float w = ab[l * P + p];
            float gx = 0.f, gy = 0.f, ga = 0.f;
            {
                const bool in = y > -1.f && x > -1.f && y < H && x < W;
                const float yc = in ? y : 0.f, xc = in ? x : 0.f;       // (an outside sample contributes exact zeros)
                const int y0 = (int)floorf(yc), x0 = (int)floorf(xc);
                const float ly = yc - y0, lx = xc - x0, hy = 1.f - ly, hx = 1.f - lx;
                float4 v00, v01, v10, v11;
                msda_corners(vl, y0, x0, H, W, MD, in, v00, v01, v10, v11);
                // per channel: bilinear value, d/dx, d/dy; dotted with the output gradient of the lane's 4 channels
                const float b0 = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
                const float b1 = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
                const float b2 = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
                const float b3 = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
                ga = go.x * b0 + go.y * b1 + go.z * b2 + go.w * b3;
                const float dx0 = hy * (v01.x - v00.x) + ly * (v11.x - v10.x), dx1 = hy * (v01.y - v00.y) + ly * (v11.y - v10.y);
                const float dx2 = hy * (v01.z - v00.z) + ly * (v11.z - v10.z), dx3 = hy * (v01.w - v00.w) + ly * (v11.w - v10.w);
                const float dy0 = hx * (v10.x - v00.x) + lx * (v11.x - v01.x), dy1 = hx * (v10.y - v00.y) + lx * (v11.y - v01.y);
                const float dy2 = hx * (v10.z - v00.z) + lx * (v11.z - v01.z), dy3 = hx * (v10.w - v00.w) + lx * (v11.w - v01.w);
                gx = w * W * (go.x * dx0 + go.y * dx1 + go.z * dx2 + go.w * dx3);     // d/dloc_x (pixel x = loc_x*W - 0.5)
                gy = w * H * (go.x * dy0 + go.y * dy1 + go.z * dy2 + go.w * dy3);
            }
            gx = head_sum(gx, dq); gy = head_sum(gy, dq); ga = head_sum(ga, dq);
            if (d4 == 0) {
                const long e = (nq * M + m) * sh.n_levels * P + l * P + p;
                gloc[e * 2] = gx;
                gloc[e * 2 + 1] = gy;
                gattn[e] = ga;
            }
        }
    }
}

// ---- fused forms (comer_engine.py): the sampling locations and the soft-maxed attention weights are computed INSIDE the
// kernels from the raw rows ow (N*Lq, ld) = [M*T*2 offsets | M*T logits] of the fused sampling_offsets | attention_weights GEMM
// (T = NL*P, compile time), their biases and the reference points: loc = ref + (off + b_off) / (W_l, H_l), attn = softmax_T.
// The forward still writes loc / attn once (the value-gradient bucket / gather kernels and the backward read them); the
// backward turns the location / weight gradients into the gradient of the raw row in registers (soft-max backward needs all
// T weight gradients of a head: they are kept by every lane after the head reduction) and writes it as the f16 operand
// dow16 (N*Lq, ld) of the input-gradient / weight-gradient GEMMs, padding columns zeroed: no gloc / gattn tensors, no
// separate prep kernels.
template <typename VT, int NL, int P>
__global__ __launch_bounds__(256) void msda_fwd4f_kernel(const VT* __restrict__ value, const float* __restrict__ ow,
                                const float* __restrict__ boff, const float* __restrict__ baw, const float* __restrict__ ref,
                                float* __restrict__ loc, float* __restrict__ attn, float* __restrict__ out,
                                __half* __restrict__ out16, MsdaShapes sh, int S, long NQ, int Lq, int M, int D, int ld, int nl_ref) {
    constexpr int T = NL * P;
    const int tpq = (M * D) >> 2, j = threadIdx.x % tpq;
    const long nq = (long)blockIdx.x * (256 / tpq) + threadIdx.x / tpq;
    if (nq >= NQ) return;
    const int n = (int)(nq / Lq), q = (int)(nq - (long)n * Lq), dq = D >> 2, m = j / dq, d4 = (j - m * dq) * 4;
    const long MD = (long)M * D;
    const float* row = ow + nq * ld;
    const float* off = row + (long)m * T * 2;
    const float* lg = row + (long)M * T * 2 + (long)m * T;
    float aw[T], sx[T], sy[T];
    float mx = -3.4e38f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        aw[t] = lg[t] + (baw ? baw[m * T + t] : 0.f);
        mx = fmaxf(mx, aw[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        aw[t] = __expf(aw[t] - mx);
        sum += aw[t];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const float* rp = ref + ((long)q * nl_ref + (nl_ref > 1 ? l : 0)) * 2;
        const float rx = rp[0], ry = rp[1], iw = 1.f / sh.W[l], ih = 1.f / sh.H[l];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            aw[t] *= inv;
            sx[t] = rx + (off[t * 2] + (boff ? boff[(m * T + t) * 2] : 0.f)) * iw;
            sy[t] = ry + (off[t * 2 + 1] + (boff ? boff[(m * T + t) * 2 + 1] : 0.f)) * ih;
        }
    }
    if (d4 == 0) {
        const long e = (nq * M + m) * T;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            loc[(e + t) * 2] = sx[t];
            loc[(e + t) * 2 + 1] = sy[t];
            attn[e + t] = aw[t];
        }
    }
    const VT* vb = value + (long)n * S * MD + (long)m * D + d4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const VT* vl = vb + (long)sh.start[l] * MD;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            const float x = sx[t] * W - 0.5f, y = sy[t] * H - 0.5f, w = aw[t];
            {
                const bool in = y > -1.f && x > -1.f && y < H && x < W;
                const float yc = in ? y : 0.f, xc = in ? x : 0.f;       // (an outside sample contributes exact zeros)
                const int y0 = (int)floorf(yc), x0 = (int)floorf(xc);
                const float ly = yc - y0, lx = xc - x0, hy = 1.f - ly, hx = 1.f - lx;
                float4 v00, v01, v10, v11;
                msda_corners(vl, y0, x0, H, W, MD, in, v00, v01, v10, v11);
                const float w00 = w * hy * hx, w01 = w * hy * lx, w10 = w * ly * hx, w11 = w * ly * lx;
                acc.x += w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
                acc.y += w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
                acc.z += w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
                acc.w += w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
            }
        }
    }
    const long o = nq * MD + (long)m * D + d4;
    if (out) *reinterpret_cast<float4*>(out + o) = acc;
    if (out16) {
        __half h[4] = {__float2half(acc.x), __float2half(acc.y), __float2half(acc.z), __float2half(acc.w)};
        *reinterpret_cast<uint2*>(out16 + o) = *reinterpret_cast<const uint2*>(h);
    }
}

template <typename VT, typename GT, int NL, int P>
__global__ __launch_bounds__(256) void msda_bwd4f_kernel(const VT* __restrict__ value, const float* __restrict__ loc,
                                const float* __restrict__ attn, const GT* __restrict__ gout, __half* __restrict__ dow16,
                                MsdaShapes sh, int S, long NQ, int Lq, int M, int D, int ld) {
    constexpr int T = NL * P;
    const int tpq = (M * D) >> 2, j = threadIdx.x % tpq;
    const long nq = (long)blockIdx.x * (256 / tpq) + threadIdx.x / tpq;
    if (nq >= NQ) return;
    const int n = (int)(nq / Lq), dq = D >> 2, m = j / dq, d4 = (j - m * dq) * 4;
    const long MD = (long)M * D;
    const VT* vb = value + (long)n * S * MD + (long)m * D + d4;
    const float* lb = loc + (nq * M + m) * T * 2;
    const float* ab = attn + (nq * M + m) * T;
    const float4 go = msda_ld4(gout + nq * MD + (long)m * D + d4);
    // Loads first, reductions last: all T locations / weights of the (query, head), then per level the 4 x P corner rows in flight
    // together, the un-reduced per-lane sums kept in registers; the 3 T head reductions run afterwards as ONE loop over the shuffle
    // distance with 3 T independent shuffles per step.  (The sample-after-sample form -- location load, wait, four corner loads,
    // wait, three shuffle loops -- exposed two memory latencies per sample: 24 per thread at T = 12.)
    float aw[T], ga[T], gx[T], gy[T], px[T], py[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        px[t] = lb[t * 2];
        py[t] = lb[t * 2 + 1];
        aw[t] = ab[t];
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int H = sh.H[l], W = sh.W[l];
        const VT* vl = vb + (long)sh.start[l] * MD;
        float4 c00[P], c01[P], c10[P], c11[P];
        float fy[P], fx[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            const float x = px[t] * W - 0.5f, y = py[t] * H - 0.5f;
            const bool in = y > -1.f && x > -1.f && y < H && x < W;
            const float yc = in ? y : 0.f, xc = in ? x : 0.f;       // (an outside sample contributes exact zeros)
            const int y0 = (int)floorf(yc), x0 = (int)floorf(xc);
            fy[p] = yc - y0;
            fx[p] = xc - x0;
            msda_corners(vl, y0, x0, H, W, MD, in, c00[p], c01[p], c10[p], c11[p]);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            const float w = aw[t];
            const float ly = fy[p], lx = fx[p], hy = 1.f - ly, hx = 1.f - lx;
            const float4 v00 = c00[p], v01 = c01[p], v10 = c10[p], v11 = c11[p];
            const float b0 = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            const float b1 = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            const float b2 = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            const float b3 = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
            ga[t] = go.x * b0 + go.y * b1 + go.z * b2 + go.w * b3;
            const float dx0 = hy * (v01.x - v00.x) + ly * (v11.x - v10.x), dx1 = hy * (v01.y - v00.y) + ly * (v11.y - v10.y);
            const float dx2 = hy * (v01.z - v00.z) + ly * (v11.z - v10.z), dx3 = hy * (v01.w - v00.w) + ly * (v11.w - v10.w);
            const float dy0 = hx * (v10.x - v00.x) + lx * (v11.x - v01.x), dy1 = hx * (v10.y - v00.y) + lx * (v11.y - v01.y);
            const float dy2 = hx * (v10.z - v00.z) + lx * (v11.z - v01.z), dy3 = hx * (v10.w - v00.w) + lx * (v11.w - v01.w);
            // d out / d offset = d out / d loc / (W, H) and d loc = pixel / (W, H): the two level sizes cancel
            gx[t] = w * (go.x * dx0 + go.y * dx1 + go.z * dx2 + go.w * dx3);
            gy[t] = w * (go.x * dy0 + go.y * dy1 + go.z * dy2 + go.w * dy3);
        }
    }
    for (int o = dq >> 1; o > 0; o >>= 1) {          // sums over the D / 4 lanes of the head (same order as head_sum)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            gx[t] += __shfl_xor(gx[t], o, 64);
            gy[t] += __shfl_xor(gy[t], o, 64);
            ga[t] += __shfl_xor(ga[t], o, 64);
        }
    }
    __half* drow = dow16 + nq * ld;
    if (d4 == 0) {
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) dot = fmaf(aw[t], ga[t], dot);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            drow[(m * T + t) * 2] = __float2half(gx[t]);
            drow[(m * T + t) * 2 + 1] = __float2half(gy[t]);
            drow[M * T * 2 + m * T + t] = __float2half(aw[t] * (ga[t] - dot));
        }
    }
    for (int c = 3 * M * T + j; c < ld; c += tpq) drow[c] = __float2half(0.f);      // K padding of the gradient GEMMs
}

// max |gout| as the bit pattern of a non-negative float (unsigned compare == float compare); *gmax zeroed by the caller.
// 16-byte loads, four in flight per thread (n is a multiple of 16 / sizeof(GT), g 16-byte aligned: checked by the launcher).
template <typename GT>
__global__ __launch_bounds__(256) void msda_absmax_kernel(const GT* __restrict__ g, unsigned int* __restrict__ gmax, long n) {
    constexpr int VEC = 16 / (int)sizeof(GT);
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    const long n4 = n / VEC, stride = (long)gridDim.x * 256;
    float m = 0.f;
    auto fold = [&](const uint4& u) {
        if constexpr (sizeof(GT) == 2) {
            const __half2* h = reinterpret_cast<const __half2*>(&u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 f = __half22float2(h[k]);
                m = fmaxf(m, fmaxf(fabsf(f.x), fabsf(f.y)));
            }
        } else {
            const float* f = reinterpret_cast<const float*>(&u);
            m = fmaxf(fmaxf(m, fabsf(f[0])), fmaxf(fabsf(f[1]), fmaxf(fabsf(f[2]), fabsf(f[3]))));
        }
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const uint4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
        fold(a); fold(b); fold(c); fold(d);
    }
    for (; i < n4; i += stride) fold(g4[i]);
    __shared__ float red[16];
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(gmax, __float_as_uint(m));
}

// grad_value as a BUCKETED GATHER, two kernels:
//   msda_bucket_kernel (one workgroup per (level, head, image)):
//     pass 1  counts, per pixel of the level, the (sample, corner) pairs that land on it   (one LDS integer add per pair --
//             not one per pair AND channel, which made the first LDS-scatter version LDS-atomic bound: 774 us per launch);
//     scan    turns the counts into bucket offsets;
//     pass 2  writes every pair's id into its pixel's bucket (the order inside a bucket depends on the wave scheduling);
//   msda_gather_kernel (one group of D lanes per pixel, all pixels of all levels / heads / images in parallel):
//     walks the pixel's bucket, recomputes the bilinear weight of each pair and accumulates
//     attn * weight * grad_out[q, m, d] in 64-bit FIXED POINT (2^24 / max|grad_out| per product): integer sums do not depend on the
//     order of the bucket, so the result is bit-reproducible, and every element of grad_value is written exactly once
//     (no atomics on HBM at all).
// LDS of the bucket kernel: 2 * H_l*W_l ints -> levels up to 16384 pixels.
// ws (ints): [N*M*S] bucket starts | [N*M*S] bucket sizes | per (image, head, level) Lq*P*4 bucket entries of TWO ints:
// {row of grad_out = query * M + head, bit pattern of bilinear weight * attention weight}.  (Round 4: the entries were pair ids,
// and the gather re-derived row and weight from loc / attn with two dependent random loads per pair.)
#define MSDA_VT 1024
#define MSDA_UN 8              // samples per thread whose loads are in flight together
__global__ __launch_bounds__(MSDA_VT) void msda_bucket_kernel(const float* __restrict__ loc, const float* __restrict__ attn,
                                                           int* __restrict__ ws, MsdaShapes sh, int S, int Lq, int M, int P, int N) {
    extern __shared__ int sm[];
    const int l = blockIdx.x, m = blockIdx.y, n = blockIdx.z;
    const int H = sh.H[l], W = sh.W[l], HW = H * W;
    int* cnt = sm;                 // [HW] counts, then exclusive offsets
    int* cur = sm + HW;            // [HW] fill cursors
    __shared__ int wsum[MSDA_VT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long nsamp = (long)Lq * P;
    const long NMS = (long)N * M * S;
    int* gstart = ws + ((long)n * M + m) * S + sh.start[l];
    int* gsize = gstart + NMS;
    int* list = ws + 2 * NMS + (((long)n * M + m) * sh.n_levels + l) * nsamp * 8;
    for (int i = tid; i < HW; i += MSDA_VT) { cnt[i] = 0; cur[i] = 0; }
    __syncthreads();
    // (both passes: the locations [and weights] of MSDA_UN samples per thread are requested together from clamped addresses --
    //  one sample per iteration exposed a scattered-load latency per sample, 21 per pass at 5 376 queries x 4 points)
    auto loc_of = [&](long e) {
        const int q = (int)(e / P), p = (int)(e - (long)q * P);
        return loc + ((((long)n * Lq + q) * M + m) * sh.n_levels + l) * P * 2 + p * 2;
    };
    for (long e0 = tid; e0 < nsamp; e0 += MSDA_UN * MSDA_VT) {
        float lx_[MSDA_UN], ly_[MSDA_UN];
#pragma unroll
        for (int u = 0; u < MSDA_UN; ++u) {
            const float* lb = loc_of(min(e0 + (long)u * MSDA_VT, nsamp - 1));
            lx_[u] = lb[0];
            ly_[u] = lb[1];
        }
#pragma unroll
        for (int u = 0; u < MSDA_UN; ++u) {
            const float x = lx_[u] * W - 0.5f, y = ly_[u] * H - 0.5f;
            if (e0 + (long)u * MSDA_VT >= nsamp || !(y > -1.f && x > -1.f && y < H && x < W)) continue;
            const int y0 = (int)floorf(y), x0 = (int)floorf(x);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int px = x0 + (c & 1), py = y0 + (c >> 1);
                if (px >= 0 && px < W && py >= 0 && py < H) atomicAdd(&cnt[py * W + px], 1);
            }
        }
    }
    __syncthreads();
    // exclusive scan of cnt[0..HW): each thread a contiguous chunk, wave scan of the chunk sums, then the wave sums
    const int chunk = (HW + MSDA_VT - 1) / MSDA_VT;
    const int c0 = min(tid * chunk, HW), c1 = min(c0 + chunk, HW);
    int tsum = 0;
    for (int i = c0; i < c1; ++i) tsum += cnt[i];
    int incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
    int run = base + incl - tsum;
    for (int i = c0; i < c1; ++i) { const int v = cnt[i]; cnt[i] = run; gstart[i] = run; gsize[i] = v; run += v; }
    __syncthreads();
    for (long e0 = tid; e0 < nsamp; e0 += MSDA_UN * MSDA_VT) {
        float lx_[MSDA_UN], ly_[MSDA_UN], aw_[MSDA_UN];
#pragma unroll
        for (int u = 0; u < MSDA_UN; ++u) {
            const long e = min(e0 + (long)u * MSDA_VT, nsamp - 1);
            const float* lb = loc_of(e);
            lx_[u] = lb[0];
            ly_[u] = lb[1];
            const int q = (int)(e / P), p = (int)(e - (long)q * P);
            aw_[u] = attn[((((long)n * Lq + q) * M + m) * sh.n_levels + l) * P + p];
        }
#pragma unroll
        for (int u = 0; u < MSDA_UN; ++u) {
            const long e = e0 + (long)u * MSDA_VT;
            const float x = lx_[u] * W - 0.5f, y = ly_[u] * H - 0.5f;
            if (e >= nsamp || !(y > -1.f && x > -1.f && y < H && x < W)) continue;
            const int q = (int)(e / P);
            const int y0 = (int)floorf(y), x0 = (int)floorf(x);
            const float lx = x - floorf(x), ly = y - floorf(y);
            const long qm = ((long)n * Lq + q) * M + m;
            const float aw = aw_[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int px = x0 + (c & 1), py = y0 + (c >> 1);
                if (px >= 0 && px < W && py >= 0 && py < H) {
                    const int pix = py * W + px;
                    const int pos = cnt[pix] + atomicAdd(&cur[pix], 1);
                    const float wgt = ((c & 1) ? lx : 1.f - lx) * ((c >> 1) ? ly : 1.f - ly) * aw;
                    *reinterpret_cast<int2*>(list + 2 * (long)pos) = make_int2((int)qm, __float_as_int(wgt));
                }
            }
        }
    }
}

// grid: (blocks of all levels) x M x N workgroups of 256 threads.  A pixel is served by GL lanes, GL chosen PER LEVEL by the
// launcher from the level's mean bucket length (GatherPlan): the GL lanes work as SL = GL / CL pair SLOTS of CL = D / VEC lanes
// (VEC = 16 / sizeof(GT) channels per lane): a slot fetches one pair's grad_out row (D values) as 16 bytes per lane -- 512 B per
// 32 lanes at fp16, where the round-3 form (one value per lane, one pair per group and instruction) moved 64 B; the bucket entries
// already hold the row and the weight (no dependent loads).  With D lanes per pixel on every level (the first round-4 form) a
// level whose buckets hold ~4 entries (the 64 x 64 map of the three-level direction: 76 % of its pixels) kept 7 of 8 slots idle
// and paid the three dependent latencies of a pixel (bucket bounds -> entries -> rows) for 4 useful loads: 1.6 TB/s algorithmic,
// against 7.9 TB/s for the one-level direction with ~84 entries per bucket; 4 lanes per pixel there put 8x the pixels in flight.
// A lane accumulates its VEC channels over its slot's pairs in 64-bit FIXED POINT (2^24 / max|grad_out| per product, 64-bit
// sums); the slots are added at the end -- integer sums, so the result does not depend on the bucket order, on the slot a pair
// falls into or on GL (bit-reproducible).
struct GatherPlan {
    int blk0[MSDA_MAX_LEVELS + 1];       // first workgroup of each level (+ total)
    int gl[MSDA_MAX_LEVELS];             // lanes per pixel of the level: CL, 2 CL, ... <= D
};

template <typename GT>
__global__ __launch_bounds__(256) void msda_gather_kernel(const GT* __restrict__ gout, const unsigned int* __restrict__ gmax,
                                                           const int* __restrict__ ws, float* __restrict__ gvalue,
                                                           __half* __restrict__ gvalue16, MsdaShapes sh, GatherPlan plan,
                                                           int S, int Lq, int M, int D, int P, int N) {
    constexpr int VEC = 16 / sizeof(GT);                 // channels per lane
    const int m = blockIdx.y, n = blockIdx.z;
    int l = 0;
    while (l + 1 < sh.n_levels && (int)blockIdx.x >= plan.blk0[l + 1]) ++l;        // (uniform: a workgroup serves one level)
    const int GL = plan.gl[l];
    const int grp = threadIdx.x / GL, d = threadIdx.x - grp * GL;
    const int pix = ((int)blockIdx.x - plan.blk0[l]) * (256 / GL) + grp;
    if (pix >= sh.H[l] * sh.W[l]) return;                    // (a whole group leaves together: GL divides 64)
    const int s = sh.start[l] + pix;                         // pixel index over all levels
    const int CL = D / VEC, SL = GL / CL;                    // lanes per pair, pair slots of the pixel
    const int slot = d / CL, ch = (d - slot * CL) * VEC;     // this lane's pair slot and first channel
    const long nsamp = (long)Lq * P;
    const long NMS = (long)N * M * S;
    const long pm = ((long)n * M + m) * S + s;
    const int b0 = ws[pm], nb = ws[NMS + pm];
    const int2* list = reinterpret_cast<const int2*>(ws + 2 * NMS + (((long)n * M + m) * sh.n_levels + l) * nsamp * 8) + b0;
    const float gm = __uint_as_float(*gmax);
    // fixed point with 2^24 / max|gout| per unit: |weight| <= 1, so a product is at most 2^24 and a lane's <= 4 * 16 / VEC
    // products of one round of D bucket entries fit an int32 (v_rndne + v_cvt_i32 + one 32-bit add per product; the float ->
    // int64 conversion of the round-3 form was ~10 instructions and made the kernel VALU-bound: 235 us for 6-11 M pairs); the
    // int32 partial sums are widened into the 64-bit accumulators once per round.  Quantum = max|gout| * 2^-24 per product.
    const float scale = gm > 0.f ? 16777216.0f / gm : 0.f;
    long long acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    for (int k0 = 0; k0 < nb; k0 += GL) {
        int part[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[v] = 0;
        // lane j holds entry k0 + j of the bucket (coalesced 8-byte reads; clamped index, masked weight: no branch around the
        // load); rows / weights travel to their slots by shuffles
        const int2 e = list[min(k0 + d, nb - 1)];
        const int rowj = e.x;
        const float wj = __int_as_float(e.y) * scale * (k0 + d < nb ? 1.f : 0.f);
        const int cntk = min(GL, nb - k0);
        for (int k = 0; k < cntk; k += 4 * SL) {             // 4 rounds of SL pairs: four 16-byte loads in flight per lane
            GT gk[4][VEC];
            float wk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + u * SL + slot;
                const int kc = kk < cntk ? kk : 0;            // (clamped: the weight of a padded slot is zeroed)
                const long row = (long)__shfl(rowj, kc, GL) * D;
                const float wsh = __shfl(wj, kc, GL);     // (every lane of the group takes part: the condition below differs per slot,
                wk[u] = kk < cntk ? wsh : 0.f;            //  and a lane skipped by a conditional shuffle cannot serve as a source)
                *reinterpret_cast<u32x4*>(gk[u]) = *reinterpret_cast<const u32x4*>(gout + row + ch);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) part[v] += __float2int_rn((float)gk[u][v] * wk[u]);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += part[v];
    }
    // add the SL slots: lanes d, d + CL, d + 2 CL, ... hold the same channels
    for (int o = CL; o < GL; o <<= 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int lo = __shfl_xor((int)(acc[v] & 0xffffffffLL), o, GL);
            const int hi = __shfl_xor((int)(acc[v] >> 32), o, GL);
            acc[v] += ((long long)hi << 32) | (unsigned int)lo;
        }
    }
    if (slot == 0) {
        const float inv = gm > 0.f ? gm / 16777216.0f : 0.f;
        const long o = (((long)n * S + s) * M + m) * D + ch;
        float gv[VEC];
        __half hv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            gv[v] = gm > 0.f ? (float)acc[v] * inv : 0.f;
            hv[v] = __float2half(gv[v]);
        }
        if (gvalue) {
#pragma unroll
            for (int v = 0; v < VEC; v += 4) *reinterpret_cast<float4*>(gvalue + o + v) = make_float4(gv[v], gv[v + 1], gv[v + 2], gv[v + 3]);
        }
        if (gvalue16) {      // the value projection's gradient operand
            if constexpr (VEC == 8) *reinterpret_cast<u32x4*>(gvalue16 + o) = *reinterpret_cast<u32x4*>(hv);
            else *reinterpret_cast<u32x2*>(gvalue16 + o) = *reinterpret_cast<u32x2*>(hv);
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

extern "C" int wc_msda_fwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                             const float* attn, float* out, void* out16, int N, int Lq, int M, int D, int P, void* stream);

extern "C" int wc_msda_fwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                           float* out, int N, int Lq, int M, int D, int P, void* stream) {
    return wc_msda_fwd_h(value, 0, h_shapes, n_levels, loc, attn, out, nullptr, N, Lq, M, D, P, stream);
}

// value f32 or f16 (value_is_f16: needs the channel-quad form, D % 4 == 0); out (f32) and / or out16 (f16): (N, Lq, M*D)
extern "C" int wc_msda_fwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                             const float* attn, float* out, void* out16, int N, int Lq, int M, int D, int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && loc && attn && (out || out16) && N > 0 && Lq > 0 && M > 0 && P > 0, "wc_msda_fwd: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0, "wc_msda_fwd: 1..8 levels with positive sizes");
    WC_CHECK_ARG((D == 16 || D == 32 || D == 64) && M * D <= 1024 && (M * D) % 64 == 0,
                 "wc_msda_fwd: head dim 16/32/64, heads*dim a multiple of 64 and <= 1024");
    const int tpq = M * D / 4;
    const bool quad = D % 4 == 0 && 256 % tpq == 0 && tpq % (D / 4) == 0 && ((uintptr_t)value % 16 == 0) &&
                      (!out || (uintptr_t)out % 16 == 0) && (!out16 || (uintptr_t)out16 % 8 == 0);
    WC_CHECK_ARG(quad || !value_is_f16, "wc_msda_fwd: an f16 value tensor needs the channel-quad form (M*D/4 dividing 256, aligned buffers)");
    if (quad) {
        const long NQ = (long)N * Lq;
        const dim3 gd((unsigned)wc_cdiv(NQ, 256 / tpq));
        if (value_is_f16)
            hipLaunchKernelGGL(msda_fwd4_kernel<__half>, gd, dim3(256), 0, (hipStream_t)stream, (const __half*)value, loc, attn, out,
                               (__half*)out16, sh, S, NQ, Lq, M, D, P);
        else
            hipLaunchKernelGGL(msda_fwd4_kernel<float>, gd, dim3(256), 0, (hipStream_t)stream, (const float*)value, loc, attn, out,
                               (__half*)out16, sh, S, NQ, Lq, M, D, P);
        WC_LAUNCH_CHECK("msda_fwd4_kernel");
        return WC_OK;
    }
    hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)((long)N * Lq)), dim3(M * D), 0, (hipStream_t)stream, (const float*)value, loc,
                       attn, out, (__half*)out16, sh, S, Lq, M, D, P);
    WC_LAUNCH_CHECK("msda_fwd_kernel");
    return WC_OK;
}

// gvalue needs no initialisation (every element is written exactly once); gmax: 1 x u32 workspace;
// ws: N*M*S*2 + N*M*n_levels*Lq*P*4 ints (bucket starts, sizes, and the per-pixel buckets of (sample, corner) ids).
extern "C" int wc_msda_bwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                             const float* attn, const void* gout, int gout_is_f16, float* gvalue, void* gvalue16, float* gloc,
                             float* gattn, void* gmax, void* ws, int N, int Lq, int M, int D, int P, void* stream);

extern "C" int wc_msda_bwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                           const float* gout, float* gvalue, float* gloc, float* gattn, void* gmax, void* ws, int N, int Lq,
                           int M, int D, int P, void* stream) {
    return wc_msda_bwd_h(value, 0, h_shapes, n_levels, loc, attn, gout, 0, gvalue, nullptr, gloc, gattn, gmax, ws, N, Lq, M, D, P, stream);
}

template <typename GT>
static int msda_bwd_value(const float* loc, const float* attn, const GT* gout, float* gvalue, void* gvalue16, void* gmax, void* ws,
                          const MsdaShapes& sh, int n_levels, int S, int N, int Lq, int M, int D, int P, hipStream_t st) {
    hipMemsetAsync(gmax, 0, sizeof(unsigned int), st);
    const long ng = (long)N * Lq * M * D;
    WC_CHECK_ARG((uintptr_t)gout % 16 == 0 && D % (16 / (int)sizeof(GT)) == 0, "wc_msda_bwd: grad_out must be 16-byte aligned");
    const long nv = ng / (16 / (long)sizeof(GT)) / 256 / 4 + 1;
    hipLaunchKernelGGL(msda_absmax_kernel<GT>, dim3((unsigned)(nv > 2048 ? 2048 : nv)), dim3(256), 0, st, gout, (unsigned int*)gmax, ng);
    WC_LAUNCH_CHECK("msda_absmax_kernel");
    int maxhw = 0;
    for (int l = 0; l < n_levels; ++l) maxhw = sh.H[l] * sh.W[l] > maxhw ? sh.H[l] * sh.W[l] : maxhw;
    WC_CHECK_ARG(maxhw <= 16384 && ws, "wc_msda_bwd: a level may have at most 16384 pixels; ws workspace missing");
    WC_CHECK_ARG(((uintptr_t)gout | (uintptr_t)gvalue | (uintptr_t)gvalue16 | (uintptr_t)ws) % 16 == 0 && D % (16 / (int)sizeof(GT)) == 0,
                 "wc_msda_bwd: grad_out / grad_value / ws must be 16-byte aligned (16-byte gathers)");
    WC_CHECK_ARG(D <= 64 && (D & (D - 1)) == 0 && D >= 16 / (int)sizeof(GT),
                 "wc_msda_bwd: the head width must be a power of two of at most 64 (lane groups of the gather)");
    static bool attr_set = false;
    if (!attr_set) {
        WC_CHECK_ARG(hipFuncSetAttribute((const void*)msda_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess,
                     "wc_msda_bwd: cannot reserve 128 KiB of LDS");
        attr_set = true;
    }
    const int prk = wc_prof_begin(st);
    hipLaunchKernelGGL(msda_bucket_kernel, dim3(n_levels, M, N), dim3(MSDA_VT), (size_t)maxhw * 2 * sizeof(int), st, loc, attn, (int*)ws,
                       sh, S, Lq, M, P, N);
    // two reads of the locations, one of the weights, one 8-byte entry per (sample, corner) written
    wc_prof_end2(prk, n_levels == 1 ? "msda_bucket_kernel<1>" : "msda_bucket_kernel<n>", 0.0,
                 (double)N * Lq * M * n_levels * P * (2 * 8 + 4 + 4 * 8), st);
    WC_LAUNCH_CHECK("msda_bucket_kernel");
    const int prg = wc_prof_begin(st);
    // lanes per pixel, per level: the smallest power-of-two multiple of CL that covers the level's mean bucket length
    GatherPlan plan;
    const int CL = D / (16 / (int)sizeof(GT));
    plan.blk0[0] = 0;
    for (int l = 0; l < n_levels; ++l) {
        const long hw = (long)sh.H[l] * sh.W[l], mean = ((long)Lq * P * 4 + hw - 1) / hw;
        int gl = CL;
        while (gl < D && gl < mean) gl <<= 1;
        plan.gl[l] = gl;
        plan.blk0[l + 1] = plan.blk0[l] + (int)wc_cdiv(hw, 256 / gl);
    }
    hipLaunchKernelGGL(msda_gather_kernel<GT>, dim3(plan.blk0[n_levels], M, N), dim3(256), 0, st, gout,
                       (const unsigned int*)gmax, (const int*)ws, gvalue, (__half*)gvalue16, sh, plan, S, Lq, M, D, P, N);
    // one dh-row of the output gradient per (sample, corner) pair (all corners inside: the upper bound) + the pair id, its
    // location / weight (12 B) + the value-gradient rows written
    wc_prof_end2(prg, n_levels == 1 ? "msda_gather_kernel<1>" : "msda_gather_kernel<n>", 0.0,
                 (double)N * Lq * M * n_levels * P * 4 * ((double)D * sizeof(GT) + 8) +
                     (double)N * S * M * D * ((gvalue ? 4 : 0) + (gvalue16 ? 2 : 0)), st);
    WC_LAUNCH_CHECK("msda_gather_kernel");
    return WC_OK;
}

// value / gout f32 or f16 (the f16 forms need the channel-quad kernels); gvalue (f32) and / or gvalue16 (f16)
extern "C" int wc_msda_bwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                             const float* attn, const void* gout, int gout_is_f16, float* gvalue, void* gvalue16, float* gloc,
                             float* gattn, void* gmax, void* ws, int N, int Lq, int M, int D, int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && loc && attn && gout && (gvalue || gvalue16) && gloc && gattn && gmax && N > 0 && Lq > 0 && M > 0 && P > 0,
                 "wc_msda_bwd: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0, "wc_msda_bwd: 1..8 levels with positive sizes");
    WC_CHECK_ARG((D == 16 || D == 32 || D == 64) && M * D <= 1024 && (M * D) % 64 == 0 && M <= 65535 && N <= 65535,
                 "wc_msda_bwd: head dim 16/32/64, heads*dim a multiple of 64 and <= 1024");
    hipStream_t st = (hipStream_t)stream;
    const int tpq = M * D / 4;
    const bool quad = D % 4 == 0 && 256 % tpq == 0 && ((uintptr_t)value % 16 == 0) && ((uintptr_t)gout % 16 == 0);
    WC_CHECK_ARG(quad || (!value_is_f16 && !gout_is_f16), "wc_msda_bwd: f16 value / gout need the channel-quad form");
    if (quad) {
        const long NQ = (long)N * Lq;
        const dim3 gd((unsigned)wc_cdiv(NQ, 256 / tpq));
#define MSDA_BWD4(VT, GT)                                                                                            \
        hipLaunchKernelGGL((msda_bwd4_kernel<VT, GT>), gd, dim3(256), 0, st, (const VT*)value, loc, attn, (const GT*)gout, gloc, \
                           gattn, sh, S, NQ, Lq, M, D, P)
        if (value_is_f16) { if (gout_is_f16) MSDA_BWD4(__half, __half); else MSDA_BWD4(__half, float); }
        else { if (gout_is_f16) MSDA_BWD4(float, __half); else MSDA_BWD4(float, float); }
#undef MSDA_BWD4
        WC_LAUNCH_CHECK("msda_bwd4_kernel");
    } else {
        hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)((long)N * Lq)), dim3(M * D), 0, st, (const float*)value, loc, attn,
                           (const float*)gout, gloc, gattn, sh, S, Lq, M, D, P);
        WC_LAUNCH_CHECK("msda_bwd_kernel");
    }
    if (gout_is_f16)
        return msda_bwd_value<__half>(loc, attn, (const __half*)gout, gvalue, gvalue16, gmax, ws, sh, n_levels, S, N, Lq, M, D, P, st);
    return msda_bwd_value<float>(loc, attn, (const float*)gout, gvalue, gvalue16, gmax, ws, sh, n_levels, S, N, Lq, M, D, P, st);
}

// Fused forms: see msda_fwd4f_kernel / msda_bwd4f_kernel.  (n_levels, P) must be (3, 4) or (1, 4) [the CTI configurations];
// wc_msda_fused_supported tells; value f16 or f32 as above.
extern "C" int wc_msda_fused_supported(int n_levels, int M, int D, int P) {
    const int tpq = M * D / 4;
    return (P == 4 && (n_levels == 1 || n_levels == 3) && D % 4 == 0 && tpq > 0 && 256 % tpq == 0 && tpq % (D / 4) == 0) ? 1 : 0;
}

extern "C" int wc_msda_fwd_f(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* ow, int ld,
                             const float* bias_off, const float* bias_aw, const float* ref, int nl_ref, float* loc, float* attn,
                             float* out, void* out16, int N, int Lq, int M, int D, int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && ow && ref && loc && attn && (out || out16) && N > 0 && Lq > 0 && M > 0, "wc_msda_fwd_f: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0 && wc_msda_fused_supported(n_levels, M, D, P) &&
                 ld >= 3 * M * n_levels * P && (nl_ref == 1 || nl_ref == n_levels) && (uintptr_t)value % 16 == 0 &&
                 (!out || (uintptr_t)out % 16 == 0) && (!out16 || (uintptr_t)out16 % 8 == 0),
                 "wc_msda_fwd_f: unsupported configuration (see wc_msda_fused_supported) or misaligned buffers");
    const long NQ = (long)N * Lq;
    const dim3 gd((unsigned)wc_cdiv(NQ, 256 / (M * D / 4)));
    hipStream_t st = (hipStream_t)stream;
    const int prf = wc_prof_begin(stream);
#define MSDA_F(VT, NL_)                                                                                                    \
    hipLaunchKernelGGL((msda_fwd4f_kernel<VT, NL_, 4>), gd, dim3(256), 0, st, (const VT*)value, ow, bias_off, bias_aw, ref, loc, attn, \
                       out, (__half*)out16, sh, S, NQ, Lq, M, D, ld, nl_ref)
    if (value_is_f16) { if (n_levels == 3) MSDA_F(__half, 3); else MSDA_F(__half, 1); }
    else { if (n_levels == 3) MSDA_F(float, 3); else MSDA_F(float, 1); }
#undef MSDA_F
    // algorithmic bytes (SURVEY.md section 8 row a-9 / VERDICT r03: nL * nP * 4 corner reads of dh values per (query, head)):
    // the corner rows + the offset / weight logits read + the fp16 / fp32 output row written
    wc_prof_end2(prf, n_levels == 3 ? "msda_fwd4f_kernel<3>" : "msda_fwd4f_kernel<1>", 0.0,
                 (double)NQ * M * ((double)n_levels * P * 4 * D * (value_is_f16 ? 2 : 4) + D * ((out16 ? 2 : 0) + (out ? 4 : 0))) +
                     (double)NQ * 3 * M * n_levels * P * 4, stream);
    WC_LAUNCH_CHECK("msda_fwd4f_kernel");
    return WC_OK;
}

extern "C" int wc_msda_bwd_f(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                             const float* attn, const void* gout, int gout_is_f16, float* gvalue, void* gvalue16, void* dow16,
                             int ld, void* gmax, void* ws, int N, int Lq, int M, int D, int P, void* stream) {
    MsdaShapes sh;
    int S = 0;
    WC_CHECK_ARG(value && h_shapes && loc && attn && gout && (gvalue || gvalue16) && dow16 && gmax && ws && N > 0 && Lq > 0 && M > 0 &&
                 M <= 65535 && N <= 65535, "wc_msda_bwd_f: bad argument");
    WC_CHECK_ARG(fill_shapes(&sh, h_shapes, n_levels, &S) == 0 && wc_msda_fused_supported(n_levels, M, D, P) &&
                 ld >= 3 * M * n_levels * P && (uintptr_t)value % 16 == 0 && (uintptr_t)gout % 16 == 0,
                 "wc_msda_bwd_f: unsupported configuration (see wc_msda_fused_supported) or misaligned buffers");
    const long NQ = (long)N * Lq;
    const dim3 gd((unsigned)wc_cdiv(NQ, 256 / (M * D / 4)));
    hipStream_t st = (hipStream_t)stream;
    const int prb = wc_prof_begin(stream);
#define MSDA_B(VT, GT, NL_)                                                                                                \
    hipLaunchKernelGGL((msda_bwd4f_kernel<VT, GT, NL_, 4>), gd, dim3(256), 0, st, (const VT*)value, loc, attn, (const GT*)gout, \
                       (__half*)dow16, sh, S, NQ, Lq, M, D, ld)
#define MSDA_B2(VT, GT) { if (n_levels == 3) MSDA_B(VT, GT, 3); else MSDA_B(VT, GT, 1); }
    if (value_is_f16) { if (gout_is_f16) MSDA_B2(__half, __half) else MSDA_B2(__half, float) }
    else { if (gout_is_f16) MSDA_B2(float, __half) else MSDA_B2(float, float) }
#undef MSDA_B2
#undef MSDA_B
    // the same corner rows + the output-gradient row read + the fp16 offset / weight gradient row written
    wc_prof_end2(prb, n_levels == 3 ? "msda_bwd4f_kernel<3>" : "msda_bwd4f_kernel<1>", 0.0,
                 (double)NQ * M * ((double)n_levels * P * 4 * D * (value_is_f16 ? 2 : 4) + D * (gout_is_f16 ? 2 : 4)) + (double)NQ * ld * 2, stream);
    WC_LAUNCH_CHECK("msda_bwd4f_kernel");
    if (gout_is_f16)
        return msda_bwd_value<__half>(loc, attn, (const __half*)gout, gvalue, gvalue16, gmax, ws, sh, n_levels, S, N, Lq, M, D, P, st);
    return msda_bwd_value<float>(loc, attn, (const float*)gout, gvalue, gvalue16, gmax, ws, sh, n_levels, S, N, Lq, M, D, P, st);
}
