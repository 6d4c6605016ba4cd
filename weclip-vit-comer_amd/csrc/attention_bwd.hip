// Flash-style attention backward for gfx950 (training path of the decoder blocks,
// reference autograd through clip/myAtt.py:21-64 as used by WeCLIP_model/Decoder/TransDecoder.py:63-85).
// No L x L tensor is stored: P is recomputed from the packed qkv and the forward's log-sum-exp.
//   dS = P * (dP - delta),  dP = dO V^T,  delta = rowsum(dO * O)
//   dq = dS K / sqrt(dh),   dk = dS^T q / sqrt(dh),   dv = P^T dO
// Two kernels, no atomics, deterministic:
//   attn_bwd_dq_kernel   : one wave per 32 queries (lane = query), streams 64-key tiles;
//                          S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T (dS^T accumulators as MFMA B operand).
//   attn_bwd_dkdv_kernel : one wave per 32 keys (lane = key), streams 64-query tiles;
//                          S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS (P / dS accumulators as B operand).
// q in `qkv` is pre-scaled by log2(e)/sqrt(dh) (exp2 domain), so dq = acc/sqrt(dh) and dk = acc/log2(e).
// Transposed operand copies (keys / queries contiguous, zero padded) come from head_transpose_kernel.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NEG_BIG (-1.0e30f)

// dst[b,h,d,l] = src[(b*L + l)*ld + col_off + h*DH + d], zero for l in [L, Lp)
// One launch for the three operands of a layer (blockIdx.y = operand * E/64 + column tile): q and k out of the packed
// in-projection output, dO; 16-byte global loads and stores (the element-wise form moved 2 bytes per lane: 9 launches of
// 15 us per step).
__global__ __launch_bounds__(256) void head_transpose_kernel(const __half* __restrict__ qkv, const __half* __restrict__ dO,
                                                              __half* __restrict__ qt, __half* __restrict__ kt,
                                                              __half* __restrict__ dot, int L, int Lp, int H, int DH) {
    __shared__ __half tile[64][66];
    const int E = H * DH, ct = E >> 6;
    const int which = blockIdx.y / ct, c0 = (blockIdx.y - which * ct) * 64;
    const __half* src = which == 2 ? dO : qkv;
    const long ld = which == 2 ? E : 3L * E;
    const int col_off = which == 1 ? E : 0;
    __half* dst = which == 0 ? qt : (which == 1 ? kt : dot);
    const int l0 = blockIdx.x * 64, b = blockIdx.z;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int i = threadIdx.x + 256 * p;
        const int r = i >> 3, cv = (i & 7) * 8;
        const int l = l0 + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (l < L) v = *reinterpret_cast<const u32x4*>(src + ((long)b * L + l) * ld + col_off + c0 + cv);
        unsigned* t32 = reinterpret_cast<unsigned*>(&tile[r][cv]);          // (r * 66 + cv) halves: 4-byte aligned
        t32[0] = v[0]; t32[1] = v[1]; t32[2] = v[2]; t32[3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int i = threadIdx.x + 256 * p;
        const int c = i >> 3, rv = (i & 7) * 8;
        __half o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = tile[rv + j][c];
        const int e = c0 + c, h = e / DH, d = e - h * DH;
        *reinterpret_cast<u32x4*>(dst + (((long)b * H + h) * DH + d) * Lp + l0 + rv) = *reinterpret_cast<const u32x4*>(o);
    }
}

__global__ __launch_bounds__(256) void attn_delta_rows_kernel(const __half* __restrict__ dO, const float* __restrict__ o32,
                                                               float* __restrict__ delta, int L, int H, int DH, long total) {
    // one wave per token row (b, q): coalesced E-wide reads, segmented shuffle reduction per head
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total) return;
    const int lane = threadIdx.x & 63;
    const int q = row % L;
    const long b = row / L;
    const int E = H * DH;
    const __half* a = dO + row * E;
    const float* bq = o32 + row * E;
    // 8 consecutive elements per lane (16-byte dO load, two 16-byte O loads): a head of DH elements lives in DH / 8
    // adjacent lanes, summed by 2-3 shuffles (the element-per-lane form: 2- and 4-byte loads, 5-6 shuffles per 64 elements)
    const int nch = E >> 3, lph = DH >> 3;             // 16-byte chunks per row, lanes per head
    for (int c0 = 0; c0 < nch; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < nch) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(a + c * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(bq + c * 8), b1 = *reinterpret_cast<const float4*>(bq + c * 8 + 4);
            const __half* ah = reinterpret_cast<const __half*>(&av);
            s = __half2float(ah[0]) * b0.x + __half2float(ah[1]) * b0.y + __half2float(ah[2]) * b0.z + __half2float(ah[3]) * b0.w +
                __half2float(ah[4]) * b1.x + __half2float(ah[5]) * b1.y + __half2float(ah[6]) * b1.z + __half2float(ah[7]) * b1.w;
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        if (lph == 8) s += __shfl_xor(s, 4, 64);
        if (c < nch && (c & (lph - 1)) == 0) delta[(b * H + c / lph) * L + q] = s;
    }
}

__device__ __forceinline__ void store_split4(__half* hi, __half* lo, const float* v) {
    __half h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = __float2half(v[k]);
        l[k] = __float2half(v[k] - __half2float(h[k]));
    }
    *reinterpret_cast<u32x2*>(hi) = *reinterpret_cast<u32x2*>(h);
    if (lo) *reinterpret_cast<u32x2*>(lo) = *reinterpret_cast<u32x2*>(l);
}

// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, (DH == 32 ? 3 : 1)) void attn_bwd_dq_kernel(const __half* __restrict__ qkv, const __half* __restrict__ kt,
                                                           const __half* __restrict__ dO, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, __half* __restrict__ dhi,
                                                           __half* __restrict__ dlo, int L, int Lp, int H, int E) {
    constexpr int KS = DH / 16, DT = DH / 32;
    constexpr int KROW = DH * 2 + 16, TROW = 136;
    constexpr int KBUF = 64 * KROW, TBUF = DH * TROW, STAGE = 2 * KBUF + TBUF;
    constexpr int KCH = DH / 8, NKC = 64 * KCH / 256, NTC = DH * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K | V | K^T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const int qr = qrow < L ? qrow : L - 1;
    const long ldq = 3L * E;
    const __half* base = qkv + (long)b * L * ldq + (long)h * DH;
    const __half* ktb = kt + ((long)b * H + h) * DH * Lp;
    f16x8 qf[KS], dof[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = *reinterpret_cast<const f16x8*>(base + (long)qr * ldq + 16 * s + 8 * hh);
        dof[s] = *reinterpret_cast<const f16x8*>(dO + ((long)b * L + qr) * E + h * DH + 16 * s + 8 * hh);
    }
    const float nlse = -lse[((long)b * H + h) * L + qr], ndel = -delta[((long)b * H + h) * L + qr];

    u32x4 rk[NKC], rv[NKC], rt[NTC];
#undef GLOAD
#define GLOAD(t_)                                                                                         \
    {                                                                                                     \
        const int t__ = (t_);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                                 \
            const int c = tid + 256 * i;                                                                  \
            int key = t__ * 64 + c / KCH;                                                                 \
            if (key > L - 1) key = L - 1;                                                                 \
            rk[i] = *reinterpret_cast<const u32x4*>(base + E + (long)key * ldq + (c % KCH) * 8);          \
            rv[i] = *reinterpret_cast<const u32x4*>(base + 2 * E + (long)key * ldq + (c % KCH) * 8);      \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NTC; ++i) {                                                 \
            const int c = tid + 256 * i;                                                                  \
            rt[i] = *reinterpret_cast<const u32x4*>(ktb + (long)(c >> 3) * Lp + t__ * 64 + (c & 7) * 8);  \
        }                                                                                                 \
    }
#undef LSTORE
#define LSTORE(buf_)                                                                                      \
    {                                                                                                     \
        char* sb = smem + (buf_) * STAGE;                                                                 \
        _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                                 \
            const int c = tid + 256 * i;                                                                  \
            *reinterpret_cast<u32x4*>(sb + (c / KCH) * KROW + (c % KCH) * 16) = rk[i];                    \
            *reinterpret_cast<u32x4*>(sb + KBUF + (c / KCH) * KROW + (c % KCH) * 16) = rv[i];             \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NTC; ++i) {                                                 \
            const int c = tid + 256 * i;                                                                  \
            u32x2* p = reinterpret_cast<u32x2*>(sb + 2 * KBUF + (c >> 3) * TROW + (c & 7) * 16);          \
            p[0] = rt[i].xy;                                                                              \
            p[1] = rt[i].zw;                                                                              \
        }                                                                                                 \
    }
    f32x16 acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    const int nt = (L + 63) / 64;
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) GLOAD(t + 1);
        const char* kb = smem + buf * STAGE;
        const char* vb = kb + KBUF;
        const char* tb = kb + 2 * KBUF;
        f32x16 s[2], dp[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = s[1][r] = nlse; dp[0][r] = dp[1][r] = ndel; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 k0 = *reinterpret_cast<const f16x8*>(kb + l31 * KROW + ks * 32 + hh * 16);
            const f16x8 k1 = *reinterpret_cast<const f16x8*>(kb + (32 + l31) * KROW + ks * 32 + hh * 16);
            const f16x8 v0 = *reinterpret_cast<const f16x8*>(vb + l31 * KROW + ks * 32 + hh * 16);
            const f16x8 v1 = *reinterpret_cast<const f16x8*>(vb + (32 + l31) * KROW + ks * 32 + hh * 16);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[ks], s[1], 0, 0, 0);
            dp[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, dof[ks], dp[0], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, dof[ks], dp[1], 0, 0, 0);
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float p = (key < L) ? __builtin_amdgcn_exp2f(s[ti][r]) : 0.f;
                s[ti][r] = p * dp[ti][r];   // dS^T
            }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            f16x8 db;
#pragma unroll
            for (int j = 0; j < 8; ++j) db[j] = (_Float16)s[s2 >> 1][(s2 & 1) * 8 + j];
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const char* tr = tb + (d * 32 + l31) * TROW + (16 * s2 + 4 * hh) * 2;
                const f16x4 a0 = *reinterpret_cast<const f16x4*>(tr);
                const f16x4 a1 = *reinterpret_cast<const f16x4*>(tr + 16);
                f16x8 ka;
                ka[0] = a0[0]; ka[1] = a0[1]; ka[2] = a0[2]; ka[3] = a0[3];
                ka[4] = a1[0]; ka[5] = a1[1]; ka[6] = a1[2]; ka[7] = a1[3];
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, db, acc[d], 0, 0, 0);
            }
        }
        if (t + 1 < nt) LSTORE(buf ^ 1);
        __syncthreads();
    }
    if (qrow < L) {
        const float sc = rsqrtf((float)DH);
        const long o = ((long)b * L + qrow) * ldq + (long)h * DH;     // dq columns [0, E) of the (M, 3E) gradient
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {acc[d][g * 4] * sc, acc[d][g * 4 + 1] * sc, acc[d][g * 4 + 2] * sc, acc[d][g * 4 + 3] * sc};
                const long oo = o + d * 32 + 8 * g + 4 * hh;
                store_split4(dhi + oo, dlo ? dlo + oo : nullptr, v);
            }
    }
}

// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, (DH == 32 ? 3 : 1)) void attn_bwd_dkdv_kernel(const __half* __restrict__ qkv, const __half* __restrict__ qt,
                                                             const __half* __restrict__ dot, const __half* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             __half* __restrict__ dhi, __half* __restrict__ dlo, int L, int Lp,
                                                             int H, int E) {
    constexpr int KS = DH / 16, DT = DH / 32;
    constexpr int QROW = DH * 2 + 16, TROW = 136;
    constexpr int QBUF = 64 * QROW, TBUF = DH * TROW, STAGE = 2 * QBUF + 2 * TBUF + 512;
    constexpr int QCH = DH / 8, NQC = 64 * QCH / 256, NTC = DH * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Q | dO | Q^T | dO^T | lse[64] | delta[64]]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int key = blockIdx.x * 128 + wave * 32 + l31;
    const int kr = key < L ? key : L - 1;
    const long ldq = 3L * E;
    const __half* base = qkv + (long)b * L * ldq + (long)h * DH;
    const __half* dob = dO + (long)b * L * E + (long)h * DH;
    const __half* qtb = qt + ((long)b * H + h) * DH * Lp;
    const __half* dtb = dot + ((long)b * H + h) * DH * Lp;
    const float* lseb = lse + ((long)b * H + h) * L;
    const float* delb = delta + ((long)b * H + h) * L;
    f16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = *reinterpret_cast<const f16x8*>(base + E + (long)kr * ldq + 16 * s + 8 * hh);
        vf[s] = *reinterpret_cast<const f16x8*>(base + 2 * E + (long)kr * ldq + 16 * s + 8 * hh);
    }
    u32x4 rq[NQC], rd[NQC], rqt[NTC], rdt[NTC];
    float rl = 0.f;
#undef GLOAD
#define GLOAD(t_)                                                                                          \
    {                                                                                                      \
        const int t__ = (t_);                                                                              \
        _Pragma("unroll") for (int i = 0; i < NQC; ++i) {                                                  \
            const int c = tid + 256 * i;                                                                   \
            int q = t__ * 64 + c / QCH;                                                                    \
            if (q > L - 1) q = L - 1;                                                                      \
            rq[i] = *reinterpret_cast<const u32x4*>(base + (long)q * ldq + (c % QCH) * 8);                 \
            rd[i] = *reinterpret_cast<const u32x4*>(dob + (long)q * E + (c % QCH) * 8);                    \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NTC; ++i) {                                                  \
            const int c = tid + 256 * i;                                                                   \
            rqt[i] = *reinterpret_cast<const u32x4*>(qtb + (long)(c >> 3) * Lp + t__ * 64 + (c & 7) * 8);  \
            rdt[i] = *reinterpret_cast<const u32x4*>(dtb + (long)(c >> 3) * Lp + t__ * 64 + (c & 7) * 8);  \
        }                                                                                                  \
        if (tid < 128) {                                                                                   \
            const int q = t__ * 64 + (tid & 63);                                                           \
            rl = (tid < 64) ? (q < L ? lseb[q] : 1.0e30f) : (q < L ? delb[q] : 0.f);                       \
        }                                                                                                  \
    }
#undef LSTORE
#define LSTORE(buf_)                                                                                       \
    {                                                                                                      \
        char* sb = smem + (buf_) * STAGE;                                                                  \
        _Pragma("unroll") for (int i = 0; i < NQC; ++i) {                                                  \
            const int c = tid + 256 * i;                                                                   \
            *reinterpret_cast<u32x4*>(sb + (c / QCH) * QROW + (c % QCH) * 16) = rq[i];                     \
            *reinterpret_cast<u32x4*>(sb + QBUF + (c / QCH) * QROW + (c % QCH) * 16) = rd[i];              \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NTC; ++i) {                                                  \
            const int c = tid + 256 * i;                                                                   \
            u32x2* p = reinterpret_cast<u32x2*>(sb + 2 * QBUF + (c >> 3) * TROW + (c & 7) * 16);           \
            p[0] = rqt[i].xy;                                                                              \
            p[1] = rqt[i].zw;                                                                              \
            u32x2* p2 = reinterpret_cast<u32x2*>(sb + 2 * QBUF + TBUF + (c >> 3) * TROW + (c & 7) * 16);   \
            p2[0] = rdt[i].xy;                                                                             \
            p2[1] = rdt[i].zw;                                                                             \
        }                                                                                                  \
        if (tid < 128) reinterpret_cast<float*>(sb + 2 * QBUF + 2 * TBUF)[tid] = rl;                       \
    }
    f32x16 dv[DT], dk[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[d][r] = dk[d][r] = 0.f;
    const int nt = (L + 63) / 64;
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) GLOAD(t + 1);
        const char* qb = smem + buf * STAGE;
        const char* db_ = qb + QBUF;
        const char* qtb_ = qb + 2 * QBUF;
        const char* dtb_ = qtb_ + TBUF;
        const float* ls = reinterpret_cast<const float*>(qb + 2 * QBUF + 2 * TBUF);
        f32x16 s[2], dp[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                s[mi][r] = -ls[rr];
                dp[mi][r] = -ls[64 + rr];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 q0 = *reinterpret_cast<const f16x8*>(qb + l31 * QROW + ks * 32 + hh * 16);
            const f16x8 q1 = *reinterpret_cast<const f16x8*>(qb + (32 + l31) * QROW + ks * 32 + hh * 16);
            const f16x8 d0 = *reinterpret_cast<const f16x8*>(db_ + l31 * QROW + ks * 32 + hh * 16);
            const f16x8 d1 = *reinterpret_cast<const f16x8*>(db_ + (32 + l31) * QROW + ks * 32 + hh * 16);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, kf[ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, kf[ks], s[1], 0, 0, 0);
            dp[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, vf[ks], dp[0], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, vf[ks], dp[1], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[mi][r]);
                s[mi][r] = p;               // P
                dp[mi][r] = p * dp[mi][r];  // dS
            }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {    // 4 k-steps of 16 queries
            f16x8 pb, sb2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[j] = (_Float16)s[s2 >> 1][(s2 & 1) * 8 + j];
                sb2[j] = (_Float16)dp[s2 >> 1][(s2 & 1) * 8 + j];
            }
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int off = (d * 32 + l31) * TROW + (16 * s2 + 4 * hh) * 2;
                const f16x4 a0 = *reinterpret_cast<const f16x4*>(dtb_ + off);
                const f16x4 a1 = *reinterpret_cast<const f16x4*>(dtb_ + off + 16);
                const f16x4 b0 = *reinterpret_cast<const f16x4*>(qtb_ + off);
                const f16x4 b1 = *reinterpret_cast<const f16x4*>(qtb_ + off + 16);
                f16x8 da, qa;
                da[0] = a0[0]; da[1] = a0[1]; da[2] = a0[2]; da[3] = a0[3];
                da[4] = a1[0]; da[5] = a1[1]; da[6] = a1[2]; da[7] = a1[3];
                qa[0] = b0[0]; qa[1] = b0[1]; qa[2] = b0[2]; qa[3] = b0[3];
                qa[4] = b1[0]; qa[5] = b1[1]; qa[6] = b1[2]; qa[7] = b1[3];
                dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(da, pb, dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, sb2, dk[d], 0, 0, 0);
            }
        }
        if (t + 1 < nt) LSTORE(buf ^ 1);
        __syncthreads();
    }
    if (key < L) {
        const float sck = 1.0f / 1.4426950408889634f;
        const long o = ((long)b * L + key) * ldq + (long)h * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float vk[4] = {dk[d][g * 4] * sck, dk[d][g * 4 + 1] * sck, dk[d][g * 4 + 2] * sck, dk[d][g * 4 + 3] * sck};
                const float vv[4] = {dv[d][g * 4], dv[d][g * 4 + 1], dv[d][g * 4 + 2], dv[d][g * 4 + 3]};
                const long ok = o + E + d * 32 + 8 * g + 4 * hh, ov = o + 2 * E + d * 32 + 8 * g + 4 * hh;
                store_split4(dhi + ok, dlo ? dlo + ok : nullptr, vk);
                store_split4(dhi + ov, dlo ? dlo + ov : nullptr, vv);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole attention backward.  qkv (B*L,3E) fp16 (q pre-scaled), dO (B*L,E) fp16, o32 (B*L,E) f32, lse (B,H,L).
// Output dqkv (B*L,3E) as fp16 hi (+lo).  Workspaces: qt, kt, dot each B*H*DH*Lp halves; delta B*H*L floats.
extern "C" int wc_attn_bwd(const void* qkv, const void* dO, const float* o32, const float* lse, void* qt, void* kt,
                           void* dot, float* delta, void* dqkv_hi, void* dqkv_lo, int B, int L, int Lp, int H, int DH,
                           void* stream) {
    const int E = H * DH;
    WC_CHECK_ARG(qkv && dO && o32 && lse && qt && kt && dot && delta && dqkv_hi && B > 0 && L > 0 && Lp >= L &&
                     Lp % 64 == 0 && E % 64 == 0,
                 "wc_attn_bwd: bad argument (Lp %% 64 == 0, (H*DH) %% 64 == 0)");
    WC_CHECK_ARG(DH == 64 || DH == 32, "wc_attn_bwd: head dim must be 32 or 64");
    hipStream_t st = (hipStream_t)stream;
    WC_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)dO | (uintptr_t)qt | (uintptr_t)kt | (uintptr_t)dot) % 16 == 0,
                 "wc_attn_bwd: operands must be 16-byte aligned");
    dim3 tg(Lp / 64, 3 * (E / 64), B);
    hipLaunchKernelGGL(head_transpose_kernel, tg, dim3(256), 0, st, (const __half*)qkv, (const __half*)dO, (__half*)qt, (__half*)kt,
                       (__half*)dot, L, Lp, H, DH);
    WC_LAUNCH_CHECK("head_transpose_kernel");
    const long total = (long)B * L;
    hipLaunchKernelGGL(attn_delta_rows_kernel, dim3(wc_cdiv(total, 4)), dim3(256), 0, st, (const __half*)dO, o32, delta,
                       L, H, DH, total);
    WC_LAUNCH_CHECK("attn_delta_rows_kernel");
    dim3 grid(wc_cdiv(L, 128), H, B);
    if (DH == 64) {
        const size_t l1 = 2 * (2 * 64 * (64 * 2 + 16) + 64 * 136);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, dim3(256), l1, st, (const __half*)qkv, (const __half*)kt,
                           (const __half*)dO, lse, delta, (__half*)dqkv_hi, (__half*)dqkv_lo, L, Lp, H, E);
        const size_t l2 = 2 * (2 * 64 * (64 * 2 + 16) + 2 * 64 * 136 + 512);
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<64>, grid, dim3(256), l2, st, (const __half*)qkv, (const __half*)qt,
                           (const __half*)dot, (const __half*)dO, lse, delta, (__half*)dqkv_hi, (__half*)dqkv_lo, L, Lp, H, E);
    } else {
        const size_t l1 = 2 * (2 * 64 * (32 * 2 + 16) + 32 * 136);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, grid, dim3(256), l1, st, (const __half*)qkv, (const __half*)kt,
                           (const __half*)dO, lse, delta, (__half*)dqkv_hi, (__half*)dqkv_lo, L, Lp, H, E);
        const size_t l2 = 2 * (2 * 64 * (32 * 2 + 16) + 2 * 32 * 136 + 512);
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<32>, grid, dim3(256), l2, st, (const __half*)qkv, (const __half*)qt,
                           (const __half*)dot, (const __half*)dO, lse, delta, (__half*)dqkv_hi, (__half*)dqkv_lo, L, Lp, H, E);
    }
    WC_LAUNCH_CHECK("attn_bwd kernels");
    return WC_OK;
}
