// MFMA GEMM for gfx950:  C[M,N] = epilogue( sum_seg A_seg[M,K] * W_seg[N,K]^T )
//
// Replaces the F.linear / nn.Linear / 1x1 nn.Conv2d / torch.bmm call sites of the hot path
// (reference clip/myAtt.py:201,321; clip/model.py:198-202,264-268,420;
//  WeCLIP_model/segformer_head.py:22-28,76; WeCLIP_model/Decoder/TransDecoder.py:122;
//  WeCLIP_model/model_attn_aff_voc.py:136) with one tiled kernel on v_mfma_f32_32x32x16_f16.
//
// Both operands are fp16, K-contiguous ("TN": nn.Linear weight layout), fp32 accumulate.
// Up to 3 K-segments are accumulated into the same tile: this is how split precision is
// expressed -- x = hi + lo in fp16 gives x*w ~= hi*w_hi + lo*w_hi + hi*w_lo, i.e. a GEMM over
// the concatenated K of 3 (activation, weight) pointer pairs; 1 segment = plain fp16 GEMM.
//
// Tile 128x128x64, 256 threads = 2x2 waves, each wave 64x64 = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs).  Global -> registers -> LDS (rows padded to 144 B: conflict-free
// ds_read_b128 of 16-B k-slices across 16 consecutive rows), double-buffered LDS, the global
// loads of tile t+1 are issued before the MFMAs of tile t, one barrier per K-tile.
// blockIdx.x walks N tiles (they share the A tile through L2), blockIdx.y M tiles.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 64
#define LDS_ROW 144   // bytes per padded tile row (64 halfs = 128 B + 16 B pad)

struct GemmArgs {
    const __half* A[3];
    const __half* W[3];
    int nseg;
    int M, N, K;
    long lda, ldw;
    long sA, sW, sC, sR;  // batch strides in elements (blockIdx.z); sR = residual's
    const float* bias;    // [N] fp32 or null
    const float* resid;   // [M,ldr] fp32 or null
    long ldr;
    float* C32;
    __half* C16;
    __half* C16lo;
    long ldc;
    int act;              // 0 none, 1 QuickGELU x*sigmoid(1.702x), 2 ReLU, 3 sigmoid
    int round16;          // round (acc+bias) through fp16 first (forced-fp16 out-proj, myAtt.py:321)
    float scale;          // multiply columns n < scale_cols by scale (q / sqrt(dh), myAtt.py:54)
    int scale_cols;
    float* P32;           // optional fp32 copy of the pre-activation value (acc + bias)
    const float* aux;     // act 4: v *= QuickGELU'(aux[arow*ldaux + n]), arow = rowmap[m / rpg]*rpg + m % rpg
    const int* rowmap;
    int rpg;
    long ldaux;
    const __half* auxh;   // act 5: v *= (auxh[m*ldaux + n] > 0)  (ReLU backward from the saved fp16 output)
    const float* cscale;  // optional per-batch column scale after bias: v *= cscale[z*sCS + n] (Dropout2d)
    long sCS;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v * (1.0f / (1.0f + __expf(-1.702f * v)));
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return 1.0f / (1.0f + __expf(-v));
    return v;
}

template <bool AUX>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: buf b (0/1): A rows [0,128) then W rows [0,128), each LDS_ROW bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long zb = blockIdx.z;

    // per-thread staging coordinates: 4 chunks of 16 B for A and 4 for W per K-tile
    int srow[4], scol[4], soff[4];
    long aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        srow[i] = c >> 3;
        scol[i] = c & 7;
        int ar = m0 + srow[i];
        if (ar > g.M - 1) ar = g.M - 1;
        int wrow = n0 + srow[i];
        if (wrow > g.N - 1) wrow = g.N - 1;
        soff[i] = srow[i] * LDS_ROW + scol[i] * 16;
        aoff[i] = zb * g.sA + (long)ar * g.lda + scol[i] * 8;
        woff[i] = zb * g.sW + (long)wrow * g.ldw + scol[i] * 8;
    }
    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;

    // staging registers as scalars + macros (arrays captured by lambdas ended up in scratch)
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define GLOAD(t)                                                                         \
    {                                                                                    \
        const int seg_ = (t) / ktiles;                                                   \
        const long k0_ = (long)((t) - seg_ * ktiles) * BK;                               \
        const __half* Ap_ = seg_ == 0 ? g.A[0] : (seg_ == 1 ? g.A[1] : g.A[2]);          \
        const __half* Wp_ = seg_ == 0 ? g.W[0] : (seg_ == 1 ? g.W[1] : g.W[2]);          \
        ra0 = *reinterpret_cast<const uint4*>(Ap_ + aoff[0] + k0_);                      \
        ra1 = *reinterpret_cast<const uint4*>(Ap_ + aoff[1] + k0_);                      \
        ra2 = *reinterpret_cast<const uint4*>(Ap_ + aoff[2] + k0_);                      \
        ra3 = *reinterpret_cast<const uint4*>(Ap_ + aoff[3] + k0_);                      \
        rw0 = *reinterpret_cast<const uint4*>(Wp_ + woff[0] + k0_);                      \
        rw1 = *reinterpret_cast<const uint4*>(Wp_ + woff[1] + k0_);                      \
        rw2 = *reinterpret_cast<const uint4*>(Wp_ + woff[2] + k0_);                      \
        rw3 = *reinterpret_cast<const uint4*>(Wp_ + woff[3] + k0_);                      \
    }
#define LSTORE(buf)                                                                      \
    {                                                                                    \
        char* base_ = smem + (buf) * (2 * BM * LDS_ROW);                                 \
        *reinterpret_cast<uint4*>(base_ + soff[0]) = ra0;                                \
        *reinterpret_cast<uint4*>(base_ + soff[1]) = ra1;                                \
        *reinterpret_cast<uint4*>(base_ + soff[2]) = ra2;                                \
        *reinterpret_cast<uint4*>(base_ + soff[3]) = ra3;                                \
        *reinterpret_cast<uint4*>(base_ + BM * LDS_ROW + soff[0]) = rw0;                 \
        *reinterpret_cast<uint4*>(base_ + BM * LDS_ROW + soff[1]) = rw1;                 \
        *reinterpret_cast<uint4*>(base_ + BM * LDS_ROW + soff[2]) = rw2;                 \
        *reinterpret_cast<uint4*>(base_ + BM * LDS_ROW + soff[3]) = rw3;                 \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    GLOAD(0);
    LSTORE(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 16;   // fragment row / byte offset of k-half
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) GLOAD(t + 1);
        const char* As = smem + buf * (2 * BM * LDS_ROW) + (wr * 64 + frow) * LDS_ROW + fk;
        const char* Ws = smem + buf * (2 * BM * LDS_ROW) + (BM + wc * 64 + frow) * LDS_ROW + fk;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 a0 = *reinterpret_cast<const f16x8*>(As + ks * 32);
            f16x8 a1 = *reinterpret_cast<const f16x8*>(As + 32 * LDS_ROW + ks * 32);
            f16x8 b0 = *reinterpret_cast<const f16x8*>(Ws + ks * 32);
            f16x8 b1 = *reinterpret_cast<const f16x8*>(Ws + 32 * LDS_ROW + ks * 32);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (t + 1 < nt) LSTORE(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Side inputs (residual / aux) of a 32x32 tile are fetched before use so the loads overlap;
    // out-of-range rows/cols read a clamped address and are not stored.  AUX (act 4/5) is a
    // separate instantiation so the common epilogue carries no aux registers.
    const long cb = zb * g.sC;
    const int act = g.act;
    const bool has_res = g.resid != nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wc * 64 + ni * 32 + (lane & 31);
            const bool colok = col < g.N;
            const int colc = colok ? col : g.N - 1;
            const float bv = g.bias ? g.bias[colc] : 0.f;
            float sc = (col < g.scale_cols) ? g.scale : 1.0f;
            if (g.cscale) sc *= g.cscale[zb * g.sCS + colc];
            const int rbase = m0 + wr * 64 + mi * 32 + 4 * (lane >> 5);
            f32x16 rv, uv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row > g.M - 1) row = g.M - 1;
                rv[r] = has_res ? g.resid[zb * g.sR + (long)row * g.ldr + colc] : 0.f;
                if constexpr (AUX) {
                    if (act == 4) {
                        const long arow = g.rowmap ? (long)g.rowmap[row / g.rpg] * g.rpg + row % g.rpg : row;
                        const float u = g.aux[arow * g.ldaux + colc];
                        const float sg = 1.0f / (1.0f + __expf(-1.702f * u));
                        uv[r] = sg * (1.0f + 1.702f * u * (1.0f - sg));   // d/du [u*sigmoid(1.702u)]
                    } else {
                        uv[r] = __half2float(g.auxh[(long)row * g.ldaux + colc]) > 0.f ? 1.f : 0.f;   // ReLU'
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[mi][ni][r] + bv;
                if (g.round16) v = __half2float(__float2half(v));
                v *= sc;
                const float pre = v;
                if constexpr (AUX)
                    v *= uv[r];
                else
                    v = apply_act(v, act);
                v += rv[r];
                if (colok && row < g.M) {
                    const long o = cb + (long)row * g.ldc + col;
                    if (g.P32) g.P32[o] = pre;
                    if (g.C32) g.C32[o] = v;
                    if (g.C16) {
                        const __half h = __float2half(v);
                        g.C16[o] = h;
                        if (g.C16lo) g.C16lo[o] = __float2half(v - __half2float(h));
                    }
                }
            }
        }
}

// out[i] = alpha * sum_s part[s*n + i]   (split-K reduction: slices are a batched GEMM over K ranges)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int nslices, long n, float alpha) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += part[(long)k * n + i];
    out[i] = s * alpha;
}

extern "C" int wc_sum_slices(const float* part, float* out, int nslices, long n, float alpha, void* stream) {
    WC_CHECK_ARG(part && out && nslices > 0 && n > 0, "wc_sum_slices: bad argument");
    hipLaunchKernelGGL(sum_slices_kernel, dim3(wc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, part, out, nslices,
                       n, alpha);
    WC_LAUNCH_CHECK("sum_slices_kernel");
    return WC_OK;
}

extern "C" int wc_gemm_f16(const void* A0, const void* A1, const void* A2, const void* W0,
                           const void* W1, const void* W2, int nseg, int M, int N, int K, long lda,
                           long ldw, int batch, long sA, long sW, long sC, const float* bias,
                           const float* resid, long ldr, long sR, float* C32, void* C16, void* C16lo, long ldc, int act,
                           int round16, float scale, int scale_cols, float* P32, const float* aux,
                           const int* rowmap, int rpg, long ldaux, const void* auxh, const float* cscale,
                           long sCS, void* stream) {
    WC_CHECK_ARG(nseg >= 1 && nseg <= 3, "wc_gemm_f16: nseg must be 1..3");
    WC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0, "wc_gemm_f16: need M,N>0 and K %% 64 == 0 (got M=%d N=%d K=%d)", M, N, K);
    WC_CHECK_ARG(A0 && W0 && (nseg < 2 || (A1 && W1)) && (nseg < 3 || (A2 && W2)),
                 "wc_gemm_f16: null operand");
    WC_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && sA % 8 == 0 && sW % 8 == 0,
                 "wc_gemm_f16: operand rows must be 16-byte aligned (lda, ldw, strides %% 8 == 0)");
    WC_CHECK_ARG(((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)A2 |
                  (uintptr_t)W2) % 16 == 0, "wc_gemm_f16: operands must be 16-byte aligned");
    WC_CHECK_ARG((C32 || C16) && ldc >= N && batch >= 1 && batch <= 65535, "wc_gemm_f16: bad output");
    WC_CHECK_ARG(act >= 0 && act <= 5, "wc_gemm_f16: act must be 0..5");
    WC_CHECK_ARG(act != 5 || (auxh && ldaux >= N), "wc_gemm_f16: act 5 needs auxh, ldaux");
    WC_CHECK_ARG(act != 4 || (aux && rpg > 0 && ldaux >= N), "wc_gemm_f16: act 4 needs aux, rpg, ldaux");
    GemmArgs g;
    g.A[0] = (const __half*)A0; g.A[1] = (const __half*)A1; g.A[2] = (const __half*)A2;
    g.W[0] = (const __half*)W0; g.W[1] = (const __half*)W1; g.W[2] = (const __half*)W2;
    g.nseg = nseg; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
    g.sA = sA; g.sW = sW; g.sC = sC; g.sR = sR; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.C32 = C32; g.C16 = (__half*)C16; g.C16lo = (__half*)C16lo; g.ldc = ldc;
    g.act = act; g.round16 = round16; g.scale = scale; g.scale_cols = scale_cols;
    g.P32 = P32; g.aux = aux; g.rowmap = rowmap; g.rpg = rpg > 0 ? rpg : 1; g.ldaux = ldaux;
    g.auxh = (const __half*)auxh; g.cscale = cscale; g.sCS = sCS;
    dim3 grid(wc_cdiv(N, BN), wc_cdiv(M, BM), batch);
    WC_CHECK_ARG(grid.y <= 65535, "wc_gemm_f16: M too large for one launch");
    const size_t lds = 2 * 2 * BM * LDS_ROW;
    if (act >= 4)
        hipLaunchKernelGGL(gemm_f16_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(gemm_f16_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, g);
    WC_LAUNCH_CHECK("gemm_f16_kernel");
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 -> fp16 (hi) [+ fp16 residual (lo)] conversion: weights at load time, activations that
// do not come out of a fused epilogue.
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x,
                                                         __half* __restrict__ hi,
                                                         __half* __restrict__ lo, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        const float f[4] = {v.x, v.y, v.z, v.w};
        __half h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = __float2half(f[k]);
            l[k] = __float2half(f[k] - __half2float(h[k]));
        }
        *reinterpret_cast<uint2*>(hi + i) = *reinterpret_cast<uint2*>(h);
        if (lo) *reinterpret_cast<uint2*>(lo + i) = *reinterpret_cast<uint2*>(l);
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long j = (n & ~3L) + threadIdx.x;
        const __half h = __float2half(x[j]);
        hi[j] = h;
        if (lo) lo[j] = __float2half(x[j] - __half2float(h));
    }
}

extern "C" int wc_split_f16(const float* x, void* hi, void* lo, long n, void* stream) {
    WC_CHECK_ARG(x && hi && n > 0, "wc_split_f16: bad argument");
    WC_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)hi % 8 == 0) && ((uintptr_t)lo % 8 == 0),
                 "wc_split_f16: misaligned buffer");
    int blocks = wc_cdiv(n, 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x,
                       (__half*)hi, (__half*)lo, n);
    WC_LAUNCH_CHECK("split_f16_kernel");
    return WC_OK;
}
