// Weight-gradient GEMM on row-major operands + the split-K slice reductions (split out of gemm.hip in round 4).
#include "gemm_common.h"
#include <cstring>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM on row-major operands ("KM" layout):  C[n, k] = sum_m dY[m, n] * X[m, k]
// (reference: what autograd computes for nn.Linear / 1x1 nn.Conv2d weights, WeCLIP_model/segformer_head.py:22-28,
//  Decoder/TransDecoder.py:98-125).  The contraction index m (tokens) is the ROW index of both operands, so the
// K-contiguous kernels above would need dY^T and X^T materialised (one transpose kernel per operand per
// layer).  Here both [64 tokens][128 columns] tiles are DMA'd as they lie in memory and the MFMA fragments
// (8 consecutive tokens of one column) are fetched with gfx950's transposing LDS read ds_read_b64_tr_b16:
// a 16-lane group reads a 4-row x 16-column block and each lane receives one column of it.
// LDS image: 256-B rows, 16-B chunk ch of row r stored at chunk ch ^ (((r&3)<<2) | ((r>>2)&3)) (swizzle on the
// DMA source address): the four rows of a block then sit in different bank quarters.
// Split-K over blockIdx.z (token slices) into fp32 partials; optional extra output column K = sum_m dY[m, n]
// (the bias gradient) from one more MFMA against a fragment of ones.
struct KmArgs {
    const __half* A;      // dY (M, lda)
    const __half* X;      // X  (M, ldx)
    const __half* zeros;  // >= 16 B of zeros: source of out-of-range rows / column chunks
    int M, N, K;          // tokens, dY columns (output rows), X columns (output columns)
    long lda, ldx;
    int x_rpg, x_gs, x_off;   // X row of token m = (m / x_rpg) * x_gs + m % x_rpg + x_off  (skips CLS rows)
    int mslice;           // tokens per slice (multiple of 64)
    int bias;
    int bias_edge;        // K % 128 == 0: the bias column K lies just past the last column tile; that tile's right-hand waves
                          // accumulate it and store it themselves (no third column tile that re-reads dY for one column)
    int gx;
    int tiles, ns, units; // output tiles per (group, slice) unit, slices per group, units = groups * ns
    int xcd;              // XCD-aware workgroup order (units % 8 == 0)
    long gA, gX, gP;      // group strides of dY, X and the partials, in elements
    GemmArgs e;           // epilogue: M = N, N = K + bias, C32 = partials, ldc, sC
};

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));

// NST = stages of the ring: 5 (80 KiB: two workgroups fill the CU's 160 KiB, 2 x 64 KiB in flight) or 8 (128 KiB: for grids of at
// most one workgroup per CU).  What a stage costs (round 4, 86 016 x 256 x 256, debug switches that drop the DMA or the
// arithmetic): an EMPTY iteration 0.27 us (counted wait + barrier), fragment reads + MFMA issue 0.52 us, DMA issue 0.12 us -- and
// they ADD UP inside a wave (one wave per SIMD: nothing to overlap with), 1.1 us per 32 tokens although the matrix pipe needs
// 0.2 us.  Two workgroups per CU interleave their chains (0.85 us per stage and CU) but are then bound by the bytes in flight:
// the DMA alone needed 33.6 us with 2 x 3 stages in flight against 25.5 us with 7 -- hence five stages per workgroup, not four.
//
// HV = 2 (grids of at most one workgroup per CU): the workgroup has EIGHT waves = two halves of four; a stage is 64 tokens, the
// halves take its first / second 32 tokens into accumulators of their own (a split over tokens inside the workgroup), all eight
// waves stage it, and the second half's accumulators are added to the first's through LDS at the end (fixed order).  Two waves
// per SIMD interleave their chains like two workgroups would, the ring is 4 x 32 KiB with 96 KiB in flight, and the launch
// writes HALF the partial tiles of the two-workgroups-per-CU form (17 MB instead of 34 MB per 256 x 257 gradient, read once
// more by the slice reduction).
template <int NST, int HV>
__global__ __launch_bounds__(256 * HV, (HV == 1 && NST <= 5) ? 2 : 1) void gemm_km_kernel(KmArgs g) {
    // ring of NST stages x [ST tokens]: [A tile | X tile] of ST x 256 B each; NST - 1 stages in flight while one is consumed,
    // counted vmcnt waits + raw s_barrier (a __syncthreads() drains every outstanding LDS-DMA: with 2 stages of 64
    // tokens the loop ran at one global-memory latency per stage)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ST = 32 * HV;           // tokens per stage
    constexpr int TILE = ST * 256;
    constexpr int NT_ = 256 * HV;         // threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = HV == 2 ? wave >> 2 : 0, w4 = wave & 3;
    const int wr = w4 >> 1, wc = w4 & 1;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, so unit u (= one token slice of one group, whose
    // `tiles` workgroups all read the same dY / X rows) takes the ids congruent to u mod 8 of its block of 8 units:
    // the slice's operands (~2-3 MB) are fetched into ONE 4-MiB L2 instead of all eight (33 -> 36 us at N 256, K 1024,
    // 16 slices).  Only when the units divide evenly over the XCDs: 33 units of 14 long tiles (the grouped adapters)
    // would put 70 workgroups on the first XCD's 64 slots and run two rounds there (173 -> 265 us).
    int unit, tile;
    if (g.xcd) {
        const int chunk = blockIdx.x / (8 * g.tiles), within = blockIdx.x - chunk * (8 * g.tiles);
        unit = chunk * 8 + (within & 7);
        tile = within >> 3;
    } else {
        unit = blockIdx.x / g.tiles;
        tile = blockIdx.x - unit * g.tiles;
    }
    const int grp = unit / g.ns;
    const int z = unit - grp * g.ns;
    const int ty = tile / g.gx, tx = tile - ty * g.gx;
    const int n0 = ty * 128, k0 = tx * 128;
    const __half* Ag = g.A + (long)grp * g.gA;      // group (e.g. adapter) of a grouped launch
    const __half* Xg = g.X + (long)grp * g.gX;
    const int mbeg = z * g.mslice;
    const int mend = (mbeg + g.mslice < g.M) ? mbeg + g.mslice : g.M;
    const int nt = (mend - mbeg + ST - 1) / ST;

    // DMA bookkeeping: chunk q = i*256 + tid -> tile row q>>4 (token), physical chunk q&15
    int rowi[2], acol[2], xcol[2], xg[2], xr[2];
    unsigned offA[2], offX[2];
    bool aok[2], xok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * NT_ + tid;
        const int row = q >> 4, pc = q & 15;
        const int lc = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
        rowi[i] = row;
        acol[i] = n0 + lc * 8;
        xcol[i] = k0 + lc * 8;
        aok[i] = acol[i] + 8 <= g.lda && acol[i] < g.N;
        xok[i] = xcol[i] + 8 <= g.ldx && xcol[i] < g.K;
        const int m = mbeg + row;
        xg[i] = m / g.x_rpg;
        xr[i] = m - xg[i] * g.x_rpg;
        // fast path: 32-bit BYTE offsets from the (uniform) operand bases, advanced by constants per stage -- the 64-bit
        // row * pitch products per DMA instruction were ~80 of the ~110 VALU instructions of an iteration (PMC: a quarter
        // of a wave's cycles).  A column chunk outside the operand reads column 0 instead: it only feeds outputs that are not
        // stored.  (The launcher keeps operands of 4 GiB and more off this kernel.)
        offA[i] = (unsigned)(((long)m * g.lda + (aok[i] ? acol[i] : 0)) * 2);
        offX[i] = (unsigned)((((long)xg[i] * g.x_gs + xr[i] + g.x_off) * g.ldx + (xok[i] ? xcol[i] : 0)) * 2);
    }
    const unsigned dA = (unsigned)(ST * g.lda * 2), dX = (unsigned)(ST * g.ldx * 2);
    const unsigned dXg = (unsigned)(((long)g.x_gs - g.x_rpg) * g.ldx * 2);      // extra step when a token group is crossed
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
#define KM_LOAD(t_, buf_)                                                                                       \
    {                                                                                                            \
        char* dst_ = smem + (buf_) * (2 * TILE) + wave * 1024;                                                   \
        if (mbeg + ((t_) + 1) * ST <= mend) {      /* (uniform) every token row of the stage exists */           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
                __builtin_amdgcn_global_load_lds((gbl_ptr)(reinterpret_cast<const char*>(Ag) + offA[i]), (lds_ptr)(dst_ + i * (4096 * HV)), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((gbl_ptr)(reinterpret_cast<const char*>(Xg) + offX[i]), (lds_ptr)(dst_ + TILE + i * (4096 * HV)), 16, 0, 0); \
            }                                                                                                    \
        } else {                                   /* ragged last stage: rows past the slice read zeros */       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
                const int m_ = mbeg + (t_) * ST + rowi[i];                                                       \
                const bool in_ = m_ < mend;                                                                      \
                const __half* pa_ = (in_ && aok[i]) ? Ag + (long)m_ * g.lda + acol[i] : g.zeros;                 \
                const __half* px_ = (in_ && xok[i]) ? Xg + ((long)xg[i] * g.x_gs + xr[i] + g.x_off) * g.ldx + xcol[i] : g.zeros; \
                __builtin_amdgcn_global_load_lds((gbl_ptr)pa_, (lds_ptr)(dst_ + i * (4096 * HV)), 16, 0, 0);            \
                __builtin_amdgcn_global_load_lds((gbl_ptr)px_, (lds_ptr)(dst_ + TILE + i * (4096 * HV)), 16, 0, 0);     \
            }                                                                                                    \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            offA[i] += dA;                                                                                       \
            offX[i] += dX;                                                                                       \
            xr[i] += ST;                                                                                         \
            while (xr[i] >= g.x_rpg) { xr[i] -= g.x_rpg; xg[i] += 1; offX[i] += dXg; }                           \
        }                                                                                                        \
    }
    // transposing fragment reads: lane = 16*grp + 4*q + p; grp = 2*hh + gi
    const int hh = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    int aaddr[2][2], baddr[2][2];      // [mi / ni][r]: byte offset inside an operand tile for k-step 0
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * hh + 4 * r + q + 32 * half;           // + 16 * ks
        const int f = (q << 2) | ((2 * hh + r) & 3);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cha = wr * 8 + t * 4 + 2 * gi + (p >> 1);
            const int chb = wc * 8 + t * 4 + 2 * gi + (p >> 1);
            aaddr[t][r] = 256 * row + 16 * (cha ^ f) + 8 * (p & 1);
            baddr[t][r] = 256 * row + 16 * (chb ^ f) + 8 * (p & 1);
        }
    }
    const unsigned lbase = (unsigned)(size_t)(lds_ptr)smem;        // LDS byte address of the ring
#define KM_TR(dst_, addr_) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst_) : "v"(addr_))
#define KM_JOIN(lo_, hi_) __builtin_bit_cast(f16x8, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7))

    f32x16 acc[2][2], bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // the wave whose 64 output columns contain column K (the bias column) also accumulates dY^T 1
    const int kb = g.K - (k0 + wc * 64);
    const bool edge_bias = g.bias_edge && tx == g.gx - 1 && wc == 1;
    const bool own_bias = (g.bias && kb >= 0 && kb < 64) || edge_bias;
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;

    // KM_LOAD advances the token -> X row map, so the stages must be requested in order: 0 .. NST-2, then t + NST-1 in the loop
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nt) KM_LOAD(s0, s0);
    int buf = 0, lbuf = NST - 1;           // ring slots of stage t and of stage t + NST - 1
    for (int t = 0; t < nt; ++t, buf = buf + 1 == NST ? 0 : buf + 1, lbuf = lbuf + 1 == NST ? 0 : lbuf + 1) {
        // stage t has landed once at most the requests of stages t+1 .. t+NST-2 (4 DMA instructions each) are outstanding
        const int ahead = nt - 1 - t < NST - 2 ? nt - 1 - t : NST - 2;
        switch (ahead) {
            case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();     // every wave's part of stage t is in LDS; stage t-1's buffer is free
        if (t + NST - 1 < nt) KM_LOAD(t + NST - 1, lbuf);
        // Fragment reads as inline asm: hipcc puts a full `s_waitcnt vmcnt(0)` in front of the ds_read_tr builtin
        // whenever LDS-DMA is outstanding (it cannot tell which LDS bytes the DMA writes), which would drain the ring.
        // Both k-steps' 16 transposing reads are issued, then one lgkmcnt(0) that carries the registers.
        const unsigned sa = lbase + buf * (2 * TILE), sx = sa + TILE;
        s16x4 ra[2][2][2], rb[2][2][2];           // [ks][mi / ni][half]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    KM_TR(ra[ks][t2][r], sa + ks * 4096 + aaddr[t2][r]);
                    KM_TR(rb[ks][t2][r], sx + ks * 4096 + baddr[t2][r]);
                }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[1][0][0]),
                       "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]), "+v"(rb[0][0][0]), "+v"(rb[0][0][1]),
                       "+v"(rb[0][1][0]), "+v"(rb[0][1][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]),
                       "+v"(rb[1][1][1]));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 a0 = KM_JOIN(ra[ks][0][0], ra[ks][0][1]);
            const f16x8 a1 = KM_JOIN(ra[ks][1][0], ra[ks][1][1]);
            const f16x8 b0 = KM_JOIN(rb[ks][0][0], rb[ks][0][1]);
            const f16x8 b1 = KM_JOIN(rb[ks][1][0], rb[ks][1][1]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
            if (own_bias) {
                bacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, ones, bacc[0], 0, 0, 0);
                bacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, ones, bacc[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                      // the epilogue reuses the ring as scratch
#undef KM_TR
#undef KM_JOIN
#undef KM_LOAD
    if constexpr (HV == 2) {
        // second half -> first half, register by register through 96 KiB of the (idle) ring behind the 32 KiB the storing
        // waves use as epilogue scratch; a fixed order (first + second), so the result does not depend on timing
        float* ex = reinterpret_cast<float*>(smem + 32768) + w4 * (6 * 16 * 64) + lane;
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[((i * 2 + j) * 16 + r) * 64] = acc[i][j][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) ex[((4 + i) * 16 + r) * 64] = bacc[i][r];
            }
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((i * 2 + j) * 16 + r) * 64];
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[i][r] += ex[((4 + i) * 16 + r) * 64];
        }
    }
    if (edge_bias) {     // every column of bacc holds the row sums: one lane per row group stores them into output column K
        if ((lane & 31) == 0) {
            float* pb = g.e.C32 + (long)z * g.e.sC + (long)grp * g.gP + g.K;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = n0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < g.N) pb[(long)row * g.e.ldc] = bacc[mi][r];
                }
        }
    } else if (own_bias) {      // ... or drop them into output column K inside this wave's block
        const int ni = kb >> 5, cl = kb & 31;
        if ((lane & 31) == cl) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (ni == 0) acc[mi][0][r] = bacc[mi][r];
                    else acc[mi][1][r] = bacc[mi][r];
                }
        }
    }
    const float bv[2] = {0.f, 0.f}, sc[2] = {1.f, 1.f};
    gemm_epilogue<false>(g.e, acc, n0, k0, wr, wc, lane, z, smem + w4 * 8192, bv, sc, (long)z * g.e.sC + (long)grp * g.gP);
}

// part: (nslices, N, K + bias) fp32 with nslices = ceil(M / mslice); zeros: device buffer of >= 16 zero bytes.
extern "C" int wc_gemm_km_f16_grouped(const void* dY, long lda, const void* X, long ldx, const void* zeros, int M, int N,
                                      int K, int x_rpg, int x_gs, int x_off, int mslice, int bias, float* part, int groups,
                                      long gA, long gX, void* stream);

extern "C" int wc_gemm_km_f16(const void* dY, long lda, const void* X, long ldx, const void* zeros, int M, int N, int K,
                              int x_rpg, int x_gs, int x_off, int mslice, int bias, float* part, void* stream) {
    return wc_gemm_km_f16_grouped(dY, lda, X, ldx, zeros, M, N, K, x_rpg, x_gs, x_off, mslice, bias, part, 1, 0, 0, stream);
}

// groups > 1: `groups` weight gradients of one shape in one launch (blockIdx.y): group i reads dY + i*gA and X + i*gX
// (elements) and writes part + i * nslices * N * (K + bias).
extern "C" int wc_gemm_km_f16_grouped(const void* dY, long lda, const void* X, long ldx, const void* zeros, int M, int N,
                                      int K, int x_rpg, int x_gs, int x_off, int mslice, int bias, float* part, int groups,
                                      long gA, long gX, void* stream) {
    WC_CHECK_ARG(dY && X && zeros && part && M > 0 && N > 0 && K > 0, "wc_gemm_km_f16: bad argument");
    WC_CHECK_ARG(groups >= 1 && groups <= 65535 && gA % 8 == 0 && gX % 8 == 0, "wc_gemm_km_f16_grouped: bad group strides");
    WC_CHECK_ARG(lda % 8 == 0 && ldx % 8 == 0 && lda >= N && ldx >= K && ((uintptr_t)dY | (uintptr_t)X | (uintptr_t)zeros) % 16 == 0,
                 "wc_gemm_km_f16: operand rows must be 16-byte aligned (lda, ldx %% 8 == 0)");
    WC_CHECK_ARG(mslice > 0 && mslice % 64 == 0, "wc_gemm_km_f16: mslice must be a positive multiple of 64");
    WC_CHECK_ARG(x_rpg >= 1 && x_gs >= 0 && x_off >= 0, "wc_gemm_km_f16: bad row map");
    {   // the kernel addresses both operands with 32-bit byte offsets from their (group) bases
        const long xrows = (long)((M - 1) / x_rpg) * x_gs + (x_rpg - 1 < M - 1 ? x_rpg - 1 : M - 1) + x_off + 1;
        WC_CHECK_ARG((long)M * lda * 2 < (1L << 32) && xrows * ldx * 2 < (1L << 32), "wc_gemm_km_f16: operands of 4 GiB and more are not supported");
    }
    const int ns = wc_cdiv(M, mslice);
    WC_CHECK_ARG(ns <= 65535, "wc_gemm_km_f16: too many slices");
    KmArgs g;
    g.A = (const __half*)dY; g.X = (const __half*)X; g.zeros = (const __half*)zeros;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldx = ldx;
    g.x_rpg = x_rpg; g.x_gs = x_gs; g.x_off = x_off; g.mslice = mslice; g.bias = bias ? 1 : 0;
    const int K1 = K + g.bias;
    g.bias_edge = (g.bias && K % 128 == 0) ? 1 : 0;
    g.gx = g.bias_edge ? K / 128 : wc_cdiv(K1, 128);
    GemmArgs& e = g.e;
    e.A[0] = e.A[1] = e.A[2] = nullptr; e.W[0] = e.W[1] = e.W[2] = nullptr;
    e.nseg = 1; e.M = N; e.N = K1; e.K = 0; e.lda = e.ldw = 0; e.sA = e.sW = 0;
    e.sC = (long)N * K1; e.sR = 0; e.bias = nullptr; e.resid = nullptr; e.ldr = 0;
    e.C32 = part; e.C16 = nullptr; e.C16lo = nullptr; e.ldc = K1; e.act = 0; e.round16 = 0; e.scale = 1.f; e.scale_cols = 0;
    e.P32 = nullptr; e.aux = nullptr; e.rowmap = nullptr; e.row0 = 0; e.rpg = 1; e.ldaux = 0; e.auxh = nullptr; e.cscale = nullptr;
    e.sCS = 0; e.gx = g.gx; e.gy = wc_cdiv(N, 128); e.vec = 0; e.auxvec = 0;
    e.zdiv = 1; e.sA2 = e.sW2 = e.sC2 = e.sB2 = e.sX2 = 0;
    g.gA = gA; g.gX = gX; g.gP = (long)ns * N * K1;
    g.tiles = g.gx * wc_cdiv(N, 128); g.ns = ns; g.units = groups * ns;
    g.xcd = g.units % 8 == 0 ? 1 : 0;
    WC_CHECK_ARG((long)g.tiles * g.units < (1L << 31), "wc_gemm_km_f16: grid too large");
    dim3 grid((unsigned)(g.tiles * g.units));
    const int pr = wc_prof_begin(stream);
    const int sl = shape_log_begin(stream);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n_cu <= 0)
            n_cu = 256;
        WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_km_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 256) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_km_kernel<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 64 * 256) == hipSuccess,
                     "wc_gemm_km_f16: cannot reserve 128 KiB of LDS");
    }
    if ((long)g.tiles * g.units <= n_cu && mslice >= 128)      // at most one workgroup per CU: eight waves, 64-token stages
        hipLaunchKernelGGL((gemm_km_kernel<4, 2>), grid, dim3(512), 4 * 128 * 256, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((gemm_km_kernel<5, 1>), grid, dim3(256), 5 * 64 * 256, (hipStream_t)stream, g);
    shape_log_end(sl, "km", M, N, K1, 1, groups, ns, 0, stream);
    wc_prof_end(pr, "gemm_km_kernel", 2.0 * M * N * K1 * groups, stream);
    WC_LAUNCH_CHECK("gemm_km_kernel");
    return WC_OK;
}

// out[i] = alpha * sum_s part[s*n + i]   (split-K reduction: slices are a batched GEMM over K ranges)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int nslices, long n, float alpha) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 slice loads in flight (a serial chain ran at one load latency per slice)
    int k = 0;
    for (; k + 4 <= nslices; k += 4) {
        s0 += part[(long)k * n + i];
        s1 += part[(long)(k + 1) * n + i];
        s2 += part[(long)(k + 2) * n + i];
        s3 += part[(long)(k + 3) * n + i];
    }
    for (; k < nslices; ++k) s0 += part[(long)k * n + i];
    out[i] = ((s0 + s1) + (s2 + s3)) * alpha;
}

// Same reduction for a weight-gradient GEMM whose operand carried a ones row: part is (slices, rows, cols+1),
// columns 0..cols-1 go to the dense weight gradient out_w (rows, cols), the last column to the bias gradient.
__global__ __launch_bounds__(256) void sum_slices_wb_kernel(const float* __restrict__ part, float* __restrict__ out_w,
                                                             float* __restrict__ out_b, int nslices, int rows, int cols,
                                                             float alpha, long gW, long gB) {
    const long n = (long)rows * (cols + 1);
    part += (long)blockIdx.y * nslices * n;        // group of a grouped weight-gradient launch
    out_w += (long)blockIdx.y * gW;
    out_b += (long)blockIdx.y * gB;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 slice loads in flight
    int k = 0;
    for (; k + 4 <= nslices; k += 4) {
        s0 += part[(long)k * n + i];
        s1 += part[(long)(k + 1) * n + i];
        s2 += part[(long)(k + 2) * n + i];
        s3 += part[(long)(k + 3) * n + i];
    }
    for (; k < nslices; ++k) s0 += part[(long)k * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    const int r = (int)(i / (cols + 1)), c = (int)(i - (long)r * (cols + 1));
    if (c < cols) out_w[(long)r * cols + c] = s * alpha;
    else out_b[r] = s * alpha;
}

extern "C" int wc_sum_slices_wb_grouped(const float* part, float* out_w, float* out_b, int nslices, int rows, int cols,
                                        float alpha, int groups, long gW, long gB, void* stream);

extern "C" int wc_sum_slices_wb(const float* part, float* out_w, float* out_b, int nslices, int rows, int cols,
                                float alpha, void* stream) {
    return wc_sum_slices_wb_grouped(part, out_w, out_b, nslices, rows, cols, alpha, 1, 0, 0, stream);
}

// part (groups, nslices, rows, cols + 1); group i writes out_w + i*gW and out_b + i*gB (elements)
extern "C" int wc_sum_slices_wb_grouped(const float* part, float* out_w, float* out_b, int nslices, int rows, int cols,
                                        float alpha, int groups, long gW, long gB, void* stream) {
    WC_CHECK_ARG(part && out_w && out_b && nslices > 0 && rows > 0 && cols > 0 && groups >= 1 && groups <= 65535,
                 "wc_sum_slices_wb: bad argument");
    hipLaunchKernelGGL(sum_slices_wb_kernel, dim3(wc_cdiv((long)rows * (cols + 1), 256), groups), dim3(256), 0,
                       (hipStream_t)stream, part, out_w, out_b, nslices, rows, cols, alpha, gW, gB);
    WC_LAUNCH_CHECK("sum_slices_wb_kernel");
    return WC_OK;
}

// Many split-K reductions in ONE launch (a training step has 16 of them, 5-7 us each at the launch floor): the jobs
// travel BY VALUE in the kernel arguments (no table in device memory: nothing to copy, nothing a captured graph could
// find overwritten on replay).  blockIdx.y = job, blockIdx.x strides over its elements; same summation order as
// sum_slices_wb_kernel.
#define SUMJ_MAX 64
struct SumJobs {
    const float* part[SUMJ_MAX];
    float* out_w[SUMJ_MAX];
    float* out_b[SUMJ_MAX];
    int nslices[SUMJ_MAX], rows[SUMJ_MAX], cols[SUMJ_MAX];
    float alpha[SUMJ_MAX];
    long sstride[SUMJ_MAX];      // elements between two slices (a job may reduce a ROW RANGE of a wider partial matrix)
};
__global__ __launch_bounds__(256) void sum_slices_wb_multi_kernel(SumJobs j) {
    const int q = blockIdx.y;
    const float* __restrict__ part = j.part[q];
    float* __restrict__ out_w = j.out_w[q];
    float* __restrict__ out_b = j.out_b[q];
    const int nslices = j.nslices[q], cols = j.cols[q];
    const float alpha = j.alpha[q];
    const long n = (long)j.rows[q] * (cols + 1);
    const long ss = j.sstride[q];
    if (((n | ss) & 3) == 0 && ((uintptr_t)part & 15) == 0) {
        // four consecutive elements per thread, 16-byte loads (the same four partial sums per element, the same order): the
        // one-element form moved 256 B per wave instruction and ran the 260 MB of the head's partials at 3 TB/s
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
            int k = 0;
#define SUMJ_ADD(d_, k_) { const float4 t_ = *reinterpret_cast<const float4*>(part + (long)(k_) * ss + i); d_.x += t_.x; d_.y += t_.y; d_.z += t_.z; d_.w += t_.w; }
            for (; k + 4 <= nslices; k += 4) {
                SUMJ_ADD(s0, k) SUMJ_ADD(s1, k + 1) SUMJ_ADD(s2, k + 2) SUMJ_ADD(s3, k + 3)
            }
            for (; k < nslices; ++k) SUMJ_ADD(s0, k)
#undef SUMJ_ADD
            const float sv[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                 (s0.w + s1.w) + (s2.w + s3.w)};
            int r = (int)(i / (cols + 1)), c = (int)(i - (long)r * (cols + 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c < cols) out_w[(long)r * cols + c] = sv[e] * alpha;
                else out_b[r] = sv[e] * alpha;
                if (++c > cols) { c = 0; ++r; }
            }
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 slice loads in flight
        int k = 0;
        for (; k + 4 <= nslices; k += 4) {
            s0 += part[(long)k * ss + i];
            s1 += part[(long)(k + 1) * ss + i];
            s2 += part[(long)(k + 2) * ss + i];
            s3 += part[(long)(k + 3) * ss + i];
        }
        for (; k < nslices; ++k) s0 += part[(long)k * ss + i];
        const float s = (s0 + s1) + (s2 + s3);
        const int r = (int)(i / (cols + 1)), c = (int)(i - (long)r * (cols + 1));
        if (c < cols) out_w[(long)r * cols + c] = s * alpha;
        else out_b[r] = s * alpha;
    }
}

// jobs: count x 8 host int64 {part, out_w, out_b (device pointers), nslices, rows, cols, alpha as float bits,
// slice stride in elements (0 = rows * (cols + 1): the job covers the whole partial matrix)}
extern "C" int wc_sum_slices_wb_multi(const int64_t* jobs, int count, void* stream) {
    WC_CHECK_ARG(jobs && count > 0, "wc_sum_slices_wb_multi: bad argument");
    for (int base = 0; base < count; base += SUMJ_MAX) {
        SumJobs j;
        const int m = count - base < SUMJ_MAX ? count - base : SUMJ_MAX;
        long nmax = 0;
        for (int q = 0; q < SUMJ_MAX; ++q) {
            const int64_t* e = jobs + (long)(base + (q < m ? q : 0)) * 8;
            j.part[q] = reinterpret_cast<const float*>(e[0]);
            j.out_w[q] = reinterpret_cast<float*>(e[1]);
            j.out_b[q] = reinterpret_cast<float*>(e[2]);
            j.nslices[q] = (int)e[3]; j.rows[q] = (int)e[4]; j.cols[q] = (int)e[5];
            const unsigned bits = (unsigned)e[6];
            memcpy(&j.alpha[q], &bits, 4);
            WC_CHECK_ARG(j.part[q] && j.out_w[q] && j.out_b[q] && j.nslices[q] > 0 && j.rows[q] > 0 && j.cols[q] > 0,
                         "wc_sum_slices_wb_multi: bad job");
            const long n = (long)j.rows[q] * (j.cols[q] + 1);
            j.sstride[q] = e[7] > 0 ? (long)e[7] : n;
            WC_CHECK_ARG(j.sstride[q] >= n, "wc_sum_slices_wb_multi: slice stride smaller than the job");
            if (q < m && n > nmax) nmax = n;
        }
        long bx = wc_cdiv(nmax, 256 * 4);          // <= 4 elements per thread of the largest job
        if (bx < 1) bx = 1;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(sum_slices_wb_multi_kernel, dim3((unsigned)bx, m), dim3(256), 0, (hipStream_t)stream, j);
        WC_LAUNCH_CHECK("sum_slices_wb_multi_kernel");
    }
    return WC_OK;
}

extern "C" int wc_sum_slices(const float* part, float* out, int nslices, long n, float alpha, void* stream) {
    WC_CHECK_ARG(part && out && nslices > 0 && n > 0, "wc_sum_slices: bad argument");
    hipLaunchKernelGGL(sum_slices_kernel, dim3(wc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, part, out, nslices,
                       n, alpha);
    WC_LAUNCH_CHECK("sum_slices_kernel");
    return WC_OK;
}

