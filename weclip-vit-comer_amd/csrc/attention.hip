// Multi-head self-attention forward for gfx950 (MFMA fp16, fp32 softmax), in two kernels.
//
// Replaces reference clip/myAtt.py:21-64 (`_scaled_dot_product_attention`: q/sqrt(d), bmm,
// softmax, bmm -- two materialised (B*H, L, L) fp32 tensors) and :325-326 (head-mean of the
// probabilities), fed by the packed in-projection output (myAtt.py:201,257-268).
//
//   attn_fwd_kernel  : flash-style O = softmax(Q K^T) V with online softmax; never writes the
//                      L x L scores; emits O rounded to fp16 (what myAtt.py:321 feeds the fp16
//                      out-projection) and the per-row log-sum-exp (base 2).
//   attn_mean_kernel : mean_h P_h = (1/H) sum_h exp2(Q_h K_h^T - LSE_h) for a 128x128 tile,
//                      the H-head sum held in MFMA accumulators, written once, coalesced
//                      (the returned `attn_output_weights.sum(dim=1) / num_heads`).
//   attn_row_body (extra workgroups of attn_fwd_kernel) / the edge part of attn_mean_kernel: the first L % 128 query
//                      rows (and key columns) when that remainder is tiny (the CLS token of a 1 + 32*32 sequence): a
//                      129th row must not cost a whole 128-row tile, so the tiled work starts at row L % 128; the mean
//                      kernel's first tile row / column take the edge entries from the tiles they hold in LDS.
// V stays row-major [key][dh] (as the in-projection wrote it); the O^T = V^T P^T fragments (8 consecutive keys
// of one dh column) come from gfx950's transposing LDS read ds_read_b64_tr_b16 -- no V^T copy in HBM.
//
// Input `qkv` is the in-projection GEMM output (B*L, 3E) fp16 with q pre-multiplied by
// log2(e)/sqrt(dh) (wc_gemm_f16 scale/scale_cols), so every exponential is a bare v_exp_f32.
//
// MFMA layouts (v_mfma_f32_32x32x16_f16): A[i=lane&31][k=8*(lane>>5)+j], B[k][j=lane&31],
// D col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The forward computes S^T = K Q^T so
// a lane owns one query (softmax statistics are lane-local) and then O^T = V^T P^T taking the
// S^T accumulators as B operand directly (k order inside a 16-step: 8*(j>>2) + 4*(lane>>5) + (j&3)).
#include "common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NEG_BIG (-1.0e30f)

// ------------------------------------------------------------------------------------------------
// The first `q_origin` query rows of every (b, h): one workgroup per row.  scores -> LDS, block max / sum,
// P rounded to fp16 like the MFMA operand of the tiled kernel, O = P V with lanes along dh.
// Runs as extra workgroups of attn_fwd_kernel (they fill CU slots as the tiled workgroups drain).
// lds: >= (L + 16 + (NT / (DH / 8)) * DH) floats (NT = threads per workgroup).
template <int DH, int NT>
__device__ __forceinline__ void attn_row_body(const __half* __restrict__ qkv, __half* __restrict__ out,
                                              float* __restrict__ out32, float* __restrict__ lse, int L, int H,
                                              int E, int q, int h, int b, float* lds) {
    constexpr int NCH = DH / 8;              // 16-B chunks per row
    constexpr int NPT = NT / NCH;            // key slices in the P V pass
    float* sc = lds;                         // [L] scores, then probabilities
    float* red = lds + ((L + 3) & ~3);
    float (*part)[DH] = reinterpret_cast<float (*)[DH]>(red + 16);
    const int tid = threadIdx.x;
    const long ldq = 3L * E;
    const __half* base = qkv + (long)b * L * ldq + (long)h * DH;
    f16x8 qv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) qv[c] = *reinterpret_cast<const f16x8*>(base + (long)q * ldq + c * 8);
    float mx = NEG_BIG;
    for (int key = tid; key < L; key += NT) {
        const __half* kr = base + E + (long)key * ldq;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(kr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf((float)qv[c][j], (float)kv[j], s);
        }
        sc[key] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int key = tid; key < L; key += NT) {
        const float pr = __builtin_amdgcn_exp2f(sc[key] - mx);
        sum += pr;
        sc[key] = (float)(_Float16)pr;
    }
    sum = block_sum(sum, red);               // (its barriers also publish sc[])
    // O = P V: thread = (16-B dh chunk, key slice); 4 row loads in flight per thread
    const int ch = tid % NCH, pt = tid / NCH;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    const __half* vcol = base + 2 * E + ch * 8;
    int key = pt;
    for (; key + 3 * NPT < L; key += 4 * NPT) {
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f16x8*>(vcol + (long)(key + u * NPT) * ldq);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float pr = sc[key + u * NPT];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf(pr, (float)v[u][j], o[j]);
        }
    }
    for (; key < L; key += NPT) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(vcol + (long)key * ldq);
        const float pr = sc[key];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(pr, (float)v[j], o[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[pt][ch * 8 + j] = o[j];
    __syncthreads();
    if (tid < DH) {
        float v = 0.f;
        for (int k = 0; k < NPT; ++k) v += part[k][tid];
        v /= sum;
        const long oi = ((long)b * L + q) * E + (long)h * DH + tid;
        out[oi] = __float2half(v);
        if (out32) out32[oi] = v;
        if (tid == 0) lse[((long)b * H + h) * L + q] = mx + log2f(sum);
    }
}

// ------------------------------------------------------------------------------------------------
// NW waves per workgroup = NW * 32 queries per workgroup sharing one K/V stream: every K/V tile is staged once per
// NW * 32 queries, so 8 waves (256 queries) halve the L2 -> LDS traffic of the 4-wave form (a (b, h) streams its
// 2 x L x dh K/V bytes once per query block: 403 -> 202 MB per launch at B = 16, L = 1025), at the same waves per SIMD.
#define ATT_EDGE_MAX 8      // largest L % 128 (mean map) / L % 64 (forward keys) remainder handled off the tiles
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_kernel(const __half* __restrict__ qkv,
                                                        __half* __restrict__ out,
                                                        float* __restrict__ out32,
                                                        float* __restrict__ lse, int L, int H,
                                                        int E, int q_origin, int nqb, int Bn, int k_origin) {
    constexpr int NT = NW * 64;          // threads per workgroup
    constexpr int KS = DH / 16;          // k-steps of QK^T
    constexpr int DT = DH / 32;          // 32-row tiles of O^T
    constexpr int KROW = DH * 2 + 16;    // bytes per K row in LDS (padded)
    constexpr int VROW = DH * 2;         // bytes per V row in LDS ([key][dh], 16-B chunks XOR-swizzled)
    constexpr int KBUF = 64 * KROW, VBUF = 64 * VROW;
    constexpr int KCH = DH / 8;          // 16-B chunks per K row
    constexpr int TCH = 64 * KCH;        // 16-B chunks of one K (or V) tile
    constexpr int NKC = (TCH + NT - 1) / NT;   // K chunks per thread
    constexpr int NVC = NKC;             // V chunks per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][KBUF + VBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l31 = lane & 31;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (private L2s); give every XCD whole
    // images, heads in turn, the q blocks of a head back to back -- K/V of a (b, h) are then fetched from HBM
    // once instead of once per XCD (measured: 458 MB -> per-launch traffic near the 100 MB algorithmic).
    const int per_img = nqb * H;
    const int main_blocks = per_img * ((Bn + 7) / 8 * 8);
    if ((int)blockIdx.x >= main_blocks) {        // remainder rows [0, q_origin): one workgroup per (row, head, image)
        const int e = blockIdx.x - main_blocks;
        const int hb = e / q_origin;
        attn_row_body<DH, NT>(qkv, out, out32, lse, L, H, E, e - hb * q_origin, hb % H, hb / H, reinterpret_cast<float*>(smem));
        return;
    }
    const int slot = blockIdx.x >> 3;
    const int b = (slot / per_img) * 8 + (blockIdx.x & 7);
    if (b >= Bn) return;
    const int rem = slot - (slot / per_img) * per_img;
    const int h = rem / nqb, qb = rem - h * nqb;
    const int qrow = q_origin + qb * (NW * 32) + wave * 32 + l31;
    const int qr = qrow < L ? qrow : L - 1;
    const long ldq = 3L * E;
    const __half* base = qkv + (long)b * L * ldq + (long)h * DH;

    f16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = *reinterpret_cast<const f16x8*>(base + (long)qr * ldq + 16 * s + 8 * hh);

    u32x4 rk[NKC], rv[NVC];
    // V tile swizzle: with 128-B rows, keys q and q+2 of a 4-key block would share banks; flip the 64-B half
    // of every second key pair (64-B rows need nothing: four keys already cover the 64 banks)
#define VSWZ(key_) (DH == 64 ? ((((key_) >> 1) & 1) << 2) : 0)
    // transposing read of the PV A-operand: lane = 32*hh + 16*gi + 4*q + p supplies key row 4*hh + q (+ 8 for the
    // second read, + 16 per k-step) and dh columns 16*gi + 4*p .. +3; it receives column 16*gi + (lane & 15)
    int vaddr[2];
    {
        const int gi = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int key = 8 * r + 4 * hh + q4;                 // + 16 * s2: does not change VSWZ
            vaddr[r] = key * VROW + (((2 * gi + (p4 >> 1)) ^ VSWZ(key)) << 4) + 8 * (p4 & 1);
        }
    }
    typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
    typedef __attribute__((address_space(3))) s16x4* tr_ptr;
#undef GLOAD
#define GLOAD(t_) \
    { \
        const int t__ = (t_); \
        _Pragma("unroll") \
        for (int i = 0; i < NKC; ++i) { \
            const int c = (NKC * NT == TCH) ? tid + NT * i : (tid + NT * i) % TCH;   /* (threads beyond the tile re-fetch a chunk they do not store) */ \
            int key = k_origin + t__ * 64 + c / KCH; \
            if (key > L - 1) key = L - 1; \
            rk[i] = *reinterpret_cast<const u32x4*>(base + E + (long)key * ldq + (c % KCH) * 8); \
        } \
        _Pragma("unroll") \
        for (int i = 0; i < NVC; ++i) { \
            const int c = (NKC * NT == TCH) ? tid + NT * i : (tid + NT * i) % TCH; \
            int key = k_origin + t__ * 64 + c / KCH; \
            if (key > L - 1) key = L - 1; \
            rv[i] = *reinterpret_cast<const u32x4*>(base + 2 * E + (long)key * ldq + (c % KCH) * 8); \
        } \
    }
#undef LSTORE
#define LSTORE(buf_) \
    { \
        const int buf__ = (buf_); \
        char* kb = smem + buf__ * (KBUF + VBUF); \
        char* vb = kb + KBUF; \
        _Pragma("unroll") \
        for (int i = 0; i < NKC; ++i) { \
            const int c = tid + NT * i; \
            if (c < TCH) *reinterpret_cast<u32x4*>(kb + (c / KCH) * KROW + (c % KCH) * 16) = rk[i]; \
        } \
        _Pragma("unroll") \
        for (int i = 0; i < NVC; ++i) { \
            const int c = tid + NT * i; \
            const int key__ = c / KCH; \
            if (c < TCH) *reinterpret_cast<u32x4*>(vb + key__ * VROW + (((c % KCH) ^ VSWZ(key__)) << 4)) = rv[i]; \
        } \
    }

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = NEG_BIG, lsum = 0.f;

    const int nt = (L - k_origin + 63) / 64;     // keys [k_origin, L) in tiles of 64; keys [0, k_origin) are folded in below
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) GLOAD(t + 1);
        const char* kb = smem + buf * (KBUF + VBUF);
        const char* vb = kb + KBUF;
        f32x16 s[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(kb + l31 * KROW + ks * 32 + hh * 16);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(kb + (32 + l31) * KROW + ks * 32 + hh * 16);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[ks], s[1], 0, 0, 0);
        }
        if (t == nt - 1) {   // mask padded keys of the last tile
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k_origin + t * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= L) s[ti][r] = NEG_BIG;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        // (skipping the rescale when no query saw a new maximum was measured: the wave-uniform branch costs more
        // in lost MFMA/VALU interleaving than the 33 multiplies it saves: 99 -> 114 us)
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] *= alpha;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // packed fp32 add (v_pk_add_f32: two values per instruction) for the max subtraction and the row sum
        const f32x2 nm = {-m, -m};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 e = {s[ti][r], s[ti][r + 1]};
                e += nm;
                f32x2 p;
                p[0] = __builtin_amdgcn_exp2f(e[0]);
                p[1] = __builtin_amdgcn_exp2f(e[1]);
                s[ti][r] = p[0];
                s[ti][r + 1] = p[1];
                ps2 += p;
            }
        lsum += ps2[0] + ps2[1];
        // O^T += V^T P^T over the 64 keys = 4 k-steps of 16
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            f16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (_Float16)s[s2 >> 1][(s2 & 1) * 8 + j];
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                // dh tile d = chunks 4*d .. 4*d+3 of a row: XOR-ing the chunk index with 4*d adds d*64 B
                const char* vr = vb + s2 * 16 * VROW;
                const f16x8 va = __builtin_bit_cast(f16x8, __builtin_shufflevector(
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(vr + (vaddr[0] ^ (d << 6)))),
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(vr + (vaddr[1] ^ (d << 6)))), 0, 1, 2, 3, 4, 5, 6, 7));
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, o[d], 0, 0, 0);
            }
        }
        if (t + 1 < nt) LSTORE(buf ^ 1);
        __syncthreads();
    }
    // The first k_origin keys (L % 64 <= ATT_EDGE_MAX: the CLS token of a 1 + 32*32 sequence must not cost a 17th K/V tile
    // that holds one key): one online-softmax step per key on the VALU.  The lane pair (hh = 0, 1) of a query holds the
    // two halves of every 16-wide k-step of q; o[d][r] is dh row d*32 + (r&3) + 8*(r>>2) + 4*hh of this lane's query.
    // (8-wave form only: in the 4-wave form, which stages twice as many K/V chunks per thread, the extra code spills)
    if constexpr (NW == 8)
    for (int kx = 0; kx < k_origin; ++kx) {
        const __half* krow = base + E + (long)kx * ldq;
        float sdot = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(krow + 16 * ks + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; ++j) sdot = fmaf((float)qf[ks][j], (float)kv[j], sdot);
        }
        sdot += __shfl_xor(sdot, 32, 64);
        const float mn = fmaxf(m, sdot);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        const float pr = __builtin_amdgcn_exp2f(sdot - mn);
        m = mn;
        lsum = lsum * alpha + (hh == 0 ? pr : 0.f);          // the pair's partial sums are added once below
        const float p16 = (float)(_Float16)pr;                 // P is an fp16 MFMA operand in the tiled path
        const __half* vrow = base + 2 * E + (long)kx * ldq;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16x4 vv = *reinterpret_cast<const f16x4*>(vrow + d * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[d][g * 4 + k] = fmaf(p16, (float)vv[k], o[d][g * 4 + k] * alpha);
            }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
#ifdef WC_NO_OSTAGE
    constexpr bool OSTAGE = false;
#else
    constexpr bool OSTAGE = NW == 8;
#endif
    if constexpr (OSTAGE) {
        // O (fp16) leaves through LDS as whole rows: a lane owns a query, so direct stores are 8-byte pieces at a row pitch
        // (64 lines touched per store instruction); each wave parks its 32 x DH tile in its own slice of the (now free) K/V
        // buffers and writes 16 bytes per lane, DH * 2 contiguous bytes per row.  All waves are past the loop's last barrier.
        constexpr int OPITCH = DH * 2 + 8;                 // bytes; 8-byte aligned rows, 2-way bank spread
        constexpr int CPR = DH / 8;                        // 16-byte chunks per row
        char* stg = smem + wave * (32 * OPITCH);
    #pragma unroll
        for (int d = 0; d < DT; ++d)
    #pragma unroll
            for (int g = 0; g < 4; ++g) {
                __half hv[4];
    #pragma unroll
                for (int k = 0; k < 4; ++k) hv[k] = __float2half(o[d][g * 4 + k] * inv);
                *reinterpret_cast<uint2*>(stg + l31 * OPITCH + (d * 32 + 8 * g + 4 * hh) * 2) = *reinterpret_cast<uint2*>(hv);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int qw0 = q_origin + qb * (NW * 32) + wave * 32;          // first query row of this wave
    #pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int r = i * (64 / CPR) + lane / CPR, ch = lane % CPR;
            if (qw0 + r < L) {
                const uint2 lo2 = *reinterpret_cast<const uint2*>(stg + r * OPITCH + ch * 16);
                const uint2 hi2 = *reinterpret_cast<const uint2*>(stg + r * OPITCH + ch * 16 + 8);
                u32x4 v = {lo2.x, lo2.y, hi2.x, hi2.y};
                *reinterpret_cast<u32x4*>(out + ((long)b * L + qw0 + r) * E + (long)h * DH + ch * 8) = v;
            }
        }
    } else if (qrow < L) {          // 4-wave form (small batches): direct stores -- the staging code makes it spill (90 -> 102 us)
        __half* orow = out + ((long)b * L + qrow) * E + (long)h * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __half hv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) hv[k] = __float2half(o[d][g * 4 + k] * inv);
                *reinterpret_cast<uint2*>(orow + d * 32 + 8 * g + 4 * hh) = *reinterpret_cast<uint2*>(hv);
            }
    }
    if (qrow < L) {
        if (out32) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(out32 + ((long)b * L + qrow) * E + (long)h * DH + d * 32 + 8 * g + 4 * hh) =
                        make_float4(o[d][g * 4] * inv, o[d][g * 4 + 1] * inv, o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv);
        }
        if (hh == 0) lse[((long)b * H + h) * L + qrow] = m + log2f(ltot);
    }
}

// ------------------------------------------------------------------------------------------------
template <int DH, int R>
__global__ __launch_bounds__(256, 2) void attn_mean_kernel(const __half* __restrict__ qkv,
                                                         const float* __restrict__ lse,
                                                         float* __restrict__ mean, int L, int H, int E, int origin,
                                                         int nt, int Bn) {
    constexpr int KS = DH / 16;
    constexpr int ROW = DH * 2 + 16;
    constexpr int TB = 128 * ROW;        // bytes of one 128-row operand tile
    constexpr int CH = DH / 8;           // 16-B chunks per row
    constexpr int NC = 128 * CH / 256;   // chunks per thread per operand (4 for DH=64, 2 for 32)
    // [2][Q tile | K tile | lse 128 f32 | edge: q_e (R rows of DH halves) | k_e (R rows) | lse_e (R f32, padded to 32 B)]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EDGE = 2 * TB + 512;                  // byte offset of the edge area inside a buffer
    constexpr int BUF = EDGE + (R > 0 ? 2 * R * DH * 2 + 32 : 0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l31 = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware order (see attn_fwd_kernel): one XCD walks all tiles of an image, so that image's Q and K
    // (2 x L x E halves, ~3 MB) stay in its L2 across the nt x nt tiles
    const int slot = blockIdx.x >> 3;
    const int b = (slot / (nt * nt)) * 8 + (blockIdx.x & 7);
    if (b >= Bn) return;
    const int rem = slot - (slot / (nt * nt)) * (nt * nt);
    const int k0 = origin + (rem % nt) * 128, q0 = origin + (rem / nt) * 128;
    const long ldq = 3L * E;
    const __half* base = qkv + (long)b * L * ldq;

    // Edge entries (rows / columns < origin, the CLS token of 1 + 32*32): the workgroups of the first tile row / column
    // take them from the K / Q tile they already hold in LDS -- threads 0..127 one key each (all edge rows), threads
    // 128..255 one query each (all edge columns); the edge rows' own q / k / LSE of the current head travel with the
    // tiles (one 16-B chunk per thread of the first 2 * origin * CH threads).  R = compiled edge capacity: 0 (origin
    // == 0), 1 (the CLS-token case) or ATT_EDGE_MAX; the origin x origin corner is finished after the loop.
    const bool edge_wg = R > 0 && (q0 == origin || k0 == origin);
    const bool edge_row = R > 0 && q0 == origin && tid < 128;
    const bool edge_col = R > 0 && k0 == origin && tid >= 128;
    const int e_which = tid / (origin * CH > 0 ? origin * CH : 1);         // 0: q_e chunk, 1: k_e chunk, >= 2: none
    const int e_rem = tid - e_which * origin * CH;
    const __half* e_src = base + (long)(e_rem / CH) * ldq + (e_which ? E : 0) + (e_rem % CH) * 8;
    const int e_dst = EDGE + (e_which * R * CH + e_rem) * 16;
    u32x4 rq[NC], rk[NC], re = {0u, 0u, 0u, 0u};
    float rl = 0.f;
#undef GLOAD
#define GLOAD(h_) \
    { \
        const int h__ = (h_); \
        _Pragma("unroll") \
        for (int i = 0; i < NC; ++i) { \
            const int c = tid + 256 * i; \
            int qr = q0 + c / CH, kr = k0 + c / CH; \
            if (qr > L - 1) qr = L - 1; \
            if (kr > L - 1) kr = L - 1; \
            rq[i] = *reinterpret_cast<const u32x4*>(base + (long)qr * ldq + h__ * DH + (c % CH) * 8); \
            rk[i] = *reinterpret_cast<const u32x4*>(base + (long)kr * ldq + E + h__ * DH + (c % CH) * 8); \
        } \
        if (tid < 128) { \
            int qr = q0 + tid; \
            if (qr > L - 1) qr = L - 1; \
            rl = lse[((long)b * H + h__) * L + qr]; \
        } \
        if constexpr (R > 0) \
            if (edge_wg) { \
                if (e_which < 2) re = *reinterpret_cast<const u32x4*>(e_src + h__ * DH); \
                if (tid >= 128 && tid < 128 + origin) rl = lse[((long)b * H + h__) * L + tid - 128]; \
            } \
    }
#undef LSTORE
#define LSTORE(buf_) \
    { \
        const int buf__ = (buf_); \
        char* qb = smem + buf__ * BUF; \
        _Pragma("unroll") \
        for (int i = 0; i < NC; ++i) { \
            const int c = tid + 256 * i; \
            *reinterpret_cast<u32x4*>(qb + (c / CH) * ROW + (c % CH) * 16) = rq[i]; \
            *reinterpret_cast<u32x4*>(qb + TB + (c / CH) * ROW + (c % CH) * 16) = rk[i]; \
        } \
        if (tid < 128) reinterpret_cast<float*>(qb + 2 * TB)[tid] = rl; \
        if constexpr (R > 0) \
            if (edge_wg) { \
                if (e_which < 2) *reinterpret_cast<u32x4*>(qb + e_dst) = re; \
                if (tid >= 128 && tid < 128 + origin) reinterpret_cast<float*>(qb + EDGE + 2 * R * DH * 2)[tid - 128] = rl; \
            } \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // -LSE enters the scores through one more MFMA k-step instead of 64 accumulator moves per head: the Q side carries
    // (-lse_hi, -lse_lo, 0, ...) (two fp16 that sum to -LSE to 2^-22 relative), the K side (1, 1, 0, ...).
    f16x8 b_aug;
#pragma unroll
    for (int j = 0; j < 8; ++j) b_aug[j] = (_Float16)((hh == 0 && j < 2) ? 1.0f : 0.0f);
    float eacc[R > 0 ? R : 1];
#pragma unroll
    for (int e = 0; e < (R > 0 ? R : 1); ++e) eacc[e] = 0.f;

    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    for (int h = 0; h < H; ++h) {
        const int buf = h & 1;
        if (h + 1 < H) GLOAD(h + 1);
        const char* qb = smem + buf * BUF;
        const float* ls = reinterpret_cast<const float*>(qb + 2 * TB);
        f16x8 a_aug[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const float nl = -ls[wr * 64 + mi * 32 + l31];
            const _Float16 hi = (_Float16)nl;
            const _Float16 lo = (_Float16)(nl - (float)hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) a_aug[mi][j] = (_Float16)0.0f;
            if (hh == 0) { a_aug[mi][0] = hi; a_aug[mi][1] = lo; }
        }
        f32x16 s[2][2];
        {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            s[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_aug[0], b_aug, zero, 0, 0, 0);
            s[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_aug[0], b_aug, zero, 0, 0, 0);
            s[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_aug[1], b_aug, zero, 0, 0, 0);
            s[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_aug[1], b_aug, zero, 0, 0, 0);
        }
        const char* As = qb + (wr * 64 + l31) * ROW + hh * 16;
        const char* Bs = qb + TB + (wc * 64 + l31) * ROW + hh * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(As + ks * 32);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(As + 32 * ROW + ks * 32);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(Bs + ks * 32);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(Bs + 32 * ROW + ks * 32);
            s[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, s[0][0], 0, 0, 0);
            s[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, s[0][1], 0, 0, 0);
            s[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, s[1][0], 0, 0, 0);
            s[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, s[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] += __builtin_amdgcn_exp2f(s[mi][ni][r]);
        if constexpr (R > 0)
        if (edge_row || edge_col) {            // wave-uniform
            // own operand row (a key for the row edge, a query for the column edge) against the edge rows' q / k of
            // this head: both from LDS (the latter broadcast reads)
            const int t = tid & 127;
            const char* own = qb + (edge_row ? TB : 0) + t * ROW;
            const char* oth = qb + EDGE + (edge_row ? 0 : R * DH * 2);
            float sd[R];
#pragma unroll
            for (int e = 0; e < R; ++e) sd[e] = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f16x8 ov = *reinterpret_cast<const f16x8*>(own + c * 16);
#pragma unroll
                for (int e = 0; e < R; ++e)
                    if (R == 1 || e < origin) {
                        const f16x8 xv = *reinterpret_cast<const f16x8*>(oth + e * DH * 2 + c * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            sd[e] = __builtin_amdgcn_fdot2(f16x2{xv[2 * j], xv[2 * j + 1]}, f16x2{ov[2 * j], ov[2 * j + 1]}, sd[e], false);
                    }
            }
            const float own_lse = ls[t];      // the query's LSE (column edge)
            const float* els = reinterpret_cast<const float*>(qb + EDGE + 2 * R * DH * 2);
#pragma unroll
            for (int e = 0; e < R; ++e)
                if (R == 1 || e < origin) eacc[e] += __builtin_amdgcn_exp2f(sd[e] - (edge_row ? els[e] : own_lse));
        }
        if (h + 1 < H) LSTORE(buf ^ 1);
        __syncthreads();
    }
    const float invh = 1.0f / H;
    float* mb = mean + (long)b * L * L;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int key = k0 + wc * 64 + ni * 32 + l31;
            if (key >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (q < L) mb[(long)q * L + key] = acc[mi][ni][r] * invh;
            }
        }
    if constexpr (R > 0)
    if (edge_row || edge_col) {
        const int t = tid & 127;
#pragma unroll
        for (int e = 0; e < R; ++e)
            if (R == 1 || e < origin) {
                if (edge_row) { if (k0 + t < L) mb[(long)e * L + k0 + t] = eacc[e] * invh; }
                else if (q0 + t < L) mb[(long)(q0 + t) * L + e] = eacc[e] * invh;
            }
    }
    if (R > 0 && q0 == origin && k0 == origin && tid < origin * origin) {      // the origin x origin corner
        const int qe = tid / origin, ke = tid - qe * origin;
        float a = 0.f;
        for (int h = 0; h < H; ++h) {
            const __half* qr = base + (long)qe * ldq + h * DH;
            const __half* kr = base + (long)ke * ldq + E + h * DH;
            float sd = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f16x8 xv = *reinterpret_cast<const f16x8*>(qr + c * 8);
                const f16x8 kv = *reinterpret_cast<const f16x8*>(kr + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) sd = fmaf((float)xv[j], (float)kv[j], sd);
            }
            a += __builtin_amdgcn_exp2f(sd - lse[((long)b * H + h) * L + qe]);
        }
        mb[(long)qe * L + ke] = a * invh;
    }
}

// ------------------------------------------------------------------------------------------------
// A remainder of at most ATT_EDGE_MAX rows (L % 128) is finished by the row / edge paths; the tiles then start at it.
static int attn_origin(int L) {
    const int r = L % 128;
    return (L >= 128 && r > 0 && r <= ATT_EDGE_MAX) ? r : 0;
}

extern "C" int wc_attn_fwd(const void* qkv, void* out, float* out32, float* lse, int B, int L, int H, int DH,
                           void* stream) {
    const int E = H * DH;
    WC_CHECK_ARG(qkv && out && lse && B > 0 && L > 0, "wc_attn_fwd: bad argument");
    WC_CHECK_ARG(DH == 64 || DH == 32, "wc_attn_fwd: head dim must be 32 or 64 (got %d)", DH);
    WC_CHECK_ARG(E % 8 == 0 && B <= 65535 && H <= 65535 && (long)wc_cdiv(L, 128) * H * (B + 7) < (1L << 30),
                 "wc_attn_fwd: bad shape");
    const size_t lds = DH == 64 ? 2 * (64 * (64 * 2 + 16) + 64 * 128) : 2 * (64 * (32 * 2 + 16) + 64 * 64);
    const int nw_env = 0;      // 0: by batch size (a process switch existed through round 3)
    // 8 waves (256 queries per workgroup) when that still fills the chip; the small decoder / tiny cases keep 4
    const int NWv = nw_env ? nw_env : ((DH == 64 && (long)wc_cdiv(L, 256) * H * B >= 512) ? 8 : 4);
    const int nthr = NWv * 64;
    int r = attn_origin(L);
    if (r && ((size_t)L + 4 + 16 + (nthr / (DH / 8)) * DH) * 4 > lds) r = 0;     // row path needs its scores in LDS
    const int kr = (NWv == 8 && L >= 64 && L % 64 > 0 && L % 64 <= ATT_EDGE_MAX) ? L % 64 : 0;   // keys handled outside the K/V tiles
    const int nqb = wc_cdiv(L - r, NWv * 32);
    dim3 grid((unsigned)(nqb * H * ((B + 7) / 8 * 8) + r * H * B));
    hipStream_t st = (hipStream_t)stream;
    const int pr = wc_prof_begin(stream);
    if (DH == 64 && NWv == 8)
        hipLaunchKernelGGL((attn_fwd_kernel<64, 8>), grid, dim3(512), lds, st, (const __half*)qkv, (__half*)out, out32, lse, L, H,
                           E, r, nqb, B, kr);
    else if (DH == 64)
        hipLaunchKernelGGL((attn_fwd_kernel<64, 4>), grid, dim3(256), lds, st, (const __half*)qkv, (__half*)out, out32, lse, L, H,
                           E, r, nqb, B, kr);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<32, 4>), grid, dim3(256), lds, st, (const __half*)qkv, (__half*)out, out32, lse, L, H,
                           E, r, nqb, B, kr);
    wc_prof_end(pr, DH == 64 ? (NWv == 8 ? "attn_fwd_kernel<64, 8>" : "attn_fwd_kernel<64, 4>") : "attn_fwd_kernel<32, 4>",
                4.0 * B * H * (double)L * L * DH, stream);
    WC_LAUNCH_CHECK("attn_fwd_kernel");
    return WC_OK;
}

extern "C" int wc_attn_mean(const void* qkv, const float* lse, float* mean, int B, int L, int H, int DH,
                            void* stream) {
    const int E = H * DH;
    WC_CHECK_ARG(qkv && lse && mean && B > 0 && L > 0 && H > 0, "wc_attn_mean: bad argument");
    WC_CHECK_ARG(DH == 64 || DH == 32, "wc_attn_mean: head dim must be 32 or 64 (got %d)", DH);
    const int r = attn_origin(L);
    const int nt = wc_cdiv(L - r, 128);
    dim3 grid((unsigned)(nt * nt * ((B + 7) / 8 * 8)));
    hipStream_t st = (hipStream_t)stream;
    const int rcap = r == 0 ? 0 : (r == 1 ? 1 : ATT_EDGE_MAX);
    const size_t lds = 2 * ((size_t)2 * 128 * (DH * 2 + 16) + 512 + (rcap ? 2 * rcap * DH * 2 + 32 : 0));   // 2 x BUF of the kernel
    static bool lds_attr_set = false;
    if (!lds_attr_set) {         // more than 64 KiB of dynamic LDS
#define WC_MEAN_ATTR(DH_, R_) \
    (hipFuncSetAttribute((const void*)attn_mean_kernel<DH_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess)
        WC_CHECK_ARG(WC_MEAN_ATTR(64, 0) && WC_MEAN_ATTR(64, 1) && WC_MEAN_ATTR(64, ATT_EDGE_MAX) && WC_MEAN_ATTR(32, 0) &&
                     WC_MEAN_ATTR(32, 1) && WC_MEAN_ATTR(32, ATT_EDGE_MAX), "wc_attn_mean: cannot reserve 80 KiB of LDS");
#undef WC_MEAN_ATTR
        lds_attr_set = true;
    }
    const int pr = wc_prof_begin(stream);
#define WC_MEAN_LAUNCH(DH_, R_) \
    hipLaunchKernelGGL((attn_mean_kernel<DH_, R_>), grid, dim3(256), lds, st, (const __half*)qkv, lse, mean, L, H, E, r, nt, B)
    if (DH == 64) {
        if (r == 0) WC_MEAN_LAUNCH(64, 0); else if (r == 1) WC_MEAN_LAUNCH(64, 1); else WC_MEAN_LAUNCH(64, ATT_EDGE_MAX);
    } else {
        if (r == 0) WC_MEAN_LAUNCH(32, 0); else if (r == 1) WC_MEAN_LAUNCH(32, 1); else WC_MEAN_LAUNCH(32, ATT_EDGE_MAX);
    }
#undef WC_MEAN_LAUNCH
    static const char* const mean_names[2][3] = {{"attn_mean_kernel<64, 0>", "attn_mean_kernel<64, 1>", "attn_mean_kernel<64, 8>"},
                                                 {"attn_mean_kernel<32, 0>", "attn_mean_kernel<32, 1>", "attn_mean_kernel<32, 8>"}};
    wc_prof_end(pr, mean_names[DH == 64 ? 0 : 1][r == 0 ? 0 : (r == 1 ? 1 : 2)], 2.0 * B * H * (double)L * L * DH, stream);
    WC_LAUNCH_CHECK("attn_mean_kernel");
    return WC_OK;
}
