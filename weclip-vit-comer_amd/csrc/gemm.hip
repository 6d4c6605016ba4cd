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
// (64 accumulator VGPRs).  Operand tiles are DMA'd global -> LDS (global_load_lds_dwordx4) into an
// XOR-swizzled, unpadded image (conflict-free ds_read_b128 fragment reads), two LDS stages: the
// DMA of tile t+1 is issued before the MFMAs of tile t, one barrier per K-tile, 64 KiB LDS -> two
// workgroups per CU.  blockIdx.x walks N tiles (they share the A tile through L2), blockIdx.y M tiles.
#include "common.h"
#include <cstring>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

// Per-shape timing of the GEMM entry points (WECLIP_GEMM_LOG=1, tools/gemm_shapes.py): an event pair around every call,
// aggregated by (entry, M, N, K, segments, batch, kernel plan).  Off by default: one branch per call.
struct GemmShapeRec { char key[96]; hipEvent_t e0, e1; double flop; };
static int g_shape_log = -1;
static std::vector<GemmShapeRec> g_shape_recs;
static int shape_log_begin(void* stream) {
    if (g_shape_log < 0) g_shape_log = getenv("WECLIP_GEMM_LOG") ? atoi(getenv("WECLIP_GEMM_LOG")) : 0;
    if (!g_shape_log) return -1;
    GemmShapeRec r;
    r.key[0] = 0; r.flop = 0;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, (hipStream_t)stream);
    g_shape_recs.push_back(r);
    return (int)g_shape_recs.size() - 1;
}
static void shape_log_end(int idx, const char* kind, int M, int N, int K, int nseg, int batch, int plan, int act, void* stream) {
    if (idx < 0) return;
    GemmShapeRec& r = g_shape_recs[idx];
    snprintf(r.key, sizeof(r.key), "%s M=%d N=%d K=%d seg=%d batch=%d plan=%d act=%d", kind, M, N, K, nseg, batch, plan, act);
    r.flop = 2.0 * M * N * K * nseg * batch;
    hipEventRecord(r.e1, (hipStream_t)stream);
}
// "key\tcalls\tms\tflop" lines, cleared afterwards
extern "C" int wc_gemm_log_report(char* buf, int cap) {
    hipDeviceSynchronize();
    std::map<std::string, std::pair<double, std::pair<double, long>>> agg;
    for (auto& r : g_shape_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto& a = agg[r.key];
            a.first += ms; a.second.first += r.flop; a.second.second += 1;
        }
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_shape_recs.clear();
    int off = 0;
    if (cap > 0) buf[0] = 0;
    for (auto& kv : agg) {
        const int n = snprintf(buf + off, off < cap ? cap - off : 0, "%s\t%ld\t%.6f\t%.6e\n", kv.first.c_str(), kv.second.second.second,
                               kv.second.first, kv.second.second.first);
        if (n < 0 || off + n >= cap) break;
        off += n;
    }
    return (int)agg.size();
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    int act;              // 0 none, 1 QuickGELU x*sigmoid(1.702x), 2 ReLU, 3 sigmoid, 6 GELU (erf); 4 / 5 / 7: see aux / auxh
    int round16;          // round (acc+bias) through fp16 first (forced-fp16 out-proj, myAtt.py:321)
    float scale;          // multiply columns n < scale_cols by scale (q / sqrt(dh), myAtt.py:54)
    int scale_cols;
    float* P32;           // optional fp32 copy of the pre-activation value (acc + bias)
    const float* aux;     // act 4: v *= QuickGELU'(aux[arow*ldaux + n]), arow = rowmap[m / rpg]*rpg + m % rpg; act 7: v *= GELU'(aux[..])
    const int* rowmap;
    int row0;             // row index of this launch's first row in the caller's matrix (rowmap arithmetic after a row split)
    int rpg;
    long ldaux;
    const __half* auxh;   // act 5: v *= (auxh[m*ldaux + n] > 0)  (ReLU backward from the saved fp16 output)
    const float* cscale;  // optional per-batch column scale after bias: v *= cscale[z*sCS + n] (Dropout2d)
    long sCS;
    int gx, gy;           // tile grid (N tiles, M tiles); the launch is 1-D over gx * roundup8(gy)
    int auxvec;           // act 5: auxh rows are 8-byte addressable per 4 columns
    int vec;              // outputs / residual are 16-byte addressable per 4 columns: LDS-transposed wide epilogue
    // two-level batch (grouped small GEMMs, e.g. 11 adapters x B images in one launch): z2 = z / zdiv, z1 = z % zdiv;
    // A, W, C move by z1 * s? + z2 * s?2; bias by z2 * sB2 and the act-5 aux by z2 * sX2 (elements)
    int zdiv;
    long sA2, sW2, sC2, sB2, sX2;
};

// Epilogue shared by the kernel variants.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Every run-time option (activation, fp16 rounding, which outputs
// exist) is tested once per block of 8/16 accumulator values, never per value: the per-value scalar
// branches of a naive epilogue cost more than its stores.  Side inputs (residual / aux) of a block are
// fetched together so the loads overlap; out-of-range rows/cols read a clamped address and are not stored.
// AUX (act 4/5) is a separate instantiation so the common epilogue carries no aux registers.
#define WC_EPI_ACT(v_, n_)                                                                        \
    if (act == 1) {                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < (n_); ++e_)                                       \
            (v_)[e_] = (v_)[e_] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * (v_)[e_]));                   \
    } else if (act == 2) {                                                                        \
        _Pragma("unroll") for (int e_ = 0; e_ < (n_); ++e_) (v_)[e_] = fmaxf((v_)[e_], 0.f);      \
    } else if (act == 3) {                                                                        \
        _Pragma("unroll") for (int e_ = 0; e_ < (n_); ++e_) (v_)[e_] = __builtin_amdgcn_rcpf(1.0f + __expf(-(v_)[e_])); \
    } else if (ERF && act == 6) {      /* (erff costs registers: only in the builds that serve act 6 / 7) */ \
        _Pragma("unroll") for (int e_ = 0; e_ < (n_); ++e_)                                       \
            (v_)[e_] = 0.5f * (v_)[e_] * (1.0f + erff((v_)[e_] * 0.70710678118654752f));          \
    }

// Per-column epilogue constants of a lane's two output columns (bias, scale): fetched BEFORE the K loop of a
// tile so that their latency (and, with LDS-DMA in flight, the in-order wait behind it) is off the epilogue.
__device__ __forceinline__ void gemm_colvals(const GemmArgs& g, int n0, int wc, int lane, long zb, float (&bv)[2],
                                             float (&sc)[2], long bbase = 0) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wc * 64 + ni * 32 + (lane & 31);
        const int colc = col < g.N ? col : g.N - 1;
        bv[ni] = g.bias ? g.bias[bbase + colc] : 0.f;
        sc[ni] = (col < g.scale_cols) ? g.scale : 1.0f;
        if (g.cscale) sc[ni] *= g.cscale[zb * g.sCS + colc];
    }
}

// The same for the 16x16 MFMA layout (L16 below): a lane owns FOUR columns of the 64-wide block, 16 apart.
__device__ __forceinline__ void gemm_colvals16(const GemmArgs& g, int n0, int wc, int lane, float (&bv)[4], float (&sc)[4]) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int col = n0 + wc * 64 + ci * 16 + (lane & 15);
        const int colc = col < g.N ? col : g.N - 1;
        bv[ci] = g.bias ? g.bias[colc] : 0.f;
        sc[ci] = (col < g.scale_cols) ? g.scale : 1.0f;
        if (g.cscale) sc[ci] *= g.cscale[colc];
    }
}

// NI = column tiles (of 32) of the wave's block: 2 (64 x 64) or 1 (64 x 32: the third column tile of the 256x192 kernel).
// L16: the accumulators come from v_mfma_f32_16x16x32_f16.  A 32x32 block is then FOUR 16x16 tiles (tr, tc) packed into the
// same 16 registers, r = (tr*2 + tc)*4 + i, holding row tr*16 + (lane>>4)*4 + i, column tc*16 + (lane&15); registers 8c'..8c'+7
// still cover the rows [16c', 16c'+16) of the block, so the chunking of the wide path is unchanged, and a lane's per-column
// constants are bv / sc[ni*2 + tc] (gemm_colvals16).
template <int NI>
__device__ __forceinline__ float epi_acc(const f32x16 (&acc)[2][NI], int mi, int ni, int r) { return acc[mi][ni][r]; }
template <int NI>
__device__ __forceinline__ float epi_acc(const f32x4 (&acc)[2][NI][4], int mi, int ni, int r) { return acc[mi][ni][r >> 2][r & 3]; }

// EK = epilogue kind of the build: 0 plain (act 0..3), 1 side input (act 4 QuickGELU', act 5 ReLU'), 2 erf GELU (act 6),
// 3 erf GELU' with side input (act 7).  The erf forms live in builds of their own: carried by every build they cost the
// 128x128 kernel its second workgroup per CU (244 -> 260 registers: 65 -> 95 us on the decoder shapes, round 3).
template <int EK, int NI = 2, bool L16 = false, class ACC>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, ACC& acc, int m0, int n0, int wr, int wc,
                                              int lane, long zb, char* scratch, const float (&bv)[L16 ? 2 * NI : 2],
                                              const float (&sc)[L16 ? 2 * NI : 2], long cb, long xb = 0) {
    constexpr bool AUX = (EK & 1) != 0, ERF = EK >= 2;
    const int act = g.act;
    // layout of accumulator register r (0..15) of block (mi, ni): row inside the 32-row block, column inside the 64-wide block,
    // index of the lane's per-column constants
    // (row = lane part + compile-time part: kept apart so that row * ldc stays one lane-dependent base + scalar multiples of ldc)
    const int rowl = L16 ? ((lane >> 4) << 2) : 4 * (lane >> 5);
#define EPI_ROWC(r_) (L16 ? ((((r_) >> 3) << 4) + ((r_) & 3)) : (((r_) & 3) + 8 * (((r_) >> 2) & 3)))
#define EPI_ROW(r_) (rowl + EPI_ROWC(r_))
#define EPI_COL(ni_, r_) (L16 ? ((ni_) * 32 + ((((r_) >> 2) & 1) << 4) + (lane & 15)) : ((ni_) * 32 + (lane & 31)))
#define EPI_CI(ni_, r_) (L16 ? ((ni_) * 2 + (((r_) >> 2) & 1)) : (ni_))
    const bool has_res = g.resid != nullptr;
    const bool r16 = g.round16 != 0;
    if (g.vec && !g.P32 && g.C16 && !g.C32) {   // fp32 outputs are already 128-B coalesced per half-wave: measured slower there
        // Wide epilogue: the MFMA C layout gives a lane one column and 16 scattered rows (64 narrow
        // stores per lane, store-issue bound).  Each wave instead drops 16 finished rows at a time into
        // its own 4 KiB of LDS scratch and re-reads them row-major, 4 columns per lane: residual / aux
        // side inputs are one 16-B (8-B) load, outputs one 8-B store per 4 values.
        float* tile0 = reinterpret_cast<float*>(scratch);     // two 4-KiB buffers per wave, alternated by chunk
        const int c4 = (lane & 15) * 4;
        const int gcol = n0 + wc * 64 + c4;
        const bool full = gcol + 3 < g.N;
        const bool has_lo = g.C16lo != nullptr;
        const bool has_sc = g.scale_cols > 0 || g.cscale != nullptr;     // uniform: most launches carry no column scale
        // act 4: the fp32 aux rows (row-mapped, 16 B per lane and row) of chunk c+1 are requested before chunk c is
        // processed, so their HBM latency hides behind one chunk of epilogue work instead of stalling every chunk
        float ua[2][4][4];
        auto aux_load = [&](int c, float (&dst)[4][4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int grow = m0 + wr * 64 + c * 16 + it * 4 + (lane >> 4);
                dst[it][0] = dst[it][1] = dst[it][2] = dst[it][3] = 0.f;
                if (grow < g.M && gcol < g.N && (NI == 2 || c4 < 32)) {
                    const long arow = g.rowmap ? (long)g.rowmap[(grow + g.row0) / g.rpg] * g.rpg + (grow + g.row0) % g.rpg : grow;
                    const float* up = g.aux + arow * g.ldaux + gcol;
                    if (full) {
                        const float4 u4 = *reinterpret_cast<const float4*>(up);
                        dst[it][0] = u4.x; dst[it][1] = u4.y; dst[it][2] = u4.z; dst[it][3] = u4.w;
                    } else {
                        for (int k = 0; k < 4 && gcol + k < g.N; ++k) dst[it][k] = up[k];
                    }
                }
            }
        };
        if constexpr (AUX) {
            if (ERF ? act == 7 : act == 4) aux_load(0, ua[0]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {            // rows [16c, 16c+16) of the wave's 64x64 sub-tile
            const int mi = c >> 1, rq0 = (c & 1) * 8;
            float* tile = tile0 + (c & 1) * 1024;
            float v[NI * 8];
            if constexpr (AUX) {
                if ((ERF ? act == 7 : act == 4) && c + 1 < 4) aux_load(c + 1, ua[(c + 1) & 1]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) v[ni * 8 + rr] = epi_acc<NI>(acc, mi, ni, rq0 + rr) + bv[EPI_CI(ni, rr)];
            if (r16) {
#pragma unroll
                for (int e = 0; e < NI * 8; ++e) v[e] = __half2float(__float2half(v[e]));
            }
            if (has_sc) {
#pragma unroll
                for (int e = 0; e < NI * 8; ++e) v[e] *= sc[EPI_CI(e >> 3, e & 7)];
            }
            if constexpr (!AUX) { WC_EPI_ACT(v, NI * 8) }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
                    tile[EPI_ROW(rr) * 64 + EPI_COL(ni, rr)] = v[ni * 8 + rr];      // (rr < 8: rows 0..15 of the chunk)
            // same wave reads what it wrote: LDS ops of a wave complete in order, no barrier needed
            float f[4][4];
            bool ok[4];
            long o[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 4 + (lane >> 4);
                const int grow = m0 + wr * 64 + c * 16 + rl;
                const float4 t4 = *reinterpret_cast<const float4*>(tile + rl * 64 + c4);
                f[it][0] = t4.x; f[it][1] = t4.y; f[it][2] = t4.z; f[it][3] = t4.w;
                ok[it] = grow < g.M && gcol < g.N && (NI == 2 || c4 < 32);      // NI == 1: the lanes of columns 32..63 idle
                o[it] = (long)(grow < g.M ? grow : g.M - 1) * g.ldc + (gcol < g.N ? gcol : 0);
            }
            if constexpr (AUX) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (!ok[it]) continue;
                    const int grow = m0 + wr * 64 + c * 16 + it * 4 + (lane >> 4);
                    if (!ERF && act == 4) {
                        const float (&u)[4] = ua[c & 1][it];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * u[k]));
                            f[it][k] *= sg * (1.0f + 1.702f * u[k] * (1.0f - sg));   // d/du [u*sigmoid(1.702u)]
                        }
                    } else if (ERF && act == 7) {
                        const float (&u)[4] = ua[c & 1][it];
#pragma unroll
                        for (int k = 0; k < 4; ++k)      // d/du [u * Phi(u)] = Phi(u) + u * phi(u)
                            f[it][k] *= 0.5f * (1.0f + erff(u[k] * 0.70710678118654752f)) + u[k] * 0.3989422804014327f * __expf(-0.5f * u[k] * u[k]);
                    } else if (!ERF) {
                        const __half* hp = g.auxh + xb + (long)grow * g.ldaux + gcol;
                        if (full && g.auxvec) {          // one 8-byte load of the four saved activations
                            typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
                            const f16x4_ hv = *reinterpret_cast<const f16x4_*>(hp);
#pragma unroll
                            for (int k = 0; k < 4; ++k) f[it][k] *= (float)hv[k] > 0.f ? 1.f : 0.f;
                        } else {
                            for (int k = 0; k < 4 && gcol + k < g.N; ++k) f[it][k] *= __half2float(hp[k]) > 0.f ? 1.f : 0.f;   // ReLU'
                        }
                    }
                }
            }
            if (has_res) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (!ok[it]) continue;
                    const int grow = m0 + wr * 64 + c * 16 + it * 4 + (lane >> 4);
                    const float* rp = g.resid + zb * g.sR + (long)grow * g.ldr + gcol;
                    if (full) { const float4 rr4 = *reinterpret_cast<const float4*>(rp); f[it][0] += rr4.x; f[it][1] += rr4.y; f[it][2] += rr4.z; f[it][3] += rr4.w; }
                    else for (int k = 0; k < 4 && gcol + k < g.N; ++k) f[it][k] += rp[k];
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                __half h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = __float2half(f[it][k]);
                if (has_lo) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) l[k] = __float2half(f[it][k] - __half2float(h[k]));
                }
                if (!ok[it]) continue;
                if (full) {
                    // non-temporal: the fp16 outputs (75 MB per QKV launch) are not read again by this kernel and would
                    // push the operand rows out of the L2s (A/B on the step, interleaved: 13.45 -> 13.36 ms)
                    __builtin_nontemporal_store(*reinterpret_cast<u32x2*>(h), reinterpret_cast<u32x2*>(g.C16 + cb + o[it]));
                    if (has_lo) __builtin_nontemporal_store(*reinterpret_cast<u32x2*>(l), reinterpret_cast<u32x2*>(g.C16lo + cb + o[it]));
                } else {
                    for (int k = 0; k < 4 && gcol + k < g.N; ++k) {
                        g.C16[cb + o[it] + k] = h[k];
                        if (has_lo) g.C16lo[cb + o[it] + k] = l[k];
                    }
                }
            }
        }
        return;
    }
    // all 64 residual values of the wave's sub-tile are requested up front: one memory latency, not four
    // (the output may alias the residual, so the compiler cannot hoist these loads over the stores itself)
    float rva[AUX ? 1 : 2][AUX ? 1 : NI][16];
    if (!AUX && has_res) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int rbase = m0 + wr * 64 + mi * 32 + rowl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = n0 + wc * 64 + EPI_COL(ni, r);
                    const int colc = col < g.N ? col : g.N - 1;
                    int row = rbase + EPI_ROWC(r);
                    if (row > g.M - 1) row = g.M - 1;
                    rva[AUX ? 0 : mi][AUX ? 0 : ni][r] = g.resid[zb * g.sR + (long)row * g.ldr + colc];
                }
            }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            // a lane's columns of this block: one (32x32 layout) or two, 16 apart (L16)
            int colv[2], colcv[2];
            bool colokv[2];
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                colv[tc] = n0 + wc * 64 + EPI_COL(ni, tc * 4);
                colokv[tc] = colv[tc] < g.N;
                colcv[tc] = colokv[tc] ? colv[tc] : g.N - 1;
            }
#define EPI_TC(r_) (L16 ? (((r_) >> 2) & 1) : 0)
            const int rbase = m0 + wr * 64 + mi * 32 + rowl;
            float uv[16], v[16], pre[16];
            float (&rv)[16] = rva[AUX ? 0 : mi][AUX ? 0 : ni];
            if constexpr (AUX) {      // the aux variants are register-bound: residual per block
                if (has_res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = rbase + EPI_ROWC(r);
                        if (row > g.M - 1) row = g.M - 1;
                        rv[r] = g.resid[zb * g.sR + (long)row * g.ldr + colcv[EPI_TC(r)]];
                    }
                }
            }
            if constexpr (AUX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = rbase + EPI_ROWC(r);
                    if (row > g.M - 1) row = g.M - 1;
                    const int colc = colcv[EPI_TC(r)];
                    if (!ERF && act == 4) {
                        const long arow = g.rowmap ? (long)g.rowmap[(row + g.row0) / g.rpg] * g.rpg + (row + g.row0) % g.rpg : row;
                        const float u = g.aux[arow * g.ldaux + colc];
                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * u));
                        uv[r] = sg * (1.0f + 1.702f * u * (1.0f - sg));   // d/du [u*sigmoid(1.702u)]
                    } else if (ERF && act == 7) {
                        const long arow = g.rowmap ? (long)g.rowmap[(row + g.row0) / g.rpg] * g.rpg + (row + g.row0) % g.rpg : row;
                        const float u = g.aux[arow * g.ldaux + colc];
                        uv[r] = 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
                    } else if (!ERF) {
                        uv[r] = __half2float(g.auxh[xb + (long)row * g.ldaux + colc]) > 0.f ? 1.f : 0.f;   // ReLU'
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = epi_acc<NI>(acc, mi, ni, r) + bv[EPI_CI(ni, r)];
            if (r16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __half2float(__float2half(v[r]));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[r] *= sc[EPI_CI(ni, r)]; pre[r] = v[r]; }
            if constexpr (AUX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= uv[r];
            } else {
                WC_EPI_ACT(v, 16)
            }
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rv[r];
            }
            // element r lives at o0 + drow(r) * ldc + (its column - the lane's first column)
            const long o0 = cb + (long)rbase * g.ldc + colv[0];
#define EPI_OFF(r_) ((long)EPI_ROWC(r_) * g.ldc + (EPI_TC(r_) ? 16 : 0))
#define EPI_OK(r_) (colokv[EPI_TC(r_)] && rbase + EPI_ROWC(r_) < g.M)
            if (g.P32) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (EPI_OK(r)) g.P32[o0 + EPI_OFF(r)] = pre[r];
            }
            if (g.C32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (non-temporal here was measured on the step and is slightly slower: the fp32 outputs are the
                    // residual stream, re-read at once by the LayerNorm that follows)
                    if (EPI_OK(r)) g.C32[o0 + EPI_OFF(r)] = v[r];
                }
            }
            if (g.C16) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (EPI_OK(r)) g.C16[o0 + EPI_OFF(r)] = __float2half(v[r]);
                if (g.C16lo) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (EPI_OK(r)) g.C16lo[o0 + EPI_OFF(r)] = __float2half(v[r] - __half2float(__float2half(v[r])));
                }
            }
        }
#undef EPI_OFF
#undef EPI_OK
#undef EPI_TC
#undef EPI_ROW
#undef EPI_ROWC
#undef EPI_COL
#undef EPI_CI
}

// Main kernel.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging and no
// ds_write: the LDS store path, ~79 B/clk/CU for ds_write_b128, was the bottleneck of the register-staged
// version).  An LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB linearly (8 unpadded 128-B tile
// rows), so bank conflicts are avoided by an XOR swizzle carried on the per-lane GLOBAL source address:
// physical 16-B chunk pc of row r holds logical chunk pc ^ ((r >> 1) & 7); fragment reads apply the same
// XOR (16 consecutive rows then cover all 64 banks exactly once per ds_read_b128 lane group).
// NST = 2: two stages, one in flight, two workgroups per CU hide each other's load latency (grids of many tiles).
// NST = 4: a ring of four stages, three in flight behind counted vmcnt waits and raw s_barriers, 128 KiB: for grids of
// at most one workgroup per CU, where the 2-stage loop runs at one global-memory latency per 64-deep K-tile.
template <int EK, int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_f16_kernel(GemmArgs g) {      // NST = 2: two workgroups per CU (<= 256 registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NST stages][A tile 16 KiB | W tile 16 KiB]
    constexpr int TILE = BM * BK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so linear
    // ids l, l+8, l+16.. share an L2.  Give each XCD whole M-tile rows: its consecutive workgroups walk
    // the N tiles of one M tile and re-use the A tile from that XCD's L2 instead of 8 L2s fetching it.
    // (Only for tall grids: with fewer than 16 M tiles the remap would park the work on a few XCDs.)
    const int gx = g.gx, gy = g.gy;
    const int lin = blockIdx.x;
    int tx, ty;
    if (gy >= 16) {
        const int slot = lin >> 3;
        ty = (slot / gx) * 8 + (lin & 7);
        tx = slot - (slot / gx) * gx;
    } else {
        ty = lin / gx;
        tx = lin - ty * gx;
    }
    if (ty >= gy) return;
    const int m0 = ty * BM, n0 = tx * BN;
    const long zb = blockIdx.z;
    const long z2 = (int)blockIdx.z / g.zdiv, z1 = zb - z2 * g.zdiv;

    long aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 256 + i * 64 + lane;        // 16-B chunk index inside the tile image
        const int row = q >> 3;
        const int c = (q & 7) ^ ((row >> 1) & 7);        // logical chunk this lane must fetch
        int ar = m0 + row;
        if (ar > g.M - 1) ar = g.M - 1;
        int wrow = n0 + row;
        if (wrow > g.N - 1) wrow = g.N - 1;
        aoff[i] = z1 * g.sA + z2 * g.sA2 + (long)ar * g.lda + c * 8;
        woff[i] = z1 * g.sW + z2 * g.sW2 + (long)wrow * g.ldw + c * 8;
    }
    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
#define GLDS(t, buf)                                                                                          \
    {                                                                                                         \
        const int seg_ = (t) / ktiles;                                                                        \
        const long k0_ = (long)((t) - seg_ * ktiles) * BK;                                                    \
        const __half* Ap_ = seg_ == 0 ? g.A[0] : (seg_ == 1 ? g.A[1] : g.A[2]);                               \
        const __half* Wp_ = seg_ == 0 ? g.W[0] : (seg_ == 1 ? g.W[1] : g.W[2]);                               \
        char* dst_ = smem + (buf) * (2 * TILE) + wave * 4096;                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
            __builtin_amdgcn_global_load_lds((gbl_ptr)(Ap_ + aoff[i] + k0_), (lds_ptr)(dst_ + i * 1024), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((gbl_ptr)(Wp_ + woff[i] + k0_), (lds_ptr)(dst_ + TILE + i * 1024), 16, 0, 0); \
        }                                                                                                     \
    }
    // fragment read addresses (bytes inside an operand tile) for this lane
    const int hh = lane >> 5, l31 = lane & 31;
    int arow[2], aswz[2], brow[2], bswz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
        arow[i] = ra * 128; aswz[i] = (ra >> 1) & 7;
        brow[i] = rb * 128; bswz[i] = (rb >> 1) & 7;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float bv[2], sc[2];
    gemm_colvals(g, n0, wc, lane, zb, bv, sc, z2 * g.sB2);
    // software-pipelined fragment reads: the ds_reads of k-step ks+1 are issued before the MFMAs of
    // ks (the compiler otherwise parks the wave on lgkmcnt(0) in front of every MFMA group);
    // sched_barrier(0) pins the issue order (hipcc otherwise sinks the reads back in front of their use)
#define FRAG_LOAD(ks_, a0_, a1_, b0_, b1_)                                                           \
        {                                                                                            \
            const int ch_ = 2 * (ks_) + hh;                                                          \
            a0_ = *reinterpret_cast<const f16x8*>(As + arow[0] + ((ch_ ^ aswz[0]) << 4));            \
            a1_ = *reinterpret_cast<const f16x8*>(As + arow[1] + ((ch_ ^ aswz[1]) << 4));            \
            b0_ = *reinterpret_cast<const f16x8*>(Ws + brow[0] + ((ch_ ^ bswz[0]) << 4));            \
            b1_ = *reinterpret_cast<const f16x8*>(Ws + brow[1] + ((ch_ ^ bswz[1]) << 4));            \
        }
#define FRAG_MMA(a0_, a1_, b0_, b1_)                                                                 \
        {                                                                                            \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_, b0_, acc[0][0], 0, 0, 0);        \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_, b1_, acc[0][1], 0, 0, 0);        \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_, b0_, acc[1][0], 0, 0, 0);        \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_, b1_, acc[1][1], 0, 0, 0);        \
        }
#define STAGE_MMA(buf_)                                                                              \
    {                                                                                                \
        const char* As = smem + (buf_) * (2 * TILE);                                                 \
        const char* Ws = As + TILE;                                                                  \
        f16x8 fa0, fa1, fb0, fb1, na0, na1, nb0, nb1;                                                \
        FRAG_LOAD(0, fa0, fa1, fb0, fb1);                                                            \
        FRAG_LOAD(1, na0, na1, nb0, nb1);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        FRAG_MMA(fa0, fa1, fb0, fb1);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        FRAG_LOAD(2, fa0, fa1, fb0, fb1);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        FRAG_MMA(na0, na1, nb0, nb1);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        FRAG_LOAD(3, na0, na1, nb0, nb1);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        FRAG_MMA(fa0, fa1, fb0, fb1);                                                                \
        FRAG_MMA(na0, na1, nb0, nb1);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    }
    if constexpr (NST == 2) {
        GLDS(0, 0);
        __syncthreads();     // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) GLDS(t + 1, buf ^ 1);
            STAGE_MMA(buf);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int s_ = 0; s_ < NST - 1; ++s_)
            if (s_ < nt) GLDS(s_, s_);
        int buf = 0, nbuf = NST - 1;
        for (int t = 0; t < nt; ++t) {
            // stage t has landed once at most the NST - 2 later requests (8 DMA instructions each) are outstanding
            const int ahead = nt - 1 - t;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // all parts of stage t are in LDS; stage t-1's buffer is free
            if (t + NST - 1 < nt) GLDS(t + NST - 1, nbuf);
            STAGE_MMA(buf);
            buf = buf + 1 == NST ? 0 : buf + 1;
            nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
        }
        __syncthreads();                      // the epilogue reuses the ring as scratch
    }
#undef STAGE_MMA
#undef FRAG_LOAD
#undef FRAG_MMA
#undef GLDS
    gemm_epilogue<EK>(g, acc, m0, n0, wr, wc, lane, zb, smem + wave * 8192, bv, sc, z1 * g.sC + z2 * g.sC2, z2 * g.sX2);
}

// ---------------------------------------------------------------------------------------------
// M <= 32 rows (the ragged last rows of a tall GEMM: 16 images x 1025 tokens leave 16 rows past the last 256-row tile;
// the 128x128 kernel ran them as N/128 workgroups, each pulling its 128 weight rows through ONE CU's ~35 GB/s miss path:
// 17.8 us for 16 x 3072 x 768).  Here a workgroup owns 64 output columns, its four waves split K, operands go straight
// from global memory to MFMA fragments (no LDS: nothing is shared between waves), all loads of a chunk of 12 k-steps in
// flight at once; the four partial tiles meet in LDS and wave 0 runs the shared epilogue.
template <int EK>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float red[3 * 32 * 64];      // partial tiles of waves 1..3: [wave-1][reg][lane]
    __shared__ __attribute__((aligned(16))) char scr[8192];              // epilogue scratch of wave 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.x * 64;
    const int arow = l31 < g.M ? l31 : g.M - 1;
    int wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        wrow[j] = n0 + j * 32 + l31;
        if (wrow[j] > g.N - 1) wrow[j] = g.N - 1;
    }
    const int per = g.K / 64;                    // k-steps (of 16) per wave; K % 64 == 0
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bv[2], sc[2];
    if (wave == 0) gemm_colvals(g, n0, 0, lane, 0, bv, sc, 0);
    for (int seg = 0; seg < g.nseg; ++seg) {
        const __half* Ap = (seg == 0 ? g.A[0] : (seg == 1 ? g.A[1] : g.A[2])) + (long)arow * g.lda + (long)wave * per * 16 + hh * 8;
        const __half* Wb = seg == 0 ? g.W[0] : (seg == 1 ? g.W[1] : g.W[2]);
        const __half* Wp0 = Wb + (long)wrow[0] * g.ldw + (long)wave * per * 16 + hh * 8;
        const __half* Wp1 = Wb + (long)wrow[1] * g.ldw + (long)wave * per * 16 + hh * 8;
        for (int ks0 = 0; ks0 < per; ks0 += 12) {
            f16x8 fa[12], fb0[12], fb1[12];
            // a lane pair reads 32 contiguous bytes of its row per k-step, so a 128-B line serves four consecutive k-steps:
            // the four loads of a line are issued back to back per operand (with A, B0, B1 interleaved per k-step the four
            // waves' streams pushed each line out of the 32-KiB L1 before its next use: 4x over-fetch from L2)
#pragma unroll
            for (int i4 = 0; i4 < 12; i4 += 4) {
#pragma unroll
                for (int i = i4; i < i4 + 4; ++i)
                    if (ks0 + i < per) fb0[i] = *reinterpret_cast<const f16x8*>(Wp0 + (ks0 + i) * 16);
#pragma unroll
                for (int i = i4; i < i4 + 4; ++i)
                    if (ks0 + i < per) fb1[i] = *reinterpret_cast<const f16x8*>(Wp1 + (ks0 + i) * 16);
#pragma unroll
                for (int i = i4; i < i4 + 4; ++i)
                    if (ks0 + i < per) fa[i] = *reinterpret_cast<const f16x8*>(Ap + (ks0 + i) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (ks0 + i < per) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb0[i], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb1[i], acc[0][1], 0, 0, 0);
                }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 32 + j * 16 + r) * 64 + lane] = acc[0][j][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] += red[(w * 32 + j * 16 + r) * 64 + lane];
    gemm_epilogue<EK>(g, acc, 0, n0, 0, 0, lane, 0, scr, bv, sc, 0);
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 "ping-pong" kernel for tall GEMMs (the token dimension): ONE workgroup of 8 waves per CU,
// wave (wr, wc) = (wave>>2, wave&3) owns a 128x64 block of the tile (128 accumulator registers).
//
// The 128x128 kernel above is bound by the L2->LDS path (32 KiB of operands per 2.1 MFLOP); this tile needs
// half the bytes per flop.  Each K-tile is staged as FOUR 16-KiB half-tiles, in the order they are consumed:
//   A0 = rows 0..63 of both row groups, B0 = columns 0..31 of all four column groups, B1 = columns 32..63,
//   A1 = rows 64..127 -- and computed in four phases: (A0,B0) (A0,B1) (A1,B1) (A1,B0), 8 MFMAs each.
// LDS holds 8 half-tile slots (128 KiB).  Phase p reads its fragments from half-tiles <= p+1, waits (counted
// vmcnt) until half-tile p+2 has landed -- three half-tiles stay in flight across the barriers -- and issues
// the LDS-DMA of half-tile p+6 between its MFMAs; a slot is re-staged at least two phases after its last read.
// The two row groups run one barrier apart: while the waves of one group multiply (and issue DMA), the other
// group's waves (their SIMD neighbours) read fragments, so LDS reads, DMA and MFMA overlap.
//
// M16 (round 3): the same schedule on v_mfma_f32_16x16x32_f16 -- per phase 16 MFMAs of 16 cycles instead of 8 of 32, the same
// fragment bytes (a 16-B fragment is now 16 rows x 8 of the 32 k of an MFMA: lane l reads row l & 15, 16-B chunk l >> 4; the
// XOR swizzle is conflict-free for that pattern as well) and the same 128 accumulator registers.  The matrix pipe takes the
// same cycles either way; what differs is the clock the chip holds under the load (MI355X_MICROARCH.md, DVFS item 7).
#define PP_SLOT 16384
// RING = half-tile slots in LDS: 8 (128 KiB, three half-tiles in flight across the barriers) or 10 (all 160 KiB, FIVE in flight;
// the default).  A fifth of the staged lines miss the XCD's L2 and come from the Infinity Cache, and a half-tile is complete
// only when its slowest line is; two more half-tiles in flight give 1-3 % (8192^3: 907 -> 881 us, QKV 78.9 -> 76.4 us,
// bit-identical outputs) -- the depth of the ring is a small part of the transport side's 41 GB/s per CU, not its cause.
template <int EK, bool M16, int RING = 8>
__global__ __launch_bounds__(512) void gemm_f16_pp_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // RING half-tile slots [128 rows][64 halfs], XOR-swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int gx = g.gx, gy = g.gy;
    const int lin = blockIdx.x;
    int tx, ty;
    if (gy >= 16) {          // XCD-aware order, as in the 128x128 kernel
        const int slot = lin >> 3;
        ty = (slot / gx) * 8 + (lin & 7);
        tx = slot - (slot / gx) * gx;
    } else {
        ty = lin / gx;
        tx = lin - ty * gx;
    }
    if (ty >= gy) return;
    const int m0 = ty * 256, n0 = tx * 256;

    // per-thread DMA sources of the four half-tile kinds (two 16-B chunks each)
    // (32-bit BYTE offsets from the operand's base: the launcher keeps operands of 4 GiB and more off this kernel; with a
    // uniform 64-bit base the LDS-DMA takes them as its 32-bit VGPR offset, no 64-bit vector add per instruction)
    unsigned offA0[2], offA1[2], offB0[2], offB1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;                 // chunk index inside the half-tile image
        const int row = q >> 3;
        const int c = (q & 7) ^ ((row >> 1) & 7);    // logical chunk this lane must fetch (swizzle on the source)
        int a0 = m0 + (row >> 6) * 128 + (row & 63), a1 = a0 + 64;
        int b0 = n0 + (row >> 5) * 64 + (row & 31), b1 = b0 + 32;
        a0 = a0 < g.M ? a0 : g.M - 1; a1 = a1 < g.M ? a1 : g.M - 1;
        b0 = b0 < g.N ? b0 : g.N - 1; b1 = b1 < g.N ? b1 : g.N - 1;
        offA0[i] = (unsigned)(((long)a0 * g.lda + c * 8) * 2); offA1[i] = (unsigned)(((long)a1 * g.lda + c * 8) * 2);
        offB0[i] = (unsigned)(((long)b0 * g.ldw + c * 8) * 2); offB1[i] = (unsigned)(((long)b1 * g.ldw + c * 8) * 2);
    }
    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;
    const int nj = 4 * nt;                           // half-tiles in the stream
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    // half-tile j = 4*t + q (q: 0 A0, 1 B0, 2 B1, 3 A1) goes to slot j & 7; q_ and slot_ are compile-time
#define PP_STAGE(t_, q_, slot_)                                                                              \
    {                                                                                                         \
        const int seg__ = (t_) / ktiles;                                                                      \
        const long k0__ = (long)((t_) - seg__ * ktiles) * BK;                                                 \
        const __half* P__ = ((q_) == 0 || (q_) == 3) ? (seg__ == 0 ? g.A[0] : (seg__ == 1 ? g.A[1] : g.A[2])) \
                                                     : (seg__ == 0 ? g.W[0] : (seg__ == 1 ? g.W[1] : g.W[2])); \
        const unsigned o0__ = (q_) == 0 ? offA0[0] : (q_) == 1 ? offB0[0] : (q_) == 2 ? offB1[0] : offA1[0];  \
        const unsigned o1__ = (q_) == 0 ? offA0[1] : (q_) == 1 ? offB0[1] : (q_) == 2 ? offB1[1] : offA1[1];  \
        char* d__ = smem + (slot_) * PP_SLOT + wave * 1024;                                                   \
        const char* B__ = reinterpret_cast<const char*>(P__ + k0__);                                          \
        __builtin_amdgcn_global_load_lds((gbl_ptr)(B__ + o0__), (lds_ptr)d__, 16, 0, 0);                      \
        __builtin_amdgcn_global_load_lds((gbl_ptr)(B__ + o1__), (lds_ptr)(d__ + 8192), 16, 0, 0);             \
    }
    // leave the n_ newest half-tiles (2 DMA instructions each) in flight
#define PP_WAIT(n_)                                                                 \
    {                                                                               \
        const int w__ = (n_);                                                       \
        if (RING == 10 && w__ >= 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   \
        else if (RING == 10 && w__ == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); \
        else if (w__ >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         \
        else if (w__ == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         \
        else if (w__ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         \
        else if (w__ == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
    }
    // fragment read addresses (bytes inside a slot): A rows wr*64 + mi*32 + l31, B rows wc*32 + l31
    // M16: fragment x of a 32-row block is (row tile tr = x >> 1, k half x & 1): rows tr*16 + (lane & 15), chunk (x & 1)*4 + (lane >> 4)
    const int hh = lane >> 5, l31 = lane & 31, q16 = lane >> 4, l15 = lane & 15;
    int aaddr[2][4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int ra = M16 ? wr * 64 + mi * 32 + (ks >> 1) * 16 + l15 : wr * 64 + mi * 32 + l31;
            const int ch = M16 ? (ks & 1) * 4 + q16 : 2 * ks + hh;
            aaddr[mi][ks] = ra * 128 + ((ch ^ ((ra >> 1) & 7)) << 4);
        }
        const int rb = M16 ? wc * 32 + (ks >> 1) * 16 + l15 : wc * 32 + l31;
        const int ch = M16 ? (ks & 1) * 4 + q16 : 2 * ks + hh;
        baddr[ks] = rb * 128 + ((ch ^ ((rb >> 1) & 7)) << 4);
    }
    f32x16 acc[M16 ? 1 : 2][2][2];      // [row half a][mi][column half b]
    f32x4 acc4[M16 ? 2 : 1][2][2][4];   // M16: [a][mi][b][tr*2 + tc]
    if constexpr (M16) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[a][i][j][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
    }
    float bv[M16 ? 4 : 2], sc[M16 ? 4 : 2];
    if constexpr (M16) gemm_colvals16(g, n0, wc, lane, bv, sc);
    else gemm_colvals(g, n0, wc, lane, 0, bv, sc);

    // prologue: half-tiles 0..5 (0..7 with the 10-slot ring; nt >= 2 is guaranteed by the launcher), the first two landed
    PP_STAGE(0, 0, 0); PP_STAGE(0, 1, 1); PP_STAGE(0, 2, 2); PP_STAGE(0, 3, 3); PP_STAGE(1, 0, 4); PP_STAGE(1, 1, 5);
    if constexpr (RING == 10) {
        PP_STAGE(1, 2, 6); PP_STAGE(1, 3, 7);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // row group 1 runs one barrier behind group 0

    f16x8 fa[2][4], fb0[4], fb1[4];
#define PP_READ_A(slot_)                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                        \
        fa[0][ks] = *reinterpret_cast<const f16x8*>(smem + (slot_) * PP_SLOT + aaddr[0][ks]);                 \
        fa[1][ks] = *reinterpret_cast<const f16x8*>(smem + (slot_) * PP_SLOT + aaddr[1][ks]);                 \
    }
#define PP_READ_B(fb_, slot_)                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                          \
        fb_[ks] = *reinterpret_cast<const f16x8*>(smem + (slot_) * PP_SLOT + baddr[ks]);
    // M16: MFMA (k2, mi, tr, tc) multiplies fragment fa[mi][tr*2 + k2] with fb[tc*2 + k2] into acc4[a][mi][b][tr*2 + tc]; the
    // quarter q_ (0..3) of a phase is k half k2 = q_ >> 1, mi = q_ & 1: four MFMAs on four different accumulators
#define PP_MMA16_Q(a_, fb_, b_, q_)                                                                           \
    _Pragma("unroll") for (int tt = 0; tt < 4; ++tt)                                                          \
        acc4[a_][(q_) & 1][b_][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                  \
            fa[(q_) & 1][(tt >> 1) * 2 + ((q_) >> 1)], fb_[(tt & 1) * 2 + ((q_) >> 1)], acc4[a_][(q_) & 1][b_][tt], 0, 0, 0);
#define PP_PIN16(a_, b_)                                                                                      \
    asm volatile("" : "+v"(acc4[a_][0][b_][0]), "+v"(acc4[a_][0][b_][1]), "+v"(acc4[a_][0][b_][2]), "+v"(acc4[a_][0][b_][3]),  \
                      "+v"(acc4[a_][1][b_][0]), "+v"(acc4[a_][1][b_][1]), "+v"(acc4[a_][1][b_][2]), "+v"(acc4[a_][1][b_][3]));
#define PP_MMA(a_, fb_, b_)                                                                                   \
    if constexpr (M16) {                                                                                      \
        PP_MMA16_Q(a_, fb_, b_, 0) PP_MMA16_Q(a_, fb_, b_, 1) PP_MMA16_Q(a_, fb_, b_, 2) PP_MMA16_Q(a_, fb_, b_, 3) \
        PP_PIN16(a_, b_)                                                                                      \
    } else {                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                    \
            acc[a_][0][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb_[ks], acc[a_][0][b_], 0, 0, 0); \
            acc[a_][1][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb_[ks], acc[a_][1][b_], 0, 0, 0); \
        }                                                                                                     \
        asm volatile("" : "+v"(acc[a_][0][b_]), "+v"(acc[a_][1][b_]));   /* keeps the MFMAs inside their phase */ \
    }
    // one phase of the guarded form (last K-tiles): [fragment reads] -> DMA of half-tile phi+6 -> counted wait ->
    // barrier -> 8 MFMAs -> barrier
#define PP_PHASE(phi_, READS_, tj_, qj_, slotj_, MMA_)                                                        \
    {                                                                                                         \
        READS_;                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if ((phi_) + (RING - 2) < nj) PP_STAGE(tj_, qj_, slotj_);                                             \
        PP_WAIT(nj - 3 - (phi_));                                                                             \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        MMA_;                                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_barrier();                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // Steady state (every stage target exists): no guards, constant vmcnt.  The two DMA instructions of a phase are
    // issued INSIDE the MFMA segment (after the 1st and the 3rd MFMA pair) and fetch half-tile phi+6: in-kernel
    // s_memtime stamps showed an LDS-DMA wave-instruction costing its wave 125-165 cycles of issue time when the
    // four waves of a row group issue together right after a barrier (the CU's address path takes ~38 cycles
    // per 1-KiB piece), which made the read segment (reads + 2 DMA, 360-460 cycles) the long pole beside the
    // partner's 8 MFMAs (~290).  Among the MFMAs the issue stall overlaps the matrix pipe's own latency.
    // The wait before the barrier then leaves three half-tiles in flight (phi+3..phi+5) and half-tile phi+2 landed.
#define PP_STAGE_H(t_, q_, slot_, h_)                                                                         \
    {                                                                                                         \
        const int seg__ = (t_) / ktiles;                                                                      \
        const long k0__ = (long)((t_) - seg__ * ktiles) * BK;                                                 \
        const __half* P__ = ((q_) == 0 || (q_) == 3) ? (seg__ == 0 ? g.A[0] : (seg__ == 1 ? g.A[1] : g.A[2])) \
                                                     : (seg__ == 0 ? g.W[0] : (seg__ == 1 ? g.W[1] : g.W[2])); \
        unsigned o__ = (q_) == 0 ? offA0[h_] : (q_) == 1 ? offB0[h_] : (q_) == 2 ? offB1[h_] : offA1[h_];     \
        asm volatile("" : "+v"(o__));      /* keeps the 32-bit offset a 32-bit register (no hoisted 64-bit copy) */ \
        char* d__ = smem + (slot_) * PP_SLOT + wave * 1024 + (h_) * 8192;                                     \
        const char* B__ = reinterpret_cast<const char*>(P__ + k0__);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_global_load_lds((gbl_ptr)(B__ + o__), (lds_ptr)d__, 16, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define PP_PHASE_S(READS_, tj_, qj_, slotj_, a_, fb_, b_)                                                     \
    {                                                                                                         \
        READS_;                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (RING == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                           \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        if constexpr (M16) {                                                                                  \
            PP_MMA16_Q(a_, fb_, b_, 0)                                                                        \
            PP_STAGE_H(tj_, qj_, slotj_, 0)                                                                   \
            PP_MMA16_Q(a_, fb_, b_, 1) PP_MMA16_Q(a_, fb_, b_, 2)                                             \
            PP_STAGE_H(tj_, qj_, slotj_, 1)                                                                   \
            PP_MMA16_Q(a_, fb_, b_, 3)                                                                        \
            PP_PIN16(a_, b_)                                                                                  \
        } else {                                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                \
                acc[a_][0][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb_[ks], acc[a_][0][b_], 0, 0, 0); \
                acc[a_][1][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb_[ks], acc[a_][1][b_], 0, 0, 0); \
                if (ks == 0) PP_STAGE_H(tj_, qj_, slotj_, 0)                                                  \
                if (ks == 2) PP_STAGE_H(tj_, qj_, slotj_, 1)                                                  \
            }                                                                                                 \
            asm volatile("" : "+v"(acc[a_][0][b_]), "+v"(acc[a_][1][b_]));                                    \
        }                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_barrier();                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    int t = 0;
    if constexpr (RING == 10) {
        // half-tile j -> slot j % 10: the pattern repeats every 20 half-tiles = 5 K-tiles; phase c stages half-tile c + 8
        for (; t + 7 <= nt; t += 5) {
            PP_PHASE_S(PP_READ_B(fb0, 1) PP_READ_A(0), t + 2, 0, 8, 0, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb1, 2), t + 2, 1, 9, 0, fb1, 1);
            PP_PHASE_S(PP_READ_A(3), t + 2, 2, 0, 1, fb1, 1);
            PP_PHASE_S(, t + 2, 3, 1, 1, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb0, 5) PP_READ_A(4), t + 3, 0, 2, 0, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb1, 6), t + 3, 1, 3, 0, fb1, 1);
            PP_PHASE_S(PP_READ_A(7), t + 3, 2, 4, 1, fb1, 1);
            PP_PHASE_S(, t + 3, 3, 5, 1, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb0, 9) PP_READ_A(8), t + 4, 0, 6, 0, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb1, 0), t + 4, 1, 7, 0, fb1, 1);
            PP_PHASE_S(PP_READ_A(1), t + 4, 2, 8, 1, fb1, 1);
            PP_PHASE_S(, t + 4, 3, 9, 1, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb0, 3) PP_READ_A(2), t + 5, 0, 0, 0, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb1, 4), t + 5, 1, 1, 0, fb1, 1);
            PP_PHASE_S(PP_READ_A(5), t + 5, 2, 2, 1, fb1, 1);
            PP_PHASE_S(, t + 5, 3, 3, 1, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb0, 7) PP_READ_A(6), t + 6, 0, 4, 0, fb0, 0);
            PP_PHASE_S(PP_READ_B(fb1, 8), t + 6, 1, 5, 0, fb1, 1);
            PP_PHASE_S(PP_READ_A(9), t + 6, 2, 6, 1, fb1, 1);
            PP_PHASE_S(, t + 6, 3, 7, 1, fb0, 0);
        }
        for (; t < nt; t += 5) {          // the last K-tiles: guarded stages, draining waits (t % 5 == 0 here)
            const int phi = 4 * t;
            PP_PHASE(phi + 0, PP_READ_B(fb0, 1) PP_READ_A(0), t + 2, 0, 8, PP_MMA(0, fb0, 0));
            PP_PHASE(phi + 1, PP_READ_B(fb1, 2), t + 2, 1, 9, PP_MMA(0, fb1, 1));
            PP_PHASE(phi + 2, PP_READ_A(3), t + 2, 2, 0, PP_MMA(1, fb1, 1));
            PP_PHASE(phi + 3, , t + 2, 3, 1, PP_MMA(1, fb0, 0));
            if (t + 1 < nt) {
                PP_PHASE(phi + 4, PP_READ_B(fb0, 5) PP_READ_A(4), t + 3, 0, 2, PP_MMA(0, fb0, 0));
                PP_PHASE(phi + 5, PP_READ_B(fb1, 6), t + 3, 1, 3, PP_MMA(0, fb1, 1));
                PP_PHASE(phi + 6, PP_READ_A(7), t + 3, 2, 4, PP_MMA(1, fb1, 1));
                PP_PHASE(phi + 7, , t + 3, 3, 5, PP_MMA(1, fb0, 0));
            }
            if (t + 2 < nt) {
                PP_PHASE(phi + 8, PP_READ_B(fb0, 9) PP_READ_A(8), t + 4, 0, 6, PP_MMA(0, fb0, 0));
                PP_PHASE(phi + 9, PP_READ_B(fb1, 0), t + 4, 1, 7, PP_MMA(0, fb1, 1));
                PP_PHASE(phi + 10, PP_READ_A(1), t + 4, 2, 8, PP_MMA(1, fb1, 1));
                PP_PHASE(phi + 11, , t + 4, 3, 9, PP_MMA(1, fb0, 0));
            }
            if (t + 3 < nt) {
                PP_PHASE(phi + 12, PP_READ_B(fb0, 3) PP_READ_A(2), t + 5, 0, 0, PP_MMA(0, fb0, 0));
                PP_PHASE(phi + 13, PP_READ_B(fb1, 4), t + 5, 1, 1, PP_MMA(0, fb1, 1));
                PP_PHASE(phi + 14, PP_READ_A(5), t + 5, 2, 2, PP_MMA(1, fb1, 1));
                PP_PHASE(phi + 15, , t + 5, 3, 3, PP_MMA(1, fb0, 0));
            }
            if (t + 4 < nt) {
                PP_PHASE(phi + 16, PP_READ_B(fb0, 7) PP_READ_A(6), t + 6, 0, 4, PP_MMA(0, fb0, 0));
                PP_PHASE(phi + 17, PP_READ_B(fb1, 8), t + 6, 1, 5, PP_MMA(0, fb1, 1));
                PP_PHASE(phi + 18, PP_READ_A(9), t + 6, 2, 6, PP_MMA(1, fb1, 1));
                PP_PHASE(phi + 19, , t + 6, 3, 7, PP_MMA(1, fb0, 0));
            }
        }
    } else {
    for (; t + 4 <= nt; t += 2) {
        PP_PHASE_S(PP_READ_B(fb0, 1) PP_READ_A(0), t + 1, 2, 6, 0, fb0, 0);
        PP_PHASE_S(PP_READ_B(fb1, 2), t + 1, 3, 7, 0, fb1, 1);
        PP_PHASE_S(PP_READ_A(3), t + 2, 0, 0, 1, fb1, 1);
        PP_PHASE_S(, t + 2, 1, 1, 1, fb0, 0);
        PP_PHASE_S(PP_READ_B(fb0, 5) PP_READ_A(4), t + 2, 2, 2, 0, fb0, 0);
        PP_PHASE_S(PP_READ_B(fb1, 6), t + 2, 3, 3, 0, fb1, 1);
        PP_PHASE_S(PP_READ_A(7), t + 3, 0, 4, 1, fb1, 1);
        PP_PHASE_S(, t + 3, 1, 5, 1, fb0, 0);
    }
    for (; t < nt; t += 2) {          // the last K-tiles: guarded stages, draining waits
        const int phi = 4 * t;
        // K-tile t (even): slots 0..3; stages half-tiles phi+6.. = (t+1: B1 A1), (t+2: A0 B0)
        PP_PHASE(phi + 0, PP_READ_B(fb0, 1) PP_READ_A(0), t + 1, 2, 6, PP_MMA(0, fb0, 0));
        PP_PHASE(phi + 1, PP_READ_B(fb1, 2), t + 1, 3, 7, PP_MMA(0, fb1, 1));
        PP_PHASE(phi + 2, PP_READ_A(3), t + 2, 0, 0, PP_MMA(1, fb1, 1));
        PP_PHASE(phi + 3, , t + 2, 1, 1, PP_MMA(1, fb0, 0));
        if (t + 1 < nt) {
            // K-tile t+1 (odd): slots 4..7; stages (t+2: B1 A1), (t+3: A0 B0)
            PP_PHASE(phi + 4, PP_READ_B(fb0, 5) PP_READ_A(4), t + 2, 2, 2, PP_MMA(0, fb0, 0));
            PP_PHASE(phi + 5, PP_READ_B(fb1, 6), t + 2, 3, 3, PP_MMA(0, fb1, 1));
            PP_PHASE(phi + 6, PP_READ_A(7), t + 3, 0, 4, PP_MMA(1, fb1, 1));
            PP_PHASE(phi + 7, , t + 3, 1, 5, PP_MMA(1, fb0, 0));
        }
    }
    }
#undef PP_STAGE_H
#undef PP_PHASE_S
#undef PP_PHASE
#undef PP_MMA
#undef PP_MMA16_Q
#undef PP_PIN16
#undef PP_READ_A
#undef PP_READ_B
#undef PP_WAIT
#undef PP_STAGE
    if (wr == 0) __builtin_amdgcn_s_barrier();       // re-align the two row groups
    __syncthreads();                                 // every wave is done with the operand slots: epilogue scratch
    if constexpr (M16) {      // (the epilogue reads the 16x16 tiles as register r = (tr*2 + tc)*4 + i of a 32x32 block)
        gemm_epilogue<EK, 2, true>(g, acc4[0], m0 + wr * 128, n0, 0, wc, lane, 0, smem + wave * 8192, bv, sc, 0);
        gemm_epilogue<EK, 2, true>(g, acc4[1], m0 + wr * 128 + 64, n0, 0, wc, lane, 0, smem + wave * 8192, bv, sc, 0);
    } else {
        gemm_epilogue<EK, 2, false>(g, acc[0], m0 + wr * 128, n0, 0, wc, lane, 0, smem + wave * 8192, bv, sc, 0);
        gemm_epilogue<EK, 2, false>(g, acc[1], m0 + wr * 128 + 64, n0, 0, wc, lane, 0, smem + wave * 8192, bv, sc, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 kernel with FOUR waves (round 3): one wave per SIMD, wave (wr, wc) = (wave >> 1, wave & 1) owns a 128x128 block
// (4 x 4 MFMA tiles, 256 accumulator registers: one wave per SIMD may use all 512 registers of a lane).
//
// Why: the ping-pong kernel above brings its operands in by LDS-DMA, and the CU's address path takes ~38 cycles per 1-KiB
// DMA piece (in-kernel stamps, round 2): 64 pieces per 64-deep K-tile = ~2400 cycles against the 2048 cycles the K-tile's
// MFMAs take -- its K loop is bound by the DMA issue path (1.26 us per K-tile with 24 CUs busy), which is why its time follows
// the bytes staged and not the number of busy CUs.  Here the operands take the ordinary vector-memory path instead:
// global_load_dwordx4 (1 KiB per wave-instruction at the L1's 64 B/clk = 16 cycles) into 64 staging registers, ds_write_b128
// into a double-buffered LDS image (~13 cycles each), ds_read_b128 fragments.  Per K-tile and CU: 1024 cycles of L1 path,
// ~830 + 512 cycles of LDS writes + reads (the 128x128 wave tile reads a third fewer fragment bytes than 128x64), 2048 of MFMA.
// One workgroup barrier per K-tile: K-tile t+1 is loaded during the first half of K-tile t's MFMAs, written to the other LDS
// image during the second half, and the barrier sits in the middle of the last k-step, with MFMAs queued on both sides.
#define W4_IMG 65536      // one LDS image: A rows 0..255 (32 KiB) then W rows 0..255 (32 KiB), 128-B rows, XOR-swizzled chunks
template <int EK, int EXP = 0>      // EXP: timing experiments (wrong results): 1 no global loads, 2 no LDS writes, 4 no barrier, 8 no fragment reads
__global__ __launch_bounds__(256) void gemm_f16_w4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int gx = g.gx, gy = g.gy;
    const int lin = blockIdx.x;
    int tx, ty;
    if (gy >= 16) {          // XCD-aware order, as in the kernels above
        const int slot = lin >> 3;
        ty = (slot / gx) * 8 + (lin & 7);
        tx = slot - (slot / gx) * gx;
    } else {
        ty = lin / gx;
        tx = lin - ty * gx;
    }
    if (ty >= gy) return;
    const int m0 = ty * 256, n0 = tx * 256;

    // staging: wave w fetches rows [64w, 64w + 64) of the A and of the W tile, 8 rows x 128 B per wave-instruction
    const int r8 = lane >> 3, c8 = lane & 7;
    unsigned offA[8], offB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 64 + i * 8 + r8;
        const int ra = m0 + row < g.M ? m0 + row : g.M - 1, rb = n0 + row < g.N ? n0 + row : g.N - 1;
        offA[i] = (unsigned)(((long)ra * g.lda + c8 * 8) * 2);
        offB[i] = (unsigned)(((long)rb * g.ldw + c8 * 8) * 2);
    }
    // LDS write address of staging register i: row 64w + 8i + r8, physical chunk c8 ^ ((row >> 1) & 7); (row >> 1) & 7 =
    // (r8 >> 1) ^ 4 (i & 1): two bases, + (i >> 1) * 2048
    int wbase[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = wave * 64 + par * 8 + r8;
        wbase[par] = row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4);
    }
    // fragment reads: A rows wr*128 + mt*32 + l31 (+ mt * 4096 B), W rows wc*128 + nt*32 + l31 in the second half of the image
    const int hh = lane >> 5, l31 = lane & 31;
    int aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ra = wr * 128 + l31, rb = wc * 128 + l31;
        aaddr[ks] = ra * 128 + (((2 * ks + hh) ^ ((ra >> 1) & 7)) << 4);
        baddr[ks] = 32768 + rb * 128 + (((2 * ks + hh) ^ ((rb >> 1) & 7)) << 4);
    }
    f32x16 acc[2][2][2][2];      // [row block rb][column block cb][mi][ni]: MFMA tile (mt, nt) = (2 rb + mi, 2 cb + ni)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][i][j][r] = 0.f;

    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;
    f16x8 G[16];                 // staging registers: 0..7 A, 8..15 W
    f16x8 fa[2][4], fb[2][4];    // fragments of two k-steps
    // K-tile t_ -> uniform byte bases of its A / W columns
#define W4_BASES(t_)                                                                                          \
    const int seg__ = (t_) / ktiles;                                                                          \
    const long k0__ = (long)((t_) - seg__ * ktiles) * BK;                                                     \
    const char* PA__ = reinterpret_cast<const char*>((seg__ == 0 ? g.A[0] : (seg__ == 1 ? g.A[1] : g.A[2])) + k0__); \
    const char* PW__ = reinterpret_cast<const char*>((seg__ == 0 ? g.W[0] : (seg__ == 1 ? g.W[1] : g.W[2])) + k0__);
    // (the empty asm keeps the 32-bit offset a 32-bit register: uniform base + VGPR offset form of the load, no 64-bit vector add)
#define W4_LOAD_A(i_) { unsigned o__ = offA[i_]; asm volatile("" : "+v"(o__)); G[i_] = *reinterpret_cast<const f16x8*>(PA__ + o__); }
#define W4_LOAD_B(i_) { unsigned o__ = offB[i_]; asm volatile("" : "+v"(o__)); G[8 + (i_)] = *reinterpret_cast<const f16x8*>(PW__ + o__); }
#define W4_WRITE_A(i_) *reinterpret_cast<f16x8*>(smem + wbase[(i_) & 1] + ((i_) >> 1) * 2048) = G[i_];
#define W4_WRITE_B(i_) *reinterpret_cast<f16x8*>(smem + 32768 + wbase[(i_) & 1] + ((i_) >> 1) * 2048) = G[8 + (i_)];
#define W4_READ(buf_, ks_, j_)                                                                                \
    if ((j_) < 4) fa[buf_][(j_) & 3] = *reinterpret_cast<const f16x8*>(smem + aaddr[ks_] + ((j_) & 3) * 4096); \
    else fb[buf_][(j_) & 3] = *reinterpret_cast<const f16x8*>(smem + baddr[ks_] + ((j_) & 3) * 4096);
#define W4_SB __builtin_amdgcn_sched_barrier(0);
    // MFMA j of a k-step: tile (mt, nt) in snake order over the 4 x 4 tiles
#define W4_MMA(buf_, j_)                                                                                      \
    {                                                                                                         \
        constexpr int mt__ = (j_) >> 2, nt__ = (mt__ & 1) ? 3 - ((j_) & 3) : (j_) & 3;                        \
        acc[mt__ >> 1][nt__ >> 1][mt__ & 1][nt__ & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                \
            fa[buf_][mt__], fb[buf_][nt__], acc[mt__ >> 1][nt__ >> 1][mt__ & 1][nt__ & 1], 0, 0, 0);           \
    }
#define W4_REP16(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) X_(8) X_(9) X_(10) X_(11) X_(12) X_(13) X_(14) X_(15)

    // prologue: K-tile 0 -> image 0, K-tile 1 requested into the staging registers, fragments of k-step 0 of K-tile 0.  The
    // images alternate by toggling bit 16 of the addresses: wbase points into the image being WRITTEN (the next K-tile's),
    // aaddr / baddr into the one being read.
    {
        W4_BASES(0)
#define W4_P0(j_) if ((j_) < 8) { W4_LOAD_A((j_) & 7) } else { W4_LOAD_B((j_) & 7) }
        W4_REP16(W4_P0)
#define W4_P1(j_) if ((j_) < 8) { W4_WRITE_A((j_) & 7) } else { W4_WRITE_B((j_) & 7) }
        W4_REP16(W4_P1)
#undef W4_P1
    }
    if (nt > 1) {
        W4_BASES(1)
        W4_REP16(W4_P0)
    }
#undef W4_P0
    wbase[0] ^= W4_IMG; wbase[1] ^= W4_IMG;
    __syncthreads();
#define W4_P2(j_) if ((j_) < 8) { W4_READ(0, 0, (j_) & 7) }
    W4_REP16(W4_P2)
#undef W4_P2

    // One K-tile t.  WR: K-tile t+1 exists (its rows wait in the staging registers: write them to the other image, barrier,
    // pre-read its k-step 0); LD: K-tile t+2 exists (a staging register is re-loaded with its K-tile t+2 data two slots after
    // it has been written out, so a load has a whole K-tile of MFMAs, ~1.4 us, to arrive).
    // ONE memory instruction per MFMA slot, the three kinds spread evenly over the K-tile: the four waves run in lockstep
    // (barrier), so 8 loads (writes) in 8 consecutive slots of every wave ask the CU's L1 (LDS store path) for 128 (~100) B/clk
    // against 64 (79) it delivers, and the waves stall at ISSUE, matrix pipe idle (tools/gemm_w4_exp.py: bunched, the loads cost
    // 0.29 us and the writes 0.13 us of a 1.9-us K-tile).
    //   slot j of k-steps 0..2: j even: fragment j/2 of the next k-step (order: A0 B0 B1 B2 B3 A1 A2 A3, the order of first use)
    //                           j = 4q+1: write staging register 4s+q;  j = 4q+3: load it again
    //   k-step 3: j < 8: write (even) / load (odd) staging register 12 + j/2; barrier after MFMA 7; j >= 8: fragments of
    //             k-step 0 of K-tile t+1
#define W4_FR(r_) ((r_) == 0 ? 0 : (r_) <= 4 ? (r_) + 3 : (r_) - 4)      /* read order -> fragment index (0..3 A, 4..7 B) */
#define W4_WRITE(gi_) if ((gi_) < 8) { W4_WRITE_A((gi_) & 7) } else { W4_WRITE_B((gi_) & 7) }
#define W4_LOAD(gi_) if ((gi_) < 8) { W4_LOAD_A((gi_) & 7) } else { W4_LOAD_B((gi_) & 7) }
#define W4_SLOT(s_, buf_, j_)                                                                                 \
    if (((j_) & 1) == 0) { if (!(EXP & 8)) { W4_READ((buf_) ^ 1, (s_) + 1, W4_FR((j_) >> 1)) } }              \
    else if (((j_) & 3) == 1) { if (WR && !(EXP & 2)) { W4_WRITE((s_) * 4 + ((j_) >> 2)) } }                  \
    else { if (LD && !(EXP & 1)) { W4_LOAD((s_) * 4 + ((j_) >> 2)) } }                                        \
    W4_SB W4_MMA(buf_, j_) W4_SB
#define W4_S0(j_) W4_SLOT(0, 0, j_)
#define W4_S1(j_) W4_SLOT(1, 1, j_)
#define W4_S2(j_) W4_SLOT(2, 0, j_)
#define W4_S3(j_)                                                                                             \
    if ((j_) < 8) {                                                                                           \
        if (((j_) & 1) == 0) { if (WR && !(EXP & 2)) { W4_WRITE(12 + (((j_) & 7) >> 1)) } }                   \
        else { if (LD && !(EXP & 1)) { W4_LOAD(12 + (((j_) & 7) >> 1)) } }                                    \
    } else if (WR && !(EXP & 8)) { W4_READ(0, 0, W4_FR((j_) & 7)) }                                           \
    W4_SB W4_MMA(1, j_) W4_SB                                                                                 \
    if (WR && (j_) == 7 && !(EXP & 4)) { __syncthreads(); W4_SB }
#define W4_TILE(t_)                                                                                           \
    {                                                                                                         \
        W4_BASES((t_) + 2)                                                                                    \
        W4_SB                                                                                                 \
        W4_REP16(W4_S0) W4_REP16(W4_S1) W4_REP16(W4_S2)                                                       \
        if (WR) {      /* every read of this image has been issued: the read addresses move to the other one */ \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { aaddr[ks] ^= W4_IMG; baddr[ks] ^= W4_IMG; }    \
        }                                                                                                     \
        W4_REP16(W4_S3)                                                                                       \
        wbase[0] ^= W4_IMG; wbase[1] ^= W4_IMG;                                                               \
    }
    int t = 0;
    for (; t + 2 < nt; ++t) { constexpr bool WR = true, LD = true; W4_TILE(t) }
    if (t + 1 < nt) { constexpr bool WR = true, LD = false; W4_TILE(t) ++t; }
    { constexpr bool WR = false, LD = false; W4_TILE(t) }
#undef W4_TILE
#undef W4_SLOT
#undef W4_WRITE
#undef W4_LOAD
#undef W4_FR
#undef W4_S0
#undef W4_S1
#undef W4_S2
#undef W4_S3
#undef W4_REP16
#undef W4_MMA
#undef W4_READ
#undef W4_WRITE_A
#undef W4_WRITE_B
#undef W4_LOAD_A
#undef W4_LOAD_B
#undef W4_BASES
#undef W4_SB
    __syncthreads();                                 // every wave is done with the operand images: epilogue scratch
    float bv0[2], sc0[2], bv1[2], sc1[2];
    gemm_colvals(g, n0, wc * 2, lane, 0, bv0, sc0);
    gemm_colvals(g, n0, wc * 2 + 1, lane, 0, bv1, sc1);
    gemm_epilogue<EK, 2, false>(g, acc[0][0], m0 + wr * 128, n0, 0, wc * 2, lane, 0, smem + wave * 8192, bv0, sc0, 0);
    gemm_epilogue<EK, 2, false>(g, acc[0][1], m0 + wr * 128, n0, 0, wc * 2 + 1, lane, 0, smem + wave * 8192, bv1, sc1, 0);
    gemm_epilogue<EK, 2, false>(g, acc[1][0], m0 + wr * 128 + 64, n0, 0, wc * 2, lane, 0, smem + wave * 8192, bv0, sc0, 0);
    gemm_epilogue<EK, 2, false>(g, acc[1][1], m0 + wr * 128 + 64, n0, 0, wc * 2 + 1, lane, 0, smem + wave * 8192, bv1, sc1, 0);
}

// ---------------------------------------------------------------------------------------------
// 256x256x64, four waves, a K-tile's fragments RESIDENT IN REGISTERS (round 3, experiment; the schedule the vendor library's
// hand-written kernel uses): wave (wr, wc) owns 128x128 (256 accumulator registers), operands come in by
// LDS-DMA into two LDS images, and a wave reads a WHOLE K-tile of its fragments (32 x 16 B = 128 registers) early:
//   iteration t (image t & 1):
//     phase A  MFMAs of k-step 0 | fragment reads of k-steps 2, 3 (k-steps 0, 1 were read in the previous iteration)
//              barrier: every wave holds all of K-tile t in registers -> image t & 1 is FREE although 3/4 of the MFMAs remain
//     phase B  MFMAs of k-step 1 | the 16 DMA instructions of K-tile t + 2 into that image
//     phase C  MFMAs of k-step 2
//     phase D  wait until K-tile t + 1 has landed (vmcnt(16): K-tile t + 2 stays in flight), barrier,
//              MFMAs of k-step 3 | fragment reads of k-steps 0, 1 of K-tile t + 1 from the other image
// so a DMA has ~1.6 K-tiles to land with only two images.
// BUF: the DMA as `buffer_load_dwordx4 v, s[rsrc], s_off offen lds` (row-group offsets in SGPRs, two lane-offset registers for all
// sixteen pieces, rows past the matrix read as zeros by the buffer's range check) instead of global_load_lds with a lane offset each.
template <int EK, bool BUF = false>
__global__ __launch_bounds__(256) void gemm_f16_r4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int gx = g.gx, gy = g.gy;
    const int lin = blockIdx.x;
    int tx, ty;
    if (gy >= 16) {          // XCD-aware order, as in the kernels above
        const int slot = lin >> 3;
        ty = (slot / gx) * 8 + (lin & 7);
        tx = slot - (slot / gx) * gx;
    } else {
        ty = lin / gx;
        tx = lin - ty * gx;
    }
    if (ty >= gy) return;
    const int m0 = ty * 256, n0 = tx * 256;

    // DMA sources: wave w stages rows [64w, 64w + 64) of the A and of the W tile, 8 rows x 128 B per instruction; the XOR
    // swizzle of the LDS image is applied on the source side (lane -> logical chunk), the destination is linear
    const int r8 = lane >> 3;
    unsigned offA[8], offB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 64 + i * 8 + r8;
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int ra = m0 + row < g.M ? m0 + row : g.M - 1, rb = n0 + row < g.N ? n0 + row : g.N - 1;
        offA[i] = (unsigned)(((long)ra * g.lda + c * 8) * 2);
        offB[i] = (unsigned)(((long)rb * g.ldw + c * 8) * 2);
    }
    // BUF: lane offsets inside an 8-row group (two variants: the swizzle of row 64w + 8i + r8 depends on the parity of i),
    // scalar offsets of the row groups
    int vA[2], vB[2], sA[8], sB[8];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int c = (lane & 7) ^ (((r8 >> 1) ^ (par * 4)) & 7);
        vA[par] = (int)(((long)r8 * g.lda + c * 8) * 2);
        vB[par] = (int)(((long)r8 * g.ldw + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sA[i] = (int)(((long)(m0 + wave * 64 + i * 8) * g.lda) * 2);
        sB[i] = (int)(((long)(n0 + wave * 64 + i * 8) * g.ldw) * 2);
    }
    const int hh = lane >> 5, l31 = lane & 31;
    int aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ra = wr * 128 + l31, rb = wc * 128 + l31;
        aaddr[ks] = ra * 128 + (((2 * ks + hh) ^ ((ra >> 1) & 7)) << 4);
        baddr[ks] = 32768 + rb * 128 + (((2 * ks + hh) ^ ((rb >> 1) & 7)) << 4);
    }
    f32x16 acc[2][2][2][2];      // [row block rb][column block cb][mi][ni]: MFMA tile (mt, nt) = (2 rb + mi, 2 cb + ni)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][i][j][r] = 0.f;

    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    f16x8 fa[4][4], fb[4][4];      // [k-step][row / column tile]: one whole K-tile
    int wimg = 0;                  // byte offset of the image the next DMAs go to
    const long bytesA = (long)g.M * g.lda * 2, bytesW = (long)g.N * g.ldw * 2;      // (< 4 GiB: checked by the launcher)
#define R4_BASES(t_)                                                                                          \
    const int seg__ = (t_) / ktiles;                                                                          \
    const long k0__ = (long)((t_) - seg__ * ktiles) * BK;                                                     \
    const char* PA__ = reinterpret_cast<const char*>((seg__ == 0 ? g.A[0] : (seg__ == 1 ? g.A[1] : g.A[2])) + k0__); \
    const char* PW__ = reinterpret_cast<const char*>((seg__ == 0 ? g.W[0] : (seg__ == 1 ? g.W[1] : g.W[2])) + k0__); \
    __amdgpu_buffer_rsrc_t RA__ = __builtin_amdgcn_make_buffer_rsrc((void*)PA__, 0, BUF ? (int)(bytesA - k0__ * 2) : 0, 0x00020000); \
    __amdgpu_buffer_rsrc_t RW__ = __builtin_amdgcn_make_buffer_rsrc((void*)PW__, 0, BUF ? (int)(bytesW - k0__ * 2) : 0, 0x00020000);
    // DMA piece i (0..7 A, 8..15 W) of the K-tile whose bases are in scope
#define R4_DMA(i_)                                                                                            \
    {                                                                                                         \
        char* dst__ = smem + wimg + ((i_) < 8 ? 0 : 32768) + (wave * 64 + ((i_) & 7) * 8) * 128;              \
        if constexpr (BUF) {                                                                                  \
            if ((i_) < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(RA__, (lds_ptr)dst__, 16, vA[(i_) & 1], sA[(i_) & 7], 0, 0); \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(RW__, (lds_ptr)dst__, 16, vB[(i_) & 1], sB[(i_) & 7], 0, 0);          \
        } else {                                                                                              \
            unsigned o__ = (i_) < 8 ? offA[(i_) & 7] : offB[(i_) & 7];                                        \
            asm volatile("" : "+v"(o__));                                                                     \
            const char* src__ = ((i_) < 8 ? PA__ : PW__) + o__;                                               \
            __builtin_amdgcn_global_load_lds((gbl_ptr)src__, (lds_ptr)dst__, 16, 0, 0);                       \
        }                                                                                                     \
    }
    // fragment j (0..3 A row tiles, 4..7 W column tiles) of k-step ks_
#define R4_READ(ks_, j_)                                                                                      \
    if ((j_) < 4) fa[ks_][(j_) & 3] = *reinterpret_cast<const f16x8*>(smem + aaddr[ks_] + ((j_) & 3) * 4096);   \
    else fb[ks_][(j_) & 3] = *reinterpret_cast<const f16x8*>(smem + baddr[ks_] + ((j_) & 3) * 4096);
#define R4_SB __builtin_amdgcn_sched_barrier(0);
#define R4_MMA(ks_, j_)                                                                                       \
    {                                                                                                         \
        constexpr int mt__ = (j_) >> 2, nt__ = (mt__ & 1) ? 3 - ((j_) & 3) : (j_) & 3;                        \
        acc[mt__ >> 1][nt__ >> 1][mt__ & 1][nt__ & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                \
            fa[ks_][mt__], fb[ks_][nt__], acc[mt__ >> 1][nt__ >> 1][mt__ & 1][nt__ & 1], 0, 0, 0);             \
    }
#define R4_REP16(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) X_(8) X_(9) X_(10) X_(11) X_(12) X_(13) X_(14) X_(15)

    // prologue: K-tiles 0 and 1 requested, K-tile 0 landed, its k-steps 0, 1 in registers
    {
        R4_BASES(0)
#define R4_P(j_) R4_DMA(j_)
        R4_REP16(R4_P)
    }
    wimg ^= W4_IMG;
    if (nt > 1) {
        R4_BASES(1)
        R4_REP16(R4_P)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef R4_P
    wimg ^= W4_IMG;
    __builtin_amdgcn_s_barrier();
#define R4_P(j_) if ((j_) < 8) { R4_READ(0, (j_) & 7) } else { R4_READ(1, (j_) & 7) }
    R4_REP16(R4_P)
#undef R4_P

    // read order inside a phase: the fragments the next MFMAs need first (A0 B0 B1 B2 B3 A1 A2 A3 of a k-step)
#define R4_FR(r_) ((r_) == 0 ? 0 : (r_) <= 4 ? (r_) + 3 : (r_) - 4)
#define R4_A(j_) { R4_READ(2 + ((j_) >> 3), R4_FR((j_) & 7)) } R4_SB R4_MMA(0, j_) R4_SB
#define R4_B(j_) if (LD) { R4_DMA(j_) } R4_SB R4_MMA(1, j_) R4_SB
#define R4_C(j_) R4_SB R4_MMA(2, j_) R4_SB
#define R4_D(j_) if (WR) { R4_READ((j_) >> 3, R4_FR((j_) & 7)) } R4_SB R4_MMA(3, j_) R4_SB
#define R4_ITER(t_)                                                                                           \
    {                                                                                                         \
        R4_BASES((t_) + 2)                                                                                    \
        R4_SB                                                                                                 \
        R4_REP16(R4_A)                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_s_barrier();      /* the image of K-tile t is free */                                \
        R4_SB                                                                                                 \
        R4_REP16(R4_B)                                                                                        \
        R4_REP16(R4_C)                                                                                        \
        if (WR) {                                                                                             \
            if (LD) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                         \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
            __builtin_amdgcn_s_barrier();  /* K-tile t + 1 is in the other image */                           \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { aaddr[ks] ^= W4_IMG; baddr[ks] ^= W4_IMG; }    \
        }                                                                                                     \
        R4_SB                                                                                                 \
        R4_REP16(R4_D)                                                                                        \
        wimg ^= W4_IMG;                                                                                       \
    }
    int t = 0;
    for (; t + 2 < nt; ++t) { constexpr bool WR = true, LD = true; R4_ITER(t) }
    if (t + 1 < nt) { constexpr bool WR = true, LD = false; R4_ITER(t) ++t; }
    { constexpr bool WR = false, LD = false; R4_ITER(t) }
#undef R4_ITER
#undef R4_A
#undef R4_B
#undef R4_C
#undef R4_D
#undef R4_FR
#undef R4_REP16
#undef R4_MMA
#undef R4_SB
#undef R4_READ
#undef R4_DMA
#undef R4_BASES
    __syncthreads();                                 // every wave is done with the operand images: epilogue scratch
    float bv0[2], sc0[2], bv1[2], sc1[2];
    gemm_colvals(g, n0, wc * 2, lane, 0, bv0, sc0);
    gemm_colvals(g, n0, wc * 2 + 1, lane, 0, bv1, sc1);
    gemm_epilogue<EK, 2, false>(g, acc[0][0], m0 + wr * 128, n0, 0, wc * 2, lane, 0, smem + wave * 8192, bv0, sc0, 0);
    gemm_epilogue<EK, 2, false>(g, acc[0][1], m0 + wr * 128, n0, 0, wc * 2 + 1, lane, 0, smem + wave * 8192, bv1, sc1, 0);
    gemm_epilogue<EK, 2, false>(g, acc[1][0], m0 + wr * 128 + 64, n0, 0, wc * 2, lane, 0, smem + wave * 8192, bv0, sc0, 0);
    gemm_epilogue<EK, 2, false>(g, acc[1][1], m0 + wr * 128 + 64, n0, 0, wc * 2 + 1, lane, 0, smem + wave * 8192, bv1, sc1, 0);
}

// ---------------------------------------------------------------------------------------------
// 256x192x64 variant of the ping-pong kernel (round 3): for N = 768 / 2304 a 192-column tile gives 4 / 12 tile columns, so
// 16 images x 1025 tokens (64 full 256-row tiles) make 256 / 768 tiles = EXACTLY 1 / 3 rounds of the 256 CUs, where the
// 256x256 tile leaves a quarter of the chip idle (192 tiles = 0.75 round, 576 = 2.25 rounds).
//
// 8 waves as 4 (rows) x 2 (columns), wave tile 64 x 96 = 2 x 3 MFMA tiles (96 accumulator registers).  A K-tile is staged as
// SEVEN 8-KiB units (one LDS-DMA instruction per wave and unit), in the order of use:
//   u0, u1 = A rows mi = 0 of the wave rows {0,1} / {2,3};  u2, u3, u4 = B column tiles j = 0, 1, 2 (both column groups);
//   u5, u6 = A rows mi = 1
// into a ring of 14 slots (two K-tiles, 112 KiB), and computed in three phases of 8 MFMAs in snake order:
//   P0 (mi0,j0) (mi0,j1)   P1 (mi0,j2) (mi1,j2)   P2 (mi1,j1) (mi1,j0)
// with the fragment reads spread 8 / 8 / 4: P0 reads B0, B1; P1 reads B2, A(mi1); P2 pre-reads A(mi0) of the NEXT K-tile into
// the registers P1 has just finished with.  As in the 256x256 kernel the two wave halves (waves w and w + 4 share a SIMD) run
// one barrier apart, so one half's MFMAs overlap the other half's LDS reads, and the DMA instructions are issued between the
// MFMAs of a phase: P0(t) requests units 5,6 of K-tile t+1, P1(t) units 0,1 of t+2, P2(t) units 2,3,4 of t+2 -- every unit is
// requested 3-4 phases (1 - 1.3 K-tiles) before the counted wait that needs it, 5 units (40 KiB) stay in flight across the
// barriers (`s_waitcnt vmcnt(5)` at every wait), and a slot is re-staged only after both halves have passed the
// `lgkmcnt(0)` behind its last fragment read.
#define P192_UNIT 8192
template <int EK>
__global__ __launch_bounds__(512) void gemm_f16_p192_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 14 units [64 rows][64 halfs], XOR-swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, grp = wave >> 2;      // grp: the half that runs one barrier behind (1)
    const int gx = g.gx, gy = g.gy;
    const int lin = blockIdx.x;
    int tx, ty;
    if (gy >= 16) {          // XCD-aware order, as in the other kernels
        const int slot = lin >> 3;
        ty = (slot / gx) * 8 + (lin & 7);
        tx = slot - (slot / gx) * gx;
    } else {
        ty = lin / gx;
        tx = lin - ty * gx;
    }
    if (ty >= gy) return;
    const int m0 = ty * 256, n0 = tx * 192;

    // per-thread DMA source of the seven unit kinds: chunk q = wave * 64 + lane -> unit row q >> 3, physical chunk q & 7
    long offA[2][2], offB[3];
    {
        const int row = tid >> 3;
        const int c = (tid & 7) ^ ((row >> 1) & 7);      // logical chunk this lane must fetch (swizzle on the source)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int a = m0 + (2 * h + (row >> 5)) * 64 + mi * 32 + (row & 31);
                a = a < g.M ? a : g.M - 1;
                offA[mi][h] = (long)a * g.lda + c * 8;
            }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int b = n0 + (row >> 5) * 96 + j * 32 + (row & 31);
            b = b < g.N ? b : g.N - 1;
            offB[j] = (long)b * g.ldw + c * 8;
        }
    }
    const int ktiles = g.K / BK;
    const int nt = ktiles * g.nseg;                  // even and >= 4 (the launcher checks)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    // unit u_ (compile time) of K-tile t_ into ring half par_ (compile time)
#define P192_STAGE(t_, u_, par_)                                                                              \
    {                                                                                                         \
        const int seg__ = (t_) / ktiles;                                                                      \
        const long k0__ = (long)((t_) - seg__ * ktiles) * BK;                                                 \
        const __half* P__ = ((u_) < 2 || (u_) > 4) ? (seg__ == 0 ? g.A[0] : (seg__ == 1 ? g.A[1] : g.A[2]))   \
                                                   : (seg__ == 0 ? g.W[0] : (seg__ == 1 ? g.W[1] : g.W[2]));  \
        const long o__ = (u_) == 0 ? offA[0][0] : (u_) == 1 ? offA[0][1] : (u_) == 2 ? offB[0] : (u_) == 3 ? offB[1] \
                       : (u_) == 4 ? offB[2] : (u_) == 5 ? offA[1][0] : offA[1][1];                           \
        char* d__ = smem + ((par_) * 7 + (u_)) * P192_UNIT + wave * 1024;                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_global_load_lds((gbl_ptr)(P__ + o__ + k0__), (lds_ptr)d__, 16, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // fragment read addresses (bytes inside the ring half): A unit row (wr & 1) * 32 + l31 of unit (mi ? 5 : 0) + (wr >> 1),
    // B unit row wc * 32 + l31 of unit 2 + j
    const int hh = lane >> 5, l31 = lane & 31;
    int aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ra = (wr & 1) * 32 + l31, rb = wc * 32 + l31;
        aaddr[ks] = grp * P192_UNIT + ra * 128 + (((2 * ks + hh) ^ ((ra >> 1) & 7)) << 4);
        baddr[ks] = rb * 128 + (((2 * ks + hh) ^ ((rb >> 1) & 7)) << 4);
    }
    f32x16 accP[2][2], accQ[2][1];      // [mi][j = 0, 1], [mi][j = 2]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accQ[i][0][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accP[i][j][r] = 0.f;
    }
    float bvP[2], scP[2], bvQ[2], scQ[2];
    gemm_colvals(g, n0 + wc * 96, 0, lane, 0, bvP, scP);
    gemm_colvals(g, n0 + wc * 96 + 64, 0, lane, 0, bvQ, scQ);

    f16x8 fa0[4], fa1[4], fb0[4], fb1[4], fb2[4];
#define P192_RD_A(f_, par_, mi_)                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                          \
        f_[ks] = *reinterpret_cast<const f16x8*>(smem + ((par_) * 7 + ((mi_) ? 5 : 0)) * P192_UNIT + aaddr[ks]);
#define P192_RD_B(f_, par_, j_)                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                          \
        f_[ks] = *reinterpret_cast<const f16x8*>(smem + ((par_) * 7 + 2 + (j_)) * P192_UNIT + baddr[ks]);
    // one phase: [fragment reads] -> counted wait -> barrier -> 8 MFMAs (two accumulators) with the DMA instructions ST0_..ST2_
    // after the 1st / 2nd / 3rd MFMA pair -> barrier
#define P192_PHASE(READS_, WAIT_, ACC0_, FA0_, FB0_, ACC1_, FA1_, FB1_, ST0_, ST1_, ST2_)                     \
    {                                                                                                         \
        READS_;                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        WAIT_;                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                    \
            ACC0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA0_[ks], FB0_[ks], ACC0_, 0, 0, 0);               \
            ACC1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA1_[ks], FB1_[ks], ACC1_, 0, 0, 0);               \
            if (ks == 0) { ST0_; }                                                                            \
            if (ks == 1) { ST1_; }                                                                            \
            if (ks == 2) { ST2_; }                                                                            \
        }                                                                                                     \
        asm volatile("" : "+v"(ACC0_), "+v"(ACC1_));      /* keeps the MFMAs inside their phase */            \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_barrier();                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define P192_W5 asm volatile("s_waitcnt vmcnt(5)" ::: "memory")
#define P192_W3 asm volatile("s_waitcnt vmcnt(3)" ::: "memory")
#define P192_W0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    // K-tile t_ in ring half par_.  KIND_ 0: steady state (t + 2 < nt); 1: t = nt - 2; 2: t = nt - 1.
#define P192_TILE(t_, par_, KIND_)                                                                            \
    {                                                                                                         \
        if ((KIND_) == 0) {                                                                                   \
            P192_PHASE(P192_RD_B(fb0, par_, 0) P192_RD_B(fb1, par_, 1), P192_W5, accP[0][0], fa0, fb0, accP[0][1], fa0, fb1, \
                       P192_STAGE((t_) + 1, 5, (par_) ^ 1), , P192_STAGE((t_) + 1, 6, (par_) ^ 1));           \
            P192_PHASE(P192_RD_B(fb2, par_, 2) P192_RD_A(fa1, par_, 1), P192_W5, accQ[0][0], fa0, fb2, accQ[1][0], fa1, fb2, \
                       P192_STAGE((t_) + 2, 0, par_), , P192_STAGE((t_) + 2, 1, par_));                       \
            P192_PHASE(P192_RD_A(fa0, (par_) ^ 1, 0), P192_W5, accP[1][1], fa1, fb1, accP[1][0], fa1, fb0,    \
                       P192_STAGE((t_) + 2, 2, par_), P192_STAGE((t_) + 2, 3, par_), P192_STAGE((t_) + 2, 4, par_)); \
        } else if ((KIND_) == 1) {                                                                            \
            P192_PHASE(P192_RD_B(fb0, par_, 0) P192_RD_B(fb1, par_, 1), P192_W5, accP[0][0], fa0, fb0, accP[0][1], fa0, fb1, \
                       P192_STAGE((t_) + 1, 5, (par_) ^ 1), , P192_STAGE((t_) + 1, 6, (par_) ^ 1));           \
            P192_PHASE(P192_RD_B(fb2, par_, 2) P192_RD_A(fa1, par_, 1), P192_W5, accQ[0][0], fa0, fb2, accQ[1][0], fa1, fb2, , , ); \
            P192_PHASE(P192_RD_A(fa0, (par_) ^ 1, 0), P192_W3, accP[1][1], fa1, fb1, accP[1][0], fa1, fb0, , , ); \
        } else {                                                                                              \
            P192_PHASE(P192_RD_B(fb0, par_, 0) P192_RD_B(fb1, par_, 1), P192_W0, accP[0][0], fa0, fb0, accP[0][1], fa0, fb1, , , ); \
            P192_PHASE(P192_RD_B(fb2, par_, 2) P192_RD_A(fa1, par_, 1), , accQ[0][0], fa0, fb2, accQ[1][0], fa1, fb2, , , ); \
            P192_PHASE(, , accP[1][1], fa1, fb1, accP[1][0], fa1, fb0, , , );                                  \
        }                                                                                                     \
    }
    // prologue: K-tile 0 complete (slots 0..6) and units 0..4 of K-tile 1 (slots 7..11): what P1(-1) and P2(-1) would have
    // requested; K-tile 0's units 0..3 landed (8 newer requests may stay in flight)
    P192_STAGE(0, 0, 0); P192_STAGE(0, 1, 0); P192_STAGE(0, 2, 0); P192_STAGE(0, 3, 0); P192_STAGE(0, 4, 0); P192_STAGE(0, 5, 0);
    P192_STAGE(0, 6, 0); P192_STAGE(1, 0, 1); P192_STAGE(1, 1, 1); P192_STAGE(1, 2, 1); P192_STAGE(1, 3, 1); P192_STAGE(1, 4, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    P192_RD_A(fa0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();      // the second half runs one barrier behind the first

    int t = 0;
    for (; t + 2 < nt; t += 2) {
        P192_TILE(t, 0, 0);
        P192_TILE(t + 1, 1, 0);
    }
    P192_TILE(t, 0, 1);
    P192_TILE(t + 1, 1, 2);
#undef P192_TILE
#undef P192_PHASE
#undef P192_RD_A
#undef P192_RD_B
#undef P192_STAGE
#undef P192_W5
#undef P192_W3
#undef P192_W0
    if (grp == 0) __builtin_amdgcn_s_barrier();      // re-align the two halves
    __syncthreads();                                 // every wave is done with the ring: epilogue scratch
    gemm_epilogue<EK, 2>(g, accP, m0 + wr * 64, n0 + wc * 96, 0, 0, lane, 0, smem + wave * 8192, bvP, scP, 0);
    gemm_epilogue<EK, 1>(g, accQ, m0 + wr * 64, n0 + wc * 96 + 64, 0, 0, lane, 0, smem + wave * 8192, bvQ, scQ, 0);
}

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
    int gx;
    int tiles, ns, units; // output tiles per (group, slice) unit, slices per group, units = groups * ns
    int xcd;              // XCD-aware workgroup order (units % 8 == 0)
    long gA, gX, gP;      // group strides of dY, X and the partials, in elements
    GemmArgs e;           // epilogue: M = N, N = K + bias, C32 = partials, ldc, sC
};

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));

__global__ __launch_bounds__(256, 2) void gemm_km_kernel(KmArgs g) {
    // ring of 4 stages x [32 tokens]: [A tile 8 KiB | X tile 8 KiB]; three stages in flight while one is consumed,
    // counted vmcnt waits + raw s_barrier (a __syncthreads() drains every outstanding LDS-DMA: with 2 stages of 64
    // tokens the loop ran at one global-memory latency per stage)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 32 * 256;
    constexpr int ST = 32;                // tokens per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
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
    bool aok[2], xok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 256 + tid;
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
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
#define KM_LOAD(t_, buf_)                                                                                       \
    {                                                                                                            \
        char* dst_ = smem + (buf_) * (2 * TILE) + wave * 1024;                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int m_ = mbeg + (t_) * ST + rowi[i];                                                           \
            const bool in_ = m_ < mend;                                                                          \
            const __half* pa_ = (in_ && aok[i]) ? Ag + (long)m_ * g.lda + acol[i] : g.zeros;                     \
            const __half* px_ = (in_ && xok[i]) ? Xg + ((long)xg[i] * g.x_gs + xr[i] + g.x_off) * g.ldx + xcol[i] : g.zeros; \
            __builtin_amdgcn_global_load_lds((gbl_ptr)pa_, (lds_ptr)(dst_ + i * 4096), 16, 0, 0);                \
            __builtin_amdgcn_global_load_lds((gbl_ptr)px_, (lds_ptr)(dst_ + TILE + i * 4096), 16, 0, 0);         \
            xr[i] += ST;                                                                                         \
            while (xr[i] >= g.x_rpg) { xr[i] -= g.x_rpg; xg[i] += 1; }                                           \
        }                                                                                                        \
    }
    // transposing fragment reads: lane = 16*grp + 4*q + p; grp = 2*hh + gi
    const int hh = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    int aaddr[2][2], baddr[2][2];      // [mi / ni][r]: byte offset inside an operand tile for k-step 0
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * hh + 4 * r + q;                       // + 16 * ks
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
    const bool own_bias = g.bias && kb >= 0 && kb < 64;
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;

    // KM_LOAD advances the token -> X row map, so the stages must be requested in order: 0, 1, 2, then t + 3 in the loop
    if (nt > 0) KM_LOAD(0, 0);
    if (nt > 1) KM_LOAD(1, 1);
    if (nt > 2) KM_LOAD(2, 2);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 3;
        // stage t has landed once at most the requests of stages t+1, t+2 (4 DMA instructions each) are outstanding
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // every wave's part of stage t is in LDS; stage t-1's buffer is free
        if (t + 3 < nt) KM_LOAD(t + 3, (t + 3) & 3);
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
    if (own_bias) {      // every column of bacc holds the row sums: drop them into output column K
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
    gemm_epilogue<false>(g.e, acc, n0, k0, wr, wc, lane, z, smem + wave * 8192, bv, sc, (long)z * g.e.sC + (long)grp * g.gP);
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
    const int ns = wc_cdiv(M, mslice);
    WC_CHECK_ARG(ns <= 65535, "wc_gemm_km_f16: too many slices");
    KmArgs g;
    g.A = (const __half*)dY; g.X = (const __half*)X; g.zeros = (const __half*)zeros;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldx = ldx;
    g.x_rpg = x_rpg; g.x_gs = x_gs; g.x_off = x_off; g.mslice = mslice; g.bias = bias ? 1 : 0;
    const int K1 = K + g.bias;
    g.gx = wc_cdiv(K1, 128);
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
    hipLaunchKernelGGL(gemm_km_kernel, grid, dim3(256), 4 * 64 * 256, (hipStream_t)stream, g);
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
};
__global__ __launch_bounds__(256) void sum_slices_wb_multi_kernel(SumJobs j) {
    const int q = blockIdx.y;
    const float* __restrict__ part = j.part[q];
    float* __restrict__ out_w = j.out_w[q];
    float* __restrict__ out_b = j.out_b[q];
    const int nslices = j.nslices[q], cols = j.cols[q];
    const float alpha = j.alpha[q];
    const long n = (long)j.rows[q] * (cols + 1);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
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
}

// jobs: count x 7 host int64 {part, out_w, out_b (device pointers), nslices, rows, cols, alpha as float bits}
extern "C" int wc_sum_slices_wb_multi(const int64_t* jobs, int count, void* stream) {
    WC_CHECK_ARG(jobs && count > 0, "wc_sum_slices_wb_multi: bad argument");
    for (int base = 0; base < count; base += SUMJ_MAX) {
        SumJobs j;
        const int m = count - base < SUMJ_MAX ? count - base : SUMJ_MAX;
        long nmax = 0;
        for (int q = 0; q < SUMJ_MAX; ++q) {
            const int64_t* e = jobs + (long)(base + (q < m ? q : 0)) * 7;
            j.part[q] = reinterpret_cast<const float*>(e[0]);
            j.out_w[q] = reinterpret_cast<float*>(e[1]);
            j.out_b[q] = reinterpret_cast<float*>(e[2]);
            j.nslices[q] = (int)e[3]; j.rows[q] = (int)e[4]; j.cols[q] = (int)e[5];
            const unsigned bits = (unsigned)e[6];
            memcpy(&j.alpha[q], &bits, 4);
            WC_CHECK_ARG(j.part[q] && j.out_w[q] && j.out_b[q] && j.nslices[q] > 0 && j.rows[q] > 0 && j.cols[q] > 0,
                         "wc_sum_slices_wb_multi: bad job");
            const long n = (long)j.rows[q] * (j.cols[q] + 1);
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

// Which kernel a shape takes: 0 = 128x128 kernel, 1 = 256x256 ping-pong kernel, 2 = ping-pong kernel on the
// full 256-row tiles + the 128x128 / few-rows kernel on the ragged last M % 256 rows (see wc_gemm_f16), 3 / 4 = the same
// two with the 256x192 tile.
// Tile width (round 3, measured: profiles/r03_gemm_tiles.txt): the 192-column tile was built to remove the tile
// quantisation of N = 768 / 2304 (256 / 768 tiles = whole rounds of the 256 CUs instead of 0.75 / 2.25 rounds) and is
// bit-identical to the 256-column one, but on the encoder shapes it is SLOWER: proj 37.7 -> 45.0 us, fc2 91.4 -> 109.7,
// QKV 79.7 -> 81.1, fc1 106.0 -> 114.4.  The launch time follows the bytes staged into LDS (tiles x (256 + tile width) rows
// per K-tile: +17 % for the narrower tile), not the number of busy CUs: the K loop is paced by the operand delivery
// (per CU the LDS-DMA path, 1.26 us per K-tile with 24 CUs active; contention between CUs adds to it: 1.66 us at 192 CUs),
// so filling the idle quarter of the chip only raises the contention.  The 192-column tile is therefore taken only where it
// stages FEWER bytes (N = 192, 384, 576: the 256-column tile would carry dead columns), never in the training step.
static int g_r4 = 0;                     // 256x256 tiles on the 4-wave register-resident-fragments kernel (wc_gemm_set_r4; experiment)
extern "C" void wc_gemm_set_r4(int on) { g_r4 = on; }      // 1: global_load_lds, 2: buffer_load ... lds
static int g_w4 = -1;                    // 256x256 tiles on the 4-wave register-staged kernel (WECLIP_GEMM_W4 / wc_gemm_set_w4)
extern "C" void wc_gemm_set_w4(int on) { g_w4 = on; }
static int g_pp_ring10 = -1;             // 256x256 kernel with ten half-tile slots (WECLIP_GEMM_RING10 / wc_gemm_set_ring10)
extern "C" void wc_gemm_set_ring10(int on) { g_pp_ring10 = on ? 1 : 0; }
static int g_pp_m16 = -1;                // 256x256 kernel on 16x16x32 MFMAs (WECLIP_GEMM_M16 / wc_gemm_set_m16)
extern "C" void wc_gemm_set_m16(int on) { g_pp_m16 = on ? 1 : 0; }
static int g_p192_mode = -1;             // 0: never, 1: by staged bytes, 2: whenever the shape allows
static float g_p192_cost = 1.0f;         // relative cost of a byte staged by the 192-column kernel
extern "C" void wc_gemm_set_p192(int mode, float cost) {
    g_p192_mode = mode;
    if (cost > 0.f) g_p192_cost = cost;
}

static int gemm_plan(int M, int N, int K, int nseg, int batch, bool row_mapped_aux, long lda = 0, long ldw = 0) {
    if (lda <= 0) lda = K;
    if (ldw <= 0) ldw = K;
    static const int pp_mode = getenv("WECLIP_GEMM_PP") ? atoi(getenv("WECLIP_GEMM_PP")) : 1;
    static const int pp_min_tiles = getenv("WECLIP_GEMM_PP_MIN_TILES") ? atoi(getenv("WECLIP_GEMM_PP_MIN_TILES")) : 160;
    if (g_p192_mode < 0) {
        g_p192_mode = getenv("WECLIP_GEMM_P192") ? atoi(getenv("WECLIP_GEMM_P192")) : 1;
        if (getenv("WECLIP_GEMM_P192_COST")) g_p192_cost = (float)atof(getenv("WECLIP_GEMM_P192_COST"));
    }
    const long gx = wc_cdiv(N, 256), gy = wc_cdiv(M, 256);
    if (!pp_mode || batch != 1 || (long)K * nseg < 2 * BK || gx * gy < pp_min_tiles) return 0;
    if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return 0;      // the tall kernels carry 32-bit byte offsets
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    const bool can_split = M % 256 != 0 && !row_mapped_aux;
    const bool split = can_split && wc_cdiv(gx * (gy - 1), n_cu) < wc_cdiv(gx * gy, n_cu);
    const long nt = (long)(K / BK) * nseg;
    if (g_p192_mode && N % 192 == 0 && nt >= 4 && nt % 2 == 0) {
        const long gx2 = N / 192;
        const bool split2 = can_split && wc_cdiv(gx2 * (gy - 1), n_cu) < wc_cdiv(gx2 * gy, n_cu);
        const float c256 = (float)(gx * (split ? gy - 1 : gy)) * (256 + 256);
        const float c192 = (float)(gx2 * (split2 ? gy - 1 : gy)) * (256 + 192) * g_p192_cost;
        if (g_p192_mode >= 2 || c192 < c256) return split2 ? 4 : 3;
    }
    return split ? 2 : 1;
}

extern "C" int wc_gemm_f16_grouped(const void* A0, const void* A1, const void* A2, const void* W0, const void* W1,
                                   const void* W2, int nseg, int M, int N, int K, long lda, long ldw, int batch, long sA,
                                   long sW, long sC, const float* bias, const float* resid, long ldr, long sR, float* C32,
                                   void* C16, void* C16lo, long ldc, int act, int round16, float scale, int scale_cols,
                                   float* P32, const float* aux, const int* rowmap, int rpg, long ldaux, const void* auxh,
                                   const float* cscale, long sCS, int zdiv, long sA2, long sW2, long sC2, long sB2,
                                   long sX2, void* stream);

extern "C" int wc_gemm_plan(int M, int N, int K, int nseg, int batch) {
    return gemm_plan(M, N, K, nseg, batch, false);
}

extern "C" int wc_gemm_f16(const void* A0, const void* A1, const void* A2, const void* W0,
                           const void* W1, const void* W2, int nseg, int M, int N, int K, long lda,
                           long ldw, int batch, long sA, long sW, long sC, const float* bias,
                           const float* resid, long ldr, long sR, float* C32, void* C16, void* C16lo, long ldc, int act,
                           int round16, float scale, int scale_cols, float* P32, const float* aux,
                           const int* rowmap, int rpg, long ldaux, const void* auxh, const float* cscale,
                           long sCS, void* stream) {
    return wc_gemm_f16_grouped(A0, A1, A2, W0, W1, W2, nseg, M, N, K, lda, ldw, batch, sA, sW, sC, bias, resid, ldr, sR,
                               C32, C16, C16lo, ldc, act, round16, scale, scale_cols, P32, aux, rowmap, rpg, ldaux, auxh,
                               cscale, sCS, batch, 0, 0, 0, 0, 0, stream);
}

static int gemm_f16_grouped_impl(const void* A0, const void* A1, const void* A2, const void* W0,
                                   const void* W1, const void* W2, int nseg, int M, int N, int K, long lda,
                                   long ldw, int batch, long sA, long sW, long sC, const float* bias,
                                   const float* resid, long ldr, long sR, float* C32, void* C16, void* C16lo, long ldc,
                                   int act, int round16, float scale, int scale_cols, float* P32, const float* aux,
                                   const int* rowmap, int rpg, long ldaux, const void* auxh, const float* cscale,
                                   long sCS, int zdiv, long sA2, long sW2, long sC2, long sB2, long sX2, void* stream);

extern "C" int wc_gemm_f16_grouped(const void* A0, const void* A1, const void* A2, const void* W0,
                                   const void* W1, const void* W2, int nseg, int M, int N, int K, long lda,
                                   long ldw, int batch, long sA, long sW, long sC, const float* bias,
                                   const float* resid, long ldr, long sR, float* C32, void* C16, void* C16lo, long ldc,
                                   int act, int round16, float scale, int scale_cols, float* P32, const float* aux,
                                   const int* rowmap, int rpg, long ldaux, const void* auxh, const float* cscale,
                                   long sCS, int zdiv, long sA2, long sW2, long sC2, long sB2, long sX2, void* stream) {
    const int sl = shape_log_begin(stream);
    const int rc = gemm_f16_grouped_impl(A0, A1, A2, W0, W1, W2, nseg, M, N, K, lda, ldw, batch, sA, sW, sC, bias, resid, ldr, sR, C32, C16,
                                         C16lo, ldc, act, round16, scale, scale_cols, P32, aux, rowmap, rpg, ldaux, auxh, cscale, sCS, zdiv,
                                         sA2, sW2, sC2, sB2, sX2, stream);
    if (sl >= 0) shape_log_end(sl, C32 ? (C16 ? "f16+f32" : "f32") : "f16", M, N, K, nseg, batch, gemm_plan(M, N, K, nseg, batch, false, lda, ldw), act, stream);
    return rc;
}

static int gemm_f16_grouped_impl(const void* A0, const void* A1, const void* A2, const void* W0,
                                   const void* W1, const void* W2, int nseg, int M, int N, int K, long lda,
                                   long ldw, int batch, long sA, long sW, long sC, const float* bias,
                                   const float* resid, long ldr, long sR, float* C32, void* C16, void* C16lo, long ldc,
                                   int act, int round16, float scale, int scale_cols, float* P32, const float* aux,
                                   const int* rowmap, int rpg, long ldaux, const void* auxh, const float* cscale,
                                   long sCS, int zdiv, long sA2, long sW2, long sC2, long sB2, long sX2, void* stream) {
    WC_CHECK_ARG(zdiv >= 1 && sA2 % 8 == 0 && sW2 % 8 == 0, "wc_gemm_f16_grouped: zdiv >= 1, sA2 / sW2 %% 8 == 0");
    WC_CHECK_ARG(nseg >= 1 && nseg <= 3, "wc_gemm_f16: nseg must be 1..3");
    WC_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0, "wc_gemm_f16: need M,N>0 and K %% 64 == 0 (got M=%d N=%d K=%d)", M, N, K);
    WC_CHECK_ARG(A0 && W0 && (nseg < 2 || (A1 && W1)) && (nseg < 3 || (A2 && W2)),
                 "wc_gemm_f16: null operand");
    WC_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && sA % 8 == 0 && sW % 8 == 0,
                 "wc_gemm_f16: operand rows must be 16-byte aligned (lda, ldw, strides %% 8 == 0)");
    WC_CHECK_ARG(((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)A2 |
                  (uintptr_t)W2) % 16 == 0, "wc_gemm_f16: operands must be 16-byte aligned");
    WC_CHECK_ARG((C32 || C16) && ldc >= N && batch >= 1 && batch <= 65535, "wc_gemm_f16: bad output");
    WC_CHECK_ARG(act >= 0 && act <= 7, "wc_gemm_f16: act must be 0..7");
    WC_CHECK_ARG(act != 5 || (auxh && ldaux >= N), "wc_gemm_f16: act 5 needs auxh, ldaux");
    WC_CHECK_ARG((act != 4 && act != 7) || (aux && rpg > 0 && ldaux >= N), "wc_gemm_f16: act 4 / 7 need aux, rpg, ldaux");
    const bool use_aux = act == 4 || act == 5 || act == 7;      // the epilogue variant that reads a side input
    const bool erf = act == 6 || act == 7;                      // the erf-GELU builds (gemm_epilogue EK 2 / 3)
    GemmArgs g;
    g.A[0] = (const __half*)A0; g.A[1] = (const __half*)A1; g.A[2] = (const __half*)A2;
    g.W[0] = (const __half*)W0; g.W[1] = (const __half*)W1; g.W[2] = (const __half*)W2;
    g.nseg = nseg; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
    g.sA = sA; g.sW = sW; g.sC = sC; g.sR = sR; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.C32 = C32; g.C16 = (__half*)C16; g.C16lo = (__half*)C16lo; g.ldc = ldc;
    g.act = act; g.round16 = round16; g.scale = scale; g.scale_cols = scale_cols;
    g.P32 = P32; g.aux = aux; g.rowmap = rowmap; g.row0 = 0; g.rpg = rpg > 0 ? rpg : 1; g.ldaux = ldaux;
    g.auxh = (const __half*)auxh; g.cscale = cscale; g.sCS = sCS;
    g.zdiv = zdiv; g.sA2 = sA2; g.sW2 = sW2; g.sC2 = sC2; g.sB2 = sB2; g.sX2 = sX2;
    // wide epilogue needs every 4-column group of a row 16-B (fp32) / 8-B (fp16) addressable
    g.vec = (ldc % 4 == 0 && sC % 4 == 0 && sC2 % 4 == 0 && (!resid || (ldr % 4 == 0 && sR % 4 == 0 && (uintptr_t)resid % 16 == 0)) &&
             (!C32 || (uintptr_t)C32 % 16 == 0) && (!C16 || (uintptr_t)C16 % 8 == 0) && (!C16lo || (uintptr_t)C16lo % 8 == 0) &&
             ((act != 4 && act != 7) || (ldaux % 4 == 0 && (uintptr_t)aux % 16 == 0)))
                ? 1 : 0;
    g.auxvec = (act == 5 && ldaux % 4 == 0 && sX2 % 4 == 0 && (uintptr_t)auxh % 8 == 0) ? 1 : 0;
    g.gx = wc_cdiv(N, BN);
    int plan = gemm_plan(M, N, K, nseg, batch, false, lda, ldw);
    if (erf && plan >= 3) plan -= 2;      // the erf epilogues exist for the 256x256 tile only
    if (plan) {   // tall GEMM: 256x256 / 256x192 ping-pong kernel
        const bool p192 = plan >= 3;
        const int tn = p192 ? 192 : 256;
        g.gx = wc_cdiv(N, tn);
        g.gy = wc_cdiv(M, 256);
        dim3 gridp((unsigned)(g.gx * (g.gy >= 16 ? (g.gy + 7) / 8 * 8 : g.gy)), 1, 1);
        static bool lds_attr_set = false;
        if (!lds_attr_set) {      // 128 / 112 KiB of dynamic LDS are above the default per-kernel limit
            WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_p192_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 14 * P192_UNIT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_p192_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 14 * P192_UNIT) == hipSuccess,
                         "wc_gemm_f16: cannot reserve 128 KiB of LDS");
            lds_attr_set = true;
        }
        // A ragged last row of tiles (M % 256 rows) costs every CU a whole extra round when it tips the tile
        // count over a multiple of the CU count (ViT-B fc1 at 16 x 1025 tokens: 65 x 12 tiles = 3.05 rounds):
        // those rows then go to the 128x128 kernel in a second, small launch.
        const int m_main = M / 256 * 256, m_rem = M - m_main;
        const bool split = plan == 2 || plan == 4;
        if (split) {
            g.M = m_main;
            g.gy -= 1;
            gridp.x = (unsigned)(g.gx * (g.gy >= 16 ? (g.gy + 7) / 8 * 8 : g.gy));
        }
        const int pr = wc_prof_begin(stream);
        if (p192) {
            if (use_aux)
                hipLaunchKernelGGL(gemm_f16_p192_kernel<true>, gridp, dim3(512), 14 * P192_UNIT, (hipStream_t)stream, g);
            else
                hipLaunchKernelGGL(gemm_f16_p192_kernel<false>, gridp, dim3(512), 14 * P192_UNIT, (hipStream_t)stream, g);
            wc_prof_end(pr, use_aux ? "gemm_f16_p192_kernel<1>" : "gemm_f16_p192_kernel<0>", 2.0 * g.M * N * K, stream);
            WC_LAUNCH_CHECK("gemm_f16_p192_kernel");
        } else {
            if (g_pp_m16 < 0) g_pp_m16 = getenv("WECLIP_GEMM_M16") ? atoi(getenv("WECLIP_GEMM_M16")) : 0;
            if (g_pp_ring10 < 0) g_pp_ring10 = getenv("WECLIP_GEMM_RING10") ? atoi(getenv("WECLIP_GEMM_RING10")) : 1;
            if (g_w4 < 0) g_w4 = getenv("WECLIP_GEMM_W4") ? atoi(getenv("WECLIP_GEMM_W4")) : 0;
            if (erf) {
                static bool erf_attr = false;
                if (!erf_attr) {
                    WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                                 hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess,
                                 "wc_gemm_f16: cannot reserve 128 KiB of LDS");
                    erf_attr = true;
                }
                if (use_aux)
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<3, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
                else
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<2, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
                wc_prof_end(pr, use_aux ? "gemm_f16_pp_kernel<3, false, 8>" : "gemm_f16_pp_kernel<2, false, 8>", 2.0 * g.M * N * K, stream);
                WC_LAUNCH_CHECK("gemm_f16_pp_kernel");
            } else if (g_r4) {
                static bool r4_attr = false;
                if (!r4_attr) {
                    WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_r4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG) == hipSuccess &&
                                 hipFuncSetAttribute((const void*)gemm_f16_r4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG) == hipSuccess,
                                 "wc_gemm_f16: cannot reserve 128 KiB of LDS");
                    r4_attr = true;
                }
                if (g_r4 == 2 && !use_aux) {
                    static bool r4b_attr = false;
                    if (!r4b_attr) {
                        WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_r4_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG) == hipSuccess,
                                     "wc_gemm_f16: cannot reserve 128 KiB of LDS");
                        r4b_attr = true;
                    }
                    hipLaunchKernelGGL((gemm_f16_r4_kernel<0, true>), gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g);
                } else if (use_aux)
                    hipLaunchKernelGGL(gemm_f16_r4_kernel<1>, gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g);
                else
                    hipLaunchKernelGGL(gemm_f16_r4_kernel<0>, gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g);
                wc_prof_end(pr, use_aux ? "gemm_f16_r4_kernel<1>" : "gemm_f16_r4_kernel<0>", 2.0 * g.M * N * K, stream);
                WC_LAUNCH_CHECK("gemm_f16_r4_kernel");
            } else if (g_w4) {
                if (use_aux)
                    hipLaunchKernelGGL(gemm_f16_w4_kernel<true>, gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g);
#ifdef W4_EXPERIMENTS
                else if (g_w4 > 1) {
                    static bool exp_attr = false;
#define W4_EXP_CASE(e_) case e_: if (!exp_attr) hipFuncSetAttribute((const void*)gemm_f16_w4_kernel<false, e_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_IMG); \
                    hipLaunchKernelGGL((gemm_f16_w4_kernel<false, e_>), gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g); break;
                    switch (g_w4 - 1) { W4_EXP_CASE(1) W4_EXP_CASE(3) W4_EXP_CASE(4) W4_EXP_CASE(7) W4_EXP_CASE(15) W4_EXP_CASE(8) default: break; }
                }
#endif
                else
                    hipLaunchKernelGGL(gemm_f16_w4_kernel<false>, gridp, dim3(256), 2 * W4_IMG, (hipStream_t)stream, g);
                wc_prof_end(pr, use_aux ? "gemm_f16_w4_kernel<1, 0>" : "gemm_f16_w4_kernel<0, 0>", 2.0 * g.M * N * K, stream);
                WC_LAUNCH_CHECK("gemm_f16_w4_kernel");
            } else if (g_pp_m16) {
                if (use_aux)
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<true, true>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
                else
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<false, true>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
            } else if (g_pp_ring10) {
                static bool r10_attr = false;
                if (!r10_attr) {
                    WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<0, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PP_SLOT) == hipSuccess &&
                                 hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<1, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PP_SLOT) == hipSuccess,
                                 "wc_gemm_f16: cannot reserve 160 KiB of LDS");
                    r10_attr = true;
                }
                if (use_aux)
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<1, false, 10>), gridp, dim3(512), 10 * PP_SLOT, (hipStream_t)stream, g);
                else
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<0, false, 10>), gridp, dim3(512), 10 * PP_SLOT, (hipStream_t)stream, g);
            } else {
                if (use_aux)
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<true, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
                else
                    hipLaunchKernelGGL((gemm_f16_pp_kernel<false, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
            }
            wc_prof_end(pr, g_pp_m16 ? (use_aux ? "gemm_f16_pp_kernel<1, true, 8>" : "gemm_f16_pp_kernel<0, true, 8>")
                            : g_pp_ring10 ? (use_aux ? "gemm_f16_pp_kernel<1, false, 10>" : "gemm_f16_pp_kernel<0, false, 10>")
                                          : (use_aux ? "gemm_f16_pp_kernel<1, false, 8>" : "gemm_f16_pp_kernel<0, false, 8>"), 2.0 * g.M * N * K, stream);
            WC_LAUNCH_CHECK("gemm_f16_pp_kernel");
        }
        if (!split) return WC_OK;
        for (int i = 0; i < nseg; ++i) g.A[i] += (long)m_main * lda;
        if (g.resid) g.resid += (long)m_main * ldr;
        if (g.C32) g.C32 += (long)m_main * ldc;
        if (g.C16) g.C16 += (long)m_main * ldc;
        if (g.C16lo) g.C16lo += (long)m_main * ldc;
        if (g.P32) g.P32 += (long)m_main * ldc;
        if (g.aux && !g.rowmap) g.aux += (long)m_main * ldaux;
        if (g.rowmap) g.row0 = m_main;          // row-mapped aux rows: keep the pointer, shift the row index
        if (g.auxh) g.auxh += (long)m_main * ldaux;
        g.M = M = m_rem;
        g.gx = wc_cdiv(N, BN);
    }
    static const int skinny_env = getenv("WECLIP_GEMM_SKINNY") ? atoi(getenv("WECLIP_GEMM_SKINNY")) : 1;
    if (skinny_env && M <= 32 && batch == 1 && N >= 256) {      // a few rows against many weight rows
        const int prs = wc_prof_begin(stream);
        if (erf && use_aux)
            hipLaunchKernelGGL(gemm_skinny_kernel<3>, dim3(wc_cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, g);
        else if (erf)
            hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3(wc_cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, g);
        else if (use_aux)
            hipLaunchKernelGGL(gemm_skinny_kernel<true>, dim3(wc_cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(gemm_skinny_kernel<false>, dim3(wc_cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, g);
        wc_prof_end(prs, erf ? "gemm_skinny_kernel<erf>" : use_aux ? "gemm_skinny_kernel<1>" : "gemm_skinny_kernel<0>", 2.0 * M * N * K, stream);
        WC_LAUNCH_CHECK("gemm_skinny_kernel");
        return WC_OK;
    }
    g.gy = wc_cdiv(M, BM);
    dim3 grid((unsigned)(g.gx * ((g.gy + 7) / 8 * 8)), 1, batch);
    // at most one workgroup per CU: the 4-stage ring (128 KiB); else two 2-stage workgroups per CU
    static int n_cu128 = 0;
    if (!n_cu128) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu128, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu128 <= 0)
            n_cu128 = 256;
        WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * BM * BK * 2) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_f16_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * BM * BK * 2) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_f16_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * BM * BK * 2) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_f16_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * BM * BK * 2) == hipSuccess,
                     "wc_gemm_f16: cannot reserve 128 KiB of LDS");
    }
    static const int ring_env = getenv("WECLIP_GEMM_RING") ? atoi(getenv("WECLIP_GEMM_RING")) : 1;
    const bool ring = ring_env && (long)g.gx * g.gy * batch <= n_cu128 && K / BK * nseg >= 3;
    const size_t lds = (ring ? 4 : 2) * 2 * BM * BK * 2;
    const int pr = wc_prof_begin(stream);
    if (erf) {
        if (use_aux) {
            if (ring) hipLaunchKernelGGL((gemm_f16_kernel<3, 4>), grid, dim3(256), lds, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((gemm_f16_kernel<3, 2>), grid, dim3(256), lds, (hipStream_t)stream, g);
        } else {
            if (ring) hipLaunchKernelGGL((gemm_f16_kernel<2, 4>), grid, dim3(256), lds, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((gemm_f16_kernel<2, 2>), grid, dim3(256), lds, (hipStream_t)stream, g);
        }
    } else if (use_aux) {
        if (ring) hipLaunchKernelGGL((gemm_f16_kernel<true, 4>), grid, dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_f16_kernel<true, 2>), grid, dim3(256), lds, (hipStream_t)stream, g);
    } else {
        if (ring) hipLaunchKernelGGL((gemm_f16_kernel<false, 4>), grid, dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_f16_kernel<false, 2>), grid, dim3(256), lds, (hipStream_t)stream, g);
    }
    wc_prof_end(pr, erf ? (ring ? "gemm_f16_kernel<erf, 4>" : "gemm_f16_kernel<erf, 2>")
                    : use_aux ? (ring ? "gemm_f16_kernel<1, 4>" : "gemm_f16_kernel<1, 2>")
                              : (ring ? "gemm_f16_kernel<0, 4>" : "gemm_f16_kernel<0, 2>"), 2.0 * g.M * N * K * batch, stream);
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
