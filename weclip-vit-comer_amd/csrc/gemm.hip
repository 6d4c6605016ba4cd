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
int shape_log_begin(void* stream) {
    if (g_shape_log < 0) g_shape_log = getenv("WECLIP_GEMM_LOG") ? atoi(getenv("WECLIP_GEMM_LOG")) : 0;
    if (!g_shape_log) return -1;
    GemmShapeRec r;
    r.key[0] = 0; r.flop = 0;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, (hipStream_t)stream);
    g_shape_recs.push_back(r);
    return (int)g_shape_recs.size() - 1;
}
void shape_log_end(int idx, const char* kind, int M, int N, int K, int nseg, int batch, int plan, int act, void* stream) {
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

#include "gemm_common.h"

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

// Which kernel a shape takes: 0 = 128x128 kernel, 1 = 256x256 ping-pong kernel, 2 = ping-pong kernel on the
// full 256-row tiles + the 128x128 / few-rows kernel on the ragged last M % 256 rows (see wc_gemm_f16).
// (Round 3 built three more schedules of the 256x256 tile -- a 256x192 tile, a 4-wave register-staged kernel, a 4-wave kernel
// with register-resident fragments -- and the 16x16x32-MFMA build of this one; all bit-identical, all slower on the step's
// shapes: DESIGN.md section 5, profiles/r03_gemm_tiles.txt, r03_gemm_w4_m16.txt; sources archived under tools/probes/rejected_r03/.)
static int gemm_plan(int M, int N, int K, int nseg, int batch, bool row_mapped_aux, long lda = 0, long ldw = 0) {
    if (lda <= 0) lda = K;
    if (ldw <= 0) ldw = K;
    static const int pp_mode = getenv("WECLIP_GEMM_PP") ? atoi(getenv("WECLIP_GEMM_PP")) : 1;
    const int pp_min_tiles = 160;
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
    const int plan = gemm_plan(M, N, K, nseg, batch, false, lda, ldw);
    if (plan) {   // tall GEMM: 256x256 ping-pong kernel
        g.gx = wc_cdiv(N, 256);
        g.gy = wc_cdiv(M, 256);
        dim3 gridp((unsigned)(g.gx * (g.gy >= 16 ? (g.gy + 7) / 8 * 8 : g.gy)), 1, 1);
        static bool lds_attr_set = false;
        if (!lds_attr_set) {      // 160 / 128 KiB of dynamic LDS are above the default per-kernel limit
            WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<0, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<1, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess &&
                         hipFuncSetAttribute((const void*)gemm_f16_pp_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PP_SLOT) == hipSuccess,
                         "wc_gemm_f16: cannot reserve 160 KiB of LDS");
            lds_attr_set = true;
        }
        // A ragged last row of tiles (M % 256 rows) costs every CU a whole extra round when it tips the tile
        // count over a multiple of the CU count (ViT-B fc1 at 16 x 1025 tokens: 65 x 12 tiles = 3.05 rounds):
        // those rows then go to the 128x128 kernel in a second, small launch.
        const int m_main = M / 256 * 256, m_rem = M - m_main;
        const bool split = plan == 2;
        if (split) {
            g.M = m_main;
            g.gy -= 1;
            gridp.x = (unsigned)(g.gx * (g.gy >= 16 ? (g.gy + 7) / 8 * 8 : g.gy));
        }
        const int pr = wc_prof_begin(stream);
        if (erf) {      // the erf-GELU epilogues: 8-slot ring builds of their own (register budget)
            if (use_aux)
                hipLaunchKernelGGL((gemm_f16_pp_kernel<3, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
            else
                hipLaunchKernelGGL((gemm_f16_pp_kernel<2, false>), gridp, dim3(512), 8 * PP_SLOT, (hipStream_t)stream, g);
            wc_prof_end(pr, use_aux ? "gemm_f16_pp_kernel<3, false, 8>" : "gemm_f16_pp_kernel<2, false, 8>", 2.0 * g.M * N * K, stream);
        } else {
            if (use_aux)
                hipLaunchKernelGGL((gemm_f16_pp_kernel<1, false, 10>), gridp, dim3(512), 10 * PP_SLOT, (hipStream_t)stream, g);
            else
                hipLaunchKernelGGL((gemm_f16_pp_kernel<0, false, 10>), gridp, dim3(512), 10 * PP_SLOT, (hipStream_t)stream, g);
            wc_prof_end(pr, use_aux ? "gemm_f16_pp_kernel<1, false, 10>" : "gemm_f16_pp_kernel<0, false, 10>", 2.0 * g.M * N * K, stream);
        }
        WC_LAUNCH_CHECK("gemm_f16_pp_kernel");
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
    const int skinny_env = 1;
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
    const int ring_env = 1;
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
