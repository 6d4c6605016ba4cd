// Row-streaming GEMM for TALL, NARROW products:  C[M, N <= 256] = epilogue(A[M, K] W[N, K]^T),  K = 128 or 256.
//
// Call sites: the Linear / 1x1-conv layers of the ViT-CoMer inserts (86 016 pyramid rows x 256 channels per image batch; no
// reference code: ViT_CoMer.pdf section 3.2-3.3) and every other F.linear of the hot path whose
// output row is one complete channel vector (WeCLIP_model/segformer_head.py:22-28 proj_2, Decoder/TransDecoder.py).
//
// Why a kernel of its own.  At K = N = 256 such a product moves 0.5 KiB of operand and 1-2.5 KiB of output / residual per
// row for 131 kflop: it is bound by HBM bytes (86 016 rows: 44 MB in, 88-220 MB out), not by the matrix pipe, and the
// 256x256 tile kernel ran it at a third of the HBM rate (1.3 rounds of one workgroup per CU that load, multiply and store
// one after the other; weights re-staged through LDS for every tile; 64-byte output segments).  Here
//   * the WEIGHTS ARE STATIONARY IN REGISTERS: a wave owns 64 output columns and keeps their whole K extent as MFMA B
//     fragments (128 registers at K = 256), loaded once per workgroup; workgroups are persistent (two per CU, each walks
//     its 32-row tiles), so the only operand traffic is A, read exactly once;
//   * A tiles (32 rows x K) arrive by LDS-DMA into a two-slot ring, the next tile's DMA issued before this tile's MFMAs;
//     XOR-swizzled 16-byte chunks (chunk c of row r at c ^ (r & 15)) keep the ds_read_b128 fragment reads conflict-free;
//   * the epilogue works on WHOLE ROWS: the accumulators are dropped into a [32][256] fp32 tile in LDS, then a wave takes a
//     row at a time -- lane l owns columns 4l..4l+3 -- so that every side input (residual, saved pre-activation) is one
//     1-KiB coalesced load and every output one 1-KiB (fp32) / 512-B (fp16) coalesced store;
//   * and because a workgroup holds complete rows, the LayerNorm(s) that consume the output (up to two different affine
//     parameter sets) are computed right there and leave as the fp16 operand of the next GEMM: no LayerNorm launch, no
//     second read of the 88 MB activation.
// Two workgroups of four waves per CU overlap each other's load / multiply / store phases.
//
// vmcnt discipline: LDS-DMA, loads and stores share one in-order counter.  Each iteration issues the next tile's DMA first
// and contains ONE explicit `s_waitcnt vmcnt(0)` -- after the row phase's first side loads, before its first store -- which
// therefore also covers that DMA; the barrier at the top of the next iteration then makes every wave's pieces visible.  No
// counted waits (a spilled register or a skipped store cannot break the accounting).
#include "gemm_common.h"
#include <stdlib.h>

struct RowArgs {
    const __half* A;
    const __half* W;
    int M, N, K;
    long lda, ldw;
    const float* bias;      // [N] or null
    const float* cscale;    // [N] or null: v *= cscale[n] after the bias (CTI gate gamma)
    int act;                // 0 none, 2 ReLU, 6 GELU (erf); 5: v *= (auxh > 0); 7: v *= GELU'(aux)
    const float* aux;       // act 7: saved pre-activation, fp32 (M, ldaux)
    const __half* auxh;     // act 5: saved fp16 output (M, ldaux)
    long ldaux;
    const float* resid;     // [M, ldr] fp32 or null, added last
    long ldr;
    float* C32;             // any subset of the three outputs
    __half* C16;
    float* P32;             // value before the activation (after bias and column scale)
    long ldc, ldc16;        // row pitch of C32 / P32, and of C16 (e.g. a column slice of a wider concat buffer)
    const float* ln_g[2];   // fused LayerNorm outputs of the final value (N == 256 only): gamma, beta, fp16 out (M, N)
    const float* ln_b[2];
    __half* ln_o[2];
    float eps;
    int ntiles;
    // grouped launch: `groups` products of one shape (the eleven adapters of WeCLIP_model/segformer_head.py:69-80); group i reads
    // A + i*gA, W + i*gW, bias + i*gB, the side input + i*gX and writes its outputs at + i*gC (all in elements); `wpg`
    // persistent workgroups per group (blockIdx.x = group * wpg + member)
    int groups, wpg;
    long gA, gW, gB, gC, gX;
};

#define ROW_TM 16
#define ROW_BAR()                                              \
    {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    }
// LDS access inside the loop is inline asm throughout: hipcc cannot tell which LDS bytes an in-flight LDS-DMA writes and puts
// `s_waitcnt vmcnt(0)` in front of any LDS read it can see, which would drain the ring; the waits below are placed by hand.
typedef float rf32x4 __attribute__((ext_vector_type(4)));
#define ROW_LD128(dst_, addr_) asm volatile("ds_read_b128 %0, %1" : "=v"(dst_) : "v"(addr_))
#define ROW_LD64(dst_, addr_) asm volatile("ds_read_b64 %0, %1" : "=v"(dst_) : "v"(addr_))
#define ROW_ST32(addr_, val_) asm volatile("ds_write_b32 %0, %1" ::"v"(addr_), "v"(val_) : "memory")

// LDS: 3 A slots (16 rows x K fp16) | 2 side slots (16 x 256 fp32) | C tile (16 x 256 fp32) | 6 x 256 column constants
// = 78 KiB at K = 256: two workgroups per CU.
template <int KT, bool ERF>
__global__ __launch_bounds__(256, 2) void gemm_row_kernel(RowArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = KT * 64;
    constexpr int SLOT = ROW_TM * K * 2;      // bytes of one A tile
    constexpr int CPR = K / 8;                // 16-byte chunks per A row
    constexpr int KS = K / 32;                // MFMA k-steps (v_mfma_f32_16x16x32_f16)
    constexpr int NDMA = ROW_TM * CPR / 256;  // A pieces per thread and tile (2 at K = 256, 1 at K = 128)
    constexpr int SSLOT = ROW_TM * 256 * 4;   // bytes of one side-input slot
    constexpr int SIDE0 = 3 * SLOT, CB0 = SIDE0 + 2 * SSLOT, CT0 = CB0 + ROW_TM * 256 * 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const unsigned lbase = (unsigned)(size_t)(lds_ptr)smem;
    float* Ct = reinterpret_cast<float*>(smem + CT0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q16 = lane >> 4;
    const int G = g.wpg;                                     // workgroups walking this group's tiles
    const int grp = __builtin_amdgcn_readfirstlane((int)blockIdx.x / g.wpg);
    const int wg0 = (int)blockIdx.x - grp * g.wpg;           // this workgroup's first tile
    const int nmy = (g.ntiles - wg0 + G - 1) / G;
    g.A += (long)grp * g.gA;
    g.W += (long)grp * g.gW;
    if (g.bias) g.bias += (long)grp * g.gB;
    if (g.C32) g.C32 += (long)grp * g.gC;
    if (g.C16) g.C16 += (long)grp * g.gC;
    if (g.P32) g.P32 += (long)grp * g.gC;
    if (g.resid) g.resid += (long)grp * g.gX;
    if (g.aux) g.aux += (long)grp * g.gX;
    if (g.auxh) g.auxh += (long)grp * g.gX;
    const int act = g.act;
    // the ONE side input of the launch (checked by the host): fp32 rows (the residual, or the saved pre-activation of act 7)
    // or fp16 rows (the saved output of act 5); it arrives by LDS-DMA like A, as a dense [16][N] image, so that the row phase
    // costs no registers and no load sits between its stores in the in-order vmcnt queue
    const char* sptr = g.resid ? reinterpret_cast<const char*>(g.resid)
                               : (ERF && act == 7) ? reinterpret_cast<const char*>(g.aux)
                                                   : act == 5 ? reinterpret_cast<const char*>(g.auxh) : nullptr;
    const int sel = act == 5 ? 2 : 4;                          // bytes per side element
    const long sld = g.resid ? g.ldr : g.ldaux;                // side row pitch in elements
    const int spr = g.N * sel / 16;                            // 16-byte pieces per side row (N % 8 == 0 for fp16 sides)
    const int spieces = ROW_TM * spr;                          // <= 1024
    const int nsd = sptr ? (spieces + 255) / 256 : 0;          // side DMA instructions per thread and tile (uniform)

    // A pieces of this thread: chunk q = j * 256 + tid of the tile image = (row q / CPR, physical chunk q % CPR), which holds
    // the row's logical chunk (q % CPR) ^ (row & 15)
    int drow[NDMA], dcol[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int q = j * 256 + tid;
        drow[j] = q / CPR;
        dcol[j] = (((q % CPR) ^ (drow[j] & 15))) * 8;
    }
#define ROW_DMA_A(tile_, slot_)                                                                                   \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NDMA; ++j) {                                                        \
            long r_ = (long)(tile_) * ROW_TM + drow[j];                                                           \
            if (r_ > g.M - 1) r_ = g.M - 1;                                                                       \
            __builtin_amdgcn_global_load_lds((gbl_ptr)(g.A + r_ * g.lda + dcol[j]),                               \
                                             (lds_ptr)(smem + (slot_) * SLOT + (j * 256 + wave * 64) * 16), 16, 0, 0); \
        }                                                                                                         \
    }
#define ROW_DMA_S(tile_, slot_)                                                                                   \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
            if (j < nsd) {      /* (uniform) */                                                                   \
                int q_ = j * 256 + tid;                                                                           \
                if (q_ > spieces - 1) q_ = spieces - 1;      /* surplus lanes re-fetch the last piece */          \
                const int sr_ = q_ / spr;                                                                         \
                long r_ = (long)(tile_) * ROW_TM + sr_;                                                           \
                if (r_ > g.M - 1) r_ = g.M - 1;                                                                   \
                __builtin_amdgcn_global_load_lds((gbl_ptr)(sptr + (r_ * sld) * sel + (q_ - sr_ * spr) * 16),      \
                                                 (lds_ptr)(smem + SIDE0 + (slot_) * SSLOT + (j * 256 + wave * 64) * 16), 16, 0, 0); \
            }                                                                                                     \
        }                                                                                                         \
    }
    if (nmy > 0) { ROW_DMA_A(wg0, 0); ROW_DMA_S(wg0, 0); }
    if (nmy > 1) ROW_DMA_A(wg0 + G, 1);

    // stationary weights: B fragments (16 columns x 32 k: lane = column l15, k group q16) of this wave's 64 columns, whole K
    const int cw = wave * 64;
    const bool has_cols = cw < g.N;
    f16x8 wf[4][KS];
    if (has_cols) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            int n = cw + cb * 16 + l15;
            if (n > g.N - 1) n = g.N - 1;
            const __half* wp = g.W + (long)n * g.ldw + q16 * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[cb][ks] = *reinterpret_cast<const f16x8*>(wp + ks * 32);
        }
    }
    // column constants in LDS (bias, column scale, two LayerNorm gamma / beta sets): as registers they would not fit beside
    // the 128 weight registers, as global loads inside the row phase their waits would drain the stores issued before them
    {
        // (six loads issued together, a missing vector reading one dummy word of W; the pin keeps hipcc from sinking them back
        //  under their conditions, where each was waited for alone: five dependent L2 latencies at the head of every workgroup)
        const int n = tid < g.N ? tid : g.N - 1;
        const float* dz = reinterpret_cast<const float*>(g.W);
        const bool l0 = g.ln_o[0] != nullptr, l1 = g.ln_o[1] != nullptr;
        float r0 = (g.bias ? g.bias : dz)[g.bias ? n : 0], r1 = (g.cscale ? g.cscale : dz)[g.cscale ? n : 0];
        float r2 = (l0 ? g.ln_g[0] : dz)[l0 ? n : 0], r3 = (l0 ? g.ln_b[0] : dz)[l0 ? n : 0];
        float r4 = (l1 ? g.ln_g[1] : dz)[l1 ? n : 0], r5 = (l1 ? g.ln_b[1] : dz)[l1 ? n : 0];
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5));
        Ct[tid] = g.bias ? r0 : 0.f;
        Ct[256 + tid] = g.cscale ? r1 : 1.f;
        Ct[512 + tid] = l0 ? r2 : 0.f;
        Ct[768 + tid] = l0 ? r3 : 0.f;
        Ct[1024 + tid] = l1 ? r4 : 0.f;
        Ct[1280 + tid] = l1 ? r5 : 0.f;
    }
    const bool has_res = g.resid != nullptr;
    const int c4 = lane * 4;                           // row phase: lane l owns columns 4l .. 4l+3
    const bool cact = c4 < g.N;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const unsigned aoff = lbase + l15 * (K * 2);
    const unsigned x4 = (unsigned)(q16 ^ l15);         // chunk (ks * 4 + q16) ^ l15 = (ks * 4) ^ x4: the two terms share no bit
    const unsigned cwr = lbase + CB0 + ((4 * q16) * 256 + cw + l15) * 4;       // C-tile address of accumulator element 0, block 0
    const unsigned crd = lbase + CB0 + (wave * 4 * 256 + c4) * 4;              // C-tile address of this wave's first row
    const unsigned ctc = lbase + CT0 + c4 * 4;
    int sa = 0;                                        // A slot of tile i (i % 3)
    for (int i = 0; i < nmy; ++i) {
        const int t = wg0 + i * G;
        const int ss = i & 1;
        ROW_BAR();                                   // tile i (A and side) is in LDS for all waves; the C tile and the slots refilled below are free
        const bool full = i + 2 < nmy;
        if (full) ROW_DMA_A(t + 2 * G, sa == 0 ? 2 : sa - 1);
        if (i + 1 < nmy) ROW_DMA_S(t + G, ss ^ 1);
        if (has_cols) {
            f32x4 acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned ab = aoff + sa * SLOT;
            f16x8 fa[4];      // the fragments of three k-steps are in flight ahead of the MFMAs (an LDS read takes longer than 4 MFMAs)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) ROW_LD128(fa[ks], ab + (((ks * 4) ^ x4) << 4));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 3 < KS) {
                    ROW_LD128(fa[(ks + 3) & 3], ab + ((((ks + 3) * 4) ^ x4) << 4));
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[ks & 3]));
                } else if (ks + 2 < KS) {
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 3]));
                } else if (ks + 1 < KS) {
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[ks & 3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 3]));
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks & 3], wf[cb][ks], acc[cb], 0, 0, 0);
            }
            // accumulators -> C tile: element e of block cb = row 4 q16 + e, column cw + 16 cb + l15.
            // The stores are inline asm, which the compiler's hazard recognizer does not see: an LDS store may read an MFMA
            // result only a number of wait states after the MFMA issued (software-managed on CDNA), so all MFMAs are retired
            // behind explicit s_nops first (the "+v" ties make every accumulator final before the nops).
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) ROW_ST32(cwr + (e * 256 + cb * 16) * 4, acc[cb][e]);
        }
        ROW_BAR();
        // ---- the iteration's one vmcnt wait, in front of the row phase's stores.  Steady state: leave the DMA batch issued at the
        // top of THIS iteration (A of tile i + 2, side of tile i + 1: NDMA + nsd instructions per wave, uniform) in flight.
        // LDS-DMA / loads complete in issue order among themselves, so "at most that many operations outstanding" implies that
        // the previous iteration's batch (this tile's side input, the next tile's A) has landed.
        // NOT vmcnt(NDMA + nsd + stores of the previous row phase): stores do not complete in order with older loads -- with
        // that count the wait could pass on completed stores while a piece of the previous batch was still in flight (built in
        // round 4: same speed, and the full-size reproducibility test caught the race as 1e-6 differences of the loss).
        // The last two iterations drain.
        if (full) {
            switch (NDMA + nsd) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // ---- row phase: this wave's 4 rows TOGETHER (one LDS round trip, one uniform branch per feature instead of one per row
        // and feature: the per-row form spent ~3 us per tile on LDS latencies and taken scalar branches, a floor of ~30 us
        // under the 86 016-row launches whatever their HBM traffic)
        constexpr int RB = ERF ? 2 : 4;      // rows per batch (the erf builds carry more live values per row)
        rf32x4 bias4, cs4;
        ROW_LD128(bias4, ctc);
        ROW_LD128(cs4, ctc + 1024);
#pragma unroll
        for (int hb = 0; hb < 4 / RB; ++hb) {
        rf32x4 c[RB], sd[RB];
        u32x2 sh[RB];
        const unsigned srd = lbase + SIDE0 + ss * SSLOT + (((wave * 4 + hb * RB) * g.N + (cact ? c4 : 0)) * sel);
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            ROW_LD128(c[j], crd + (hb * RB + j) * 1024);
            sd[j] = rf32x4{0.f, 0.f, 0.f, 0.f};
            sh[j] = u32x2{0u, 0u};
        }
        if (sptr && sel == 4) {
#pragma unroll
            for (int j = 0; j < RB; ++j) ROW_LD128(sd[j], srd + j * g.N * 4);
        }
        if (sel == 2) {
#pragma unroll
            for (int j = 0; j < RB; ++j) ROW_LD64(sh[j], srd + j * g.N * 2);
        }
        if constexpr (RB == 4)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(bias4), "+v"(cs4), "+v"(c[0]), "+v"(c[1]), "+v"(c[RB - 2]), "+v"(c[RB - 1]), "+v"(sd[0]), "+v"(sd[1]),
                           "+v"(sd[RB - 2]), "+v"(sd[RB - 1]), "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[RB - 2]), "+v"(sh[RB - 1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias4), "+v"(cs4), "+v"(c[0]), "+v"(c[1]), "+v"(sd[0]), "+v"(sd[1]), "+v"(sh[0]), "+v"(sh[1]));
        const long grow0 = (long)t * ROW_TM + wave * 4 + hb * RB;
        const int nrow = g.M - grow0 >= RB ? RB : (int)(g.M - grow0);      // < RB (or <= 0) in the ragged last tile only
        float v[RB][4];
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[j][k] = (c[j][k] + bias4[k]) * cs4[k];
        if (g.P32) {
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (cact && j < nrow) *reinterpret_cast<float4*>(g.P32 + (grow0 + j) * g.ldc + c4) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
        if (act == 2) {
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[j][k] = fmaxf(v[j][k], 0.f);
        } else if (act == 5) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const __half* hp = reinterpret_cast<const __half*>(&sh[j]);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[j][k] *= __half2float(hp[k]) > 0.f ? 1.f : 0.f;
            }
        } else if (ERF && (act == 6 || act == 7)) {
            // exact GELU / its derivative by the 12-instruction form of common.h (the library erff needed a non-unrolled loop
            // rotating one value at a time through it to fit beside the 128 weight registers, and made this epilogue VALU-bound).
            // act 7: u = the saved pre-activation, d/du [u * Phi(u)] = Phi(u) + u * phi(u)
            if (act == 7) {
#pragma unroll
                for (int j = 0; j < RB; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[j][k] *= wc_gelu_grad(sd[j][k]);
            } else {
#pragma unroll
                for (int j = 0; j < RB; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[j][k] = wc_gelu(v[j][k]);
            }
        }
        if (has_res) {
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[j][k] += sd[j][k];
        }
        if (g.C32) {
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (cact && j < nrow) *reinterpret_cast<float4*>(g.C32 + (grow0 + j) * g.ldc + c4) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
        if (g.C16) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                __half h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = __float2half(v[j][k]);
                if (cact && j < nrow) *reinterpret_cast<u32x2*>(g.C16 + (grow0 + j) * g.ldc16 + c4) = *reinterpret_cast<u32x2*>(h);
            }
        }
        if (g.ln_o[0]) {      // N == 256: all 64 lanes hold 4 columns; same arithmetic as layernorm_kernel<1> (norm.hip)
            float mean[RB], rstd[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) mean[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) / 256.f;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[j][k] -= mean[j];
                rstd[j] = rsqrtf(wave_sum((v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3])) / 256.f + g.eps);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!g.ln_o[s]) continue;
                rf32x4 ww, bb;
                ROW_LD128(ww, ctc + 2048 + s * 2048);
                ROW_LD128(bb, ctc + 3072 + s * 2048);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ww), "+v"(bb));
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    __half h[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) h[k] = __float2half(v[j][k] * rstd[j] * ww[k] + bb[k]);
                    if (j < nrow) *reinterpret_cast<u32x2*>(g.ln_o[s] + (grow0 + j) * 256 + c4) = *reinterpret_cast<u32x2*>(h);
                }
            }
        }
        }
        sa = sa == 2 ? 0 : sa + 1;
    }
#undef ROW_DMA_A
#undef ROW_DMA_S
}

static int g_row_cus = 0;

// C ABI: see include/weclip_hip.h
extern "C" int wc_gemm_row_supported(int M, int N, int K) {
    return (M > 0 && N > 0 && N <= 256 && N % 4 == 0 && (K == 128 || K == 256)) ? 1 : 0;
}

static int gemm_row_impl(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                         const float* cscale, int act, const float* aux, const void* auxh, long ldaux, const float* resid,
                         long ldr, float* C32, void* C16, float* P32, long ldc, long ldc16, const float* ln_g0, const float* ln_b0,
                         void* ln_o0, const float* ln_g1, const float* ln_b1, void* ln_o1, float eps, int groups, long gA, long gW,
                         long gB, long gC, long gX, void* stream);

extern "C" int wc_gemm_row_f16(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                               const float* cscale, int act, const float* aux, const void* auxh, long ldaux, const float* resid,
                               long ldr, float* C32, void* C16, float* P32, long ldc, long ldc16, const float* ln_g0, const float* ln_b0,
                               void* ln_o0, const float* ln_g1, const float* ln_b1, void* ln_o1, float eps, void* stream) {
    return gemm_row_impl(A, lda, W, ldw, M, N, K, bias, cscale, act, aux, auxh, ldaux, resid, ldr, C32, C16, P32, ldc, ldc16, ln_g0,
                         ln_b0, ln_o0, ln_g1, ln_b1, ln_o1, eps, 1, 0, 0, 0, 0, 0, stream);
}

extern "C" int wc_gemm_row_f16_grouped(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                                       int act, const void* auxh, long ldaux, float* C32, void* C16, long ldc, long ldc16, int groups,
                                       long gA, long gW, long gB, long gC, long gX, void* stream) {
    WC_CHECK_ARG(groups >= 1 && groups <= 256 && gA % 8 == 0 && gW % 8 == 0 && gB % 4 == 0 && gC % 4 == 0 && gX % 8 == 0,
                 "wc_gemm_row_f16_grouped: 1..256 groups, group strides that keep every row 16-byte aligned");
    return gemm_row_impl(A, lda, W, ldw, M, N, K, bias, nullptr, act, nullptr, auxh, ldaux, nullptr, 0, C32, C16, nullptr, ldc, ldc16,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1e-5f, groups, gA, gW, gB, gC, gX, stream);
}

static int gemm_row_impl(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                         const float* cscale, int act, const float* aux, const void* auxh, long ldaux, const float* resid,
                         long ldr, float* C32, void* C16, float* P32, long ldc, long ldc16, const float* ln_g0, const float* ln_b0,
                         void* ln_o0, const float* ln_g1, const float* ln_b1, void* ln_o1, float eps, int groups, long gA, long gW,
                         long gB, long gC, long gX, void* stream) {
    WC_CHECK_ARG(A && W && wc_gemm_row_supported(M, N, K), "wc_gemm_row_f16: need N <= 256, N %% 4 == 0, K = 128 | 256 (got M=%d N=%d K=%d)", M, N, K);
    WC_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ((uintptr_t)A | (uintptr_t)W) % 16 == 0,
                 "wc_gemm_row_f16: operand rows must be 16-byte aligned");
    WC_CHECK_ARG((C32 || C16 || P32 || ln_o0) && ldc >= N && ldc % 4 == 0 && (!C16 || (ldc16 >= N && ldc16 % 4 == 0)),
                 "wc_gemm_row_f16: bad output / ldc, ldc16 %% 4 != 0");
    WC_CHECK_ARG(((uintptr_t)C32 | (uintptr_t)P32 | (uintptr_t)resid | (uintptr_t)aux | (uintptr_t)bias | (uintptr_t)cscale) % 16 == 0 &&
                 ((uintptr_t)C16 | (uintptr_t)auxh | (uintptr_t)ln_o0 | (uintptr_t)ln_o1) % 8 == 0, "wc_gemm_row_f16: misaligned buffer");
    WC_CHECK_ARG(act == 0 || act == 2 || act == 5 || act == 6 || act == 7, "wc_gemm_row_f16: act must be 0, 2, 5, 6 or 7");
    WC_CHECK_ARG(act != 5 || (auxh && ldaux >= N && ldaux % 8 == 0 && N % 8 == 0 && (uintptr_t)auxh % 16 == 0 && !resid), "wc_gemm_row_f16: act 5 needs auxh, ldaux %% 4 == 0, and takes no residual (one side input per launch)");
    WC_CHECK_ARG(act != 7 || (aux && ldaux >= N && ldaux % 4 == 0 && !resid), "wc_gemm_row_f16: act 7 needs aux, ldaux %% 4 == 0, and takes no residual (one fp32 side input per launch)");
    WC_CHECK_ARG(!resid || (ldr >= N && ldr % 4 == 0), "wc_gemm_row_f16: ldr %% 4 != 0");
    WC_CHECK_ARG((!ln_o0 && !ln_o1) || (N == 256 && ln_o0 && ln_g0 && ln_b0 && (!ln_o1 || (ln_g1 && ln_b1))),
                 "wc_gemm_row_f16: fused LayerNorm needs N == 256 and gamma / beta (slot 0 first)");
    if (!g_row_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&g_row_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_row_cus <= 0)
            g_row_cus = 256;
        WC_CHECK_ARG(hipFuncSetAttribute((const void*)gemm_row_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 79872) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_row_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 79872) == hipSuccess,
                     "wc_gemm_row_f16: cannot reserve 78 KiB of LDS");
    }
    RowArgs g;
    g.A = (const __half*)A; g.W = (const __half*)W; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw;
    g.bias = bias; g.cscale = cscale; g.act = act; g.aux = aux; g.auxh = (const __half*)auxh; g.ldaux = ldaux;
    g.resid = resid; g.ldr = ldr; g.C32 = C32; g.C16 = (__half*)C16; g.P32 = P32; g.ldc = ldc; g.ldc16 = ldc16;
    g.ln_g[0] = ln_g0; g.ln_b[0] = ln_b0; g.ln_o[0] = (__half*)ln_o0;
    g.ln_g[1] = ln_g1; g.ln_b[1] = ln_b1; g.ln_o[1] = (__half*)ln_o1;
    g.eps = eps;
    g.ntiles = wc_cdiv(M, ROW_TM);
    g.groups = groups; g.gA = gA; g.gW = gW; g.gB = gB; g.gC = gC; g.gX = gX;
    // two persistent workgroups per CU (measured at 86 016 x 256 x 256, fp16 out: 35.8 / 29.8 / 34.4 / 34.9 us with 1 / 2 / 3 / 4 per CU),
    // dealt evenly over the groups
    int wpg = 2 * g_row_cus / groups;
    if (wpg < 1) wpg = 1;
    if (wpg > g.ntiles) wpg = g.ntiles;
    g.wpg = wpg;
    const int grid = wpg * groups;
    const bool erf = act == 6 || act == 7;
    const size_t lds = 3 * (size_t)ROW_TM * K * 2 + 3 * ROW_TM * 256 * 4 + 6 * 256 * 4;
    const int pr = wc_prof_begin(stream);
    const int sl = shape_log_begin(stream);
    if (K == 256) {
        if (erf) hipLaunchKernelGGL((gemm_row_kernel<4, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_row_kernel<4, false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
    } else {
        if (erf) hipLaunchKernelGGL((gemm_row_kernel<2, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_row_kernel<2, false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
    }
    shape_log_end(sl, "row", M, N, K, 1, groups, 9, act, stream);
    // algorithmic bytes: A once, the side input once, every output once
    const double nb = (double)groups * M * K * 2 + (double)groups * M * N * ((resid || act == 7 ? 4 : 0) + (act == 5 ? 2 : 0) + (C32 ? 4 : 0) + (P32 ? 4 : 0) +
                                                           (C16 ? 2 : 0) + (ln_o0 ? 2 : 0) + (ln_o1 ? 2 : 0));
    wc_prof_end2(pr, K == 256 ? (erf ? "gemm_row_kernel<4, true>" : "gemm_row_kernel<4, false>")
                              : (erf ? "gemm_row_kernel<2, true>" : "gemm_row_kernel<2, false>"), 2.0 * groups * M * N * K, nb, stream);
    WC_LAUNCH_CHECK("gemm_row_kernel");
    return WC_OK;
}
