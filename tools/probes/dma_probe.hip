// L2 -> LDS transport probe: what one CU can pull through global_load_lds_dwordx4 with the GEMM's access pattern and
// with two alternatives, nothing else running on the CU (no fragment reads, no MFMA).
//   pattern 0: the 256x256 GEMM's half-tile: a wave instruction = 8 rows x 128 B (row stride = lda)
//   pattern 1: 4 rows x 256 B per instruction (a BK = 128 layout)
//   pattern 2: 16 rows x 64 B per instruction
// One 512-thread workgroup per CU, each streams the A rows of "its tile" over K like the GEMM does; depth = DMA
// instructions kept in flight per wave.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int PAT, int DEPTH, int READS = 0>
__global__ __launch_bounds__(512) void probe(const __half* __restrict__ A, long lda, int M, int K, int iters, int gx, int rot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = (blockIdx.x / gx) * 256 % M;
    // per-lane source: which row / 16-B chunk this lane fetches inside one instruction's 1 KiB
    int row, ch;
    if (PAT == 0) { row = lane >> 3; ch = lane & 7; }            // 8 rows x 8 chunks
    else if (PAT == 1) { row = lane >> 4; ch = lane & 15; }      // 4 rows x 16 chunks
    else { row = lane >> 2; ch = lane & 3; }                     // 16 rows x 4 chunks
    const int rows_per_inst = PAT == 0 ? 8 : (PAT == 1 ? 4 : 16);
    const int bytes_per_row = PAT == 0 ? 128 : (PAT == 1 ? 256 : 64);
    // a "K-step" moves 64 KiB per workgroup = 8 instructions per wave
    const int ksteps = K * 2 / bytes_per_row;                    // K-steps until the row is exhausted
    for (int it = 0; it < iters; ++it) {
        for (int ks0 = 0; ks0 < ksteps; ++ks0) {
            // rot: the gx workgroups that share a row block start at different K offsets (wrap around)
            const int ks = rot ? (ks0 + (int)(blockIdx.x % gx) * rot) % ksteps : ks0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (m0 + (wave * 8 + i) * rows_per_inst + row) % M;
                const __half* src = A + (long)r * lda + (long)ks * (bytes_per_row / 2) + ch * 8;
                char* dst = smem + ((ks & 1) * 8 + i) * 8192 + wave * 1024;
                __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)dst, 16, 0, 0);
                if (READS) {          // fragment-read load beside the DMA: READS x ds_read_b128 per DMA instruction
                    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                    u4 acc = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < READS; ++r) {
                        const u4 v = *reinterpret_cast<const u4*>(smem + (((ks + 1) & 1) * 8 + ((i + r) & 7)) * 8192 + ((lane * 16 + wave * 1024 + r * 2048) & 8191));
                        acc += v;
                    }
                    asm volatile("" :: "v"(acc));
                }
                if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (DEPTH == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (DEPTH == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PAT, int DEPTH, int READS = 0>
static void run(const __half* A, long lda, int M, int K, int gx, int rot = 0) {
    const int iters = 20;
    hipFuncSetAttribute((const void*)probe<PAT, DEPTH, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<PAT, DEPTH, READS><<<256, 512, 131072>>>(A, lda, M, K, 2, gx, rot);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<PAT, DEPTH, READS><<<256, 512, 131072>>>(A, lda, M, K, iters, gx, rot);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * (double)(K * 2 / (PAT == 0 ? 128 : (PAT == 1 ? 256 : 64))) * 65536.0;
    printf("pattern %d depth %2d reads/DMA %d gx %d rot %d K %4d: %7.1f GB/s per CU  (%5.2f TB/s chip)  %.1f us\n", PAT, DEPTH, READS, gx, rot, K,
           bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
    const int M = 16384;
    for (int K : {768, 3072}) {
        __half* A;
        hipMalloc(&A, (size_t)M * K * 2);
        hipMemset(A, 0x11, (size_t)M * K * 2);
        run<0, 6, 0>(A, K, M, K, 1); run<0, 6, 1>(A, K, M, K, 1); run<0, 6, 3>(A, K, M, K, 1); run<0, 6, 6>(A, K, M, K, 1);
        run<0, 12, 3>(A, K, M, K, 1);
        hipFree(A);
    }
    return 0;
}
