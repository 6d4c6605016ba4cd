// In-kernel shader clock under a dense fp16 MFMA load (MI355X guide, "DVFS give-back" item 6): delta(s_memtime) /
// delta(s_memrealtime) * 100 MHz around a loop of v_mfma_f32_32x32x16_f16 on random operands, 8 waves per CU (2 per
// SIMD), every CU busy, after ~2 s of back-to-back launches.  Also prints the MFMA rate the loop sustained.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 65535]; b[i] = in[(tid * 8 + 4 + i) & 65535]; }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + 1) & 3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 1) & 3], b[k], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 2) & 3], b[(k + 3) & 3], acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    f16x8* in; float* out; unsigned long long* st;
    hipMalloc(&in, 65536 * sizeof(f16x8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;                      // 16 MFMAs per iteration per wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 60; ++rep) mfma_loop<<<256, 512>>>(in, out, st, iters);     // ~2 s of load first
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<<<256, 512>>>(in, out, st, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(512);
    hipMemcpy(hs.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int i = 0; i < 256; ++i) clk.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 100e6);
    std::sort(clk.begin(), clk.end());
    const double flops = 256.0 * 8 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("dense fp16 MFMA loop (2 waves/SIMD, random operands): %.1f TFLOP/s; in-kernel clock median %.3f GHz (min %.3f, max %.3f); "
           "cycles per MFMA per SIMD %.1f\n", flops / (ms * 1e-3) / 1e12, clk[128] / 1e9, clk[0] / 1e9, clk[255] / 1e9,
           (double)hs[0] / (iters * 16.0 * 2));
    return 0;
}
