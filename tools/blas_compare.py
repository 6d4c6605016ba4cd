"""Reference point for the hand-written GEMMs: the vendor library (torch.matmul -> hipBLASLt / rocBLAS, fp16 in,
fp32 accumulate, fp16 out, no epilogue) on the encoder shapes, next to wc_gemm_f16 with the same operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops

def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in ((16400, 2304, 768), (16400, 3072, 768), (16400, 768, 3072), (16400, 768, 768), (16384, 256, 1024), (8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    wt = w.t().contiguous()
    lib_nt = t(lambda: torch.matmul(a, w.t(), out=o))          # W stored (N, K) as nn.Linear does
    lib_nn = t(lambda: torch.matmul(a, wt, out=o))
    ours = t(lambda: ops.gemm(a, w, M, N, K, out16=o))
    f = 2.0 * M * N * K / 1e6
    print(f"M={M} N={N} K={K}: library {lib_nt:7.1f} us ({f / lib_nt:6.1f} TF/s) [W^T pre-transposed {lib_nn:7.1f} us], "
          f"wc_gemm_f16 {ours:7.1f} us ({f / ours:6.1f} TF/s)", flush=True)
