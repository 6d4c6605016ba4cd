import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from weclip_vit_comer_amd import ops, _lib as L
lib = L.lib().cdll
lib.wc_gemm_set_p192.argtypes = [ctypes.c_int, ctypes.c_float]; lib.wc_gemm_set_p192.restype = None
M, N, K = 16400, 768, 768
g = torch.Generator().manual_seed(1)
a = torch.randn(M, K, generator=g).half().cuda(); w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
outs = {}
for mode in (0, 2):
    lib.wc_gemm_set_p192(mode, 0.0)
    o = torch.zeros(M, N, device="cuda")
    ops.gemm(a, w, M, N, K, out32=o)
    torch.cuda.synchronize()
    outs[mode] = o
d = (outs[0] - outs[2]).abs()
ref = a.double() @ w.double().t()
print("max diff", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel())
print("err mode0", (outs[0].double() - ref).abs().max().item(), "mode2", (outs[2].double() - ref).abs().max().item())
nz = (d > 0).nonzero()
print("rows mod 256 hist:", torch.bincount(nz[:, 0] % 256 // 32, minlength=8).tolist())
print("cols mod 192 hist (32-col tiles):", torch.bincount(nz[:, 1] % 192 // 32, minlength=6).tolist())
print("cols hist by 32:", torch.bincount(nz[:, 1] // 32, minlength=24).tolist())
print("row tile hist (first 10):", torch.bincount(nz[:, 0] // 256, minlength=65).tolist()[:10])
