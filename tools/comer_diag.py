"""Engine vs module-by-module form vs an fp64 CPU evaluation of the same network (deformable attention by grid_sample)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import comer_oracle as CO
from weclip_vit_comer_amd.WeCLIP_model import comer as CM

B, H, W, dim = 2, 128, 160, 256
h, w = H // 16, W // 16
torch.manual_seed(0)
net = CM.CoMerInteraction(dim)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for t in net.cti:
        t.gamma.copy_(torch.randn(dim, generator=g) * 0.5)
        for a in (t.to_v, t.to_c):
            a.sampling_offsets.weight.copy_(torch.randn(a.sampling_offsets.weight.shape, generator=g) * 0.02)
            a.attention_weights.weight.copy_(torch.randn(a.attention_weights.weight.shape, generator=g) * 0.05)
            a.attention_weights.bias.copy_(torch.randn(a.attention_weights.bias.shape, generator=g) * 0.2)
img = torch.randn(B, 3, H, W, generator=g)
maps0 = [torch.randn(B, h * w, dim, generator=g) for _ in range(11)]
gy = torch.randn(B, dim, h, w, generator=g)

# fp64 CPU reference
ref = copy.deepcopy(net).double()
orig = CM.ms_deform_attn_core
CM.ms_deform_attn_core = lambda value, shapes, loc, attn: CO.ms_deform_attn(value, shapes, loc, attn)
maps = [m.double().requires_grad_(True) for m in maps0]
y = ref(img.double(), maps, (h, w))
y.backward(gy.double())
CM.ms_deform_attn_core = orig
R = (y.detach(), [maps[b].grad for b in net.stage_blocks], {n: p.grad for n, p in ref.named_parameters()})

net = net.cuda()
res = {}
for mode in ("0", "1"):
    os.environ["WECLIP_COMER_ENGINE"] = mode
    for p in net.parameters():
        p.grad = None
    maps = [m.cuda().requires_grad_(True) for m in maps0]
    y = net(img.cuda(), maps, (h, w))
    y.backward(gy.cuda())
    res[mode] = (y.detach().cpu().double(), [maps[b].grad.cpu().double() for b in net.stage_blocks],
                 {n: p.grad.cpu().double() for n, p in net.named_parameters()})
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
for mode, name in (("0", "module form"), ("1", "engine")):
    r = res[mode]
    errs = {n: rel(r[2][n], R[2][n]) for n in R[2]}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print(f"{name:12s} vs fp64: y {rel(r[0], R[0]):.1e}  d(maps) {max(rel(a, b) for a, b in zip(r[1], R[1])):.1e}  worst param grads:")
    for n, v in worst:
        print(f"      {n:45s} {v:.2e}   |ref| max {R[2][n].abs().max().item():.2e}")
