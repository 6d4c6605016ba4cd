"""List the GEMM shapes of one training step with their event-timed durations (sorted by total time)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import synth, ops
from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
from weclip_vit_comer_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
sd = synth.make_clip_state_dict(seed=0, with_text=False)
bg, fg = synth.make_text_features(20, 25, 512)
fuse, dec = synth.make_head_state_dicts()
model = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[768] * 4, dataset_root_path=None,
               device=dev, text_features=(bg.to(dev), fg.to(dev)))
model.decoder_fts_fuse.load_state_dict(fuse); model.decoder.load_state_dict(dec); model.train()
step = TrainStep(model)
img = synth.make_images(16, 512, 512, seed=100).to(dev)
labels = synth.make_label_lists(16, 2, seed=7)
for _ in range(3):
    step(img, labels=labels)
torch.cuda.synchronize()
rec = []
orig = ops.gemm
def traced(a, w, M, N, K, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); orig(a, w, M, N, K, **kw); e1.record()
    nseg = 1 + (getattr(a, "lo", None) is not None) + (getattr(w, "lo", None) is not None)
    rec.append(((M, N, K, nseg, kw.get("batch", 1), kw.get("act", 0), "res" if kw.get("resid") is not None else "",
                 "o32" if kw.get("out32") is not None else "o16"), e0, e1))
ops.gemm = traced
import weclip_vit_comer_amd.head_engine as he, weclip_vit_comer_amd.clip.vit_engine as ve, weclip_vit_comer_amd.gradcam_engine as ge
for m in (he, ve, ge):
    m.ops.gemm = traced
step(img, labels=labels)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in rec:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"{len(rec)} GEMM calls, {tot/1e3:.2f} ms")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, nseg, batch = k[:5]
    tf = 2.0 * M * N * K * batch * n / us / 1e6
    print(f"{str(k):60s} x{n:3d} {us/n:8.1f} us each {us/1e3:7.3f} ms  {tf:7.1f} TF/s")
