"""Host-side view of one training step: device->host synchronisations (torch sync debug mode) and the
host time to enqueue a step vs the GPU time to run it."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import synth
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
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step(img, labels=labels)
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))
for x in w[:20]:
    print("  ", x.filename.split("/")[-1], x.lineno, str(x.message)[:80])
torch.cuda.synchronize()
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step(img, labels=labels)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
