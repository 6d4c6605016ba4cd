"""Pure host (Python + launch) time of one train step: enqueue a few steps onto an idle GPU queue and time the host
side only, against the synchronised step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from weclip_vit_comer_amd import synth
from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
from weclip_vit_comer_amd.train_step import TrainStep

dev = torch.device("cuda", 0)
sd = synth.make_clip_state_dict(seed=0, with_text=False)
bg, fg = synth.make_text_features(20, 25, 512)
fuse, dec = synth.make_head_state_dicts()
model = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[768] * 4, dataset_root_path=None,
               device=dev, text_features=(bg.to(dev), fg.to(dev)))
model.decoder_fts_fuse.load_state_dict(fuse)
model.decoder.load_state_dict(dec)
model.train()
step = TrainStep(model)
img = synth.make_images(16, 512, 512, seed=100).to(dev)
labels = synth.make_label_lists(16, 2, seed=7)
for _ in range(5):
    step(img, labels=labels)
torch.cuda.synchronize()
for n in (1, 2, 3, 3, 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(img, labels=labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, synchronised {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)
