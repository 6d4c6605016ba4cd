"""attn_pred / seg error at 512^2 against the reference fixture for every config.head_lo mask of interest, and the head's time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weclip_vit_comer_amd import synth
from weclip_vit_comer_amd import config
from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
g = np.load("tests/golden/vitb_512.npz")
sd = synth.make_clip_state_dict(seed=0, with_text=False)
bg, fg = synth.make_text_features(20, 25, 512)
fuse, dec = synth.make_head_state_dicts()
m = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[768] * 4, dataset_root_path=None, device="cuda",
           text_features=(bg.cuda(), fg.cuda()))
m.decoder_fts_fuse.load_state_dict(fuse); m.decoder.load_state_dict(dec); m.eval()
img = synth.make_images(16, 512, 512, seed=100).cuda()
labels = synth.make_label_lists(16, 2, seed=7)
i = int(g["img_index"])
for mask in (0, 31, 32, 63, 32 + 16, 32 + 16 + 4 + 8):
    config.head_lo = mask
    with torch.no_grad():
        seg, lab, ap = m(img, [""] * 16, labels=labels)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(3):
            seg, lab, ap = m(img, [""] * 16, labels=labels)
        torch.cuda.synchronize(); dt = (time.time() - t) / 3
    e_ap = np.abs(ap[i, ::64].cpu().numpy() - g["attn_pred_rows"]).max()
    e_seg = np.abs(seg[i].cpu().numpy() - g["seg"]).max() / np.abs(g["seg"]).max()
    print(f"head_lo {mask:2d}: attn_pred abs {e_ap:.2e}  seg rel {e_seg:.2e}  forward {dt*1e3:.2f} ms", flush=True)
