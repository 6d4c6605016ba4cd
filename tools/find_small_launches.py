"""Which Python lines launch the small torch kernels (fills, copies, elementwise) inside one train step:
a TorchDispatchMode over one step of the bench workload logs every aten op with the innermost repo frame."""
import collections, os, sys
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
               device=dev, text_features=(bg.to(dev), fg.to(dev)), comer=len(sys.argv) > 1 and sys.argv[1] == "comer")
model.decoder_fts_fuse.load_state_dict(fuse)
model.decoder.load_state_dict(dec)
model.train()
step = TrainStep(model)
img = synth.make_images(16, 512, 512, seed=100).to(dev)
labels = synth.make_label_lists(16, 2, seed=7)
for _ in range(4):
    step(img, labels=labels)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.Counter()
SKIP = ("view", "reshape", "detach", "alias", "as_strided", "expand", "permute", "transpose", "t.default", "select",
        "slice", "unsqueeze", "squeeze", "empty", "_unsafe_view", "unbind", "split", "lift_fresh", "is_pinned", "stride",
        "sym_", "size", "numel", "_local_scalar_dense")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if root in f.filename and "tools/" not in f.filename]
            where = f"{fr[-1].filename.replace(root + '/', '')}:{fr[-1].lineno}" if fr else "?"
            agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    step(img, labels=labels)
torch.cuda.synchronize()
for k, n in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:4d} x  {k[0]:40s} {k[1]}")
