"""`GradCAM` with the call contract the WeCLIP code uses (reference pytorch_grad_cam/base_cam.py:
186-199,62-114 patched for list inputs):

    cam = GradCAM(model=clip_model, target_layers=[...resblocks[-1].ln_1], reshape_transform=fn)
    grayscale_cam, probs, attn_last = cam(input_tensor=[feats (L,N,D), text (T,Ed), h, w],
                                          targets=[ClipOutputTarget(k)], target_size=None)

returns `grayscale_cam` np.float32 (N, h/16, w/16) in [0,1], the class probabilities (N,T) and the
last block's head-mean attention (N,L,L).  The gradient is analytic (gradcam_engine), so hooks on
`target_layers` are not used; `batch()` is the device-resident multi-pair entry the model uses.
"""
import numpy as np
import torch

from ..clip import vit_engine as VE
from ..gradcam_engine import last_layer_forward


class GradCAM:
    def __init__(self, model, target_layers=None, use_cuda=False, reshape_transform=None):
        self.model = model.eval()
        self.target_layers = target_layers
        self.reshape_transform = reshape_transform

    @staticmethod
    def _category(target):
        cat = getattr(target, "category", None)
        if cat is None:
            raise RuntimeError("targets must carry a `.category` (e.g. ClipOutputTarget)")
        return int(cat)

    def batch(self, state, text_hat, text_idx, n_text, pair_img, pair_cls, Tmax):
        return state.grad_cam(text_hat, text_idx, n_text, pair_img, pair_cls, Tmax)

    def __call__(self, input_tensor, targets=None, target_size=None, aug_smooth=False, eigen_smooth=False):
        if aug_smooth or eigen_smooth or target_size is not None:
            raise NotImplementedError("aug_smooth / eigen_smooth / target_size are unused by WeCLIP")
        feats, text, H, W = input_tensor
        rows, N, Lq = VE.to_rows(feats)
        st = last_layer_forward(self.model, rows, N, Lq)
        dev = rows.device
        T = text.shape[0]
        text = text.detach().float().to(dev)
        that = (text / text.norm(dim=1, keepdim=True)).contiguous()
        if targets is None:
            probs = st.class_probs(text)
            cats = probs.argmax(dim=-1).tolist()
        else:
            cats = [self._category(t) for t in targets]
        i32 = dict(dtype=torch.int32, device=dev)
        idx = torch.arange(T, **i32).repeat(N, 1).contiguous()
        cams, probs, _ = st.grad_cam(that, idx, torch.full((N,), T, **i32), torch.arange(N, **i32),
                                     torch.tensor(cats, **i32), T)
        cam = cams.view(N, H // 16, W // 16).cpu().numpy().astype(np.float32)
        return cam, probs, st.mean
