"""`WeCLIP` (COCO) -- reference WeCLIP_model/model_attn_aff_coco.py:56-170: same pipeline as the VOC
model with CAM threshold 0.7, the last 10 attention maps in the seg-trans branch (switch at
iteration 40000), an encoder that is not frozen by name, and `mode='val'` returning
`(seg, None, attn_pred)` right after the decoder."""
import os

from .model_attn_aff_voc import WeCLIP as _VocWeCLIP, reshape_transform  # noqa: F401


class WeCLIP(_VocWeCLIP):
    cam_threshold = 0.7
    seg_trans_last = 10
    seg_trans_after = 40000
    val_runs_cam = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for p in self.encoder.parameters():     # the COCO model never freezes the encoder (:62-63);
            p.requires_grad = True              # it still only ever runs under no_grad / GradCAM
        # GT PNGs: <root>/SegmentationClass/train/<name>.png (:78, :134) -- the VOC model reads <root>/SegmentationClassAug
        root = kwargs.get("dataset_root_path", args[4] if len(args) > 4 else None)
        self.root_path = os.path.join(root, "SegmentationClass", "train") if root else None
