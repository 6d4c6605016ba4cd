"""`scoremap2bbox` (reference clip/utils.py:115-142) on the device: 8-connected components of
the thresholded u8 CAM and their clamped bounding boxes (csrc/affinity.hip box_mask_kernel)."""
import numpy as np
import torch

from .. import cam_pipeline as CP


def scoremap2bbox(scoremap, threshold, multi_contour_eval=False):
    """scoremap (h, w) float in [0,1] -> (boxes int array (n,4) [x0,y0,x1,y1], n).
    No component: ([[0,0,0,0]], 1) like the reference."""
    sm = torch.as_tensor(np.ascontiguousarray(scoremap, dtype=np.float32))
    h, w = sm.shape
    i32 = dict(dtype=torch.int32, device="cuda")
    _, _, boxes, nbox = CP.box_masks(sm.reshape(1, -1).cuda(), torch.zeros(1, **i32), torch.zeros(1, **i32),
                                     1, 1, h, w, threshold, want_boxes=True)
    n = int(nbox.item())
    if n == 0:
        return np.asarray([[0, 0, 0, 0]]), 1
    b = boxes[0, :min(n, 64)].cpu().numpy()
    if not multi_contour_eval:
        areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        b = b[[int(np.argmax(areas))]]
    return b, len(b)
