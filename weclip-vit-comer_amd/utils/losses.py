"""Live loss functions of the reference training step (utils/losses.py:11-22 get_aff_loss;
scripts/dist_clip_voc.py:105-113 get_seg_loss).  Stock PyTorch-ROCm ops."""
import torch
import torch.nn.functional as F


def get_aff_loss(inputs, targets):
    pos = targets == 1
    neg = targets == 0
    pos_count = pos.sum() + 1
    neg_count = neg.sum() + 1
    pos_loss = (pos * (1 - inputs)).sum() / pos_count
    neg_loss = (neg * inputs).sum() / neg_count
    return 0.5 * pos_loss + 0.5 * neg_loss, pos_count, neg_count


def get_seg_loss(pred, label, ignore_index=255):
    bg = label.clone()
    bg[label != 0] = ignore_index
    fg = label.clone()
    fg[label == 0] = ignore_index
    return 0.5 * (F.cross_entropy(pred, bg.long(), ignore_index=ignore_index)
                  + F.cross_entropy(pred, fg.long(), ignore_index=ignore_index))


class _SegLossFn(torch.autograd.Function):
    """get_seg_loss(F.interpolate(seg, (H,W), bilinear), label) without materialising the up-sampled
    logits (csrc/losses.hip); backward = per-pixel softmax gradient + separable bilinear backward."""

    @staticmethod
    def forward(ctx, seg, label, ignore_index):
        from .. import _lib as L
        seg = seg.float().contiguous()
        label = label.long().contiguous()
        B, nc, h, w = seg.shape
        H, W = label.shape[1:]
        sums = torch.empty(8, device=seg.device, dtype=torch.float32)
        ctx.ignore = int(ignore_index)
        ctx.one_pass = bool(ctx.needs_input_grad[0]) and nc <= 24
        if ctx.one_pass:
            # training: loss and gradient (for an upstream gradient of 1) in one pass over the pixels -- the soft-max of a
            # pixel is formed once instead of three times (csrc/losses.hip seg_loss_fused_kernel)
            nblk = ((W + 63) // 64) * ((h + 3) // 4) * B
            part = torch.empty(nblk * 4, device=seg.device, dtype=torch.float32)
            cnt = torch.empty(2048, device=seg.device, dtype=torch.float32)
            tmp = torch.empty(2 * B * nc * h * W, device=seg.device, dtype=torch.float32)
            grad = torch.empty_like(seg)
            L.lib().wc_seg_loss_fwd_bwd(L.ptr(seg, torch.float32, "seg"), L.ptr(label, torch.int64, "label"), L.ptr(cnt),
                                        L.ptr(part), L.ptr(sums), L.ptr(tmp), L.ptr(grad), B, nc, h, w, H, W, ctx.ignore,
                                        L.stream())
            ctx.save_for_backward(grad)
            return sums[4].clone()
        nblk = ((W + 63) // 64) * ((H + 3) // 4) * B
        part = torch.empty(nblk * 4, device=seg.device, dtype=torch.float32)
        L.lib().wc_seg_loss_fwd(L.ptr(seg, torch.float32, "seg"), L.ptr(label, torch.int64, "label"), L.ptr(part),
                                L.ptr(sums), B, nc, h, w, H, W, ctx.ignore, L.stream())
        ctx.save_for_backward(seg, label, sums)
        # mean over an empty set is NaN in F.cross_entropy too (0/0); the scalar tail is computed by the reduce kernel
        return sums[4].clone()

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        if ctx.one_pass:
            (grad,) = ctx.saved_tensors
            return grad * g, None, None
        seg, label, sums = ctx.saved_tensors
        B, nc, h, w = seg.shape
        H, W = label.shape[1:]
        wts = sums[5:7] * g            # (0.5 / n_bg, 0.5 / n_fg) from the forward reduce kernel: no host sync
        out = torch.empty_like(seg)
        # soft-max gradient formed inside the Y pass of the separable bilinear backward: the (B, nc, H, W) gradient
        # (352 MB at 16 x 21 x 512 x 512) is never written (csrc/losses.hip seg_loss_bwd_y_kernel)
        tmp = torch.empty(B * nc * h * W, device=seg.device, dtype=torch.float32)
        L.lib().wc_seg_loss_bwd_fused(L.ptr(seg), L.ptr(label), L.ptr(wts, torch.float32, "wts"), L.ptr(tmp), L.ptr(out), B, nc,
                                      h, w, H, W, ctx.ignore, L.stream())
        return out, None, None


def get_seg_loss_fused(seg_lowres, label, ignore_index=255):
    """== get_seg_loss(F.interpolate(seg_lowres, label.shape[1:], 'bilinear', align_corners=False), label)."""
    return _SegLossFn.apply(seg_lowres, label, ignore_index)


class _AffLossFn(torch.autograd.Function):
    """get_aff_loss(attn_pred, cams_to_affinity_label(cam_label, radius mask)) in one pass (csrc/losses.hip),
    without the (B, hw, hw) label / mask tensors."""

    @staticmethod
    def forward(ctx, attn_pred, cam_label, radius, ignore_index):
        from .. import _lib as L
        ap = attn_pred.float().contiguous()
        lab = cam_label.long().contiguous()
        B, hw, _ = ap.shape
        H, W = lab.shape[1:]
        h, w = H // 16, W // 16
        if h * w != hw:
            raise RuntimeError("attn_pred does not match the 1/16 token grid of the labels")
        part = torch.empty(4 * B * ((hw + 7) // 8), device=ap.device, dtype=torch.float32)
        sums = torch.empty(8, device=ap.device, dtype=torch.float32)
        L.lib().wc_aff_loss_fwd(L.ptr(ap, torch.float32, "attn_pred"), L.ptr(lab, torch.int64, "cam_label"), L.ptr(part),
                                L.ptr(sums), B, h, w, H, W, int(radius), int(ignore_index), L.stream())
        ctx.save_for_backward(lab, sums)
        ctx.meta = (B, h, w, H, W, int(radius), int(ignore_index))
        return sums[4].clone()

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        lab, sums = ctx.saved_tensors
        B, h, w, H, W, radius, ignore = ctx.meta
        coef = sums[5:7] * g           # (-0.5 / (n_pos + 1), 0.5 / (n_neg + 1)) from the forward reduce kernel
        dap = torch.empty(B, h * w, h * w, device=lab.device, dtype=torch.float32)
        L.lib().wc_aff_loss_bwd(L.ptr(lab), L.ptr(coef, torch.float32, "coef"), L.ptr(dap), B, h, w, H, W, radius, ignore,
                                L.stream())
        return dap, None, None, None


def get_aff_loss_fused(attn_pred, cam_label, radius=8, ignore_index=255):
    """== get_aff_loss(attn_pred, cams_to_affinity_label(cam_label, get_mask_by_radius(h, w, radius)))[0]."""
    return _AffLossFn.apply(attn_pred, cam_label, radius, ignore_index)
