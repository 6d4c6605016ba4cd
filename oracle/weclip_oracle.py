"""CPU oracle: a plain torch/numpy restatement of the WeCLIP forward/CAM hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module, and only as the checker.  The
product package (weclip-vit-comer_amd/) never imports it and has no CPU fallback.

Every function restates the arithmetic of the reference at the cited file:line
(paths relative to /root/reference) in fp32 on the CPU, following the reference's CPU
path (`clip.load(..., device='cpu')` => `model.float()`), where the only reduced-precision
steps are the forced fp16 out-projection (clip/myAtt.py:321) and the fp16-rounded resized
position embedding (clip/model.py:26).

Parity pinning: tests/golden/make_golden.py imports the unmodified reference in the build
container (oracle/refharness.py), runs it on the seeded inputs of oracle/synth.py and
stores inputs+outputs under tests/golden/*.npz; tests/test_oracle_golden.py checks this
restatement against those fixtures (function level and whole `WeCLIP.forward`).
The contour/bounding-box step (cv2) is pinned only against an OpenCV-documentation stand-in:
"unverified vs cv2" (SURVEY.md §8c).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- helpers

PAR_DILATIONS = (1, 2, 4, 8, 12, 24)
# tap order of get_kernel() (WeCLIP_model/PAR.py:10-24): TL,T,TR,L,R,BL,B,BR
PAR_DIRS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


def _ln(x, w, b, eps=1e-5):
    """clip/model.py:177-183 -- LayerNorm evaluated in fp32."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def _linear(x, w, b, mm=None):
    if mm is not None:
        return mm(x, w) + b
    return F.linear(x, w, b)


def fp16_inputs_mm(x, w):
    """Emulation of a single-pass fp16 MFMA GEMM (inputs rounded to fp16, fp32 accumulate).
    Used only to *study* precision modes on the CPU (precision section)."""
    return x.half().float() @ w.half().float().t()


# ----------------------------------------------------------------------------- A.1

def resized_pos_embed(pos, h, w):
    """upsample_pos_emb, clip/model.py:11-27: keep CLS row, bilinear (align_corners=False)
    resize of the grid rows, result rounded through fp16."""
    first, grid = pos[:1], pos[1:]
    n, d = grid.shape
    s = int(round(math.sqrt(n)))
    g = grid.t().reshape(1, d, s, s)
    g = F.interpolate(g, size=(h, w), mode="bilinear", align_corners=False)
    g = g.reshape(d, h * w).t()
    return torch.cat([first, g], 0).half().float()


def patch_tokens(img, sd, patch=16):
    """VisionTransformer.forward up to ln_pre, clip/model.py:264-273.  Returns (L, B, D)."""
    B, _, H, W = img.shape
    h, w = H // patch, W // patch
    x = F.conv2d(img.float(), sd["visual.conv1.weight"].float(), stride=patch)
    x = x.reshape(B, x.shape[1], -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"].float().expand(B, 1, -1)
    x = torch.cat([cls, x], 1) + resized_pos_embed(sd["visual.positional_embedding"].float(), h, w)
    x = _ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    return x.permute(1, 0, 2).contiguous()


# ----------------------------------------------------------------------------- A.2

def attention(a, in_w, in_b, out_w, out_b, heads, mm=None):
    """clip/myAtt.py:199-326.  a: (L, N, E) = LN1(x).  Returns (out (L,N,E) fp32 holding
    fp16-rounded values, head-mean probabilities (N, L, L))."""
    L, N, E = a.shape
    d = E // heads
    qkv = _linear(a.float(), in_w.float(), in_b.float(), mm)          # fp32 in-proj :199-201
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.contiguous().view(L, N * heads, d).transpose(0, 1)           # :257-268
    k = k.contiguous().view(L, N * heads, d).transpose(0, 1)
    v = v.contiguous().view(L, N * heads, d).transpose(0, 1)
    s = torch.bmm(q / math.sqrt(d), k.transpose(1, 2))                 # :53-56
    p = torch.softmax(s, dim=-1)                                       # :59
    o = torch.bmm(p, v)                                                # :63
    o = o.transpose(0, 1).contiguous().view(L, N, E)
    o = F.linear(o.half(), out_w.half(), out_b.half())                 # forced fp16, :321
    pm = p.view(N, heads, L, L).sum(1) / heads                         # :325-326
    return o.float(), pm


def block(x, sd, prefix, heads, mm=None, return_ln1=False, ln1_override=None):
    """ResidualAttentionBlock.forward, clip/model.py:210-214 (decoder twin
    WeCLIP_model/Decoder/TransDecoder.py:81-85)."""
    p = prefix
    a = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]) if ln1_override is None else ln1_override
    o, pm = attention(a, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                      sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, mm)
    x1 = x + o
    z = _linear(_ln(x1, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]),
                sd[p + "mlp.c_fc.weight"].float(), sd[p + "mlp.c_fc.bias"].float(), mm)
    z = z * torch.sigmoid(1.702 * z)                                   # QuickGELU :186-188
    x2 = x1 + _linear(z, sd[p + "mlp.c_proj.weight"].float(), sd[p + "mlp.c_proj.bias"].float(), mm)
    if return_ln1:
        return x2, pm, a
    return x2, pm


def n_vision_layers(sd):
    return len([k for k in sd if k.startswith("visual.") and k.endswith("attn.in_proj_weight")])


def encode_image(img, sd, heads, mm=None):
    """CLIP.encode_image(require_all_fts=True), clip/model.py:387-390,264-287,225-243:
    blocks 1..layers-1 under no_grad; returns (list of x (L,B,D), list of maps (B,L,L))."""
    layers = n_vision_layers(sd)
    with torch.no_grad():
        x = patch_tokens(img, sd, sd["visual.conv1.weight"].shape[-1])
        xs, maps = [], []
        for i in range(layers - 1):
            x, pm = block(x, sd, f"visual.transformer.resblocks.{i}.", heads, mm)
            xs.append(x)
            maps.append(pm)
    return xs, maps


# ----------------------------------------------------------------------------- A.3 / A.4

def forward_last_layer(feats, text, sd, heads, mm=None, ln1_override=None):
    """CLIP.forward_last_layer, clip/model.py:407-429.  feats (L,N,D), text (T,E).
    Returns (probs (N,T), head-mean map (N,L,L), a = LN1 output)."""
    layers = n_vision_layers(sd)
    x2, pm, a = block(feats, sd, f"visual.transformer.resblocks.{layers - 1}.", heads, mm,
                      return_ln1=True, ln1_override=ln1_override)
    x = x2.permute(1, 0, 2)
    x = _ln(x, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    y = x[:, 1:, :].mean(dim=1)
    y = y @ sd["visual.proj"].float()
    y = y / y.norm(dim=1, keepdim=True)
    t = text.float() / text.float().norm(dim=1, keepdim=True)
    logits = sd["logit_scale"].float().exp() * y @ t.t()
    return logits.softmax(dim=-1), pm, a


def scale_cam(z):
    """pytorch_grad_cam/utils/image.py:51-61 without resize."""
    z = z - z.min()
    return z / (1e-7 + z.max())


def grad_cam(feats, text, cls, sd, heads, h, w, mm=None):
    """GradCAM on block-12 ln_1 output: base_cam.py:62-154, grad_cam.py:16-23,
    activations_and_gradients.py:19-37, reshape_transform model_attn_aff_voc.py:23-30.
    feats (L,1,D).  Returns (cam (h,w) float32 numpy, probs (1,T), map (1,L,L), weights (D,))."""
    layers = n_vision_layers(sd)
    with torch.enable_grad():
        a = _ln(feats, sd[f"visual.transformer.resblocks.{layers - 1}.ln_1.weight"],
                sd[f"visual.transformer.resblocks.{layers - 1}.ln_1.bias"]).detach().requires_grad_(True)
        probs, pm, _ = forward_last_layer(feats.detach(), text, sd, heads, mm, ln1_override=a)
        (g,) = torch.autograd.grad(probs[0, cls], a)
    A = a.detach()[1:, 0, :].reshape(h, w, -1).permute(2, 0, 1).numpy()      # (D,h,w)
    G = g[1:, 0, :].reshape(h, w, -1).permute(2, 0, 1).numpy()
    wts = G.mean(axis=(1, 2))                                                # grad_cam.py:23
    cam = np.maximum((wts[:, None, None] * A).sum(0), 0).astype(np.float32)  # base_cam.py:56-60,144
    cam = scale_cam(cam)                                                     # :145
    cam = scale_cam(np.maximum(cam, 0))                                      # :150-154
    return cam.astype(np.float32), probs.detach(), pm.detach(), wts


# ----------------------------------------------------------------------------- A.5

def compute_trans_mat(w):
    """clip/clip_tool.py:64-80."""
    t = w / w.sum(0, keepdim=True)
    t = t / t.sum(1, keepdim=True)
    for _ in range(2):
        t = t / t.sum(0, keepdim=True)
        t = t / t.sum(1, keepdim=True)
    t = (t + t.t()) / 2
    return t @ t


def affinity_weight(maps12, seg_attn=None, seg_trans=False, n_last=6):
    """clip/clip_tool.py:152-175 (VOC n_last=6) / :271-293 (COCO n_last=10).
    maps12: (12, L, L) = 11 encoder maps + block-12 map for ONE image."""
    s = maps12[:, 1:, 1:]
    if not seg_trans:
        return s[-8:].mean(0)
    s = s[-n_last:]
    diff = (seg_attn[None] - s).flatten(1).sum(1)
    keep = (diff <= diff.mean()).float().view(-1, 1, 1)
    wgt = (keep * s).sum(0) / (keep.expand_as(s).sum(0) + 1e-5)
    return wgt * seg_attn


# ----------------------------------------------------------------------------- A.6

def _components8(binary):
    """8-connected components of a small boolean grid (flood fill).  Stand-in for
    cv2.findContours(RETR_TREE): the union of contour bounding boxes equals the union of
    component bounding boxes (hole contours lie inside their component's box)."""
    hh, ww = binary.shape
    lab = -np.ones((hh, ww), np.int32)
    comps = []
    for y0 in range(hh):
        for x0 in range(ww):
            if not binary[y0, x0] or lab[y0, x0] >= 0:
                continue
            cid = len(comps)
            stack = [(y0, x0)]
            lab[y0, x0] = cid
            ymin = ymax = y0
            xmin = xmax = x0
            while stack:
                y, x = stack.pop()
                ymin, ymax, xmin, xmax = min(ymin, y), max(ymax, y), min(xmin, x), max(xmax, x)
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < hh and 0 <= xx < ww and binary[yy, xx] and lab[yy, xx] < 0:
                            lab[yy, xx] = cid
                            stack.append((yy, xx))
            comps.append((xmin, ymin, xmax, ymax))
    return comps


def box_mask(cam, thr):
    """scoremap2bbox + caller's fill: clip/utils.py:115-142, clip/clip_tool.py:179-183.
    cam (h,w) float32 in [0,1].  Returns float32 (h,w) mask of 0/1."""
    hh, ww = cam.shape
    u = (cam * 255).astype(np.uint8)
    theta = int(thr * np.max(u))
    comps = _components8(u > theta)
    m = np.zeros((hh, ww), np.float32)
    for (xmin, ymin, xmax, ymax) in comps:
        x0, y0 = xmin, ymin
        x1, y1 = min(xmax + 1, ww - 1), min(ymax + 1, hh - 1)
        m[y0:y1, x0:x1] = 1
    return m


# ----------------------------------------------------------------------------- A.7

def refine_cam(trans, cam, mask):
    """clip/clip_tool.py:185-191: (T * mask_row) @ cam."""
    hh, ww = cam.shape
    t = trans * torch.from_numpy(mask).reshape(1, -1)
    return (t @ torch.from_numpy(cam).reshape(-1, 1)).reshape(hh, ww)


def upsample_cam(r, H, W):
    """generate_cam_label, clip/clip_tool.py:202-216: min-max then cv2.resize bilinear
    (half-pixel centres == interpolate(align_corners=False) for up-scaling)."""
    z = torch.from_numpy(scale_cam(r.numpy().astype(np.float32)))[None, None]
    return F.interpolate(z, size=(H, W), mode="bilinear", align_corners=False)[0, 0]


# ----------------------------------------------------------------------------- A.8

def par_affinity(img, dilations=PAR_DILATIONS, w1=0.3, w2=0.01):
    """PAR.forward up to `aff`, WeCLIP_model/PAR.py:64-88, with neighbour fetch by index
    clamping (== replicate padding + one-hot dilated conv, :39-49).  img (1,3,H,W).
    Returns aff (48,H,W)."""
    _, C, H, W = img.shape
    ys = torch.arange(H)
    xs = torch.arange(W)
    nbrs, pos = [], []
    for d in dilations:
        for (dy, dx) in PAR_DIRS:
            yy = (ys + dy * d).clamp(0, H - 1)
            xx = (xs + dx * d).clamp(0, W - 1)
            nbrs.append(img[0][:, yy][:, :, xx])
            pos.append(d * (math.sqrt(2.0) if (dy != 0 and dx != 0) else 1.0))
    nb = torch.stack(nbrs, 1)                                  # (C,48,H,W)
    std = nb.std(dim=1, keepdim=True)                          # unbiased, :77
    e = -(((nb - img[0][:, None]).abs() / (std + 1e-8) / w1) ** 2)
    e = e.mean(0)                                              # mean over channels :81
    pos = torch.tensor(pos, dtype=torch.float32)
    pe = -((pos / (pos.std() + 1e-8) / w1) ** 2)               # :78,83
    return torch.softmax(e, 0) + w2 * torch.softmax(pe, 0)[:, None, None]


def par_iterate(aff, masks, num_iter=20, dilations=PAR_DILATIONS):
    """PAR.forward loop, WeCLIP_model/PAR.py:88-92.  masks (C,H,W)."""
    _, H, W = masks.shape
    ys = torch.arange(H)
    xs = torch.arange(W)
    for _ in range(num_iter):
        acc = torch.zeros_like(masks)
        t = 0
        for d in dilations:
            for (dy, dx) in PAR_DIRS:
                yy = (ys + dy * d).clamp(0, H - 1)
                xx = (xs + dx * d).clamp(0, W - 1)
                acc = acc + aff[t] * masks[:, yy][:, :, xx]
                t += 1
        masks = acc
    return masks


def par(img, masks, num_iter=20, dilations=PAR_DILATIONS):
    """PAR.forward, WeCLIP_model/PAR.py:64-92.  img (1,3,Hi,Wi), masks (1,C,H,W)."""
    img = F.interpolate(img.float(), size=masks.shape[-2:], mode="bilinear", align_corners=True)
    aff = par_affinity(img, dilations)
    return par_iterate(aff, masks[0].float(), num_iter, dilations)[None]


# ----------------------------------------------------------------------------- A.9

def segformer_head(x_all, sd, mm=None):
    """SegFormerHead.forward in eval mode (Dropout2d = identity),
    WeCLIP_model/segformer_head.py:69-80.  x_all (11,B,768,h,w) -> (B,256,h,w)."""
    outs = []
    for i in range(x_all.shape[0]):
        x = x_all[i].float()
        n, _, hh, ww = x.shape
        t = x.flatten(2).transpose(1, 2)
        t = F.relu(_linear(t, sd[f"linears_modulelist.{i}.proj.weight"],
                           sd[f"linears_modulelist.{i}.proj.bias"], mm))
        t = _linear(t, sd[f"linears_modulelist.{i}.proj_2.weight"],
                    sd[f"linears_modulelist.{i}.proj_2.bias"], mm)
        outs.append(t.permute(0, 2, 1).reshape(n, -1, hh, ww))
    cat = torch.cat(outs, 1)
    return F.conv2d(cat, sd["linear_fuse.weight"], sd["linear_fuse.bias"])


def decoder(fts, sd, heads=8, mm=None):
    """DecoderTransformer.forward, WeCLIP_model/Decoder/TransDecoder.py:113-125."""
    b, c, hh, ww = fts.shape
    x = fts.reshape(b, c, hh * ww).permute(2, 0, 1)
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    maps = []
    for i in range(layers):
        x, pm = block(x, sd, f"transformer.resblocks.{i}.", heads, mm)
        maps.append(pm)
    x = x.permute(1, 2, 0).reshape(b, c, hh, ww)
    return F.conv2d(x, sd["linear_pred.weight"], sd["linear_pred.bias"]), maps


def attn_pred(fts):
    """WeCLIP_model/model_attn_aff_voc.py:134-137."""
    b, c, hh, ww = fts.shape
    f = fts.reshape(b, c, hh * ww)
    return torch.sigmoid(f.transpose(2, 1).bmm(f))


# ----------------------------------------------------------------------------- whole forward

def weclip_forward(img, label_lists, clip_sd, fuse_sd, dec_sd, bg_text, fg_text, heads=12,
                   dec_heads=8, seg_trans=False, dataset="voc", mm=None, return_aux=False,
                   out_size=None):
    """WeCLIP.forward, WeCLIP_model/model_attn_aff_voc.py:107-175 (COCO twin
    model_attn_aff_coco.py:100-170), with image-level labels injected (the reference reads
    them from the GT PNG, clip/clip_tool.py:111-124) and Dropout2d in eval mode.
    Returns (seg, cam_labels int64 (B,H,W), attn_pred); `seg`/`attn_pred` carry autograd
    history w.r.t. fuse_sd / dec_sd tensors that require grad."""
    B, _, H, W = img.shape
    h, w = H // 16, W // 16
    thr, n_last = (0.4, 6) if dataset == "voc" else (0.7, 10)
    xs, maps = encode_image(img, clip_sd, heads, mm)
    stack = torch.stack(xs, 0)                                          # (11,L,B,D)
    toks = stack[:, 1:].permute(0, 2, 3, 1).reshape(len(xs), B, -1, h, w)
    fts = segformer_head(toks, fuse_sd, mm)
    seg, _ = decoder(fts, dec_sd, dec_heads, mm)
    ap = attn_pred(fts)
    labels_out, aux = [], []
    for i in range(B):
        ids = list(label_lists[i])
        text = torch.cat([fg_text[ids], bg_text], 0)
        feats = xs[-1][:, i:i + 1]
        cams, refined = [], []
        trans = None
        for j, cid in enumerate(ids):
            cam, probs, pm12, _ = grad_cam(feats, text, j, clip_sd, heads, h, w, mm)
            if j == 0:
                maps12 = torch.cat([torch.stack([m[i] for m in maps], 0), pm12], 0)
                wgt = affinity_weight(maps12, ap[i].detach(), seg_trans, n_last)
                trans = compute_trans_mat(wgt.detach())
            m = box_mask(cam, thr)
            refined.append(refine_cam(trans, cam, m))
            cams.append(cam)
        oh, ow = (H, W) if out_size is None else out_size
        R = torch.stack([upsample_cam(r, oh, ow) for r in refined], 0)
        bg = (1 - R.max(0, keepdim=True)[0]) ** 1
        C = torch.cat([bg, R], 0)
        valid = torch.tensor([0] + [c + 1 for c in ids], dtype=torch.int64)
        with torch.no_grad():
            ref = par(img[i:i + 1], C[None])
            lab = valid[ref.argmax(1)][0]
        labels_out.append(lab)
        aux.append(dict(cams=np.stack(cams), probs=probs.numpy(), trans=trans.numpy(),
                        refined=torch.stack(refined).numpy(), C=C.numpy(), par=ref[0].numpy()))
    out = (seg, torch.stack(labels_out, 0), ap)
    return out + (aux,) if return_aux else out


def seg_logits(img, clip_sd, fuse_sd, dec_sd, heads=12, dec_heads=8, mm=None):
    """`WeCLIP.forward(mode='val')` of the COCO model (model_attn_aff_coco.py:100-132): encoder -> adapters ->
    decoder, returned right after the decoder (no CAM stage).  -> seg (B, nc, h, w)."""
    B, _, H, W = img.shape
    h, w = H // 16, W // 16
    xs, _ = encode_image(img, clip_sd, heads, mm)
    stack = torch.stack(xs, 0)
    toks = stack[:, 1:].permute(0, 2, 3, 1).reshape(len(xs), B, -1, h, w)
    seg, _ = decoder(segformer_head(toks, fuse_sd, mm), dec_sd, dec_heads, mm)
    return seg


def msc_flip_predict(model_fn, inputs, label_hw, scales=(1.0, 0.75), resize_long=512):
    """One image of `validate`, test_msc_flip_coco.py:52-94: long side -> resize_long (:52-57); [img, flip] at
    scale 1 (:60-68); every other scale through F.interpolate(scale_factor=s), its logits resized to the scale-1
    logit grid, un-flipped and pair-averaged (:72-86); mean over scales (:88); bilinear to the label size + arg-max
    for the plain scale-1 logits and for the multi-scale average (:90-94).
    model_fn(x (2,3,H,W)) -> seg (2,nc,h,w).  Returns (seg_pred, msc_pred) int64 (Hl, Wl)."""
    _, _, h, w = inputs.shape
    if resize_long:
        ratio = resize_long / max(h, w)
        inputs = F.interpolate(inputs, size=(int(h * ratio), int(w * ratio)), mode="bilinear", align_corners=False)
    segs_cat = model_fn(torch.cat([inputs, inputs.flip(-1)], 0))
    segs = segs_cat[0].unsqueeze(0)
    seg_list = [(segs_cat[0] + segs_cat[1].flip(-1)) / 2]
    gh, gw = segs_cat.shape[2:]
    for s in scales:
        if s != 1.0:
            x = F.interpolate(inputs, scale_factor=s, mode="bilinear", align_corners=False)
            sc = model_fn(torch.cat([x, x.flip(-1)], 0))
            sc = F.interpolate(sc, size=(gh, gw), mode="bilinear", align_corners=False)
            seg_list.append((sc[0] + sc[1].flip(-1)) / 2)
    msc = torch.mean(torch.stack(seg_list, 0), 0).unsqueeze(0)
    up = lambda t: torch.argmax(F.interpolate(t, size=tuple(label_hw), mode="bilinear", align_corners=False), dim=1)[0]
    return up(segs), up(msc)


def fast_hist(label_true, label_pred, num_classes):
    """utils/evaluate.py:10-16."""
    lt = np.asarray(label_true).reshape(-1).astype(np.int64)
    lp = np.asarray(label_pred).reshape(-1).astype(np.int64)
    keep = (lt >= 0) & (lt < num_classes)
    return np.bincount(num_classes * lt[keep] + lp[keep], minlength=num_classes ** 2).reshape(num_classes, num_classes)


def _pil_coeffs(in_size, out_size):
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` for Image.BILINEAR (what datasets/transforms.py:41 calls):
    triangle filter of support max(in/out, 1); per output coordinate the first tap, the tap count and the coefficients,
    normalised in double precision (taps accumulated in order) and rounded to 22-bit fixed point."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / fscale
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)             # C (int) cast: truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    k = np.zeros((out_size, ksize), np.float64)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):
        a = np.abs(((x + xmin).astype(np.float64) - center + 0.5) * ss)
        k[:, x] = np.where((x < n) & (a < 1.0), 1.0 - a, 0.0)
        ww = ww + k[:, x]
    k = np.where(ww[:, None] != 0.0, k / ww[:, None], k)
    return xmin, n, (0.5 + k * float(1 << 22)).astype(np.int64)


def pil_bilinear_u8(img_u8, rw, rh):
    """`Image.fromarray(img).resize([rw, rh], resample=Image.BILINEAR)` for an (H,W,C) uint8 array, restated: horizontal
    pass (fixed-point accumulate from 2^21, >> 22, clip to uint8), then the vertical pass over that uint8 image.
    Equal to real Pillow on every pixel (tests/test_oracle_golden.py checks it when PIL is importable)."""
    src = np.asarray(img_u8).astype(np.int64)
    H, W, C = src.shape
    xmin, _, kx = _pil_coeffs(W, rw)
    ymin, _, ky = _pil_coeffs(H, rh)
    tmp = np.full((H, rw, C), 1 << 21, np.int64)
    for j in range(kx.shape[1]):
        tmp += src[:, np.minimum(xmin + j, W - 1), :] * kx[None, :, j, None]     # taps beyond the count have weight 0
    tmp = np.clip(tmp >> 22, 0, 255)
    out = np.full((rh, rw, C), 1 << 21, np.int64)
    for j in range(ky.shape[1]):
        out += tmp[np.minimum(ymin + j, H - 1)] * ky[:, j, None, None]
    return np.clip(out >> 22, 0, 255).astype(np.uint8)


def augment_normalize(img_u8, scale, flip, pad_y, pad_x, crop_y, crop_x, crop, mean=(123.675, 116.28, 103.53),
                      std=(58.395, 57.12, 57.375)):
    """One image of the train-time input chain, datasets/voc.py:108-143 with the random draws given:
    random_scaling (transforms.py:26-49: PIL BILINEAR to (int(s*w), int(s*h)), uint8 result), random_fliplr (:70-84),
    random_crop (:119-176, zero canvas), normalize_img (:8-15, float32), HWC -> CHW.
    img_u8 (H,W,3) uint8 -> (3,crop,crop) f32."""
    img = np.asarray(img_u8)
    H, W, _ = img.shape
    rh, rw = int(scale * H), int(scale * W)
    x = pil_bilinear_u8(img, rw, rh).astype(np.float32)
    if flip:
        x = x[:, ::-1]
    ch, cw = max(crop, rh), max(crop, rw)
    canvas = np.zeros((ch, cw, 3), np.float32)
    canvas[pad_y:pad_y + rh, pad_x:pad_x + rw] = x
    out = canvas[crop_y:crop_y + crop, crop_x:crop_x + crop]
    out = (out - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return torch.from_numpy(np.ascontiguousarray(out.transpose(2, 0, 1)))


# ----------------------------------------------------------------------------- A.10

def radius_mask(h, w, radius=8):
    """get_mask_by_radius, scripts/dist_clip_voc.py:116-133 (Chebyshev window)."""
    yy, xx = np.divmod(np.arange(h * w), w)
    m = (np.abs(yy[:, None] - yy[None]) <= radius) & (np.abs(xx[:, None] - xx[None]) <= radius)
    return m.astype(np.float64)


def cams_to_affinity_label(cam_label, mask, ignore_index=255):
    """utils/camutils.py:226-247."""
    b, H, W = cam_label.shape
    r = F.interpolate(cam_label[:, None].float(), size=[H // 16, W // 16], mode="nearest")
    l = r.reshape(b, 1, -1)
    eq = (l.transpose(1, 2) == l).long()
    ign = (l == ignore_index)
    bad = ign | ign.transpose(1, 2) | (torch.as_tensor(mask)[None] == 0)
    eq[bad] = ignore_index
    return eq


def aff_loss(pred, target):
    """get_aff_loss, utils/losses.py:11-22."""
    pos = (target == 1)
    neg = (target == 0)
    pl = (pos * (1 - pred)).sum() / (pos.sum() + 1)
    nl = (neg * pred).sum() / (neg.sum() + 1)
    return 0.5 * pl + 0.5 * nl


def seg_loss(pred, label, ignore_index=255):
    """get_seg_loss, scripts/dist_clip_voc.py:105-113."""
    bg = label.clone()
    bg[label != 0] = ignore_index
    fg = label.clone()
    fg[label == 0] = ignore_index
    return 0.5 * (F.cross_entropy(pred, bg, ignore_index=ignore_index)
                  + F.cross_entropy(pred, fg, ignore_index=ignore_index))


def train_losses(seg, cam_labels, ap, radius=8):
    """Loss section of the step, scripts/dist_clip_voc.py:246-260."""
    H, W = cam_labels.shape[1:]
    segs = F.interpolate(seg, size=(H, W), mode="bilinear", align_corners=False)
    al = cams_to_affinity_label(cam_labels, radius_mask(H // 16, W // 16, radius))
    la = aff_loss(ap, al)
    ls = seg_loss(segs, cam_labels.long())
    return ls + 0.1 * la, ls, la
