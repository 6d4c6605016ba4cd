"""Generate the golden fixtures under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

Build-container only (needs /root/reference; see oracle/refharness.py for how it is
imported on CPU without its missing third-party packages).  Re-run with
    python tests/golden/make_golden.py
Outputs (committed, data only -- no reference source):
  tiny_func.npz    function-level outputs of the reference on the tiny config
                   (encode_image, GradCAM, compute_trans_mat, PAR, SegFormerHead,
                   DecoderTransformer)
  tiny_voc.npz     whole `WeCLIP.forward` (VOC model) + losses + backward, normal branch
  tiny_voc_seg.npz same with the seg-trans branch (iter_num > 15000)
  vitb_224.npz     BASELINE config 0: ViT-B/16-sized synthetic weights, one 224x224 image,
                   encode + GradCAM(2 classes) + transition matrix + refinement
  tiny_coco_msc.npz the reference's own `validate` (test_msc_flip_coco.py:33-121, executed from the script's source
                   without importing the script) on a tiny 81-class COCO-headed model and three synthetic images of
                   different sizes at scales (1, 0.75): per-image predictions, both confusion histograms, scores
  vitb_512.npz     the benchmark size: whole `WeCLIP.forward` (VOC model, ViT-B/16-sized synthetic
  vitb_512_seg.npz weights) on image 3 of bench.py's B=16 512x512 batch, normal / seg-trans branch;
  vitb_512_seg_sink.npz  the seg-trans branch once more with per-block CLS attention sinks (synth.SINK_512) so that
                   the discrete layer selection is decided by the data, not by fp32 rounding;
                   every stage of the CAM chain recorded by wrapping (not editing) the reference's own
                   functions: class probabilities, CAM maps, affinity, T rows, refined CAMs, PAR rows, labels
  vitb_320.npz     the same whole forward at the REFERENCE'S OWN default geometry (crop 320, batch 4: scripts/dist_clip_voc.py:34,
                   configs/voc_attn_reg.yaml:5): image 1 of a 4 x 320 x 320 batch (L = 401 tokens: neither a multiple of the
                   attention tiles nor the CLS-only remainder of 512 x 512)
  vitb_512_train.npz  the benchmark-size forward once more followed by the reference's losses and `loss.backward()`: loss
                   values, the norm of every adapter / decoder gradient, ten gradient tensors in full
  augment_ref.npz  the reference's train-time input chain (datasets/transforms.py: PIL BILINEAR rescale, flip, zero-pad
                   + crop, normalise) on six small uint8 images, up- and down-scaling, with every random draw recorded
Inputs are regenerated from oracle/synth.py seeds; each fixture stores a checksum of the
weights and the image so a drifting RNG is detected instead of silently mis-compared.
"""
import ast
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refharness, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

TINY, TINY_HW, TINY_LABELS, checksum = synth.TINY, synth.TINY_HW, synth.TINY_LABELS, synth.checksum


def _script_fn(name):
    """Pull one function out of scripts/dist_clip_voc.py without importing the script
    (it needs omegaconf/tensorboard at import time)."""
    src = open(os.path.join(refharness.REF, "scripts", "dist_clip_voc.py")).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            ns = {"np": np, "torch": torch, "F": torch.nn.functional}
            exec(compile(ast.Module([node], []), "dist_clip_voc.py", "exec"), ns)
            return ns[name]
    raise KeyError(name)


def make_tiny_func():
    from clip.clip_tool import ClipOutputTarget, compute_trans_mat
    from pytorch_grad_cam import GradCAM
    from WeCLIP_model.model_attn_aff_voc import reshape_transform
    from WeCLIP_model.PAR import PAR
    from WeCLIP_model.segformer_head import SegFormerHead
    from WeCLIP_model.Decoder.TransDecoder import DecoderTransformer

    sd = synth.make_clip_state_dict(**TINY)
    m = refharness.build_clip(sd)
    H, W = TINY_HW
    img = synth.make_images(2, H, W)
    with torch.no_grad():
        fts, attns = m.encode_image(img, H, W, require_all_fts=True)
    for name, p in m.named_parameters():
        p.requires_grad = "11" in name
    cam = GradCAM(model=m, target_layers=[m.visual.transformer.resblocks[-1].ln_1],
                  reshape_transform=reshape_transform)
    bg, fg = synth.make_text_features(20, 25, TINY["embed_dim"])
    out = dict(weights_ck=checksum(sd.values()), img_ck=checksum([img]),
               fts_last=fts[-1].numpy(), fts_first=fts[0].numpy(),
               attn=torch.stack(attns, 0).numpy())
    cams, probs, last = [], [], []
    for i, ids in enumerate(TINY_LABELS):
        text = torch.cat([fg[ids], bg], 0)
        for j in range(len(ids)):
            g, p, a = cam(input_tensor=[fts[-1][:, i:i + 1], text, H, W],
                          targets=[ClipOutputTarget(j)], target_size=None)
            cams.append(g[0])
            probs.append(p.detach().numpy()[0])
            last.append(a.detach().numpy()[0])
    out.update(cams=np.stack(cams), probs=np.stack(probs), attn_last=np.stack(last))
    g = torch.Generator().manual_seed(11)
    wgt = torch.rand(24, 24, generator=g) + 0.05
    out.update(trans_in=wgt.numpy(), trans_out=compute_trans_mat(wgt).numpy())
    par = PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])
    masks = torch.rand(1, 3, H, W, generator=g)
    out.update(par_masks=masks.numpy(), par_out=par(img[:1], masks).numpy())
    par2 = PAR(num_iter=3, dilations=[1, 2, 4, 8, 12, 24])   # odd size, 5 channels
    img2 = synth.make_images(1, 37, 53, seed=5)
    masks2 = torch.rand(1, 5, 37, 53, generator=g)
    out.update(par2_masks=masks2.numpy(), par2_out=par2(img2, masks2).numpy())
    fuse_sd, dec_sd = synth.make_head_state_dicts(width=TINY["width"])
    head = SegFormerHead(in_channels=[TINY["width"]] * 4, embedding_dim=256, num_classes=21, index=11).eval()
    head.load_state_dict(fuse_sd)
    dec = DecoderTransformer(width=256, layers=3, heads=8, output_dim=21).eval()
    dec.load_state_dict(dec_sd)
    toks = torch.stack(fts, 0)[:, 1:].permute(0, 2, 3, 1).reshape(11, 2, -1, H // 16, W // 16)
    with torch.no_grad():
        f = head(toks)
        seg, dmaps = dec(f)
    out.update(head_out=f.numpy(), dec_out=seg.numpy(), dec_map0=dmaps[0].numpy())
    np.savez_compressed(os.path.join(OUT, "tiny_func.npz"), **out)
    print("tiny_func.npz written")


def make_tiny_whole(seg_trans):
    from PIL import Image
    from WeCLIP_model.model_attn_aff_voc import WeCLIP
    from utils.camutils import cams_to_affinity_label
    from utils.losses import get_aff_loss
    get_seg_loss = _script_fn("get_seg_loss")
    get_mask_by_radius = _script_fn("get_mask_by_radius")

    sd = synth.make_clip_state_dict(**TINY)
    H, W = TINY_HW
    img = synth.make_images(2, H, W)
    bg, fg = synth.make_text_features(20, 25, TINY["embed_dim"])
    fuse_sd, dec_sd = synth.make_head_state_dicts(width=TINY["width"])
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, "clip_tiny.pt")
        # clip.load: torch.jit.load fails -> torch.load state-dict branch (clip/clip.py:128-143)
        torch.save(sd, ck)
        os.makedirs(os.path.join(tmp, "SegmentationClassAug"))
        names = []
        for i, ids in enumerate(TINY_LABELS):
            png = np.zeros((H, W), np.uint8)
            for j, c in enumerate(ids):
                png[4 + 8 * j: 12 + 8 * j, 4:20] = c + 1
            png[-3:, -3:] = 255
            Image.fromarray(png).save(os.path.join(tmp, "SegmentationClassAug", f"im{i}.png"))
            names.append(f"im{i}")
        model = WeCLIP(num_classes=21, clip_model=ck, embedding_dim=256,
                       in_channels=[TINY["width"]] * 4, dataset_root_path=tmp, device="cpu")
        model.bg_text_features, model.fg_text_features = bg, fg
        model.decoder_fts_fuse.load_state_dict(fuse_sd)
        model.decoder.load_state_dict(dec_sd)
        model.eval()
        if seg_trans:
            model.iter_num = 20000
        seg, cam_labels, ap = model(img, names)
    segs = torch.nn.functional.interpolate(seg, size=cam_labels.shape[1:], mode="bilinear",
                                           align_corners=False)
    mask = get_mask_by_radius(h=H // 16, w=W // 16, radius=8)
    aff_label = cams_to_affinity_label(cam_labels.clone(), mask=torch.from_numpy(mask), ignore_index=255)
    attn_loss, _, _ = get_aff_loss(ap, aff_label)
    seg_loss = get_seg_loss(segs, cam_labels.type(torch.long), ignore_index=255)
    loss = seg_loss + 0.1 * attn_loss
    loss.backward()
    grads = {}
    for n, p in list(model.decoder.named_parameters()) + list(model.decoder_fts_fuse.named_parameters()):
        grads[n] = p.grad
    keep = ["linear_pred.weight", "linear_pred.bias", "transformer.resblocks.2.attn.in_proj_bias",
            "transformer.resblocks.0.ln_1.weight", "transformer.resblocks.0.mlp.c_fc.bias",
            "linears_modulelist.10.proj_2.bias", "linears_modulelist.0.proj.bias", "linear_fuse.bias"]
    out = dict(weights_ck=checksum(sd.values()), img_ck=checksum([img]),
               seg=seg.detach().numpy(), cam_labels=cam_labels.numpy().astype(np.uint8),
               attn_pred=ap.detach().numpy(), aff_label=aff_label.numpy().astype(np.uint8),
               attn_loss=np.float32(attn_loss.item()), seg_loss=np.float32(seg_loss.item()),
               grad_norms=np.array([float(grads[n].norm()) for n in sorted(grads)], np.float64),
               grad_names=np.array(sorted(grads)))
    for n in keep:
        out["grad:" + n] = grads[n].numpy()
    fn = "tiny_voc_seg.npz" if seg_trans else "tiny_voc.npz"
    np.savez_compressed(os.path.join(OUT, fn), **out)
    print(fn, "written; labels present:", np.unique(out["cam_labels"]))


def make_vitb_224():
    from clip.clip_tool import ClipOutputTarget, compute_trans_mat
    from clip.utils import scoremap2bbox
    from pytorch_grad_cam import GradCAM
    from WeCLIP_model.model_attn_aff_voc import reshape_transform

    sd = synth.make_clip_state_dict(seed=0, with_text=True)
    m = refharness.build_clip(sd)
    H = W = 224
    img = synth.make_images(1, H, W, seed=100)
    with torch.no_grad():
        fts, attns = m.encode_image(img, H, W, require_all_fts=True)
    for name, p in m.named_parameters():
        p.requires_grad = "11" in name
    cam = GradCAM(model=m, target_layers=[m.visual.transformer.resblocks[-1].ln_1],
                  reshape_transform=reshape_transform)
    bg, fg = synth.make_text_features(20, 25, 512)
    ids = [0, 1]
    text = torch.cat([fg[ids], bg], 0)
    cams, probs, refined = [], None, []
    for j in range(2):
        g, p, a = cam(input_tensor=[fts[-1], text, H, W], targets=[ClipOutputTarget(j)], target_size=None)
        cams.append(g[0])
        probs = p.detach().numpy()
        if j == 0:
            aw = torch.cat([torch.stack(attns, 0)[:, 0], a.detach()], 0)[:, 1:, 1:][-8:].mean(0)
            T = compute_trans_mat(aw)
        box, cnt = scoremap2bbox(scoremap=g[0], threshold=0.4, multi_contour_eval=True)
        mask = torch.zeros(14, 14)
        for i_ in range(cnt):
            x0, y0, x1, y1 = box[i_]
            mask[y0:y1, x0:x1] = 1
        refined.append((T * mask.view(1, -1)) @ torch.from_numpy(g[0]).view(-1, 1))
    out = dict(weights_ck=checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]),
               img_ck=checksum([img]),
               fts_last=fts[-1][:, 0].numpy().astype(np.float32),
               fts5_cls=fts[5][0, 0].numpy(), attn10_rows=attns[10][0, ::16].numpy(),
               attn_last_rows=a.detach()[0, ::16].numpy(),
               probs=probs, cams=np.stack(cams), trans_diag=T.diagonal().numpy(),
               trans_rows=T[::28].numpy(), refined=torch.stack(refined).reshape(2, 14, 14).numpy())
    np.savez_compressed(os.path.join(OUT, "vitb_224.npz"), **out)
    print("vitb_224.npz written; probs[:3] =", probs[0, :3])


BENCH_IMG = 3          # image of bench.py's batch (synth.make_images(16, 512, 512, seed=100)) the 512x512 fixture is made of


def make_vitb_512(seg_trans, sink=None, size=512, batch=16, index=None, seed=100, label_seed=7, fn_override=None):
    """Whole reference `WeCLIP.forward` at the benchmark size on ONE image of the benchmark batch.  The
    intermediate stages are recorded by wrapping the reference's own callables at generation time
    (GradCAM.__call__, compute_trans_mat, perform_single_voc_cam, PAR.forward); nothing is edited."""
    from PIL import Image
    import clip.clip_tool as ref_ct
    import WeCLIP_model.model_attn_aff_voc as ref_voc
    from pytorch_grad_cam.base_cam import BaseCAM

    H = W = size
    IDX = BENCH_IMG if index is None else index
    sd = synth.make_clip_state_dict(seed=0, with_text=True, cls_sink=sink)
    img = synth.make_images(batch, H, W, seed=seed)[IDX:IDX + 1].contiguous()
    ids = synth.make_label_lists(batch, 2, seed=label_seed)[IDX]
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse_sd, dec_sd = synth.make_head_state_dicts()
    rec = {"cam": [], "probs": [], "attn_last": [], "trans_in": [], "trans_out": [], "refined": [], "par_in": [], "par_out": []}

    orig_call, orig_tm, orig_single = BaseCAM.__call__, ref_ct.compute_trans_mat, ref_voc.perform_single_voc_cam

    def call(self, *a, **k):
        out = orig_call(self, *a, **k)
        rec["cam"].append(np.array(out[0][0])); rec["probs"].append(out[1].detach().numpy()[0].copy())
        rec["attn_last"].append(out[2].detach().numpy()[0].copy())
        return out

    def tm(x):
        out = orig_tm(x)
        rec["trans_in"].append(x.detach().numpy().copy()); rec["trans_out"].append(out.detach().numpy().copy())
        return out

    def single(*a, **k):
        out = orig_single(*a, **k)
        rec["refined"].append(torch.stack([c.detach() for c in out[0]]).numpy())
        return out

    BaseCAM.__call__, ref_ct.compute_trans_mat, ref_voc.perform_single_voc_cam = call, tm, single
    try:
        with tempfile.TemporaryDirectory() as tmp:
            ck = os.path.join(tmp, "clip_vitb.pt")
            torch.save(sd, ck)
            os.makedirs(os.path.join(tmp, "SegmentationClassAug"))
            png = np.zeros((H, W), np.uint8)
            for j, c in enumerate(ids):
                png[32 + 64 * j: 96 + 64 * j, 32:160] = c + 1
            png[-3:, -3:] = 255
            Image.fromarray(png).save(os.path.join(tmp, "SegmentationClassAug", "im.png"))
            model = ref_voc.WeCLIP(num_classes=21, clip_model=ck, embedding_dim=256, in_channels=[768] * 4,
                                   dataset_root_path=tmp, device="cpu")
            model.bg_text_features, model.fg_text_features = bg, fg
            model.decoder_fts_fuse.load_state_dict(fuse_sd)
            model.decoder.load_state_dict(dec_sd)
            model.eval()
            model.par.register_forward_hook(lambda m, i, o: (rec["par_in"].append(i[1].detach().numpy().copy()),
                                                             rec["par_out"].append(o.detach().numpy().copy())) and None)
            if seg_trans:
                model.iter_num = 20000
            with torch.no_grad():
                fts, attns = model.encoder.encode_image(img, H, W, require_all_fts=True)
            seg, cam_labels, ap = model(img, ["im"])
    finally:
        BaseCAM.__call__, ref_ct.compute_trans_mat, ref_voc.perform_single_voc_cam = orig_call, orig_tm, orig_single
    T = rec["trans_out"][0]
    Wa = rec["trans_in"][0]
    # the seg-trans layer selection, recomputed with the reference's own expressions (clip_tool.py:152-162) from the
    # tensors it was given, plus the same sums in fp64: at |diff| ~ 1e6 the fp32 sums are quantised to 1/8
    aw = torch.cat([torch.stack(attns, 0)[:, 0], torch.from_numpy(rec["attn_last"][0])[None]], 0)[:, 1:, 1:][-6:]
    seg_attn = ap.detach()[0:1]
    attn_diff = torch.sum((seg_attn - aw).flatten(1), dim=1)
    keep_ref = attn_diff <= torch.mean(attn_diff)
    A64 = aw.double().flatten(1).sum(1)
    out = dict(keep_ref=keep_ref.numpy(), diff_ref=attn_diff.numpy(), A64=A64.numpy(), weights_ck=checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]), img_ck=checksum([img]),
               img_index=np.int64(IDX), ids=np.array(ids),
               fts_last_rows=fts[-1][::64, 0].numpy().astype(np.float32),       # (17, 768): every 64th token of block 11
               fts5_rows=fts[5][::128, 0].numpy().astype(np.float32),
               attn10_rows=attns[10][0, ::128].numpy(), attn_last_rows=rec["attn_last"][0][::128],
               probs=np.stack(rec["probs"]), cams=np.stack(rec["cam"]),
               aff_rows=Wa[::64].astype(np.float32), aff_rowsum=Wa.sum(1).astype(np.float32),
               trans_rows=T[::64].astype(np.float32), trans_diag=np.diagonal(T).astype(np.float32),
               trans_rowsum=T.sum(1).astype(np.float32),
               refined=rec["refined"][0].astype(np.float32),
               par_in_rows=rec["par_in"][0][0][:, ::16].astype(np.float32),      # (3, 32, 512): every 16th pixel row
               par_out_rows=rec["par_out"][0][0][:, ::16].astype(np.float32),
               cam_labels=cam_labels[0].numpy().astype(np.uint8),
               seg=seg[0].detach().numpy().astype(np.float32), attn_pred_rows=ap[0, ::64].detach().numpy().astype(np.float32))
    fn = "vitb_512_seg.npz" if seg_trans else "vitb_512.npz"
    if sink is not None:
        assert seg_trans
        fn = "vitb_512_seg_sink.npz"
        out["cls_sink"] = np.array(sink, np.float64)
    if fn_override:
        fn = fn_override
        out.update(size=np.int64(size), batch=np.int64(batch), seed=np.int64(seed), label_seed=np.int64(label_seed))
    np.savez_compressed(os.path.join(OUT, fn), **out)
    print(fn, "written; probs:", out["probs"][:, :2], "labels:", np.unique(out["cam_labels"], return_counts=True),
          "layer selection:", out["keep_ref"].astype(int), "A_l:", np.round(out["A64"], 3))


def make_vitb_512_train():
    """BASELINE configs[2] at the benchmark resolution, trainable half: the reference's `WeCLIP.forward` (ViT-B/16-sized
    synthetic weights, image 3 of bench.py's batch, 512x512) followed by its losses (scripts/dist_clip_voc.py:250-260:
    bilinear up-sampling, get_seg_loss, cams_to_affinity_label + get_aff_loss) and `loss.backward()`: both loss values, the
    norm of every adapter / decoder gradient and a few gradient tensors in full."""
    from PIL import Image
    import WeCLIP_model.model_attn_aff_voc as ref_voc
    from utils.camutils import cams_to_affinity_label
    from utils.losses import get_aff_loss
    get_seg_loss = _script_fn("get_seg_loss")
    get_mask_by_radius = _script_fn("get_mask_by_radius")
    H = W = 512
    sd = synth.make_clip_state_dict(seed=0, with_text=True)
    img = synth.make_images(16, H, W, seed=100)[BENCH_IMG:BENCH_IMG + 1].contiguous()
    ids = synth.make_label_lists(16, 2, seed=7)[BENCH_IMG]
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse_sd, dec_sd = synth.make_head_state_dicts()
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, "clip_vitb.pt")
        torch.save(sd, ck)
        os.makedirs(os.path.join(tmp, "SegmentationClassAug"))
        png = np.zeros((H, W), np.uint8)
        for j, c in enumerate(ids):
            png[32 + 64 * j: 96 + 64 * j, 32:160] = c + 1
        png[-3:, -3:] = 255
        Image.fromarray(png).save(os.path.join(tmp, "SegmentationClassAug", "im.png"))
        model = ref_voc.WeCLIP(num_classes=21, clip_model=ck, embedding_dim=256, in_channels=[768] * 4, dataset_root_path=tmp,
                               device="cpu")
        model.bg_text_features, model.fg_text_features = bg, fg
        model.decoder_fts_fuse.load_state_dict(fuse_sd)
        model.decoder.load_state_dict(dec_sd)
        model.eval()
        seg, cam_labels, ap = model(img, ["im"])
    segs = torch.nn.functional.interpolate(seg, size=cam_labels.shape[1:], mode="bilinear", align_corners=False)
    mask = get_mask_by_radius(h=H // 16, w=W // 16, radius=8)
    aff_label = cams_to_affinity_label(cam_labels.clone(), mask=torch.from_numpy(mask), ignore_index=255)
    attn_loss, _, _ = get_aff_loss(ap, aff_label)
    seg_loss = get_seg_loss(segs, cam_labels.type(torch.long), ignore_index=255)
    (seg_loss + 0.1 * attn_loss).backward()
    grads = {}
    for n, p in list(model.decoder.named_parameters()) + list(model.decoder_fts_fuse.named_parameters()):
        grads[n] = p.grad
    keep = ["linear_pred.weight", "linear_pred.bias", "transformer.resblocks.2.attn.in_proj_bias",
            "transformer.resblocks.0.ln_1.weight", "transformer.resblocks.0.mlp.c_fc.bias", "transformer.resblocks.1.attn.out_proj.weight",
            "linears_modulelist.10.proj_2.bias", "linears_modulelist.0.proj.bias", "linears_modulelist.5.proj_2.weight", "linear_fuse.bias"]
    out = dict(weights_ck=checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]), img_ck=checksum([img]),
               img_index=np.int64(IDX), ids=np.array(ids), cam_labels=cam_labels[0].numpy().astype(np.uint8),
               attn_loss=np.float32(attn_loss.item()), seg_loss=np.float32(seg_loss.item()),
               n_pos=np.int64((aff_label == 1).sum()), n_neg=np.int64((aff_label == 0).sum()),
               grad_norms=np.array([float(grads[n].norm()) for n in sorted(grads)], np.float64), grad_names=np.array(sorted(grads)))
    for n in keep:
        out["grad:" + n] = grads[n].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "vitb_512_train.npz"), **out)
    print("vitb_512_train.npz written; losses", out["seg_loss"], out["attn_loss"], "labels", np.unique(out["cam_labels"], return_counts=True))


COCO_SIZES = [(80, 112), (96, 64), (71, 100)]      # synthetic "original" image sizes (the last one is odd on purpose)
COCO_LONG = 96                                     # --resize_long


def _msc_script_fn(name, ns):
    """One function of test_msc_flip_coco.py, compiled from the script's source without importing the script
    (it parses argv and imports omegaconf / joblib / imageio / pydensecrf at import time)."""
    src = open(os.path.join(refharness.REF, "test_msc_flip_coco.py")).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.Module([node], []), "test_msc_flip_coco.py", "exec"), ns)
            return ns[name]
    raise KeyError(name)


def coco_inputs():
    """(name, image (3,H,W), label (H,W) uint8) triples shared by the generator and the tests."""
    out = []
    for i, (H, W) in enumerate(COCO_SIZES):
        img = synth.make_images(1, H, W, seed=400 + i)[0]
        g = torch.Generator().manual_seed(500 + i)
        lab = torch.randint(0, 81, (max(H // 8, 1), max(W // 8, 1)), generator=g)
        lab = lab.repeat_interleave(8, 0).repeat_interleave(8, 1)[:H, :W].contiguous()
        lab[:3, :5] = 255
        out.append((f"im{i}", img, lab.to(torch.uint8)))
    return out


def make_tiny_coco_msc():
    import types
    from WeCLIP_model.model_attn_aff_coco import WeCLIP
    from utils import evaluate

    sd = synth.make_clip_state_dict(**TINY)
    fuse_sd, dec_sd = synth.make_head_state_dicts(width=TINY["width"], num_classes=81, seed=3)
    data = coco_inputs()

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(data)

        def __getitem__(self, i):
            n, img, lab = data[i]
            return n, img, lab.long(), torch.zeros(80)

    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, "clip_tiny.pt")
        torch.save(sd, ck)
        model = WeCLIP(num_classes=81, clip_model=ck, embedding_dim=256, in_channels=[TINY["width"]] * 4,
                       dataset_root_path=tmp, device="cpu")
        model.decoder_fts_fuse.load_state_dict(fuse_sd)
        model.decoder.load_state_dict(dec_sd)
        model.eval()
        tu = types.SimpleNamespace(data=types.SimpleNamespace(
            DataLoader=lambda ds, **kw: torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)))
        tproxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in ("cat", "mean", "stack", "argmax")}, utils=tu)
        ns = {"np": np, "torch": tproxy, "F": torch.nn.functional, "tqdm": lambda it, **kw: it, "evaluate": evaluate,
              "args": types.SimpleNamespace(resize_long=COCO_LONG)}
        validate = _msc_script_fn("validate", ns)
        with torch.no_grad():
            gts, preds, msc_preds, cams, h1, h2, h3 = validate(model, DS(), test_scales=[1.0, 0.75])
    assert len(preds) == len(data) and not h1.any()          # the script only folds the lists into its histograms every 2000 images
    hist, score = evaluate.scores(gts, preds, np.zeros((81, 81)), 81)
    msc_hist, msc_score = evaluate.scores(gts, msc_preds, np.zeros((81, 81)), 81)
    out = dict(weights_ck=checksum(sd.values()), head_ck=checksum(list(fuse_sd.values()) + list(dec_sd.values())),
               img_ck=checksum([d[1] for d in data]), sizes=np.array(COCO_SIZES), resize_long=np.int64(COCO_LONG),
               hist=hist.astype(np.int64), msc_hist=msc_hist.astype(np.int64),
               miou=np.float64(score["miou"]), msc_miou=np.float64(msc_score["miou"]),
               pacc=np.float64(score["pAcc"]), msc_pacc=np.float64(msc_score["pAcc"]))
    for i in range(len(data)):
        out[f"pred{i}"] = np.asarray(preds[i]).astype(np.uint8)
        out[f"msc_pred{i}"] = np.asarray(msc_preds[i]).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "tiny_coco_msc.npz"), **out)
    print("tiny_coco_msc.npz written; mIoU", score["miou"], "msc mIoU", msc_score["miou"],
          "classes predicted:", len(np.unique(np.concatenate([np.asarray(p).ravel() for p in msc_preds]))))


COCO_TRAIN_LABELS = [[2, 41], [0, 17, 79]]       # class ids (0..79) present in the two images


def make_tiny_coco_train(seg_trans):
    """The COCO model's TRAIN forward (model_attn_aff_coco.py:100-170 -> clip_tool.perform_single_coco_cam :221-319:
    80 classes + 23 background prompts, CAM threshold 0.7, GT PNGs under <root>/SegmentationClass/train; beyond
    iteration 40 000 the seg-trans branch over the last 10 maps)."""
    from PIL import Image
    from WeCLIP_model.model_attn_aff_coco import WeCLIP
    sd = synth.make_clip_state_dict(**TINY)
    H, W = TINY_HW
    img = synth.make_images(2, H, W, seed=600)
    bg, fg = synth.make_text_features(80, 23, TINY["embed_dim"], seed=5)
    fuse_sd, dec_sd = synth.make_head_state_dicts(width=TINY["width"], num_classes=81, seed=3)
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, "clip_tiny.pt")
        torch.save(sd, ck)
        d = os.path.join(tmp, "SegmentationClass", "train")
        os.makedirs(d)
        names = []
        for i, ids in enumerate(COCO_TRAIN_LABELS):
            png = np.zeros((H, W), np.uint8)
            for j, c in enumerate(ids):
                png[4 + 8 * j: 12 + 8 * j, 4:20] = c + 1
            png[-3:, -3:] = 255
            Image.fromarray(png).save(os.path.join(d, f"{2000 + i}.png"))
            names.append(2000 + i)
        model = WeCLIP(num_classes=81, clip_model=ck, embedding_dim=256, in_channels=[TINY["width"]] * 4,
                       dataset_root_path=tmp, device="cpu")
        model.bg_text_features, model.fg_text_features = bg, fg
        model.decoder_fts_fuse.load_state_dict(fuse_sd)
        model.decoder.load_state_dict(dec_sd)
        model.eval()
        if seg_trans:
            model.iter_num = 50000
        with torch.no_grad():
            pass
        seg, cam_labels, ap = model(img, names)
    out = dict(weights_ck=checksum(sd.values()), img_ck=checksum([img]), seg=seg.detach().numpy(),
               cam_labels=cam_labels.numpy().astype(np.uint8), attn_pred=ap.detach().numpy())
    fn = "tiny_coco_train_seg.npz" if seg_trans else "tiny_coco_train.npz"
    np.savez_compressed(os.path.join(OUT, fn), **out)
    print(fn, "written; labels present:", np.unique(out["cam_labels"]))


AUG_SEED, AUG_CROP, AUG_SRC = 31, 64, (54, 76)


def augment_inputs():
    """uint8 (6, 54, 76, 3) "decoded JPEGs": smooth structure + noise, shared by the generator and the tests."""
    f = synth.make_images(6, AUG_SRC[0], AUG_SRC[1], seed=700)
    return (f * 58.0 + 118.0).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def make_augment():
    """The reference's train-time input chain on six small images: datasets/transforms.py random_scaling (PIL BILINEAR)
    -> random_fliplr -> random_crop -> normalize_img -> CHW, called in the order of datasets/voc.py:109-143
    (`VOC12ClsDataset.__transforms`; that module itself needs numpy.lib.utils / imageio / torchvision at import time, so the
    four calls are made here).  The reference's random sources are replaced by seeded recording proxies of the same
    generators (`random.Random`, `np.random.RandomState`), so the fixture holds every draw next to the output."""
    import random as pyrandom
    import datasets.transforms as T
    assert T.__file__.startswith(refharness.REF)
    imgs = augment_inputs().numpy()
    draws = []

    class Rec:
        def __init__(self, rng):
            self.rng = rng

        def __getattr__(self, name):
            fn = getattr(self.rng, name)

            def call(*a, **k):
                v = fn(*a, **k)
                draws[-1].append((name, float(v)))
                return v
            return call

    py = Rec(pyrandom.Random(AUG_SEED))
    nprs = Rec(np.random.RandomState(AUG_SEED))
    orig_random, orig_randint = T.random, np.random.randint
    T.random, np.random.randint = py, nprs.randint
    outs, boxes = [], []
    try:
        for b in range(len(imgs)):
            draws.append([])
            image = np.array(imgs[b])
            image = T.random_scaling(image, scale_range=[0.5, 2.0])
            image = T.random_fliplr(image)
            image, img_box = T.random_crop(image, crop_size=AUG_CROP, mean_rgb=[0, 0, 0], ignore_index=255)
            image = T.normalize_img(image)
            outs.append(np.transpose(image, (2, 0, 1)))
            boxes.append(img_box)
    finally:
        T.random, np.random.randint = orig_random, orig_randint
    names = [[n for n, _ in d] for d in draws]
    assert all(n == ["uniform", "random", "randint", "randint", "randrange", "randrange"] for n in names), names
    vals = np.array([[v for _, v in d] for d in draws], np.float64)            # (6, 6): scale, p_flip, pad_y, pad_x, crop_y, crop_x
    assert (vals[:, 0] < 0.9).sum() >= 2 and (vals[:, 0] > 1.1).sum() >= 2, vals[:, 0]
    out = dict(img_ck=checksum([torch.from_numpy(imgs)]), seed=np.int64(AUG_SEED), crop=np.int64(AUG_CROP), draws=vals,
               out=np.stack(outs).astype(np.float32), img_box=np.stack(boxes).astype(np.int64))
    np.savez_compressed(os.path.join(OUT, "augment_ref.npz"), **out)
    print("augment_ref.npz written; scales", np.round(vals[:, 0], 3), "flips", (vals[:, 1] > 0.5).astype(int))


if __name__ == "__main__":
    refharness.install()
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        make_augment()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coco":
        make_tiny_coco_msc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coco_train":
        make_tiny_coco_train(False)
        make_tiny_coco_train(True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "512":        # only the benchmark-size fixtures
        make_vitb_512(False)
        make_vitb_512(True)
        make_vitb_512(True, synth.SINK_512)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "320":
        make_vitb_512(False, size=320, batch=4, index=1, seed=320, label_seed=9, fn_override="vitb_320.npz")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "512train":
        make_vitb_512_train()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "512sink":
        make_vitb_512(True, synth.SINK_512)
        sys.exit(0)
    make_tiny_func()
    make_tiny_whole(False)
    make_tiny_whole(True)
    make_vitb_224()
    make_vitb_512(False)
    make_vitb_512(True)
    make_vitb_512(True, synth.SINK_512)
    make_vitb_512_train()
    make_vitb_512(False, size=320, batch=4, index=1, seed=320, label_seed=9, fn_override="vitb_320.npz")
    make_tiny_coco_msc()
    make_tiny_coco_train(False)
    make_tiny_coco_train(True)
    make_augment()
