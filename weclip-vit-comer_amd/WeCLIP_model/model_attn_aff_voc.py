"""`WeCLIP` (VOC) with the reference's constructor / forward / state-dict contract
(WeCLIP_model/model_attn_aff_voc.py:60-175): frozen CLIP ViT encoder -> adapters -> decoder ->
attn_pred = sigmoid(F^T F) -> per image GradCAM + affinity refinement + PAR -> pseudo labels.
The per-image Python loop with its numpy/cv2/PIL round trips is replaced by one batched,
device-resident pipeline (all images x classes at once)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import cam_pipeline as CP
from ..clip import clip_tool as CT
from ..clip import vit_engine as VE
from ..clip.clip import load as clip_load
from ..head_engine import DecoderFunction, HeadEngine, HeadFunction
from ..pytorch_grad_cam import GradCAM
from .comer import CoMerInteraction
from .Decoder.TransDecoder import DecoderTransformer
from .PAR import PAR, refine_labels
from .segformer_head import SegFormerHead


def reshape_transform(tensor, height=28, width=28):
    """(L, N, D) tokens -> (N, D, h, w) patch grid (reference :23-30); kept for API parity."""
    t = tensor.permute(1, 0, 2)[:, 1:, :]
    return t.reshape(t.size(0), height, width, t.size(2)).permute(0, 3, 1, 2)


class WeCLIP(nn.Module):
    cam_threshold = 0.4      # scoremap2bbox threshold (clip_tool.py:179)
    seg_trans_last = 6       # maps used in the seg-trans branch (clip_tool.py:155)
    seg_trans_after = 15000  # iteration after which the seg-trans branch is used (:146)
    val_runs_cam = True      # the VOC model runs the CAM/PAR path in 'val' too (:146-155)

    def __init__(self, num_classes=None, clip_model=None, embedding_dim=256, in_channels=512,
                 dataset_root_path=None, device="cuda", text_features=None, comer=False):
        """`text_features=(bg (n_bg, Ed), fg (n_fg, Ed))`: the zero-shot text rows.  The reference
        computes them at construction with the CLIP text tower + tokenizer (:34-46,81-82); that
        init-time step is outside this package, so they are passed in (or assigned later to
        `bg_text_features` / `fg_text_features`)."""
        super().__init__()
        self.num_classes, self.embedding_dim, self.in_channels = num_classes, embedding_dim, in_channels
        self.encoder, _ = clip_load(clip_model, device=device)
        for name, p in self.encoder.named_parameters():
            p.requires_grad = "11" in name                      # reference :68-70
        self.decoder_fts_fuse = SegFormerHead(in_channels=in_channels, embedding_dim=embedding_dim,
                                              num_classes=num_classes, index=11)
        self.decoder = DecoderTransformer(width=embedding_dim, layers=3, heads=8, output_dim=num_classes)
        self.bg_text_features, self.fg_text_features = (None, None) if text_features is None else text_features
        self.target_layers = [self.encoder.visual.transformer.resblocks[-1].ln_1]
        self.grad_cam = GradCAM(model=self.encoder, target_layers=self.target_layers,
                                reshape_transform=reshape_transform)
        self.root_path = os.path.join(dataset_root_path, "SegmentationClassAug") if dataset_root_path else None
        self.cam_bg_thres = 1
        self.encoder.eval()
        self.par = PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])
        self.iter_num = 0
        self.require_all_fts = True
        self.head_impl = os.environ.get("WECLIP_HEAD", "hip")   # "hip" (head_engine.py) | "torch" (stock autograd)
        self.head_engine = HeadEngine(self.decoder_fts_fuse, self.decoder)
        # optional ViT-CoMer inserts (comer.py): MRFP + bidirectional CTI on the adapter outputs of
        # blocks [2,5,8,11]; their fusion replaces `linear_fuse` as the decoder input.  Extra keys
        # `comer.*` in the state dict; absent (and the reference contract untouched) by default.
        self.comer = CoMerInteraction(embedding_dim) if comer else None
        self.fork_head = os.environ.get("WECLIP_FORK_HEAD", "1") != "0"      # head forward beside the CAM chain (second stream)
        self.to(device)

    def get_param_groups(self):
        groups = [[], [], [], []]   # backbone; backbone_norm; cls_head; seg_head
        groups[3].extend(self.decoder.parameters())
        groups[3].extend(self.decoder_fts_fuse.parameters())
        if self.comer is not None:
            groups[3].extend(self.comer.parameters())
        return groups

    # ------------------------------------------------------------------------------------------
    def _labels_for(self, img_names, labels, shape):
        if labels is not None:
            return [list(l) for l in labels], [tuple(shape)] * len(labels)
        if self.root_path is None:
            raise RuntimeError("pass labels=[[class ids]...] or construct WeCLIP with dataset_root_path")
        out, sizes = [], []
        for name in img_names:
            ids, osz = CT.read_image_labels(os.path.join(self.root_path, str(name) + ".png"))
            out.append(ids)
            sizes.append(osz)
        return out, sizes

    def _maps_needed(self, seg_trans):
        n = self.seg_trans_last if seg_trans else 8
        first = 12 - n                      # index into the 12 maps (11 encoder + last block)
        return [i >= first for i in range(11)]

    def encode(self, img, seg_trans, x16=None, want_maps=True):
        """Frozen encoder: token rows of blocks 1..11 and the head-mean maps the affinity needs (none when the
        caller returns before the CAM stage: the COCO model in 'val', model_attn_aff_coco.py:131-132).
        `x16` (a list) additionally receives fp16 copies of the block outputs (adapter operands)."""
        vis = self.encoder.visual
        rows, B, Lq = vis.embed(img)
        need = self._maps_needed(seg_trans) if want_maps else [False] * 11
        xs, maps = [], []
        for i in range(vis.transformer.layers - 1):
            rows, m = VE.run_block(vis.transformer.resblocks[i].pack(), rows, B, Lq, want_mean=need[i], x16_out=x16,
                                   tag=b"@vit_attn")
            xs.append(rows)
            maps.append(m)
        return xs, maps, B, Lq

    def forward(self, img, img_names="2007_000032", mode="train", labels=None, plan=None):
        """-> (seg (B,nc,h,w), cam_labels (B,H,W) int64 [list of per-image maps in 'val' when the
        original sizes differ], attn_pred (B,hw,hw)).  `plan`: a ready clip_tool.PairPlan for `labels`
        (TrainStep's graph mode keeps one per batch signature and refills it in place)."""
        B, _, H, W = img.shape
        h, w = H // 16, W // 16
        self.encoder.eval()
        self.iter_num += 1
        seg_trans = self.iter_num > self.seg_trans_after or mode == "val"
        img = img.cuda().float().contiguous()
        hip_head = self.head_impl == "hip" and self.comer is None
        self._bwd_forked = False              # set when this forward created autograd nodes on the side stream
        comer_tokens = self.comer is not None and self.head_impl == "hip" and self.comer.engine_ok(img)
        x16 = VE.X16Stack(self.encoder.visual.transformer.layers - 1) if (hip_head or comer_tokens) else None
        with torch.no_grad():
            want_cam = not (mode == "val" and not self.val_runs_cam)
            xs, maps, _, Lq = self.encode(img, seg_trans, x16, want_maps=want_cam)
        if hip_head:
            # adapters + decoder + attn_pred, forward and backward as HIP launches (head_engine.py)
            drop = None
            if self.training:
                p = self.decoder_fts_fuse.dropout.p
                drop = ((torch.rand(B, self.embedding_dim, device=img.device) >= p).float() / (1.0 - p)).contiguous()
            # Outside the seg-trans branch the CAM -> affinity -> PAR chain reads nothing the head produces (clip_tool.py:146-176
            # takes attn_pred only after iteration 15000): the head's forward runs on a second stream beside it and joins before the
            # losses (measured: the gain comes from running beside the last-layer / GradCAM GEMMs; beside the PAR sweeps it is a loss).
            # In the seg-trans branch the affinity does read attn_pred: the fork then covers the last-layer forward + GradCAM,
            # and the CAM chain joins the head's stream right before the affinity weight (the callable below).
            fork = self.fork_head and want_cam and img.is_cuda
            if fork:
                main = torch.cuda.current_stream()
                side = CT.side_stream(img.device)
                self.head_engine.fwd_stream = side
                try:
                    seg, attn_pred = HeadFunction.apply(self.head_engine, x16, B, Lq, h, w, drop, *self.head_engine.params())
                finally:
                    self.head_engine.fwd_stream = None

                def joined_attn_pred():
                    main.wait_stream(side)
                    return attn_pred.detach()

                with torch.no_grad():
                    cam_labels = self.cam_labels(img, xs[-1], maps, joined_attn_pred if seg_trans else None, img_names, labels, mode,
                                                 seg_trans, h, w, plan=plan)
                main.wait_stream(side)
                return seg, cam_labels, attn_pred
            seg, attn_pred = HeadFunction.apply(self.head_engine, x16, B, Lq, h, w, drop, *self.head_engine.params())
        else:
            # (same fork for the module-by-module head: its autograd nodes then belong to the side stream, their backward runs
            #  there, and TrainStep joins `side_streams()` after loss.backward())
            fork = self.fork_head and want_cam and not seg_trans and img.is_cuda and self.head_impl == "hip" and self.training
            if fork:
                main = torch.cuda.current_stream()
                side = CT.side_stream(img.device)
                side.wait_stream(main)
                self._bwd_forked = True
                with torch.cuda.stream(side):
                    seg, attn_pred = self._module_head(img, xs, x16, comer_tokens, B, Lq, h, w)
                with torch.no_grad():
                    cam_labels = self.cam_labels(img, xs[-1], maps, None, img_names, labels, mode, seg_trans, h, w, plan=plan)
                main.wait_stream(side)
                return seg, cam_labels, attn_pred
            seg, attn_pred = self._module_head(img, xs, x16, comer_tokens, B, Lq, h, w)
        if mode == "val" and not self.val_runs_cam:
            return seg, None, attn_pred
        with torch.no_grad():
            cam_labels = self.cam_labels(img, xs[-1], maps, attn_pred.detach(), img_names, labels, mode,
                                         seg_trans, h, w, plan=plan)
        return seg, cam_labels, attn_pred

    def side_streams(self):
        """Streams (besides the caller's) that BACKWARD work of the last forward runs on: TrainStep joins them after
        `loss.backward()`.  Only the module-form head forked under `torch.cuda.stream(side)` leaves autograd nodes there (the
        fused HeadFunction keeps its node on the caller's stream); a forward that did not fork returns nothing, so a step being
        captured into a HIP graph never waits on a stream outside its capture.  The stream objects live in clip_tool (a stream
        must not end up in a pickled / deep-copied model).
        Allocator note: tensors allocated on the side stream (seg, attn_pred, the inserts' saved activations) are used by the
        caller's stream only after `main.wait_stream(side)`, and are freed on the host after `loss.backward()` has been enqueued;
        the side stream's next use starts with `side.wait_stream(main)` in the next forward, i.e. behind every kernel that read
        them, so the caching allocator cannot hand their blocks to a kernel that overtakes a reader."""
        if not getattr(self, "_bwd_forked", False):
            return []
        return CT.side_streams()

    def _module_head(self, img, xs, x16, comer_tokens, B, Lq, h, w):
        """Adapters [+ ViT-CoMer inserts] -> decoder -> attn_pred through autograd nodes (every head form but the fused engine)."""
        if comer_tokens:       # adapters of the four stage blocks + inserts as one engine, fed by the encoder's f16 block outputs
            fts = self.decoder_fts_fuse.dropout(self.comer.forward_tokens(img, x16, Lq, self.decoder_fts_fuse.linears_modulelist,
                                                                          (h, w)))
        elif self.comer is not None:
            used = set(self.comer.stage_blocks)       # only these adapter outputs enter the inserts
            toks = [mlp.tokens(r.view(B, Lq, -1)[:, 1:, :]) if i in used else None for i, (mlp, r) in
                    enumerate(zip(self.decoder_fts_fuse.linears_modulelist, xs))]
            fts = self.decoder_fts_fuse.dropout(self.comer(img, toks, (h, w)))
        else:
            fts = self.decoder_fts_fuse.forward_rows(xs, B, Lq, h, w)
        if self.head_impl == "hip":      # decoder + linear_pred + attn_pred on the HIP path, from the fused features
            rows = fts.permute(0, 2, 3, 1).reshape(B * h * w, fts.shape[1])
            seg, attn_pred = DecoderFunction.apply(self.head_engine, rows, B, h, w, *self.head_engine.dec_params())
        else:
            seg, _ = self.decoder(fts, need_weights=False)
            f = fts.reshape(B, fts.shape[1], h * w)
            attn_pred = torch.sigmoid(f.transpose(2, 1).bmm(f))
        return seg, attn_pred

    def cam_labels(self, img, last_rows, maps, attn_pred, img_names, labels, mode, seg_trans, h, w, plan=None):
        if self.bg_text_features is None or self.fg_text_features is None:
            raise RuntimeError("WeCLIP needs text_features=(bg, fg) (see __init__)")
        B, _, H, W = img.shape
        if isinstance(img_names, str):
            img_names = [img_names]
        dev = img.device
        if plan is None:
            label_lists, sizes = self._labels_for(img_names, labels, (H, W))
            plan = CT.PairPlan(label_lists, self.fg_text_features.shape[0], self.bg_text_features.shape[0], dev)
        else:
            sizes = [(H, W)] * plan.B
        text_hat = CT.normalised_text(self.fg_text_features, self.bg_text_features, dev)
        R, _, _, _ = CT.batch_refined_cams(self.encoder, last_rows, maps, attn_pred if seg_trans else None,
                                           plan, text_hat, h, w, self.cam_threshold, seg_trans,
                                           self.seg_trans_last)
        C = plan.K + 1
        if mode == "train" or all(tuple(s) == (H, W) for s in sizes):
            cams = CP.upsample_with_bg(R, plan.nk, h, w, H, W, C)          # cam_bg_thres = 1: pow is identity
            refined = self.par(img, cams)
            return refine_labels(refined, plan.valid_key, plan.nch)
        out = []                                                            # 'val': original image sizes
        for i, (oh, ow) in enumerate(sizes):
            cams = CP.upsample_with_bg(R[i:i + 1].contiguous(), plan.nk[i:i + 1].contiguous(), h, w, oh, ow, C)
            refined = self.par(img[i:i + 1], cams)
            out.append(refine_labels(refined, plan.valid_key[i:i + 1].contiguous(),
                                     plan.nch[i:i + 1].contiguous())[0])
        return out
