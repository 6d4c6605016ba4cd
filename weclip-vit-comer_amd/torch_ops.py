"""PyTorch custom-op registration of the hot-path operators (`torch.ops.weclip.*`).

BASELINE.json's north star asks for the HIP kernels to be "exposed to Python through PyTorch-ROCm custom ops so
WeCLIP_model.* and the segformer_head keep their nn.Module API".  This module registers the operator-level entry
points of libweclip_hip.so with the PyTorch dispatcher (`torch.library.custom_op`, CUDA/HIP device only) together
with shape-propagating fake kernels, so the ops can be called as `torch.ops.weclip.<name>`, show up under their own
names in the PyTorch profiler and can be traced (`torch.export` / `make_fx`) like any ATen op.  The nn.Modules of
this package call them at module granularity (PAR.forward, compute_trans_mat, the fused losses' forward,
confusion_hist, ...); inside a block the engines keep talking to the C ABI directly (vit_engine / head_engine:
one Python frame per launch matters there).

There is NO CPU kernel: a CPU tensor raises NotImplementedError from the dispatcher (no backend registered),
which is the "fail loudly" contract of the package.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

F32 = torch.float32
_LIB = "weclip"


def _par_module(dilations, num_iter):
    from .WeCLIP_model.PAR import PAR
    key = (tuple(dilations), int(num_iter))
    m = _par_cache.get(key)
    if m is None:
        m = _par_cache[key] = PAR(list(dilations), int(num_iter))
    return m


_par_cache = {}


@torch.library.custom_op(f"{_LIB}::par_forward", mutates_args=(), device_types="cuda")
def par_forward(imgs: Tensor, masks: Tensor, dilations: List[int], num_iter: int) -> Tensor:
    """PAR.forward (reference WeCLIP_model/PAR.py:64-92): refined masks (b, C, H, W)."""
    return _par_module(dilations, num_iter)._forward_impl(imgs, masks)


@par_forward.register_fake
def _(imgs, masks, dilations, num_iter):
    return masks.new_empty(masks.shape, dtype=F32)


@torch.library.custom_op(f"{_LIB}::par_labels", mutates_args=(), device_types="cuda")
def par_labels(masks: Tensor, valid_key: Tensor) -> Tensor:
    """valid_key[argmax_c masks] (reference `_refine_cams`, model_attn_aff_voc.py:49-57) -> (B, H, W) int64."""
    from .WeCLIP_model.PAR import refine_labels
    return refine_labels(masks.float().contiguous(), valid_key.long().contiguous())


@par_labels.register_fake
def _(masks, valid_key):
    B, _, H, W = masks.shape
    return masks.new_empty((B, H, W), dtype=torch.int64)


@torch.library.custom_op(f"{_LIB}::trans_mat", mutates_args=(), device_types="cuda")
def trans_mat(attn_weight: Tensor) -> Tensor:
    """compute_trans_mat (reference clip/clip_tool.py:64-80) of a batch of (hw, hw) affinities."""
    from . import cam_pipeline as CP
    w = attn_weight.detach().float().contiguous()
    return CP.trans_mat(w if w.dim() == 3 else w[None]).view(attn_weight.shape)


@trans_mat.register_fake
def _(attn_weight):
    return attn_weight.new_empty(attn_weight.shape, dtype=F32)


@torch.library.custom_op(f"{_LIB}::attention", mutates_args=(), device_types="cuda")
def attention(qkv16: Tensor, B: int, L: int, H: int, DH: int, want_mean: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """Flash attention forward + head-mean probabilities (reference clip/myAtt.py:21-64, 325-326) on the packed fp16
    in-projection output (B*L, 3*H*DH), q pre-scaled by log2(e)/sqrt(DH).  -> (O fp16 (B*L, E), LSE (B,H,L),
    mean (B,L,L); a (0,) tensor when want_mean is False)."""
    from . import ops
    o16, lse, mean = ops.attention(qkv16, B, L, H, DH, want_mean=want_mean)
    return o16, lse, mean if mean is not None else qkv16.new_empty((0,), dtype=F32)


@attention.register_fake
def _(qkv16, B, L, H, DH, want_mean):
    return (qkv16.new_empty((B * L, H * DH)), qkv16.new_empty((B, H, L), dtype=F32),
            qkv16.new_empty((B, L, L) if want_mean else (0,), dtype=F32))


@torch.library.custom_op(f"{_LIB}::linear_f16", mutates_args=(), device_types="cuda")
def linear_f16(x16: Tensor, w16: Tensor, bias: Tensor, act: int) -> Tensor:
    """y = act(x W^T + b) on the MFMA path (F.linear call sites of the reference): x16 (M, K), w16 (N, K) fp16,
    bias (N,) f32, act 0 none / 1 QuickGELU / 2 ReLU / 3 sigmoid -> (M, N) f32.  K % 64 == 0."""
    from . import ops
    M, K = x16.shape
    N = w16.shape[0]
    out = torch.empty(M, N, device=x16.device, dtype=F32)
    ops.gemm(x16.contiguous(), w16.contiguous(), M, N, K, bias=bias.float().contiguous(), out32=out, act=act)
    return out


@linear_f16.register_fake
def _(x16, w16, bias, act):
    return x16.new_empty((x16.shape[0], w16.shape[0]), dtype=F32)


@torch.library.custom_op(f"{_LIB}::layernorm", mutates_args=(), device_types="cuda")
def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """Row LayerNorm with fp32 statistics (reference clip/model.py:177-183) -> f32, same shape."""
    from . import ops
    y, _ = ops.layernorm(x.float().contiguous().view(-1, x.shape[-1]), weight.float().contiguous(), bias.float().contiguous(),
                         eps=eps, want32=True, want16=False)
    return y.view(x.shape)


@layernorm.register_fake
def _(x, weight, bias, eps):
    return x.new_empty(x.shape, dtype=F32)


@torch.library.custom_op(f"{_LIB}::bilinear_resize", mutates_args=(), device_types="cuda")
def bilinear_resize(x: Tensor, out_h: int, out_w: int, align_corners: bool) -> Tensor:
    """F.interpolate(x, (out_h, out_w), 'bilinear', align_corners) of (N, C, H, W) f32 planes."""
    from .resize import bilinear_resize as br
    return br(x.float().contiguous(), (out_h, out_w), align_corners=align_corners)


@bilinear_resize.register_fake
def _(x, out_h, out_w, align_corners):
    return x.new_empty(x.shape[:-2] + (out_h, out_w), dtype=F32)


@torch.library.custom_op(f"{_LIB}::confusion_hist", mutates_args=(), device_types="cuda")
def confusion_hist(label_true: Tensor, label_pred: Tensor, num_classes: int) -> Tensor:
    """_fast_hist (reference utils/evaluate.py:10-16) -> (nc, nc) int64."""
    from .utils.evaluate import confusion_hist as ch
    return ch(label_true, label_pred, num_classes)


@confusion_hist.register_fake
def _(label_true, label_pred, num_classes):
    return label_true.new_empty((num_classes, num_classes), dtype=torch.int64)


@torch.library.custom_op(f"{_LIB}::seg_loss", mutates_args=(), device_types="cuda")
def seg_loss(seg_lowres: Tensor, label: Tensor, ignore_index: int) -> Tensor:
    """get_seg_loss(F.interpolate(seg, label.shape[1:]), label) (reference scripts/dist_clip_voc.py:105-113,250), forward
    value only (the differentiable form is utils.losses.get_seg_loss_fused)."""
    from .utils.losses import get_seg_loss_fused
    return get_seg_loss_fused(seg_lowres.detach(), label, ignore_index).detach()


@seg_loss.register_fake
def _(seg_lowres, label, ignore_index):
    return seg_lowres.new_empty((), dtype=F32)


@torch.library.custom_op(f"{_LIB}::aff_loss", mutates_args=(), device_types="cuda")
def aff_loss(attn_pred: Tensor, cam_label: Tensor, radius: int, ignore_index: int) -> Tensor:
    """get_aff_loss(attn_pred, cams_to_affinity_label(cam_label, radius mask)) (reference utils/losses.py:11-22,
    utils/camutils.py:226-247), forward value only (differentiable form: utils.losses.get_aff_loss_fused)."""
    from .utils.losses import get_aff_loss_fused
    return get_aff_loss_fused(attn_pred.detach(), cam_label, radius, ignore_index).detach()


@aff_loss.register_fake
def _(attn_pred, cam_label, radius, ignore_index):
    return attn_pred.new_empty((), dtype=F32)


# ---------------------------------------------------------------------------------------------------------------------
# Trainable operators: forward and backward are both registered ops, tied together with `register_autograd`
# ("ops used by trainable modules need ... register_autograd backward").  hip_functional.linear /
# layer_norm -- the nn.Linear / 1x1 nn.Conv2d / nn.LayerNorm layers of the module-by-module ViT-CoMer form -- call these.
GRAD_SCALE = 4096.0


@torch.library.custom_op(f"{_LIB}::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int) -> Tensor:
    """y = act(x W^T + b) for fp32 x (M, K), W (N, K) [or a 1x1 conv kernel (N, K, 1, 1)], b (N) or None on the MFMA path
    (fp16 operands [hi + lo in `exact` precision], fp32 accumulate).  act 0 none / 2 ReLU.  K % 64 == 0."""
    from . import config, ops
    x = x.detach().float().contiguous()
    M, K = x.shape
    N = weight.shape[0]
    if K % 64:
        raise RuntimeError(f"weclip::linear: K = {K} must be a multiple of 64")
    ex = config.exact()
    y = torch.empty(M, N, device=x.device, dtype=F32)
    ops.gemm(ops.split_f16(x, with_lo=ex), ops.split_f16(weight.detach().float().reshape(N, K).contiguous(), with_lo=ex), M, N, K,
             bias=bias.detach().float().contiguous() if bias is not None else None, out32=y, act=act)
    return y


@linear.register_fake
def _(x, weight, bias, act):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=F32)


@torch.library.custom_op(f"{_LIB}::linear_bwd", mutates_args=(), device_types="cuda")
def linear_bwd(dy: Tensor, x: Tensor, weight: Tensor, y: Tensor, act: int, need_dx: bool, need_dw: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """Gradients of weclip::linear: (dx (M, K), dW (N, K), db (N)); a (0,) tensor where not needed.  The input gradient is
    one MFMA GEMM against the transposed weight, the weight + bias gradients the split-K row-major GEMM (csrc/gemm.hip
    gemm_km_kernel, bias = extra output column); gradients travel multiplied by 2^12 so they sit in fp16's normal range."""
    from . import _lib as L
    from . import config, ops
    M, K = x.shape
    N = weight.shape[0]
    dev = dy.device
    ex = config.exact()
    dy = dy.float().contiguous()
    if act == 2:
        dy = dy * (y > 0)
    Np = (N + 63) // 64 * 64
    if Np != N:                                   # the contraction dimension of dX = dY W must be a multiple of 64
        pad = torch.zeros(M, Np, device=dev, dtype=F32)
        pad[:, :N] = dy
        dy = pad
    _, dS = ops.colscale_split(dy, None, M, alpha=GRAD_SCALE, want32=False, with_lo=ex)
    w2 = weight.detach().float().reshape(N, K).contiguous()
    # one placeholder PER slot: outputs of a custom op may not alias each other
    dx, dw, db = (torch.empty((0,), device=dev, dtype=F32) for _ in range(3))
    if need_dx:
        wT, Kp = ops.transpose_f16(w2, N, K, with_lo=ex)          # (K, Np) fp16: W^T with the N columns zero padded
        dx = torch.empty(M, K, device=dev, dtype=F32)
        ops.gemm(dS, wT, M, K, Kp, out32=dx, scale=1.0 / GRAD_SCALE, scale_cols=K)
    if need_dw:
        xhi = ops.split_f16(x.detach().float().contiguous()).hi
        tiles = ops.wgrad_tiles(N, K)
        ns = 1
        while ns * 2 * tiles <= 512 and M // (ns * 2) >= 256:
            ns *= 2
        part, ns = ops.wgrad_partials(dS.hi, xhi, M, N, K, lda=Np, slices=ns, bias=True)
        dw = torch.empty(N, K, device=dev, dtype=F32)
        db = torch.empty(N, device=dev, dtype=F32)
        L.lib().wc_sum_slices_wb(L.ptr(part, F32), L.ptr(dw, F32), L.ptr(db, F32), ns, N, K, 1.0 / GRAD_SCALE, L.stream())
    return dx, dw, db


@linear_bwd.register_fake
def _(dy, x, weight, y, act, need_dx, need_dw):
    N, K = weight.shape[0], x.shape[1]
    def e():
        return dy.new_empty((0,), dtype=F32)
    return (x.new_empty(x.shape, dtype=F32) if need_dx else e(), x.new_empty((N, K), dtype=F32) if need_dw else e(),
            x.new_empty((N,), dtype=F32) if need_dw else e())


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act = inputs
    ctx.save_for_backward(x, weight, output if act == 2 else None)
    ctx.act, ctx.has_bias, ctx.wshape = act, bias is not None, weight.shape


def _linear_backward(ctx, dy):
    x, weight, y = ctx.saved_tensors
    need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
    dx, dw, db = torch.ops.weclip.linear_bwd(dy, x, weight, y if y is not None else dy.new_empty((0,)), ctx.act, need_dx, need_dw)
    return (dx if need_dx else None, dw.view(ctx.wshape) if need_dw else None, db if (need_dw and ctx.has_bias) else None, None)


torch.library.register_autograd(f"{_LIB}::linear", _linear_backward, setup_context=_linear_setup)


@torch.library.custom_op(f"{_LIB}::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """Row LayerNorm (fp32 statistics, reference clip/model.py:177-183) of x (rows, D) -> f32; differentiable
    (weclip::layer_norm_bwd through register_autograd)."""
    from . import ops
    y, _ = ops.layernorm(x.detach().float().contiguous(), weight.detach().float().contiguous(), bias.detach().float().contiguous(),
                         eps=eps, want32=True, want16=False)
    return y


@layer_norm.register_fake
def _(x, weight, bias, eps):
    return x.new_empty(x.shape, dtype=F32)


@torch.library.custom_op(f"{_LIB}::layer_norm_bwd", mutates_args=(), device_types="cuda")
def layer_norm_bwd(dy: Tensor, x: Tensor, weight: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """(dx, dgamma, dbeta) of weclip::layer_norm (csrc/train_ops.hip ln_bwd, fixed-order column reductions)."""
    from . import ops
    dx, _, dgb = ops.layernorm_bwd(dy.float().contiguous(), x.detach().float().contiguous(), weight.detach().float().contiguous(),
                                   want32=True, eps=eps)
    return dx, dgb[0].clone(), dgb[1].clone()


@layer_norm_bwd.register_fake
def _(dy, x, weight, eps):
    return x.new_empty(x.shape, dtype=F32), weight.new_empty(weight.shape, dtype=F32), weight.new_empty(weight.shape, dtype=F32)


def _ln_setup(ctx, inputs, output):
    x, weight, bias, eps = inputs
    ctx.save_for_backward(x, weight)
    ctx.eps = eps


def _ln_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dg, db = torch.ops.weclip.layer_norm_bwd(dy, x, weight, ctx.eps)
    return dx, dg, db, None


torch.library.register_autograd(f"{_LIB}::layer_norm", _ln_backward, setup_context=_ln_setup)


OPS = ("par_forward", "par_labels", "trans_mat", "attention", "linear_f16", "layernorm", "bilinear_resize", "confusion_hist",
       "seg_loss", "aff_loss", "linear", "linear_bwd", "layer_norm", "layer_norm_bwd")
