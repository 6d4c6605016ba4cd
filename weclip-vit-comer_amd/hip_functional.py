"""Autograd functions on the HIP kernels for the small trainable layers around the deformable-attention cores of
the ViT-CoMer inserts: nn.Linear / 1x1 nn.Conv2d -> the MFMA GEMM (csrc/gemm.hip, forward,
input gradient, split-K weight + bias gradient), nn.LayerNorm -> csrc/norm.hip / train_ops.hip, 3x3 stride-s
nn.Conv2d -> im2col gather + the same GEMM, nn.GroupNorm + ReLU -> csrc/convstem.hip.

Same numerics as the WeCLIP head (head_engine.py): fp16 MFMA operands (hi [+ lo in `exact` precision]), fp32
accumulation; gradients travel multiplied by GRAD_SCALE so they sit in fp16's normal range.
The nn.Modules keep owning the parameters (state-dict keys unchanged); only their forward goes through here.
"""
import torch

from . import _lib as L
from . import config, ops
from .ops import F16, F32, Split

GRAD_SCALE = 4096.0


def _rows(x):
    K = x.shape[-1]
    return x.reshape(-1, K), x.shape[:-1]


def linear(x, weight, bias=None, act=0):
    """F.linear(x, weight, bias) [+ ReLU] for x (..., K) on the HIP path; weight may be a 1x1 conv kernel (N, K, 1, 1).
    Runs as the registered, differentiable custom op `torch.ops.weclip.linear` (torch_ops.py: forward = MFMA GEMM,
    backward = weclip::linear_bwd via register_autograd)."""
    from . import torch_ops  # noqa: F401  (registers torch.ops.weclip.*)
    rows, lead = _rows(x)
    y = torch.ops.weclip.linear(rows, weight, bias, act)
    return y.view(*lead, weight.shape[0])


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dimension as the differentiable custom op `torch.ops.weclip.layer_norm`."""
    from . import torch_ops  # noqa: F401
    rows, lead = _rows(x)
    return torch.ops.weclip.layer_norm(rows, weight, bias, eps).view(*lead, x.shape[-1])


def module_linear(mod, x, act=0):
    """nn.Linear / 1x1 nn.Conv2d module applied to token rows x (..., K) through the HIP path (CUDA) or stock torch."""
    if x.is_cuda and x.shape[-1] % 64 == 0:
        return linear(x, mod.weight, mod.bias, act)
    y = torch.nn.functional.linear(x, mod.weight.flatten(1), mod.bias)
    return torch.relu(y) if act == 2 else y


def module_layer_norm(mod, x):
    if x.is_cuda:
        return layer_norm(x, mod.weight, mod.bias, mod.eps)
    return mod(x)


class _Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(C, O, 3, stride, padding=1, bias=False) on NHWC rows: im2col gather + MFMA GEMM (csrc/convstem.hip)."""

    @staticmethod
    def forward(ctx, x, weight, N, H, W, stride):
        L.require_gpu()
        x = x.detach().float().contiguous()
        C, O = x.shape[-1], weight.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        M = N * Ho * Wo
        Kp = (9 * C + 63) // 64 * 64
        ex = config.exact()
        cols = Split(torch.empty(M, Kp, device=x.device, dtype=F16), torch.empty(M, Kp, device=x.device, dtype=F16) if ex else None)
        L.lib().wc_im2col3x3(L.ptr(x, F32, "x"), L.ptr(cols.hi), L.ptr(cols.lo), N, H, W, C, stride, Kp, L.stream())
        wmat = torch.zeros(O, Kp, device=x.device, dtype=F32)
        wmat[:, :9 * C] = weight.detach().float().permute(0, 2, 3, 1).reshape(O, 9 * C)      # [o][(ky*3+kx)*C + c]
        y = torch.empty(M, O, device=x.device, dtype=F32)
        ops.gemm(cols, ops.split_f16(wmat, with_lo=ex), M, O, Kp, out32=y)
        ctx.save_for_backward(cols.hi, wmat)
        ctx.meta = (N, H, W, C, O, stride, Kp, M, ex, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        cols_hi, wmat = ctx.saved_tensors
        N, H, W, C, O, stride, Kp, M, ex, wshape = ctx.meta
        dev = dy.device
        dy = dy.float().contiguous()
        # fp16 operand of both gradient GEMMs (scaled into fp16's normal range).  The input gradient contracts over the O output
        # channels, which must then be padded to a multiple of 64; the weight gradient reads the O columns as they are.  The first
        # stem layer (O = 32, 1 048 576 rows at 512^2) has no input gradient: no padding, no 268 MB zero fill + strided copy.
        need_dx = ctx.needs_input_grad[0]
        Op = (O + 63) // 64 * 64 if need_dx else O
        _, dS = ops.colscale_split(dy, None, M, alpha=GRAD_SCALE, want32=False, with_lo=ex)
        if Op != O:
            pad = Split(torch.zeros(M, Op, device=dev, dtype=F16), torch.zeros(M, Op, device=dev, dtype=F16) if ex else None)
            for src, dst in ((dS.hi, pad.hi), (dS.lo, pad.lo)):
                if src is not None:
                    L.lib().wc_rows_copy_f16(L.ptr(src, F16), 0, L.ptr(dst, F16), 1, M, O, O, 0, Op, 0, L.stream())
            dS = pad
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wT, Kc = ops.transpose_f16(wmat, O, Kp, with_lo=ex)                   # (Kp, Op)
            dcols = torch.empty(M, Kp, device=dev, dtype=F32)
            ops.gemm(dS, wT, M, Kp, Kc, out32=dcols, scale=1.0 / GRAD_SCALE, scale_cols=Kp)
            dx = torch.empty(N * H * W, C, device=dev, dtype=F32)
            L.lib().wc_col2im3x3(L.ptr(dcols, F32), L.ptr(dx), N, H, W, C, stride, Kp, L.stream())
        if ctx.needs_input_grad[1]:
            tiles = ((O + 127) // 128) * ((Kp + 1 + 127) // 128)
            ns = 1
            while ns * 2 * tiles <= 512 and M // (ns * 2) >= 256:
                ns *= 2
            part, ns = ops.wgrad_partials(dS.hi, cols_hi, M, O, Kp, lda=Op, slices=ns, bias=False)
            dwm = torch.empty(O, Kp, device=dev, dtype=F32)
            L.lib().wc_sum_slices(L.ptr(part, F32), L.ptr(dwm, F32), ns, O * Kp, 1.0 / GRAD_SCALE, L.stream())
            dw = dwm[:, :9 * C].reshape(O, 3, 3, C).permute(0, 3, 1, 2).contiguous().view(wshape)
        return dx, dw, None, None, None, None


def conv3x3_rows(x_rows, weight, N, H, W, stride):
    """x_rows (N*H*W, C) NHWC rows -> ((N*Ho*Wo, O) rows, Ho, Wo)."""
    y = _Conv3x3Fn.apply(x_rows, weight, N, H, W, stride)
    return y, (H - 1) // stride + 1, (W - 1) // stride + 1


class _GroupNormReluFn(torch.autograd.Function):
    """relu(nn.GroupNorm(G, C)(x)) on rows (N, HW, C); every reduction in a fixed order (csrc/convstem.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, N, G, eps):
        L.require_gpu()
        x = x.detach().float().contiguous()
        C = x.shape[-1]
        HW = x.numel() // (N * C)
        nblk = (HW + 63) // 64
        dev = x.device
        y = torch.empty_like(x)
        stats = torch.empty(N * G * 2, device=dev, dtype=F32)
        part = torch.empty(N * nblk * G * 2, device=dev, dtype=F32)
        g32 = gamma.detach().float().contiguous()
        L.lib().wc_groupnorm_relu_fwd(L.ptr(x, F32, "x"), L.ptr(g32, F32), L.ptr(beta.detach().float().contiguous(), F32), L.ptr(y),
                                      L.ptr(stats), L.ptr(part), N, HW, C, G, float(eps), L.stream())
        ctx.save_for_backward(x, y, stats, g32)
        ctx.meta = (N, HW, C, G, nblk)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, g32 = ctx.saved_tensors
        N, HW, C, G, nblk = ctx.meta
        dev = dy.device
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty(C, device=dev, dtype=F32), torch.empty(C, device=dev, dtype=F32)
        gpart = torch.empty(N * nblk * G * 2, device=dev, dtype=F32)
        cpart = torch.empty(N * nblk * C * 2, device=dev, dtype=F32)
        gsum = torch.empty(N * G * 2, device=dev, dtype=F32)
        L.lib().wc_groupnorm_relu_bwd(L.ptr(x), L.ptr(y), L.ptr(dy, F32, "dy"), L.ptr(stats), L.ptr(g32), L.ptr(dx), L.ptr(dgamma),
                                      L.ptr(dbeta), L.ptr(gpart), L.ptr(cpart), L.ptr(gsum), N, HW, C, G, L.stream())
        return dx, dgamma, dbeta, None, None, None


def groupnorm_relu_rows(x_rows, gn, N):
    """relu(gn(x)) for an nn.GroupNorm module on rows (N*HW, C)."""
    return _GroupNormReluFn.apply(x_rows, gn.weight, gn.bias, N, gn.num_groups, gn.eps)
