"""Backward building blocks of the trainable path (attention backward, LN backward, transposes,
column sums, sigmoid-Gram backward) vs torch autograd on the CPU in fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from weclip_vit_comer_amd import ops
    return ops


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.mark.parametrize("B,L,H,DH", [(2, 24, 8, 32), (1, 197, 12, 64), (2, 1024, 8, 32), (1, 130, 1, 64)])
def test_attention_backward_matches_autograd(ops, B, L, H, DH):
    E = H * DH
    g = torch.Generator().manual_seed(L + DH)
    qkv = torch.randn(B * L, 3 * E, generator=g)
    qs = (qkv[:, :E] * ops.q_scale(DH)).half()
    k, v = qkv[:, E:2 * E].half(), qkv[:, 2 * E:].half()
    packed = torch.cat([qs, k, v], 1).contiguous().cuda()
    do = (torch.randn(B * L, E, generator=g) * 0.1).half()
    o16, lse, _, o32 = ops.attention(packed, B, L, H, DH, want_mean=False, want_o32=True)
    d = ops.attention_bwd(packed, do.cuda(), o32, lse, B, L, H, DH)
    got = (d.hi.float() + d.lo.float()).cpu().double()
    # reference: unscaled q = qs / (log2e/sqrt(dh)) (exactly what the kernel differentiates w.r.t.)
    q0 = (qs.double() / ops.q_scale(DH)).requires_grad_(True)
    k0, v0 = k.double().requires_grad_(True), v.double().requires_grad_(True)
    sh = lambda t: t.view(B, L, H, DH).permute(0, 2, 1, 3)
    s = sh(q0) @ sh(k0).transpose(-1, -2) / DH ** 0.5
    o = (torch.softmax(s, -1) @ sh(v0)).permute(0, 2, 1, 3).reshape(B * L, E)
    (o * do.double()).sum().backward()
    ref = torch.cat([q0.grad, k0.grad, v0.grad], 1)
    for name, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
        e = _rel(got[:, sl], ref[:, sl])
        assert e < 6e-3, f"{name} rel err {e:.2e}"      # P, dS are rounded to fp16 before the second MFMA


@pytest.mark.parametrize("D", [64, 256, 768])
def test_layernorm_backward(ops, D):
    g = torch.Generator().manual_seed(D)
    rows = 333
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5)
    w = torch.randn(D, generator=g)
    dy = torch.randn(rows, D, generator=g)
    add = torch.randn(rows, D, generator=g)
    dx32, dx16, dgb = ops.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), add=add.cuda(), want16=True, out_scale=0.5,
                                        alpha=2.0)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = torch.zeros(D, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5)
    (y * dy.double()).sum().backward()
    assert _rel(dx32.cpu(), xr.grad + add.double()) < 1e-5
    assert _rel(dx16.float().cpu(), 0.5 * (xr.grad + add.double())) < 2e-3
    assert _rel(dgb[0].cpu(), 2 * wr.grad) < 1e-5 and _rel(dgb[1].cpu(), 2 * br.grad) < 1e-5


@pytest.mark.parametrize("D,rows", [(64, 333), (256, 1000), (256, 40000)])
def test_two_layernorms_of_one_input_backward_in_one_pass(ops, D, rows):
    """wc_layernorm_bwd2_h: y_a = LN(x; w_a), y_b = LN(x; w_b) (the CTI's two norms of c1) -> dx = dLN_a + dLN_b + add and the
    four parameter gradients, against fp64 autograd on the fp16-rounded gradients; and bit-for-bit the four column sums of the
    one-norm kernel (same summation order per norm)."""
    g = torch.Generator().manual_seed(D + rows)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5)
    wa, wb = torch.randn(D, generator=g), torch.randn(D, generator=g)
    dya, dyb = torch.randn(rows, D, generator=g).half(), torch.randn(rows, D, generator=g).half()
    add = torch.randn(rows, D, generator=g)
    dx32, dx16, ga, gb = ops.layernorm_bwd2(dya.cuda(), wa.cuda(), dyb.cuda(), wb.cuda(), x.cuda(), add=add.cuda(), want16=True,
                                            out_scale=0.5, alpha=2.0)
    xr = x.double().requires_grad_(True)
    war, wbr = wa.double().requires_grad_(True), wb.double().requires_grad_(True)
    bar, bbr = [torch.zeros(D, dtype=torch.float64, requires_grad=True) for _ in range(2)]
    ya = torch.nn.functional.layer_norm(xr, (D,), war, bar, 1e-5)
    yb = torch.nn.functional.layer_norm(xr, (D,), wbr, bbr, 1e-5)
    ((ya * dya.double()).sum() + (yb * dyb.double()).sum()).backward()
    assert _rel(dx32.cpu(), xr.grad + add.double()) < 1e-5
    assert _rel(dx16.float().cpu(), 0.5 * (xr.grad + add.double())) < 2e-3
    tol = 1e-5 if rows < 10000 else 1e-4          # fp32 column sums over 40 000 rows
    assert _rel(ga[0].cpu(), 2 * war.grad) < tol and _rel(ga[1].cpu(), 2 * bar.grad) < tol
    assert _rel(gb[0].cpu(), 2 * wbr.grad) < tol and _rel(gb[1].cpu(), 2 * bbr.grad) < tol
    _, _, ga1 = ops.layernorm_bwd(dya.cuda(), x.cuda(), wa.cuda(), alpha=2.0)
    _, _, gb1 = ops.layernorm_bwd(dyb.cuda(), x.cuda(), wb.cuda(), alpha=2.0)
    assert torch.equal(ga, ga1) and torch.equal(gb, gb1)


def test_transpose_colsum_and_wgrad(ops):
    """dW = dY^T X through transposes + the TN GEMM, db = colsum(dY)."""
    g = torch.Generator().manual_seed(0)
    Bn, R, Cy, Cx = 2, 75, 96, 130     # rows per batch not a multiple of 64 -> zero padded K
    dy = torch.randn(Bn * R, Cy, generator=g)
    xfull = torch.randn(Bn, R + 1, Cx, generator=g).half()      # x with a CLS row per batch to skip
    dyT, Kp = ops.transpose_f16(dy.cuda(), Bn * R, Cy, with_lo=True)
    xsrc = xfull.cuda()
    xT, Kp2 = ops.transpose_f16(xsrc.view(-1)[Cx:], R, Cx, ld=Cx, batch=Bn, sSrc=(R + 1) * Cx)
    assert Kp == Kp2 == 192
    out = torch.empty(Cy, Cx, device="cuda")
    ops.gemm(dyT, xT, Cy, Cx, Kp, out32=out, scale=0.25, scale_cols=Cx)
    ref = 0.25 * dy.double().t() @ xfull[:, 1:].reshape(Bn * R, Cx).double()
    assert _rel(out.cpu(), ref) < 1e-5
    b = ops.colsum(dy.cuda(), Bn * R, Cy, alpha=0.5)
    assert _rel(b.cpu(), 0.5 * dy.double().sum(0)) < 1e-5
    bh = ops.colsum(dy.half().cuda(), Bn * R, Cy, round16=True)
    assert _rel(bh.cpu(), dy.half().double().sum(0)) < 2e-3


def test_sigmoid_gram_backward_and_colscale(ops):
    g = torch.Generator().manual_seed(3)
    B, n, c = 2, 40, 64
    F = torch.randn(B, n, c, generator=g, dtype=torch.float64, requires_grad=True)
    AP = torch.sigmoid(F @ F.transpose(1, 2))
    dAP = torch.randn(B, n, n, generator=g, dtype=torch.float64)
    (AP * dAP).sum().backward()
    S = ops.sigmoid_gram_bwd(dAP.float().cuda(), AP.detach().float().cuda())
    Sf = (S.hi.float() + S.lo.float())[:, :, :n].cpu().double()
    assert _rel(Sf @ F.detach(), F.grad) < 1e-4
    x = torch.randn(B * n, c, generator=g)
    cs = torch.rand(B, c, generator=g)
    y32, ys = ops.colscale_split(x.cuda(), cs.cuda(), n)
    ref = x.view(B, n, c) * cs[:, None, :]
    assert _rel(y32.cpu(), ref.reshape(B * n, c)) < 1e-6
    assert _rel((ys.hi.float() + ys.lo.float()).cpu(), ref.reshape(B * n, c)) < 1e-6


def test_gemm_relu_grad_and_colscale_epilogues(ops):
    g = torch.Generator().manual_seed(4)
    Bn, M, N, K = 2, 70, 64, 64
    a = torch.randn(Bn * M, K, generator=g).half()
    w = torch.randn(N, K, generator=g).half()
    t1 = torch.randn(Bn * M, N, generator=g).half()
    out = torch.empty(Bn * M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), Bn * M, N, K, out32=out, act=5, auxh=t1.cuda(), ldaux=N)
    ref = (a.double() @ w.double().t()) * (t1.double() > 0)
    assert _rel(out.cpu(), ref) < 1e-5
    cs = torch.rand(Bn, N, generator=g)
    bias = torch.randn(N, generator=g)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), out32=out, batch=Bn, sA=M * K, sW=0, sC=M * N,
             cscale=cs.cuda(), sCS=N)
    ref2 = ((a.double() @ w.double().t() + bias.double()).view(Bn, M, N) * cs.double()[:, None, :]).reshape(Bn * M, N)
    assert _rel(out.cpu(), ref2) < 1e-5


@pytest.mark.parametrize("M,N,K,lda,slices", [(1000, 21, 200, 64, 4), (4096, 256, 768, None, 8), (70, 130, 64, 136, 1)])
def test_weight_gradient_gemm_row_major_operands(M, N, K, lda, slices):
    """dW = dY^T X and db = dY^T 1 straight from row-major fp16 operands (transposing LDS reads):
    ragged token count, ragged N / K tiles, padded dY rows, split-K slices."""
    from weclip_vit_comer_amd import ops
    g = torch.Generator().manual_seed(M + N)
    lda = N if lda is None else lda
    dy = torch.zeros(M, lda).half()
    dy[:, :N] = torch.randn(M, N, generator=g).half()
    x = torch.randn(M, K, generator=g).half()
    part, ns = ops.wgrad_partials(dy.cuda(), x.cuda(), M, N, K, lda=lda, slices=slices, bias=True)
    got = part.sum(0).cpu().double()
    ref = dy[:, :N].double().t() @ x.double()
    refb = dy[:, :N].double().sum(0)
    scale = ref.abs().max()
    assert (got[:, :K] - ref).abs().max() / scale < 2e-6
    assert (got[:, K] - refb).abs().max() / refb.abs().max() < 2e-6


def test_grouped_weight_gradient_and_slice_reduction():
    """wc_gemm_km_f16_grouped + wc_sum_slices_wb_grouped: G weight gradients of one shape in one launch each (the
    adapters): dY taken as column slices of one matrix, X from a stacked (G, B, 1 + hw, C) token buffer through the row
    map, results written at a uniform stride into one flat buffer (the gradient bucket layout)."""
    from weclip_vit_comer_amd import _lib as L, ops
    G, B, hw, C, N = 3, 2, 160, 192, 64
    M = B * hw
    g = torch.Generator().manual_seed(17)
    dy = torch.randn(M, G * N, generator=g).half()
    tok = torch.randn(G, B, 1 + hw, C, generator=g).half()
    part, ns = ops.wgrad_partials(dy.cuda(), tok.cuda(), M, N, C, lda=G * N, ldx=C, slices=3, bias=True, xmap=(hw, 1 + hw, 1),
                                  groups=G, gA=N, gX=B * (1 + hw) * C)
    assert tuple(part.shape) == (G, ns, N, C + 1)
    stride = N * C + N + 40                      # weight, bias, then unrelated parameters of the group
    flat = torch.full((G * stride,), 7.0, device="cuda")
    gw, gb = flat[:N * C], flat[N * C:N * C + N]
    L.lib().wc_sum_slices_wb_grouped(L.ptr(part, torch.float32), L.ptr(gw, torch.float32), L.ptr(gb, torch.float32), ns, N, C,
                                     0.5, G, stride, stride, L.stream())
    flat = flat.cpu().double()
    for gi in range(G):
        dyg = dy[:, gi * N:(gi + 1) * N].double()
        ref = 0.5 * dyg.t() @ tok[gi, :, 1:].reshape(M, C).double()
        refb = 0.5 * dyg.sum(0)
        o = gi * stride
        assert (flat[o:o + N * C].view(N, C) - ref).abs().max() / ref.abs().max() < 2e-6
        assert (flat[o + N * C:o + N * C + N] - refb).abs().max() / refb.abs().max() < 2e-6
        assert (flat[o + N * C + N:o + stride] == 7.0).all()          # nothing else touched


def test_weight_gradient_gemm_skips_cls_rows():
    """X = the patch rows of a (B, 1 + hw, C) token tensor, addressed through the row map."""
    from weclip_vit_comer_amd import ops
    B, hw, C, N = 3, 24, 64, 32
    g = torch.Generator().manual_seed(5)
    tok = torch.randn(B, 1 + hw, C, generator=g).half()
    dy = torch.randn(B * hw, N, generator=g).half()
    part, ns = ops.wgrad_partials(dy.cuda(), tok.view(-1, C).cuda(), B * hw, N, C, slices=1, bias=True, xmap=(hw, 1 + hw, 1))
    ref = dy.double().t() @ tok[:, 1:].reshape(B * hw, C).double()
    got = part.sum(0).cpu().double()
    assert (got[:, :C] - ref).abs().max() / ref.abs().max() < 2e-6


def test_fused_adamw_matches_torch_adamw():
    """PolyWarmupAdamW's one-launch HIP step vs torch.optim.AdamW driven with the same schedule, 5 steps,
    two parameter groups with different lr / weight decay."""
    from weclip_vit_comer_amd.utils.optimizer import PolyWarmupAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(257, 33), (64,), (5, 7, 3)]
    init = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (0.1 + k) for s in shapes] for k in range(5)]

    def make(dev):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        return ps, [{"params": ps[:2], "lr": 2e-3, "weight_decay": 0.01}, {"params": ps[2:], "lr": 5e-4, "weight_decay": 0.0}]

    ps_h, groups_h = make("cuda")
    opt_h = PolyWarmupAdamW(groups_h, lr=2e-4, weight_decay=0.01, betas=[0.9, 0.999], warmup_iter=3, max_iter=100,
                            warmup_ratio=1e-6, power=1.0)
    ps_t, groups_t = make("cpu")
    opt_t = PolyWarmupAdamW(groups_t, lr=2e-4, weight_decay=0.01, betas=[0.9, 0.999], warmup_iter=3, max_iter=100,
                            warmup_ratio=1e-6, power=1.0)      # CPU parameters -> stock torch.optim.AdamW path
    for k in range(5):
        for p, gr in zip(ps_h, grads[k]):
            p.grad = gr.cuda()
        for p, gr in zip(ps_t, grads[k]):
            p.grad = gr.clone().double().float()
        assert opt_h._hip_ok() and not opt_t._hip_ok()
        opt_h.step()
        opt_t.step()
    for a, b in zip(ps_h, ps_t):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())
    sa, sb = opt_h.state[ps_h[0]], opt_t.state[ps_t[0]]
    assert float(sa["step"]) == float(sb["step"]) == 5
    assert (sa["exp_avg_sq"].cpu() - sb["exp_avg_sq"]).abs().max().item() < 1e-6 * sb["exp_avg_sq"].abs().max().item()
