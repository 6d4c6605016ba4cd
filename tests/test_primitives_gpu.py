"""GEMM / LayerNorm / attention HIP kernels vs fp32 CPU references (through the C ABI)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from weclip_vit_comer_amd import ops
    return ops


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def test_mfma_layout_identity(ops):
    """A = I with an ASYMMETRIC W catches swapped row/col maps."""
    K = 64
    a = torch.eye(K, dtype=torch.float16, device="cuda")
    w = (torch.arange(96 * K, device="cuda").reshape(96, K) % 251).to(torch.float16)  # asymmetric ints
    out = torch.empty(K, 96, device="cuda")
    ops.gemm(a, w, K, 96, K, out32=out)
    assert torch.equal(out, w.float().t())


@pytest.mark.parametrize("M,N,K", [(200, 21, 64), (1025, 2304, 768), (333, 768, 3072), (128, 128, 128)])
def test_gemm_fp16_exact_products(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    out = torch.empty(M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), out32=out)
    assert _rel(out.cpu().double(), ref) < 2e-6          # fp32 accumulation of exact fp16 products
    # epilogue: QuickGELU + residual + fp16 hi/lo outputs
    out16 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    lo16 = torch.empty_like(out16)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), resid=res.cuda(), out32=out, out16=out16,
             out16lo=lo16, act=1)
    ref2 = ref * torch.sigmoid(1.702 * ref) + res.double()
    assert _rel(out.cpu().double(), ref2) < 2e-6
    assert _rel((out16.float() + lo16.float()).cpu().double(), ref2) < 2e-6
    # forced-fp16 rounding before the residual (out-projection semantics)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), resid=res.cuda(), out32=out, round16=True)
    ref3 = (a.double() @ w.double().t() + bias.double()).float().half().double() + res.double()
    assert ((out.cpu().double() - ref3).abs() > 1e-6).float().mean().item() < 2e-3   # fp16 rounding ties


def test_gemm_split_precision_and_scale(ops):
    M, N, K = 300, 192, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.04
    ref = x.double() @ w.double().t()
    out = torch.empty(M, N, device="cuda")
    ops.gemm(ops.split_f16(x.cuda()), ops.split_f16(w.cuda()), M, N, K, out32=out)
    e1 = _rel(out.cpu().double(), ref)
    ops.gemm(ops.split_f16(x.cuda(), True), ops.split_f16(w.cuda(), True), M, N, K, out32=out)
    e3 = _rel(out.cpu().double(), ref)
    assert e1 < 2e-3 and e3 < 5e-6, (e1, e3)
    ops.gemm(ops.split_f16(x.cuda(), True), ops.split_f16(w.cuda(), True), M, N, K, out32=out,
             scale=0.5, scale_cols=64)
    ref[:, :64] *= 0.5
    assert _rel(out.cpu().double(), ref) < 5e-6


def test_gemm_tall_pingpong(ops):
    """Tall GEMMs (>= 160 tiles of 256x256) take the 256x256 ping-pong kernel: ragged M/N edges, odd and even
    K-tile counts, wide (fp16-only) and narrow (fp32 + residual) epilogues, 2 K-segments, ReLU' aux path."""
    M, N, K = 8300, 1300, 192
    g = torch.Generator().manual_seed(7)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    ac, wcu, bc = a.cuda(), w.cuda(), bias.cuda()
    out16 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    lo16 = torch.zeros_like(out16)
    ops.gemm(ac, wcu, M, N, K, bias=bc, out16=out16, out16lo=lo16, act=1)
    ref1 = ref * torch.sigmoid(1.702 * ref)
    assert _rel((out16.float() + lo16.float()).cpu().double(), ref1) < 2e-6
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(ac, wcu, M, N, K, bias=bc, resid=res.cuda(), out32=out)
    assert _rel(out.cpu().double(), ref + res.double()) < 2e-6
    # two K-segments (hi/lo activation against the same weight) through the same stream
    x = torch.randn(M, K, generator=g)
    sp = ops.split_f16(x.cuda(), True)
    ops.gemm(sp, ops.Split(wcu, None), M, N, K, out32=out)
    assert _rel(out.cpu().double(), x.double() @ w.double().t()) < 5e-6
    # ReLU backward from a saved fp16 activation (act 5)
    saved = torch.randn(M, N, generator=g).half()
    ops.gemm(ac, wcu, M, N, K, out32=out, act=5, auxh=saved.cuda(), ldaux=N)
    assert _rel(out.cpu().double(), (a.double() @ w.double().t()) * (saved.double() > 0)) < 2e-6
    # long K (many ring revolutions) against the fp32-accumulating reference
    K2 = 1536
    a2 = torch.randn(M, K2, generator=g).half()
    w2 = (torch.randn(N, K2, generator=g) * 0.05).half()
    ops.gemm(a2.cuda(), w2.cuda(), M, N, K2, out32=out)
    assert _rel(out.cpu().double(), a2.double() @ w2.double().t()) < 2e-6


@pytest.mark.parametrize("nt", list(range(2, 15)) + [17, 23])
def test_gemm_tall_ring_every_tail_length(ops, nt):
    """The 256x256 kernel's K loop is unrolled over 5 K-tiles (10-slot LDS ring) with a guarded tail: every number of K-tiles
    from 2 to 14 (all tail lengths on both sides of the steady loop's entry condition) and two longer odd ones, against the
    fp64 product, and against the erf build of the same kernel (8-slot ring, 2-K-tile unroll: another tail code path) bit for
    bit in front of the activation (`pre32`)."""
    from weclip_vit_comer_amd import _lib as L
    M, N, K = 20480 + 16, 512, 64 * nt
    g = torch.Generator().manual_seed(nt)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    assert L.lib().cdll.wc_gemm_plan(M, N, K, 1, 1) in (1, 2)
    o = torch.zeros(M, N, device="cuda")
    ops.gemm(a, w, M, N, K, bias=bias, out32=o)
    pre, o6 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda", dtype=torch.float16)
    ops.gemm(a, w, M, N, K, bias=bias, pre32=pre, out16=o6, act=6)
    torch.cuda.synchronize()
    mm = M // 256 * 256
    assert torch.equal(o[:mm], pre[:mm])
    ref = a.double() @ w.double().t() + bias.double()
    assert _rel(o.double(), ref) < 2e-6


def test_gemm_ragged_rows_split(ops):
    """65 x 4 tiles of 256x256 on 256 CUs: the 16 ragged rows go to a second launch (128x128 kernel);
    residual / fp16 hi+lo outputs / saved pre-activation must line up across the seam."""
    M, N, K = 64 * 256 + 16, 1024, 128
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    out = torch.zeros(M, N, device="cuda")
    pre = torch.zeros(M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), resid=res.cuda(), out32=out, pre32=pre, act=1)
    assert _rel(pre.cpu().double(), ref) < 2e-6
    assert _rel(out.cpu().double(), ref * torch.sigmoid(1.702 * ref) + res.double()) < 2e-6
    hi = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    lo = torch.zeros_like(hi)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), out16=hi, out16lo=lo)
    assert _rel((hi.float() + lo.float()).cpu().double(), ref) < 2e-6
    # QuickGELU' from a row-mapped pre-activation (GradCAM: 16 (image, class) pairs of 1025 rows reading 8 images)
    rpg = 1025
    rowmap = torch.tensor([3, 3, 0, 7, 1, 1, 2, 5, 6, 6, 4, 0, 2, 7, 5, 4], dtype=torch.int32)
    u = torch.randn(8 * rpg, N, generator=g)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, out32=out, act=4, aux=u.cuda(), rowmap=rowmap.cuda(), rpg=rpg, ldaux=N)
    um = u.view(8, rpg, N)[rowmap.long()].reshape(M, N).double()
    sg = torch.sigmoid(1.702 * um)
    ref4 = (a.double() @ w.double().t()) * (sg * (1 + 1.702 * um * (1 - sg)))
    assert _rel(out.cpu().double(), ref4) < 2e-6


@pytest.mark.parametrize("M,N,K", [(16, 3072, 768), (32, 768, 3072), (5, 300, 192)])
def test_gemm_few_rows(ops, M, N, K):
    """M <= 32 rows against N >= 256 weight rows take the K-split kernel without LDS staging (the ragged rows of the tall
    encoder GEMMs): plain, QuickGELU + residual + hi/lo outputs, three K-segments (split precision), ragged N."""
    g = torch.Generator().manual_seed(M * N)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), out32=out)
    assert _rel(out.cpu().double(), ref) < 2e-6
    hi = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    lo = torch.zeros_like(hi)
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), resid=res.cuda(), out32=out, out16=hi, out16lo=lo, act=1)
    ref2 = ref * torch.sigmoid(1.702 * ref) + res.double()
    assert _rel(out.cpu().double(), ref2) < 2e-6
    assert _rel((hi.float() + lo.float()).cpu().double(), ref2) < 2e-6
    ops.gemm(a.cuda(), w.cuda(), M, N, K, bias=bias.cuda(), out16=hi)           # fp16-only output: the wide epilogue
    assert _rel(hi.float().cpu().double(), ref) < 1e-3
    x = torch.randn(M, K, generator=g)
    w32 = torch.randn(N, K, generator=g) * 0.04
    ops.gemm(ops.split_f16(x.cuda(), True), ops.split_f16(w32.cuda(), True), M, N, K, out32=out)
    assert _rel(out.cpu().double(), x.double() @ w32.double().t()) < 5e-6


def test_gemm_grouped_two_level_batch(ops):
    """wc_gemm_f16_grouped: batch index z = group * zdiv + member.  (a) groups x members with per-group weights / biases
    and outputs written side by side (the adapters' first Linear: blocks x images); (b) groups only, A taken as column
    slices of one matrix, act 5 with a per-group fp16 aux (the adapters' ReLU backward)."""
    g = torch.Generator().manual_seed(21)
    G, Bm, Mi, N, K = 3, 4, 192, 128, 192             # groups, members, rows per member
    a = torch.randn(G, Bm, Mi + 1, K, generator=g).half()        # one extra leading row per member (like the CLS row)
    w = (torch.randn(G, N, K, generator=g) * 0.05).half()
    bias = torch.randn(G, N, generator=g)
    out = torch.zeros(Bm * Mi, G * N, device="cuda", dtype=torch.float16)
    ad = a.cuda()
    ops.gemm(ad.view(-1)[K:], w.cuda(), Mi, N, K, bias=bias.cuda(), out16=out, act=2, batch=G * Bm, zdiv=Bm,
             sA=(Mi + 1) * K, sA2=Bm * (Mi + 1) * K, sW=0, sW2=N * K, sC=Mi * G * N, sC2=N, sB2=N, ldc=G * N)
    for gi in range(G):
        ref = torch.relu(a[gi, :, 1:].double() @ w[gi].double().t() + bias[gi].double()).reshape(Bm * Mi, N)
        got = out[:, gi * N:(gi + 1) * N].cpu().double()
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), gi
    # (b)
    M = 512
    dcat = torch.randn(M, G * N, generator=g).half()
    w2t = (torch.randn(G, N, N, generator=g) * 0.05).half()          # (out, k) per group
    t1 = torch.randn(G, M, N, generator=g).half()
    dt1 = torch.zeros(G, M, N, device="cuda", dtype=torch.float16)
    ops.gemm(dcat.cuda(), w2t.cuda(), M, N, N, lda=G * N, out16=dt1, act=5, auxh=t1.cuda(), ldaux=N, batch=G, zdiv=1,
             sA2=N, sW2=N * N, sC2=M * N, sX2=M * N)
    for gi in range(G):
        ref = (dcat[:, gi * N:(gi + 1) * N].double() @ w2t[gi].double().t()) * (t1[gi].double() > 0)
        assert (dt1[gi].cpu().double() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), gi


def test_gemm_batched(ops):
    Bn, M, N, K = 3, 130, 70, 64
    g = torch.Generator().manual_seed(1)
    a = torch.randn(Bn, M, K, generator=g).half()
    w = torch.randn(Bn, N, K, generator=g).half()
    out = torch.empty(Bn, M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), M, N, K, out32=out, batch=Bn, sA=M * K, sW=N * K, sC=M * N, act=3)
    ref = torch.sigmoid(torch.bmm(a.double(), w.double().transpose(1, 2)))
    assert _rel(out.cpu().double(), ref) < 2e-6


@pytest.mark.parametrize("D", [64, 256, 768])
def test_layernorm(ops, D):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(1000, D, generator=g) * 3 + 1
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    y32, sp = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), want32=True, with_lo=True)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    assert (y32.cpu() - ref).abs().max().item() < 5e-6
    assert ((sp.hi.float() + sp.lo.float()).cpu() - ref).abs().max().item() < 5e-6
    assert torch.equal(sp.hi.cpu(), y32.cpu().half())


@pytest.mark.parametrize("B,L,H,DH", [(2, 25, 1, 64), (1, 197, 12, 64), (2, 1025, 12, 64), (2, 1024, 8, 32),
                                      (1, 130, 8, 32), (1, 257, 2, 32), (2, 136, 3, 64), (1, 137, 2, 64)])
def test_attention_vs_reference(ops, B, L, H, DH):
    """Same arithmetic as clip/myAtt.py:21-64,325-326 on fp16-rounded q,k,v."""
    E = H * DH
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B * L, 3 * E, generator=g)
    qkv[:, :E] *= 1.5
    q = (qkv[:, :E] * ops.q_scale(DH)).half()
    k, v = qkv[:, E:2 * E].half(), qkv[:, 2 * E:].half()
    packed = torch.cat([q, k, v], 1).contiguous()
    o16, lse, mean = ops.attention(packed.cuda(), B, L, H, DH)
    qf = q.float().view(B, L, H, DH).permute(0, 2, 1, 3).double() / ops.LOG2E
    kf = k.float().view(B, L, H, DH).permute(0, 2, 1, 3).double()
    vf = v.float().view(B, L, H, DH).permute(0, 2, 1, 3).double()
    s = qf @ kf.transpose(-1, -2)
    p = torch.softmax(s, -1)
    o = (p @ vf).permute(0, 2, 1, 3).reshape(B * L, E)
    np.testing.assert_allclose(mean.cpu().double().numpy(), p.mean(1).numpy(), rtol=2e-4, atol=1e-7)
    assert abs(mean.sum().item() / (B * L) - 1.0) < 1e-4                    # rows sum to 1
    ref_lse = torch.logsumexp(s, -1) * ops.LOG2E
    assert (lse.cpu().double() - ref_lse).abs().max().item() < 1e-4
    # P is rounded to fp16 before PV and O to fp16 on output
    assert (o16.cpu().double() - o).abs().max().item() < 4e-3 * max(1.0, o.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(40013, 256, 256), (86016, 256, 256), (4100, 256, 128), (16384, 128, 256), (9000, 96, 256),
                                   (777, 256, 256), (16, 64, 128)])
def test_gemm_row_kernel_every_epilogue(ops, M, N, K):
    """csrc/gemm_row.hip (round 4: weights stationary in registers, 16-row tiles streamed through LDS, whole-row epilogue)
    against the fp64 product and, where the tile kernels cover the same epilogue, against wc_gemm_f16 on the same operands:
    plain fp32 / fp16 outputs, bias + column scale + fp32 residual, ReLU, ReLU' from the saved fp16 output, exact GELU with
    the saved pre-activation, GELU' from it, and the fused LayerNorm outputs (bit-identical to layernorm_kernel on the fp32
    output).  Persistent workgroups (more tiles than 2 x CUs), ragged row counts, N < 256 (idle waves / lanes), K = 128."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.06).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    cs = (torch.rand(N, generator=g) + 0.5).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    saved = torch.randn(M, N, generator=g).half().cuda()
    u = torch.randn(M, N, generator=g).cuda()
    ref = (a.double() @ w.double().t()).cpu()
    refb = ref + bias.double().cpu()
    z32 = lambda: torch.zeros(M, N, device="cuda")
    z16 = lambda: torch.zeros(M, N, device="cuda", dtype=torch.float16)
    # plain
    o32, o16 = z32(), z16()
    ops.gemm_row(a, w, M, N, K, out32=o32, out16=o16)
    assert _rel(o32.cpu().double(), ref) < 2e-6
    assert torch.equal(o16, o32.half())
    # bias * column scale + residual
    o = z32()
    ops.gemm_row(a, w, M, N, K, bias=bias, cscale=cs, resid=res, out32=o)
    assert _rel(o.cpu().double(), refb * cs.double().cpu() + res.double().cpu()) < 2e-6
    t = z32()
    ops.gemm(a, w, M, N, K, bias=bias, cscale=cs.view(1, N), sCS=0, resid=res, out32=t)
    assert (o - t).abs().max().item() <= 2e-5 * max(1.0, t.abs().max().item())        # (another fp32 summation order)
    # ReLU with fp16 output; ReLU' from a saved fp16 tensor
    h = z16()
    ops.gemm_row(a, w, M, N, K, bias=bias, act=2, out16=h)
    assert _rel(h.float().cpu().double(), refb.clamp_min(0)) < 1e-3
    if N % 8 == 0:
        o = z32()
        ops.gemm_row(a, w, M, N, K, act=5, auxh=saved, ldaux=N, out32=o)
        assert _rel(o.cpu().double(), ref * (saved.double().cpu() > 0)) < 2e-6
    # exact GELU with the saved pre-activation, and its derivative
    pre, gl = z32(), z16()
    ops.gemm_row(a, w, M, N, K, bias=bias, act=6, pre32=pre, out16=gl)
    assert _rel(pre.cpu().double(), refb) < 2e-6
    assert _rel(gl.float().cpu().double(), torch.nn.functional.gelu(refb)) < 1e-3
    o = z32()
    ops.gemm_row(a, w, M, N, K, act=7, aux=u, ldaux=N, out32=o)
    ud = u.double().cpu()
    dg = 0.5 * (1 + torch.erf(ud / 2 ** 0.5)) + ud * torch.exp(-0.5 * ud * ud) / (2 * torch.pi) ** 0.5
    assert _rel(o.cpu().double(), ref * dg) < 2e-6
    # fused LayerNorm outputs: bit-identical to the LayerNorm kernel run on the fp32 output
    if N == 256:
        g0, b0 = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
        g1, b1 = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
        o, l0, l1 = z32(), z16(), z16()
        ops.gemm_row(a, w, M, N, K, bias=bias, resid=res, out32=o, ln=[(g0, b0, l0), (g1, b1, l1)], eps=1e-5)
        assert _rel(o.cpu().double(), refb + res.double().cpu()) < 2e-6
        for gam, bet, got in ((g0, b0, l0), (g1, b1, l1)):
            want = ops.layernorm(o, gam, bet, eps=1e-5, want32=False, want16=True)[1].hi
            assert torch.equal(got, want)
        only = z16()
        ops.gemm_row(a, w, M, N, K, bias=bias, resid=res, ln=[(g0, b0, only)])      # LayerNorm output alone
        assert torch.equal(only, l0)
    # determinism
    o2 = z32()
    ops.gemm_row(a, w, M, N, K, out32=o2)
    assert torch.equal(o2, o32)


def test_gemm_row_grouped_matches_per_group_products(ops):
    """wc_gemm_row_f16_grouped (the eleven adapters' second Linear and its input gradient in one launch each): every group against
    the fp64 product of its own operands -- outputs dropped into column slices of a wider buffer (forward), ReLU' from the saved
    fp16 activations with the A operand read from column slices (backward)."""
    G, M, E = 5, 4100, 256
    g = torch.Generator().manual_seed(7)
    t1 = torch.randn(G, M, E, generator=g).half().cuda()
    w = (torch.randn(G, E, E, generator=g) * 0.06).half().cuda()
    b = torch.randn(G, E, generator=g).cuda()
    cat = torch.zeros(M, G * E, device="cuda", dtype=torch.float16)
    ops.gemm_row_grouped(t1, w, M, E, E, G, bias=b, out16=cat, ldc16=G * E, gA=M * E, gW=E * E, gB=E, gC=E)
    for i in range(G):
        ref = t1[i].double().cpu() @ w[i].double().cpu().t() + b[i].double().cpu()
        assert _rel(cat[:, i * E:(i + 1) * E].float().cpu().double(), ref) < 1e-3, i
    dcat = torch.randn(M, G * E, generator=g).half().cuda()
    dt1 = torch.zeros(G, M, E, device="cuda", dtype=torch.float16)
    ops.gemm_row_grouped(dcat, w, M, E, E, G, lda=G * E, out16=dt1, act=5, auxh=t1, ldaux=E, gA=E, gW=E * E, gC=M * E, gX=M * E)
    for i in range(G):
        ref = (dcat[:, i * E:(i + 1) * E].double().cpu() @ w[i].double().cpu().t()) * (t1[i].double().cpu() > 0)
        assert _rel(dt1[i].float().cpu().double(), ref) < 1e-3, i
