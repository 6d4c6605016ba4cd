"""BASELINE configs[2] AS WRITTEN -- the WeCLIP VOC train step WITH the ViT-CoMer inserts -- at its full size: batch 16,
512 x 512, ViT-B/16-sized weights (VERDICT r03 "missing" 1).  No reference code exists for the inserts (SURVEY.md section 8
row a-9: parity unpinned), so what can be pinned at this size is (a) the fused engine (comer_engine.py) against the
module-by-module autograd form of the same network (comer.py + hip_functional.py), whose small-size agreement with an fp64
evaluation is tests/test_comer_gpu.py; (b) determinism: two runs bit-identical, the HIP-graph replay bit-identical to the
eager step."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, S = 16, 512


def _randomise_gates(net, seed=1):
    """The zero-initialised gates / offset / weight Linears get values so that every path carries signal."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for t in net.cti:
            t.gamma.copy_((torch.randn(t.gamma.shape, generator=g) * 0.5).to(t.gamma.device))
            for a in (t.to_v, t.to_c):
                a.sampling_offsets.weight.copy_((torch.randn(a.sampling_offsets.weight.shape, generator=g) * 0.02).to(t.gamma.device))
                a.attention_weights.weight.copy_((torch.randn(a.attention_weights.weight.shape, generator=g) * 0.05).to(t.gamma.device))
                a.attention_weights.bias.copy_((torch.randn(a.attention_weights.bias.shape, generator=g) * 0.2).to(t.gamma.device))


def test_comer_inserts_engine_vs_module_form_at_bench_size(monkeypatch):
    from weclip_vit_comer_amd.WeCLIP_model.comer import CoMerInteraction
    h = w = S // 16
    torch.manual_seed(0)
    net = CoMerInteraction(256).cuda()
    _randomise_gates(net)
    g = torch.Generator().manual_seed(2)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    maps0 = [torch.randn(B, h * w, 256, generator=g).cuda() if i in net.stage_blocks else None for i in range(11)]
    gy = torch.randn(B, 256, h, w, generator=g).cuda()
    res = {}
    for mode in ("0", "1", "1b"):
        monkeypatch.setenv("WECLIP_COMER_ENGINE", mode[0])
        for p in net.parameters():
            p.grad = None
        maps = [m.clone().requires_grad_(True) if m is not None else None for m in maps0]
        y = net(img, maps, (h, w))
        y.backward(gy)
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), [maps[b].grad.clone() for b in net.stage_blocks],
                     {n: p.grad.clone() for n, p in net.named_parameters()})
    rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    # two runs of the engine: bit-identical (fixed-order reductions, no float atomics)
    assert torch.equal(res["1"][0], res["1b"][0])
    assert all(torch.equal(a, b) for a, b in zip(res["1"][1], res["1b"][1]))
    assert all(torch.equal(res["1"][2][n], res["1b"][2][n]) for n in res["1"][2])
    ey = rel(res["1"][0], res["0"][0])
    em = max(rel(a, b) for a, b in zip(res["1"][1], res["0"][1]))
    ep = {n: rel(res["1"][2][n], v) for n, v in res["0"][2].items()}
    off = {n: v for n, v in ep.items() if "sampling_offsets" in n}
    rest = {n: v for n, v in ep.items() if "sampling_offsets" not in n}
    worst = sorted(rest.items(), key=lambda kv: -kv[1])[:3]
    print(f"CoMer inserts at B={B}, {S}x{S}: engine vs module form: output {ey:.1e}, d(adapter maps) {em:.1e}, parameter gradients "
          f"worst {worst}, sampling offsets worst {max(off.values()):.1e}")
    assert all(v.abs().max().item() > 0 for v in res["1"][2].values())
    assert ey < 1e-3, ey
    # each form is within ~1e-2 of an fp64 evaluation at small size (tests/test_comer_gpu.py: sums of a kinked bilinear derivative
    # under fp16 operand rounding); two such forms against each other, on the largest entry of 16 x 1024 x 256 gradients: 2.4e-2
    assert em < 4e-2, em
    # parameter gradients: the same two-forms-against-each-other allowance (measured worst 1.9e-2 ... 2.4e-2 on nc_q.weight, a sum of
    # 86 016 x 16 signed terms whose largest entry is small against the terms; it moves with any change of summation order upstream)
    assert all(v < 4e-2 for v in rest.values()), worst
    assert all(v < 0.15 for v in off.values()), off


def _step_run(graph, train, n=3, seed=0):
    import bench
    from weclip_vit_comer_amd.data import SyntheticVOCLoader
    from weclip_vit_comer_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    m = bench.make_model(dev, comer=True)
    _randomise_gates(m.comer)
    m.train(train)
    step = TrainStep(m, graph=graph)
    loader = SyntheticVOCLoader(B, S, 2, rank=0, world=1, device=dev, source="uint8")
    losses, grads = [], []
    for _ in range(n + (2 if graph else 0)):
        img, labels = loader.next()
        out = step(img, labels=labels)
        losses.append([o.item() for o in out])
        grads.append(step.bucket.flat.clone())
    if graph:
        assert all(e["graph"] is not None for e in step._graphs.values()) and len(step._graphs) == 1
        losses, grads = losses[2:], grads[2:]          # (the eager warm-up step and the capturing step of the graph mode)
    params = torch.cat([p.detach().flatten() for p in m.get_param_groups()[3]])
    del step, m
    torch.cuda.empty_cache()
    return losses, grads, params


def test_configs2_with_comer_step_is_reproducible_at_bench_size():
    """Two runs of the whole train step (train mode: Dropout2d on, head forked to the side stream, direct gradient writes into
    the all-reduce bucket) from the same seed: every loss, the whole gradient bucket and the parameters after the AdamW steps
    equal bit for bit."""
    l1, g1, p1 = _step_run(True, True)
    l2, g2, p2 = _step_run(True, True)
    assert l1 == l2, (l1, l2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert torch.equal(p1, p2)
    assert all(v == v for l in l1 for v in l) and g1[0].abs().max().item() > 0
    assert len({tuple(l) for l in l1}) == len(l1)


def test_configs2_with_comer_graph_replay_equals_eager_at_bench_size():
    """eval mode (no Dropout2d draw): the HIP-graph replay of forward + losses + backward equals the eager step bit for bit on
    the same batches (the loader is seeded: both runs see the same five / three batches in the same order)."""
    import bench  # noqa: F401
    le, ge, pe = _step_run(False, False, n=5)
    lg, gg, pg = _step_run(True, False, n=3)
    assert le[2:] == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(ge[2:], gg))
    assert torch.equal(pe, pg)
