"""TrainStep(graph=True): the HIP-graph replay of forward + losses + backward must produce bit-identical losses,
gradients and parameters to the eager path (same kernels, same launch order), for batches that change from step to
step (images and label lists are refilled in the graph's static buffers)."""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
H, W = synth.TINY_HW


def _model(train, comer=False):
    from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
    sd = synth.make_clip_state_dict(**synth.TINY)
    bg, fg = synth.make_text_features(20, 25, synth.TINY["embed_dim"])
    fuse, dec = synth.make_head_state_dicts(width=synth.TINY["width"])
    m = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[synth.TINY["width"]] * 4,
               dataset_root_path=None, device="cuda", text_features=(bg.cuda(), fg.cuda()), comer=comer)
    m.decoder_fts_fuse.load_state_dict(fuse)
    m.decoder.load_state_dict(dec)
    return m.train() if train else m.eval()


BATCHES = [(11, [[3, 7], [0, 14]]), (12, [[1, 2], [5, 19]]), (13, [[4, 6], [8, 9]]), (14, [[3, 7], [10, 11]]),
           (15, [[0, 1], [2, 3]])]


def _run(graph, seg_trans=False, comer=False):
    from weclip_vit_comer_amd.train_step import TrainStep
    torch.manual_seed(0)
    m = _model(train=False, comer=comer)     # eval: no Dropout2d, so eager and replayed steps see the same arithmetic
    if seg_trans:
        m.iter_num = 20000
    step = TrainStep(m, graph=graph)
    losses, grads = [], []
    for seed, labels in BATCHES:
        img = synth.make_images(2, H, W, seed=seed).cuda()
        out = step(img, labels=labels)
        losses.append([o.item() for o in out])
        grads.append(step.bucket.flat.clone())
    params = torch.cat([p.detach().flatten() for p in m.get_param_groups()[3]])
    return losses, grads, params, step, m


@pytest.mark.parametrize("seg_trans", [False, True])
def test_graph_replay_is_bit_identical_to_eager(seg_trans):
    le, ge, pe, _, me = _run(False, seg_trans)
    lg, gg, pg, step, mg = _run(True, seg_trans)
    assert len(step._graphs) == 1 and next(iter(step._graphs.values()))["graph"] is not None
    assert me.iter_num == mg.iter_num
    assert le == lg, (le, lg)
    for a, b in zip(ge, gg):
        assert torch.equal(a, b)
    assert torch.equal(pe, pg)
    assert len({tuple(l) for l in lg}) == len(lg)          # the batches really differ from step to step


def test_graph_replay_with_comer_inserts_is_bit_identical_to_eager():
    """The ViT-CoMer inserts (conv stem, MRFP, CTI with deformable attention: hip_functional autograd Functions + torch
    glue) are captured and replayed like the rest of the step."""
    le, ge, pe, _, _ = _run(False, comer=True)
    lg, gg, pg, step, _ = _run(True, comer=True)
    assert len(step._graphs) == 1 and next(iter(step._graphs.values()))["graph"] is not None
    assert le == lg, (le, lg)
    for a, b in zip(ge, gg):
        assert torch.equal(a, b)
    assert torch.equal(pe, pg)


def test_graph_mode_new_signature_gets_its_own_graph_and_dropout_varies():
    from weclip_vit_comer_amd.train_step import TrainStep
    torch.manual_seed(0)
    m = _model(train=True)
    step = TrainStep(m, graph=True)
    img = synth.make_images(2, H, W, seed=11).cuda()
    l2 = [step(img, labels=[[3, 7], [0, 14]])[0].item() for _ in range(4)]        # signature (2 images, 4 pairs, K=2)
    l3 = [step(img, labels=[[3, 7, 9], [0]])[0].item() for _ in range(3)]         # (2, 4, 3): another graph
    assert len(step._graphs) == 2
    assert all(v == v for v in l2 + l3)
    # same input every step: the loss still moves (optimizer updates + a fresh Dropout2d mask per replay)
    assert len(set(l2)) == len(l2)


def test_bucketed_signatures_keep_real_label_distributions_to_a_few_graphs():
    """The reference takes every image's classes from its GT mask (clip/clip_tool.py:111-124): 1..5 per image, so the pair
    count and the channel capacity change from batch to batch.  TrainStep(graph=True) buckets the signature (pairs padded to
    a multiple of 8 with dropped dummy pairs, K to the next even number): 50 random label lists need <= 8 captured graphs,
    and every loss, gradient and the final parameters equal the UNPADDED eager step bit for bit."""
    import numpy as np
    from weclip_vit_comer_amd.train_step import TrainStep
    rs = np.random.RandomState(0)
    batches = []
    for s in range(50):
        labels = [sorted(rs.choice(20, size=int(rs.randint(1, 6)), replace=False).tolist()) for _ in range(2)]
        batches.append((100 + s, labels))

    class NoStep:          # parameters stay put: a batch whose pseudo-labels hold no foreground pixel has a NaN loss (the
        def step(self):    # reference's CE over an empty set, scripts/dist_clip_voc.py:105-113) and must not poison the rest
            pass

        def zero_grad(self):
            pass

    def run(graph):
        torch.manual_seed(0)
        m = _model(train=False)
        step = TrainStep(m, optimizer=NoStep(), graph=graph)
        losses, grads = [], []
        for seed, labels in batches:
            out = step(synth.make_images(2, H, W, seed=seed).cuda(), labels=labels)
            losses.append([o.item() for o in out])
            grads.append(step.bucket.flat.clone())
        return losses, grads, torch.cat([p.detach().flatten() for p in m.get_param_groups()[3]]), step

    le, ge, pe, _ = run(False)
    lg, gg, pg, step = run(True)
    sigs = {(sum(len(l) for l in lab), max(len(l) for l in lab)) for _, lab in batches}
    captured = [e for e in step._graphs.values() if e["graph"] is not None]
    print(f"{len(sigs)} distinct (pairs, max classes) among 50 batches -> {len(step._graphs)} bucketed signatures, "
          f"{len(captured)} captured graphs")
    assert len(sigs) > 8 >= len(step._graphs)
    te, tg = torch.tensor(le, dtype=torch.float64), torch.tensor(lg, dtype=torch.float64)
    assert torch.isfinite(te).all(1).sum().item() >= 40
    assert torch.equal(torch.isnan(te), torch.isnan(tg)) and torch.equal(te.nan_to_num(), tg.nan_to_num()), (le, lg)
    for i, (a, b) in enumerate(zip(ge, gg)):
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(a.nan_to_num(), b.nan_to_num()), \
            (i, batches[i][1], (a - b).abs().max().item())
    assert torch.equal(pe, pg)


@pytest.mark.parametrize("graph", [False, True])
def test_forked_module_head_in_train_mode_equals_single_stream(graph):
    """with the ViT-CoMer inserts (module-form head) a TRAINING forward forks the head to the side stream, so its
    autograd nodes -- and the direct gradient writes into the all-reduce bucket -- run there and TrainStep joins the stream
    after backward().  Train mode (Dropout2d on), fixed seed, eager and graph mode: losses and the whole gradient bucket must
    equal the single-stream step bit for bit (a missing join / a racing bucket write would show here)."""
    from weclip_vit_comer_amd.train_step import TrainStep

    def run(fork):
        torch.manual_seed(0)
        m = _model(train=True, comer=True)
        with torch.no_grad():
            for t in m.comer.cti:
                t.gamma.fill_(0.3)
        m.fork_head = fork
        step = TrainStep(m, graph=graph)
        assert bool(m.comer.direct_grads)
        losses, grads, forked = [], [], []
        for seed, labels in BATCHES:
            out = step(synth.make_images(2, H, W, seed=seed).cuda(), labels=labels)
            losses.append([o.item() for o in out])
            grads.append(step.bucket.flat.clone())
            forked.append(bool(m.side_streams()))
        if graph:
            assert len(step._graphs) == 1 and next(iter(step._graphs.values()))["graph"] is not None
        return losses, grads, forked

    l1, g1, f1 = run(True)
    l0, g0, f0 = run(False)
    assert all(f1) and not any(f0)            # the fork really happened in one run and not in the other
    assert l1 == l0, (l1, l0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b), (a - b).abs().max().item()
    assert all(g.abs().max().item() > 0 for g in g1)
