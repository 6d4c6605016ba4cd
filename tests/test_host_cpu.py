"""CPU tests of the host-side logic (no GPU): pair planning, label reading, loss/label helpers vs the
oracle, optimizer schedule, and the data-parallel gradient bucket over gloo (world_size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import weclip_oracle as O


def test_pair_plan_indices():
    from weclip_vit_comer_amd.clip.clip_tool import PairPlan
    plan = PairPlan([[3, 7], [5]], n_fg=20, n_bg=25, device="cpu")
    assert plan.P == 3 and plan.K == 2 and plan.Tmax == 27
    assert plan.pair_img.tolist() == [0, 0, 1] and plan.pair_cls.tolist() == [0, 1, 0]
    assert plan.n_text.tolist() == [27, 27, 26]
    assert plan.text_idx[0, :3].tolist() == [3, 7, 20] and plan.text_idx[2, :2].tolist() == [5, 20]
    assert plan.valid_key.tolist() == [[0, 4, 8], [0, 6, 0]] and plan.nch.tolist() == [3, 2]
    with pytest.raises(RuntimeError):
        PairPlan([[]], 20, 25, "cpu")


def test_read_image_labels_uint8_wrap(tmp_path):
    """np.unique(uint8) - 1 wraps background 0 -> 255, which is why 255/254 are dropped
    (reference clip/clip_tool.py:111-118)."""
    from PIL import Image
    from weclip_vit_comer_amd.clip.clip_tool import read_image_labels
    png = np.zeros((30, 40), np.uint8)
    png[:5] = 4
    png[5:9] = 15
    png[-2:] = 255
    Image.fromarray(png).save(tmp_path / "a.png")
    ids, size = read_image_labels(str(tmp_path / "a.png"))
    assert ids == [3, 14] and tuple(size) == (30, 40)


def test_affinity_label_and_losses_match_oracle():
    from weclip_vit_comer_amd.utils.camutils import cams_to_affinity_label, get_mask_by_radius
    from weclip_vit_comer_amd.utils.losses import get_aff_loss, get_seg_loss
    g = torch.Generator().manual_seed(0)
    lab = torch.randint(0, 4, (2, 64, 96), generator=g)
    lab[0, :16, :16] = 255
    mask = get_mask_by_radius(4, 6, 2)
    assert np.array_equal(mask.numpy(), O.radius_mask(4, 6, 2).astype(np.float32))
    a = cams_to_affinity_label(lab, mask=mask)
    b = O.cams_to_affinity_label(lab, O.radius_mask(4, 6, 2))
    assert torch.equal(a, b)
    pred = torch.rand(2, 24, 24, generator=g)
    assert abs(get_aff_loss(pred, a)[0].item() - O.aff_loss(pred, b).item()) < 1e-7
    seg = torch.randn(2, 5, 64, 96, generator=g)
    assert abs(get_seg_loss(seg, lab).item() - O.seg_loss(seg, lab).item()) < 1e-6


def test_poly_warmup_schedule():
    from weclip_vit_comer_amd.utils.optimizer import PolyWarmupAdamW
    p = torch.nn.Parameter(torch.ones(3))
    opt = PolyWarmupAdamW([{"params": [p], "lr": 2e-3}], lr=2e-4, weight_decay=0.01, betas=[0.9, 0.999],
                          warmup_iter=50, max_iter=30000, warmup_ratio=1e-6, power=1.0)
    lrs = []
    for _ in range(60):
        p.grad = torch.ones(3)
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert abs(lrs[0] - 2e-3 * 1e-6) < 1e-12                                   # step 0: base * warmup_ratio
    assert abs(lrs[25] - 2e-3 * (1 - (1 - 25 / 50) * (1 - 1e-6))) < 1e-12
    assert abs(lrs[55] - 2e-3 * (1 - 55 / 30000)) < 1e-12                      # poly decay, power 1


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from weclip_vit_comer_amd.train_step import GradBucket
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    bucket = GradBucket(list(net.parameters()))
    bucket.zero()
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()                       # accumulates into the flat bucket views
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(bucket.flat, sum(gathered) / world) and all(
        p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    # every view starts on a 16-byte boundary: odd-sized tensors are followed by <= 3 zero padding elements
    padded = sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
    ret[rank] = bool(ok) and bucket.flat.numel() == padded and all(
        (p.grad.data_ptr() - bucket.flat.data_ptr()) % 16 == 0 for p in net.parameters())
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(0) and ret.get(1)


def test_pair_plan_update_refills_in_place():
    """TrainStep's graph mode keeps ONE PairPlan per batch signature and refills its device tensors in place."""
    from weclip_vit_comer_amd.clip.clip_tool import PairPlan
    plan = PairPlan([[3, 7], [5, 9]], n_fg=20, n_bg=25, device="cpu")
    ptrs = (plan.pair_img.data_ptr(), plan.text_idx.data_ptr(), plan.valid_key.data_ptr())
    assert PairPlan.signature([[3, 7], [5, 9]]) == (2, 4, 2)
    plan.update([[0, 1], [2, 19]])
    assert (plan.pair_img.data_ptr(), plan.text_idx.data_ptr(), plan.valid_key.data_ptr()) == ptrs
    assert plan.text_idx[2, :2].tolist() == [2, 19] and plan.valid_key.tolist() == [[0, 1, 2], [0, 3, 20]]
    ref = PairPlan([[0, 1], [2, 19]], 20, 25, "cpu")
    for a, b in ((plan.pair_img, ref.pair_img), (plan.pair_cls, ref.pair_cls), (plan.text_idx, ref.text_idx),
                 (plan.n_text, ref.n_text), (plan.nk, ref.nk), (plan.nch, ref.nch), (plan.valid_key, ref.valid_key)):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        plan.update([[0], [2, 19, 4]])            # another signature (same pair count, other K) needs its own plan
