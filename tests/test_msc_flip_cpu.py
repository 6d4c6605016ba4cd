"""CPU side of the multi-scale + flip inference path (BASELINE configs[4]): the oracle's
restatement of `validate` (test_msc_flip_coco.py:52-94) against the fixture produced by running the reference's own
`validate`, the histogram / score helpers against the reference's histograms, and the data-parallel histogram sum
over two gloo ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import synth
from oracle import weclip_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def coco_inputs(g):
    from make_golden import coco_inputs as ci      # pure synth code: the reference is only imported inside its make_* functions
    data = ci()
    assert abs(float(synth.checksum([d[1] for d in data])) - float(g["img_ck"])) <= 1e-9 * abs(float(g["img_ck"]))
    return data


def test_oracle_msc_flip_matches_reference_validate(golden):
    g = golden("tiny_coco_msc.npz")
    sd = synth.make_clip_state_dict(**synth.TINY)
    fuse, dec = synth.make_head_state_dicts(width=synth.TINY["width"], num_classes=81, seed=3)
    assert abs(float(synth.checksum(list(fuse.values()) + list(dec.values()))) - float(g["head_ck"])) <= 1e-9 * abs(float(g["head_ck"]))
    data = coco_inputs(g)
    hist = np.zeros((81, 81), np.int64)
    msc_hist = np.zeros((81, 81), np.int64)
    with torch.no_grad():
        fn = lambda x: O.seg_logits(x, sd, fuse, dec, heads=1)
        for i, (_, img, lab) in enumerate(data):
            p, mp_ = O.msc_flip_predict(fn, img[None], lab.shape, scales=(1.0, 0.75), resize_long=int(g["resize_long"]))
            assert np.array_equal(p.numpy().astype(np.uint8), g[f"pred{i}"])
            assert np.array_equal(mp_.numpy().astype(np.uint8), g[f"msc_pred{i}"])
            hist += O.fast_hist(lab.numpy(), p.numpy(), 81)
            msc_hist += O.fast_hist(lab.numpy(), mp_.numpy(), 81)
    assert np.array_equal(hist, g["hist"]) and np.array_equal(msc_hist, g["msc_hist"])


def test_host_histogram_and_scores_match_reference(golden):
    from weclip_vit_comer_amd.utils import evaluate
    g = golden("tiny_coco_msc.npz")
    data = coco_inputs(g)
    gts = [d[2].numpy().astype(np.int16) for d in data]
    preds = [g[f"msc_pred{i}"].astype(np.int16) for i in range(len(data))]
    hist, score = evaluate.scores(gts, preds, np.zeros((81, 81)), 81)
    assert np.array_equal(hist.astype(np.int64), g["msc_hist"])
    assert abs(score["miou"] - float(g["msc_miou"])) < 1e-12 and abs(score["pAcc"] - float(g["msc_pacc"])) < 1e-12
    assert hist.sum() == sum(int((x != 255).sum()) for x in gts)          # every non-ignored pixel counted once
    # pseudo_scores: 255 in the prediction means "ignore"
    lp = preds[0].copy()
    lp[:4] = 255
    s = evaluate.pseudo_scores([gts[0]], [lp], 81)
    assert 0.0 <= s["pAcc"] <= 1.0


def _hist_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from weclip_vit_comer_amd.msc_flip import reduce_hist, shard
    items = list(range(7))
    mine = shard(items, rank, world)
    g = torch.Generator().manual_seed(0)
    per_item = torch.randint(0, 50, (7, 5, 5), generator=g, dtype=torch.int64)      # same table on both ranks
    h = per_item[mine].sum(0)
    reduce_hist(h)
    ret[rank] = (mine, bool(torch.equal(h, per_item.sum(0))))
    dist.destroy_process_group()


def test_histogram_sum_over_two_gloo_ranks():
    """Replicas evaluate disjoint shards of the images; ONE int64 all-reduce gives every rank the full histogram."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_hist_worker, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret[0][0] == [0, 2, 4, 6] and ret[1][0] == [1, 3, 5]
    assert ret[0][1] and ret[1][1]
