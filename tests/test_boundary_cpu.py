"""CPU tests of the drop-in boundary: the reference scripts' own import lines resolve to this package from a
fresh interpreter, `bench.py --gpus N` starts N ranks, and the whole TrainStep host logic
keeps two gloo ranks in lock-step."""
import os
import subprocess
import sys
import textwrap

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, capture_output=True, text=True,
                          env=dict(os.environ, PYTHONPATH=ROOT))


def test_reference_import_lines_resolve_after_install_dropin():
    """scripts/dist_clip_voc.py:18-23 and test_msc_flip_coco.py:16-17 of the reference, verbatim, in a fresh process."""
    r = _run("""
        import weclip_vit_comer_amd
        weclip_vit_comer_amd.install_dropin()
        from utils.losses import get_aff_loss
        from utils import evaluate
        from utils.AverageMeter import AverageMeter
        from utils.camutils import cams_to_affinity_label
        from utils.optimizer import PolyWarmupAdamW
        from WeCLIP_model.model_attn_aff_voc import WeCLIP
        from WeCLIP_model.model_attn_aff_coco import WeCLIP as WeCLIPCoco
        from WeCLIP_model.PAR import PAR
        from WeCLIP_model.segformer_head import SegFormerHead
        from WeCLIP_model.Decoder.TransDecoder import DecoderTransformer
        import clip
        from clip.clip_tool import generate_cam_label, generate_clip_fts, perform_single_voc_cam, perform_single_coco_cam
        from clip.model import build_model
        from pytorch_grad_cam import GradCAM
        import weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc as real
        assert WeCLIP is real.WeCLIP, "alias and package must be ONE module object"
        import WeCLIP_model.model_attn_aff_voc as alias
        assert alias is real and alias.__spec__.name == "weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc"
        assert issubclass(WeCLIPCoco, WeCLIP)
        assert evaluate.scores and AverageMeter and get_aff_loss and cams_to_affinity_label and PolyWarmupAdamW
        print("DROPIN-OK")
    """)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout + r.stderr


def test_dropin_leaves_other_packages_alone_and_can_fall_back_to_a_reference_checkout(tmp_path):
    """Names this package does not provide are left to the regular finders; `reference_root` lets the helper
    modules of the user's checkout (utils/imutils.py ...) import next to the HIP-backed ones."""
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "imutils.py").write_text("MARK = 41\n")
    r = _run(f"""
        import weclip_vit_comer_amd
        weclip_vit_comer_amd.install_dropin(reference_root={str(tmp_path)!r})
        import json, numpy                      # unrelated imports still work
        from utils.imutils import MARK          # not in this package: comes from the checkout
        from utils.losses import get_aff_loss_fused   # in this package: the HIP-backed one
        try:
            import clip.does_not_exist
        except ImportError:
            print("FALLBACK-OK", MARK)
    """)
    assert r.returncode == 0 and "FALLBACK-OK 41" in r.stdout, r.stdout + r.stderr


def test_bench_gpus_n_spawns_n_ranks_before_any_gpu_call(monkeypatch):
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    started = []

    class P:
        def __init__(self, cmd, env=None):
            started.append((cmd, env))

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", P)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    inited = torch.cuda.is_initialized()
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert len(started) == 4
    assert [e["RANK"] for _, e in started] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["LOCAL_RANK"] == e["RANK"] and e["MASTER_ADDR"] == "127.0.0.1" for _, e in started)
    assert all(c[-4:] == ["--gpus", "4", "--steps", "3"] for c, _ in started)
    assert torch.cuda.is_initialized() == inited           # the parent made no GPU call


def test_bench_rejects_mismatched_world(monkeypatch):
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)


class _StubWeCLIP(torch.nn.Module):
    """CPU stand-in with WeCLIP's forward / get_param_groups contract (seg, cam_labels, attn_pred)."""
    seg_trans_after = 15000

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Conv2d(3, 8, 16, stride=16)
        self.pred = torch.nn.Conv2d(8, 5, 1)
        self.iter_num = 0

    def get_param_groups(self):
        return [[], [], [], list(self.parameters())]

    def forward(self, img, names=None, mode="train", labels=None):
        f = self.body(img)
        seg = self.pred(f)
        B, C, h, w = f.shape
        ff = f.reshape(B, C, h * w)
        ap = torch.sigmoid(ff.transpose(2, 1).bmm(ff))
        cam = (img[:, 0] > 0).long() * 2          # deterministic pseudo labels from the input
        return seg, cam, ap


def _train_worker(rank, world, port, bucket, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from weclip_vit_comer_amd.train_step import TrainStep
    torch.manual_seed(0)
    m = _StubWeCLIP()
    step = TrainStep(m, bucket=bucket)
    g = torch.Generator().manual_seed(100 + rank)
    losses = []
    for _ in range(3):
        img = torch.randn(2, 3, 64, 64, generator=g)
        losses.append(step(img, labels=[[0], [1]])[0].item())
    flat = torch.cat([p.detach().flatten() for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ret[rank] = (bool(torch.equal(gathered[0], gathered[1])), flat.clone(), losses)
    dist.destroy_process_group()


def _single_process_reference():
    """The same three steps on one process with both ranks' batches: gradient = mean of the two rank gradients."""
    sys.path.insert(0, ROOT)
    from weclip_vit_comer_amd.train_step import TrainStep, make_optimizer
    torch.manual_seed(0)
    m = _StubWeCLIP()
    opt = make_optimizer(m)
    helper = TrainStep(m, optimizer=opt, bucket=False)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    for _ in range(3):
        grads = None
        for g in gens:
            img = torch.randn(2, 3, 64, 64, generator=g)
            seg, cam, ap = m(img)
            loss, _, _ = helper.losses(seg, cam, ap)
            opt.zero_grad()
            loss.backward()
            cur = [p.grad.clone() for p in m.parameters()]
            grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
        for p, gr in zip(m.parameters(), grads):
            p.grad = gr / 2
        opt.step()
    return torch.cat([p.detach().flatten() for p in m.parameters()])


def _run_world2(bucket):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29700 + (os.getpid() + (7 if bucket else 0)) % 200
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, bucket, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return ret


def test_train_step_gloo_world2_matches_single_process_mean_gradient():
    """Whole TrainStep (forward, losses, backward, bucket all-reduce, PolyWarmupAdamW) on two gloo ranks with
    different batches: both ranks end with identical parameters, equal to single-process training on the mean
    gradient of the two batches."""
    ret = _run_world2(True)
    same, flat, _ = ret[0]
    assert same and ret[1][0]
    ref = _single_process_reference()
    assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-7), (flat - ref).abs().max()


def test_train_step_without_bucket_still_reduces_under_dp():
    """bucket=False used to skip the exchange silently (ranks diverged): it now all-reduces tensor by tensor."""
    ret = _run_world2(False)
    same, flat, _ = ret[0]
    assert same
    assert torch.allclose(flat, _single_process_reference(), rtol=1e-5, atol=1e-7)


def test_custom_ops_are_registered_with_the_dispatcher_and_have_no_cpu_kernel():
    """north_star: kernels "exposed to Python through PyTorch-ROCm custom ops".  Registration and the shape-propagating
    fake kernels are checked here; the GPU tests call the ops for real."""
    import pytest
    from torch._subclasses.fake_tensor import FakeTensorMode
    sys.path.insert(0, ROOT)
    import weclip_vit_comer_amd as W
    names = W.register_torch_ops()
    for n in names:
        assert hasattr(torch.ops.weclip, n), n
    with FakeTensorMode():
        img = torch.empty(2, 3, 64, 96, device="cuda")
        masks = torch.empty(2, 4, 64, 96, device="cuda")
        out = torch.ops.weclip.par_forward(img, masks, [1, 2, 4, 8, 12, 24], 20)
        assert tuple(out.shape) == (2, 4, 64, 96) and out.dtype == torch.float32
        lab = torch.ops.weclip.par_labels(masks, torch.empty(2, 4, device="cuda", dtype=torch.int64))
        assert tuple(lab.shape) == (2, 64, 96) and lab.dtype == torch.int64
        o, lse, mean = torch.ops.weclip.attention(torch.empty(100, 192, device="cuda", dtype=torch.float16), 2, 50, 2, 32, True)
        assert tuple(o.shape) == (100, 64) and tuple(lse.shape) == (2, 2, 50) and tuple(mean.shape) == (2, 50, 50)
        y = torch.ops.weclip.linear_f16(torch.empty(10, 64, device="cuda", dtype=torch.float16),
                                        torch.empty(7, 64, device="cuda", dtype=torch.float16), torch.empty(7, device="cuda"), 1)
        assert tuple(y.shape) == (10, 7)
        h = torch.ops.weclip.confusion_hist(torch.empty(5, 5, device="cuda", dtype=torch.int64),
                                            torch.empty(5, 5, device="cuda", dtype=torch.int64), 81)
        assert tuple(h.shape) == (81, 81) and h.dtype == torch.int64
    with pytest.raises(NotImplementedError):          # no CPU backend: the dispatcher refuses, nothing falls back
        torch.ops.weclip.par_forward(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8), [1], 1)
