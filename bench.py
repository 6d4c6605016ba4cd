#!/usr/bin/env python3
"""Benchmark of the WeCLIP hot path on MI355X.

A step = one full training step of `WeCLIP` (VOC head, 21 classes) on one synthetic VOC-shaped
batch: frozen CLIP ViT-B/16 encoder forward, adapters + decoder forward, block-12 forward +
GradCAM for K=2 classes per image, affinity refinement, PAR (20 iterations), losses, backward of
adapters/decoder, (DP: one RCCL all-reduce of the 5.99 M gradients), AdamW step.
BASELINE.json metric: images/sec (train fwd+bwd) ViT-B/16 512x512; per-GPU batch 16 (configs[2]).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline`: dominant kernel family timed with HIP events on the
launch stream during the timed steps; `cpu_baseline`: the CPU oracle (oracle/weclip_oracle.py, a
port of the reference's CPU path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes-per-image", type=int, default=2)
    ap.add_argument("--precision", default=None, choices=[None, "fast", "exact"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comer", action="store_true",
                    help="enable the ViT-CoMer inserts (no reference code exists for them; off = the reference model)")
    ap.add_argument("--timer-stride", type=int, default=7,
                    help="HIP-event pair around 1 of every n instrumented kernel launches (0: none, roofline = null)")
    ap.add_argument("--cpu-images", type=int, default=2, help="images in the bounded CPU-oracle sample")
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (port of the reference CPU path) on `cpu_images` images of the same workload:
    forward + losses + backward of the trainable heads."""
    from oracle import synth
    from oracle import weclip_oracle as O
    n = args.cpu_images
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    fuse, dec = synth.make_head_state_dicts()
    fuse = {k: v.requires_grad_(True) for k, v in fuse.items()}
    dec = {k: v.requires_grad_(True) for k, v in dec.items()}
    bg, fg = synth.make_text_features(20, 25, 512)
    img = synth.make_images(n, args.size, args.size, seed=100)
    labels = synth.make_label_lists(n, args.classes_per_image)
    t0 = time.time()
    seg, cam, ap = O.weclip_forward(img, labels, sd, fuse, dec, bg, fg, heads=12)
    loss, _, _ = O.train_losses(seg, cam, ap)
    loss.backward()
    dt = time.time() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} images {args.size}x{args.size}, K={args.classes_per_image}: oracle forward "
                      f"(encoder, GradCAM, affinity, PAR) + losses + head backward, {dt:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from weclip_vit_comer_amd import config, ops, synth
    from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
    from weclip_vit_comer_amd.train_step import TrainStep
    if args.precision:
        config.precision = args.precision

    dev = torch.device("cuda", local)
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse, dec = synth.make_head_state_dicts()
    model = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[768] * 4,
                   dataset_root_path=None, device=dev, text_features=(bg.to(dev), fg.to(dev)), comer=args.comer)
    model.decoder_fts_fuse.load_state_dict(fuse)
    model.decoder.load_state_dict(dec)
    model.train()
    step = TrainStep(model)
    B, S, K = args.batch, args.size, args.classes_per_image
    img = synth.make_images(B, S, S, seed=100 + rank).to(dev)
    labels = synth.make_label_lists(B, K, seed=7 + rank)

    for _ in range(args.warmup):
        step(img, labels=labels)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.KernelTimer.enable(args.timer_stride)       # HIP-event pair around every launch of the dominant kernels (csrc/core.hip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(img, labels=labels)
    t_enq = time.perf_counter() - t0          # host time to enqueue the steps (the GPU may still be running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    summ = ops.KernelTimer.summary()
    ops.KernelTimer.enable(0)
    stride = max(args.timer_stride, 1)
    # peaks from guides/MI355X_MICROARCH.md: dense fp16 MFMA 2.5 PFLOP/s, HBM3E 8 TB/s
    def peak_of(name):
        return ("hbm", 8000.0, "GB/s") if name.startswith("par_") else ("mfma", 2500.0, "TFLOP/s")
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (tools/refresh_profiles.sh + tools/pmc_traffic.py -> profiles/r01_traffic.json; 2*FETCH_SIZE + WRITE_SIZE, KiB,
    # per the MI355X guide); keyed by the same kernel names
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath) and B == 16 and S == 512 and K == 2:
        traffic = {k: round(v["hbm_bytes_per_launch"]) for k, v in json.load(open(tpath)).items()}
    roofs = []
    for name, r in summ.items():
        bound, peak, unit = peak_of(name)
        sec = r["ms"] * 1e-3
        if sec <= 0:
            continue
        ach = r["work"] / sec / (1e12 if bound == "mfma" else 1e9)
        roofs.append({"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                      "frac": round(ach / peak, 4), "traffic": traffic.get(name), "launches_timed": r["launches"],
                      "sampling": f"1 of {stride} launches",
                      "avg_launch_us": round(r["ms"] * 1e3 / max(r["launches"], 1), 2),
                      "share_of_step": round(r["ms"] * stride * 1e-3 / dt, 4)})
    roofs.sort(key=lambda x: -x["share_of_step"])
    out = {
        "metric": "images/sec (train fwd+bwd) ViT-B/16 512x512 VOC",
        "value": round(world * B * args.steps / dt, 3),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f16 MFMA operands / f32 accumulate, f32 residual+softmax+LN, PAR f32 arithmetic on 16-bit fixed-point "
                  "affinities (precision=fast)") if config.precision == "fast" else
                 "f16 hi+lo MFMA operands / f32 accumulate, f32 residual+softmax+LN+PAR (precision=%s)" % config.precision,
        "data": "synthetic",
        "config": {"workload": f"WeCLIP VOC full train step, batch {B}/GPU at {S}x{S}, K={K} classes/image "
                               f"(BASELINE configs[2]{'/[3] DP' if world > 1 else ''})" + (" + ViT-CoMer inserts" if args.comer else ""),
                   "global_batch": world * B, "parallelism": f"dp{world}"},
        "roofline": roofs[0] if roofs else None,
        "roofline_other": roofs[1:],
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
