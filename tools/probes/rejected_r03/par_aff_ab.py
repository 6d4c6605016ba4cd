"""PAR set-up kernel A/B (WECLIP_PAR_AFF_LDS=0/1 in two processes): time of PAR.forward and a checksum of its output."""
import hashlib, os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from weclip_vit_comer_amd import synth
    from weclip_vit_comer_amd.WeCLIP_model.PAR import PAR
    B, C, H, W = 16, 3, 512, 512
    img = synth.make_images(B, H, W).cuda()
    masks = torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    mod = PAR([1, 2, 4, 8, 12, 24], 20).cuda()
    for _ in range(3):
        out = mod(img, masks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = mod(img, masks)
    e1.record(); torch.cuda.synchronize()
    print(f"WECLIP_PAR_AFF_LDS={os.environ.get('WECLIP_PAR_AFF_LDS')}: PAR.forward {e0.elapsed_time(e1) / 10:.3f} ms, sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]}")
else:
    for v in ("0", "1", "0", "1"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=dict(os.environ, WECLIP_PAR_AFF_LDS=v))
