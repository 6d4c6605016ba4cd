"""RCCL process group of ONE rank on the 1-GPU box + TrainStep(graph=True): exercises what the 8-GPU run does around the
graph (NCCL watchdog thread alive during the capture, all-reduce between the replay and the optimizer step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29431", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
import bench
class A: pass
a = A(); a.batch, a.size, a.classes_per_image, a.steps, a.warmup = 4, 256, 2, 6, 2
import __graft_entry__; __graft_entry__.build()
res, step, loader = bench.run_leg(a, torch.device("cuda", 0), 0, 1, steps=6, warmup=2, graph=True)
print("graph + RCCL(1 rank):", res)
dist.barrier(); dist.destroy_process_group()
print("DP-CHECK-OK")
