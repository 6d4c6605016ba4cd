"""Where the HOST time of a graph-replayed training step goes (cProfile over N steps)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
args = A(); args.batch, args.size, args.classes_per_image, args.steps, args.warmup = 16, 512, 2, 10, 2
dev = torch.device("cuda", 0)
import __graft_entry__; __graft_entry__.build()
res, step, loader = bench.run_leg(args, dev, 0, 1, steps=5, warmup=2, graph=("--eager" not in sys.argv))
print(res)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); c0 = time.process_time()
pr.enable()
for _ in range(20):
    img, labels = loader.next()
    step(img, labels=labels)
pr.disable()
t1 = time.perf_counter() - t0; c1 = time.process_time() - c0
torch.cuda.synchronize()
print(f"20 steps: host wall {t1*50:.2f} ms/step, cpu {c1*50:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
