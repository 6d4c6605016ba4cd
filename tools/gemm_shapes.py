"""Which GEMM shapes the training step launches and what each costs: WECLIP_GEMM_LOG=1 puts an event pair around every call of the
GEMM entry points; eager steps (no graph) are aggregated per (entry, M, N, K, segments, batch, kernel plan).
    python tools/gemm_shapes.py [comer]"""
import ctypes, os, sys
os.environ["WECLIP_GEMM_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from weclip_vit_comer_amd import _lib as L
from weclip_vit_comer_amd.data import SyntheticVOCLoader
from weclip_vit_comer_amd.train_step import TrainStep

comer = len(sys.argv) > 1 and sys.argv[1] == "comer"
dev = torch.device("cuda", 0)
model = bench.make_model(dev, comer=comer)
step = TrainStep(model, graph=False)
loader = SyntheticVOCLoader(16, 512, 2, rank=0, world=1, device=dev, source="uint8")
cd = L.lib().cdll
cd.wc_gemm_log_report.argtypes = [ctypes.c_char_p, ctypes.c_int]
cd.wc_gemm_log_report.restype = ctypes.c_int
buf = ctypes.create_string_buffer(1 << 20)
for _ in range(2):
    img, labels = loader.next()
    step(img, labels=labels)
cd.wc_gemm_log_report(buf, len(buf))      # drop the warm-up steps
n = 3
for _ in range(n):
    img, labels = loader.next()
    step(img, labels=labels)
cd.wc_gemm_log_report(buf, len(buf))
rows = []
for line in buf.value.decode().splitlines():
    key, calls, ms, flop = line.split("\t")
    rows.append((float(ms) / n, int(calls) / n, float(flop) / n, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"GEMM entry points, per step (eager, event pair around every call: each adds a few us of fencing): {tot:.3f} ms in {sum(r[1] for r in rows):.0f} calls")
for ms, calls, flop, key in rows:
    print(f"{ms:8.3f} ms {calls:5.1f} calls {ms / calls * 1e3:8.1f} us each {flop / ms / 1e9:8.1f} TF/s   {key}")
