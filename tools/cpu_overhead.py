"""Host-side cost of enqueueing one training step (Python autograd + ctypes launches) against its GPU time:
if the two approach each other the step is launch-bound.  Usage: python tools/cpu_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench
import rpnet_amd.functional as RF
from rpnet_amd.parallel import FlatGradBucket

dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(bench.ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = 5
RF.set_async_wgrad(True)
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, 8, 256, dev, 1)
for _ in range(3):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
    enq.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
print(f"enqueue per step: min {min(enq)*1e3:.1f} ms  median {sorted(enq)[5]*1e3:.1f} ms;  wall per step {tot*1e3:.1f} ms")
# enqueue cost with the GPU idle at the start of every step (no back-pressure from a full queue)
enq2 = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
    enq2.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print(f"enqueue per step from an idle GPU: min {min(enq2)*1e3:.1f} ms  median {sorted(enq2)[2]*1e3:.1f} ms")
