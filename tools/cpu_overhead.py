"""Host-side cost of enqueueing one training step (Python autograd + ctypes launches) against its GPU time:
if the two approach each other the step is launch-bound.
Usage: python tools/cpu_overhead.py [--size 512 --iters 10 --ways 2 --batch 4 --conv-math f16]   (default: the headline step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench
import rpnet_amd.functional as RF
from rpnet_amd.parallel import FlatGradBucket

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--shots", type=int, default=1)
ap.add_argument("--ways", type=int, default=1); ap.add_argument("--conv-math", default=None)
a_ = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(bench.ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = a_.iters
RF.set_async_wgrad(True)
if a_.conv_math:
    RF.set_conv_math(a_.conv_math)
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, a_.batch, a_.size, dev, a_.shots, a_.ways)
print(f"{a_.ways}-way {a_.shots}-shot {a_.size}x{a_.size} T={a_.iters} batch {a_.batch} [{RF.conv_math()}]")
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

# C-ABI calls of one step (each is one or a few kernel launches) and torch's own kernels (profiler)
calls = {}
orig = RF.call
def counting(name, *args):
    calls[name] = calls.get(name, 0) + 1
    return orig(name, *args)
from rpnet_amd import hip
RF.call = counting; hip_call = hip.call; hip.call = counting
try:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    n_rp = sum(1 for e in evs if "rpnet" in e.name)
    n_other = len(evs) - n_rp
    t_other = sum(e.device_time for e in evs if "rpnet" not in e.name) / 1e3
    print(f"kernel launches per step: {len(evs)} = {n_rp} rpnet + {n_other} torch / runtime ({t_other:.2f} ms); C-ABI calls {sum(calls.values())}")
    top = {}
    for e in evs:
        if "rpnet" not in e.name:
            k = e.name[:70]
            top[k] = top.get(k, 0) + 1
    for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {v:4d} x {k}")
finally:
    RF.call = orig; hip.call = hip_call
