"""Per-call timing of one bench step: every C-ABI call with its shape, duration and TFLOP/s
(HIP events on the launch stream).  Debug/tuning aid; prints a table sorted by time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench
from rpnet_amd import hip
import rpnet_amd.functional as RF
from rpnet_amd.parallel import FlatGradBucket

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("batch", type=int, nargs="?", default=8); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--shots", type=int, default=1)
ap.add_argument("--ways", type=int, default=1); ap.add_argument("--conv-math", default=None)
a_ = ap.parse_args()
B = a_.batch
cfg = yaml.load(open("yamls/example.yml"), Loader=yaml.FullLoader); cfg["n_iter_refinement"] = a_.iters
dev = torch.device("cuda", 0)
if a_.conv_math:
    RF.set_conv_math(a_.conv_math)
net = bench.build_model(cfg, dev); bucket = FlatGradBucket(net); inp = bench.make_inputs(1234, B, a_.size, dev, a_.shots, a_.ways)
for _ in range(2): bench.step(net, bucket, inp, 1.0)
torch.cuda.synchronize()
rec = []; orig = hip.call
def timed(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    info, fl = "", 0.0
    if name in ("rpnet_conv_fwd", "rpnet_conv_wgrad"):
        d = args[0]._obj
        fl = 2.0 * d.N * d.H * d.W * (d.C0 + d.C1) * (d.Co0 + d.Co1) * d.taps
        info = f"M={d.N*d.H*d.W:7d} Cin={d.C0+d.C1:4d} Cout={d.Co0+d.Co1:4d} taps={d.taps} ups={d.upsample} sc={d.in_scale_mode}{d.out_scale_mode}"
    a.record(); orig(name, *args); b.record(); rec.append((name, info, fl, a, b))
hip.call = timed; RF.call = timed
bench.step(net, bucket, inp, 1.0); torch.cuda.synchronize()
hip.call = orig; RF.call = orig
rows = {}
for name, info, fl, a, b in rec:
    r = rows.setdefault((name, info), [0, 0.0, 0.0]); r[0] += 1; r[1] += a.elapsed_time(b); r[2] += fl
tot = sum(r[1] for r in rows.values())
print(f"total kernel ms {tot:.2f}")
for (name, info), r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tf = r[2] / (r[1] * 1e-3) / 1e12 if r[2] else 0
    print(f"{r[1]:8.3f} ms {100*r[1]/tot:5.1f}% x{r[0]:3d} {tf:6.1f} TF  {name:28s} {info}")
