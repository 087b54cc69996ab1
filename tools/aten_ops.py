"""Which torch (non-rpnet) GPU kernels does one training step launch, from where?  torch.profiler over one bench step,
grouped by operator + input shapes, with the Python call site (debug aid for removing element-wise launches).
Environment: B (8), SIZE (256), ITERS (5), WAYS (1), MATH (library default) — configs[4] is B=4 SIZE=512 ITERS=10 WAYS=2 MATH=f16."""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = int(os.environ.get("ITERS", "5"))
if os.environ.get("MATH"):
    RF.set_conv_math(os.environ["MATH"])
RF.set_async_wgrad(True)
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, int(os.environ.get("B", "8")), int(os.environ.get("SIZE", "256")), dev,
                        n_ways=int(os.environ.get("WAYS", "1")))
for _ in range(2):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.device_time_total > 0 and e.key.startswith("aten::"):
        stack = [s for s in e.stack if "rpnet_amd" in s or "bench.py" in s or "parallel.py" in s][:2]
        rows.append((e.count, e.device_time_total, e.key, str(e.input_shapes)[:90], " <- ".join(s.split("/")[-1] for s in stack)))
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print(f"aten ops with device time: {sum(r[0] for r in rows)} calls, {tot / 1e3:.2f} ms")
for r in rows[:70]:
    print(f"{r[0]:4d} x {r[1]:9.1f} us  {r[2]:28s} {r[3]:92s} {r[4]}")
# memcpy / memset activity of the step (runtime copies are not aten kernels: listed by kind and by the aten op that issued them)
mem = {}
for e in prof.events():
    if "Memcpy" in e.name or "Memset" in e.name:
        mem.setdefault(e.name, [0, 0.0])
        mem[e.name][0] += 1
        mem[e.name][1] += e.device_time
for k, v in mem.items():
    print(f"{v[0]:4d} x {v[1]:9.1f} us  {k}")
cp = {}
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    if e.key in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like"):
        stack = [s_ for s_ in e.stack if "rpnet_amd" in s_ or "bench.py" in s_ or "parallel.py" in s_][:2]
        cp[(e.key, str(e.input_shapes)[:60], " <- ".join(s_.split("/")[-1] for s_ in stack))] = e.count
for k, v in sorted(cp.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v:4d} x {k[0]:18s} {k[1]:62s} {k[2]}")
