"""Per-launch table of the conv kernels of ONE eval-mode call (test_rpnet.py call shape: batch B, T = 10, no_grad):
shape, arithmetic, HIP-event time and TFLOP/s of every rpnet_conv_fwd launch, plus the call's wall time.
Usage (GPU box): python tools/eval_layers.py [B] [size]"""
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
net = bench.build_model(cfg, dev)
net.eval()
net.num_iter = cfg.get("n_test_iter_refinement", 10)
si, fg, bg, qi, ql, appr = bench.make_inputs(77, B, size, dev)
with torch.no_grad():
    for _ in range(3):
        net(si, fg, bg, qi, appr_query_labels=appr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        net(si, fg, bg, qi, appr_query_labels=appr)
    torch.cuda.synchronize()
    print(f"batch {B} {size}x{size}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
    recs, orig = [], RF.call

    def timed(name, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = orig(name, *args)
        b.record()
        d = args[0]._obj if name == "rpnet_conv_fwd" else None
        recs.append((name, d and (d.N, d.H, d.W, d.C0 + d.C1, d.Co0 + d.Co1, d.taps, d.split_planes, d.upsample), a, b))
        return r
    RF.call = timed
    t0 = time.perf_counter()
    net(si, fg, bg, qi, appr_query_labels=appr)
    enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    RF.call = orig
tot = {}
for name, d, a, b in recs:
    ms = a.elapsed_time(b)
    tot.setdefault(name, [0, 0.0])
    tot[name][0] += 1
    tot[name][1] += ms
    if d:
        fl = 2.0 * d[0] * d[1] * d[2] * d[3] * d[4] * d[5]
        print(f"  conv N{d[0]} {d[1]}x{d[2]} {d[3]:4d}->{d[4]:4d} taps {d[5]} planes {d[6]} up {d[7]}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF")
print(f"enqueue of the instrumented call {enq:.2f} ms; per entry point (launches, ms):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:36s} {v[0]:4d} {v[1]:7.3f}")
