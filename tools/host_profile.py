"""cProfile of the host side of the headline step (or of the eval call with --eval): where the enqueue time goes.
Usage (GPU box): python tools/host_profile.py [--eval] [--steps 20]"""
import argparse
import cProfile
import os
import pstats
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--eval", action="store_true")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--sort", default="tottime")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
net = bench.build_model(cfg, dev)
if a.eval:
    net.eval()
    net.num_iter = cfg.get("n_test_iter_refinement", 10)
    si, fg, bg, qi, ql, appr = bench.make_inputs(77, a.batch or 2, 256, dev)

    def one():
        with torch.no_grad():
            net(si, fg, bg, qi, appr_query_labels=appr)
else:
    import rpnet_amd.functional as RF
    RF.set_async_wgrad(True)          # the bench's step: weight gradients on the side streams, straight into the bucket
    bucket = FlatGradBucket(net)
    inp = bench.make_inputs(1234, a.batch or 8, 256, dev)

    def one():
        bench.step(net, bucket, inp, 1.0)
for _ in range(3):
    one()
torch.cuda.synchronize()
# backward on the calling thread, where cProfile can see it (the engine's device thread is not a Python thread)
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(a.sort)
print(f"{a.steps} calls; times below are totals over them")
st.print_stats(a.top)
