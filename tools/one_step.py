"""One warm-up step + N training steps of a given workload, nothing else (no probes, no legs): the smallest command a profiler can be
pointed at.  Usage: python tools/one_step.py --size 512 --ways 2 --iters 10 --batch 1 --conv-math f16 [--steps 1] [--serial]"""
import argparse
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--ways", type=int, default=1)
ap.add_argument("--shots", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--conv-math", default="f16x2")
ap.add_argument("--serial", action="store_true", help="everything on one stream")
ap.add_argument("--mask-skip", action="store_true", help="zero-tile skip of the masked CRE convolutions ON (the library's default; bench.py's "
                "headline and every roofline figure are measured with it OFF, and so is this script unless asked)")
a = ap.parse_args()
if a.serial:
    os.environ.update(RPNET_ASYNC_WGRAD="0", RPNET_CRE_STREAMS_TRAIN="0", RPNET_ENC_STREAMS="0")
import torch  # noqa: E402
import yaml  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = a.iters
RF.set_conv_math(a.conv_math)
RF._MASK_SKIP = bool(a.mask_skip)          # dense launches, as in bench.py's headline
# bench.py's default schedule: weight gradients on their side stream (RPNET_ASYNC_WGRAD=0 / --serial: through autograd on the main one)
RF.set_async_wgrad(not a.serial and os.environ.get("RPNET_ASYNC_WGRAD", "1") == "1")
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, a.batch, a.size, dev, a.shots, a.ways)
for i in range(1 + a.steps):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
    torch.cuda.synchronize()
    print(f"step {i} done", flush=True)
print("OK", float(bucket.flat.abs().max()), flush=True)
