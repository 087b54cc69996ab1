"""HIP-graph replay of the training step beside an RCCL communicator: does the ORDER of capture and dist.init_process_group matter?
Round 4 captured the step in a process that already held a one-rank RCCL group and saw a segmentation fault inside
hipStreamEndCapture in 3 of 12 runs (profiles/r04_graph_capture_under_rccl.txt).  Orders tried here, one fresh process each:
    before   build the model, CAPTURE the step (no communicator, no watchdog thread exists yet), then init_process_group("nccl",
             world 1), then 20 x { replay + all-reduce of the flat bucket }; gradients compared with one eager step
    after    init_process_group first, then capture (round 4's order; thread-local capture mode)
    python tools/graph_capture_order.py before|after   -> prints "OK <order> ..." and exits 0, or dies
tools/graph_capture_order.sh N runs each order N times and counts the exit codes -> profiles/r05_graph_capture_order.txt"""
import os
import sys

import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.graph import GraphedTrainStep  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

order = sys.argv[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def init_group():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)                      # the communicator and its watchdog exist from here on
    torch.cuda.synchronize()


if order == "after":
    init_group()
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = 5
RF.set_async_wgrad(True)
RF._MASK_SKIP = False
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net, force_active=True)
inp = bench.make_inputs(1234, 8, 256, dev)
scaler = cfg["align_loss_scaler"]
bench.step(net, bucket, inp, scaler)
torch.cuda.synchronize()
ref = bucket.flat.clone()                    # (order "before": no group yet, plain sums; world 1: the same numbers afterwards)


def loss_fn(out, ql):
    return RF.dice_ce_sum([out["output"], *out["refinement"].values()], ql) + scaler * out["align_loss"]


g = GraphedTrainStep(net, bucket, loss_fn)
si, fg, bg, qi, ql, appr = inp
g.capture(si, fg, bg, qi, ql, appr)
torch.cuda.synchronize()
if order == "before":
    init_group()
for _ in range(20):
    g(si, fg, bg, qi, ql, appr)
torch.cuda.synchronize()
same = torch.equal(bucket.flat, ref)
err = float((bucket.flat - ref).abs().max() / ref.abs().max())
print(f"OK {order}: captured, 20 replays + all-reduce beside the communicator; gradients equal to the eager step: {same} (max rel diff {err:.1e})", flush=True)
dist.destroy_process_group()
