"""Per-launch table of the GEMM launches of ONE training step (BASELINE configs[1], streams serialised): the duration of every 3x3 /
1x1 convolution, input-gradient and weight-gradient launch INSIDE the step (HIP events around the C-ABI call, the kernel in front
of it named) beside the duration of the SAME launch — same descriptor, same operands, still resident — repeated alone 12 times after
the step.  Answers VERDICT r05 item 2: where does the model's conv `frac` (0.465) stand below the isolated kernels' (0.51 - 0.56),
layer by layer.  Usage: python tools/layer_table.py [out.txt]      (env: B, SIZE, ITERS, MATH as tools/aten_ops.py)"""
import collections
import ctypes as C
import os
import sys

os.environ.update(RPNET_ASYNC_WGRAD="0", RPNET_CRE_STREAMS_TRAIN="0", RPNET_ENC_STREAMS="0")
import torch  # noqa: E402
import yaml  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = int(os.environ.get("ITERS", "5"))
if os.environ.get("MATH"):
    RF.set_conv_math(os.environ["MATH"])
RF.set_async_wgrad(False)
B, SIZE = int(os.environ.get("B", "8")), int(os.environ.get("SIZE", "256"))
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, B, SIZE, dev)
for _ in range(3):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()

# every tensor of the logged step stays alive (the launches are repeated on the same operands afterwards)
KEEP = []
for _n in ("empty", "zeros", "ones", "full", "empty_like", "zeros_like", "ones_like", "stack", "cat"):
    def _wrap(fn):
        def f(*a, **k):
            o = fn(*a, **k)
            KEEP.append(o)
            return o
        return f
    setattr(torch, _n, _wrap(getattr(torch, _n)))

GEMM = ("rpnet_conv_fwd", "rpnet_conv_up4", "rpnet_conv_wgrad", "rpnet_conv_wgrad_up4")
log, prev = [], ["-"]
orig = RF.call


def spy(name, *args):
    if name in GEMM:
        d = args[0]._obj               # the descriptor behind C.byref
        keep = type(d)()                 # a private copy of it
        C.memmove(C.addressof(keep), C.addressof(d), C.sizeof(d))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = orig(name, *args)
        b.record()
        Cin, Cout = d.C0 + d.C1, d.Co0 + d.Co1
        mode = args[1] if name == "rpnet_conv_up4" else None
        kind = {"rpnet_conv_fwd": "conv", "rpnet_conv_up4": "up4 fwd" if mode == 1 else "up4 dgrad", "rpnet_conv_wgrad": "wgrad",
                "rpnet_conv_wgrad_up4": "wgrad up4"}[name]
        if name in ("rpnet_conv_wgrad", "rpnet_conv_wgrad_up4") and args[1] is None:
            kind += " (reduce only)"
        log.append({"name": name, "kind": kind, "desc": keep, "args": args[1:], "ev": (a, b), "prev": prev[0],
                    "shape": (d.N, d.H, d.W, Cin, Cout, d.taps, d.upsample, d.split_planes),
                    "flop": 2.0 * d.N * d.H * d.W * Cin * Cout * (4 if "up4" in kind else d.taps)})
        prev[0] = name
        return r
    prev[0] = name
    return orig(name, *args)


RF.call = spy
bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()
RF.call = orig
rows = []
for e in log:
    e["insitu"] = e["ev"][0].elapsed_time(e["ev"][1]) * 1e3
for e in log:
    d = e["desc"]
    args = (C.byref(d),) + tuple(e["args"])
    for _ in range(2):
        orig(e["name"], *args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(12):
        orig(e["name"], *args)
    b.record()
    torch.cuda.synchronize()
    e["alone"] = a.elapsed_time(b) * 1e3 / 12
print(f"# tools/layer_table.py: B={B} SIZE={SIZE} T={cfg['n_iter_refinement']} arithmetic {RF.conv_math()}, streams serialised; {len(log)} GEMM launches of one step", file=out)
print("# in-step = HIP events around the C-ABI call inside the training step (incl. the weight-gradient reduce / the tile-flag launch where the call has one);", file=out)
print("# alone = the same call (same descriptor, same operands) 12 times back to back after the step; TF = algorithmic FLOPs of the launch (up4: the 4 executed taps)", file=out)
print(f"{'#':>3s} {'kind':18s} {'N x H x W':>14s} {'Cin':>5s} {'Cout':>5s} {'taps':>4s}  {'in-step us':>10s} {'alone us':>9s} {'ratio':>6s} {'TF in-step':>10s} {'TF alone':>9s}  in front of it", file=out)
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for i, e in enumerate(log):
    N, H, W, Cin, Cout, taps, ups, planes = e["shape"]
    tf = lambda us: e["flop"] / us / 1e6 if us > 0 else 0.0  # noqa: E731
    print(f"{i:3d} {e['kind']:18s} {f'{N}x{H}x{W}':>14s} {Cin:5d} {Cout:5d} {taps:4d}  {e['insitu']:10.1f} {e['alone']:9.1f} {e['insitu'] / e['alone']:6.2f} "
          f"{tf(e['insitu']):10.1f} {tf(e['alone']):9.1f}  {e['prev'].replace('rpnet_', '')}", file=out)
    k = "wgrad" if "wgrad" in e["kind"] else "conv fwd + dgrad"
    tot[k][0] += e["insitu"]; tot[k][1] += e["alone"]; tot[k][2] += e["flop"]
for k, (a, b, f) in tot.items():
    print(f"# {k}: in-step {a / 1e3:.3f} ms ({f / a / 1e6:.1f} TF), alone {b / 1e3:.3f} ms ({f / b / 1e6:.1f} TF), in-step / alone = {a / b:.3f}", file=out)
