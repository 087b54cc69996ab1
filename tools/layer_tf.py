"""Per-launch table of the convolution calls of one profiled step (streams serialised): shape, duration, TFLOP/s
of algorithmic fp32 FLOPs.  Usage: python tools/layer_tf.py [--conv-math f16x2|bf16x3|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench
import rpnet_amd.functional as RF
from rpnet_amd import hip
from rpnet_amd.parallel import FlatGradBucket

if "--conv-math" in sys.argv:
    RF.set_conv_math(sys.argv[sys.argv.index("--conv-math") + 1])
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(bench.ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = 5
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, 8, 256, dev, 1)
for _ in range(3):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
rec = []
orig = hip.call
def timed(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    info = None
    if name in ("rpnet_conv_fwd", "rpnet_conv_wgrad"):
        d = args[0]._obj
        info = (d.N * d.H * d.W, d.C0 + d.C1, d.Co0 + d.Co1, d.taps, d.H, d.upsample, d.split_planes)
    a.record(); orig(name, *args); b.record()
    rec.append((name, info, a, b))
hip.call = timed; RF.call = timed
RF.set_async_wgrad(False)
import rpnet_amd.modules as RM
RM._CRE_STREAMS_TRAIN = False       # one launch at a time on the GPU
RM._ENC_STREAMS = 0
bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()
hip.call = orig; RF.call = orig
agg = {}
for name, info, a, b in rec:
    if info is None: continue
    k = (name,) + info
    e = agg.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += a.elapsed_time(b)
tot = {}
print(f"{'call':18s} {'M':>8s} {'Cin':>5s} {'Cout':>5s} taps  H ups pl   n   ms/call   TF    ms total")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    name, M, ci, co, taps, H, ups, pl = k
    fl = 2.0 * M * ci * co * taps
    print(f"{name:18s} {M:8d} {ci:5d} {co:5d} {taps:4d} {H:3d} {ups:3d} {pl:2d} {n:3d} {ms/n:9.3f} {fl*n/ms/1e9:6.1f} {ms:9.3f}")
    tot[name] = tot.get(name, 0.0) + ms
print(tot)
