"""What does an HBM-bound pass of the main chain cost while a weight-gradient GEMM runs beside it?  One training step of BASELINE
configs[1] is logged (streams serialised, every operand kept alive); then every logged call of the VICTIMS (BatchNorm backward, glue
backward, correlation backward ...) is repeated on the same operands (a) alone and (b) while the AGGRESSOR — the longest 3x3
weight-gradient launch of the step — loops on a second stream.  The default schedule puts exactly these pairs on the machine together
(functional._WGRAD_DEFER); in the traced step the passes take about twice their time alone (profiles/r06_kernel_stats_async.csv).
Usage: python tools/corun_probe.py [out.txt]     env: VICTIMS="rpnet_bn_bwd,rpnet_refine_glue_bwd,..." AGG=<log index> AGG_TUNE=<tune> REPS=6 B SIZE ITERS MATH"""
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
RF._MASK_SKIP = False
B, SIZE = int(os.environ.get("B", "8")), int(os.environ.get("SIZE", "256"))
REPS = int(os.environ.get("REPS", "6"))
VICTIMS = os.environ.get("VICTIMS", "rpnet_bn_bwd,rpnet_bn_relu,rpnet_refine_glue_bwd,rpnet_local_corr_split_bwd,rpnet_bn_stats_from_partial").split(",")
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, B, SIZE, dev)
for _ in range(3):
    bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()

KEEP = []
for _n in ("empty", "zeros", "ones", "full", "empty_like", "zeros_like", "ones_like", "stack", "cat"):
    def _wrap(fn):
        def f(*a, **k):
            o = fn(*a, **k)
            KEEP.append(o)
            return o
        return f
    setattr(torch, _n, _wrap(getattr(torch, _n)))

log = []
orig = RF.call


def spy(name, *args):
    if name in ("rpnet_conv_wgrad",) or name in VICTIMS:
        a2 = list(args)
        desc = None
        if name == "rpnet_conv_wgrad":
            d = args[0]._obj
            desc = type(d)()
            C.memmove(C.addressof(desc), C.addressof(d), C.sizeof(d))
            a2[0] = C.byref(desc)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = orig(name, *args)
        b.record()
        log.append({"name": name, "args": tuple(a2), "desc": desc, "ev": (a, b)})
        return r
    return orig(name, *args)


RF.call = spy
bench.step(net, bucket, inp, cfg["align_loss_scaler"])
torch.cuda.synchronize()
RF.call = orig
for e in log:
    e["insitu"] = e["ev"][0].elapsed_time(e["ev"][1]) * 1e3
SPIN = None
if os.environ.get("AGG", "").startswith("spin"):      # AGG=spin:<blocks>[:<lds bytes>]: the MFMA spinner instead of a weight gradient
    f = os.environ.pop("AGG").split(":")
    SPIN = (int(f[1]) if len(f) > 1 else 256, int(f[2]) if len(f) > 2 else 144 * 1024)
    spin_out = torch.zeros(4, device=dev, dtype=torch.int64)
    spin_log = []
wg = [i for i, e in enumerate(log) if e["name"] == "rpnet_conv_wgrad" and e["args"][1] is not None and e["desc"].taps == 9]
agg = int(os.environ["AGG"]) if os.environ.get("AGG") else max(wg, key=lambda i: log[i]["insitu"])
A = log[agg]
d = A["desc"]
if os.environ.get("AGG_TUNE"):      # another kernel form / an ablation of the aggressor (rpnet_conv_desc.tune; csrc/conv_wgrad_ring.hip)
    d.tune = int(os.environ["AGG_TUNE"], 0)
if SPIN:
    print(f"# tools/corun_probe.py: B={B} SIZE={SIZE} T={cfg['n_iter_refinement']} {RF.conv_math()}; aggressor = MFMA spinner, {SPIN[0]} blocks of 4 waves, "
          f"{SPIN[1]} bytes of LDS each (rpnet_debug_mfma_spin); {REPS} repeats per victim call", file=out, flush=True)
else:
    print(f"# tools/corun_probe.py: B={B} SIZE={SIZE} T={cfg['n_iter_refinement']} {RF.conv_math()}; aggressor = weight gradient #{agg}: "
          f"{d.N}x{d.H}x{d.W}, {d.C0 + d.C1} -> {d.Co0 + d.Co1} channels, tune {d.tune:#x}, {A['insitu']:.0f} us in the serialised step; {REPS} repeats per victim call",
          file=out, flush=True)
side, main = torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)
if SPIN:
    for nb in (1, 64, 128, 256):
        for _ in range(2):
            orig("rpnet_debug_mfma_spin", nb, SPIN[1], 200000, RF.ptr(spin_out))      # 2 ms
        torch.cuda.synchronize()
        c, t, n = (int(v) for v in spin_out[:3].tolist())
        print(f"# spinner alone, {nb:3d} blocks, 2 ms: shader clock {c / t * 100.0:.0f} MHz, MFMA issue {n * 32.0 / max(c, 1):.2f} of back-to-back", file=out, flush=True)


def timed(e, corun):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if corun and SPIN:
        started = torch.cuda.Event()
        with torch.cuda.stream(side):
            orig("rpnet_debug_mfma_spin", SPIN[0], SPIN[1], 300, RF.ptr(spin_out))          # 3 us: the kernel is loaded and the stream busy
            started.record(side)
            orig("rpnet_debug_mfma_spin", SPIN[0], SPIN[1], int(REPS * 6 * max(e["insitu"], 20.0) * 100), RF.ptr(spin_out))
        main.wait_event(started)
    elif corun:
        started = torch.cuda.Event()
        with torch.cuda.stream(side):
            orig(A["name"], *A["args"])
            started.record(side)
            k = max(3, int(REPS * 2.5 * max(e["insitu"], 20.0) / max(A["insitu"], 1.0)) + 2)
            for _ in range(k):
                orig(A["name"], *A["args"])
        main.wait_event(started)
    a.record()
    for _ in range(REPS):
        orig(e["name"], *e["args"])
    b.record()
    torch.cuda.synchronize()
    if corun and SPIN:
        c, t, n = (int(v) for v in spin_out[:3].tolist())
        spin_log.append((c / t * 100.0, n * 32.0 / max(c, 1)))      # shader MHz; MFMA-busy fraction at 32 cycles per 32x32x16
    return a.elapsed_time(b) * 1e3 / REPS


tot = {}
print(f"{'#':>4s} {'call':32s} {'in step us':>10s} {'alone us':>9s} {'beside us':>10s} {'ratio':>6s}", file=out)
for i, e in enumerate(log):
    if e["name"] not in VICTIMS:
        continue
    timed(e, False)
    al = timed(e, False)
    co = timed(e, True)
    t = tot.setdefault(e["name"], [0, 0.0, 0.0])
    t[0] += 1; t[1] += al; t[2] += co
    print(f"{i:4d} {e['name']:32s} {e['insitu']:10.1f} {al:9.1f} {co:10.1f} {co / al:6.2f}", file=out, flush=True)
for n, (k, al, co) in tot.items():
    print(f"# {n}: {k} calls, alone {al / 1e3:.3f} ms, beside the weight gradient {co / 1e3:.3f} ms ({co / al:.2f} x)", file=out)
if SPIN and spin_log:
    mhz = sorted(x[0] for x in spin_log)
    busy = sorted(x[1] for x in spin_log)
    print(f"# the spinner's shader clock while the victims ran: median {mhz[len(mhz) // 2]:.0f} MHz (min {mhz[0]:.0f}, max {mhz[-1]:.0f}); "
          f"its waves' MFMA issue: median {busy[len(busy) // 2]:.2f} of back-to-back", file=out)
