"""Micro-benchmark of the implicit-GEMM conv (rpnet_conv_fwd) and wgrad on the layer shapes of
the 256x256, batch-8 step.  Prints TFLOP/s per shape (HIP events, 20 launches)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpnet_amd import hip
from rpnet_amd.functional import PackedWeight, _desc, _ws
from rpnet_amd.hip import call, ptr, query

SHAPES = [  # N, H, W, Cin, Cout
    (16, 256, 256, 64, 64), (16, 128, 128, 128, 128), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512),
    (16, 16, 16, 1024, 1024), (16, 32, 32, 1024, 512), (8, 64, 64, 256, 256),
]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for (N, H, W, ci, co) in SHAPES:
    zero = os.environ.get("ZERO") == "1"
    x = torch.zeros(N, H, W, ci, device=dev) if zero else torch.randn(N, H, W, ci, device=dev)
    w = torch.zeros(co, ci, 3, 3, device=dev) if zero else torch.randn(co, ci, 3, 3, device=dev) * 0.05
    pw = PackedWeight(w)
    y = torch.empty(N, H, W, co, device=dev)
    dy = torch.zeros(N, H, W, co, device=dev) if zero else torch.randn(N, H, W, co, device=dev)
    dw = torch.empty_like(w)
    d = _desc(x, None, pw.wp, None, None, 0, y, None, N, H, W, 9, 0)
    wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, ci, co, 9)
    ws = _ws(wb, x)
    fl = 2.0 * N * H * W * ci * co * 9
    def run(fn, n=20):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    line = f"M={N*H*W:8d} {ci:4d}->{co:4d}"
    if which in ("both", "fwd"):
        try:
            t = run(lambda: call("rpnet_conv_fwd", C.byref(d)))
            line += f"  fwd {t:7.3f} ms {fl/t/1e9:6.1f} TF"
        except RuntimeError as e:
            line += f"  fwd n/a ({str(e)[-40:]})"
    if which in ("both", "wgrad"):
        t = run(lambda: call("rpnet_conv_wgrad", C.byref(d), ptr(dy), ptr(dw), ci, 0, ci, ci, ptr(ws), wb))
        line += f"  wgrad {t:7.3f} ms {fl/t/1e9:6.1f} TF"
    print(line, flush=True)
