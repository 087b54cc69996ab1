"""Accuracy and speed of the split implicit GEMM (conv_split.hip: planes=3 bf16, planes=2 scaled fp16) against the fp32-MFMA kernel.
Accuracy: max |err| / max |ref| against an fp64 CPU convolution on a small shape; speed: TFLOP/s
(algorithmic fp32 FLOPs) on the layer shapes of the 256x256, batch-8 step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rpnet_amd.functional import PackedWeight, _desc, split_bf16, split_f16
from rpnet_amd.hip import call

dev = "cuda:0"
import rpnet_amd.functional as _RF
if os.environ.get("TILE"):          # force tile variant TILE of the split forward kernels (rpnet_conv_desc.tune)
    _RF.TUNE["tile"] = int(os.environ["TILE"]) + 1 + 256 * int(os.environ.get("DBG", "0"))   # DBG: ablation bits of variant 11
PLANES = tuple(int(v) for v in os.environ.get("PLANES", "0,3,2").split(","))


def planes_of(x, planes):
    """-> (planes tensor, tensor scale or None): fp16 planes take a power-of-two scale with max|x| / s <= 2^15 — here
    from the data (a micro-benchmark); the model takes it from the BatchNorm bound instead"""
    if planes == 3:
        return split_bf16(x, 3), None
    s_in = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(x.abs().max())).item()) - 15)], device=dev)
    return split_f16(x, s_in, planes=planes)


def conv(x, pw, co, planes, xs=None):
    N, H, W, ci = x.shape
    y = torch.empty(N, H, W, co, device=dev)
    if planes:
        pk = pw.split_packs(planes)
        xs = planes_of(x, planes) if xs is None else xs
        d = _desc(xs[0], None, pk[0], None, None, 0, y, None, N, H, W, 9, 0)
        d.split_planes = planes
        if planes <= 2:
            d.acc_scale_col, d.acc_scale_x = pk[2].data_ptr(), xs[1].data_ptr()
        d._keep = (xs, pk)
    else:
        d = _desc(x, None, pw.wp, None, None, 0, y, None, N, H, W, 9, 0)
    call("rpnet_conv_fwd", C.byref(d))
    return y, d


torch.manual_seed(0)
FWD_ONLY = os.environ.get("FWD_ONLY") == "1"      # only the forward speed table
WG_PLANES = (2,)                                  # WG_ONLY=1: only the f16x2 weight-gradient speed table
RELU = os.environ.get("RELU") == "1"              # half of the activations exact zeros, as behind a ReLU (the clock the chip holds depends on the data)
ZERO = os.environ.get("ZERO", "")                 # "x": all-zero activations, "xw": all-zero activations and weights (no toggling in the matrix pipe: the clock the loop COULD hold)
WG_ONLY = os.environ.get("WG_ONLY") == "1"
for (N, H, W, ci, co) in ([] if (FWD_ONLY or WG_ONLY) else [(2, 32, 32, 256, 256), (1, 16, 16, 1024, 128)]):
    x = torch.randn(N, H, W, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    pw = PackedWeight(w)
    for planes in (0, 3, 2):
        y, _ = conv(x, pw, co, planes)
        err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        print(f"{N}x{H}x{W} {ci}->{co} planes={planes}: max err / max|ref| = {err:.3e}", flush=True)
    xs = split_bf16(x, 3)
    print("split exact:", (xs.float().sum(0) - x).abs().max().item())

SHAPES = [(16, 256, 256, 64, 64), (16, 128, 128, 128, 128), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512),
          (16, 16, 16, 1024, 1024), (16, 32, 32, 1024, 512), (8, 64, 64, 256, 256)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for (N, H, W, ci, co) in ([] if WG_ONLY else SHAPES):
    x = torch.randn(N, H, W, ci, device=dev)
    if RELU:
        x = torch.relu(x)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    if "x" in ZERO:
        x = x * 0 + 1e-30        # (planes of an all-zero tensor; the scale search needs a nonzero maximum)
    if "w" in ZERO:
        w = w * 0
    pw = PackedWeight(w)
    fl = 2.0 * N * H * W * ci * co * 9
    line = f"M={N*H*W:8d} {ci:4d}->{co:4d}"
    for planes in PLANES:
        xs = planes_of(x, planes) if planes else None
        _, d = conv(x, pw, co, planes, xs)
        for _ in range(3):
            call("rpnet_conv_fwd", C.byref(d))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call("rpnet_conv_fwd", C.byref(d))
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20
        line += f"  p{planes} {t:6.3f} ms {fl/t/1e9:6.1f} TF"
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        split_bf16(x, 3)
    b.record(); torch.cuda.synchronize()
    line += f"  split3 {a.elapsed_time(b)/10:6.3f} ms"
    print(line, flush=True)

if FWD_ONLY:
    sys.exit(0)
# ---- weight gradient: accuracy (fp64 CPU reference) and speed
from rpnet_amd.functional import _ws
from rpnet_amd.hip import ptr, query


def wgrad(x, dy, pw, planes):
    N, H, W, ci = x.shape
    co = dy.shape[-1]
    dw = torch.empty(co, ci, 3, 3, device=dev)
    wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, ci, co, 9)
    ws = _ws(wb, x)
    if planes:
        (xs, sx), (dys, sdy) = planes_of(x, planes), planes_of(dy, planes)
        d = _desc(xs, None, pw.wp, None, None, 0, dy, None, N, H, W, 9, 0)
        d.split_planes = planes
        if planes == 2:
            d.acc_scale_x, d.acc_scale_dy = sx.data_ptr(), sdy.data_ptr()
        d.tune = (int(os.environ.get("WG_ABL", "0")) << 8) | int(os.environ.get("WG_TUNE", "0"))   # WG_TUNE=16: round 5's row-major K order      # ablation forms of conv_wgrad9_dma_kernel (tools/wgrad_anatomy.sh)
        # WG_GEMM_ONLY=1: the split-K GEMM launch alone (dw == NULL: the two-phase form without its reduce)
        args = (C.byref(d), ptr(dys), None if os.environ.get("WG_GEMM_ONLY") == "1" else ptr(dw), ci, 0, ci, ci, ptr(ws), wb)
        keep = (xs, dys, sx, sdy)
    else:
        d = _desc(x, None, pw.wp, None, None, 0, dy, None, N, H, W, 9, 0)
        args = (C.byref(d), ptr(dy), ptr(dw), ci, 0, ci, ci, ptr(ws), wb)
        keep = ()
    call("rpnet_conv_wgrad", *args)
    return dw, args, (d, ws, keep)


for (N, H, W, ci, co) in ([] if WG_ONLY else [(2, 32, 32, 64, 128), (3, 8, 8, 128, 64), (2, 12, 20, 64, 64)]):
    x = torch.randn(N, H, W, ci, device=dev)
    dy = torch.randn(N, H, W, co, device=dev)
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w0, padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = w0.grad
    pw = PackedWeight(torch.zeros(co, ci, 3, 3, device=dev))
    for planes in (0, 3, 2):
        dw, _, _k = wgrad(x, dy, pw, planes)
        err = (dw.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        print(f"wgrad {N}x{H}x{W} {ci}->{co} planes={planes}: max err / max|ref| = {err:.3e}", flush=True)

for (N, H, W, ci, co) in SHAPES:
    x = torch.randn(N, H, W, ci, device=dev)
    dy = torch.randn(N, H, W, co, device=dev)
    pw = PackedWeight(torch.zeros(co, ci, 3, 3, device=dev))
    if "x" in ZERO:
        x = x * 0 + 1e-30
    if "w" in ZERO:      # (here: the other operand, the output gradient)
        dy = dy * 0 + 1e-30
    fl = 2.0 * N * H * W * ci * co * 9
    line = f"wgrad M={N*H*W:8d} {ci:4d}->{co:4d}"
    for planes in (WG_PLANES if os.environ.get("WG_ONLY") else (0, 3, 2)):
        _, args, _k = wgrad(x, dy, pw, planes)
        for _ in range(3):
            call("rpnet_conv_wgrad", *args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call("rpnet_conv_wgrad", *args)
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20
        line += f"  p{planes} {t:6.3f} ms {fl/t/1e9:6.1f} TF"
    print(line, flush=True)
