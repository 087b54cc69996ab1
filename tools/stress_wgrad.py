"""Race screen of the LDS-DMA weight-gradient kernel: the layer shapes of a 64x64 episode (image rows of 4 .. 64 pixels, 2-4
images: the shapes of tests/test_gpu_dist.py) over and over, each result compared BIT FOR BIT with the register-staged kernel's,
optionally while a second process keeps the GPU busy (python tools/stress_wgrad.py load &).  Prints the mismatch count."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rpnet_amd.functional as RF
from rpnet_amd.functional import PackedWeight, _desc, split_f16, _ws
from rpnet_amd.hip import call, ptr, query

dev = "cuda:0"
if len(sys.argv) > 1 and sys.argv[1] == "load":      # background load: big GEMM-ish work on the same GPU
    a = torch.randn(8192, 8192, device=dev)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
        (a @ a).sum().item()
    sys.exit(0)


def planes(x):
    s = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(x.abs().max())).item()) - 15)], device=dev)
    return split_f16(x, s, planes=2)


def wgrad(xs, sx, dys, sdy, dy, N, H, W, ci, co, tune):
    dw = torch.empty(co, ci, 3, 3, device=dev)
    wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, ci, co, 9)
    ws = _ws(wb, dy)
    d = _desc(xs, None, None, None, None, 0, dy, None, N, H, W, 9, 0)
    d.split_planes, d.tune = 2, tune
    d.acc_scale_x, d.acc_scale_dy = sx.data_ptr(), sdy.data_ptr()
    call("rpnet_conv_wgrad", C.byref(d), ptr(dys), ptr(dw), ci, 0, ci, ci, ptr(ws), wb)
    return dw


SHAPES = [(4, 64, 64, 64, 64), (4, 32, 32, 64, 128), (4, 32, 32, 128, 128), (4, 16, 16, 128, 256), (4, 16, 16, 256, 256),
          (4, 8, 8, 256, 512), (4, 8, 8, 512, 512), (4, 4, 4, 512, 1024), (4, 4, 4, 1024, 1024), (4, 8, 8, 1024, 512),
          (4, 16, 16, 512, 256), (2, 16, 16, 256, 256)]
torch.manual_seed(0)
data = []
for (N, H, W, ci, co) in SHAPES:
    x, dy = torch.randn(N, H, W, ci, device=dev), torch.randn(N, H, W, co, device=dev)
    (xs, sx), (dys, sdy) = planes(x), planes(dy)
    ref = wgrad(xs, sx, dys, sdy, dy, N, H, W, ci, co, 8)
    data.append((xs, sx, dys, sdy, dy, N, H, W, ci, co, ref))
torch.cuda.synchronize()
iters = int(os.environ.get("ITERS", "200"))
bad = {}
for it in range(iters):
    outs = [wgrad(*dd[:10], 0) for dd in data]
    torch.cuda.synchronize()
    for dd, o in zip(data, outs):
        if not torch.equal(o, dd[10]):
            k = dd[5:10]
            bad[k] = bad.get(k, 0) + 1
print("iterations", iters, "mismatches by shape (N, H, W, cin, cout):", bad if bad else "none")
